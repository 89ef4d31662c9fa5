"""gemm_nt micro-benchmark on the Unet's large 1x1-conv shapes at B = 32, 256x256 (bf16; HIP-event time per launch against the
algorithmic HBM floor).  Library: DIG_HIP_LIB (tools/lab/build_variant.sh).  python tools/lab/bench_gemm.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dig_amd import ops
torch.set_grad_enabled(False)
dt = torch.bfloat16
B = 32
def rnd(*s):
    return torch.randn(*s, device="cuda").to(dt)
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot = 0.0
# ResnetBlock tails with a concat shortcut: (H, K0, K1, N)
for H, K0, K1, N in ((64, 256, 128, 256), (32, 512, 256, 512)):
    M = B * H * H
    x0, x1, g = rnd(B, H * H, K0), rnd(B, H * H, K1), rnd(B, H * H, N)
    w = (torch.randn(N, K0 + K1, device="cuda") * 0.05).to(dt); b = torch.zeros(N, device="cuda"); aff = torch.ones(B, N, 2, device="cuda")
    us = timeit(lambda: ops.linear_rows(x0, w, bias=b, x1=x1, gsrc=g, gaff=aff, rows_per_img=H * H))
    mb = M * (K0 + K1 + 2 * N) * 2 / 1e6
    print(f"res_tail  M={M:7d} K={K0}+{K1} N={N}: {us:6.1f} us  {mb / us:5.2f} TB/s  (x2 per step)")
    tot += 2 * us
# Downsample: unshuffle + 1x1 (H of the input, C, N)
for H, C, N in ((256, 64, 64), (128, 64, 128), (64, 128, 256)):
    x = rnd(B, H, H, C); w = (torch.randn(N, 4 * C, device="cuda") * 0.05).to(dt); b = torch.zeros(N, device="cuda")
    us = timeit(lambda: ops.downsample_conv(x, w, b))
    M = B * H * H // 4
    mb = (B * H * H * C + M * N) * 2 / 1e6
    print(f"downsample M={M:7d} K={4 * C} N={N}: {us:6.1f} us  {mb / us:5.2f} TB/s")
    tot += us
# attention-side projections at 32x32 (plain, + residual)
for K, N, res in ((256, 256, False), (512, 256, False), (128, 256, True), (128, 512, True)):
    M = B * 1024
    x = rnd(M, K); w = (torch.randn(N, K, device="cuda") * 0.05).to(dt); b = torch.zeros(N, device="cuda"); r = rnd(M, N) if res else None
    us = timeit(lambda: ops.linear_rows(x, w, bias=b, res=r))
    mb = M * (K + N * (2 if res else 1)) * 2 / 1e6
    print(f"proj      M={M:7d} K={K} N={N} res={int(res)}: {us:6.1f} us  {mb / us:5.2f} TB/s")
    tot += us
print(f"sum (step-weighted) {tot:.1f} us  lib={os.environ.get('DIG_HIP_LIB', 'default')}")
