"""pw_res micro-benchmark: python tools/lab/bench_pw.py M K0 K1 N  (bf16; HIP-event time per launch, algorithmic TB/s)"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dig_amd import ops
torch.set_grad_enabled(False)
M, K0, K1, N = (int(v) for v in sys.argv[1:5])
dt = torch.bfloat16
B = 32
x0 = torch.randn(B, M // B, K0, device="cuda").to(dt); x1 = torch.randn(B, M // B, K1, device="cuda").to(dt)
g = torch.randn(B, M // B, N, device="cuda").to(dt)
w = (torch.randn(N, K0 + K1, device="cuda") * 0.05).to(dt); b = torch.zeros(N, device="cuda")
aff = torch.ones(B, N, 2, device="cuda")
f = lambda: ops.res_tail(x0, w, b, x1, g, aff, M // B)
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"M={M} K={K0}+{K1} N={N} iters={os.environ.get('DIG_PW_ITERS','-')}: {us:.1f} us, {M * (K0 + K1 + 2 * N) * 2 / us / 1e6:.2f} TB/s")
