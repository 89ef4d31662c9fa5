"""Micro-benchmark of the Unet's 1x1-conv GEMM shapes (B=32, 256x256, bf16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import ops
dev = torch.device("cuda", 0)
bf = torch.bfloat16
torch.manual_seed(0)
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).to(bf)
def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cases = []
def tail(M, K0, K1, N, rows):
    x0, x1 = rnd(M, K0), (rnd(M, K1) if K1 else None)
    w, b = rnd(N, K0 + K1), torch.randn(N, device=dev)
    g, aff = rnd(M, N), torch.randn(M // rows, N, 2, device=dev)
    return lambda: ops.linear_rows(x0, w, bias=b, x1=x1, gsrc=g, gaff=aff, rows_per_img=rows)
def down(B, H, C, N):
    x, w, b = rnd(B, H, H, C), rnd(N, 4 * C), torch.randn(N, device=dev)
    return lambda: ops.downsample_conv(x, w, b)
def lin(M, K, N, res=False, bias=True):
    x, w = rnd(M, K), rnd(N, K)
    b = torch.randn(N, device=dev) if bias else None
    r = rnd(M, N) if res else None
    return lambda: ops.linear_rows(x, w, bias=b, res=r)
cases = [
    ("tail L2up 131072x384->256", tail(131072, 256, 128, 256, 4096), 131072 * (384 + 256 + 256) * 2, 2 * 131072 * 384 * 256),
    ("tail L3up 32768x768->512", tail(32768, 512, 256, 512, 1024), 32768 * (768 + 512 + 512) * 2, 2 * 32768 * 768 * 512),
    ("down L0 524288x256->64", down(32, 256, 64, 64), 524288 * (256 + 64) * 2, 2 * 524288 * 256 * 64),
    ("down L1 131072x256->128", down(32, 128, 64, 128), 131072 * (256 + 128) * 2, 2 * 131072 * 256 * 128),
    ("down L2 32768x512->256", down(32, 64, 128, 256), 32768 * (512 + 256) * 2, 2 * 32768 * 512 * 256),
    ("qkv 32768x512->384", lin(32768, 512, 384, bias=False), 32768 * (512 + 384) * 2, 2 * 32768 * 512 * 384),
    ("out 32768x128->512 +res", lin(32768, 128, 512, res=True), 32768 * (128 + 1024) * 2, 2 * 32768 * 128 * 512),
    ("mid 32768x512->512", lin(32768, 512, 512), 32768 * 1024 * 2, 2 * 32768 * 512 * 512),
]
for name, fn, by, fl in cases:
    us = timeit(fn)
    print(f"{name:32s} {us:8.1f} us  {by / us / 1e6:7.2f} TB/s  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
