"""dig_vit_attn at the VOC token count: key-tiled matrix-core kernel vs the VALU kernel (DIG_VIT_ATTN_NO_MFMA)."""
import os, sys, torch
sys.path.insert(0, ".")
from dig_amd._lib import call
for B, N in ((1, 733), (8, 733), (1, 864), (32, 216)):
    heads, dh = 6, 64
    D = heads * dh
    qkv = torch.randn(B, N, 3 * D, device="cuda")
    out = torch.empty(B, N, D, device="cuda")
    acc = torch.zeros(B, N, N, device="cuda")
    for hook in ("", "DIG_VIT_ATTN_TILED", "DIG_VIT_ATTN_NO_MFMA"):
        if hook:
            os.environ[hook] = "1"
        f = lambda: call("vit_attn", qkv=qkv, out=out, acc_all=acc, acc_cls=None, B=B, N=N, heads=heads, dh=dh, ncls=0)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        print(f"B={B} N={N} {hook or 'default':>22}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
        if hook:
            del os.environ[hook]
