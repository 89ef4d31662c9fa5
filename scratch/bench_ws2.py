import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import ops
dt = torch.bfloat16
DEV = 'cuda'
B, H, W = 32, 256, 256
x = torch.randn(B, H, W, 64, device=DEV).to(dt)
cw = ops.ConvWeight(torch.randn(64, 64, 3, 3, device=DEV) * 0.05, torch.zeros(64, device=DEV), dt)
aff = torch.ones(B, 64, 2, device=DEV)
def run(tag, pro=False, iters=10):
    a = aff if pro else None
    for _ in range(2): ops.conv3x3(x, cw, aff=a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.conv3x3(x, cw, aff=a)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tag = os.environ.get("DIG_HIP_LIB", "default").split("lib_")[-1]
print(f"{tag:14s} skip={os.environ.get('DIG_WS_SKIP','0'):3s} plain {run('a'):7.1f} us   pro {run('b', True):7.1f} us", flush=True)
