import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import ops
dt = torch.bfloat16
DEV = 'cuda'
def run(B, H, W, Ci, Co, C1=0, up2=False, iters=10):
    h, w = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(B, h, w, Ci, device=DEV).to(dt)
    x1 = torch.randn(B, h, w, C1, device=DEV).to(dt) if C1 else None
    cw = ops.ConvWeight(torch.randn(Co, Ci + C1, 3, 3, device=DEV) * 0.05, torch.zeros(Co, device=DEV), dt, C1=C1)
    f = lambda: ops.conv3x3(x, cw, x1=x1, up2=up2)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * H * W * Co * (Ci + C1) * 9
    print(f"{os.environ.get('DIG_HIP_LIB','default')[-14:]:>14} B{B} {H}x{W} {Ci}+{C1}->{Co} up2={int(up2)}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF", flush=True)
run(32, 256, 256, 64, 64, C1=64)
run(32, 256, 256, 128, 64, up2=True)
run(32, 128, 128, 64, 64, C1=64)
