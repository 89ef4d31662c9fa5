import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import ops
dt = torch.bfloat16
DEV = 'cuda'
def run(B, H, W, Ci, Co, pro=False, C1=0, iters=10):
    x = torch.randn(B, H, W, Ci, device=DEV).to(dt)
    x1 = torch.randn(B, H, W, C1, device=DEV).to(dt) if C1 else None
    cw = ops.ConvWeight(torch.randn(Co, Ci + C1, 3, 3, device=DEV) * 0.05, torch.zeros(Co, device=DEV), dt, C1=C1)
    aff = torch.ones(B, Ci, 2, device=DEV) if pro else None
    for _ in range(3): ops.conv3x3(x, cw, x1=x1, aff=aff)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): ops.conv3x3(x, cw, x1=x1, aff=aff)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * B * H * W * Co * (Ci + C1) * 9
    print(f"{os.environ.get('DIG_HIP_LIB','default')[-14:]:>14} B{B} {H}x{W} {Ci}+{C1}->{Co} pro={int(pro)}: {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF")
for args in [(32,128,128,128,128,False),(32,64,64,256,256,False),(32,32,32,512,512,False),(32,32,32,512,512,True),(32,32,32,768,512,False)]:
    run(*args)
