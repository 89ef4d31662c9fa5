import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import ops
dt = torch.bfloat16
DEV = 'cuda'
B, H, W = 32, 256, 256
x0 = torch.randn(B, H, W, 64, device=DEV).to(dt); x1 = torch.randn(B, H, W, 64, device=DEV).to(dt)
xu = torch.randn(B, H // 2, W // 2, 128, device=DEV).to(dt)
wt = torch.randn(64, 128, 3, 3, device=DEV) * 0.03
cw = ops.ConvWeight(wt, torch.zeros(64, device=DEV), dt, C1=64)
cwu = ops.ConvWeight(wt, torch.zeros(64, device=DEV), dt)
def run(f, iters=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
tag = os.environ.get("DIG_HIP_LIB", "default").split("lib_")[-1][-12:]
for rep in range(2):
    for env in ("", "1"):
        if env: os.environ["DIG_NO_RS128"] = "1"
        else: os.environ.pop("DIG_NO_RS128", None)
        a = run(lambda: ops.conv3x3(x0, cw, x1=x1)); b = run(lambda: ops.conv3x3(xu, cwu, up2=True, want_stats=False))
        print(f"{tag} {'generic' if env else 'rs128  '} concat {a:7.1f} us   up2 {b:7.1f} us", flush=True)
