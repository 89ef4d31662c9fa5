import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dig_amd import ops
dt = torch.bfloat16
x = torch.randn(32, 32, 32, 512, device="cuda").to(dt)
cw = ops.ConvWeight(torch.randn(512, 512, 3, 3, device="cuda") * 0.02, torch.zeros(512, device="cuda"), dt)
for _ in range(4):
    ops.conv3x3(x, cw)
torch.cuda.synchronize()
