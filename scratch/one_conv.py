import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import ops
dt = torch.bfloat16; DEV='cuda'
which = sys.argv[1]
cfg = {'l0': (32,256,256,64,64), 'l3': (32,32,32,512,512), 'l1': (32,128,128,128,128)}[which]
B,H,W,Ci,Co = cfg
x = torch.randn(B,H,W,Ci,device=DEV).to(dt)
cw = ops.ConvWeight(torch.randn(Co,Ci,3,3,device=DEV)*0.05, torch.zeros(Co,device=DEV), dt)
for _ in range(3): ops.conv3x3(x, cw)
torch.cuda.synchronize()
