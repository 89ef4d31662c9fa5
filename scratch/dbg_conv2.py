import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
from dig_amd import ops, synth
DEV='cuda'
def rnd(tag, shape, scale=1.0): return synth.uniform_pm1(tag, shape) * scale
dt=torch.float32
B,Ci,H,W,Co=1,32,16,16,32
x = rnd("cx",(B,Ci,H,W))
b = torch.zeros(Co)
xd = x.permute(0,2,3,1).contiguous().to(device=DEV,dtype=dt)
for tap in range(9):
  for c in range(0,Ci,4):
    w = torch.zeros(Co,Ci,3,3); w[:,c,tap//3,tap%3]=1.0
    ref = F.conv2d(x,w,b,padding=1)
    wp = w.permute(0,2,3,1).contiguous().to(device=DEV,dtype=dt)
    y,_ = ops.conv3x3(xd, wp, b.to(DEV), want_stats=False)
    y2,_ = ops.conv3x3(xd, wp, b.to(DEV), want_stats=False)
    d = (y.float().permute(0,3,1,2).cpu()-ref).abs()
    bad = (d > 1e-5).nonzero()
    if len(bad): print('tap',tap,'c',c,'nbad',len(bad), bad[:4].tolist(), 'got', float(y[0,3,11,0]), 'ref', float(ref[0,0,3,11]), 'determ', bool(torch.equal(y,y2)))
print('done')
