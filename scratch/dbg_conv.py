import sys, torch, torch.nn.functional as F
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle')
from dig_amd import ops, synth
DEV='cuda'
def rnd(tag, shape, scale=1.0): return synth.uniform_pm1(tag, shape) * scale
for dt in (torch.float32, torch.bfloat16):
  for (B,Ci,H,W,Co) in [(1,384,14,13,20),(1,384,14,13,32),(1,384,16,16,20),(1,64,14,13,20),(1,48,14,13,24)]:
    x = rnd("cx",(B,Ci,H,W)).to(dt).float(); w = rnd("cw",(Co,Ci,3,3),(3.0/(9*Ci))**0.5).to(dt).float(); b = rnd("cb",(Co,),0.1)
    ref = F.conv2d(x,w,b,padding=1)
    wp = w.permute(0,2,3,1).contiguous().to(device=DEV,dtype=dt)
    y,_ = ops.conv3x3(x.permute(0,2,3,1).contiguous().to(device=DEV,dtype=dt), wp, b.to(DEV), want_stats=False)
    d = (y.float().permute(0,3,1,2).cpu()-ref).abs()
    bad = (d > (1e-4 if dt==torch.float32 else 3e-2)).nonzero()
    print(dt, (B,Ci,H,W,Co), 'max', float(d.max()), 'nbad', len(bad), bad[:12].tolist())
