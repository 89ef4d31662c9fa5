import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from dig_amd import ops, synth
dt = torch.bfloat16
B, H, W = 2, 32, 48
g = torch.Generator().manual_seed(3)
x0 = torch.randn(B, 64, H, W, generator=g).to(dt).float(); x1 = torch.randn(B, 64, H, W, generator=g).to(dt).float()
w = (torch.randn(64, 128, 3, 3, generator=g) * 0.03).to(dt).float(); b = torch.randn(64, generator=g) * 0.1
cw = ops.ConvWeight(w.cuda(), b.cuda(), dt, C1=64)
ref = F.conv2d(torch.cat([x0, x1], 1), w, b, padding=1)
nh = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda().to(dt)
y, st = ops.conv3x3(nh(x0), cw, x1=nh(x1))
torch.cuda.synchronize()
print("y err", float((y.float().permute(0, 3, 1, 2).cpu() - ref).abs().max()))
yb = y.float().cpu().reshape(B, H // 16, 16, W // 16, 16, 8, 8)  # b, ty, y, tx, x, g, c
ts = yb.sum((2, 4, 6)).permute(0, 3, 1, 2).reshape(B, 8, -1)  # b,g,tile
tq = (yb * yb).sum((2, 4, 6)).permute(0, 3, 1, 2).reshape(B, 8, -1)
got = st.cpu()
print("stats shape", tuple(got.shape))
ds = (got[..., 0] - ts).abs(); dq = (got[..., 1] - tq).abs()
print("per-tile sum err max", float(ds.max()), "sumsq err max", float(dq.max()), "ref sumsq typical", float(tq.mean()))
bad = (dq > 1e-2 * tq.abs() + 1e-2).nonzero()
print("bad entries", len(bad), bad[:12].tolist())
if len(bad):
    i = bad[0].tolist()
    print("got", got[i[0], i[1], i[2]].tolist(), "want", float(ts[i[0], i[1], i[2]]), float(tq[i[0], i[1], i[2]]))
