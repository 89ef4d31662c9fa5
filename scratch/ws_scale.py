import sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import ops
dt = torch.bfloat16
DEV = 'cuda'
torch.manual_seed(0)
w = (torch.randn(64, 64, 3, 3) * 0.05).to(dt).float()
b = torch.randn(64) * 0.1
cw = ops.ConvWeight(w.to(DEV), b.to(DEV), dt)
for (B, H, W) in [(4, 128, 128), (8, 256, 256), (32, 256, 256)]:
    x = torch.randn(B, 64, H, W).to(dt)
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    for rep in range(3):
        y, st = ops.conv3x3(xd, cw)
    torch.cuda.synchronize()
    ref = F.conv2d(x.float().to(DEV), w.to(DEV), b.to(DEV), padding=1)
    err = (y.float().permute(0, 3, 1, 2) - ref).abs().max().item()
    s = st.double().sum(2)
    g = ref.reshape(B, 8, -1).double()
    e2 = ((s[..., 0] - g.sum(-1)).abs().max() / g.shape[-1] ** 0.5).item()
    print(B, H, W, "max err", err, "stat err", e2, flush=True)
    assert err < 0.05
