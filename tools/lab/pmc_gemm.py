import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dig_amd import ops
dt = torch.bfloat16
B = 32
H, K0, K1, N = (32, 512, 256, 512) if len(sys.argv) < 2 or sys.argv[1] == "a" else (64, 256, 128, 256)
x0 = torch.randn(B, H * H, K0, device="cuda").to(dt); x1 = torch.randn(B, H * H, K1, device="cuda").to(dt); g = torch.randn(B, H * H, N, device="cuda").to(dt)
w = (torch.randn(N, K0 + K1, device="cuda") * 0.05).to(dt); b = torch.zeros(N, device="cuda"); aff = torch.ones(B, N, 2, device="cuda")
for _ in range(4):
    ops.linear_rows(x0, w, bias=b, x1=x1, gsrc=g, gaff=aff, rows_per_img=H * H)
torch.cuda.synchronize()
