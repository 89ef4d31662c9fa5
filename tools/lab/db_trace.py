import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dig_amd import ops, _lib
dt = torch.bfloat16
DEV = 'cuda'
B, H, W, C = 32, 32, 32, 512
x = torch.randn(B, H, W, C, device=DEV).to(dt)
cw = ops.ConvWeight(torch.randn(C, C, 3, 3, device=DEV) * 0.02, torch.zeros(C, device=DEV), dt)
for _ in range(3): ops.conv3x3(x, cw)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 64 * 8))()
L = ctypes.CDLL(_lib.LIB_PATH)
print("rc", L.dig_debug_db_trace(buf))
t0 = buf[0]
for ci in range(8, 16):
    row = []
    for w in (0, 5):
        a = [buf[(w * 64 + ci) * 8 + i] - t0 for i in range(5)]
        row.append(f"w{w}: {a[0]:7d} issue {a[1]-a[0]:5d} mfma {a[2]-a[1]:5d} lstore {a[3]-a[2]:5d} barrier {a[4]-a[3]:5d}")
    print(f"ci{ci:3d} ", " | ".join(row))
