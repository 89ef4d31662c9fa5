import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dig_amd import ops, _lib
dt = torch.bfloat16
DEV = 'cuda'
B, H, W = 32, 256, 256
x = torch.randn(B, H, W, 64, device=DEV).to(dt)
cw = ops.ConvWeight(torch.randn(64, 64, 3, 3, device=DEV) * 0.05, torch.zeros(64, device=DEV), dt)
aff = torch.ones(B, 64, 2, device=DEV) if len(sys.argv) > 1 else None
for _ in range(3): ops.conv3x3(x, cw, aff=aff)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 64 * 8))()
L = ctypes.CDLL(_lib.LIB_PATH)
print("rc", L.dig_debug_ws_trace(buf))
t0 = min(buf[(w * 64 + 0) * 8] for w in range(8))
for ph in range(0, 12):
    row = []
    for w in (0, 4):
        a, b, c, s3, s4, s5 = (buf[(w * 64 + ph) * 8 + i] - t0 for i in range(6))
        stage = (ph - 1 - (w >> 2)) & 1
        row.append(f"w{w}: {a:7d} +{b-a:6d} +{c-b:6d}" + (f" [gl {s3-a:5d} E1 {s4-s3:5d} E2 {s5-s4:5d} ls {b-s5:5d}]" if stage else " " * 41))
    print(f"ph{ph-1:3d} ", " | ".join(row))

comp, stg = [], []
for ph in range(3, 12):
    for w in (0, 4):
        a, b = buf[(w * 64 + ph) * 8 + 0], buf[(w * 64 + ph) * 8 + 1]
        (stg if (ph - 1 - (w >> 2)) & 1 else comp).append(b - a)
print("SUMMARY %s compute %.0f stage %.0f" % (os.environ.get("DIG_HIP_LIB", "default").split("lib_")[-1], sum(comp) / len(comp), sum(stg) / len(stg)))
blk = (ctypes.c_ulonglong * 512)()
print("rc", L.dig_debug_ws_blk(blk))
cyc = [blk[2 * i] for i in range(256)]
rt0 = min(blk[2 * i + 1] for i in range(256))
print("block cycles: min %d median %d max %d" % (min(cyc), sorted(cyc)[128], max(cyc)))
print("entry skew (100MHz ticks): max", max(blk[2 * i + 1] - rt0 for i in range(256)))
for i in range(0, 256, 8):
    print(i, " ".join("%6d" % c for c in cyc[i:i + 8]))
