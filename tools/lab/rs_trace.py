import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dig_amd import ops, _lib
dt = torch.bfloat16
B, H, W = 32, 256, 256
x0 = torch.randn(B, H, W, 64, device='cuda').to(dt); x1 = torch.randn(B, H, W, 64, device='cuda').to(dt)
cw = ops.ConvWeight(torch.randn(64, 128, 3, 3, device='cuda') * 0.03, torch.zeros(64, device='cuda'), dt, C1=64)
for _ in range(3): ops.conv3x3(x0, cw, x1=x1)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4 * 64 * 4))(); blk = (ctypes.c_ulonglong * 512)()
L = ctypes.CDLL(_lib.LIB_PATH)
print("rc", L.dig_debug_rs_trace(buf, blk))
tag = os.environ.get("DIG_HIP_LIB", "default").split("lib_")[-1]
t0 = buf[0]
for c in range(4, 12):
    row = []
    for w in (0, 3):
        a, b, d, e = (buf[(w * 64 + c) * 4 + i] for i in range(4))
        row.append(f"w{w}: top {a - t0:8d} wait+bar {b - a:6d} compute {d - b:6d}" + (f" epi {e - d:6d}" if c & 1 else ""))
    print(f"chunk {c:2d} ", " | ".join(row))
comp = [buf[(w * 64 + c) * 4 + 2] - buf[(w * 64 + c) * 4 + 1] for w in range(4) for c in range(4, 60)]
wait = [buf[(w * 64 + c) * 4 + 1] - buf[(w * 64 + c) * 4 + 0] for w in range(4) for c in range(4, 60)]
epi = [buf[(w * 64 + c) * 4 + 3] - buf[(w * 64 + c) * 4 + 2] for w in range(4) for c in range(5, 60, 2)]
cyc = [blk[2 * i] for i in range(256)]; rt = [blk[2 * i + 1] for i in range(256)]
print(f"SUMMARY {tag}: compute {sum(comp)/len(comp):.0f} wait+barrier {sum(wait)/len(wait):.0f} epilogue {sum(epi)/len(epi):.0f}  block cycles min {min(cyc)} med {sorted(cyc)[128]} max {max(cyc)}  clock {sorted(cyc)[128]/sorted(rt)[128]*100:.0f} MHz")
