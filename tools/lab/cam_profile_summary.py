"""Per-kernel table of the CAM path from a rocprofv3 --kernel-trace --stats CSV: time share, and for the HBM-bound kernels
(LayerNorm, cam_fuse, cam_tta_norm, cam_argmax, patchify) the achieved GB/s against algorithmic bytes, for the fp32 GEMMs the
TFLOP/s against the 157 TFLOP/s fp32 matrix peak.  usage: cam_profile_summary.py <kernel_stats.csv> B C H W calls"""
import csv, sys
f, B, C, H, W, calls = sys.argv[1], *map(int, sys.argv[2:7])
hp, wp = H // 16, W // 16
P, D = hp * wp, 384
N = C + P
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
# algorithmic bytes / flops PER forward of the batch (flip-TTA forwards run 2B images; `calls` = forwards of B-image batches profiled, counted by the caller)
def per(name):
    r = [x for x in rows if name in x["Name"]]
    return (sum(float(x["TotalDurationNs"]) for x in r), sum(int(x["Calls"]) for x in r)) if r else (0.0, 0)
print(f"B={B} C={C} {H}x{W}: {N} tokens; total kernel time {tot / 1e6:.2f} ms over the profiled run")
print(f"{'kernel':42s} {'calls':>6s} {'avg us':>9s} {'share':>6s}  note")
for r in sorted(rows, key=lambda x: -float(x["TotalDurationNs"]))[:14]:
    ns, c = float(r["TotalDurationNs"]), int(r["Calls"])
    name, note = r["Name"], ""
    avg = ns / c
    if "layernorm" in name:
        by = 2.0 * B * N * D * 4
        note = f"{by / avg:.0f} GB/s of 8000 (2 x rows x 384 x 4 B; rows = B*N, flip batches 2x)"
    elif "cam_tta_norm" in name:
        by = B * C * H * W * 4.0
        note = f"{by / avg:.0f} GB/s of 8000 (writes B*C*H*W fp32)"
    elif "cam_fuse" in name:
        by = B * (P * P + C * P) * 4.0
        note = f"{by / avg:.0f} GB/s of 8000 (A_pp + A_cp; flip batches 2x)"
    elif "gemm_nt" in name:
        note = "fp32 MFMA GEMM class (qkv / proj / MLP / patch embed)"
    print(f"{name[:42]:42s} {c:6d} {avg / 1e3:9.1f} {ns / tot * 100:5.1f}%  {note}")
g, gc = per("gemm_nt")
fl = 12 * 2.0 * B * N * D * (3 * D + D + 8 * D)  # per forward: qkv + proj + fc1 + fc2 of 12 blocks
print(f"fp32 GEMMs: {g / 1e6:.2f} ms in {gc} launches; block linears = {fl / 1e12:.3f} TFLOP per B-image forward")
