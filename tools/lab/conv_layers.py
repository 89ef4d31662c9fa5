"""Per-launch table of every conv3x3 of one B=32, 256x256 step: shape, prologue/up2 flags, us, TFLOP/s (HIP events)."""
import sys, torch
sys.path.insert(0, ".")
from dig_amd import ops, synth
from dig_amd.diffusion import GaussianDiffusion, Unet
torch.set_grad_enabled(False)
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 256
u = Unet(dim=64, dim_mults=(1, 2, 4, 8), flash_attn=True)
synth.fill_state_dict(u, seed=0)
gd = GaussianDiffusion(u, image_size=S, timesteps=1000, sampling_timesteps=150).eval().cuda()
u.set_compute_dtype(torch.bfloat16)
x0 = synth.synth_images(B, S, S, seed=1234).cuda()
nz = synth.synth_noise(B, S, S, seed=1235).cuda()
t = torch.full((B,), 100, dtype=torch.long, device="cuda")
log, on = [], [False]
orig = ops.conv3x3
def timed(x0_, cw, x1=None, aff=None, up2=False, want_stats=True, groups=8):
    if not on[0]:
        return orig(x0_, cw, x1=x1, aff=aff, up2=up2, want_stats=want_stats, groups=groups)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = orig(x0_, cw, x1=x1, aff=aff, up2=up2, want_stats=want_stats, groups=groups)
    e1.record()
    Bc, H, W, Co = out[0].shape
    log.append(((cw.C0, cw.C1, Co, H, W, int(aff is not None), int(up2)), 2.0 * Bc * H * W * Co * (cw.C0 + cw.C1) * 9, e0, e1))
    return out
ops.conv3x3 = timed
for _ in range(3):
    gd.p_losses(x0, t, noise=nz)
torch.cuda.synchronize()
on[0] = True
N = 5
for _ in range(N):
    gd.p_losses(x0, t, noise=nz)
torch.cuda.synchronize()
n = len(log) // N
print(" C0  C1  Cout    HxW  aff up2      us   TFLOP/s")
tot = 0
for i in range(n):
    us = sum(log[i + k * n][2].elapsed_time(log[i + k * n][3]) for k in range(N)) / N * 1e3
    k, fl = log[i][0], log[i][1]
    tot += us
    print(f"{k[0]:4d}{k[1]:4d}{k[2]:6d} {k[3]:4d}x{k[4]:<4d}{k[5]:3d}{k[6]:4d} {us:8.1f} {fl / us / 1e6:8.0f}")
print("total us", tot)
