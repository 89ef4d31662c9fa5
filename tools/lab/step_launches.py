"""Every C-ABI launch of one B=32, 256x256 step with its key shape fields and HIP-event time (uses dig_amd._lib.TRACE)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dig_amd import _lib, synth
from dig_amd.diffusion import GaussianDiffusion, Unet
torch.set_grad_enabled(False)
B, S = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 256
u = Unet(dim=64, dim_mults=(1, 2, 4, 8), flash_attn=True)
synth.fill_state_dict(u, seed=0)
gd = GaussianDiffusion(u, image_size=S, timesteps=1000, sampling_timesteps=150).eval().cuda()
u.set_compute_dtype(torch.bfloat16)
x0 = synth.synth_images(B, S, S, seed=1234).cuda(); nz = synth.synth_noise(B, S, S, seed=1235).cuda()
t = torch.full((B,), 100, dtype=torch.long, device="cuda")
for _ in range(3): gd.p_losses(x0, t, noise=nz)
torch.cuda.synchronize()
log, pend = [], []
KEYS = ("M", "N", "K", "K0", "K1", "B", "H", "W", "C0", "C1", "Cout", "C", "HW", "n", "hid", "rows", "up2", "a_mode", "batch")
def trace(fn, kw, phase):
    ev = torch.cuda.Event(enable_timing=True); ev.record()
    if phase == 0: pend.append(ev); return
    log.append((fn, {k: kw[k] for k in KEYS if k in kw and not hasattr(kw[k], "shape")}, "aff" if kw.get("aff0") is not None else "", pend.pop(), ev))
N = 5
_lib.TRACE = trace
for _ in range(N): gd.p_losses(x0, t, noise=nz)
torch.cuda.synchronize()
_lib.TRACE = None
n = len(log) // N
tot = 0
for i in range(n):
    us = sum(log[i + k * n][3].elapsed_time(log[i + k * n][4]) for k in range(N)) / N * 1e3
    tot += us
    print(f"{i:3d} {log[i][0]:20s} {us:8.1f} us  {log[i][2]:3s} {log[i][1]}")
print("total", tot)
