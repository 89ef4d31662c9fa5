import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dig_amd import synth
from dig_amd.diffusion import GaussianDiffusion, Unet
torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
B, S = 32, 256
unet = Unet(dim=64, dim_mults=(1, 2, 4, 8), flash_attn=True)
synth.fill_state_dict(unet, seed=0)
gd = GaussianDiffusion(unet, image_size=S, timesteps=1000, sampling_timesteps=150).eval().to(dev)
unet.set_compute_dtype(torch.bfloat16)
x0 = synth.synth_images(B, S, S, seed=1234).to(dev)
nz = synth.synth_noise(B, S, S, seed=1235).to(dev)
t = torch.full((B,), 100, dtype=torch.long, device=dev)
def one(lo, hi):
    return gd.p_losses(x0[lo:hi], t[lo:hi], noise=nz[lo:hi])
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("single stream B=32      ", timeit(lambda: one(0, B)), flush=True)
print("single stream 2 x B=16  ", timeit(lambda: (one(0, 16), one(16, 32))), flush=True)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def two():
    with torch.cuda.stream(s1): one(0, 16)
    with torch.cuda.stream(s2): one(16, 32)
print("two streams   2 x B=16  ", timeit(two), flush=True)
ss = [torch.cuda.Stream() for _ in range(4)]
def four():
    for i, s in enumerate(ss):
        with torch.cuda.stream(s): one(8 * i, 8 * i + 8)
print("four streams  4 x B=8   ", timeit(four), flush=True)
