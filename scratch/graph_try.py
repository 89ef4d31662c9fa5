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
def step():
    x_t, v, feats = gd.p_losses(x0, t, noise=nz)
    return v, unet._last_maps
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager ms", timeit(step), flush=True)
v_e, m_e = step(); v_e = v_e.clone(); m_e = m_e.clone()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    v_g, m_g = step()
print("captured", flush=True)
print("graph ms", timeit(g.replay), flush=True)
g.replay(); torch.cuda.synchronize()
print("match", torch.equal(v_g, v_e), torch.equal(m_g, m_e))
