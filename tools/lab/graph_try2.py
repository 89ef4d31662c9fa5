import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dig_amd import synth
from dig_amd.diffusion import GaussianDiffusion, Unet
torch.set_grad_enabled(False)
unet = Unet(dim=64, dim_mults=(1, 2, 4, 8), flash_attn=True)
synth.fill_state_dict(unet, seed=0)
gd = GaussianDiffusion(unet, image_size=256, timesteps=1000, sampling_timesteps=150).eval().cuda()
for B in (4, 8, 20, 32):
    x0 = synth.synth_images(B, 256, 256, seed=1).cuda(); nz = synth.synth_noise(B, 256, 256, seed=2).cuda()
    t = torch.full((B,), 100, dtype=torch.long, device='cuda')
    for _ in range(3): gd.p_losses(x0, t, noise=nz)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): gd.p_losses(x0, t, noise=nz)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 20
    # host-only cost: time to enqueue without waiting
    t0 = time.perf_counter()
    for _ in range(5): gd.p_losses(x0, t, noise=nz)
    host = (time.perf_counter() - t0) / 5
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        gd.p_losses(x0, t, noise=nz)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = gd.p_losses(x0, t, noise=nz)
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 20
    ref = gd.p_losses(x0, t, noise=nz)
    g.replay(); torch.cuda.synchronize()
    print(f"B={B}: eager {eager*1e3:.3f} ms (host enqueue {host*1e3:.3f} ms)  graph {graph*1e3:.3f} ms  equal {torch.equal(ref[1], out[1])}", flush=True)
