import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import diffusion as D, synth
from dig_amd._lib import call, dtype_code
torch.set_grad_enabled(False)
u = D.Unet(dim=64, dim_mults=(1, 2, 4, 8), flash_attn=True)
synth.fill_state_dict(u, seed=0)
gd = D.GaussianDiffusion(u, image_size=256, timesteps=1000, sampling_timesteps=150).cuda().eval()
P = u.packed()
B, S = 32, 256
x0 = synth.synth_images(B, S, S, 1234).cuda(); nz = synth.synth_noise(B, S, S, 1235).cuda()
t = torch.full((B,), 100, dtype=torch.long, device='cuda')
def stem(lo, hi):
    b = hi - lo
    x = torch.empty(b, S, S, 64, dtype=torch.bfloat16, device='cuda')
    xt = torch.empty(b, 3, S, S, device='cuda')
    call("qsample_stem", img=x0[lo:hi].contiguous(), noise=nz[lo:hi].contiguous(), t=t[lo:hi].contiguous(), sqrt_ac=gd.sqrt_alphas_cumprod, sqrt_1mac=gd.sqrt_one_minus_alphas_cumprod,
         w=P["stem_w"], bias=P["stem_b"], x_t=xt, out=x, pre_scale=(1.,)*3, pre_shift=(0.,)*3, B=b, H=S, W=S, Cout=64, dtype=0)
    return x, xt
full, fxt = stem(0, 32)
full2, _ = stem(0, 32)
print("full vs full2 equal:", torch.equal(full, full2))
for lo, hi in ((28, 32), (24, 32), (16, 32), (29, 30), (28, 30)):
    part, pxt = stem(lo, hi)
    d = (part.float() - full[lo:hi].float()).abs()
    nzr = d.nonzero()
    print(f"shard {lo}:{hi}: max {float(d.max())} nnz {len(nzr)} xt equal {torch.equal(pxt, fxt[lo:hi])}")
    if len(nzr):
        ys, xs = nzr[:, 1], nzr[:, 2]
        print("   images", sorted(set(nzr[:, 0].tolist())), "y range", int(ys.min()), int(ys.max()), "x range", int(xs.min()), int(xs.max()), "channels", sorted(set(nzr[:, 3].tolist()))[:10])
        print("   first", nzr[:5].tolist())
