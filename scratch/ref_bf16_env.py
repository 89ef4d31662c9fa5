import sys, torch, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import ref_import, numpy as np
from dig_amd import synth
torch.set_grad_enabled(False)
ddp = ref_import.load_ddp()
for name, dim, H, W, B, tv in [("unet_tiny32",16,32,32,2,[3,100]),("unet128_t100",64,128,128,1,[100])]:
    unet = ddp.Unet(dim=dim, dim_mults=(1,2,4,8), flash_attn=False)
    synth.fill_state_dict(unet, seed=0)
    gd = ddp.GaussianDiffusion(unet, image_size=H, timesteps=1000, sampling_timesteps=150).eval()
    gd.model.training=False
    x0 = synth.synth_images(B,H,W,1234); nz = synth.synth_noise(B,H,W,1235); t = torch.tensor(tv)
    x_t, v, feats = gd.p_losses(x0, t, noise=nz.clone())
    with torch.autocast('cpu', dtype=torch.bfloat16):
        x_t2, v2, feats2 = gd.p_losses(x0, t, noise=nz.clone())
    print(name, 'v', float((v-v2.float()).abs().max()), 'vmax', float(v.abs().max()), [ (round(float((a-b.float()).abs().max()),4), round(float(a.abs().max()),2)) for a,b in zip(feats,feats2)])
