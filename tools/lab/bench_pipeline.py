"""Secondary measurement (BASELINE.json configs[2]/[3]): the whole forward of model_WSSS.update / infer_multi on a batch of images:
six Unet passes (five low-t ones batched) -> five guidance maps + the t=100 reconstruction -> MCTformer with the LFCA consumer ->
MCTformer on the reconstruction -> flip-TTA CAMs -> uint8 labels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dig_amd import cam, guidance, synth
from dig_amd.diffusion import GaussianDiffusion, Unet
from dig_amd.mctformer import create_model
torch.set_grad_enabled(False)
B, S = int(os.environ.get("B", 16)), int(os.environ.get("S", 224))
u = Unet(dim=64, dim_mults=(1, 2, 4, 8), flash_attn=True)
synth.fill_state_dict(u, seed=0)
gd = GaussianDiffusion(u, image_size=S, timesteps=1000, sampling_timesteps=150).eval().cuda()
u.set_compute_dtype(torch.bfloat16)
m = create_model("deit_small_MCTformerV2_DiG", pretrained=False, num_classes=20, lfca=True)
synth.fill_state_dict(m, seed=0, skip_prefixes=("norm.",))
m = m.cuda().eval().set_lfca_dtype(torch.bfloat16)
img = synth.synth_imagenet_images(B, S, S).cuda()
label = synth.synth_labels(B, 20).cuda()
noises = [torch.randn_like(img) for _ in range(6)]
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
t_g = timeit(lambda: guidance.extract_guidance(gd, img, noises))
t_u = timeit(lambda: guidance.update_forward(gd, m, img, noises))
t_i = timeit(lambda: cam.cam_labels(cam.infer_multi_cam(m, img, label, 12), label, 0.45))
print(f"B={B} {S}x{S}: guidance extraction (6 Unet passes, bf16) {t_g:.1f} ms = {5 * B / t_g * 1e3:.0f} maps/s; "
      f"update forward (guidance + LFCA pass + pass on img_diff) {t_u:.1f} ms = {B / t_u * 1e3:.0f} img/s; "
      f"infer_multi + labels {t_i:.1f} ms = {B / t_i * 1e3:.0f} img/s")
