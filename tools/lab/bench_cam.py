"""Secondary measurement (BASELINE.json configs[2]): MCTformer class-token attention -> CAM path, B images at 224x224, C=20."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dig_amd import cam, synth
from dig_amd.mctformer import create_model
torch.set_grad_enabled(False)
B, C = int(os.environ.get("B", 32)), int(os.environ.get("C", 20))
H, W = int(os.environ.get("H", 224)), int(os.environ.get("W", 224))
m = create_model("deit_small_MCTformerV2_DiG", pretrained=False, num_classes=C, drop_rate=0.0, drop_path_rate=0.1, drop_block_rate=None)
synth.fill_state_dict(m, seed=0, skip_prefixes=("LFCA", "diff_head_p", "norm."))
m = m.cuda().eval()
img = synth.synth_imagenet_images(B, H, W).cuda()
label = synth.synth_labels(B, C).cuda()
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
ms = timeit(lambda: m(img, return_att=True, n_layers=12))
gf = 10.18 if (H, W, C) == (224, 224, 20) else None
print(f"rcams forward  B={B} C={C} {H}x{W} ({C + (H // 16) * (W // 16)} tokens): {ms:.3f} ms  = {B / ms * 1e3:.0f} img/s" + (f"  ({B * gf / ms:.1f} TFLOP/s at {gf} GFLOP/img)" if gf else ""), flush=True)
ms = timeit(lambda: cam.cam_labels(cam.infer_multi_cam(m, img, label, 12), label, 0.45))
print(f"infer_multi (flip TTA, max_norm) + argmax labels: {ms:.3f} ms = {B / ms * 1e3:.0f} img/s", flush=True)
