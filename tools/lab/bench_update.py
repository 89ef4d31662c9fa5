"""One FULL optimiser step of model_WSSS.update's MCTformer part (dig_amd.train.update_step: all five loss terms, two ViT
passes, the LFCA branch on its tape, train-mode DropPath, AdamW) at B images of 224x224, fp32: HIP-event time per step.
usage: python tools/lab/bench_update.py [B]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dig_amd import synth, train
from dig_amd.mctformer import create_model
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = create_model("deit_small_MCTformerV2_DiG", pretrained=False, num_classes=20, drop_path_rate=0.1, lfca=True)
synth.fill_state_dict(m, seed=0, skip_prefixes=("norm.",))
m = m.cuda().eval()
img = synth.synth_imagenet_images(B, 224, 224, seed=1).cuda()
img_diff = synth.synth_imagenet_images(B, 224, 224, seed=3).cuda()
label = synth.synth_labels(B, 20, seed=2).cuda()
feat = [(synth.normal("feat_diff", (B, 1024, 14, 14), seed=37 + j) * 1.5).cuda() for j in range(5)]
opt = train.AdamW(m, lr=5e-4 * B / 512.0)
gen = torch.Generator(device="cuda").manual_seed(0)
def step():
    dp = train.draw_drop_path(m, B, "cuda", generator=gen)
    return train.update_step(m, opt, img, label, img_diff=img_diff, feat_diff=feat, epo=10, drop_path=dp)
for _ in range(2): step()
torch.cuda.synchronize()
K = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(K): loss, _ = step()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
print(f"B={B}: full update step (5 terms, LFCA tape, DropPath, AdamW over {sum(p.numel() for p in m.parameters()) / 1e6:.0f} M parameters) "
      f"{ms:.1f} ms = {B / ms * 1e3:.0f} img/s, loss {float(loss):.4f}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
