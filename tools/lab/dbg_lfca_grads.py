"""print per-parameter gradient errors of the full update step against the reference fixture (debug aid)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from dig_amd import synth
from dig_amd import train
from dig_amd.mctformer import create_model
DEV = "cuda:0"
fx = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "grads_update224_c20.npz"))
W = tuple(float(v) for v in fx["W"])
m = create_model("deit_small_MCTformerV2_DiG", pretrained=False, num_classes=20, drop_rate=0.0, drop_path_rate=0.0, drop_block_rate=None, lfca=True)
synth.fill_state_dict(m, seed=0, skip_prefixes=("norm.",))
m = m.to(DEV).eval()
img = synth.synth_imagenet_images(2, 224, 224, seed=6300).to(DEV)
img_diff = synth.synth_imagenet_images(2, 224, 224, seed=6301).to(DEV)
label = synth.synth_labels(2, 20, seed=15)
feat = [(synth.normal("feat_diff", (2, 1024, 14, 14), seed=37 + j) * 1.5).to(DEV) for j in range(5)]
opt = train.AdamW(m, lr=0.0, weight_decay=0.0)
with torch.no_grad():
    loss, g = train.update_step(m, opt, img, label.to(DEV), W=W, img_diff=img_diff, feat_diff=feat, epo=10, T=float(fx["T"]), A=float(fx["A"]))
print("loss", float(loss), float(fx["loss"]))
names = [str(n) for n in fx["names"]]
params = dict(m.named_parameters())
for i, k in enumerate(names):
    if not k.startswith(("LFCA", "diff_head_p")):
        continue
    gk = g[k].reshape(params[k].shape).float().cpu()
    idx = synth.sample_index(k, gk.numel())
    rs, rn = torch.from_numpy(fx[f"g{i}_s"]), float(fx[f"g{i}_norm"])
    samp = float((gk.flatten()[idx] - rs).abs().max()) / (float(rs.abs().max()) + 1e-12)
    dn = abs(float(gk.double().norm()) - rn) / (rn + 1e-30)
    print(f"{k:45s} norm {rn:10.3e} dnorm {dn:9.2e} samp {samp:9.2e}")
