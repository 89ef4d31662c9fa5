"""print per-parameter gradient errors of the full update step against a reference fixture (debug aid).
usage: dbg_lfca_grads.py [eval|droppath]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from dig_amd import synth, train
from dig_amd.mctformer import create_model
DEV = "cuda:0"
mode = sys.argv[1] if len(sys.argv) > 1 else "eval"
name, s0, s1, sl, sf = ("grads_update224_c20", 6300, 6301, 15, 37) if mode == "eval" else ("grads_update_droppath224_c20", 6400, 6401, 16, 47)
fx = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", name + ".npz"))
W = tuple(float(v) for v in fx["W"])
B = fx["label"].shape[0]
m = create_model("deit_small_MCTformerV2_DiG", pretrained=False, num_classes=20, drop_rate=0.0, drop_path_rate=0.0, drop_block_rate=None, lfca=True)
synth.fill_state_dict(m, seed=0, skip_prefixes=("norm.",))
m = m.to(DEV).eval()
img = synth.synth_imagenet_images(B, 224, 224, seed=s0).to(DEV)
img_diff = synth.synth_imagenet_images(B, 224, 224, seed=s1).to(DEV)
label = synth.synth_labels(B, 20, seed=sl)
feat = [(synth.normal("feat_diff", (B, 1024, 14, 14), seed=sf + j) * 1.5).to(DEV) for j in range(5)]
dp = None
if mode != "eval":
    Tn = lambda a: torch.from_numpy(np.asarray(a)).to(DEV)
    d1, d2, dl = Tn(fx["dp_pass1"]), Tn(fx["dp_pass2"]), Tn(fx["dp_lfca"])
    blocks = lambda d: [(None, None) if i == 0 else (d[i, 0], d[i, 1]) for i in range(d.shape[0])]
    dp = {"pass1": {"blocks": blocks(d1), "LFCA0": (dl[0, 0], dl[0, 1]), "LFCA1": (dl[1, 0], dl[1, 1])}, "pass2": {"blocks": blocks(d2)}}
opt = train.AdamW(m, lr=0.0, weight_decay=0.0)
with torch.no_grad():
    loss, g = train.update_step(m, opt, img, label.to(DEV), W=W, img_diff=img_diff, feat_diff=feat, epo=10, T=float(fx["T"]), A=float(fx["A"]), drop_path=dp)
print("loss", float(loss), float(fx["loss"]))
names = [str(n) for n in fx["names"]]
params = dict(m.named_parameters())
for i, k in enumerate(names):
    gk = g[k].reshape(params[k].shape).float().cpu()
    idx = synth.sample_index(k, gk.numel())
    rs, rn = torch.from_numpy(fx[f"g{i}_s"]), float(fx[f"g{i}_norm"])
    sabs = float((gk.flatten()[idx] - rs).abs().max())
    samp = sabs / (float(rs.abs().max()) + 1e-12)
    dn = abs(float(gk.double().norm()) - rn) / (rn + 1e-30)
    print(f"{k:45s} norm {rn:10.3e} dnorm {dn:9.2e} samp {samp:9.2e} abs {sabs:9.2e}")
