"""Every C-ABI launch of one training step (B images, 224x224) with its shape fields and HIP-event time, by entry point.
usage: train_launches.py [B] [full]   (full: all five terms with the LFCA tape, update_step; default: the classification losses)"""
import sys, os, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dig_amd import _lib, synth, train
from dig_amd.mctformer import create_model
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
FULL = len(sys.argv) > 2 and sys.argv[2] == "full"
m = create_model("deit_small_MCTformerV2_DiG", pretrained=False, num_classes=20, lfca=FULL)
synth.fill_state_dict(m, seed=0, skip_prefixes=("norm.",))
m = m.cuda().eval()
img = synth.synth_imagenet_images(B, 224, 224, seed=1).cuda()
label = synth.synth_labels(B, 20, seed=2).cuda()
opt = train.AdamW(m, lr=1e-5)
if FULL:
    img_diff = synth.synth_imagenet_images(B, 224, 224, seed=3).cuda()
    feat = [(synth.normal("feat_diff", (B, 1024, 14, 14), seed=37 + j) * 1.5).cuda() for j in range(5)]
    step = lambda: train.update_step(m, opt, img, label, img_diff=img_diff, feat_diff=feat, epo=10)
else:
    step = lambda: train.update_cls_step(m, opt, img, label)
for _ in range(2): step()
torch.cuda.synchronize()
log, pend = [], []
KEYS = ("M", "N", "K", "batch", "ksplit", "sam", "sak", "sbk", "sbn", "rows", "D", "n", "C0", "Cout", "H")
def trace(fn, kw, phase):
    ev = torch.cuda.Event(enable_timing=True); ev.record()
    if phase == 0: pend.append(ev); return
    log.append((fn, tuple((k, kw[k]) for k in KEYS if k in kw), pend.pop(), ev))
_lib.TRACE = trace
step()
torch.cuda.synchronize()
_lib.TRACE = None
agg = collections.defaultdict(lambda: [0, 0.0])
for fn, key, a, b in log:
    k = (fn, key if fn in ("gemm_f32", "gemm_nt", "conv3x3") else ())
    agg[k][0] += 1; agg[k][1] += a.elapsed_time(b) * 1e3
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:40]:
    fl = ""
    d = dict(k[1])
    if "K" in d: fl = f"{2 * d['M'] * d['N'] * d['K'] * d.get('batch', 1) * v[0] / v[1] / 1e6:7.1f} TF"
    print(f"{k[0]:16s} x{v[0]:4d} {v[1]:9.1f} us {100 * v[1] / tot:5.1f}% {fl}  {d}")
print("total", tot)
