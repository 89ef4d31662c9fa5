"""One optimiser step of the MCTformer classification losses (dig_amd.train.update_cls_step) at B images of 224x224: time
per step and per phase (HIP events), fp32.  usage: python tools/lab/bench_train.py [B]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dig_amd import synth, train
from dig_amd.mctformer import create_model
torch.set_grad_enabled(False)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = create_model("deit_small_MCTformerV2_DiG", pretrained=False, num_classes=20, lfca=False)
synth.fill_state_dict(m, seed=0, skip_prefixes=("norm.",))
m = m.cuda().eval()
img = synth.synth_imagenet_images(B, 224, 224, seed=1).cuda()
label = synth.synth_labels(B, 20, seed=2).cuda()
opt = train.AdamW(m, lr=5e-4 * B / 512.0)
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for _ in range(2): train.update_cls_step(m, opt, img, label)
torch.cuda.synchronize()
K = 5
t = [0.0, 0.0, 0.0]
for _ in range(K):
    e0 = ev(); outs, ctx = train.forward_train(m, img)
    e1 = ev(); g = train.backward_cls(m, outs, ctx, label, 1.0)
    e2 = ev(); opt.step(g)
    e3 = ev(); torch.cuda.synchronize()
    for i, (a, b) in enumerate(((e0, e1), (e1, e2), (e2, e3))): t[i] += a.elapsed_time(b) / K
# matmul flops of one ViT-S pass at N = 216 tokens: per block 2*N*D*(3D + D + 8D) + 4*N*N*D; backward = 2x forward
D, N = 384, 216
fwd = B * 12 * (2 * N * D * 12 * D + 4 * N * N * D) / 1e12
print(f"B={B}: forward {t[0]:.2f} ms, backward {t[1]:.2f} ms, AdamW {t[2]:.2f} ms, step {sum(t):.2f} ms = {B / sum(t) * 1e3:.0f} img/s; "
      f"block matmuls {fwd:.3f} + {2 * fwd:.3f} TFLOP -> {fwd / t[0] * 1e3:.1f} / {2 * fwd / t[1] * 1e3:.1f} TFLOP/s fp32")
