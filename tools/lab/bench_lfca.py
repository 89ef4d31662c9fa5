"""Secondary measurement: MCTformer forward with the LFCA consumer of the guidance maps, fp32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dig_amd import synth
from dig_amd.mctformer import create_model
torch.set_grad_enabled(False)
B = int(os.environ.get("B", 8))
m = create_model("deit_small_MCTformerV2_DiG", pretrained=False, num_classes=20, lfca=True)
synth.fill_state_dict(m, seed=0, skip_prefixes=("norm.",))
m = m.cuda().eval()
img = synth.synth_imagenet_images(B, 224, 224).cuda()
feat = [torch.randn(B, 1024, 14, 14, device="cuda") for _ in range(5)]
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
t0 = timeit(lambda: m(img))
t1 = timeit(lambda: m(img, feat_diff=feat))
m.set_lfca_dtype(torch.bfloat16)
t2 = timeit(lambda: m(img, feat_diff=feat))
print(f"bf16 storage for the LFCA convolutions: {t2:.2f} ms -> LFCA part {t2 - t0:.2f} ms = {B * 74.0 / (t2 - t0):.1f} TFLOP/s")
print(f"B={B} 224x224: ViT/CAM forward {t0:.2f} ms; with LFCA0/1 + diff_head_p {t1:.2f} ms -> LFCA part {t1 - t0:.2f} ms = {B * 74.0 / (t1 - t0):.1f} TFLOP/s (74 GFLOP/img, fp32)")
