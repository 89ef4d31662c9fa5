import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import diffusion as D, synth
from dig_amd._lib import call, dtype_code
dt = torch.bfloat16
u = D.Unet(dim=64, dim_mults=(1, 2, 4, 8), flash_attn=True)
synth.fill_state_dict(u, seed=0)
u = u.cuda(); u.set_compute_dtype(dt)
P = u.packed()
tag = os.environ.get("DIG_HIP_LIB", "default").split("lib_")[-1][-10:]
for name, H, C, p in (("L0", 256, 64, P["downs"][0]["attn"]), ("L1", 128, 64, P["downs"][1]["attn"]), ("L2", 64, 128, P["downs"][2]["attn"]),
                      ("L1u", 128, 128, P["ups"][2]["attn"]), ("L2u", 64, 256, P["ups"][1]["attn"])):
    B = 32
    x = torch.randn(B, H, H, C, device='cuda').to(dt)
    n = H * H
    splits = D.la_splits(n)
    part = torch.empty(B, splits, 4, 32, 32, dtype=torch.float32, device='cuda')
    pmax = torch.empty(B, splits, 4, 32, dtype=torch.float32, device='cuda')
    psum = torch.empty(B, splits, 4, 32, dtype=torch.float32, device='cuda')
    f = lambda: call("la_kvctx", x=x, wkv=p["wkv_g"], part=part, pmax=pmax, psum=psum, B=B, n=n, C=C, hid=128, splits=splits, dtype=dtype_code(dt))
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{tag:10s} la_kvctx {name} {H}x{H}x{C} splits={splits}: {e0.elapsed_time(e1)/10*1e3:.1f} us", flush=True)
