import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dig_amd import diffusion as D, synth
dt = torch.bfloat16
u = D.Unet(dim=64, dim_mults=(1, 2, 4, 8), flash_attn=True)
synth.fill_state_dict(u, seed=0)
u = u.cuda(); u.set_compute_dtype(dt)
P = u.packed()
for lvl, (H, C) in enumerate([(256, 64), (128, 64), (64, 128)]):
    x = torch.randn(32, H, H, C, device='cuda').to(dt)
    p = P["downs"][lvl]["attn"]
    for _ in range(2): D._linear_attention(p, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): D._linear_attention(p, x)
    e1.record(); torch.cuda.synchronize()
    print(f"splits={os.environ.get('DIG_LA_SPLITS','64')} L{lvl} {H}x{H}x{C}: {e0.elapsed_time(e1)/10*1e3:.1f} us", flush=True)
