"""Debug build of conv3x3_ws.hip (-DWS_ASSERT_ADDR): run the weight-stationary conv at the full bench size and at every shape
class the tests use (border-only images, up2, aff on/off, in0_bstride), then read the violation counter."""
import ctypes, os, sys, torch
sys.path.insert(0, ".")
from dig_amd import ops, _lib
from dig_amd._lib import call
torch.manual_seed(0)
lib = _lib.lib()
viol = (ctypes.c_longlong * 4)()
total = 0
for (B, H, W, up2, aff) in ((32, 256, 256, 0, 1), (32, 256, 256, 0, 0), (32, 128, 128, 0, 1), (5, 16, 16, 0, 1), (3, 32, 48, 0, 0),
                            (2, 64, 64, 1, 0), (7, 48, 16, 0, 1), (160, 128, 128, 0, 1)):
    hs, ws_ = (H // 2, W // 2) if up2 else (H, W)
    x = torch.randn(B, hs, ws_, 64, device="cuda").bfloat16()
    w = (torch.randn(64, 64, 3, 3) * 0.05).cuda()
    pw = ops.ConvWeight(w, (torch.randn(64) * 0.1).cuda(), torch.bfloat16)
    a = torch.randn(B, 64, 2, device="cuda").float().contiguous() if aff else None
    y, st = ops.conv3x3(x, pw, aff=a, up2=bool(up2))
    torch.cuda.synchronize()
    rc = lib.dig_debug_ws_addr_violations(viol)
    print(f"B={B} {H}x{W} up2={up2} aff={aff}: rc={rc} violations={viol[0]} kind={viol[1]} off={viol[2]} tile={viol[3]} out_finite={bool(torch.isfinite(y.float()).all())}", flush=True)
    total += viol[0]
print("TOTAL VIOLATIONS", total)
sys.exit(1 if total else 0)
