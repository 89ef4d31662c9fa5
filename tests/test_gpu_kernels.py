"""GPU tier (-m gpu): each HIP kernel, called through the C ABI, against the CPU oracle / plain torch fp32 on
seeded inputs.  fp32 storage must agree to 2e-4 (MFMA f32 accumulation order differs from torch's);
bf16 storage is compared with a tolerance scaled by the output magnitude (bf16 has 8 significant bits)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import dig_oracle as O
from dig_amd import ops, synth
from dig_amd._lib import call, dtype_code

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def tol(dt, ref):
    return 2e-4 * max(1.0, float(ref.abs().max())) if dt == torch.float32 else 2.5e-2 * max(1.0, float(ref.abs().max()))


def nhwc(x, dt):
    return x.permute(0, 2, 3, 1).contiguous().to(device=DEV, dtype=dt)


def nchw(y):
    return y.float().permute(0, 3, 1, 2).cpu()


def rnd(tag, shape, scale=1.0):
    return synth.uniform_pm1(tag, shape) * scale


def q(x, dt):
    """round-trip through the storage dtype so the CPU reference sees the same inputs"""
    return x.to(dt).float()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 32, 20, 24, 64), (1, 64, 16, 16, 128), (1, 16, 7, 9, 16), (1, 384, 14, 13, 20), (1, 48, 33, 17, 32),
                                   (2, 24, 16, 32, 64), (1, 96, 18, 18, 256)])
def test_conv3x3_plain(dt, shape):
    B, Ci, H, W, Co = shape
    x, w, b = q(rnd("cx", (B, Ci, H, W)), dt), q(rnd("cw", (Co, Ci, 3, 3), (3.0 / (9 * Ci)) ** 0.5), dt), rnd("cb", (Co,), 0.1)
    ref = F.conv2d(x, w, b, padding=1)
    cw = ops.ConvWeight(w.to(DEV), b.to(DEV), dt)
    stats_ok = Co % 8 == 0
    y, st = ops.conv3x3(nhwc(x, dt), cw, want_stats=stats_ok)
    assert (nchw(y) - ref).abs().max() <= tol(dt, ref)
    if stats_ok:
        st = st.cpu().double().sum(2)  # B,G,2
        g = ref.reshape(B, 8, -1).double()
        assert torch.allclose(st[..., 0], g.sum(-1), rtol=1e-3, atol=1e-2 * g.shape[-1] ** 0.5)
        assert torch.allclose(st[..., 1], (g * g).sum(-1), rtol=2e-3 if dt == torch.float32 else 2e-2)


@pytest.mark.parametrize("dt", DTYPES)
def test_conv3x3_concat_up2_prologue(dt):
    B, C0, C1, H, W, Co = 2, 32, 16, 10, 12, 32
    x0, x1 = q(rnd("x0", (B, C0, H, W)), dt), q(rnd("x1", (B, C1, H, W)), dt)
    w, b = q(rnd("w", (Co, C0 + C1, 3, 3), 0.1), dt), rnd("b", (Co,), 0.1)
    # concat
    ref = F.conv2d(torch.cat([x0, x1], 1), w, b, padding=1)
    y, _ = ops.conv3x3(nhwc(x0, dt), ops.ConvWeight(w.to(DEV), b.to(DEV), dt, C1=C1), x1=nhwc(x1, dt))
    assert (nchw(y) - ref).abs().max() <= tol(dt, ref)
    # nearest x2 upsample addressing
    w2 = q(rnd("w2", (Co, C0, 3, 3), 0.1), dt)
    ref = F.conv2d(F.interpolate(x0, scale_factor=2, mode="nearest"), w2, b, padding=1)
    cw2 = ops.ConvWeight(w2.to(DEV), b.to(DEV), dt)
    y, _ = ops.conv3x3(nhwc(x0, dt), cw2, up2=True, want_stats=False)
    assert (nchw(y) - ref).abs().max() <= tol(dt, ref)
    # GroupNorm-apply + SiLU prologue
    aff = torch.stack([1.0 + 0.3 * rnd("a", (B, C0)), 0.2 * rnd("bb", (B, C0))], -1)
    xin = F.silu(x0 * aff[..., 0][:, :, None, None] + aff[..., 1][:, :, None, None])
    ref = F.conv2d(q(xin, dt), w2, b, padding=1)
    y, _ = ops.conv3x3(nhwc(x0, dt), cw2, aff=aff.to(DEV).contiguous())
    assert (nchw(y) - ref).abs().max() <= tol(dt, ref) * (1 if dt == torch.float32 else 2)


@pytest.mark.parametrize("cfg", [(64, 64, 64, 40, 36), (128, 0, 64, 33, 70), (128, 0, 64, 16, 32), (64, 32, 128, 34, 20), (256, 0, 64, 64, 64)])
def test_conv3x3_double_buffered_variants(cfg):
    """The double-buffered bf16 kernel (Cout = 128) and the generic one (Cout = 64) on wide inputs: concat, nearest-x2 addressing,
    GroupNorm-apply + SiLU prologue and the GroupNorm partials, on ragged tiles, against F.conv2d in fp32."""
    C0, C1, Co, H, W = cfg
    dt, B = torch.bfloat16, 2
    x0 = q(rnd("bx0", (B, C0, H, W)), dt)
    x1 = q(rnd("bx1", (B, C1, H, W)), dt) if C1 else None
    w, b = q(rnd("bw", (Co, C0 + C1, 3, 3), (3.0 / (9 * (C0 + C1))) ** 0.5), dt), rnd("bb", (Co,), 0.1)
    xin = torch.cat([x0, x1], 1) if C1 else x0
    ref = F.conv2d(xin, w, b, padding=1)
    cw = ops.ConvWeight(w.to(DEV), b.to(DEV), dt, C1=C1)
    y, st = ops.conv3x3(nhwc(x0, dt), cw, x1=nhwc(x1, dt) if C1 else None)
    assert (nchw(y) - ref).abs().max() <= tol(dt, ref)
    st = st.cpu().double().sum(2)
    g = ref.reshape(B, 8, -1).double()
    assert torch.allclose(st[..., 0], g.sum(-1), rtol=1e-3, atol=1e-2 * g.shape[-1] ** 0.5)
    assert torch.allclose(st[..., 1], (g * g).sum(-1), rtol=2e-2)
    if C1 == 0:
        # nearest x2 upsample addressing (Upsample, ddp.py:90-94)
        xs = x0[:, :, : H // 2, : W // 2].contiguous()
        H2, W2 = 2 * (H // 2), 2 * (W // 2)
        ref = F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest"), w, b, padding=1)
        y, _ = ops.conv3x3(nhwc(xs, dt), cw, up2=True, want_stats=False)
        assert tuple(y.shape) == (B, H2, W2, Co) and (nchw(y) - ref).abs().max() <= tol(dt, ref)
        # GroupNorm-apply + SiLU prologue
        aff = torch.stack([1.0 + 0.3 * rnd("ba", (B, C0)), 0.2 * rnd("bbb", (B, C0))], -1)
        xp = F.silu(x0 * aff[..., 0][:, :, None, None] + aff[..., 1][:, :, None, None])
        ref = F.conv2d(q(xp, dt), w, b, padding=1)
        y, _ = ops.conv3x3(nhwc(x0, dt), cw, aff=aff.to(DEV).contiguous())
        assert (nchw(y) - ref).abs().max() <= 2 * tol(dt, ref)


@pytest.mark.parametrize("cfg", [(64, 64, 40, 36), (128, 0, 33, 70), (64, 0, 16, 32), (48, 0, 20, 24)])
def test_conv3x3_db_64_channel_instantiation(cfg, monkeypatch):
    """The BN = 64 instantiation of conv3x3_db.hip (off by default: the weight-/register-stationary kernels are faster on the
    Unet's 64-channel layers; DIG_DB64=all routes every Cout = 64 layer to it, weights packed to match): plain, concat,
    nearest-x2 and prologue launches against F.conv2d, GroupNorm partials, and the same bits from both tile heights."""
    monkeypatch.setenv("DIG_DB64", "all")
    C0, C1, H, W = cfg
    dt, B, Co = torch.bfloat16, 2, 64
    x0 = q(rnd("d6x0", (B, C0, H, W)), dt)
    x1 = q(rnd("d6x1", (B, C1, H, W)), dt) if C1 else None
    w, b = q(rnd("d6w", (Co, C0 + C1, 3, 3), (3.0 / (9 * (C0 + C1))) ** 0.5), dt), rnd("d6b", (Co,), 0.1)
    ref = F.conv2d(torch.cat([x0, x1], 1) if C1 else x0, w, b, padding=1)
    cw = ops.ConvWeight(w.to(DEV), b.to(DEV), dt, C1=C1)
    outs = {}
    for th in ("16", "32"):
        monkeypatch.setenv("DIG_DB_TILE", th)
        y, st = ops.conv3x3(nhwc(x0, dt), cw, x1=nhwc(x1, dt) if C1 else None)
        assert (nchw(y) - ref).abs().max() <= tol(dt, ref)
        assert st.shape == (B, 8, ((H + 15) // 16) * ((W + 15) // 16), 2)
        g = ref.reshape(B, 8, -1).double()
        assert torch.allclose(st.cpu().double().sum(2)[..., 0], g.sum(-1), rtol=1e-3, atol=1e-2 * g.shape[-1] ** 0.5)
        outs[th] = [y, st]
        if C1 == 0:
            aff = torch.stack([1.0 + 0.3 * rnd("d6a", (B, C0)), 0.2 * rnd("d6ab", (B, C0))], -1)
            xp = F.silu(x0 * aff[..., 0][:, :, None, None] + aff[..., 1][:, :, None, None])
            yp, _ = ops.conv3x3(nhwc(x0, dt), cw, aff=aff.to(DEV).contiguous())
            assert (nchw(yp) - F.conv2d(q(xp, dt), w, b, padding=1)).abs().max() <= 2 * tol(dt, ref)
            xs = x0[:, :, : H // 2, : W // 2].contiguous()
            yu, _ = ops.conv3x3(nhwc(xs, dt), cw, up2=True, want_stats=False)
            assert (nchw(yu) - F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest"), w, b, padding=1)).abs().max() <= tol(dt, ref)
            outs[th] += [yp, yu]
    for a_, b_ in zip(outs["16"], outs["32"]):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("shape", [(2, 64, 128, 32, 32), (1, 32, 256, 48, 40), (3, 48, 128, 16, 16), (2, 32, 128, 20, 33)])
def test_conv3x3_db_tile_heights_bitwise(shape, monkeypatch):
    """conv3x3_db.hip picks its block shape - 32 or 16 rows, 128 or 64 output channels - by a cost model of the launch, i.e.
    depending on the batch.  All must give the same bits - outputs AND GroupNorm partials (one per 16x16 pixels from either height) - or a
    shard of a batch would not be bitwise equal to the same images inside the full batch.  Ragged heights included."""
    B, Cin, Co, H, W = shape
    dt = torch.bfloat16
    x = nhwc(q(rnd("dbt", (B, Cin, H, W)), dt), dt)
    w, b = q(rnd("dbtw", (Co, Cin, 3, 3), (3.0 / (9 * Cin)) ** 0.5), dt), rnd("dbtb", (Co,), 0.1)
    cw = ops.ConvWeight(w.to(DEV), b.to(DEV), dt)
    aff = torch.stack([1.0 + 0.3 * rnd("dbta", (B, Cin)), 0.2 * rnd("dbtbb", (B, Cin))], -1).to(DEV).contiguous()
    outs, up = {}, {}
    for th in ("16", "32", "16x64", "32x64"):  # rows [x output channels per block; default 128]
        monkeypatch.setenv("DIG_DB_TILE", th)
        outs[th] = ops.conv3x3(x, cw), ops.conv3x3(x, cw, aff=aff)
        up[th] = ops.conv3x3(x, cw, up2=True, want_stats=False)[0]  # nearest-x2 source addressing of the halo DMA
    assert torch.equal(up["16"], up["32"]) and torch.equal(up["16"], up["16x64"]) and torch.equal(up["16"], up["32x64"])
    for th in ("16x64", "32x64"):  # the half-N-tile blocks (lower / upper 64 rows of the 128-row weight slabs): same bits again
        for (ya, sa), (yb, sb) in zip(outs["16"], outs[th]):
            assert torch.equal(ya, yb) and torch.equal(sa, sb), th
    for (y16, s16), (y32, s32) in zip(outs["16"], outs["32"]):
        assert s16.shape == s32.shape == (B, 8, ((H + 15) // 16) * ((W + 15) // 16), 2)
        assert torch.equal(y16, y32) and torch.equal(s16, s32)
    ref = F.conv2d(nchw(x), w, b, padding=1)
    assert (nchw(outs["16"][0][0]) - ref).abs().max() <= tol(dt, ref)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("mnk", [(300, 200, 64), (128, 64, 32), (37, 3, 64), (5, 520, 256), (1000, 72, 136)])
def test_gemm_epilogues(dt, mnk):
    M, N, K = mnk
    a, w, b = q(rnd("ga", (M, K)), dt), q(rnd("gw", (N, K), K ** -0.5), dt), rnd("gb", (N,), 0.1)
    res = q(rnd("gr", (M, N)), dt)
    ref = F.gelu(F.linear(a, w, b)) + res
    y = ops.linear_rows(a.to(device=DEV, dtype=dt), w.to(device=DEV, dtype=dt), bias=b.to(DEV), res=res.to(device=DEV, dtype=dt), gelu=True)
    assert (y.float().cpu() - ref).abs().max() <= tol(dt, ref)
    y32 = ops.linear_rows(a.to(device=DEV, dtype=dt), w.to(device=DEV, dtype=dt), out_f32=True)
    assert y32.dtype == torch.float32
    assert (y32.cpu() - F.linear(a, w)).abs().max() <= tol(torch.float32 if dt == torch.float32 else dt, ref)


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_concat_gnres_unshuffle_nchw(dt):
    B, H, W, C0, C1, N = 2, 6, 8, 32, 16, 48
    x0, x1 = q(rnd("x0", (B, C0, H, W)), dt), q(rnd("x1", (B, C1, H, W)), dt)
    w, b = q(rnd("w", (N, C0 + C1, 1, 1), 0.2), dt), rnd("b", (N,), 0.1)
    raw = q(rnd("raw", (B, N, H, W)), dt)
    aff = torch.stack([1.0 + 0.3 * rnd("a", (B, N)), 0.2 * rnd("bb", (B, N))], -1)
    ref = F.conv2d(torch.cat([x0, x1], 1), w, b) + F.silu(raw * aff[..., 0][:, :, None, None] + aff[..., 1][:, :, None, None])
    y = ops.linear_rows(nhwc(x0, dt), w.reshape(N, -1).to(device=DEV, dtype=dt), bias=b.to(DEV), x1=nhwc(x1, dt), gsrc=nhwc(raw, dt),
                        gaff=aff.to(DEV).contiguous(), rows_per_img=H * W)
    assert (nchw(y) - ref).abs().max() <= tol(dt, ref)
    # Downsample: pixel-unshuffle + conv1x1 (ddp.py:102-106)
    wd = q(rnd("wd", (N, 4 * C0, 1, 1), 0.1), dt)
    ref = O.downsample({"1.weight": wd, "1.bias": b}, x0)
    wdp = wd.reshape(N, C0, 2, 2).permute(0, 2, 3, 1).reshape(N, 4 * C0).contiguous().to(device=DEV, dtype=dt)
    y = ops.downsample_conv(nhwc(x0, dt), wdp, b.to(DEV))
    assert (nchw(y) - ref).abs().max() <= tol(dt, ref)
    # NCHW fp32 store (final_conv)
    w3 = q(rnd("w3", (3, C0, 1, 1), 0.2), dt)
    ref = F.conv2d(x0, w3, b[:3])
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=DEV)
    ops.gemm_nt(nhwc(x0, dt), w3.reshape(3, C0).to(device=DEV, dtype=dt), out, B * H * W, 3, C0, dtype=dt, bias=b[:3].to(DEV).contiguous(),
                out_f32=True, nchw_hw=H * W)
    assert (out.cpu() - ref).abs().max() <= tol(dt, ref)


@pytest.mark.parametrize("shape", [(700, 256, 128, 256, 100), (130, 512, 256, 520, 65), (260, 64, 0, 64, 260), (1100, 96, 32, 40, 550),
                                   (2048, 128, 0, 128, 1024), (90, 1024, 512, 72, 30)])
def test_gemm_dma_pipeline(shape):
    """bf16 GEMMs with K % 32 == 0 take the multi-stage LDS-DMA kernel: 1..48 K steps (pipeline fill, steady state and drain),
    ragged M and N tiles, the concat loader and the GroupNorm+SiLU / bias epilogue, against F.linear in fp32."""
    M, K0, K1, N, rows = shape
    dt = torch.bfloat16
    K = K0 + K1
    x0 = q(rnd("dx0", (M, K0)), dt)
    x1 = q(rnd("dx1", (M, K1)), dt) if K1 else None
    w, b = q(rnd("dw", (N, K), K ** -0.5), dt), rnd("db", (N,), 0.1)
    raw = q(rnd("draw", (M, N)), dt)
    aff = torch.stack([1.0 + 0.3 * rnd("da", (M // rows, N)), 0.2 * rnd("dbb", (M // rows, N))], -1)
    xin = torch.cat([x0, x1], 1) if K1 else x0
    ab = aff.repeat_interleave(rows, 0)
    ref = F.linear(xin, w, b) + F.silu(raw * ab[..., 0] + ab[..., 1])
    dev = lambda t: t.to(device=DEV, dtype=dt)
    y = ops.linear_rows(dev(x0), dev(w), bias=b.to(DEV), x1=dev(x1) if K1 else None, gsrc=dev(raw), gaff=aff.to(DEV).contiguous(), rows_per_img=rows)
    assert (y.float().cpu() - ref).abs().max() <= tol(dt, ref)
    y32 = ops.linear_rows(dev(x0), dev(w), x1=dev(x1) if K1 else None, out_f32=True)
    assert (y32.cpu() - F.linear(xin, w)).abs().max() <= 2e-3 * max(1.0, float(ref.abs().max()))  # bf16 products, fp32 accumulate


@pytest.mark.parametrize("N", [1, 3, 4])
def test_final_conv_stream(N):
    """final_conv (ddp.py:377): K = 64 bf16 rows -> N <= 4 fp32 NCHW planes takes the streaming kernel inside dig_gemm_nt."""
    B, H, W, C = 3, 16, 24, 64
    dt = torch.bfloat16
    x = q(rnd("fx", (B, C, H, W)), dt)
    w, b = q(rnd("fw", (N, C, 1, 1), 0.2), dt), rnd("fb", (N,), 0.1)
    ref = F.conv2d(x, w, b)
    out = torch.empty(B, N, H, W, dtype=torch.float32, device=DEV)
    ops.gemm_nt(nhwc(x, dt), w.reshape(N, C).to(device=DEV, dtype=dt), out, B * H * W, N, C, dtype=dt, bias=b.to(DEV).contiguous(),
                out_f32=True, nchw_hw=H * W)
    assert (out.cpu() - ref).abs().max() <= 1e-4 * max(1.0, float(ref.abs().max()))  # bf16 products, fp32 accumulate


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 37, 50, 64), (1, 16, 32, 64), (3, 48, 70, 64), (2, 20, 24, 16)])
def test_qsample_stem(dt, cfg):
    """normalise + q_sample + init_conv 7x7 (ddp.py:82-83, 853-860, 296): x_t exact, conv against F.conv2d of the storage-rounded
    x_t; ragged tiles, per-image timesteps.  Cout = 64 in bf16 is the im2col-free kernel, everything else the panel kernel."""
    B, H, W, Co = cfg
    img, nz = rnd("simg", (B, 3, H, W)), rnd("snz", (B, 3, H, W), 2.0)
    w, b = q(rnd("sw", (Co, 3, 7, 7), 0.15), dt), rnd("sb", (Co,), 0.1)
    t = torch.tensor([100, 0, 999][:B])
    ca, cs = 0.5 + 0.5 * rnd("sca", (1000,)).abs(), rnd("scs", (1000,)).abs()
    ps, sh = [2.0, 1.5, 0.5], [-1.0, 0.25, 0.0]
    x0 = img * torch.tensor(ps)[None, :, None, None] + torch.tensor(sh)[None, :, None, None]
    x_t_ref = ca[t][:, None, None, None] * x0 + cs[t][:, None, None, None] * nz
    ref = F.conv2d(q(x_t_ref, dt), w, b, padding=3)
    wp = F.pad(w.permute(0, 2, 3, 1).reshape(Co, 147), (0, 13)).contiguous().to(device=DEV, dtype=dt)
    x_t = torch.empty(B, 3, H, W, device=DEV)
    out = torch.empty(B, H, W, Co, dtype=dt, device=DEV)
    call("qsample_stem", img=img.to(DEV), noise=nz.to(DEV), t=t.to(DEV), sqrt_ac=ca.to(DEV), sqrt_1mac=cs.to(DEV), w=wp, bias=b.to(DEV),
         x_t=x_t, out=out, pre_scale=ps, pre_shift=sh, B=B, H=H, W=W, Cout=Co, dtype=dtype_code(dt))
    assert (x_t.cpu() - x_t_ref).abs().max() <= 1e-6 * max(1.0, float(x_t_ref.abs().max()))
    assert (nchw(out) - ref).abs().max() <= tol(dt, ref)


@pytest.mark.parametrize("dt", DTYPES)
def test_gn_finalize_and_apply(dt):
    B, C, H, W = 2, 32, 20, 24
    x = q(rnd("x", (B, C, H, W)) * 2 + 0.5, dt)
    gamma, beta = 1 + 0.1 * rnd("g", (C,)), 0.1 * rnd("be", (C,))
    ss = rnd("ss", (B, 100))
    res = q(rnd("res", (B, C, H, W)), dt)
    sc, sh = ss[:, 10:10 + C], ss[:, 10 + C:10 + 2 * C]
    ref = F.silu(F.group_norm(x, 8, gamma, beta, eps=1e-5) * (sc[:, :, None, None] + 1) + sh[:, :, None, None]) + res
    # produce stats through an identity-free path: 1 tile per image computed on the host
    g = x.reshape(B, 8, -1).double()
    st = torch.stack([g.sum(-1), (g * g).sum(-1)], -1).float().reshape(B, 8, 1, 2)
    aff = ops.gn_finalize(st.to(DEV).contiguous(), gamma.to(DEV), beta.to(DEV), H * W, ss=ss.to(DEV).contiguous(), ss_off=10)
    y = ops.gn_silu_add(nhwc(x, dt), aff, res=nhwc(res, dt))
    assert (nchw(y) - ref).abs().max() <= tol(dt, ref)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("C", [16, 64, 96, 512])
def test_rmsnorm(dt, C):
    x = q(rnd("x", (2, C, 5, 7)), dt)
    g = 1 + 0.1 * rnd("g", (1, C, 1, 1))
    res = q(rnd("r", (2, C, 5, 7)), dt)
    ref = O.rms_norm(x, g) + res
    y = ops.rmsnorm(nhwc(x, dt), g.reshape(-1).to(DEV).contiguous(), res=nhwc(res, dt))
    assert (nchw(y) - ref).abs().max() <= tol(dt, ref)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C,H,W,h,w", [(64, 32, 32, 16, 16), (512, 32, 32, 16, 16), (100, 40, 24, 23, 31), (72, 9, 7, 16, 16), (64, 256, 256, 16, 16)])
def test_bilinear_ac_tiled(dt, C, H, W, h, w):
    """dig_bilinear_ac_concat (the LDS-transposed kernel: 64-channel x 128-pixel tiles, channel and pixel tails, several pixel
    chunks per map) against F.interpolate(mode='bilinear', align_corners=True) of the same stored values (ddp.py:418,434);
    the channels of the concat buffer outside the slice stay untouched."""
    x = rnd("bil", (3, C, H, W)).to(dt).float()
    buf = torch.full((3, C + 24, h, w), 7.0, device=DEV)
    ops.bilinear_ac_into(nhwc(x, dt), buf, 16)
    ref = F.interpolate(x, size=(h, w), mode="bilinear", align_corners=True)
    assert (buf[:, 16:16 + C].cpu() - ref).abs().max() <= 1e-5 * max(1.0, float(ref.abs().max()))
    assert torch.all(buf[:, :16] == 7.0) and torch.all(buf[:, 16 + C:] == 7.0)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C,G,tiles,with_ss,with_res", [(3, 32, 32, 512, 8, 4, True, False), (2, 64, 64, 128, 8, 16, False, True),
                                                              (5, 9, 7, 72, 8, 1, True, True), (2, 128, 128, 64, 8, 64, False, True)])
def test_gn_fin_silu_add_bitwise_equals_the_two_launches(dt, B, H, W, C, G, tiles, with_ss, with_res):
    """dig_gn_fin_silu_add (GroupNorm finalize inside the elementwise launch, ddp.py:155-166,194) against dig_gn_finalize +
    dig_gn_silu_add on the same partial statistics: BITWISE equal (the finalize arithmetic is one shared device function),
    out of place and in place, with and without the time-embedding scale / shift and the residual; more than
    dig_gn_fin_max_tiles partials per image are refused."""
    x = nhwc(rnd("gfx", (B, C, H, W), 2.0), dt)
    res = nhwc(rnd("gfr", (B, C, H, W)), dt) if with_res else None
    st = (torch.rand(B, G, tiles, 2, generator=torch.Generator().manual_seed(5)) * 40 + 1).to(DEV)
    st[..., 1] += 3000.0  # sum of squares well above (sum)^2 / n
    gamma, beta = rnd("gfg", (C,)).to(DEV), rnd("gfb", (C,)).to(DEV)
    ss = rnd("gfs", (B, 2 * C + 16)).to(DEV) if with_ss else None
    aff = ops.gn_finalize(st, gamma, beta, H * W, ss=ss, ss_off=16 if with_ss else 0, eps=1e-5)
    ref = ops.gn_silu_add(x, aff, res=res)
    got = ops.gn_fin_silu_add(x, st, gamma, beta, H * W, ss=ss, ss_off=16 if with_ss else 0, eps=1e-5, res=res)
    assert torch.equal(got, ref)
    xi = x.clone()
    assert ops.gn_fin_silu_add(xi, st, gamma, beta, H * W, ss=ss, ss_off=16 if with_ss else 0, eps=1e-5, res=res, inplace=True) is xi
    assert torch.equal(xi, ref)
    assert ops.gn_fin_supported(st, C) and not ops.gn_fin_supported(torch.empty(B, G, 65, 2), C)
    with pytest.raises(RuntimeError, match="tiles"):
        ops.gn_fin_silu_add(x, torch.ones(B, G, 65, 2, device=DEV), gamma, beta, H * W)


def test_small_kernels():
    # sinusoid (ddp.py:118-130)
    t = torch.tensor([0, 1, 4, 100, 999])
    out = torch.empty(5, 64, dtype=torch.float32, device=DEV)
    call("sinusoid", t=t.to(DEV), out=out, B=5, dim=64, dtype=dtype_code(torch.float32))
    half = 32
    fr = torch.exp(torch.arange(half) * -(np.log(10000) / (half - 1)))
    ang = t[:, None].float() * fr[None]
    ref = torch.cat([ang.sin(), ang.cos()], -1)
    assert (out.cpu() - ref).abs().max() <= 2e-4  # t=999: fp32 argument reduction differences
    # bilinear align_corners=True into a concat buffer (ddp.py:418,434)
    x = rnd("bx", (2, 16, 32, 48))
    buf = torch.zeros(2, 40, 2, 3, device=DEV)
    ops.bilinear_ac_into(nhwc(x, torch.float32), buf, 8)
    ref = F.interpolate(x, size=(2, 3), mode="bilinear", align_corners=True)
    assert (buf[:, 8:24].cpu() - ref).abs().max() <= 1e-5
    assert buf[:, :8].abs().max() == 0 and buf[:, 24:].abs().max() == 0
    # row softmax / group softmax
    s = rnd("s", (7, 1000), 5.0)
    p = ops.row_softmax(s.to(DEV), torch.float32)
    assert (p.cpu() - s.softmax(-1)).abs().max() <= 1e-6
    qq = rnd("q", (50, 128), 3.0)
    g = ops.group_softmax32_(qq.clone().to(DEV), 0.25)
    assert (g.cpu() - qq.reshape(50, 4, 32).softmax(-1).reshape(50, 128) * 0.25).abs().max() <= 1e-6
    # layout round trip
    z = rnd("z", (2, 24, 5, 6))
    assert torch.equal(ops.nhwc_to_nchw_f32(ops.nchw_to_nhwc(z.to(DEV), torch.float32)).cpu(), z)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 4, 16), (1, 4, 24), (2, 2, 200), (1, 4, 784), (1, 3, 1024)])
def test_flash_attn(dt, shape):
    """dig_flash_attn vs softmax(q k^T * scale) v in fp32 (attend.py:98-113), incl. ragged key/query tiles."""
    B, heads, n = shape
    hid = heads * 32
    qk = q(rnd("fa_qk", (B, n, 2 * hid), 2.0), dt)
    v = q(rnd("fa_v", (B, n, hid)), dt)
    qh = qk[..., :hid].reshape(B, n, heads, 32).permute(0, 2, 1, 3)
    kh = qk[..., hid:].reshape(B, n, heads, 32).permute(0, 2, 1, 3)
    vh = v.reshape(B, n, heads, 32).permute(0, 2, 1, 3)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * 32 ** -0.5, -1) @ vh).permute(0, 2, 1, 3).reshape(B, n, hid)
    out = torch.empty(B, n, hid, dtype=dt, device=DEV)
    vT = v.permute(0, 2, 1).contiguous().to(device=DEV, dtype=dt)
    call("flash_attn", qk=qk.to(device=DEV, dtype=dt), vT=vT, out=out, B=B, n=n, heads=heads, ldqk=2 * hid, dtype=dtype_code(dt), scale=32 ** -0.5)
    assert (out.float().cpu() - ref).abs().max() <= (2e-4 if dt == torch.float32 else 1.5e-2)


@pytest.mark.parametrize("shape", [(2, 32, 48, 0), (1, 16, 16, 0), (3, 64, 32, 0), (1, 112, 112, 0), (3, 64, 32, 3), (2, 48, 80, 1), (5, 32, 32, 2)])
def test_conv3x3_weight_stationary_64(shape, monkeypatch):
    """The persistent weight-stationary 64->64 bf16 variant (conv3x3_ws.hip): plain + GroupNorm partials (4 per tile),
    GroupNorm-apply + SiLU prologue, nearest-x2 addressing.  The last cases force a tiny grid (DIG_WS_GRID) so that
    every block walks many tiles, across image boundaries, with an odd tile count for one of the two wave groups."""
    dt = torch.bfloat16
    B, H, W, grid = shape
    if grid:
        monkeypatch.setenv("DIG_WS_GRID", str(grid))
    x = q(rnd("wsx", (B, 64, H, W)), dt)
    w, b = q(rnd("wsw", (64, 64, 3, 3), (3.0 / (9 * 64)) ** 0.5), dt), rnd("wsb", (64,), 0.1)
    cw = ops.ConvWeight(w.to(DEV), b.to(DEV), dt)
    ref = F.conv2d(x, w, b, padding=1)
    y, st = ops.conv3x3(nhwc(x, dt), cw)
    assert st.shape[2] == (H // 16) * (W // 16) * 4
    assert (nchw(y) - ref).abs().max() <= tol(dt, ref)
    st = st.cpu().double().sum(2)
    g = ref.reshape(B, 8, -1).double()
    assert torch.allclose(st[..., 0], g.sum(-1), rtol=1e-3, atol=1e-2 * g.shape[-1] ** 0.5)
    assert torch.allclose(st[..., 1], (g * g).sum(-1), rtol=2e-3)
    aff = torch.stack([1.0 + 0.3 * rnd("wsa", (B, 64)), 0.2 * rnd("wsbb", (B, 64))], -1)
    xin = F.silu(x * aff[..., 0][:, :, None, None] + aff[..., 1][:, :, None, None])
    ref = F.conv2d(q(xin, dt), w, b, padding=1)
    y, _ = ops.conv3x3(nhwc(x, dt), cw, aff=aff.to(DEV).contiguous())
    assert (nchw(y) - ref).abs().max() <= 2 * tol(dt, ref)
    if H <= 64:
        ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
        y, _ = ops.conv3x3(nhwc(x, dt), cw, up2=True, want_stats=False)
        assert (nchw(y) - ref).abs().max() <= tol(dt, ref)


@pytest.mark.parametrize("shape", [(2, 32, 48, 64, 0), (1, 16, 16, 64, 0), (3, 64, 32, 0, 0), (1, 112, 112, 64, 0), (3, 64, 32, 64, 3), (2, 48, 80, 0, 1),
                                   (5, 32, 32, 64, 2), (1, 16, 32, 0, 1)])
def test_conv3x3_register_stationary_128_to_64(shape, monkeypatch):
    """The persistent register-stationary 128->64 bf16 kernel (conv3x3_rs.hip): (x | skip) concat of two 64-channel tensors
    (C1 = 64) or one 128-channel tensor (C1 = 0), GroupNorm partials (one per tile and group), nearest-x2 addressing, border and
    interior tiles, against F.conv2d in fp32 - and against the generic kernel (DIG_NO_RS128) within one bf16 rounding.  The last
    cases force a tiny grid (DIG_RS_GRID) so that every block walks many tiles across image boundaries (K = 1 and odd K too)."""
    dt = torch.bfloat16
    B, H, W, C1, grid = shape
    C0 = 128 - C1
    if grid:
        monkeypatch.setenv("DIG_RS_GRID", str(grid))
    x0 = q(rnd("rsx0", (B, C0, H, W)), dt)
    x1 = q(rnd("rsx1", (B, C1, H, W)), dt) if C1 else None
    w, b = q(rnd("rsw", (64, 128, 3, 3), (3.0 / (9 * 128)) ** 0.5), dt), rnd("rsb", (64,), 0.1)
    cw = ops.ConvWeight(w.to(DEV), b.to(DEV), dt, C1=C1)
    ref = F.conv2d(torch.cat([x0, x1], 1) if C1 else x0, w, b, padding=1)
    y, st = ops.conv3x3(nhwc(x0, dt), cw, x1=nhwc(x1, dt) if C1 else None)
    assert st.shape[2] == (H // 16) * (W // 16)
    assert (nchw(y) - ref).abs().max() <= tol(dt, ref)
    g = ref.reshape(B, 8, -1).double()
    ssum = st.cpu().double().sum(2)
    assert torch.allclose(ssum[..., 0], g.sum(-1), rtol=1e-3, atol=1e-2 * g.shape[-1] ** 0.5)
    assert torch.allclose(ssum[..., 1], (g * g).sum(-1), rtol=2e-3)
    y2, st2 = ops.conv3x3(nhwc(x0, dt), cw, x1=nhwc(x1, dt) if C1 else None)
    assert torch.equal(y, y2) and torch.equal(st, st2)  # deterministic
    monkeypatch.setenv("DIG_NO_RS128", "1")
    yg, stg = ops.conv3x3(nhwc(x0, dt), cw, x1=nhwc(x1, dt) if C1 else None)
    monkeypatch.delenv("DIG_NO_RS128")
    assert (y.float() - yg.float()).abs().max().item() <= 2.0 ** -7 * max(1.0, float(ref.abs().max()))
    # (the generic kernel sums the fp32 accumulators, this one the stored bf16 values: rounding noise of ~1e4 values)
    assert torch.allclose(st.double().sum(2), stg.double().sum(2), rtol=1e-3, atol=0.5)
    if C1 == 0 and H <= 64:
        ref = F.conv2d(F.interpolate(x0, scale_factor=2, mode="nearest"), w, b, padding=1)
        y, _ = ops.conv3x3(nhwc(x0, dt), cw, up2=True, want_stats=False)
        assert tuple(y.shape) == (B, 2 * H, 2 * W, 64) and (nchw(y) - ref).abs().max() <= tol(dt, ref)


@pytest.mark.parametrize("shape", [(2, 32, 32, 64, 64, 64), (3, 16, 32, 128, 64, 128), (1, 64, 96, 64, 64, 64)])
def test_pw_res_tail(shape):
    """dig_pw_res (streaming ResnetBlock tail) vs conv1x1([x0|x1]) + SiLU(a*h2+b) in fp32, and vs the general GEMM path."""
    dt = torch.bfloat16
    B, H, W, K0, K1, N = shape
    x0, x1 = q(rnd("pwx0", (B, H, W, K0)), dt), q(rnd("pwx1", (B, H, W, K1)), dt)
    h2 = q(rnd("pwh2", (B, H, W, N), 2.0), dt)
    w = q(rnd("pww", (N, K0 + K1), (K0 + K1) ** -0.5), dt)
    bias = rnd("pwb", (N,), 0.1)
    aff = torch.stack([1.0 + 0.3 * rnd("pwa", (B, N)), 0.2 * rnd("pwab", (B, N))], -1).contiguous()
    ref = torch.cat([x0, x1], -1) @ w.t() + bias + F.silu(h2 * aff[:, None, None, :, 0] + aff[:, None, None, :, 1])
    d = lambda t: t.to(device=DEV, dtype=dt).contiguous()
    assert ops._lib.lib().dig_pw_res_supported(K0, K1, N, dtype_code(dt)) == 1
    y = ops.res_tail(d(x0), d(w), bias.to(DEV), d(x1), d(h2), aff.to(DEV), H * W)
    assert (y.float().cpu() - ref).abs().max() <= tol(dt, ref)
    y2 = ops.linear_rows(d(x0), d(w), bias=bias.to(DEV), x1=d(x1), gsrc=d(h2), gaff=aff.to(DEV), rows_per_img=H * W)
    assert (y.float() - y2.float()).abs().max().item() <= tol(dt, ref)
    if K0 + K1 == 128 and N == 64:
        # the Unet's final_conv (ddp.py:373,443-448) fused into the tail: fp32 NCHW result of conv1x1(N -> n_out) on the block
        # output, which is never stored
        for n_out in (3, 1, 4):
            fw, fb = rnd("pwfw", (n_out, N), N ** -0.5), rnd("pwfb", (n_out,), 0.1)
            assert ops.res_tail_final_supported(d(x0), d(x1), N, H * W, n_out)
            pred = torch.full((B, n_out, H, W), float("nan"), device=DEV)
            r = ops.res_tail(d(x0), d(w), bias.to(DEV), d(x1), d(h2), aff.to(DEV), H * W, final=(fw.to(DEV), fb.to(DEV), pred))
            assert r is None
            fref = (ref @ fw.t() + fb).permute(0, 3, 1, 2)
            # (the conv is applied to the block output as the unfused path stores it, i.e. rounded to bf16)
            assert (pred.cpu() - fref).abs().max() <= 5e-3 * max(1.0, float(fref.abs().max()))
            fref_b = (ref.to(dt).float() @ fw.t() + fb).permute(0, 3, 1, 2)
            assert (pred.cpu() - fref_b).abs().mean() <= 2e-4 * max(1.0, float(fref.abs().max()))


@pytest.mark.parametrize("dt", DTYPES)
def test_lfca_helper_kernels(dt):
    """dig_gn_stats + dig_gn_finalize + dig_affine_act_add (GroupNorm on a tensor that is not a conv output, with the
    activation placements of resnet101.py:35-38,214-222) and dig_bilinear_resize (F.interpolate align_corners=False)."""
    B, H, W, C = 2, 9, 7, 48
    x = q(rnd("lx", (B, C, H, W), 2.0), dt)
    res = q(rnd("lr", (B, C, H, W)), dt)
    gamma, beta = 1.0 + 0.2 * rnd("lg", (C,)), 0.1 * rnd("lb", (C,))
    xd, rd = nhwc(x, dt), nhwc(res, dt)
    for groups, tiles in ((1, 5), (3, 1), (6, 63)):
        st = torch.empty(B, groups, tiles, 2, dtype=torch.float32, device=DEV)
        call("gn_stats", x=xd, stats=st, B=B, HW=H * W, C=C, groups=groups, tiles=tiles, dtype=dtype_code(dt))
        aff = ops.gn_finalize(st, gamma.to(DEV), beta.to(DEV), H * W, eps=1e-6)
        gn = F.group_norm(x, groups, gamma, beta, 1e-6)
        for pre, post, fn in ((2, 0, lambda: F.relu(gn) + 0.5 * res), (0, 2, lambda: F.relu(gn + 0.5 * res)), (3, 1, lambda: F.silu(F.gelu(gn) + 0.5 * res))):
            out = torch.empty_like(xd)
            call("affine_act_add", x=xd, aff=aff, res=rd, out=out, B=B, HW=H * W, C=C, pre_act=pre, post_act=post, dtype=dtype_code(dt),
                 x_scale=1.0, res_scale=0.5)
            ref = fn()
            assert (nchw(out) - ref).abs().max() <= tol(dt, ref) * (1 if dt == torch.float32 else 2), (groups, pre, post)
    out = torch.empty_like(xd)
    call("affine_act_add", x=xd, aff=None, res=rd, out=out, B=B, HW=H * W, C=C, pre_act=0, post_act=0, dtype=dtype_code(dt), x_scale=0.5, res_scale=0.5)
    assert (nchw(out) - (x + res) / 2).abs().max() <= tol(dt, x)
    for ho, wo in ((9, 7), (16, 16), (5, 11), (18, 3)):
        out = torch.empty(B, ho, wo, C, dtype=dt, device=DEV)
        rr = q(rnd("lrr", (B, C, ho, wo)), dt)
        call("bilinear_resize", **{"in": xd}, res=nhwc(rr, dt), out=out, B=B, hi=H, wi=W, ho=ho, wo=wo, C=C, dtype=dtype_code(dt))
        ref = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=False) + rr
        assert (nchw(out) - ref).abs().max() <= (1e-5 if dt == torch.float32 else tol(dt, ref)), (ho, wo)


@pytest.mark.parametrize("cfg", [(2, 36, 3, 64, ""), (1, 216, 6, 64, ""), (2, 100, 2, 32, ""), (1, 330, 2, 64, ""), (3, 7, 1, 32, ""),
                                 (1, 733, 6, 64, ""), (2, 520, 3, 32, ""), (1, 928, 1, 64, ""), (1, 940, 1, 32, ""),
                                 (2, 36, 3, 64, "TILED"), (3, 7, 1, 32, "TILED"), (1, 97, 2, 64, "TILED"), (1, 128, 2, 32, "TILED"),
                                 (1, 330, 2, 64, "NO_MFMA"), (2, 36, 3, 64, "GACC"), (1, 733, 6, 64, "GACC"), (1, 97, 2, 32, "GACC"),
                                 (1, 1573, 6, 64, ""), (1, 2872, 2, 64, "")])
def test_vit_attn_rollout(cfg, monkeypatch):
    """dig_vit_attn (vision_transformer.py:87-103 + the head-mean / layer-sum of mctformer.py:302-315): output tokens and the
    two probability accumulators against torch, over two layers.  N <= 320 runs on the register-resident fp32 matrix-core
    kernel, 320 < N <= 928 (733 = the reference's 368x496 VOC crop with 20 class tokens) on the key-tiled one with the head-sum
    in LDS, beyond that (1573 / 2872 = the same crop at scales 1.5 / 2.0 of VOC12DatasetMS) on the key-tiled one that adds
    p / heads straight into the accumulators (GACC); ragged token counts, both head widths; the env hooks force the tiled /
    GACC / VALU kernels at small N (fewer key tiles than waves, a last tile with one key)."""
    B, N, heads, dh, hook = cfg
    if hook:
        monkeypatch.setenv("DIG_VIT_ATTN_" + hook, "1")
    ncls, D = min(5, N), heads * dh
    acc_all = torch.zeros(B, N, N, device=DEV)
    acc_cls = torch.zeros(B, ncls, N, device=DEV)
    ref_all = torch.zeros(B, N, N)
    for layer in range(2):
        qkv = rnd(f"va{layer}", (B, N, 3 * D), 1.5)
        q, k, v = qkv.reshape(B, N, 3, heads, dh).permute(2, 0, 3, 1, 4)
        p = ((q @ k.transpose(-2, -1)) * dh ** -0.5).softmax(-1)
        ref_o = (p @ v).transpose(1, 2).reshape(B, N, D)
        ref_all += p.mean(1)
        out = torch.empty(B, N, D, device=DEV)
        call("vit_attn", qkv=qkv.to(DEV), out=out, acc_all=acc_all, acc_cls=acc_cls if layer == 1 else None, B=B, N=N, heads=heads, dh=dh, ncls=ncls)
        assert (out.cpu() - ref_o).abs().max() <= 2e-5 * max(1.0, float(ref_o.abs().max()))
    assert (acc_all.cpu() - ref_all).abs().max() <= 2e-6
    assert (acc_cls.cpu() - p.mean(1)[:, :ncls]).abs().max() <= 2e-6


def test_head_and_loss_kernels():
    """csrc/losses.hip against torch on random inputs: head_pool (relu + NCHW transpose, top-3 pooling with ties and P < 3),
    row_mean (strided rows, two-input average), multilabel soft margin, the binary-split KL term, max_norm'd L1 with and
    without the in-kernel bilinear resize, sum_scale with accumulation."""
    g = lambda *s: rnd("hl" + "x".join(map(str, s)), s, 2.0)
    for B, P, C in ((2, 196, 20), (1, 713, 20), (3, 2, 5), (2, 64, 80)):
        x = g(B, P, C)
        x[0, : min(P, 5), 0] = 1.25  # ties inside the top 3
        relu = torch.empty(B, C, P, device=DEV)
        pooled = torch.empty(B, C, device=DEV)
        call("head_pool", **{"in": x.to(DEV)}, relu_nchw=relu, pooled=pooled, B=B, P=P, C=C)
        v = x.permute(0, 2, 1)
        k = min(3, P)
        ref = v.clamp_min(0).topk(k, dim=-1)[0].mean(-1) - (-v).clamp_min(0).topk(k, dim=-1)[0].mean(-1)
        assert torch.equal(relu.cpu(), v.clamp_min(0))
        assert (pooled.cpu() - ref).abs().max() <= 1e-6 * max(1.0, float(ref.abs().max()))
    B, C, D, N = 3, 20, 384, 216
    x = g(B, N, D)
    out = torch.empty(B, C, device=DEV)
    call("row_mean", x0=x.to(DEV), x1=None, out=out, bstride0=N * D, bstride1=0, B=B, C=C, D=D)
    assert (out.cpu() - x[:, :C].mean(-1)).abs().max() <= 1e-6
    y = g(B, C, D)
    call("row_mean", x0=x[:, :C].contiguous().to(DEV), x1=y.to(DEV), out=out, bstride0=C * D, bstride1=C * D, B=B, C=C, D=D)
    assert (out.cpu() - ((x[:, :C] + y) / 2).mean(-1)).abs().max() <= 1e-6
    for B, C in ((2, 20), (5, 80), (1, 3)):
        lx, ly = g(B, C) * 3, (g(B, C, 1)[..., 0] > 0).float()
        s, t = g(B, C, 2)[..., 0] * 4, g(B, C, 3)[..., 0] * 4
        out = torch.full((1,), 7.0, device=DEV)
        call("msm_loss", x=lx.to(DEV), y=ly.to(DEV), out=out, scale=1.5, accumulate=0, B=B, C=C)
        ref = 1.5 * F.multilabel_soft_margin_loss(lx, ly)
        assert abs(float(out) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref)))
        for Tt in (1.0, 5.0):
            call("kd_bin_loss", student=s.to(DEV), teacher=t.to(DEV), out=out, T=Tt, scale=2.0, accumulate=1, B=B, C=C)
            lg = lambda p: torch.log(torch.cat((p, 1 - p), dim=1))
            ref = ref + 2.0 * torch.nn.KLDivLoss(reduction="batchmean", log_target=True)(lg(F.softmax(s / Tt, 1)), lg(F.softmax(t / Tt, 1)))
        assert abs(float(out) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))

    def max_norm(c):
        c = F.relu(c)
        mx, mn = c.flatten(2).max(-1)[0][..., None, None], c.flatten(2).min(-1)[0][..., None, None]
        return F.relu(c - mn - 1e-5) / (mx - mn + 1e-5)

    for (ha, wa), (hb, wb), atl in (((14, 14), (14, 14), 1), ((16, 16), (14, 14), 1), ((14, 14), (14, 14), 0), ((23, 31), (12, 40), 0)):
        B, C = 2, 20
        a, b = g(B, C, ha, wa), g(B, C, hb, wb)
        lab = (g(B, C, 9)[..., 0] > 0).float()
        partial = torch.empty(B * C, device=DEV)
        call("maxnorm_l1", a=a.to(DEV), b=b.to(DEV), label=lab.to(DEV), partial=partial, B=B, C=C, ha=ha, wa=wa, hb=hb, wb=wb, a_times_label=atl)
        out = torch.empty(1, device=DEV)
        call("sum_scale", **{"in": partial}, out=out, scale=1.0 / (B * C * ha * wa), accumulate=0, n=B * C)
        l4 = lab.view(B, C, 1, 1)
        an = max_norm(a) * (l4 if atl else 1.0)
        ref = ((an - max_norm(F.interpolate(b, size=(ha, wa), mode="bilinear", align_corners=False))) * l4).abs().mean()
        assert abs(float(out) - float(ref)) <= 1e-6 * max(1.0, abs(float(ref))), ((ha, wa), (hb, wb), float(out), float(ref))


def test_conv3x3_random_shapes():
    """Seeded sweep over ragged shapes and channel combinations (every conv3x3 dispatch class: weight-stationary, double-buffered,
    generic bf16/fp32; concat / nearest-x2 / prologue), bf16 and fp32, against F.conv2d."""
    rng = np.random.RandomState(20261110)
    for it in range(24):
        dt = DTYPES[it % 2]
        Co = int(rng.choice([16, 32, 64, 128, 256]))
        C0 = int(rng.choice([16, 32, 64, 128, 192]))
        mode = it % 4  # 0 plain, 1 concat, 2 up2, 3 prologue
        C1 = int(rng.choice([16, 64])) if mode == 1 else 0
        B = int(rng.randint(1, 3))
        H, W = (int(rng.randint(1, 24)) * 2, int(rng.randint(1, 24)) * 2) if mode == 2 else (int(rng.randint(1, 45)), int(rng.randint(1, 45)))
        if it % 6 == 0:
            H, W, C0, Co, C1, mode = 16 * int(rng.randint(1, 4)), 16 * int(rng.randint(1, 4)), 64, 64, 0, 3 * (it % 12 == 0)  # weight-stationary class
        tag = f"r{it}"
        w, b = q(rnd(tag + "w", (Co, C0 + C1, 3, 3), (3.0 / (9 * (C0 + C1))) ** 0.5), dt), rnd(tag + "b", (Co,), 0.1)
        cw = ops.ConvWeight(w.to(DEV), b.to(DEV), dt, C1=C1)
        if mode == 2:
            xs = q(rnd(tag + "x", (B, C0, H // 2, W // 2)), dt)
            ref = F.conv2d(F.interpolate(xs, scale_factor=2, mode="nearest"), w, b, padding=1)
            y, _ = ops.conv3x3(nhwc(xs, dt), cw, up2=True, want_stats=False)
            k = 1
        else:
            x0 = q(rnd(tag + "x", (B, C0, H, W)), dt)
            x1 = q(rnd(tag + "y", (B, C1, H, W)), dt) if C1 else None
            aff, k = None, 1
            xin = torch.cat([x0, x1], 1) if C1 else x0
            if mode == 3:
                aff = torch.stack([1.0 + 0.3 * rnd(tag + "a", (B, C0)), 0.2 * rnd(tag + "c", (B, C0))], -1)
                xin = q(F.silu(x0 * aff[..., 0][:, :, None, None] + aff[..., 1][:, :, None, None]), dt)
                k = 1 if dt == torch.float32 else 2
            ref = F.conv2d(xin, w, b, padding=1)
            y, st = ops.conv3x3(nhwc(x0, dt), cw, x1=nhwc(x1, dt) if C1 else None, aff=aff.to(DEV).contiguous() if aff is not None else None)
            g = ref.reshape(B, 8, -1).double()
            assert torch.allclose(st.cpu().double().sum(2)[..., 0], g.sum(-1), rtol=1e-3, atol=2e-2 * g.shape[-1] ** 0.5), (it, "stats")
        assert (nchw(y) - ref).abs().max() <= k * tol(dt, ref), (it, dt, B, C0, C1, Co, H, W, mode)


def test_gemm_random_shapes():
    """Seeded sweep over M, N, K (ragged tiles, K with and without the multiple-of-32 that selects the LDS-DMA kernel), the concat
    loader, bias / residual / GELU epilogues, bf16 and fp32, against F.linear."""
    rng = np.random.RandomState(7)
    for it in range(24):
        dt = DTYPES[it % 2]
        M, N = int(rng.randint(1, 700)), int(rng.randint(1, 40)) * 8
        K0 = int(rng.choice([8, 32, 64, 96, 136, 256, 512]))
        K1 = int(rng.choice([0, 0, 32, 64]))
        tag = f"g{it}"
        x0 = q(rnd(tag + "x", (M, K0)), dt)
        x1 = q(rnd(tag + "y", (M, K1)), dt) if K1 else None
        w, b = q(rnd(tag + "w", (N, K0 + K1), (K0 + K1) ** -0.5), dt), rnd(tag + "b", (N,), 0.1)
        res = q(rnd(tag + "r", (M, N)), dt) if it % 3 == 0 else None
        gelu = it % 4 == 1
        xin = torch.cat([x0, x1], 1) if K1 else x0
        ref = F.linear(xin, w, b)
        if gelu:
            ref = F.gelu(ref)
        if res is not None:
            ref = ref + res
        dev = lambda t: t.to(device=DEV, dtype=dt)
        y = ops.linear_rows(dev(x0), dev(w), bias=b.to(DEV), x1=dev(x1) if K1 else None, res=dev(res) if res is not None else None, gelu=gelu)
        assert (y.float().cpu() - ref).abs().max() <= tol(dt, ref), (it, dt, M, N, K0, K1)


@pytest.mark.parametrize("B", [1, 5, 32, 70])
def test_time_embedding_fused(B):
    """dig_time_embed + dig_time_mlp (sinusoid -> Linear -> GELU -> Linear, SiLU, all block projections) vs plain torch fp32
    (ddp.py:118-130,316-321,172-175)."""
    dim, tdim, N = 64, 256, 8064 if B == 32 else 200
    t = torch.tensor([(37 * i + 100) % 1000 for i in range(B)], dtype=torch.long)
    w1, b1 = rnd("te.w1", (tdim, dim), dim ** -0.5), rnd("te.b1", (tdim,), 0.1)
    w3, b3 = rnd("te.w3", (tdim, tdim), tdim ** -0.5), rnd("te.b3", (tdim,), 0.1)
    wm, bm = rnd("te.wm", (N, tdim), tdim ** -0.5), rnd("te.bm", (N,), 0.1)
    temb_ref = O.time_embedding({"1.weight": w1, "1.bias": b1, "3.weight": w3, "3.bias": b3}, t, dim)
    ss_ref = F.linear(F.silu(temb_ref), wm, bm)
    d = lambda x: x.to(DEV).contiguous()
    temb, st = torch.empty(B, tdim, device=DEV), torch.empty(B, tdim, device=DEV)
    call("time_embed", t=d(t), w1=d(w1), b1=d(b1), w3=d(w3), b3=d(b3), temb=temb, st=st, B=B, dim=dim, tdim=tdim)
    ss = torch.full((B, N), float("nan"), device=DEV)
    call("time_mlp", st=st, w=d(wm), bias=d(bm), out=ss, B=B, N=N, K=tdim)
    assert (temb.cpu() - temb_ref).abs().max() <= 2e-5
    assert (st.cpu() - F.silu(temb_ref)).abs().max() <= 2e-5
    assert (ss.cpu() - ss_ref).abs().max() <= 5e-5


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(3, 64, 16), (2, 128, 1), (1, 256, 32), (2, 16, 5)])
def test_la_merge_with_mb(dt, cfg):
    """dig_la_merge with the fused M_b output vs the stand-alone merge followed by the batched GEMM it replaces
    (M_b[c][h*32+d] = sum_e Wout[c][h*32+e] ctx[h][d][e], ddp.py:231-232): identical context, M_b equal up to the storage
    rounding of a different fp32 summation order."""
    B, dim, splits = cfg
    part = rnd("mm.part", (B, splits, 4, 32, 32)).to(DEV)
    pmax = rnd("mm.pmax", (B, splits, 4, 32), 3.0).to(DEV)
    psum = (rnd("mm.psum", (B, splits, 4, 32)).abs() + 0.5).to(DEV)
    wo = q(rnd("mm.wo", (dim, 128), 128 ** -0.5), dt).to(device=DEV, dtype=dt)
    ctx = torch.empty(B * 4, 32, 32, dtype=dt, device=DEV)
    call("la_merge", part=part, pmax=pmax, psum=psum, ctx=ctx, B=B, splits=splits, dtype=dtype_code(dt))
    mb_ref = torch.einsum("che,bhde->bchd", wo.float().reshape(dim, 4, 32), ctx.float().reshape(B, 4, 32, 32)).reshape(B, dim, 128)
    ctx2 = torch.empty_like(ctx)
    mb = torch.full((B, dim, 128), float("nan"), dtype=dt, device=DEV)
    call("la_merge", part=part, pmax=pmax, psum=psum, ctx=ctx2, B=B, splits=splits, dtype=dtype_code(dt), wo=wo, mb=mb, dim=dim)
    assert torch.equal(ctx, ctx2)
    assert (mb.float() - mb_ref).abs().max().item() <= (1e-5 if dt == torch.float32 else 2.0 ** -8) * max(1.0, float(mb_ref.abs().max()))
    mb3 = torch.empty_like(mb)
    call("la_merge", part=part, pmax=pmax, psum=psum, ctx=None, B=B, splits=splits, dtype=dtype_code(dt), wo=wo, mb=mb3, dim=dim)
    assert torch.equal(mb, mb3)
