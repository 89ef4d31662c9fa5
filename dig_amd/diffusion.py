"""MI355X-native drop-in for the DiG diffusion-guidance forward: ``Unet``, ``GaussianDiffusion``, ``Attend``.

Public surface and ``state_dict`` keys follow the reference (denoising_diffusion_pytorch/
denoising_diffusion_pytorch.py:268-448, 493-634, 853-913 and attend.py:35-113) so its checkpoints load
unchanged, but the modules below are parameter containers only: every forward is executed by the
hand-written gfx950 kernels in dig_amd/csrc through the C ABI of include/dig_hip.h.  There is no
CPU/eager fallback - calling a forward off-device or without libdig_hip.so raises.

Data flow of one Unet forward (activations NHWC, bf16 or fp32 storage, fp32 accumulate/statistics):
  stem kernel      : normalise + q_sample + conv7x7                       (one pass, writes x_t too)
  time MLP         : sinusoid -> 2 linears -> ONE GEMM for all 19 per-block (scale|shift) projections
  ResnetBlock      : conv3x3 [+concat] -> GN partials -> finalize -> conv3x3 with GN+SS+SiLU fused into
                     its loader -> finalize -> (GN+SiLU + residual) fused into res_conv GEMM / one ew pass
  LinearAttention  : RMSNorm -> q GEMM + 32-wide softmax; k,v projected channel-major -> context on MFMA
                     with exp fused -> out GEMM -> to_out GEMM -> RMSNorm + residual
  Attention        : RMSNorm -> qk GEMM, v channel-major -> QK^T GEMM (fp32 logits) -> row softmax -> PV GEMM
                     -> to_out GEMM + residual
  guidance maps    : bilinear(align_corners=True) written straight into the [B,1024,h,w] concat buffer
"""
import math

import os

import torch
from torch import nn

from . import ops
from ._lib import _pver, _tver, call, dtype_code

__all__ = ["Unet", "GaussianDiffusion", "Attend", "sigmoid_beta_schedule"]


# --------------------------------------------------------------------------------------------------
# parameter containers (names chosen to reproduce the reference state_dict keys)
# --------------------------------------------------------------------------------------------------
class _NoForward(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} is a parameter container; run the enclosing Unet on a HIP device")


class _Marker(_NoForward):
    """Parameter-free placeholder that keeps nn.Sequential indices aligned with the reference."""


class RMSNorm(_NoForward):
    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))


class Block(_NoForward):
    def __init__(self, dim, dim_out, groups=8):
        super().__init__()
        self.proj = nn.Conv2d(dim, dim_out, 3, padding=1)
        self.norm = nn.GroupNorm(groups, dim_out)


class ResnetBlock(_NoForward):
    def __init__(self, dim, dim_out, *, time_emb_dim=None, groups=8):
        super().__init__()
        self.mlp = nn.Sequential(_Marker(), nn.Linear(time_emb_dim, dim_out * 2)) if time_emb_dim is not None else None
        self.block1 = Block(dim, dim_out, groups)
        self.block2 = Block(dim_out, dim_out, groups)
        self.res_conv = nn.Conv2d(dim, dim_out, 1) if dim != dim_out else _Marker()


class LinearAttention(_NoForward):
    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.heads, self.dim_head = heads, dim_head
        hidden = heads * dim_head
        self.norm = RMSNorm(dim)
        self.to_qkv = nn.Conv2d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Sequential(nn.Conv2d(hidden, dim, 1), RMSNorm(dim))


class Attention(_NoForward):
    def __init__(self, dim, heads=4, dim_head=32, flash=False):
        super().__init__()
        self.heads, self.dim_head = heads, dim_head
        hidden = heads * dim_head
        self.norm = RMSNorm(dim)
        self.attend = Attend(flash=flash)
        self.to_qkv = nn.Conv2d(dim, hidden * 3, 1, bias=False)
        self.to_out = nn.Conv2d(hidden, dim, 1)


class Attend(nn.Module):
    """attend.py:35-113.  forward(q, k, v) with [b, h, n, d] tensors -> [b, h, n, d], softmax(q k^T / sqrt(d)) v.
    `flash` is accepted for API compatibility; both settings run the same HIP kernels."""

    def __init__(self, dropout=0., flash=False):
        super().__init__()
        self.dropout, self.flash = dropout, flash

    def forward(self, q, k, v):
        if self.training and self.dropout > 0:
            raise NotImplementedError("attention dropout is a training feature outside the guidance path")
        assert q.is_cuda, "dig_amd.Attend runs on the HIP device only"
        b, h, n, d = q.shape
        nk = k.shape[-2]
        assert d % 8 == 0 and nk % 8 == 0, "head dim and key length must be multiples of 8"
        dt = q.dtype
        q, k = q.contiguous(), k.contiguous()
        vT = v.transpose(-1, -2).contiguous()  # [b,h,d,nk]: PV contracts over keys, so keys must be contiguous
        S = torch.empty(b * h, n, nk, dtype=torch.float32, device=q.device)
        ops.gemm_nt(q, k, S, n, nk, d, dtype=dt, alpha=d ** -0.5, out_f32=True, batch=b * h,
                    sa=(n * d, 0), sb=(nk * d, 0), sc=(n * nk, 0))
        P = ops.row_softmax(S, dt)
        out = torch.empty(b, h, n, d, dtype=dt, device=q.device)
        ops.gemm_nt(P, vT, out, n, d, nk, dtype=dt, batch=b * h, sa=(n * nk, 0), sb=(d * nk, 0), sc=(n * d, 0))
        return out


def _down(dim, dim_out):
    return nn.Sequential(_Marker(), nn.Conv2d(dim * 4, dim_out, 1))


def _up(dim, dim_out):
    return nn.Sequential(_Marker(), nn.Conv2d(dim, dim_out, 3, padding=1))


class Unet(nn.Module):
    """Same constructor as the reference Unet (ddp.py:268-286).  Supported on the HIP path: the DiG
    configuration family - channels=3, no self-conditioning / learned variance / learned sinusoidal
    embedding, dim a multiple of 16, attn_dim_head=32."""

    def __init__(self, dim, init_dim=None, out_dim=None, dim_mults=(1, 2, 4, 8), channels=3, self_condition=False,
                 resnet_block_groups=8, learned_variance=False, learned_sinusoidal_cond=False,
                 random_fourier_features=False, learned_sinusoidal_dim=16, attn_dim_head=32, attn_heads=4,
                 full_attn=(False, False, False, True), flash_attn=False):
        super().__init__()
        if self_condition or learned_variance or learned_sinusoidal_cond or random_fourier_features:
            raise NotImplementedError("only the DiG Unet configuration (plain sinusoidal time embedding, "
                                      "no self-conditioning, no learned variance) has a HIP path")
        if channels != 3 or attn_dim_head != 32 or dim % 16 != 0 or (init_dim not in (None, dim)):
            raise NotImplementedError("HIP path needs channels=3, attn_dim_head=32, init_dim=dim, dim % 16 == 0")
        self.channels, self.self_condition = channels, self_condition
        self.random_or_learned_sinusoidal_cond = False
        self.dim, self.groups = dim, resnet_block_groups
        self.init_conv = nn.Conv2d(channels, dim, 7, padding=3)
        dims = [dim] + [dim * m for m in dim_mults]
        pairs = list(zip(dims[:-1], dims[1:]))
        n = len(pairs)
        tdim = dim * 4
        self.time_mlp = nn.Sequential(_Marker(), nn.Linear(dim, tdim), _Marker(), nn.Linear(tdim, tdim))
        as_tuple = lambda v: v if isinstance(v, tuple) else (v,) * n
        full_attn, attn_heads = as_tuple(full_attn), as_tuple(attn_heads)
        assert len(full_attn) == n
        self.full_attn, self.heads = full_attn, attn_heads
        rb = lambda i, o: ResnetBlock(i, o, time_emb_dim=tdim, groups=resnet_block_groups)
        att = lambda full, d, hd: Attention(d, heads=hd, flash=flash_attn) if full else LinearAttention(d, heads=hd)
        self.downs = nn.ModuleList()
        for i, (ci, co) in enumerate(pairs):
            last = i == n - 1
            self.downs.append(nn.ModuleList([rb(ci, ci), rb(ci, ci), att(full_attn[i], ci, attn_heads[i]),
                                             _down(ci, co) if not last else nn.Conv2d(ci, co, 3, padding=1)]))
        mid = dims[-1]
        self.mid_block1 = rb(mid, mid)
        self.mid_attn = Attention(mid, heads=attn_heads[-1], flash=flash_attn)
        self.mid_block2 = rb(mid, mid)
        self.ups = nn.ModuleList()
        for j, i in enumerate(reversed(range(n))):
            ci, co = pairs[i]
            last = j == n - 1
            self.ups.append(nn.ModuleList([rb(co + ci, co), rb(co + ci, co), att(full_attn[i], co, attn_heads[i]),
                                           _up(co, ci) if not last else nn.Conv2d(co, ci, 3, padding=1)]))
        self.out_dim = out_dim if out_dim is not None else channels
        self.final_res_block = rb(dim * 2, dim)
        self.final_conv = nn.Conv2d(dim, self.out_dim, 1)
        self.compute_dtype = torch.bfloat16  # storage type of activations/weights on the HIP path
        self._packed = None

    @property
    def downsample_factor(self):
        return 2 ** (len(self.downs) - 1)

    # ---- weight packing ---------------------------------------------------------------------
    def set_compute_dtype(self, dtype):
        assert dtype in (torch.bfloat16, torch.float32)
        self.compute_dtype = dtype
        self._packed = None
        return self

    def _pack_key(self):
        p = next(self.parameters())
        return (p.device, self.compute_dtype, _pver(self.parameters()), p.data_ptr())

    def packed(self):
        key = self._pack_key()
        if self._packed is None or self._packed["key"] != key:
            self._packed = _pack_unet(self, self.compute_dtype)
            self._packed["key"] = key
        return self._packed

    # ---- forward ----------------------------------------------------------------------------
    def forward(self, x, time, x_self_cond=None):
        """ddp.py:379-448: x [B,3,H,W] fp32 (already noised), time [B] int64.
        Returns pred if self.training else (pred, [5 guidance maps])."""
        assert x_self_cond is None, "self-conditioning is not part of the DiG configuration"
        pred, feats, _ = _unet_forward(self, x, None, time, None, None, (1.0,) * 3, (0.0,) * 3, want_xt=False)
        return pred if self.training else (pred, feats)


def _conv3_pack(conv, dt, C1=0):
    """-> ops.ConvWeight; C1 = channels of the second half when the conv reads a fused (x | skip) concat."""
    return ops.ConvWeight(conv.weight, conv.bias, dt, C1=C1)


def _conv1_pack(conv, dt):
    w = conv.weight.detach().reshape(conv.weight.shape[0], -1).contiguous().to(dt)
    return w, (conv.bias.detach().float().contiguous() if conv.bias is not None else None)


def _pack_unet(u, dt):
    f32 = lambda t: t.detach().float().contiguous()
    P = {"dtype": dt}
    sw = u.init_conv.weight.detach().float().permute(0, 2, 3, 1).reshape(u.dim, 147)  # [Cout][(ky,kx,c)]
    P["stem_w"] = torch.nn.functional.pad(sw, (0, 13)).contiguous().to(dt)  # K padded to 160
    P["stem_b"] = f32(u.init_conv.bias)
    P["t1"] = (f32(u.time_mlp[1].weight), f32(u.time_mlp[1].bias))
    P["t3"] = (f32(u.time_mlp[3].weight), f32(u.time_mlp[3].bias))
    mlps_w, mlps_b, off = [], [], [0]

    def rb(m, C1=0):
        d = {}
        d["w1"] = _conv3_pack(m.block1.proj, dt, C1)
        d["g1"], d["be1"] = f32(m.block1.norm.weight), f32(m.block1.norm.bias)
        d["w2"] = _conv3_pack(m.block2.proj, dt)
        d["g2"], d["be2"] = f32(m.block2.norm.weight), f32(m.block2.norm.bias)
        d["eps"] = m.block1.norm.eps
        d["res"] = _conv1_pack(m.res_conv, dt) if isinstance(m.res_conv, nn.Conv2d) else None
        d["ss_off"] = off[0]
        mlps_w.append(f32(m.mlp[1].weight))
        mlps_b.append(f32(m.mlp[1].bias))
        off[0] += m.mlp[1].weight.shape[0]
        return d

    def attn(m):
        d = {"full": isinstance(m, Attention), "heads": m.heads}
        hid = m.heads * m.dim_head
        wqkv, _ = _conv1_pack(m.to_qkv, dt)
        d["g"] = f32(m.norm.g.reshape(-1))
        if d["full"]:
            d["wqk"], d["wv"] = wqkv[:2 * hid].contiguous(), wqkv[2 * hid:].contiguous()
            d["wo"], d["bo"] = _conv1_pack(m.to_out, dt)
        else:
            d["wq"], d["wkv"] = wqkv[:hid].contiguous(), wqkv[hid:].contiguous()
            wg = (m.to_qkv.weight.detach().float().reshape(3 * hid, -1) * m.norm.g.detach().float().reshape(1, -1))
            d["wq_g"], d["wkv_g"] = wg[:hid].contiguous().to(dt), wg[hid:].contiguous().to(dt)  # RMSNorm gain folded in
            d["wo"], d["bo"] = _conv1_pack(m.to_out[0], dt)
            d["g2"] = f32(m.to_out[1].g.reshape(-1))
        return d

    def stage(blocks, is_down, skip_c=(0, 0)):
        b1, b2, at, smp = blocks
        d = {"rb1": rb(b1, skip_c[0]), "rb2": rb(b2, skip_c[1]), "attn": attn(at)}
        if isinstance(smp, nn.Conv2d):
            d["kind"] = "conv"
            d["w"] = _conv3_pack(smp, dt)
            d["cout"] = smp.weight.shape[0]
        elif is_down:
            d["kind"] = "down"
            w = smp[1].weight.detach()
            co, c4 = w.shape[0], w.shape[1]
            d["w"] = w.reshape(co, c4 // 4, 2, 2).permute(0, 2, 3, 1).reshape(co, c4).contiguous().to(dt)
            d["b"] = f32(smp[1].bias)
            d["cout"] = co
        else:
            d["kind"] = "up"
            d["w"] = _conv3_pack(smp[1], dt)
            d["cout"] = smp[1].weight.shape[0]
        return d

    P["downs"] = [stage(s, True) for s in u.downs]
    P["mid1"], P["mid_attn"], P["mid2"] = rb(u.mid_block1), attn(u.mid_attn), rb(u.mid_block2)
    # skip connections popped by the ups path: stage j reads (x | h.pop()) twice; both skips of a level have dim_in channels
    n = len(u.downs)
    dims = [u.dim] + [u.downs[i][3][1].weight.shape[0] if i < n - 1 else u.downs[i][3].weight.shape[0] for i in range(n)]
    P["ups"] = [stage(s, False, (dims[n - 1 - j], dims[n - 1 - j])) for j, s in enumerate(u.ups)]
    P["final_rb"] = rb(u.final_res_block, u.dim)
    P["final_w"], P["final_b"] = _conv1_pack(u.final_conv, dt)
    P["final_w32"] = u.final_conv.weight.detach().float().reshape(u.final_conv.weight.shape[0], -1).contiguous()
    P["mlp_w"], P["mlp_b"] = torch.cat(mlps_w, 0).contiguous(), torch.cat(mlps_b, 0).contiguous()
    dev = u.init_conv.weight.device
    P["one"] = torch.ones(1, dtype=torch.float32, device=dev)
    P["zero"] = torch.zeros(1, dtype=torch.float32, device=dev)
    return P


# --------------------------------------------------------------------------------------------------
# forward building blocks (all on the current HIP stream)
# --------------------------------------------------------------------------------------------------
# Block2's GroupNorm + SiLU on its input (ddp.py:159-166): applied by the conv kernel's prologue as the halo is staged, or - for maps of
# at most this many elements per image (the 32x32 levels: 256 and 512 channels) - by one in-place elementwise pass before a
# plain conv.  The prologue costs a wide layer 22 us (every N-tile block transforms the same halo), the pass 13 us there; on
# larger maps the pass costs more than the prologue (measured A/B at B = 32: 9.93 ms with 524288, 9.99 without, 9.97 with
# 1048576, 10.05 with 2097152).  A function of the per-image shape only, never of the batch: results do not depend on B.
_GN_INPLACE_MAX = int(os.environ.get("DIG_GN_INPLACE_MAX", "524288"))
# finalize inside the elementwise launch that consumes it where the statistics are few partials per image (ops.gn_fin_silu_add)
_GN_FUSED_FIN = os.environ.get("DIG_GN_FUSED_FIN", "1") != "0"


def _resnet(p, x0, x1, ss_all, debug=None, final=None):
    """ResnetBlock (ddp.py:169-194); x1 = second half of a fused channel concat (or None).
    final = (fw, fb, pred): fuse the Unet's final_conv into the block tail (ops.res_tail); returns None then."""
    hw = x0.shape[1] * x0.shape[2]
    c1, st1 = ops.conv3x3(x0, p["w1"], x1=x1)
    if _GN_INPLACE_MAX and c1.shape[1] * c1.shape[2] * c1.shape[3] <= _GN_INPLACE_MAX:
        if _GN_FUSED_FIN and ops.gn_fin_supported(st1, c1.shape[3]):
            h1 = ops.gn_fin_silu_add(c1, st1, p["g1"], p["be1"], hw, ss=ss_all, ss_off=p["ss_off"], eps=p["eps"], inplace=True)
        else:
            h1 = ops.gn_silu_add(c1, ops.gn_finalize(st1, p["g1"], p["be1"], hw, ss=ss_all, ss_off=p["ss_off"], eps=p["eps"]), inplace=True)
        c2, st2 = ops.conv3x3(h1, p["w2"])
    else:
        aff1 = ops.gn_finalize(st1, p["g1"], p["be1"], hw, ss=ss_all, ss_off=p["ss_off"], eps=p["eps"])
        c2, st2 = ops.conv3x3(c1, p["w2"], aff=aff1)
    if p["res"] is None:
        assert x1 is None
        if _GN_FUSED_FIN and ops.gn_fin_supported(st2, c2.shape[3]):
            return ops.gn_fin_silu_add(c2, st2, p["g2"], p["be2"], hw, eps=p["eps"], res=x0)
        return ops.gn_silu_add(c2, ops.gn_finalize(st2, p["g2"], p["be2"], hw, eps=p["eps"]), res=x0)
    aff2 = ops.gn_finalize(st2, p["g2"], p["be2"], hw, eps=p["eps"])
    wr, br = p["res"]
    return ops.res_tail(x0, wr, br, x1, c2, aff2, hw, final=final)


def _linear_attention_unfused(p, x):
    """LinearAttention (ddp.py:196-233) + the caller's residual, as separate GEMM / softmax / context kernels
    (used when the fused kernels' shape constraints do not hold)."""
    B, H, W, C = x.shape
    n, heads = H * W, p["heads"]
    hid = heads * 32
    dt = x.dtype
    assert n % 8 == 0, "H*W must be a multiple of 8 at every linear-attention level"
    xn = ops.rmsnorm(x, p["g"])
    # q projection with q.softmax(dim=-2) * scale fused into the GEMM epilogue (ddp.py:224,227)
    q = ops.linear_rows(xn, p["wq"], gsm_scale=32 ** -0.5)  # [B,H,W,hid] pixel-major
    kvT = torch.empty(B, 2 * hid, n, dtype=dt, device=x.device)  # channel-major k | v
    ops.gemm_nt(p["wkv"], xn, kvT, 2 * hid, n, C, dtype=dt, batch=B, sb=(n * C, 0), sc=(2 * hid * n, 0))
    kmax = torch.empty(B * hid, dtype=torch.float32, device=x.device)
    call("row_max", **{"in": kvT}, out=kmax, rows=B * hid, group_stride=2 * hid * n, group_rows=hid, n=n, dtype=dtype_code(dt))
    chunks = (n + 63) // 64
    splits = max(1, min(32, chunks // 8))
    part = torch.empty(B * heads, splits, 32, 32, dtype=torch.float32, device=x.device)
    ksum = torch.empty(B * heads, splits, 32, dtype=torch.float32, device=x.device)
    esz = kvT.element_size()
    call("la_context", k=kvT.data_ptr(), v=kvT.data_ptr() + hid * n * esz, kmax=kmax, part=part, ksum=ksum,
         b_stride_k=2 * hid * n, h_stride_k=32 * n, b_stride_v=2 * hid * n, h_stride_v=32 * n,
         B=B, heads=heads, n=n, splits=splits, dtype=dtype_code(dt))
    ctx = torch.empty(B * heads, 32, 32, dtype=dt, device=x.device)  # [bh][d][e]
    call("la_finalize", part=part, ksum=ksum, ctxT=ctx, BH=B * heads, splits=splits, dtype=dtype_code(dt), transposed=1, scale=1.0)
    # second einsum (ddp.py:231) and to_out conv folded into ONE per-image matrix:  y = Wout . blockdiag(ctx^T) . q = M_b . q,
    # M_b[c][h*32+d] = sum_e Wout[c][h*32+e] * ctx[h][d][e]   (tiny batched GEMM), then y = q @ M_b^T + bias.
    dim = p["wo"].shape[0]
    Mb = torch.empty(B, dim, hid, dtype=dt, device=x.device)
    ops.gemm_nt(p["wo"], ctx, Mb, dim, 32, 32, dtype=dt, lda0=hid, ldb=32, ldc=hid, batch=B * heads, z_inner=heads,
                sa=(0, 32), sb=(heads * 1024, 1024), sc=(dim * hid, 32))
    y = torch.empty(B, H, W, dim, dtype=dt, device=x.device)
    ops.gemm_nt(q, Mb, y, n, dim, hid, dtype=dt, bias=p["bo"], batch=B, sa=(n * hid, 0), sb=(dim * hid, 0), sc=(n * dim, 0))
    return ops.rmsnorm(y, p["g2"], res=x)


def la_splits(n, C=0, dtype=None):
    """Pixel splits of the fused LinearAttention's pass A for an image of n pixels (independent of the batch size).
    bf16, C = 64 at >= 8192 pixels in whole 32-pixel tiles runs the wave-private kernel, whose splits are blocks of four
    waves (each wave a run of tiles, merged in the block): 16 of them per image."""
    if C == 64 and dtype == torch.bfloat16 and n >= 8192 and n % 32 == 0:
        return 16
    return max(1, min(32, ((n + 63) // 64) // 4))


def _linear_attention(p, x):
    """LinearAttention (ddp.py:196-233) + the caller's residual (ddp.py:402,430), fused: pass A (k, v, online softmax,
    context) -> merge + M_b -> pass B (q, softmax, M_b q, RMSNorm, residual).  x is read twice, the result written once."""
    B, H, W, C = x.shape
    n, heads = H * W, p["heads"]
    dim = p["wo"].shape[0]
    if heads != 4 or n % 32 != 0 or C not in (16, 32, 64, 128, 256) or dim != C:
        return _linear_attention_unfused(p, x)
    dt = x.dtype
    dc = dtype_code(dt)
    tiles = (n + 63) // 64
    # pixel splits of pass A: a function of n ONLY, so that an image goes through the same fp32 merge order whatever batch
    # it sits in (a rank's shard of a batch is then bitwise equal to the same images inside the full batch)
    splits = la_splits(n, C, dt)
    part = torch.empty(B, splits, 4, 32, 32, dtype=torch.float32, device=x.device)
    pmax = torch.empty(B, splits, 4, 32, dtype=torch.float32, device=x.device)
    psum = torch.empty(B, splits, 4, 32, dtype=torch.float32, device=x.device)
    call("la_kvctx", x=x, wkv=p["wkv_g"], part=part, pmax=pmax, psum=psum, B=B, n=n, C=C, hid=128, splits=splits, dtype=dc)
    # merge of the splits and M_b = Wout . blockdiag(ctx^T) in ONE launch (the context itself never goes to HBM)
    Mb = torch.empty(B, dim, 128, dtype=dt, device=x.device)
    call("la_merge", part=part, pmax=pmax, psum=psum, ctx=None, B=B, splits=splits, dtype=dc, wo=p["wo"], mb=Mb, dim=dim)
    out = torch.empty_like(x)
    call("la_out", x=x, wq=p["wq_g"], mb=Mb, bias=p["bo"], g2=p["g2"], out=out, B=B, n=n, C=C, dim=dim, dtype=dc, scale=32 ** -0.5)
    return out


def _full_attention(p, x):
    """Attention (ddp.py:235-264) via Attend (attend.py:98-113) + the caller's residual (ddp.py:402,412,430)."""
    B, H, W, C = x.shape
    n, heads = H * W, p["heads"]
    hid = heads * 32
    dt = x.dtype
    assert n % 8 == 0, "H*W must be a multiple of 8 at every full-attention level"
    xn = ops.rmsnorm(x, p["g"])
    qk = ops.linear_rows(xn, p["wqk"])  # [B,H,W,2*hid]: q | k, pixel-major
    vT = torch.empty(B, hid, n, dtype=dt, device=x.device)
    ops.gemm_nt(p["wv"], xn, vT, hid, n, C, dtype=dt, batch=B, sb=(n * C, 0), sc=(hid * n, 0))
    att = torch.empty(B, H, W, hid, dtype=dt, device=x.device)
    call("flash_attn", qk=qk, vT=vT, out=att, B=B, n=n, heads=heads, ldqk=2 * hid, dtype=dtype_code(dt), scale=32 ** -0.5)
    return ops.linear_rows(att, p["wo"], bias=p["bo"], res=x)


def _time_embedding(u, P, t):
    """ddp.py:118-130,316-321 and every block's SiLU+Linear (ddp.py:172-175,186-188); always fp32 (tiny).  Two launches:
    sinusoid + both Linears + GELU/SiLU per image, then all 19 block projections as one skinny GEMM."""
    B = t.shape[0]
    tdim = P["t3"][0].shape[0]
    okw = lambda k: k >= 4 and k % 4 == 0 and (k % 256 == 0 or (k <= 256 and (k // 4) & (k // 4 - 1) == 0))
    if not (okw(u.dim) and okw(tdim)) or tdim % 64:  # widths the fused kernels do not take (dig_time_embed, dig_time_mlp)
        return _time_embedding_unfused(u, P, t)
    temb = torch.empty(B, tdim, dtype=torch.float32, device=t.device)
    st = torch.empty_like(temb)
    call("time_embed", t=t, w1=P["t1"][0], b1=P["t1"][1], w3=P["t3"][0], b3=P["t3"][1], temb=temb, st=st, B=B, dim=u.dim, tdim=tdim)
    N = P["mlp_w"].shape[0]
    ss = torch.empty(B, N, dtype=torch.float32, device=t.device)
    call("time_mlp", st=st, w=P["mlp_w"], bias=P["mlp_b"], out=ss, B=B, N=N, K=tdim)
    return temb, ss  # [B, sum 2*C] fp32


def _time_embedding_unfused(u, P, t):
    """The same chain on the general kernels (widths the fused kernels do not take)."""
    B = t.shape[0]
    emb = torch.empty(B, u.dim, dtype=torch.float32, device=t.device)
    call("sinusoid", t=t, out=emb, B=B, dim=u.dim, dtype=dtype_code(torch.float32))
    e = ops.linear_rows(emb, P["t1"][0], bias=P["t1"][1], gelu=True)
    temb = ops.linear_rows(e, P["t3"][0], bias=P["t3"][1])
    st = torch.empty_like(temb)
    call("act", **{"in": temb}, out=st, n=temb.numel(), act=0, in_f32=1, out_f32=1, dtype=dtype_code(torch.float32))
    return temb, ops.linear_rows(st, P["mlp_w"], bias=P["mlp_b"])  # [B, sum 2*C] fp32


def _same_device(module, dev, what):
    """Raise (before any kernel launch) unless every parameter and buffer of `module` lives on `dev`: the kernels take raw
    pointers, so a CPU-resident or other-GPU weight would be a GPU fault, not an exception."""
    for name, p in list(module.named_parameters()) + list(module.named_buffers()):
        if p.device != dev:
            raise RuntimeError(f"dig_amd.{what}: parameter {name!r} lives on {p.device} but the input on {dev}; "
                               f"move the module with .to('{dev}') first (no CPU fallback)")


def _mark_t_checked(t, num_timesteps):
    """Remember on the tensor OBJECT (not its storage address, which the allocator reuses) that its current version holds
    valid timesteps; an in-place write bumps _version and voids the mark.  Called directly only on tensors this package has
    just built from Python ints and never writes again (also under torch.inference_mode(), where there is no version
    counter); _check_t does not mark a caller's inference tensor, which is re-validated on every call instead."""
    t._dig_t_ok = (_tver(t), num_timesteps)
    return t


def _check_t(t, num_timesteps, B):
    """extract() in the reference (ddp.py:452-455) is a gather that raises on an out-of-range index; here the tables are
    indexed by the kernels directly, so validate on the host.  Reading a device tensor back costs a stream sync, so a
    validated tensor object is marked: a loop that reuses its `t` (bench.py, a serving loop) pays once, and
    dig_amd.guidance marks the tensors it builds from Python ints."""
    if t.dim() != 1 or t.shape[0] != B:
        raise ValueError(f"t must have shape [{B}], got {tuple(t.shape)}")
    if not B or getattr(t, "_dig_t_ok", None) == (_tver(t), num_timesteps):
        return
    lo, hi = (int(v) for v in torch.aminmax(t))
    if lo < 0 or hi >= num_timesteps:
        raise IndexError(f"timestep out of range: t must lie in [0, {num_timesteps}), got [{lo}, {hi}]")
    if not t.is_inference():  # an inference tensor can be rewritten in place without a trace: never cache its validation
        _mark_t_checked(t, num_timesteps)


def _unet_forward(u, img, noise, t, sqrt_ac, sqrt_1mac, pre_scale, pre_shift, want_xt, debug=None, want_pred=True):
    """Shared body of Unet.forward and GaussianDiffusion.p_losses.  img [B,3,H,W] fp32 NCHW.
    With noise=None the stem runs with (1, 0) coefficients, i.e. x_t = pre(img).
    want_pred=False skips final_res_block + final_conv (ddp.py:443-448) and returns pred = None: the guidance loop discards
    the model output of its five low-t passes (model.py:302,312-314) and the maps do not depend on it."""
    if not img.is_cuda:
        raise RuntimeError("dig_amd.Unet runs on the HIP device only (no CPU fallback); move inputs with .cuda()")
    P = u.packed()
    if P.get("device_checked") != img.device:  # once per packing: a move or an in-place update of any parameter re-packs
        _same_device(u, img.device, "Unet")
        P["device_checked"] = img.device
    for name, buf in (("noise", noise), ("sqrt_alphas_cumprod", sqrt_ac), ("sqrt_one_minus_alphas_cumprod", sqrt_1mac)):
        if buf is not None and buf.device != img.device:
            raise RuntimeError(f"dig_amd.Unet: {name} lives on {buf.device} but the input on {img.device}; move the "
                               "GaussianDiffusion module with .to(device) first")
    B, Cin, H, W = img.shape
    f = u.downsample_factor
    assert H % f == 0 and W % f == 0, f"your input dimensions {(H, W)} need to be divisible by {f}, given the unet"
    dt = P["dtype"]
    dev = img.device
    img = img.contiguous().float()
    t = t.to(device=dev, dtype=torch.long).contiguous()
    if noise is None:
        noise, tt, ca, cs = img, torch.zeros_like(t), P["one"], P["zero"]
    else:
        noise, tt, ca, cs = noise.contiguous().float(), t, sqrt_ac, sqrt_1mac
    x_t = torch.empty_like(img) if want_xt else None
    x = torch.empty(B, H, W, u.dim, dtype=dt, device=dev)
    call("qsample_stem", img=img, noise=noise, t=tt, sqrt_ac=ca, sqrt_1mac=cs, w=P["stem_w"], bias=P["stem_b"],
         x_t=x_t, out=x, pre_scale=pre_scale, pre_shift=pre_shift, B=B, H=H, W=W, Cout=u.dim, dtype=dtype_code(dt))
    keep = (lambda k, v: debug.__setitem__(k, v)) if debug is not None else (lambda k, v: None)
    keep("init_conv", x)
    r = x
    temb, ss = _time_embedding(u, P, t)
    keep("temb", temb)
    skips = []
    for i, s in enumerate(P["downs"]):
        x = _resnet(s["rb1"], x, None, ss)
        skips.append(x)
        x = _resnet(s["rb2"], x, None, ss)
        x = _full_attention(s["attn"], x) if s["attn"]["full"] else _linear_attention(s["attn"], x)
        skips.append(x)
        keep(f"downs.{i}.attn", x)
        if s["kind"] == "down":
            x = ops.downsample_conv(x, s["w"], s["b"])
        else:
            x, _ = ops.conv3x3(x, s["w"], want_stats=False)
        keep(f"downs.{i}.out", x)
    x = _resnet(P["mid1"], x, None, ss)
    x = _full_attention(P["mid_attn"], x)
    x = _resnet(P["mid2"], x, None, ss)
    keep("mid", x)
    h, w = H // 16, W // 16
    chans = [x.shape[-1]] + [s["cout"] for s in P["ups"]]
    maps = torch.empty(B, sum(chans), h, w, dtype=torch.float32, device=dev)
    ops.bilinear_ac_into(x, maps, 0)
    coff = chans[0]
    for j, s in enumerate(P["ups"]):
        x = _resnet(s["rb1"], x, skips.pop(), ss)
        x = _resnet(s["rb2"], x, skips.pop(), ss)
        x = _full_attention(s["attn"], x) if s["attn"]["full"] else _linear_attention(s["attn"], x)
        keep(f"ups.{j}.attn", x)
        x, _ = ops.conv3x3(x, s["w"], up2=(s["kind"] == "up"), want_stats=False)
        keep(f"ups.{j}.out", x)
        ops.bilinear_ac_into(x, maps, coff)
        coff += chans[j + 1]
    pred = None
    if want_pred:
        pred = torch.empty(B, u.out_dim, H, W, dtype=torch.float32, device=dev)
        fr = P["final_rb"]
        if fr["res"] is not None and ops.res_tail_final_supported(x, r, fr["res"][0].shape[0], H * W, u.out_dim):
            # final_conv applied inside the tail kernel of final_res_block: the block's output never goes to HBM
            _resnet(fr, x, r, ss, final=(P["final_w32"], P["final_b"], pred))
        else:
            x = _resnet(fr, x, r, ss)
            keep("final_res_block", x)
            ops.gemm_nt(x, P["final_w"], pred, B * H * W, u.out_dim, u.dim, dtype=dt, bias=P["final_b"], out_f32=True, nchw_hw=H * W)
    feats, o = [], 0
    for c in chans:
        feats.append(maps[:, o:o + c])
        o += c
    u._last_maps = maps  # contiguous [B, sum C, h, w] == torch.cat(feats, 1)
    return pred, feats, x_t


# --------------------------------------------------------------------------------------------------
# GaussianDiffusion (schedule + q_sample + p_losses eval + predict_start_from_v)
# --------------------------------------------------------------------------------------------------
def sigmoid_beta_schedule(timesteps, start=-3, end=3, tau=1):
    """ddp.py:478-491, computed in fp64."""
    t = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64) / timesteps
    v_start, v_end = torch.tensor(start / tau).sigmoid(), torch.tensor(end / tau).sigmoid()
    ac = (-((t * (end - start) + start) / tau).sigmoid() + v_end) / (v_end - v_start)
    ac = ac / ac[0]
    return torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)


class GaussianDiffusion(nn.Module):
    """ddp.py:493-634, 853-913 - the subset DiG calls (model.py:132-143, 288, 302-306): schedule buffers with the
    reference's names, normalize/unnormalize, q_sample, p_losses (eval branch), predict_start_from_v, and the truncated
    DDIM sampler sample_from_t / ddim_sample_from_t (ddp.py:770-831).  The ancestral samplers and the training loss are
    outside the guidance path and are not provided."""

    def __init__(self, model, *, image_size, timesteps=1000, sampling_timesteps=None, objective='pred_v',
                 beta_schedule='sigmoid', schedule_fn_kwargs=dict(), ddim_sampling_eta=0., auto_normalize=True,
                 offset_noise_strength=0., min_snr_loss_weight=False, min_snr_gamma=5):
        super().__init__()
        assert model.channels == model.out_dim
        assert not model.random_or_learned_sinusoidal_cond
        if objective != 'pred_v' or beta_schedule != 'sigmoid':
            raise NotImplementedError("the DiG checkpoint uses objective='pred_v' with the sigmoid schedule")
        self.model = model
        self.channels, self.self_condition = model.channels, model.self_condition
        self.image_size, self.objective = image_size, objective
        betas = sigmoid_beta_schedule(timesteps, **schedule_fn_kwargs)
        alphas = 1. - betas
        ac = torch.cumprod(alphas, dim=0)
        acp = torch.cat([torch.ones(1, dtype=torch.float64), ac[:-1]])
        self.num_timesteps = int(betas.shape[0])
        self.sampling_timesteps = sampling_timesteps if sampling_timesteps is not None else timesteps
        assert self.sampling_timesteps <= timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        self.offset_noise_strength = offset_noise_strength
        pv = betas * (1. - acp) / (1. - ac)
        snr = ac / (1 - ac)
        clipped = snr.clone()
        if min_snr_loss_weight:
            clipped.clamp_(max=min_snr_gamma)
        for name, val in [
            ('betas', betas), ('alphas_cumprod', ac), ('alphas_cumprod_prev', acp),
            ('sqrt_alphas_cumprod', ac.sqrt()), ('sqrt_one_minus_alphas_cumprod', (1. - ac).sqrt()),
            ('log_one_minus_alphas_cumprod', (1. - ac).log()), ('sqrt_recip_alphas_cumprod', (1. / ac).sqrt()),
            ('sqrt_recipm1_alphas_cumprod', (1. / ac - 1).sqrt()), ('posterior_variance', pv),
            ('posterior_log_variance_clipped', pv.clamp(min=1e-20).log()),
            ('posterior_mean_coef1', betas * acp.sqrt() / (1. - ac)),
            ('posterior_mean_coef2', (1. - acp) * alphas.sqrt() / (1. - ac)),
            ('loss_weight', clipped / (snr + 1)),
        ]:
            self.register_buffer(name, val.to(torch.float32))
        self.auto_normalize = auto_normalize

    @property
    def device(self):
        return self.betas.device

    def normalize(self, img):
        return img * 2 - 1 if self.auto_normalize else img

    def unnormalize(self, t):
        return (t + 1) * 0.5 if self.auto_normalize else t

    def _axpby(self, x, y, t, sign, post_scale=(1.0,) * 3, post_shift=(0.0,) * 3):
        if not x.is_cuda:
            raise RuntimeError("dig_amd.GaussianDiffusion runs on the HIP device only (no CPU fallback)")
        B, C, H, W = x.shape
        assert C == 3 and B > 0 and H * W > 0
        if self.sqrt_alphas_cumprod.device != x.device or y.device != x.device:
            raise RuntimeError(f"dig_amd.GaussianDiffusion: schedule buffers on {self.sqrt_alphas_cumprod.device}, operands on "
                               f"{x.device} / {y.device}; move the module with .to(device) first")
        _check_t(t, self.num_timesteps, B)
        t = t.to(x.device)
        out = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
        call("sched_axpby", x=x.contiguous().float(), y=y.contiguous().float(), t=t.to(torch.long).contiguous(),
             sqrt_ac=self.sqrt_alphas_cumprod, sqrt_1mac=self.sqrt_one_minus_alphas_cumprod, out=out,
             post_scale=post_scale, post_shift=post_shift, sign=sign, B=B, HW=H * W)
        return out

    def q_sample(self, x_start, t, noise=None):
        """ddp.py:853-860."""
        if noise is None:
            noise = torch.randn_like(x_start)
        return self._axpby(x_start, noise, t, 1.0)

    def predict_start_from_v(self, x_t, t, v):
        """ddp.py:630-634."""
        return self._axpby(x_t, v, t, -1.0)

    def ddim_time_pairs(self, num_step):
        """ddp.py:775-782: (time, time_next) pairs of the DDIM schedule truncated at num_step; the last pair ends at -1."""
        slice_idx = int((num_step / self.num_timesteps) * self.sampling_timesteps)
        times = torch.linspace(-1, self.num_timesteps - 1, steps=self.sampling_timesteps + 1)[:slice_idx]
        times = list(reversed(times.int().tolist()))
        return list(zip(times[:-1], times[1:]))

    @torch.no_grad()
    def ddim_sample_from_t(self, shape, img, num_step, return_all_timesteps=False, noise=None):
        """ddp.py:770-819: q_sample `img` to timestep num_step, then DDIM back to t = 0 with clip_x_start and the re-derived
        noise prediction (ddp.py:645-668, 'pred_v').  Each step is one Unet forward on the HIP kernels + one dig_ddim_step
        launch; the step's schedule scalars are formed on the host in fp32 from a host copy of alphas_cumprod exactly as
        the reference forms them from its 0-dim tensors.  `noise` (optional) is the q_sample noise the reference draws
        (ddp.py:856); per-step noise is drawn only when ddim_sampling_eta > 0 (with eta = 0 the reference multiplies it by 0)."""
        if self.model.training:
            raise NotImplementedError("sampling runs the forward-only Unet; call .eval() first")
        if not img.is_cuda:
            raise RuntimeError("dig_amd.GaussianDiffusion runs on the HIP device only (no CPU fallback)")
        B = shape[0]
        assert tuple(img.shape) == tuple(shape), (tuple(img.shape), tuple(shape))
        if not 0 <= int(num_step) < self.num_timesteps:
            raise IndexError(f"timestep out of range: num_step must lie in [0, {self.num_timesteps}), got {num_step}")
        pairs = self.ddim_time_pairs(num_step)
        dev = img.device
        tab = {k: getattr(self, k).detach().cpu() for k in ("alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                                                             "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod")}
        eta = self.ddim_sampling_eta
        t0 = _mark_t_checked(torch.full((B,), int(num_step), device=dev, dtype=torch.long), self.num_timesteps)
        x = self.q_sample(img, t0, noise)
        stack = None
        if return_all_timesteps:
            stack = torch.empty(B, len(pairs) + 1, *img.shape[1:], dtype=torch.float32, device=dev)
            stack[:, 0].copy_(self._unnorm_copy(x))
        out_u = None
        for k, (time, time_next) in enumerate(pairs):
            tc = _mark_t_checked(torch.full((B,), time, device=dev, dtype=torch.long), self.num_timesteps)
            v, _, _ = _unet_forward(self.model, x, None, tc, None, None, (1.0,) * 3, (0.0,) * 3, want_xt=False)
            last = time_next < 0
            san = c = sigma = 0.0
            nz = None
            if not last:
                alpha, alpha_next = tab["alphas_cumprod"][time], tab["alphas_cumprod"][time_next]
                sg = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
                san, c, sigma = float(alpha_next.sqrt()), float((1 - alpha_next - sg ** 2).sqrt()), float(sg)
                if sigma != 0.0:
                    nz = torch.randn_like(x)
            nxt = torch.empty_like(x)
            final = k == len(pairs) - 1
            if stack is not None:
                out_u = torch.empty_like(x)
            elif final:
                out_u = torch.empty_like(x) if self.auto_normalize else None
            call("ddim_step", x=x, v=v.contiguous(), noise=nz, out=nxt, out_unnorm=out_u if self.auto_normalize else None,
                 sqrt_ac=float(tab["sqrt_alphas_cumprod"][time]), sqrt_1mac=float(tab["sqrt_one_minus_alphas_cumprod"][time]),
                 sqrt_recip=float(tab["sqrt_recip_alphas_cumprod"][time]), sqrt_recipm1=float(tab["sqrt_recipm1_alphas_cumprod"][time]),
                 sqrt_alpha_next=san, c=c, sigma=sigma, last=int(last), n=x.numel())
            x = nxt
            if stack is not None:
                stack[:, k + 1].copy_(out_u if self.auto_normalize else x)
        if stack is not None:
            return stack
        if not pairs:
            return self._unnorm_copy(x)
        return out_u if self.auto_normalize else x

    def _unnorm_copy(self, x):
        """unnormalize() of a device tensor through dig_sched_axpby's post-affine: 0.5 * (1 * x + 0 * x) + 0.5."""
        if not self.auto_normalize:
            return x.clone()
        t = _mark_t_checked(torch.zeros(x.shape[0], device=x.device, dtype=torch.long), self.num_timesteps)
        out = torch.empty_like(x)
        P = self.model.packed()
        call("sched_axpby", x=x, y=x, t=t, sqrt_ac=P["one"], sqrt_1mac=P["zero"], out=out, post_scale=(0.5,) * 3,
             post_shift=(0.5,) * 3, sign=1.0, B=x.shape[0], HW=x.shape[2] * x.shape[3])
        return out

    def sample_from_t(self, batch_size=16, img=None, step=None, return_all_timesteps=False, noise=None):
        """ddp.py:827-831.  Only the DDIM branch (sampling_timesteps < timesteps, the DiG configuration, model.py:135) is built."""
        if not self.is_ddim_sampling:
            raise NotImplementedError("p_sample_loop_from_t (ancestral sampling) is not built; construct with sampling_timesteps < timesteps")
        return self.ddim_sample_from_t((batch_size, self.channels, self.image_size, self.image_size), img, step,
                                       return_all_timesteps=return_all_timesteps, noise=noise)

    def p_losses(self, x_start, t, noise=None, offset_noise_strength=None, pre_scale=(1.0,) * 3, pre_shift=(0.0,) * 3, want_pred=True):
        """Eval branch of ddp.py:862-913: returns (x_t, model_out, feat_out).  The discarded MSE of the reference
        (ddp.py:896-909) is not computed.  pre_scale/pre_shift fuse a per-channel affine on x_start into the stem
        (used by dig_amd.guidance to fold denormforDiff + normalize).  want_pred=False (not a reference argument): model_out is
        None and the final ResnetBlock + 1x1 conv that only feed it are skipped - for callers that use the maps alone."""
        if self.model.training:
            raise NotImplementedError("the training loss is outside the guidance path; call .eval() first")
        if noise is None:
            noise = torch.randn_like(x_start)
        ons = offset_noise_strength if offset_noise_strength is not None else self.offset_noise_strength
        if ons > 0.:
            noise = noise + ons * torch.randn(x_start.shape[:2], device=x_start.device)[:, :, None, None]
        if x_start.is_cuda:
            _check_t(t, self.num_timesteps, x_start.shape[0])
        pred, feats, x_t = _unet_forward(self.model, x_start, noise, t, self.sqrt_alphas_cumprod,
                                         self.sqrt_one_minus_alphas_cumprod, pre_scale, pre_shift, want_xt=True, want_pred=want_pred)
        return x_t, pred, feats

    def forward(self, img, *args, **kwargs):
        """ddp.py:915-925."""
        b, c, h, w = img.shape
        assert h == self.image_size and w == self.image_size, f'height and width of image must be {self.image_size}'
        t = torch.randint(0, self.num_timesteps, (b,), device=img.device).long()
        return self.p_losses(img, t, *args, pre_scale=(2.0,) * 3 if self.auto_normalize else (1.0,) * 3,
                             pre_shift=(-1.0,) * 3 if self.auto_normalize else (0.0,) * 3, **kwargs)
