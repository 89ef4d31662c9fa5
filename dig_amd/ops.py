"""Thin tensor-level wrappers over the C-ABI kernels (dig_amd/_lib.py).  Activations are NHWC torch tensors
([B,H,W,C], bf16 or fp32) living on the HIP device; torch is used only to own memory and streams."""
import torch

from . import _lib
from ._lib import call, dtype_code

GN_GROUPS = 8


def _chk(t):
    assert t.is_cuda and t.is_contiguous(), "dig_amd ops need contiguous HIP-device tensors"
    return t


class ConvWeight:
    """A conv3x3 weight packed as the kernel's per-chunk LDS slabs: [n_tile][chunk][9][BN][KC] (zero padded),
    chunks = ceil(C0/KC) + ceil(C1/KC) for a (C0 | C1) channel concat.  (BN, KC) come from dig_conv3x3_config."""

    def __init__(self, w_oihw, bias, dtype, C1=0):
        import ctypes
        Co, Cin = w_oihw.shape[0], w_oihw.shape[1]
        C0 = Cin - C1
        bn, kc = ctypes.c_int(), ctypes.c_int()
        rc = _lib.lib().dig_conv3x3_config(Co, dtype_code(dtype), ctypes.byref(bn), ctypes.byref(kc))
        assert rc == 0
        BN, KC = bn.value, kc.value
        w = w_oihw.detach().float().permute(0, 2, 3, 1).reshape(Co, 9, Cin)  # [Co][tap][Cin]
        pieces = []
        for lo, hi in ((0, C0), (C0, Cin)):
            if hi > lo:
                pc = w[:, :, lo:hi]
                pad = (-pc.shape[2]) % KC
                pieces.append(torch.nn.functional.pad(pc, (0, pad)))
        w = torch.cat(pieces, 2)
        nch = w.shape[2] // KC
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, (-Co) % BN))  # pad Cout to a multiple of BN
        nt = w.shape[0] // BN
        layout = _lib.lib().dig_conv3x3_wlayout(C0, C1, Co, dtype_code(dtype))
        if layout == 2:
            w = _pack_pairs(w, nt, nch, BN, KC)
        else:
            w = w.reshape(nt, BN, 9, nch, KC).permute(0, 3, 2, 1, 4).contiguous()
            if layout == 1:
                assert KC == 16
                rows = (torch.arange(BN, device=w.device) >> 3) & 1 == 1
                w[:, :, :, rows, :] = torch.cat([w[:, :, :, rows, 8:], w[:, :, :, rows, :8]], -1)
        self.w = w.to(dtype)
        self.bias = bias.detach().float().contiguous()
        self.Cout, self.C0, self.C1, self.dtype = Co, C0, C1, dtype


def _pack_pairs(w, nt, nch, BN, KC):
    """The LDS image of conv3x3_db.hip (dig_conv3x3_wlayout == 2; K = 32 of one v_mfma_f32_16x16x32_bf16 = two taps x 16
    channels).  w: [nt*BN][9 taps][nch*KC] fp32 -> [nt][pair P][main(2P) | main(2P+1) | w8(P)]:
      main(c): [4 K steps q][128 rows][4 slots][8]  slot s = tap 2q + (s >> 1), channel half s & 1 of chunk c
      w8(P)  : [128 rows][4 slots][8]               slot s = tap 8 of chunk 2P + (s >> 1), channel half s & 1
    row rho = mt*16 + rr holds output channel 32*(mt >> 1) + 8*(rr >> 2) + 4*(mt & 1) + (rr & 3); slot s sits at position
    s ^ f[(rho >> 2) & 3], f = (0, 2, 3, 1).  An odd chunk count is padded with a zero chunk."""
    assert BN in (64, 128) and KC == 16
    dev = w.device
    npairs = (nch + 1) // 2
    w = torch.nn.functional.pad(w, (0, npairs * 32 - nch * KC))
    rho = torch.arange(BN, device=dev)
    mt, rr = rho >> 4, rho & 15
    ch = 32 * (mt >> 1) + 8 * (rr >> 2) + 4 * (mt & 1) + (rr & 3)
    wp = w.reshape(nt, BN, 9, npairs, 2, 2, 8)[:, ch]  # [nt][rho][tap][P][chunk in pair][half][8]
    main = wp[:, :, :8].reshape(nt, BN, 4, 2, npairs, 2, 2, 8).permute(0, 4, 5, 2, 1, 3, 6, 7)  # [nt][P][cip][q][rho][tsel][half][8]
    main = main.reshape(nt, npairs, 2, 4, BN, 4, 8)
    w8 = wp[:, :, 8].permute(0, 2, 1, 3, 4, 5).reshape(nt, npairs, BN, 4, 8)  # [nt][P][rho][(cip, half)][8]
    f = torch.tensor([0, 2, 3, 1], device=dev)[(rho >> 2) & 3]
    src = torch.arange(4, device=dev)[None, :] ^ f[:, None]  # physical slot s' of row rho holds logical slot s' ^ f(rho)
    main = torch.gather(main, 5, src[None, None, None, None, :, :, None].expand_as(main))
    w8 = torch.gather(w8, 3, src[None, None, :, :, None].expand_as(w8))
    return torch.cat([main.reshape(nt, npairs, -1), w8.reshape(nt, npairs, -1)], 2).contiguous()


def conv3x3(x0, cw, x1=None, aff=None, up2=False, want_stats=True, groups=GN_GROUPS):
    """x0,x1: [B,h,w,C] NHWC; cw: ConvWeight; returns (raw conv output [B,H,W,Cout], GroupNorm partials or None)."""
    B, h, ww, C0 = x0.shape
    H, W = (2 * h, 2 * ww) if up2 else (h, ww)
    C1 = x1.shape[-1] if x1 is not None else 0
    assert (cw.C0, cw.C1) == (C0, C1) and cw.dtype == x0.dtype, ((cw.C0, cw.C1), (C0, C1), cw.dtype, x0.dtype)
    out = torch.empty(B, H, W, cw.Cout, dtype=x0.dtype, device=x0.device)
    stats = None
    if want_stats:
        tiles = _lib.lib().dig_conv3x3_tiles(H, W, C0, C1, cw.Cout, groups, dtype_code(x0.dtype))
        stats = torch.empty(B, groups, tiles, 2, dtype=torch.float32, device=x0.device)
    call("conv3x3", in0=_chk(x0), in1=_chk(x1) if x1 is not None else None, aff0=aff, w=cw.w, bias=cw.bias, out=out,
         stats=stats, in0_bstride=0, B=B, H=H, W=W, C0=C0, C1=C1, Cout=cw.Cout, groups=groups, up2=int(up2),
         dtype=dtype_code(x0.dtype))
    return out, stats


def gn_finalize(stats, gamma, beta, hw, ss=None, ss_off=0, eps=1e-5):
    """stats [B,G,tiles,2] -> aff [B,C,2].  ss: [B,ld] fp32 rows with (scale|shift) of this block at ss_off."""
    B, G, tiles, _ = stats.shape
    C = gamma.numel()
    aff = torch.empty(B, C, 2, dtype=torch.float32, device=stats.device)
    call("gn_finalize", stats=stats, gamma=gamma, beta=beta, ss=ss, aff=aff, B=B, C=C, groups=G, tiles=tiles, hw=hw,
         ss_ld=ss.shape[1] if ss is not None else 0, ss_off=ss_off, eps=eps)
    return aff


def gn_silu_add(x, aff, res=None, inplace=False):
    B, H, W, C = x.shape
    out = x if inplace else torch.empty_like(x)  # elementwise: out may alias x
    call("gn_silu_add", x=_chk(x), aff=aff, res=_chk(res) if res is not None else None, out=out, B=B, HW=H * W, C=C,
         dtype=dtype_code(x.dtype))
    return out


_GN_FIN_MAX_TILES = None


def gn_fin_supported(stats, C):
    """whether gn_fin_silu_add takes these statistics (few partials per image and group, C <= 1024)"""
    global _GN_FIN_MAX_TILES
    if _GN_FIN_MAX_TILES is None:
        _GN_FIN_MAX_TILES = int(_lib.lib().dig_gn_fin_max_tiles())
    return stats.shape[2] <= _GN_FIN_MAX_TILES and C <= 1024


def gn_fin_silu_add(x, stats, gamma, beta, hw, ss=None, ss_off=0, eps=1e-5, res=None, inplace=False):
    """gn_finalize + gn_silu_add in one launch (stats [B,G,tiles,2] with tiles <= dig_gn_fin_max_tiles): out = silu(gn(x)) [+ res]"""
    B, H, W, C = x.shape
    _, G, tiles, _ = stats.shape
    out = x if inplace else torch.empty_like(x)
    call("gn_fin_silu_add", x=_chk(x), res=_chk(res) if res is not None else None, out=out, stats=stats, gamma=gamma, beta=beta, ss=ss,
         B=B, HW=H * W, C=C, dtype=dtype_code(x.dtype), groups=G, tiles=tiles, hw=hw, ss_ld=ss.shape[1] if ss is not None else 0,
         ss_off=ss_off, eps=eps)
    return out


def rmsnorm(x, g, res=None):
    C = x.shape[-1]
    out = torch.empty_like(x)
    call("rmsnorm", x=_chk(x), g=g, res=_chk(res) if res is not None else None, out=out, rows=x.numel() // C, C=C,
         dtype=dtype_code(x.dtype))
    return out


def gemm_nt(a0, b, c, M, N, K, *, dtype, a1=None, K0=0, lda0=None, lda1=0, ldb=None, ldc=None, bias=None, res=None, ldr=0,
            gsrc=None, gaff=None, ldg=0, rows_per_img=0, alpha=1.0, gelu=False, out_f32=False, batch=1, z_inner=1,
            sa=(0, 0), sa1=(0, 0), sb=(0, 0), sc=(0, 0), sr=(0, 0), unshuffle=None, nchw_hw=0, gsm_scale=0.0):
    """Raw access to dig_gemm_nt; strides are in elements.  a0/b/c may be tensors or (tensor, element_offset)."""

    def ptr(x):
        if x is None:
            return None
        if isinstance(x, tuple):
            t, off = x
            return t.data_ptr() + off * t.element_size()
        return x.data_ptr()

    kw = dict(a0=ptr(a0), a1=ptr(a1), b=ptr(b), c=ptr(c), bias=bias, res=ptr(res), gsrc=ptr(gsrc), gaff=gaff,
              sa_o=sa[0], sa_i=sa[1], sa1_o=sa1[0], sa1_i=sa1[1], sb_o=sb[0], sb_i=sb[1], sc_o=sc[0], sc_i=sc[1],
              sr_o=sr[0], sr_i=sr[1], M=M, N=N, K=K, K0=K0, lda0=lda0 if lda0 is not None else K, lda1=lda1,
              ldb=ldb if ldb is not None else K, ldc=ldc if ldc is not None else N, ldr=ldr, ldg=ldg, batch=batch,
              z_inner=z_inner, a_mode=0, unsh_Ho=0, unsh_Wo=0, unsh_C=0, c_mode=0, hw=0, rows_per_img=rows_per_img,
              gelu=int(gelu), out_f32=int(out_f32), dtype=dtype_code(dtype), alpha=float(alpha), gsm_scale=float(gsm_scale))
    if unshuffle is not None:
        kw.update(a_mode=1, unsh_Ho=unshuffle[0], unsh_Wo=unshuffle[1], unsh_C=unshuffle[2])
    if nchw_hw:
        kw.update(c_mode=1, hw=nchw_hw)
    call("gemm_nt", **kw)


def linear_rows(x, w, bias=None, res=None, gelu=False, out_f32=False, x1=None, gsrc=None, gaff=None, rows_per_img=0, gsm_scale=0.0):
    """y[rows, N] = [x | x1] @ w^T (+bias, +silu(gn(gsrc)), gelu, +res).  x: [..., K0] contiguous; w: [N, K]."""
    K0 = x.shape[-1]
    K1 = x1.shape[-1] if x1 is not None else 0
    N, K = w.shape
    assert K == K0 + K1
    M = x.numel() // K0
    odt = torch.float32 if out_f32 else x.dtype
    y = torch.empty(*x.shape[:-1], N, dtype=odt, device=x.device)
    gemm_nt(_chk(x), _chk(w), y, M, N, K, dtype=x.dtype, a1=_chk(x1) if x1 is not None else None, K0=K0, lda0=K0, lda1=K1,
            bias=bias, res=_chk(res) if res is not None else None, ldr=N, gsrc=_chk(gsrc) if gsrc is not None else None,
            gaff=gaff, ldg=N, rows_per_img=rows_per_img, gelu=gelu, out_f32=out_f32, gsm_scale=gsm_scale)
    return y


def res_tail_final_supported(x0, x1, N, rows_per_img, n_out):
    """Whether res_tail can apply the Unet's final 1x1 conv to the block output in the same kernel (dig_pw_res, fout)."""
    K0 = x0.shape[-1]
    K1 = x1.shape[-1] if x1 is not None else 0
    M = x0.numel() // K0
    return (M % 32 == 0 and rows_per_img % 32 == 0 and K0 + K1 == 128 and N == 64 and 1 <= n_out <= 4
            and bool(_lib.lib().dig_pw_res_supported(K0, K1, N, dtype_code(x0.dtype))))


def res_tail(x0, w, bias, x1, gsrc, gaff, rows_per_img, final=None):
    """ResnetBlock tail with a channel-changing shortcut: res_conv([x0 | x1]) + SiLU(GroupNorm-apply(gsrc)).
    Streaming kernel where the shape is supported (small K, N), otherwise the general GEMM with the same epilogue.
    final = (fw [n_out, N] fp32, fb [n_out] fp32, pred [B, n_out, H, W] fp32): the block is the Unet's final_res_block and
    final_conv is applied in the same kernel (caller checks res_tail_final_supported); the block output is then never stored
    and None is returned."""
    K0 = x0.shape[-1]
    K1 = x1.shape[-1] if x1 is not None else 0
    N = w.shape[0]
    M = x0.numel() // K0
    if final is not None:
        fw, fb, pred = final
        call("pw_res", x0=_chk(x0), x1=_chk(x1) if x1 is not None else None, w=_chk(w), bias=bias, gsrc=_chk(gsrc), gaff=gaff,
             out=None, M=M, K0=K0, K1=K1, N=N, rows_per_img=rows_per_img, dtype=dtype_code(x0.dtype),
             fw=_chk(fw), fb=_chk(fb), fout=_chk(pred), f_no=fw.shape[0])
        return None
    if M % 32 == 0 and rows_per_img % 32 == 0 and _lib.lib().dig_pw_res_supported(K0, K1, N, dtype_code(x0.dtype)):
        out = torch.empty(*x0.shape[:-1], N, dtype=x0.dtype, device=x0.device)
        call("pw_res", x0=_chk(x0), x1=_chk(x1) if x1 is not None else None, w=_chk(w), bias=bias, gsrc=_chk(gsrc), gaff=gaff,
             out=out, M=M, K0=K0, K1=K1, N=N, rows_per_img=rows_per_img, dtype=dtype_code(x0.dtype))
        return out
    return linear_rows(x0, w, bias=bias, x1=x1, gsrc=gsrc, gaff=gaff, rows_per_img=rows_per_img)


def downsample_conv(x, w, bias):
    """Pixel-unshuffle (2x2) + conv1x1; w packed [Cout, 4C] with K ordered (p1, p2, c)."""
    B, H, W, C = x.shape
    Ho, Wo = H // 2, W // 2
    N = w.shape[0]
    y = torch.empty(B, Ho, Wo, N, dtype=x.dtype, device=x.device)
    gemm_nt(_chk(x), _chk(w), y, B * Ho * Wo, N, 4 * C, dtype=x.dtype, bias=bias, unshuffle=(Ho, Wo, C))
    return y


def row_softmax(s, out_dtype):
    rows, n = s.numel() // s.shape[-1], s.shape[-1]
    out = torch.empty(s.shape, dtype=out_dtype, device=s.device)
    call("row_softmax", **{"in": _chk(s)}, out=out, rows=rows, n=n, in_f32=int(s.dtype == torch.float32), dtype=dtype_code(out_dtype))
    return out


def group_softmax32_(q, scale):
    groups = q.shape[-1] // 32
    call("group_softmax32", q=_chk(q), rows=q.numel() // q.shape[-1], groups=groups, dtype=dtype_code(q.dtype), scale=scale)
    return q


def nchw_to_nhwc(x, dtype):
    B, C, H, W = x.shape
    out = torch.empty(B, H, W, C, dtype=dtype, device=x.device)
    call("nchw_f32_to_nhwc", **{"in": _chk(x.float())}, out=out, B=B, C=C, HW=H * W, dtype=dtype_code(dtype))
    return out


def nhwc_to_nchw_f32(x):
    B, H, W, C = x.shape
    out = torch.empty(B, C, H, W, dtype=torch.float32, device=x.device)
    call("nhwc_to_nchw_f32", **{"in": _chk(x)}, out=out, B=B, C=C, HW=H * W, dtype=dtype_code(x.dtype))
    return out


def bilinear_ac_into(x, out, c_off):
    B, H, W, C = x.shape
    _, Ctot, h, w = out.shape
    call("bilinear_ac_concat", **{"in": _chk(x)}, out=out, B=B, H=H, W=W, C=C, h=h, w=w, Ctot=Ctot, c_off=c_off,
         dtype=dtype_code(x.dtype))
