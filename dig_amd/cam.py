"""HIP forward of the MCTformer class-token attention -> CAM path (fp32).

mctformer_forward reproduces MCTformerV2_DiG.forward without the LFCA branch (networks/mctformer.py:227-263,
293-350): patch embed -> 12 x (LN, qkv, attention with rollout, proj, LN, MLP) -> head conv3x3 -> CAM aggregation.
The probabilities are head-averaged and layer-summed inside the attention kernel (two accumulators), so the
12 x [B,6,N,N] tensors of the reference never exist.  infer_multi_cam / cam_labels reproduce
models/model_ECCV24_DiG.py:468-483 and evaluation.py:42-50.
"""
import math

import torch

from . import ops
from ._lib import _pver, call

F32 = torch.float32


def _pack(m):
    key = (next(m.parameters()).device, _pver(m.parameters()), m.cls_token.data_ptr())
    if m._packed is not None and m._packed["key"] == key:
        return m._packed
    f = lambda t: t.detach().float().contiguous()
    P = {"key": key}
    P["pe_w"] = f(m.patch_embed.proj.weight.reshape(m.embed_dim, -1))  # [384, 3*16*16], K = (c, ky, kx)
    P["pe_b"] = f(m.patch_embed.proj.bias)
    P["cls"], P["pos"] = f(m.cls_token[0]), f(m.pos_embed[0])
    P["cls_pos"] = P["cls"] + P["pos"][:m.num_classes]  # class-token rows incl. their (never interpolated) pos-embed, folded at pack time
    P["blocks"] = []
    for blk in m.blocks:
        P["blocks"].append(dict(
            n1=(f(blk.norm1.weight), f(blk.norm1.bias)), n2=(f(blk.norm2.weight), f(blk.norm2.bias)),
            qkv=(f(blk.attn.qkv.weight), f(blk.attn.qkv.bias)), proj=(f(blk.attn.proj.weight), f(blk.attn.proj.bias)),
            fc1=(f(blk.mlp.fc1.weight), f(blk.mlp.fc1.bias)), fc2=(f(blk.mlp.fc2.weight), f(blk.mlp.fc2.bias))))
    P["head"] = ops.ConvWeight(m.head.weight, m.head.bias, F32)
    P["pos_cache"] = {}
    m._packed = P
    return P


def _layernorm(x, wb, eps=1e-6):
    out = torch.empty_like(x)
    call("layernorm", x=x, gamma=wb[0], beta=wb[1], out=out, rows=x.numel() // x.shape[-1], D=x.shape[-1], eps=eps)
    return out


def _pos_embed(m, P, H, W):
    """interpolate_pos_encoding (mctformer.py:196-217); note the reference's (w, h) = (H, W)."""
    C, ps = m.num_classes, m.patch_embed.patch_size[0]
    pos = P["pos"]
    Ng = pos.shape[0] - C
    hp, wp = H // ps, W // ps
    if hp * wp == Ng and H == W:
        return pos
    if (H, W) in P["pos_cache"]:
        return P["pos_cache"][(H, W)]
    g = int(math.sqrt(Ng))
    s_h, s_w = (hp + 0.1) / math.sqrt(Ng), (wp + 0.1) / math.sqrt(Ng)
    ho, wo = int(math.floor(g * s_h)), int(math.floor(g * s_w))
    assert ho == hp and wo == wp
    out = torch.empty(C + hp * wp, pos.shape[1], dtype=F32, device=pos.device)
    out[:C] = pos[:C]
    call("bicubic_pos", **{"in": pos.data_ptr() + C * pos.shape[1] * 4}, out=out.data_ptr() + C * pos.shape[1] * 4,
         g=g, D=pos.shape[1], ho=hp, wo=wp, rscale_h=float(1.0 / s_h), rscale_w=float(1.0 / s_w))
    P["pos_cache"][(H, W)] = out
    return out


def _pack_lfca(m):
    dt = m.lfca_dtype
    key = (next(m.parameters()).device, _pver(m.parameters()), m.LFCA0.query.weight.data_ptr(), dt)
    if m._packed_lfca is not None and m._packed_lfca["key"] == key:
        return m._packed_lfca
    f = lambda t: t.detach().float().contiguous()
    mat = lambda conv: conv.weight.detach().reshape(conv.weight.shape[0], -1).contiguous().to(dt)   # storage type of the big GEMMs
    wb = lambda mod: (f(mod.weight), f(mod.bias))
    wbd = lambda mod: (mod.weight.detach().contiguous().to(dt), f(mod.bias))
    zeros = lambda n: torch.zeros(n, dtype=F32, device=m.head.weight.device)
    P = {"key": key, "dtype": dt}
    for name in ("LFCA0", "LFCA1"):
        L = getattr(m, name)
        di = L.diff_init
        P[name] = dict(
            c1=ops.ConvWeight(di.convbnrelu1[0].weight, zeros(m.D_f), dt), gn1=wb(di.convbnrelu1[1]),
            c2=ops.ConvWeight(di.convbn2[0].weight, zeros(m.D_f), dt), gn2=wb(di.convbn2[1]), branch=mat(di.branch),
            norm_k=wb(L.norm_k), norm_v=wb(L.norm_v), norm_p1=wb(L.norm_p1), norm_c1=wb(L.norm_c1),
            query=wb(L.query), key=(mat(L.key), f(L.key.bias)), value=ops.ConvWeight(L.value.weight, L.value.bias, dt),
            mlp_c0=wb(L.mlp_c0), mlp_p0=ops.ConvWeight(L.mlp_p0.weight, L.mlp_p0.bias, dt),
            c1a=wb(L.mlp_c1[0]), c1b=wb(L.mlp_c1[2]), p1a=(mat(L.mlp_p1[0]), f(L.mlp_p1[0].bias)), p1b=(mat(L.mlp_p1[2]), f(L.mlp_p1[2].bias)))
    P["head0"] = ops.ConvWeight(m.diff_head_p[0].weight, m.diff_head_p[0].bias, dt)
    P["head2"] = (mat(m.diff_head_p[2]), f(m.diff_head_p[2].bias))
    m._packed_lfca = P
    return P


def _affine(x, aff=None, res=None, pre=0, post=0, xs=1.0, rs=1.0):
    """out = post(pre(xs * (a*x + b)) + rs * res) on NHWC (dig_affine_act_add); activations 0 none, 1 SiLU, 2 ReLU, 3 GELU."""
    out = torch.empty_like(x)
    C = x.shape[-1]
    B = x.shape[0]
    call("affine_act_add", x=x, aff=aff, res=res, out=out, B=B, HW=x.numel() // (B * C), C=C, pre_act=pre, post_act=post,
         dtype=ops.dtype_code(x.dtype), x_scale=float(xs), res_scale=float(rs))
    return out


def _cast(x, dt):
    """storage-type conversion (dig_cast); identity when x already has dt"""
    if x.dtype == dt:
        return x
    out = torch.empty(x.shape, dtype=dt, device=x.device)
    call("cast", **{"in": x.contiguous()}, out=out, n=x.numel(), to_f32=int(dt == F32))
    return out


def _gn1(stats, wb, hw, eps):
    """GroupNorm(1, C) affine table from per-tile partials of any group count: the partials of all groups of an image are
    simply more partials of its single group."""
    B = stats.shape[0]
    return ops.gn_finalize(stats.reshape(B, 1, -1, 2), wb[0], wb[1], hw, eps=eps)


def lfca_forward(L, qkv, C, feat, heads=8):
    """LFCA.forward in eval mode (mctformer.py:124-166) on the HIP kernels.  The convolutions and the wide 1x1 GEMMs run in the
    storage type of `feat` (fp32, or bf16 with fp32 accumulation = what torch.autocast would pick for them); LayerNorm,
    GroupNorm statistics, the cross attention and the class-token branch are always fp32.
    qkv [B, C+N, 3D]: output of a ViT block's qkv Linear (its first D columns are the un-scaled query in the
    [B, C+N, D] layout of mctformer.py:280-281); feat NHWC [B, hd, wd, 5*D_f].  -> (x_c [B, C, D], x_p NHWC [B, hd, wd, D_f])."""
    B, Nq, D3 = qkv.shape
    D = D3 // 3
    N = Nq - C
    ht = wt = int(N ** 0.5)
    _, hd, wd, _ = feat.shape
    Nf = hd * wd
    dev = qkv.device
    dt = feat.dtype
    gg = 16 if dt == F32 else 8  # GroupNorm partials per 64- / 128-channel slice (the conv's N tile); folded into the single group below
    # diff_init = ResBlock_ysh(5*D_f, D_f, groups=1) (resnet101.py:195-222)
    h1, st = ops.conv3x3(feat, L["c1"], groups=gg)
    a1 = _affine(h1, _gn1(st, L["gn1"], Nf, 1e-5), pre=2)
    h2, st = ops.conv3x3(a1, L["c2"], groups=gg)
    br = ops.linear_rows(feat, L["branch"])
    kv = _affine(h2, _gn1(st, L["gn2"], Nf, 1e-5), res=br, post=2)
    Df = kv.shape[-1]
    # q / k / v projections + LayerNorms (mctformer.py:133-136)
    q = torch.empty(B, Nq, D, dtype=F32, device=dev)
    ops.gemm_nt(qkv, L["query"][0], q, B * Nq, D, D, dtype=F32, lda0=D3, bias=L["query"][1])
    q = _layernorm(q, L["norm_k"])
    k = _layernorm(ops.linear_rows(kv, L["key"][0], bias=L["key"][1], out_f32=True), L["norm_k"])    # [B,hd,wd,D] fp32
    v = _layernorm(_cast(ops.conv3x3(kv, L["value"], want_stats=False)[0], F32), L["norm_v"])        # [B,hd,wd,Df] fp32
    # 8-head cross attention (mctformer.py:138-151).  The key count (196 at 224x224, 225 for 15x15 maps) need not be a
    # multiple of the GEMM's 8-element K vectors: the logits live in rows of Nfp >= Nf columns whose tail is -inf (softmax ->
    # exactly 0 there) and v^T in rows of the same pitch whose tail is zero, so the padded K range multiplies 0 by 0.
    Nfp = (Nf + 7) // 8 * 8
    dq, dv = D // heads, Df // heads
    S = torch.full((B, heads, Nq, Nfp), float("-inf"), dtype=F32, device=dev)
    ops.gemm_nt(q, k, S, Nq, Nf, dq, dtype=F32, lda0=D, ldb=D, ldc=Nfp, alpha=1.0 / ((384 // 8) ** 0.5), batch=B * heads, z_inner=heads,
                sa=(Nq * D, dq), sb=(Nf * D, dq), sc=(heads * Nq * Nfp, Nq * Nfp))
    Pm = ops.row_softmax(S, F32)
    vT = torch.zeros(B, Df, Nfp, dtype=F32, device=dev)
    call("nhwc_to_nchw_f32", **{"in": v}, out=vT, B=B, C=Df, HW=Nf, dtype=1, out_ld=Nfp)
    o = torch.empty(B, Nq, Df, dtype=F32, device=dev)
    ops.gemm_nt(Pm, vT, o, Nq, dv, Nfp, dtype=F32, lda0=Nfp, ldb=Nfp, ldc=Df, batch=B * heads, z_inner=heads,
                sa=(heads * Nq * Nfp, Nq * Nfp), sb=(Df * Nfp, dv * Nfp), sc=(Nq * Df, dv))
    # class-token branch (mctformer.py:153-158): rows [:C] of every image
    x_c = torch.empty(B, C, D, dtype=F32, device=dev)
    ops.gemm_nt(o, L["mlp_c0"][0], x_c, C, D, Df, dtype=F32, bias=L["mlp_c0"][1], res=qkv, ldr=D3, batch=B,
                sa=(Nq * Df, 0), sc=(C * D, 0), sr=(Nq * D3, 0))
    y = _layernorm(x_c, L["norm_c1"])
    hdn = ops.linear_rows(y, L["c1a"][0], bias=L["c1a"][1], gelu=True)
    x_c = ops.linear_rows(hdn, L["c1b"][0], bias=L["c1b"][1], res=x_c)
    # patch branch (mctformer.py:154,160-164): rows [C:] of o are NHWC [B, ht, wt, Df] with batch stride Nq*Df
    o_p = torch.empty(B, ht, wt, Df, dtype=dt, device=dev)
    cw = L["mlp_p0"]
    od = _cast(o, dt)
    call("conv3x3", in0=od.data_ptr() + C * Df * od.element_size(), in1=None, aff0=None, w=cw.w, bias=cw.bias, out=o_p, stats=None,
         in0_bstride=Nq * Df, B=B, H=ht, W=wt, C0=Df, C1=0, Cout=Df, groups=1, up2=0, dtype=ops.dtype_code(dt))
    x_p = torch.empty(B, hd, wd, Df, dtype=dt, device=dev)
    call("bilinear_resize", **{"in": o_p}, res=kv, out=x_p, B=B, hi=ht, wi=wt, ho=hd, wo=wd, C=Df, dtype=ops.dtype_code(dt))
    tiles = min(Nf, 16)
    st = torch.empty(B, 1, tiles, 2, dtype=F32, device=dev)
    call("gn_stats", x=x_p, stats=st, B=B, HW=Nf, C=Df, groups=1, tiles=tiles, dtype=ops.dtype_code(dt))
    y = _affine(x_p, _gn1(st, L["norm_p1"], Nf, 1e-6))
    hdn = ops.linear_rows(y, L["p1a"][0], bias=L["p1a"][1], gelu=True)
    x_p = ops.linear_rows(hdn, L["p1b"][0], bias=L["p1b"][1], res=x_p)
    return x_c, x_p


def mctformer_forward(m, img, n_layers=12, feat_diff=None, flip_tta=False):
    """-> dict with rcams, mtatt, attn (patch_attn), cams, fcams, cls, pcls, feat (mctformer.py:331-350 keys); with
    feat_diff (five [B,1024,h,w] guidance maps) also cam_diff, diff_cls, diff_pcls (mctformer.py:265-291,324-346).
    flip_tta: run the batch [img, img.flip(-1)] (2B rows; the mirrored half is formed by the patch im2row kernel)."""
    if not img.is_cuda:
        raise RuntimeError("dig_amd.MCTformerV2_DiG runs on the HIP device only (no CPU fallback)")
    P = _pack(m)
    if P.get("device_checked") != img.device:  # once per packing: a move or an in-place update of any parameter re-packs
        from .diffusion import _same_device
        _same_device(m, img.device, "MCTformerV2_DiG")
        P["device_checked"] = img.device
    Bi, _, H, W = img.shape
    B = 2 * Bi if flip_tta else Bi
    assert not (flip_tta and feat_diff is not None)
    C, D, heads, ps = m.num_classes, m.embed_dim, m.num_heads, m.patch_embed.patch_size[0]
    hp, wp = H // ps, W // ps
    Pn, N = hp * wp, C + hp * wp
    dev = img.device
    img = img.contiguous().float()
    pos = _pos_embed(m, P, H, W)
    # patch embedding: im2row + GEMM(+bias, +pos) written straight into the token buffer behind the class tokens
    rows = torch.empty(B * Pn, 3 * ps * ps, dtype=F32, device=dev)
    call("patchify", img=img, rows=rows, B=B, H=H, W=W, ps=ps, flip_from=Bi if flip_tta else 0)
    x = torch.empty(B, N, D, dtype=F32, device=dev)
    x[:, :C] = P["cls_pos"]  # broadcast copy of the class-token rows (cls_token + pos_embed[:C], mctformer.py:240-241)
    ops.gemm_nt(rows, P["pe_w"], (x, C * D), Pn, D, 3 * ps * ps, dtype=F32, bias=P["pe_b"], res=(pos, C * D), ldr=D,
                batch=B, sa=(Pn * 3 * ps * ps, 0), sc=(N * D, 0), sr=(0, 0))
    acc_all = torch.zeros(B, N, N, dtype=F32, device=dev)
    acc_cls = torch.zeros(B, C, N, dtype=F32, device=dev)
    depth = len(P["blocks"])
    qkv_last = []
    for li, bp in enumerate(P["blocks"]):
        y = _layernorm(x, bp["n1"])
        qkv = ops.linear_rows(y, bp["qkv"][0], bias=bp["qkv"][1])
        if feat_diff is not None and li >= depth - 2:
            qkv_last.append(qkv)
        o = torch.empty(B, N, D, dtype=F32, device=dev)
        call("vit_attn", qkv=qkv, out=o, acc_all=acc_all, acc_cls=acc_cls if li >= depth - n_layers else None,
             B=B, N=N, heads=heads, dh=D // heads, ncls=C)
        x = ops.linear_rows(o, bp["proj"][0], bias=bp["proj"][1], res=x)
        y = _layernorm(x, bp["n2"])
        hdn = ops.linear_rows(y, bp["fc1"][0], bias=bp["fc1"][1], gelu=True)
        x = ops.linear_rows(hdn, bp["fc2"][0], bias=bp["fc2"][1], res=x)
    # head conv3x3 on the patch tokens viewed as NHWC [B,hp,wp,D] (batch stride N*D, offset C*D)
    head = torch.empty(B, hp, wp, C, dtype=F32, device=dev)
    call("conv3x3", in0=x.data_ptr() + C * D * 4, in1=None, aff0=None, w=P["head"].w, bias=P["head"].bias, out=head, stats=None,
         in0_bstride=N * D, B=B, H=hp, W=wp, C0=D, C1=0, Cout=C, groups=1, up2=0, dtype=1)
    mtatt = torch.empty(B, C, hp, wp, dtype=F32, device=dev)
    cams = torch.empty_like(mtatt)
    rcams = torch.empty_like(mtatt)
    call("cam_fuse", acc_all=acc_all, acc_cls=acc_cls, head=head, mtatt=mtatt, cams=cams, rcams=rcams, B=B, C=C, P=Pn)
    # outputs['cams'] = relu(head) as NCHW and the top-3 pooled patch logits (mctformer.py:296-300); class-token logits (:295)
    relu_head = torch.empty(B, C, hp, wp, dtype=F32, device=dev)
    pcls = torch.empty(B, C, dtype=F32, device=dev)
    call("head_pool", **{"in": head}, relu_nchw=relu_head, pooled=pcls, B=B, P=Pn, C=C)
    cls = torch.empty(B, C, dtype=F32, device=dev)
    call("row_mean", x0=x, x1=None, out=cls, bstride0=N * D, bstride1=0, B=B, C=C, D=D)
    outs = {"cls": cls, "pcls": pcls, "cams": relu_head, "fcams": cams, "attn": acc_all[:, C:, C:],
            "mtatt": mtatt, "rcams": rcams, "feat": x[:, C:].reshape(B, hp, wp, D).permute(0, 3, 1, 2)}
    if feat_diff is not None:
        PL = _pack_lfca(m)
        hd, wd = feat_diff[0].shape[2:]
        # torch.stack(feat_diff, 1).view(n, -1, h, w) (mctformer.py:275): timestep-major channels, then NHWC for the kernels
        cat = torch.stack([t.float() for t in feat_diff], dim=1).reshape(B, -1, hd, wd).contiguous()
        feat = ops.nchw_to_nhwc(cat, PL["dtype"])
        c0, p0 = lfca_forward(PL["LFCA0"], qkv_last[1], C, feat)   # query0 = q of the last block (mctformer.py:280)
        c1, p1 = lfca_forward(PL["LFCA1"], qkv_last[0], C, feat)   # query1 = q of the one before
        out_p = _affine(p0, res=p1, xs=0.5, rs=0.5)                # (p0 + p1) / 2 (mctformer.py:287)
        hh = _affine(ops.conv3x3(out_p, PL["head0"], want_stats=False)[0], pre=2)
        h2 = ops.linear_rows(hh, PL["head2"][0], bias=PL["head2"][1], out_f32=True)  # NHWC [B,hd,wd,C], pre-ReLU
        Cd = h2.shape[-1]
        outs["cam_diff"] = torch.empty(B, Cd, hd, wd, dtype=F32, device=dev)
        outs["diff_pcls"] = torch.empty(B, Cd, dtype=F32, device=dev)
        call("head_pool", **{"in": h2}, relu_nchw=outs["cam_diff"], pooled=outs["diff_pcls"], B=B, P=hd * wd, C=Cd)
        outs["diff_cls"] = torch.empty(B, C, dtype=F32, device=dev)  # ((c0 + c1) / 2).mean(-1) (mctformer.py:288,340)
        call("row_mean", x0=c0, x1=c1, out=outs["diff_cls"], bstride0=C * D, bstride1=C * D, B=B, C=C, D=D)
    return outs


def infer_multi_cam(m, img, label, n_layers=12):
    """model.py:468-477: CAM of the image and of its horizontal flip, bilinear upsample x label, sum, max_norm.
    img [B,3,H,W] (H, W multiples of 16), label [B,C] -> normalised CAM [B,C,H,W] fp32."""
    B, _, H, W = img.shape
    C = m.num_classes
    rc = mctformer_forward(m, img, n_layers, flip_tta=True)["rcams"]  # one batched forward for the image and its flip
    hp, wp = rc.shape[-2:]
    out = torch.empty(B, C, H, W, dtype=F32, device=img.device)
    call("cam_tta_norm", rc=rc[:B].contiguous(), rc_flip=rc[B:].contiguous(), label=label.float().contiguous(), out=out,
         B=B, C=C, hp=hp, wp=wp, H=H, W=W)
    return out


def cam_labels(norm_cam, label, threshold):
    """evaluation.py:42-50 on the device: uint8 label map [B,H,W]."""
    B, C, H, W = norm_cam.shape
    out = torch.empty(B, H, W, dtype=torch.uint8, device=norm_cam.device)
    call("cam_argmax", cam=norm_cam.contiguous(), label=label.float().contiguous(), out=out, B=B, C=C, HW=H * W,
         threshold=float(threshold))
    return out


def cam_dict(norm_cam_1, label_1):
    """model.py:479-483 for one image: {class index: [H,W] float32 numpy} for label > 1e-5."""
    nc = norm_cam_1.detach().cpu().numpy()
    return {i: nc[i] for i in range(label_1.shape[0]) if float(label_1[i]) > 1e-5}
