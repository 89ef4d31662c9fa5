"""Training step of the MCTformer classification losses on the HIP kernels.

model_WSSS.update (models/model_ECCV24_DiG.py:318-341,396-410,424-425):

    self.opt_trm.zero_grad(); outputs = self.net_trm(self.img, ...)
    loss_cls = W[0] * (multilabel_soft_margin(cls) + multilabel_soft_margin(pcls))        (+ the same on img_diff: loss_cls_diff)
    loss_trm.backward(); self.opt_trm.step()

The reference leaves the backward to autograd over vision_transformer.py:54-103,177-196 and mctformer.py:227-300.  Here the
forward is re-run with its intermediates kept (`forward_train`), and `backward_cls` walks the same operators in reverse on the
adjoint kernels of csrc/train.hip (dig_gemm_f32 with transposed operands as stride pairs, dig_layernorm_bwd, dig_gelu_bwd,
dig_softmax_bwd, dig_head_pool_bwd, dig_msm_grad, dig_colsum); `AdamW` is torch.optim.AdamW's update with timm's no-decay
filter.  fp32, eval-mode semantics (no DropPath / dropout: the product's MCTformer refuses drop rates), no atomics - a step is
bitwise reproducible.  All five loss terms are differentiated: loss_cls, loss_cls_diff, loss_cam_consistency (cam_diff enters
detached), loss_rcams_consistency (through the attention roll-up of all twelve layers) and loss_pred_diff (the LFCA branch, on
the tape of dig_amd/train_lfca.py).
torch allocates, copies and reshapes; every arithmetic operation is a C-ABI launch.  There is no CPU fallback.
"""
import math

import torch

from . import ops
from ._lib import call, lib
from .cam import _layernorm, _pack, _pos_embed

F32 = torch.float32


def _p(x):
    if x is None:
        return None
    if isinstance(x, tuple):
        t, off = x
        return t.data_ptr() + off * 4
    return x.data_ptr()


def gemm_f32(a, b, c, M, N, K, *, sam, sak, sbk, sbn, ldc=None, bias=None, alpha=1.0, batch=1, z_inner=1,
             sa=(0, 0), sb=(0, 0), sc=(0, 0), accumulate=False, ksplit=1):
    """dig_gemm_f32: C[z][m][n] = alpha * sum_k A[z][m*sam + k*sak] * B[z][k*sbk + n*sbn] (+ bias) (+ C).  a / b / c are fp32
    tensors or (tensor, element offset); strides in elements.  ksplit > 1 (batch 1, dense C): K is cut into that many slices
    whose partial products are added by dig_colsum - the weight gradients, whose output is a few tiles and K every token."""
    t = c[0] if isinstance(c, tuple) else c
    kw = dict(a=_p(a), b=_p(b), bias=bias, sam=sam, sak=sak, sbk=sbk, sbn=sbn, sa_o=sa[0], sa_i=sa[1], sb_o=sb[0], sb_i=sb[1],
              sc_o=sc[0], sc_i=sc[1], M=M, N=N, K=K, ldc=ldc if ldc is not None else N, batch=batch, z_inner=z_inner,
              accumulate=int(accumulate), alpha=float(alpha))
    with torch.cuda.device(t.device):
        if ksplit > 1:
            assert batch == 1 and bias is None and not accumulate and (ldc is None or ldc == N)
            part = torch.empty(ksplit, M * N, dtype=F32, device=t.device)
            call("gemm_f32", c=part, sc_k=M * N, ksplit=ksplit, **kw)
            colsum(part, ksplit, M * N, out=c)
        else:
            call("gemm_f32", c=_p(c), sc_k=0, ksplit=1, **kw)


def _ksplit(M, N, K):
    """K slices for a small-output product: about two blocks per CU (128 x 64 tiles), at least 64 K values per slice."""
    tiles = ((M + 127) // 128) * ((N + 63) // 64)
    return max(1, min(64, 512 // tiles, K // 64))


def colsum(x, M, N, ld=None, out=None, scale=1.0, accumulate=False):
    """out[n] (+)= scale * sum_m x[m*ld + n] (dig_colsum, two fixed-order stages)."""
    t = x[0] if isinstance(x, tuple) else x
    if out is None:
        out = torch.empty(N, dtype=F32, device=t.device)
    ws = torch.empty(lib().dig_colsum_chunks(M) * N, dtype=F32, device=t.device)
    call("colsum", x=_p(x), ws=ws, out=_p(out), M=M, ld=ld if ld is not None else N, N=N, accumulate=int(accumulate), scale=float(scale))
    return out


def linear_bwd(x, w, dy, M, need_dx=True):
    """Adjoint of y = x W^T + b for x [M][K], W [N][K], dy [M][N]: -> (dx [M][K] or None, dW [N][K], db [N])."""
    N, K = w.shape
    dw = torch.empty(N, K, dtype=F32, device=w.device)
    gemm_f32(dy, x, dw, N, K, M, sam=1, sak=N, sbk=K, sbn=1, ksplit=_ksplit(N, K, M))   # dW[n][k] = sum_m dy[m][n] x[m][k]
    db = colsum(dy, M, N)
    dx = None
    if need_dx:
        dx = torch.empty(M, K, dtype=F32, device=w.device)
        gemm_f32(dy, w, dx, M, K, N, sam=N, sak=1, sbk=K, sbn=1)      # dx[m][k] = sum_n dy[m][n] W[n][k]
    return dx, dw, db


def layernorm_bwd(x, gamma, dy, add=None, eps=1e-6):
    """-> (dx [+ add], dgamma, dbeta) of nn.LayerNorm over the last dimension."""
    D = x.shape[-1]
    rows = x.numel() // D
    nb = lib().dig_layernorm_bwd_blocks(rows)
    part = torch.empty(nb, 2 * D, dtype=F32, device=x.device)
    dx = torch.empty_like(x)
    call("layernorm_bwd", x=x, gamma=gamma, dy=dy, add=add, dx=dx, part=part, rows=rows, D=D, eps=float(eps))
    gb = colsum(part, nb, 2 * D)
    return dx, gb[:D], gb[D:]


def _dp_apply(u, s, res=None):
    """s_b * u (+ res): a DropPath factor per sample (mask_b / keep_prob) on a branch output u [B, ..., C], as the [B][C][2]
    (scale, shift) table of dig_affine_act_add"""
    from .cam import _affine
    B, C = u.shape[0], u.shape[-1]
    tab = torch.zeros(B, C, 2, dtype=F32, device=u.device)
    tab[..., 0] = s.to(device=u.device, dtype=F32).reshape(B, 1)
    r3 = None if res is None else res.reshape(B, -1, C)
    return _affine(u.reshape(B, -1, C).contiguous(), tab, res=r3).reshape(u.shape)


def draw_drop_path(m, B, device, lfca=True, passes=2, generator=None, lfca_rate=0.1):
    """The stochastic-depth factors of one update step, drawn as timm's DropPath draws them (per sample Bernoulli(keep_prob),
    divided by keep_prob; timm 0.3.x is not importable here: the rule is restated and PARITY UNPINNED on the random stream -
    the arithmetic with given factors is pinned, tests/test_gpu_train.py).  Block i of the ViT has rate
    linspace(0, m.drop_path_rate, depth)[i] (vision_transformer.py:265; rate 0 = nn.Identity, no draw), both branches of a
    block draw separately (:193-194); each LFCA has DropPath(0.1) on its two patch-branch residuals (mctformer.py:93,163-164)
    in the first pass only.  -> the `drop_path` argument of update_step, or None when nothing would be dropped."""
    rate = float(getattr(m, "drop_path_rate", 0.0) or 0.0)
    depth = len(m.blocks)
    rates = [float(r) for r in torch.linspace(0, rate, depth)]  # the reference's expression: fp32 values, .item()

    def draw(r):
        keep = 1.0 - r
        return (torch.rand(B, device=device, generator=generator) < keep).to(F32) / keep

    if rate <= 0.0 and not lfca:
        return None
    out = {}
    for ps in range(1, passes + 1):
        out[f"pass{ps}"] = {"blocks": [(draw(r), draw(r)) if r > 0 else (None, None) for r in rates]}
    if lfca:
        out["pass1"]["LFCA0"] = (draw(lfca_rate), draw(lfca_rate))
        out["pass1"]["LFCA1"] = (draw(lfca_rate), draw(lfca_rate))
    return out


def forward_train(m, img, want_cam=False, dp=None):
    """MCTformerV2_DiG.forward without the LFCA branch (mctformer.py:227-263,293-300,318), keeping what the backward needs.
    -> (outs {cls, pcls}, ctx); want_cam adds the CAM outputs the consistency terms use (cams = relu(head), mtatt, fcams, rcams:
    mctformer.py:302-321,334-338) from the head-averaged, layer-summed roll-up of the kept attention probabilities.
    dp: train-mode DropPath, [(s_attn, s_mlp)] per block, each a [B] tensor of mask / keep_prob or None (see draw_drop_path);
    None = eval-mode semantics."""
    if not img.is_cuda:
        raise RuntimeError("dig_amd.train runs on the HIP device only (no CPU fallback)")
    P = _pack(m)
    B, _, H, W = img.shape
    C, D, heads, ps = m.num_classes, m.embed_dim, m.num_heads, m.patch_embed.patch_size[0]
    hp, wp = H // ps, W // ps
    Pn, N = hp * wp, C + hp * wp
    dh = D // heads
    dev = img.device
    img = img.contiguous().float()
    pos = _pos_embed(m, P, H, W)
    if pos is not P["pos"]:
        raise NotImplementedError("dig_amd.train: the bicubic pos_embed resize (mctformer.py:210-214) has no adjoint kernel yet - "
                                  "train at the pos_embed's own grid (224x224 for the DeiT-S checkpoint)")
    rows = torch.empty(B * Pn, 3 * ps * ps, dtype=F32, device=dev)
    call("patchify", img=img, rows=rows, B=B, H=H, W=W, ps=ps, flip_from=0)
    x = torch.empty(B, N, D, dtype=F32, device=dev)
    x[:, :C] = P["cls_pos"]
    ops.gemm_nt(rows, P["pe_w"], (x, C * D), Pn, D, 3 * ps * ps, dtype=F32, bias=P["pe_b"], res=(pos, C * D), ldr=D,
                batch=B, sa=(Pn * 3 * ps * ps, 0), sc=(N * D, 0), sr=(0, 0))
    saved = []
    acc = torch.zeros(B, N, N, dtype=F32, device=dev) if want_cam else None
    if dp is not None and len(dp) != len(P["blocks"]):
        raise ValueError(f"dig_amd.train.forward_train: {len(dp)} DropPath entries for {len(P['blocks'])} blocks")
    for li, bp in enumerate(P["blocks"]):
        s1, s2 = dp[li] if dp is not None else (None, None)
        x0 = x
        y1 = _layernorm(x0, bp["n1"])
        qkv = ops.linear_rows(y1, bp["qkv"][0], bias=bp["qkv"][1])
        S = torch.empty(B, heads, N, N, dtype=F32, device=dev)
        gemm_f32(qkv, (qkv, D), S, N, N, dh, sam=3 * D, sak=1, sbk=1, sbn=3 * D, alpha=dh ** -0.5, batch=B * heads, z_inner=heads,
                 sa=(N * 3 * D, dh), sb=(N * 3 * D, dh), sc=(heads * N * N, N * N))
        Pm = ops.row_softmax(S, F32)
        if want_cam:
            call("head_sum", p=Pm, out=acc, n=N * N, B=B, H=heads, accumulate=1, scale=1.0 / heads)
        o = torch.empty(B, N, D, dtype=F32, device=dev)
        gemm_f32(Pm, (qkv, 2 * D), o, N, dh, N, sam=N, sak=1, sbk=3 * D, sbn=1, ldc=D, batch=B * heads, z_inner=heads,
                 sa=(heads * N * N, N * N), sb=(N * 3 * D, dh), sc=(N * D, dh))
        if s1 is None:
            x1 = ops.linear_rows(o, bp["proj"][0], bias=bp["proj"][1], res=x0)
        else:  # x + drop_path(attn(norm1(x))), vision_transformer.py:193
            x1 = _dp_apply(ops.linear_rows(o, bp["proj"][0], bias=bp["proj"][1]), s1, res=x0)
        y2 = _layernorm(x1, bp["n2"])
        pre = ops.linear_rows(y2, bp["fc1"][0], bias=bp["fc1"][1])
        hdn = torch.empty_like(pre)
        call("act", **{"in": pre}, out=hdn, n=pre.numel(), act=1, in_f32=1, out_f32=1, dtype=1)
        if s2 is None:
            x = ops.linear_rows(hdn, bp["fc2"][0], bias=bp["fc2"][1], res=x1)
        else:  # x + drop_path(mlp(norm2(x))), vision_transformer.py:194
            x = _dp_apply(ops.linear_rows(hdn, bp["fc2"][0], bias=bp["fc2"][1]), s2, res=x1)
        saved.append(dict(x0=x0, y1=y1, qkv=qkv, P=Pm, o=o, x1=x1, y2=y2, pre=pre, h=hdn, s1=s1, s2=s2))
    head = torch.empty(B, hp, wp, C, dtype=F32, device=dev)
    call("conv3x3", in0=x.data_ptr() + C * D * 4, in1=None, aff0=None, w=P["head"].w, bias=P["head"].bias, out=head, stats=None,
         in0_bstride=N * D, B=B, H=hp, W=wp, C0=D, C1=0, Cout=C, groups=1, up2=0, dtype=1)
    pcls = torch.empty(B, C, dtype=F32, device=dev)
    call("head_pool", **{"in": head}, relu_nchw=None, pooled=pcls, B=B, P=Pn, C=C)
    cls = torch.empty(B, C, dtype=F32, device=dev)
    call("row_mean", x0=x, x1=None, out=cls, bstride0=N * D, bstride1=0, B=B, C=C, D=D)
    ctx = dict(rows=rows, saved=saved, x=x, head=head, dims=(B, C, D, heads, dh, hp, wp, Pn, N, ps))
    outs = {"cls": cls, "pcls": pcls}
    if want_cam:
        relu_head = torch.empty(B, C, hp, wp, dtype=F32, device=dev)
        call("head_pool", **{"in": head}, relu_nchw=relu_head, pooled=None, B=B, P=Pn, C=C)
        mtatt, fcams, rcams = (torch.empty(B, C, hp, wp, dtype=F32, device=dev) for _ in range(3))
        call("cam_fuse", acc_all=acc, acc_cls=acc[:, :C].contiguous(), head=head, mtatt=mtatt, cams=fcams, rcams=rcams, B=B, C=C, P=Pn)
        outs.update(cams=relu_head, mtatt=mtatt, fcams=fcams, rcams=rcams)
        ctx.update(acc=acc, mtatt=mtatt, fcams=fcams)
    return outs, ctx


def backward_cls(m, outs, ctx, label, weight=1.0, grads=None, d_cams=None, d_rcams=None):
    """d/dparams of  weight * (F.multilabel_soft_margin_loss(cls, label) + F.multilabel_soft_margin_loss(pcls, label))
    (model.py:337-340 with weight = W[0]; :406-409 with W[3] on the img_diff pass), plus - for a forward_train(want_cam=True)
    pass - the gradients that arrive at its CAM outputs: d_cams [B,C,hp,wp] w.r.t. outs['cams'] = relu(head) (the
    CAM-consistency term) and d_rcams w.r.t. outs['rcams'] (the rcams-consistency term, which reaches the attention
    probabilities of every layer through the roll-up).  -> {named_parameters() key: gradient}; with `grads` given (a second
    pass of the same step) the new gradients are accumulated into it, as autograd's .grad does."""
    P = _pack(m)
    B, C, D, heads, dh, hp, wp, Pn, N, ps = ctx["dims"]
    dev = ctx["x"].device
    label = label.to(dev).float().contiguous()
    M = B * N
    new = {}
    # ---- loss -> cls / pcls
    dcls = torch.empty(B, C, dtype=F32, device=dev)
    call("msm_grad", x=outs["cls"], y=label, dx=dcls, scale=float(weight) / D, B=B, C=C)       # incl. the 1/D of x_cls.mean(-1)
    dpcls = torch.empty(B, C, dtype=F32, device=dev)
    call("msm_grad", x=outs["pcls"], y=label, dx=dpcls, scale=float(weight), B=B, C=C)
    # ---- top-3 pooling -> head conv (mctformer.py:298-300)
    Cp = (C + 7) // 8 * 8
    dhead = torch.empty(B, hp, wp, Cp, dtype=F32, device=dev)
    call("head_pool_bwd", head=ctx["head"], dpooled=dpcls, dhead=dhead, B=B, P=Pn, C=C, Cp=Cp)
    # ---- CAM outputs (mctformer.py:302-321): cams = relu(head); fcams = mtatt * relu(head); rcams = patch_attn @ fcams
    G = None
    if d_cams is not None:
        call("cam_fuse_bwd", dcams=d_cams.contiguous(), mtatt=None, head=ctx["head"], G=None, dhead=dhead, B=B, C=C, P=Pn, Cp=Cp, inv_heads=0.0)
    if d_rcams is not None:
        acc, drc = ctx["acc"], d_rcams.contiguous()
        G = torch.zeros(B, N, N, dtype=F32, device=dev)  # gradient of the roll-up (head mean, layer sum): added to every layer's dP
        dfc = torch.empty(B, C, Pn, dtype=F32, device=dev)
        # dfcams[c][j] = sum_i patch_attn[i][j] drc[c][i];  dpatch_attn[i][j] = sum_c drc[c][i] fcams[c][j]  (/ heads into G)
        gemm_f32(drc, (acc, C * N + C), dfc, C, Pn, Pn, sam=Pn, sak=1, sbk=N, sbn=1, batch=B, sa=(C * Pn, 0), sb=(N * N, 0), sc=(C * Pn, 0))
        gemm_f32(drc, ctx["fcams"], (G, C * N + C), Pn, Pn, C, sam=1, sak=Pn, sbk=Pn, sbn=1, ldc=N, alpha=1.0 / heads, batch=B,
                 sa=(C * Pn, 0), sb=(C * Pn, 0), sc=(N * N, 0))
        call("cam_fuse_bwd", dcams=dfc, mtatt=ctx["mtatt"], head=ctx["head"], G=G, dhead=dhead, B=B, C=C, P=Pn, Cp=Cp, inv_heads=1.0 / heads)
    new["head.bias"] = colsum(dhead, B * Pn, Cp)[:C].clone()
    # weight gradient on the zero-padded domain: a tap is then a constant element offset, dW_tap = dYpad^T Xpad[+off]
    xfin = ctx["x"]
    Hp2, Wp2 = hp + 2, wp + 2
    xpad = torch.zeros(B, Hp2, Wp2, D, dtype=F32, device=dev)
    xpad[:, 1:-1, 1:-1] = xfin[:, C:].reshape(B, hp, wp, D)
    dypad = torch.zeros(B, Hp2, Wp2, Cp, dtype=F32, device=dev)
    dypad[:, 1:-1, 1:-1] = dhead
    p0 = Wp2 + 1
    Kp = B * Hp2 * Wp2 - 2 * p0
    dwt = torch.empty(9, Cp, D, dtype=F32, device=dev)
    for tap in range(9):
        off = (tap // 3 - 1) * Wp2 + (tap % 3 - 1)
        gemm_f32((dypad, p0 * Cp), (xpad, (p0 + off) * D), (dwt, tap * Cp * D), Cp, D, Kp, sam=1, sak=Cp, sbk=D, sbn=1,
                 ksplit=_ksplit(Cp, D, Kp))
    new["head.weight"] = dwt[:, :C].permute(1, 2, 0).reshape(C, D, 3, 3).contiguous()
    # input gradient: the same conv kernel on dY with the taps mirrored and the channel roles swapped
    wflip = torch.zeros(D, Cp, 3, 3, dtype=F32, device=dev)
    wflip[:, :C] = m.head.weight.detach().float().flip(2, 3).permute(1, 0, 2, 3)
    cwT = ops.ConvWeight(wflip, torch.zeros(D, dtype=F32, device=dev), F32)
    dx = torch.empty(B, N, D, dtype=F32, device=dev)
    dxp = torch.empty(B, hp, wp, D, dtype=F32, device=dev)
    call("conv3x3", in0=dhead, in1=None, aff0=None, w=cwT.w, bias=cwT.bias, out=dxp, stats=None, in0_bstride=0, B=B, H=hp, W=wp,
         C0=Cp, C1=0, Cout=D, groups=1, up2=0, dtype=1)
    dx[:, C:] = dxp.reshape(B, Pn, D)
    dx[:, :C] = dcls.unsqueeze(-1).expand(B, C, D)   # x_cls.mean(-1): every embedding element gets dcls / D (scaled above)
    # ---- the blocks in reverse (vision_transformer.py:186-190)
    for li in range(len(P["blocks"]) - 1, -1, -1):
        bp, sv = P["blocks"][li], ctx["saved"][li]
        pre = f"blocks.{li}."
        # a dropped branch: its gradient is the residual stream's times the same per-sample factor; the stream itself passes
        dbr = dx if sv["s2"] is None else _dp_apply(dx, sv["s2"])
        dh_, dw, db = linear_bwd(sv["h"].reshape(M, -1), bp["fc2"][0], dbr.reshape(M, D), M)
        new[pre + "mlp.fc2.weight"], new[pre + "mlp.fc2.bias"] = dw, db
        call("gelu_bwd", x=sv["pre"], dy=dh_, dx=dh_, n=dh_.numel())
        dy2, dw, db = linear_bwd(sv["y2"].reshape(M, D), bp["fc1"][0], dh_, M)
        new[pre + "mlp.fc1.weight"], new[pre + "mlp.fc1.bias"] = dw, db
        dx1, dg, dbt = layernorm_bwd(sv["x1"], bp["n2"][0], dy2.reshape(B, N, D), add=dx)
        new[pre + "norm2.weight"], new[pre + "norm2.bias"] = dg.clone(), dbt.clone()
        dbr = dx1 if sv["s1"] is None else _dp_apply(dx1, sv["s1"])
        do, dw, db = linear_bwd(sv["o"].reshape(M, D), bp["proj"][0], dbr.reshape(M, D), M)
        new[pre + "attn.proj.weight"], new[pre + "attn.proj.bias"] = dw, db
        # attention (vision_transformer.py:87-97): per (image, head) on the q | k | v thirds of qkv, gradients into dqkv
        qkv, Pm = sv["qkv"], sv["P"]
        dqkv = torch.empty(B, N, 3 * D, dtype=F32, device=dev)
        bh = dict(batch=B * heads, z_inner=heads)
        sq, sP, sO = (N * 3 * D, dh), (heads * N * N, N * N), (N * D, dh)
        gemm_f32(Pm, do, (dqkv, 2 * D), N, dh, N, sam=1, sak=N, sbk=D, sbn=1, ldc=3 * D, sa=sP, sb=sO, sc=sq, **bh)            # dV = P^T dO
        dP = torch.empty(B, heads, N, N, dtype=F32, device=dev)
        gemm_f32(do, (qkv, 2 * D), dP, N, N, dh, sam=D, sak=1, sbk=1, sbn=3 * D, sa=sO, sb=sq, sc=sP, **bh)                    # dP = dO V^T
        call("softmax_bwd", p=Pm, dp=dP, g=G, rows=B * heads * N, n=N, heads=heads, N=N, scale=dh ** -0.5)                                               # dS (of q k^T)
        gemm_f32(dP, (qkv, D), dqkv, N, dh, N, sam=N, sak=1, sbk=3 * D, sbn=1, ldc=3 * D, sa=sP, sb=sq, sc=sq, **bh)            # dQ = dS K
        gemm_f32(dP, qkv, (dqkv, D), N, dh, N, sam=1, sak=N, sbk=3 * D, sbn=1, ldc=3 * D, sa=sP, sb=sq, sc=sq, **bh)            # dK = dS^T Q
        dy1, dw, db = linear_bwd(sv["y1"].reshape(M, D), bp["qkv"][0], dqkv.reshape(M, 3 * D), M)
        new[pre + "attn.qkv.weight"], new[pre + "attn.qkv.bias"] = dw, db
        dx, dg, dbt = layernorm_bwd(sv["x0"], bp["n1"][0], dy1.reshape(B, N, D), add=dx1)
        new[pre + "norm1.weight"], new[pre + "norm1.bias"] = dg.clone(), dbt.clone()
    # ---- embedding (mctformer.py:229-234): pos_embed and cls_token see the batch sum, the patch projection is a Linear
    gp = colsum(dx, B, N * D).reshape(1, N, D)
    new["pos_embed"] = gp
    new["cls_token"] = gp[:, :C].clone()
    dpatch = dx[:, C:].reshape(B * Pn, D).contiguous()
    _, dw, db = linear_bwd(ctx["rows"], P["pe_w"], dpatch, B * Pn, need_dx=False)
    new["patch_embed.proj.weight"] = dw.reshape(m.patch_embed.proj.weight.shape)
    new["patch_embed.proj.bias"] = db
    if grads is None:
        return new
    for k, g in new.items():
        if k in grads:
            _add_into(grads[k], g)
        else:
            grads[k] = g
    return grads


def _add_into(dst, src):
    """dst += src on the device (a one-row dig_colsum with accumulate: out[n] += x[0][n])."""
    n = dst.numel()
    ws = torch.empty(n, dtype=F32, device=dst.device)
    call("colsum", x=src.contiguous(), ws=ws, out=dst, M=1, ld=n, N=n, accumulate=1, scale=1.0)


def no_weight_decay(name, p):
    """timm.optim.optim_factory.add_weight_decay as create_optimizer applies it (model.py:244): 1-D parameters, biases and the
    model's no_weight_decay() set {pos_embed, cls_token} are not decayed."""
    return p.ndim <= 1 or name.endswith(".bias") or name in ("pos_embed", "cls_token")


class AdamW:
    """torch.optim.AdamW's update on dig_adamw, one launch per parameter tensor; state (exp_avg, exp_avg_sq, step) per name.
    Defaults are the reference's: lr = 5e-4 * batch / 512 (model.py:241-242), weight decay 0.05, betas (0.9, 0.999), eps 1e-8
    (train_trm.py:121-129)."""

    def __init__(self, model, lr, weight_decay=0.05, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.lr, self.wd, self.betas, self.eps = model, float(lr), float(weight_decay), betas, float(eps)
        self.state = {}
        self.t = 0

    def step(self, grads):
        self.t += 1
        b1, b2 = self.betas
        bc1, bc2 = 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
        params = dict(self.model.named_parameters())
        for k, g in grads.items():
            p = params[k]
            if p.dtype != F32 or not p.is_contiguous():
                raise RuntimeError(f"AdamW: parameter {k} must be a contiguous fp32 tensor")
            st = self.state.get(k)
            if st is None:
                st = self.state[k] = (torch.zeros_like(p.data), torch.zeros_like(p.data))
            g = g.reshape(p.shape).contiguous()
            call("adamw", p=p.data, g=g, m=st[0], v=st[1], n=p.numel(), lr=self.lr, wd=0.0 if no_weight_decay(k, p) else self.wd,
                 beta1=b1, beta2=b2, eps=self.eps, step_size=self.lr / bc1, sqrt_bc2=math.sqrt(bc2))
        # the kernels wrote p.data behind torch's version counters: drop the packed copies of the weights
        self.model._packed = None
        if hasattr(self.model, "_packed_lfca"):
            self.model._packed_lfca = None


def maxnorm_l1_bwd(a, b, label, scale, a_times_label):
    """scale * d/db sum | (max_norm(a) [* label] - max_norm(b)) * label |  (dig_maxnorm_l1_bwd; a, b [B,C,h,w] of one size)"""
    if a.shape != b.shape:
        raise NotImplementedError("dig_amd.train: the consistency terms are differentiated for equal map sizes only "
                                  f"(got {tuple(a.shape)} vs {tuple(b.shape)}: the bilinear resize has no adjoint kernel yet)")
    B, C, h, w = b.shape
    db = torch.empty(B, C, h, w, dtype=F32, device=b.device)
    call("maxnorm_l1_bwd", a=a.contiguous().float(), b=b.contiguous(), label=label, db=db, B=B, C=C, P=h * w,
         a_times_label=int(a_times_label), scale=float(scale))
    return db


def update_step(m, opt, img, label, W=(1.0, 1.0, 1.0, 1.0, 10.0), img_diff=None, feat_diff=None, cam_diff=None, epo=10, T=5.0, A=0.2,
                drop_path=None):
    """One optimiser step of model_WSSS.update's MCTformer part (model.py:319-425), all five terms:
        W[0] loss_cls + W[1] loss_pred_diff (needs feat_diff, the five guidance maps: the LFCA branch runs on a tape,
        dig_amd/train_lfca.py) + W[2] loss_cam_consistency (cam_diff = the LFCA output, a constant of the loss; epo > 9)
        + W[3] loss_cls_diff + W[4] loss_rcams_consistency (both on img_diff, the t = 100 reconstruction; epo > 9).
    zero_grad -> forward passes with kept activations -> backward -> AdamW.  cam_diff may be given (it is a constant) when
    the LFCA branch is not run here (W[1] = 0).  drop_path: the stochastic-depth factors of this step (draw_drop_path; the
    reference runs update() in train mode with drop_path_rate 0.1, train_trm.py:158, model.py:320) - None = eval-mode semantics.
    -> (loss_trm as a device scalar, grads keyed like named_parameters())."""
    dp1, dp2 = (drop_path or {}).get("pass1") or {}, (drop_path or {}).get("pass2") or {}
    from .losses import LFCA_EPOCH, _kd, _maxnorm_l1, _msm
    dev = img.device
    lab = label.to(dev).float().contiguous()
    B, C = lab.shape
    late = epo > LFCA_EPOCH
    if W[1] > 0 and feat_diff is None:
        raise KeyError("dig_amd.train.update_step: W[1] > 0 (loss_pred_diff) needs feat_diff, the guidance maps of this batch")
    use_lfca = W[1] > 0 or (W[2] > 0 and late and cam_diff is None and feat_diff is not None)
    use_cam = W[2] > 0 and late and (cam_diff is not None or use_lfca)
    use_rc = W[4] > 0 and late and img_diff is not None
    outs, ctx = forward_train(m, img, want_cam=use_cam or use_rc, dp=dp1.get("blocks"))
    loss = torch.zeros(1, dtype=F32, device=dev)
    _msm(outs["cls"], lab, loss, W[0], False)
    _msm(outs["pcls"], lab, loss, W[0], True)
    grads = None
    if use_lfca:
        from . import train_lfca
        lo, lst = train_lfca.lfca_branch_train(m, ctx, feat_diff, dp=dp1)
        if cam_diff is None:
            cam_diff = lo["cam_diff"]
        if W[1] > 0:  # model.py:346-363; the teachers (cls / pcls of this pass) are constants
            kd_scale = W[1] * (1 - A) * (T * T) * 0.5
            _msm(lo["diff_cls"], lab, loss, W[1] * A, True)
            _msm(lo["diff_pcls"], lab, loss, W[1] * A, True)
            _kd(lo["diff_cls"], outs["cls"], loss, T, kd_scale, True)
            _kd(lo["diff_pcls"], outs["pcls"], loss, T, kd_scale, True)
            dd = []
            for stu, tea in ((lo["diff_cls"], outs["cls"]), (lo["diff_pcls"], outs["pcls"])):
                d = torch.empty(B, C, dtype=F32, device=dev)
                call("msm_grad", x=stu, y=lab, dx=d, scale=float(W[1] * A), B=B, C=C)
                call("kd_bin_grad", student=stu, teacher=tea, dx=d, T=float(T), scale=float(kd_scale), accumulate=1, B=B, C=C)
                dd.append(d)
            grads = dict(train_lfca.lfca_branch_backward(lst, dd[0], dd[1]))
        del lst
    d_cams = None
    if use_cam:
        hw = outs["cams"].shape[2] * outs["cams"].shape[3]
        term = torch.zeros(1, dtype=F32, device=dev)
        _maxnorm_l1(cam_diff, outs["cams"], lab, term, W[2], True)
        _add_into(loss, term)
        d_cams = maxnorm_l1_bwd(cam_diff, outs["cams"], lab, W[2] / (B * C * hw), True)
    rc1 = outs["rcams"] if use_rc else None
    grads = backward_cls(m, outs, ctx, lab, W[0], grads=grads, d_cams=d_cams)
    del ctx
    if img_diff is not None and (W[3] > 0 or use_rc):
        outs2, ctx2 = forward_train(m, img_diff, want_cam=use_rc, dp=dp2.get("blocks"))
        _msm(outs2["cls"], lab, loss, W[3], True)
        _msm(outs2["pcls"], lab, loss, W[3], True)
        d_rc = None
        if use_rc:
            hw = rc1.shape[2] * rc1.shape[3]
            term = torch.zeros(1, dtype=F32, device=dev)
            _maxnorm_l1(rc1, outs2["rcams"], lab, term, W[4], False)
            _add_into(loss, term)
            d_rc = maxnorm_l1_bwd(rc1, outs2["rcams"], lab, W[4] / (B * C * hw), False)
        grads = backward_cls(m, outs2, ctx2, lab, W[3], grads=grads, d_rcams=d_rc)
    opt.step(grads)
    return loss, grads


def update_cls_step(m, opt, img, label, W0=1.0, img_diff=None, W3=1.0):
    """update_step restricted to the classification terms (loss_cls, and loss_cls_diff when img_diff is given)."""
    return update_step(m, opt, img, label, W=(W0, 0.0, 0.0, W3 if img_diff is not None else 0.0, 0.0), img_diff=img_diff, epo=0)
