"""Backward of the LFCA branch of MCTformerV2_DiG (networks/mctformer.py:56-166,265-291,324-346; resnet101.py:31-38,195-222)
for loss_pred_diff of model_WSSS.update (models/model_ECCV24_DiG.py:346-363), on the HIP kernels.

The branch is a DAG of two dozen operators per LFCA with three fan-outs (kv feeds key / value / the patch residual), so it is
run on a small tape: every forward helper enqueues its launches, keeps what its adjoint needs and appends that adjoint to the
tape; `Tape.backward()` replays them in reverse.  Gradients are keyed by the data pointer of the (contiguous, owned) tensor
they belong to; parameter gradients accumulate under the reference's state_dict names (norm_k is applied twice per LFCA).
The ViT queries enter detached (mctformer.py:280-281) and the guidance maps are data: nothing flows out of the branch, its
gradients are those of LFCA0 / LFCA1 / diff_head_p only.  fp32; DropPath with given per-sample factors (dpath) or eval-mode semantics; map sizes of the
ViT grid and the guidance maps equal (the bilinear resize of mctformer.py:161 is then the identity; its adjoint is not built).
torch allocates, copies and reshapes; all arithmetic is C-ABI launches.
"""
import torch

from . import ops
from ._lib import call
from .cam import _affine, _layernorm
from .train import F32, _add_into, _dp_apply, _ksplit, colsum, gemm_f32, layernorm_bwd, linear_bwd


class Tape:
    def __init__(self):
        self.ops, self.g, self.pg, self.keep = [], {}, {}, []

    def grad(self, t):
        e = self.g.get(t.data_ptr())
        return None if e is None else e[0].reshape(t.shape)

    def acc(self, t, g, own=True):
        """add g to the gradient of tensor t (own: g is a fresh buffer nobody else holds)"""
        k = t.data_ptr()
        e = self.g.get(k)
        if e is None:
            self.g[k] = [g.reshape(t.shape), own]
            self.keep.append(t)
            return
        if not e[1]:
            e[0], e[1] = e[0].clone(), True
        _add_into(e[0], g.reshape(e[0].shape))

    def pacc(self, name, g):
        if name in self.pg:
            _add_into(self.pg[name], g.contiguous())
        else:
            self.pg[name] = g

    def backward(self):
        for op in reversed(self.ops):
            op()


def _f(t):
    return t.detach().float().contiguous()


def lin(tp, x, w, b, name, need_dx=True, res=None):
    """y = x W^T + b (+ res, a constant): nn.Linear / 1x1 Conv2d on rows [M][K]"""
    W2 = _f(w).reshape(w.shape[0], -1)
    y = ops.linear_rows(x, W2, bias=_f(b) if b is not None else None, res=res)

    def bwd():
        dy = tp.grad(y)
        dx, dw, db = linear_bwd(x, W2, dy.reshape(-1, W2.shape[0]), x.numel() // W2.shape[1], need_dx=need_dx)
        tp.pacc(name + ".weight", dw.reshape(w.shape))
        if b is not None:
            tp.pacc(name + ".bias", db)
        if need_dx:
            tp.acc(x, dx)
    tp.ops.append(bwd)
    return y


def conv3x3_bwd(x, w, dy, need_dx, has_bias):
    """adjoint of a 3x3 / pad-1 convolution on NHWC fp32: the weight gradient as nine GEMMs on the zero-padded domain (a tap is
    a constant element offset there), the input gradient as the forward kernel on mirrored taps with the channel roles swapped."""
    B, h, wd, Cin = x.shape
    Cout = dy.shape[-1]
    dev = x.device
    Hp2, Wp2 = h + 2, wd + 2
    xpad = torch.zeros(B, Hp2, Wp2, Cin, dtype=F32, device=dev)
    xpad[:, 1:-1, 1:-1] = x
    dypad = torch.zeros(B, Hp2, Wp2, Cout, dtype=F32, device=dev)
    dypad[:, 1:-1, 1:-1] = dy
    p0 = Wp2 + 1
    Kp = B * Hp2 * Wp2 - 2 * p0
    dwt = torch.empty(9, Cout, Cin, dtype=F32, device=dev)
    ks = _ksplit(Cout, Cin, Kp)
    for tap in range(9):
        off = (tap // 3 - 1) * Wp2 + (tap % 3 - 1)
        gemm_f32((dypad, p0 * Cout), (xpad, (p0 + off) * Cin), (dwt, tap * Cout * Cin), Cout, Cin, Kp, sam=1, sak=Cout, sbk=Cin, sbn=1, ksplit=ks)
    dw = dwt.permute(1, 2, 0).reshape(Cout, Cin, 3, 3).contiguous()
    db = colsum(dy, B * h * wd, Cout) if has_bias else None
    dx = None
    if need_dx:
        cwT = ops.ConvWeight(w.detach().float().flip(2, 3).permute(1, 0, 2, 3).contiguous(), torch.zeros(Cin, dtype=F32, device=dev), F32)
        dx = ops.conv3x3(dy.contiguous(), cwT, want_stats=False)[0]
    return dx, dw, db


def conv3(tp, x, w, b, name, need_dx=True):
    dev = x.device
    cw = ops.ConvWeight(w, b if b is not None else torch.zeros(w.shape[0], dtype=F32, device=dev), F32)
    y = ops.conv3x3(x, cw, want_stats=False)[0]

    def bwd():
        dx, dw, db = conv3x3_bwd(x, w, tp.grad(y), need_dx, b is not None)
        tp.pacc(name + ".weight", dw)
        if b is not None:
            tp.pacc(name + ".bias", db)
        if need_dx:
            tp.acc(x, dx)
    tp.ops.append(bwd)
    return y


def gn1(tp, x, mod, name, eps):
    """nn.GroupNorm(1, C) on NHWC [B,h,w,C]"""
    B, h, w, C = x.shape
    HW = h * w
    tiles = min(HW, 16)
    st = torch.empty(B, 1, tiles, 2, dtype=F32, device=x.device)
    call("gn_stats", x=x, stats=st, B=B, HW=HW, C=C, groups=1, tiles=tiles, dtype=1)
    gamma, beta = _f(mod.weight), _f(mod.bias)
    y = _affine(x, ops.gn_finalize(st, gamma, beta, HW, eps=eps))

    def bwd():
        dy = tp.grad(y).contiguous()
        dx = torch.empty_like(x)
        part = torch.empty(B, 2 * C, dtype=F32, device=x.device)
        call("gn1_bwd", x=x, gamma=gamma, dy=dy, dx=dx, part=part, B=B, HW=HW, C=C, eps=float(eps))
        gb = colsum(part, B, 2 * C)
        tp.pacc(name + ".weight", gb[:C].clone())
        tp.pacc(name + ".bias", gb[C:].clone())
        tp.acc(x, dx)
    tp.ops.append(bwd)
    return y


def ln(tp, x, mod, name):
    gamma, beta = _f(mod.weight), _f(mod.bias)
    y = _layernorm(x, (gamma, beta), eps=mod.eps)

    def bwd():
        dx, dg, db = layernorm_bwd(x, gamma, tp.grad(y).contiguous(), eps=mod.eps)
        tp.pacc(name + ".weight", dg.clone())
        tp.pacc(name + ".bias", db.clone())
        tp.acc(x, dx)
    tp.ops.append(bwd)
    return y


def relu(tp, x):
    y = _affine(x, pre=2)

    def bwd():
        dy = tp.grad(y).contiguous()
        dx = torch.empty_like(dy)
        call("relu_bwd", y=y, dy=dy, dx=dx, n=dy.numel())
        tp.acc(x, dx)
    tp.ops.append(bwd)
    return y


def gelu(tp, x):
    y = torch.empty_like(x)
    call("act", **{"in": x}, out=y, n=x.numel(), act=1, in_f32=1, out_f32=1, dtype=1)

    def bwd():
        dy = tp.grad(y).contiguous()
        dx = torch.empty_like(dy)
        call("gelu_bwd", x=x, dy=dy, dx=dx, n=dy.numel())
        tp.acc(x, dx)
    tp.ops.append(bwd)
    return y


def scaled(t, s):
    """s * t as a fresh tensor (a one-row dig_colsum)"""
    return colsum(t.contiguous(), 1, t.numel(), scale=s).reshape(t.shape)


def add(tp, a, b, sa=1.0, sb=1.0):
    """y = sa * a + sb * b (same shapes)"""
    a4 = a.reshape(a.shape[0], -1, a.shape[-1])
    y = _affine(a4, res=b.reshape(a4.shape), xs=sa, rs=sb).reshape(a.shape)

    def bwd():
        dy = tp.grad(y)
        if sa == 1.0 and sb == 1.0:
            tp.acc(a, dy, own=False)
            tp.acc(b, dy, own=False)
        else:
            tp.acc(a, scaled(dy, sa))
            tp.acc(b, scaled(dy, sb))
    tp.ops.append(bwd)
    return y


def dpath(tp, x, s):
    """DropPath in train mode with its draw given (mctformer.py:163-164): y = s_b * x, s_b = mask_b / keep_prob; s None = identity"""
    if s is None:
        return x
    y = _dp_apply(x, s)

    def bwd():
        tp.acc(x, _dp_apply(tp.grad(y).contiguous(), s))
    tp.ops.append(bwd)
    return y


def xattn(tp, q, k, v, heads, scale):
    """multi-head cross attention (mctformer.py:138-151): q [B,Nq,D], k [B,Nf,D], v [B,Nf,Df] -> o [B,Nq,Df]"""
    B, Nq, D = q.shape
    Nf, Df = k.shape[1], v.shape[2]
    dq, dv = D // heads, Df // heads
    dev = q.device
    bh = dict(batch=B * heads, z_inner=heads)
    sQ, sK, sV, sO, sP = (Nq * D, dq), (Nf * D, dq), (Nf * Df, dv), (Nq * Df, dv), (heads * Nq * Nf, Nq * Nf)
    S = torch.empty(B, heads, Nq, Nf, dtype=F32, device=dev)
    gemm_f32(q, k, S, Nq, Nf, dq, sam=D, sak=1, sbk=1, sbn=D, alpha=scale, sa=sQ, sb=sK, sc=sP, **bh)
    P = ops.row_softmax(S, F32)
    o = torch.empty(B, Nq, Df, dtype=F32, device=dev)
    gemm_f32(P, v, o, Nq, dv, Nf, sam=Nf, sak=1, sbk=Df, sbn=1, ldc=Df, sa=sP, sb=sV, sc=sO, **bh)

    def bwd():
        do = tp.grad(o).contiguous()
        dv_ = torch.empty_like(v)
        gemm_f32(P, do, dv_, Nf, dv, Nq, sam=1, sak=Nf, sbk=Df, sbn=1, ldc=Df, sa=sP, sb=sO, sc=sV, **bh)       # dV = P^T dO
        dP = torch.empty_like(P)
        gemm_f32(do, v, dP, Nq, Nf, dv, sam=Df, sak=1, sbk=1, sbn=Df, sa=sO, sb=sV, sc=sP, **bh)               # dP = dO V^T
        call("softmax_bwd", p=P, dp=dP, g=None, rows=B * heads * Nq, n=Nf, heads=heads, N=Nq, scale=scale)
        dq_, dk_ = torch.empty_like(q), torch.empty_like(k)
        gemm_f32(dP, k, dq_, Nq, dq, Nf, sam=Nf, sak=1, sbk=D, sbn=1, ldc=D, sa=sP, sb=sK, sc=sQ, **bh)        # dQ = dS K
        gemm_f32(dP, q, dk_, Nf, dq, Nq, sam=1, sak=Nf, sbk=D, sbn=1, ldc=D, sa=sP, sb=sQ, sc=sK, **bh)        # dK = dS^T Q
        tp.acc(q, dq_)
        tp.acc(k, dk_)
        tp.acc(v, dv_)
    tp.ops.append(bwd)
    return o


def lfca_train(tp, L, prefix, qin, C, feat, dp=(None, None)):
    """LFCA.forward (mctformer.py:124-166) on the tape; dp = the DropPath factors of :163 and :164 (None: eval semantics).  qin [B,Nq,D]: the (detached) query of a ViT block;
    feat NHWC [B,hd,wd,5*Df] fp32.  -> (x_c [B,C,D], x_p NHWC [B,hd,wd,Df])."""
    B, Nq, D = qin.shape
    N = Nq - C
    ht = wt = int(N ** 0.5)
    _, hd, wd, F5 = feat.shape
    if (ht, wt) != (hd, wd):
        raise NotImplementedError("dig_amd.train: the LFCA branch is differentiated for guidance maps of the ViT grid's size only "
                                  f"({ht}x{wt} vs {hd}x{wd}: the bilinear resize of mctformer.py:161 has no adjoint kernel yet)")
    di = L.diff_init
    h1 = conv3(tp, feat, di.convbnrelu1[0].weight, None, prefix + "diff_init.convbnrelu1.0", need_dx=False)
    a1 = relu(tp, gn1(tp, h1, di.convbnrelu1[1], prefix + "diff_init.convbnrelu1.1", 1e-5))
    h2 = gn1(tp, conv3(tp, a1, di.convbn2[0].weight, None, prefix + "diff_init.convbn2.0"), di.convbn2[1], prefix + "diff_init.convbn2.1", 1e-5)
    br = lin(tp, feat.reshape(-1, F5), di.branch.weight, None, prefix + "diff_init.branch", need_dx=False).reshape(h2.shape)
    kv = relu(tp, add(tp, h2, br))
    Df = kv.shape[-1]
    q = ln(tp, lin(tp, qin.reshape(-1, D), L.query.weight, L.query.bias, prefix + "query", need_dx=False).reshape(B, Nq, D), L.norm_k, prefix + "norm_k")
    k = ln(tp, lin(tp, kv.reshape(-1, Df), L.key.weight, L.key.bias, prefix + "key").reshape(B, hd * wd, D), L.norm_k, prefix + "norm_k")
    v = ln(tp, conv3(tp, kv, L.value.weight, L.value.bias, prefix + "value").reshape(B, hd * wd, Df), L.norm_v, prefix + "norm_v")
    o = xattn(tp, q, k, v, 8, 1.0 / ((384 // 8) ** 0.5))  # the reference hard-codes 384 // 8 (mctformer.py:146)
    o_c = o[:, :C].clone()   # clones: the tape keys gradients by data pointer, a view at offset 0 would alias o
    o_p = o[:, C:].reshape(B, ht, wt, Df).clone()

    def split_bwd(o_c=o_c, o_p=o_p):  # bound now: o_p is rebound below
        do = torch.zeros_like(o)
        gc, gp = tp.grad(o_c), tp.grad(o_p)
        if gc is not None:
            do[:, :C] = gc
        if gp is not None:
            do[:, C:] = gp.reshape(B, N, Df)
        tp.acc(o, do)
    tp.ops.append(split_bwd)
    # class-token branch (mctformer.py:153-158)
    x_c = lin(tp, o_c.reshape(-1, Df), L.mlp_c0.weight, L.mlp_c0.bias, prefix + "mlp_c0", res=qin[:, :C].reshape(-1, D).contiguous())
    y = ln(tp, x_c, L.norm_c1, prefix + "norm_c1")
    hdn = gelu(tp, lin(tp, y, L.mlp_c1[0].weight, L.mlp_c1[0].bias, prefix + "mlp_c1.0"))
    x_c = add(tp, lin(tp, hdn, L.mlp_c1[2].weight, L.mlp_c1[2].bias, prefix + "mlp_c1.2"), x_c)
    # patch branch (mctformer.py:160-164); equal map sizes: F.interpolate is the identity
    o_p = conv3(tp, o_p, L.mlp_p0.weight, L.mlp_p0.bias, prefix + "mlp_p0")
    x_p = add(tp, kv, dpath(tp, o_p, dp[0]))
    y = gn1(tp, x_p, L.norm_p1, prefix + "norm_p1", 1e-6)
    hdn = gelu(tp, lin(tp, y.reshape(-1, Df), L.mlp_p1[0].weight, L.mlp_p1[0].bias, prefix + "mlp_p1.0"))
    x_p = add(tp, dpath(tp, lin(tp, hdn, L.mlp_p1[2].weight, L.mlp_p1[2].bias, prefix + "mlp_p1.2").reshape(x_p.shape), dp[1]), x_p)
    return x_c.reshape(B, C, D), x_p


def lfca_branch_train(m, ctx, feat_diff, dp=None):
    """The LFCA branch of MCTformerV2_DiG.forward (mctformer.py:265-291,324-346) on a tape, from the kept qkv of the last two
    ViT blocks of a forward_train pass.  -> (outs {cam_diff, diff_cls, diff_pcls}, tape state for lfca_branch_backward)."""
    B, C, D, heads, dh, hp, wp, Pn, N, ps = ctx["dims"]
    dev = ctx["x"].device
    hd, wd = feat_diff[0].shape[2:]
    cat = torch.stack([t.float() for t in feat_diff], dim=1).reshape(B, -1, hd, wd).contiguous()  # timestep-major channels (:275)
    feat = ops.nchw_to_nhwc(cat, F32)
    tp = Tape()
    q0 = ctx["saved"][-1]["qkv"][..., :D].contiguous()   # query0 = q of the last block, query1 = of the one before (:279-281)
    q1 = ctx["saved"][-2]["qkv"][..., :D].contiguous()
    dp = dp or {}
    c0, p0 = lfca_train(tp, m.LFCA0, "LFCA0.", q0, C, feat, dp=dp.get("LFCA0", (None, None)))
    c1, p1 = lfca_train(tp, m.LFCA1, "LFCA1.", q1, C, feat, dp=dp.get("LFCA1", (None, None)))
    out_c = add(tp, c0, c1, 0.5, 0.5)
    out_p = add(tp, p0, p1, 0.5, 0.5)
    hh = relu(tp, conv3(tp, out_p, m.diff_head_p[0].weight, m.diff_head_p[0].bias, "diff_head_p.0"))
    Df = hh.shape[-1]
    h2 = lin(tp, hh.reshape(-1, Df), m.diff_head_p[2].weight, m.diff_head_p[2].bias, "diff_head_p.2")  # [B*hd*wd, Cd] = cam_diff, pre-ReLU
    Cd = h2.shape[-1]
    cam_diff = torch.empty(B, Cd, hd, wd, dtype=F32, device=dev)
    diff_pcls = torch.empty(B, Cd, dtype=F32, device=dev)
    call("head_pool", **{"in": h2}, relu_nchw=cam_diff, pooled=diff_pcls, B=B, P=hd * wd, C=Cd)
    diff_cls = torch.empty(B, C, dtype=F32, device=dev)
    call("row_mean", x0=out_c, x1=None, out=diff_cls, bstride0=C * D, bstride1=0, B=B, C=C, D=D)
    st = dict(tape=tp, out_c=out_c, h2=h2, dims=(B, C, D, hd * wd, Cd))
    return {"cam_diff": cam_diff, "diff_cls": diff_cls, "diff_pcls": diff_pcls}, st


def lfca_branch_backward(st, d_diff_cls, d_diff_pcls):
    """d_diff_cls / d_diff_pcls [B,C]: gradients of the loss w.r.t. the branch's two logit outputs (already scaled).
    -> {state_dict key: gradient} of LFCA0 / LFCA1 / diff_head_p."""
    tp = st["tape"]
    B, C, D, P, Cd = st["dims"]
    dev = d_diff_cls.device
    dh2 = torch.empty(B, P, Cd, dtype=F32, device=dev)
    call("head_pool_bwd", head=st["h2"], dpooled=d_diff_pcls.contiguous(), dhead=dh2, B=B, P=P, C=Cd, Cp=Cd)
    tp.acc(st["h2"], dh2.reshape(st["h2"].shape))
    tp.acc(st["out_c"], scaled(d_diff_cls, 1.0 / D).unsqueeze(-1).expand(B, C, D).contiguous())   # out_c.mean(-1)
    tp.backward()
    return tp.pg
