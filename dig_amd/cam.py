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
from ._lib import call

F32 = torch.float32


def _pack(m):
    key = (next(m.parameters()).device, sum(p._version for p in m.parameters()), m.cls_token.data_ptr())
    if m._packed is not None and m._packed["key"] == key:
        return m._packed
    f = lambda t: t.detach().float().contiguous()
    P = {"key": key}
    P["pe_w"] = f(m.patch_embed.proj.weight.reshape(m.embed_dim, -1))  # [384, 3*16*16], K = (c, ky, kx)
    P["pe_b"] = f(m.patch_embed.proj.bias)
    P["cls"], P["pos"] = f(m.cls_token[0]), f(m.pos_embed[0])
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


def mctformer_forward(m, img, n_layers=12):
    """-> dict with rcams, mtatt, attn (patch_attn), cams, fcams, cls, pcls, feat (mctformer.py:331-350 keys)."""
    if not img.is_cuda:
        raise RuntimeError("dig_amd.MCTformerV2_DiG runs on the HIP device only (no CPU fallback)")
    P = _pack(m)
    B, _, H, W = img.shape
    C, D, heads, ps = m.num_classes, m.embed_dim, m.num_heads, m.patch_embed.patch_size[0]
    hp, wp = H // ps, W // ps
    Pn, N = hp * wp, C + hp * wp
    dev = img.device
    img = img.contiguous().float()
    pos = _pos_embed(m, P, H, W)
    # patch embedding: im2row + GEMM(+bias, +pos) written straight into the token buffer behind the class tokens
    rows = torch.empty(B * Pn, 3 * ps * ps, dtype=F32, device=dev)
    call("patchify", img=img, rows=rows, B=B, H=H, W=W, ps=ps)
    x = torch.empty(B, N, D, dtype=F32, device=dev)
    x[:, :C] = P["cls"] + pos[:C]  # broadcast copy of the class-token rows
    ops.gemm_nt(rows, P["pe_w"], (x, C * D), Pn, D, 3 * ps * ps, dtype=F32, bias=P["pe_b"], res=(pos, C * D), ldr=D,
                batch=B, sa=(Pn * 3 * ps * ps, 0), sc=(N * D, 0), sr=(0, 0))
    acc_all = torch.zeros(B, N, N, dtype=F32, device=dev)
    acc_cls = torch.zeros(B, C, N, dtype=F32, device=dev)
    depth = len(P["blocks"])
    for li, bp in enumerate(P["blocks"]):
        y = _layernorm(x, bp["n1"])
        qkv = ops.linear_rows(y, bp["qkv"][0], bias=bp["qkv"][1])
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
    relu_head = head.permute(0, 3, 1, 2).clamp_min(0)
    k = min(3, Pn)
    hv = head.permute(0, 3, 1, 2).reshape(B, C, -1)
    pcls = hv.clamp_min(0).topk(k, dim=-1)[0].mean(-1) - (-hv).clamp_min(0).topk(k, dim=-1)[0].mean(-1)  # mctformer.py:300
    return {"cls": x[:, :C].mean(-1), "pcls": pcls, "cams": relu_head, "fcams": cams, "attn": acc_all[:, C:, C:],
            "mtatt": mtatt, "rcams": rcams, "feat": x[:, C:].reshape(B, hp, wp, D).permute(0, 3, 1, 2)}


def infer_multi_cam(m, img, label, n_layers=12):
    """model.py:468-477: CAM of the image and of its horizontal flip, bilinear upsample x label, sum, max_norm.
    img [B,3,H,W] (H, W multiples of 16), label [B,C] -> normalised CAM [B,C,H,W] fp32."""
    B, _, H, W = img.shape
    C = m.num_classes
    both = torch.cat([img, torch.flip(img, (3,))], 0)  # one batched forward for the image and its flip
    rc = mctformer_forward(m, both, n_layers)["rcams"]
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
