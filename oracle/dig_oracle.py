"""TEST INFRASTRUCTURE ONLY - CPU oracle for the DiG diffusion-guidance hot path.

This file is the *checker*: a functional, fp32, CPU restatement of what the
reference computes on the path SURVEY.md section 8 lists.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product (``dig_amd``) never does and fails loudly when its HIP
library is missing.

Pinning: every function here is checked against the reference itself, imported
in the build container by ``oracle/gen_golden.py`` (max-abs deltas recorded in
``tests/golden/MANIFEST.json``), and against the committed golden vectors under
``tests/golden/`` (which travel to the GPU box; the reference does not).

All tensors are NCHW fp32 torch CPU tensors; weights are a flat ``state_dict``
mapping with the reference's key names (``GaussianDiffusion.state_dict()`` /
``MCTformerV2_DiG.state_dict()``).  Citations are ``file:line`` into
``/root/reference``; ``ddp.py`` abbreviates
``denoising_diffusion_pytorch/denoising_diffusion_pytorch.py``.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


# --------------------------------------------------------------------------------------
# schedule  (ddp.py:478-491 sigmoid schedule; ddp.py:534-601 buffers)
# --------------------------------------------------------------------------------------
def sigmoid_betas(timesteps, start=-3.0, end=3.0, tau=1.0):
    """fp64 sigmoid beta schedule, ddp.py:478-491."""
    t = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64) / timesteps
    v0 = torch.tensor(start / tau).sigmoid()  # fp32 scalars, as in the reference
    v1 = torch.tensor(end / tau).sigmoid()
    ac = (-((t * (end - start) + start) / tau).sigmoid() + v1) / (v1 - v0)
    ac = ac / ac[0]
    betas = 1 - ac[1:] / ac[:-1]
    return betas.clamp(0, 0.999)


def schedule_buffers(timesteps=1000):
    """The 13 fp32 buffers GaussianDiffusion registers for objective='pred_v' (ddp.py:555-601)."""
    b = sigmoid_betas(timesteps)
    a = 1.0 - b
    ac = torch.cumprod(a, 0)
    acp = torch.cat([torch.ones(1, dtype=torch.float64), ac[:-1]])
    pv = b * (1.0 - acp) / (1.0 - ac)
    snr = ac / (1 - ac)
    out = {
        "betas": b, "alphas_cumprod": ac, "alphas_cumprod_prev": acp,
        "sqrt_alphas_cumprod": ac.sqrt(),
        "sqrt_one_minus_alphas_cumprod": (1.0 - ac).sqrt(),
        "log_one_minus_alphas_cumprod": (1.0 - ac).log(),
        "sqrt_recip_alphas_cumprod": (1.0 / ac).sqrt(),
        "sqrt_recipm1_alphas_cumprod": (1.0 / ac - 1).sqrt(),
        "posterior_variance": pv,
        "posterior_log_variance_clipped": pv.clamp(min=1e-20).log(),
        "posterior_mean_coef1": b * acp.sqrt() / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - acp) * a.sqrt() / (1.0 - ac),
        "loss_weight": snr / (snr + 1),
    }
    return {k: v.to(torch.float32) for k, v in out.items()}


def _gather(table, t, ndim):
    """extract(), ddp.py:452-455."""
    return table.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))


def q_sample(sd, x0, t, noise):
    """ddp.py:853-860."""
    return (_gather(sd["sqrt_alphas_cumprod"], t, x0.dim()) * x0
            + _gather(sd["sqrt_one_minus_alphas_cumprod"], t, x0.dim()) * noise)


def predict_start_from_v(sd, x_t, t, v):
    """ddp.py:630-634."""
    return (_gather(sd["sqrt_alphas_cumprod"], t, x_t.dim()) * x_t
            - _gather(sd["sqrt_one_minus_alphas_cumprod"], t, x_t.dim()) * v)


# --------------------------------------------------------------------------------------
# U-Net pieces
# --------------------------------------------------------------------------------------
def _sub(sd, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


def time_embedding(w, t, dim):
    """SinusoidalPosEmb + time_mlp, ddp.py:118-130,316-321.  `w` = sub-dict under 'time_mlp.'."""
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    ang = t.to(torch.float32)[:, None] * freq[None, :]
    e = torch.cat([ang.sin(), ang.cos()], dim=-1)
    e = F.linear(e, w["1.weight"], w["1.bias"])
    e = F.gelu(e)  # exact erf GELU
    return F.linear(e, w["3.weight"], w["3.bias"])


def rms_norm(x, g):
    """RMSNorm, ddp.py:108-114: L2-normalise over C (eps 1e-12) * g * sqrt(C)."""
    n = x.pow(2).sum(1, keepdim=True).sqrt().clamp_min(1e-12)
    return x / n * g * (x.shape[1] ** 0.5)


def conv_gn_act(w, x, groups, scale_shift=None):
    """Block, ddp.py:151-167."""
    x = F.conv2d(x, w["proj.weight"], w["proj.bias"], padding=1)
    x = F.group_norm(x, groups, w["norm.weight"], w["norm.bias"], eps=1e-5)
    if scale_shift is not None:
        sc, sh = scale_shift
        x = x * (sc + 1) + sh
    return F.silu(x)


def resnet_block(w, x, temb, groups=8):
    """ResnetBlock, ddp.py:169-194."""
    ss = None
    if "mlp.1.weight" in w and temb is not None:
        e = F.linear(F.silu(temb), w["mlp.1.weight"], w["mlp.1.bias"])[:, :, None, None]
        ss = e.chunk(2, dim=1)
    h = conv_gn_act(_sub(w, "block1."), x, groups, ss)
    h = conv_gn_act(_sub(w, "block2."), h, groups)
    res = F.conv2d(x, w["res_conv.weight"], w["res_conv.bias"]) if "res_conv.weight" in w else x
    return h + res


def linear_attention(w, x, heads=4, dim_head=32):
    """LinearAttention, ddp.py:196-233 (residual added by the caller)."""
    b, c, hh, ww = x.shape
    n = hh * ww
    xn = rms_norm(x, w["norm.g"])
    qkv = F.conv2d(xn, w["to_qkv.weight"]).reshape(b, 3, heads, dim_head, n)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
    q = q.softmax(dim=-2) * dim_head ** -0.5
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(b, heads * dim_head, hh, ww)
    out = F.conv2d(out, w["to_out.0.weight"], w["to_out.0.bias"])
    return rms_norm(out, w["to_out.1.g"])


def attend(q, k, v):
    """Attend.forward math path, attend.py:98-113 (== SDPA path, attend.py:65-82, to 2.4e-6)."""
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * q.shape[-1] ** -0.5
    return torch.einsum("bhij,bhjd->bhid", sim.softmax(dim=-1), v)


def full_attention(w, x, heads=4, dim_head=32):
    """Attention, ddp.py:235-264."""
    b, c, hh, ww = x.shape
    n = hh * ww
    xn = rms_norm(x, w["norm.g"])
    qkv = F.conv2d(xn, w["to_qkv.weight"]).reshape(b, 3, heads, dim_head, n)
    q, k, v = (qkv[:, i].transpose(-1, -2) for i in range(3))  # b h n d
    o = attend(q, k, v)  # b h n d
    o = o.transpose(-1, -2).reshape(b, heads * dim_head, hh, ww)
    return F.conv2d(o, w["to_out.weight"], w["to_out.bias"])


def downsample(w, x):
    """Downsample, ddp.py:102-106: 'b c (h p1) (w p2) -> b (c p1 p2) h w' then conv1x1."""
    return F.conv2d(F.pixel_unshuffle(x, 2), w["1.weight"], w["1.bias"])


def upsample(w, x):
    """Upsample, ddp.py:90-94: nearest x2 then conv3x3."""
    return F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w["1.weight"], w["1.bias"], padding=1)


def unet_structure(sd):
    """Derive (dim, n_stages, full_attn flags) from the key set, so that the oracle follows
    whatever Unet(...) configuration produced the state dict (ddp.py:289-373)."""
    dim = sd["init_conv.weight"].shape[0]
    n = 0
    while f"downs.{n}.0.block1.proj.weight" in sd:
        n += 1
    full = [f"downs.{i}.2.to_out.weight" in sd for i in range(n)]
    return dim, n, full


def unet_forward(sd, x, t, groups=8, heads=4, dim_head=32, collect=None):
    """Unet.forward eval branch, ddp.py:379-448.  `sd` = Unet state dict (no 'model.' prefix).
    Returns (pred [B,3,H,W], [5 guidance maps at H/16 x W/16]).  `collect`, if a dict, receives
    named intermediates for layer-by-layer debugging of the HIP path."""
    dim, n_stage, full = unet_structure(sd)
    H, W = x.shape[-2:]
    assert H % 2 ** (n_stage - 1) == 0 and W % 2 ** (n_stage - 1) == 0  # ddp.py:380

    def keep(name, val):
        if collect is not None:
            collect[name] = val
        return val

    x = F.conv2d(x, sd["init_conv.weight"], sd["init_conv.bias"], padding=3)
    keep("init_conv", x)
    r = x
    temb = keep("temb", time_embedding(_sub(sd, "time_mlp."), t, dim))
    skips, feats = [], []
    for i in range(n_stage):
        p = f"downs.{i}."
        x = resnet_block(_sub(sd, p + "0."), x, temb, groups)
        skips.append(x)
        x = resnet_block(_sub(sd, p + "1."), x, temb, groups)
        att = full_attention if full[i] else linear_attention
        x = att(_sub(sd, p + "2."), x, heads, dim_head) + x
        skips.append(x)
        keep(p + "attn", x)
        if i < n_stage - 1:
            x = downsample(_sub(sd, p + "3."), x)
        else:
            x = F.conv2d(x, sd[p + "3.weight"], sd[p + "3.bias"], padding=1)
        keep(p + "out", x)
    x = resnet_block(_sub(sd, "mid_block1."), x, temb, groups)
    x = full_attention(_sub(sd, "mid_attn."), x, heads, dim_head) + x
    x = resnet_block(_sub(sd, "mid_block2."), x, temb, groups)
    keep("mid", x)
    size = (H // 16, W // 16)
    feats.append(F.interpolate(x, size=size, mode="bilinear", align_corners=True))  # ddp.py:418
    for j in range(n_stage):
        i = n_stage - 1 - j  # mirrors reversed(full_attn), ddp.py:357
        p = f"ups.{j}."
        x = resnet_block(_sub(sd, p + "0."), torch.cat([x, skips.pop()], 1), temb, groups)
        x = resnet_block(_sub(sd, p + "1."), torch.cat([x, skips.pop()], 1), temb, groups)
        att = full_attention if full[i] else linear_attention
        x = att(_sub(sd, p + "2."), x, heads, dim_head) + x
        keep(p + "attn", x)
        if j < n_stage - 1:
            x = upsample(_sub(sd, p + "3."), x)
        else:
            x = F.conv2d(x, sd[p + "3.weight"], sd[p + "3.bias"], padding=1)
        keep(p + "out", x)
        feats.append(F.interpolate(x, size=size, mode="bilinear", align_corners=True))  # ddp.py:434
    x = resnet_block(_sub(sd, "final_res_block."), torch.cat([x, r], 1), temb, groups)
    return F.conv2d(x, sd["final_conv.weight"], sd["final_conv.bias"]), feats


def p_losses_eval(sd, x_start, t, noise):
    """GaussianDiffusion.p_losses eval branch, ddp.py:862-913 -> (x_t, model_out, feat_out).
    `sd` = GaussianDiffusion state dict (schedule buffers + 'model.*')."""
    x_t = q_sample(sd, x_start, t, noise)
    v, feats = unet_forward(_sub(sd, "model."), x_t, t)
    return x_t, v, feats


def ddim_time_pairs(num_step, total_timesteps=1000, sampling_timesteps=150):
    """ddp.py:775-782: the (time, time_next) pairs of the truncated DDIM schedule, last pair ends at -1."""
    slice_idx = int((num_step / total_timesteps) * sampling_timesteps)
    times = torch.linspace(-1, total_timesteps - 1, steps=sampling_timesteps + 1)[:slice_idx]
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


def ddim_sample_from_t(sd, img, num_step, noise, sampling_timesteps=150, eta=0.0, step_noises=None):
    """GaussianDiffusion.ddim_sample_from_t(..., return_all_timesteps=True), ddp.py:770-819, objective pred_v:
    q_sample to num_step with the explicit `noise` (the reference draws it, ddp.py:785,856), then the DDIM updates with
    clip_x_start / rederive_pred_noise (ddp.py:645-668).  With eta = 0 the per-step noise is multiplied by sigma = 0.
    Returns the unnormalised stack [B, steps + 1, 3, H, W]."""
    B = img.shape[0]
    T = sd["betas"].shape[0]
    ac = sd["alphas_cumprod"]
    x = q_sample(sd, img, torch.full((B,), num_step, dtype=torch.long), noise)
    imgs = [x]
    for k, (time, time_next) in enumerate(ddim_time_pairs(num_step, T, sampling_timesteps)):
        t = torch.full((B,), time, dtype=torch.long)
        v, _ = unet_forward(_sub(sd, "model."), x, t)
        x0 = predict_start_from_v(sd, x, t, v).clamp(-1.0, 1.0)
        eps = ((_gather(sd["sqrt_recip_alphas_cumprod"], t, x.dim()) * x - x0)
               / _gather(sd["sqrt_recipm1_alphas_cumprod"], t, x.dim()))
        if time_next < 0:
            x = x0
        else:
            alpha, alpha_next = ac[time], ac[time_next]
            sigma = eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
            c = (1 - alpha_next - sigma ** 2).sqrt()
            nz = step_noises[k] if step_noises is not None else torch.zeros_like(x)
            x = x0 * alpha_next.sqrt() + c * eps + sigma * nz
        imgs.append(x)
    return (torch.stack(imgs, dim=1) + 1) * 0.5


# --------------------------------------------------------------------------------------
# guidance-extraction glue  (models/model_ECCV24_DiG.py:286-316, 499-519)
# --------------------------------------------------------------------------------------
def denorm_for_diff(img):
    """model.py:499-508 then diffusion.normalize (ddp.py:82-83)."""
    m = torch.tensor(IMAGENET_MEAN).reshape(1, 3, 1, 1)
    s = torch.tensor(IMAGENET_STD).reshape(1, 3, 1, 1)
    return (img * s + m) * 2 - 1


def norm_for_cls(img01):
    """model.py:510-519."""
    m = torch.tensor(IMAGENET_MEAN).reshape(1, 3, 1, 1)
    s = torch.tensor(IMAGENET_STD).reshape(1, 3, 1, 1)
    return (img01 - m) / s


def guidance_maps(sd, img, noises, t_list=(0, 1, 2, 3, 4, 100), sfi=100):
    """model.py:286-316.  `img` ImageNet-normalised [B,3,H,W]; `noises[j]` is the explicit noise
    for t_list[j] (the reference draws it on device, ddp.py:865).  Returns
    (list of len(t_list)-1 tensors [B,1024,H/16,W/16], img_diff [B,3,H,W])."""
    x0 = denorm_for_diff(img)
    out, img_diff = [], None
    for j, nz in zip(t_list, noises):
        t = torch.full((img.shape[0],), j, dtype=torch.long)
        x_t, v, feats = p_losses_eval(sd, x0, t, nz)
        if j == sfi:
            img_diff = norm_for_cls((predict_start_from_v(sd, x_t, t, v) + 1) * 0.5)
        else:
            out.append(torch.cat(feats, dim=1))
    return out, img_diff


# --------------------------------------------------------------------------------------
# MCTformer class-token attention -> CAM  (networks/mctformer.py, networks/vision_transformer.py)
# --------------------------------------------------------------------------------------
def vit_pos_embed(sd, num_classes, H, W, patch=16):
    """interpolate_pos_encoding, mctformer.py:196-217 (note the reference's w/h naming:
    its `w` is x.shape[2] = image height)."""
    pe = sd["pos_embed"]
    N = pe.shape[1] - num_classes
    npatch = (H // patch) * (W // patch)
    if npatch == N and H == W:
        return pe
    g = int(math.sqrt(N))
    w0, h0 = H // patch + 0.1, W // patch + 0.1
    pp = pe[:, num_classes:].reshape(1, g, g, -1).permute(0, 3, 1, 2)
    pp = F.interpolate(pp, scale_factor=(w0 / math.sqrt(N), h0 / math.sqrt(N)), mode="bicubic")
    assert int(w0) == pp.shape[-2] and int(h0) == pp.shape[-1]
    pp = pp.permute(0, 2, 3, 1).reshape(1, -1, pe.shape[-1])
    return torch.cat([pe[:, :num_classes], pp], dim=1)


def drop_path_apply(t, s):
    """DropPath in train mode with its Bernoulli draw GIVEN: t * s_b, s_b = mask_b / keep_prob per sample (timm.models.layers
    DropPath / drop_path with scale_by_keep, timm 0.3.x as the DiG repo pins it; timm is not importable here, so the rule is
    restated: the random DRAW is parity-unpinned, the arithmetic is pinned by the reference's own block code run with the same
    factors - oracle/gen_golden.py case_grads_update_droppath).  s None = eval mode / rate 0 (identity)."""
    return t if s is None else t * s.view(-1, *([1] * (t.dim() - 1)))


def vit_block(w, x, heads, want_q=False, dp=(None, None)):
    """Block + Attention + Mlp, vision_transformer.py:54-103,177-196.  Returns (x, probs) or, with want_q,
    (x, probs, q) where q is the un-scaled query [B, heads, N, D/heads] that Attention.forward hands back (:96-103).
    dp = (s_attn, s_mlp): train-mode DropPath factors of the two residual branches (:193-194), see drop_path_apply."""
    B, N, D = x.shape
    y = F.layer_norm(x, (D,), w["norm1.weight"], w["norm1.bias"], eps=1e-6)
    qkv = F.linear(y, w["attn.qkv.weight"], w["attn.qkv.bias"]).reshape(B, N, 3, heads, D // heads)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    p = ((q @ k.transpose(-2, -1)) * (D // heads) ** -0.5).softmax(dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, N, D)
    x = x + drop_path_apply(F.linear(o, w["attn.proj.weight"], w["attn.proj.bias"]), dp[0])
    y = F.layer_norm(x, (D,), w["norm2.weight"], w["norm2.bias"], eps=1e-6)
    y = drop_path_apply(F.linear(F.gelu(F.linear(y, w["mlp.fc1.weight"], w["mlp.fc1.bias"])), w["mlp.fc2.weight"], w["mlp.fc2.bias"]), dp[1])
    return (x + y, p, q) if want_q else (x + y, p)


def mctformer_cam(sd, img, n_layers=12, heads=6, patch=16, collect=None, dp=None):
    """MCTformerV2_DiG.forward(x, return_att=True) -> rcams, mctformer.py:227-263,293-321,347-348.
    Also returns the intermediates the product exposes (mtatt, patch_attn, relu(head))."""
    B, _, H, W = img.shape
    C = sd["cls_token"].shape[1]
    D = sd["cls_token"].shape[2]
    x = F.conv2d(img, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    hp, wp = x.shape[-2:]
    x = x.flatten(2).transpose(1, 2)
    x = torch.cat([sd["cls_token"].expand(B, -1, -1), x], dim=1) + vit_pos_embed(sd, C, H, W, patch)
    depth = 0
    while f"blocks.{depth}.norm1.weight" in sd:
        depth += 1
    probs = []
    qs = []
    for i in range(depth):
        x, p, q = vit_block(_sub(sd, f"blocks.{i}."), x, heads, want_q=True, dp=dp[i] if dp is not None else (None, None))
        probs.append(p.mean(dim=1))  # head mean, mctformer.py:303
        qs.append(q)
    if collect is not None:
        collect["tokens"] = x
        # the LFCA queries, mctformer.py:279-281: q of the last / second-to-last block as [B, C+N, D]
        collect["query0"] = qs[-1].permute(0, 2, 1, 3).reshape(B, -1, D).detach()  # (.detach(): mctformer.py:280-281 - the LFCA
        collect["query1"] = qs[-2].permute(0, 2, 1, 3).reshape(B, -1, D).detach()  #  branch sends no gradient into the ViT)
    # no final self.norm in the DiG subclass (mctformer.py:242-250)
    xp = x[:, C:].reshape(B, hp, wp, D).permute(0, 3, 1, 2).contiguous()
    head = F.conv2d(xp, sd["head.weight"], sd["head.bias"], padding=1)
    fmap = F.relu(head)
    A = torch.stack(probs)  # depth,B,N,N
    mtatt = A[-n_layers:].sum(0)[:, :C, C:].reshape(B, C, hp, wp)
    cams = mtatt * fmap
    patch_attn = A[:, :, C:, C:].sum(0)
    rcams = torch.matmul(patch_attn.unsqueeze(1), cams.reshape(B, C, -1, 1)).reshape(B, C, hp, wp)
    if collect is not None:
        collect.update(mtatt=mtatt, patch_attn=patch_attn, fmap=fmap, head=head)
    return rcams


def _gn1(x, w, b, eps):
    return F.group_norm(x, 1, w, b, eps)


def res_block_ysh(w, x):
    """ResBlock_ysh(in, out, groups=1), resnet101.py:31-38,195-222: conv3x3(no bias) - GroupNorm(1) - ReLU,
    conv3x3(no bias) - GroupNorm(1), + 1x1 branch (no bias), ReLU.  (`groups` is passed to both the conv and the
    GroupNorm; LFCA uses groups=1, mctformer.py:64-67.)"""
    y = F.relu(_gn1(F.conv2d(x, w["convbnrelu1.0.weight"], None, padding=1), w["convbnrelu1.1.weight"], w["convbnrelu1.1.bias"], 1e-5))
    y = _gn1(F.conv2d(y, w["convbn2.0.weight"], None, padding=1), w["convbn2.1.weight"], w["convbn2.1.bias"], 1e-5)
    return F.relu(y + F.conv2d(x, w["branch.weight"]))


def lfca(w, query_vit, feat_cat, C, heads=8, dp=(None, None)):
    """LFCA.forward, mctformer.py:124-166; dp = the two train-mode DropPath factors of :163-164 (None: eval mode, identity).
    query_vit [B, C+N, D]; feat_cat [B, 5*D_f, h, w] -> (x_c [B, C, D], x_p [B, D_f, h, w])."""
    B, _, hd, wd = feat_cat.shape
    D = query_vit.shape[-1]
    N = query_vit.shape[1] - C
    ht = wt = int(N ** 0.5)
    kv = res_block_ysh(_sub(w, "diff_init."), feat_cat)
    Df = kv.shape[1]
    nk = lambda t: F.layer_norm(t, (D,), w["norm_k.weight"], w["norm_k.bias"], 1e-6)
    q = nk(F.linear(query_vit, w["query.weight"], w["query.bias"]))
    k = nk(F.conv2d(kv, w["key.weight"], w["key.bias"]).reshape(B, D, -1).permute(0, 2, 1))
    v = F.layer_norm(F.conv2d(kv, w["value.weight"], w["value.bias"], padding=1).reshape(B, Df, -1).permute(0, 2, 1),
                     (Df,), w["norm_v.weight"], w["norm_v.bias"], 1e-6)
    q = q.reshape(B, -1, heads, D // heads).permute(0, 2, 1, 3)
    k = k.reshape(B, -1, heads, D // heads).permute(0, 2, 1, 3)
    v = v.reshape(B, -1, heads, Df // heads).permute(0, 2, 1, 3)
    a = F.softmax(torch.matmul(q, k.transpose(-2, -1)) / ((384 // 8) ** 0.5), dim=-1)  # the reference hard-codes 384//8 (:146)
    o = torch.matmul(a, v).transpose(1, 2).reshape(B, N + C, -1)
    o_c = o[:, :C]
    o_p = o[:, C:].permute(0, 2, 1).reshape(B, Df, ht, wt)
    x_c = F.linear(o_c, w["mlp_c0.weight"], w["mlp_c0.bias"]) + query_vit[:, :C]
    y = F.layer_norm(x_c, (D,), w["norm_c1.weight"], w["norm_c1.bias"], 1e-6)
    x_c = F.linear(F.gelu(F.linear(y, w["mlp_c1.0.weight"], w["mlp_c1.0.bias"])), w["mlp_c1.2.weight"], w["mlp_c1.2.bias"]) + x_c
    o_p = F.conv2d(o_p, w["mlp_p0.weight"], w["mlp_p0.bias"], padding=1)
    o_p = F.interpolate(o_p, size=(hd, wd), mode="bilinear", align_corners=False)
    x_p = kv + drop_path_apply(o_p, dp[0])
    y = _gn1(x_p, w["norm_p1.weight"], w["norm_p1.bias"], 1e-6)
    x_p = drop_path_apply(F.conv2d(F.gelu(F.conv2d(y, w["mlp_p1.0.weight"], w["mlp_p1.0.bias"])), w["mlp_p1.2.weight"], w["mlp_p1.2.bias"]), dp[1]) + x_p
    return x_c, x_p


def mctformer_lfca(sd, img, feat_diff, n_layers=12, dp=None):
    """MCTformerV2_DiG.forward(x, feat_diff=[5 x [B,1024,h,w]]) in eval mode: the extra dict entries of the LFCA
    branch, mctformer.py:265-291,324-346.  Returns {rcams, cam_diff, diff_cls, diff_pcls}."""
    col = {}
    dp = dp or {}
    rcams = mctformer_cam(sd, img, n_layers, collect=col, dp=dp.get("blocks"))
    B = img.shape[0]
    C = sd["cls_token"].shape[1]
    hd, wd = feat_diff[0].shape[2:]
    cat = torch.stack(feat_diff, dim=1).reshape(B, -1, hd, wd)  # timestep-major channel order (:275)
    c0, p0 = lfca(_sub(sd, "LFCA0."), col["query0"], cat, C, dp=dp.get("LFCA0", (None, None)))
    c1, p1 = lfca(_sub(sd, "LFCA1."), col["query1"], cat, C, dp=dp.get("LFCA1", (None, None)))
    out_c, out_p = (c0 + c1) / 2, (p0 + p1) / 2
    y = F.relu(F.conv2d(out_p, sd["diff_head_p.0.weight"], sd["diff_head_p.0.bias"], padding=1))
    cam_diff = F.conv2d(y, sd["diff_head_p.2.weight"], sd["diff_head_p.2.bias"])
    top = lambda t: torch.topk(t.reshape(B, cam_diff.shape[1], -1), dim=-1, k=3)[0].mean(-1)
    return {"rcams": rcams, "cam_diff": F.relu(cam_diff), "diff_cls": out_c.mean(-1),
            "diff_pcls": top(F.relu(cam_diff)) - top(F.relu(-cam_diff))}


def mctformer_outputs(sd, img, feat_diff=None, n_layers=12, dp=None):
    """The entries of MCTformerV2_DiG.forward's output dict that model_WSSS.update consumes (mctformer.py:293-346,
    model.py:324-335): cls, pcls, cams (= relu(head)), mtatt, rcams and, with feat_diff, cam_diff / diff_cls / diff_pcls.
    dp: train-mode DropPath factors of this pass {"blocks": [(s_attn, s_mlp)] * depth, "LFCA0": (s, s), "LFCA1": (s, s)}."""
    col = {}
    B = img.shape[0]
    C = sd["cls_token"].shape[1]
    dp = dp or {}
    if feat_diff is not None:
        out = mctformer_lfca(sd, img, feat_diff, n_layers, dp=dp)
        mctformer_cam(sd, img, n_layers, collect=col, dp=dp.get("blocks"))
    else:
        out = {"rcams": mctformer_cam(sd, img, n_layers, collect=col, dp=dp.get("blocks"))}
    head = col["head"]
    top = lambda t: torch.topk(t.reshape(B, C, -1), dim=-1, k=3)[0].mean(-1)
    out.update(cls=col["tokens"][:, :C].mean(-1), pcls=top(F.relu(head)) - top(F.relu(-head)), cams=col["fmap"], mtatt=col["mtatt"])
    return out



# --------------------------------------------------------------------------------------
# (f)-4: backward of the classification losses through the MCTformer + the AdamW step (model.py:337-341,406-410,424-425)
# --------------------------------------------------------------------------------------
def mct_cls_loss(sd, img, label, w0=1.0):
    """loss_cls of model_WSSS.update (model.py:337-340) - also the form of loss_cls_diff (:406-409) on the reconstructed
    image: W * (multilabel_soft_margin(cls) + multilabel_soft_margin(pcls)) of one MCTformer pass without the LFCA branch."""
    o = mctformer_outputs(sd, img)
    return w0 * (F.multilabel_soft_margin_loss(o["cls"], label) + F.multilabel_soft_margin_loss(o["pcls"], label))


def mct_cls_loss_grads(sd, img, label, w0=1.0):
    """loss_trm.backward() (model.py:424) restricted to the loss_cls term, by autograd over this file's own forward
    restatement: -> (loss, {state_dict key: gradient}) for every parameter the loss reaches (eval mode: no DropPath)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    with torch.enable_grad():
        loss = mct_cls_loss(leaves, img, label, w0)
        names = [k for k in leaves]
        grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    return float(loss.detach()), {k: g for k, g in zip(names, grads) if g is not None}


def mct_consistency_loss_grads(sd, img, img_diff, cam_diff, label, W=(1.0, 0.0, 1.0, 1.0, 10.0)):
    """loss_trm.backward() (model.py:424) for every term that does not differentiate through the LFCA branch:
      W[0] loss_cls (:337-340) + W[2] loss_cam_consistency (:368-390, cam_diff enters detached) + W[3] loss_cls_diff (:406-409)
      + W[4] loss_rcams_consistency (:414-418, the first pass's rcams detached),
    by autograd over this file's forward restatement of the two MCTformer passes (eval mode).  cam_diff: the LFCA output of the
    first pass (any tensor of its shape: it is a constant of the loss).  W[1] (loss_pred_diff) must be 0 here.
    -> (loss_trm, {state_dict key: gradient})."""
    assert W[1] == 0
    B, C = label.shape
    lab4 = label.view(B, C, 1, 1)
    msm = F.multilabel_soft_margin_loss
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    with torch.enable_grad():
        o1 = mctformer_outputs(leaves, img)
        o2 = mctformer_outputs(leaves, img_diff)
        loss = W[0] * (msm(o1["cls"], label) + msm(o1["pcls"], label))
        if W[2] > 0:
            _cams = max_norm(F.interpolate(o1["cams"], size=cam_diff.shape[2:], mode="bilinear", align_corners=False))
            _cam_diff = max_norm(cam_diff) * lab4
            loss = loss + W[2] * ((_cam_diff.detach() - _cams) * lab4).abs().mean()
        loss = loss + W[3] * (msm(o2["cls"], label) + msm(o2["pcls"], label))
        if W[4] > 0:
            loss = loss + W[4] * ((max_norm(o1["rcams"]).detach() - max_norm(o2["rcams"])) * lab4).abs().mean()
        names = [k for k in leaves]
        grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    return float(loss.detach()), {k: g for k, g in zip(names, grads) if g is not None}


def mct_update_loss_grads(sd, img, feat_diff, img_diff, label, W=(1.0, 1.0, 1.0, 1.0, 10.0), T=5.0, A=0.2, epo=10, drop_path=None):
    """loss_trm.backward() of model_WSSS.update (model.py:322-424) with ALL five terms, by autograd over this file's forward
    restatement: first pass with the LFCA branch (its ViT queries detached, mctformer.py:280-281; teachers and cam_diff detached,
    model.py:349-350,389), second pass on img_diff.  sd must hold the LFCA0 / LFCA1 / diff_head_p parameters.
    -> (loss_trm, {state_dict key: gradient})."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    with torch.enable_grad():
        o1 = mctformer_outputs(leaves, img, feat_diff, dp=(drop_path or {}).get("pass1"))  # drop_path: {"pass1": .., "pass2": ..}
        o2 = mctformer_outputs(leaves, img_diff, dp=(drop_path or {}).get("pass2"))         # (net_trm.train(), model.py:320)
        loss = _update_loss_tensor(o1, o2, label, W, T, A, epo)
        names = [k for k in leaves]
        grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    return float(loss.detach()), {k: g for k, g in zip(names, grads) if g is not None}


def _update_loss_tensor(o1, o2, label, W, T, A, epo, lfca_epoch=9):
    """loss_trm of model.py:337-421 as a differentiable tensor (update_losses above returns floats), same expressions."""
    B, C = label.shape
    kd_log = torch.nn.KLDivLoss(reduction="batchmean", log_target=True)
    msm = F.multilabel_soft_margin_loss
    lab4 = label.view(B, C, 1, 1)
    loss = W[0] * (msm(o1["cls"], label) + msm(o1["pcls"], label))
    if W[1] > 0:
        sp, sc = F.softmax(o1["diff_pcls"] / T, dim=1), F.softmax(o1["diff_cls"] / T, dim=1)
        tp, tc = F.softmax(o1["pcls"].detach() / T, dim=1), F.softmax(o1["cls"].detach() / T, dim=1)
        lg = lambda p: torch.log(torch.cat((p, 1 - p), dim=1))
        loss = loss + W[1] * (A * msm(o1["diff_cls"], label) + A * msm(o1["diff_pcls"], label)
                              + (1 - A) * (T * T) * kd_log(lg(sc), lg(tc)) * 0.5 + (1 - A) * (T * T) * kd_log(lg(sp), lg(tp)) * 0.5)
    if W[2] > 0 and epo > lfca_epoch:
        _cams = max_norm(F.interpolate(o1["cams"], size=o1["cam_diff"].shape[2:], mode="bilinear", align_corners=False))
        loss = loss + W[2] * (((max_norm(o1["cam_diff"]) * lab4).detach() - _cams) * lab4).abs().mean()
    if W[3] > 0 or W[4] > 0:
        loss = loss + W[3] * (msm(o2["cls"], label) + msm(o2["pcls"], label))
    if W[4] > 0 and epo > lfca_epoch:
        loss = loss + W[4] * ((max_norm(o1["rcams"]).detach() - max_norm(o2["rcams"])) * lab4).abs().mean()
    return loss


def adamw_no_decay(name, p):
    """timm.optim.create_optimizer's weight-decay filter (optim_factory.add_weight_decay, timm 0.3.x - the version the DiG
    repo pins; timm itself is not importable here, so this is a restatement of its published rule and PARITY UNPINNED on
    that rule): 1-D parameters, names ending in .bias, and the model's no_weight_decay() set {pos_embed, cls_token} get
    weight_decay = 0."""
    return p.ndim <= 1 or name.endswith(".bias") or name in ("pos_embed", "cls_token")


def adamw_step(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.999, eps=1e-8):
    """One torch.optim.AdamW update (decoupled decay first, bias-corrected moments, eps added to sqrt(v_hat)); `step` is the
    1-based count after this update.  Returns (p, m, v)."""
    p = p * (1.0 - lr * wd)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * (m / denom), m, v


def update_forward(sd_diffusion, sd_mct, img, noises, n_layers=12):
    """Forward part of model_WSSS.update, model.py:286-322 (guidance maps -> net_trm(img, feat_diff_list)) and
    model.py:396-404 (net_trm(img_diff)), composed from the restated pieces; eval mode.
    Returns (LFCA-branch dict, rcams of the img_diff pass, maps, img_diff)."""
    maps, img_diff = guidance_maps(sd_diffusion, img, noises)
    outs = mctformer_lfca(sd_mct, img, maps, n_layers)
    return outs, mctformer_cam(sd_mct, img_diff, n_layers), maps, img_diff


LOSS_NAMES = ("loss_cls", "loss_pred_diff", "loss_cam_consistency", "loss_cls_diff", "loss_rcams_consistency")


def update_losses(outputs, outputs_diff, label, W=(1.0, 1.0, 1.0, 1.0, 10.0), T=5.0, A=0.2, epo=0, lfca_epoch=9):
    """The loss terms of model_WSSS.update, models/model_ECCV24_DiG.py:337-421, with the reference's own torch calls
    (F.multilabel_soft_margin_loss, nn.KLDivLoss(reduction='batchmean', log_target=True), max_norm model.py:572-578).
    -> dict of python floats: the five loss_names and loss_trm."""
    B, C = label.shape
    kd_log = torch.nn.KLDivLoss(reduction="batchmean", log_target=True)  # model.py:203
    msm = F.multilabel_soft_margin_loss
    lab4 = label.view(B, C, 1, 1)
    out = {}
    out["loss_cls"] = W[0] * (msm(outputs["cls"], label) + msm(outputs["pcls"], label))
    total = out["loss_cls"]
    zero = torch.zeros(())
    if W[1] > 0:
        sp = F.softmax(outputs["diff_pcls"] / T, dim=1)
        sc = F.softmax(outputs["diff_cls"] / T, dim=1)
        tp = F.softmax(outputs["pcls"] / T, dim=1)
        tc = F.softmax(outputs["cls"] / T, dim=1)
        lg = lambda p: torch.log(torch.cat((p, 1 - p), dim=1))
        out["loss_pred_diff"] = W[1] * (A * msm(outputs["diff_cls"], label) + A * msm(outputs["diff_pcls"], label)
                                        + (1 - A) * (T * T) * kd_log(lg(sc), lg(tc)) * 0.5
                                        + (1 - A) * (T * T) * kd_log(lg(sp), lg(tp)) * 0.5)
        total = total + out["loss_pred_diff"]
    else:
        out["loss_pred_diff"] = zero
    if W[2] > 0 and epo > lfca_epoch:
        cd = outputs["cam_diff"]
        _cams = max_norm(F.interpolate(outputs["cams"], size=cd.shape[2:], mode="bilinear", align_corners=False))
        _cam_diff = max_norm(cd) * lab4
        out["loss_cam_consistency"] = W[2] * ((_cam_diff - _cams) * lab4).abs().mean()
        total = total + out["loss_cam_consistency"]
    else:
        out["loss_cam_consistency"] = zero
    if W[3] > 0 or W[4] > 0:
        out["loss_cls_diff"] = W[3] * (msm(outputs_diff["cls"], label) + msm(outputs_diff["pcls"], label))
        total = total + out["loss_cls_diff"]
    else:
        out["loss_cls_diff"] = zero
    if W[4] > 0 and epo > lfca_epoch:
        out["loss_rcams_consistency"] = W[4] * ((max_norm(outputs["rcams"]) - max_norm(outputs_diff["rcams"])) * lab4).abs().mean()
        total = total + out["loss_rcams_consistency"]
    else:
        out["loss_rcams_consistency"] = zero
    out["loss_trm"] = total
    return {k: float(v) for k, v in out.items()}


def max_norm(cam):
    """model.py:572-578 == mctformer.py:219-225."""
    N, C, H, W = cam.shape
    cam = F.relu(cam)
    mx = cam.reshape(N, C, -1).max(dim=-1)[0].reshape(N, C, 1, 1)
    mn = cam.reshape(N, C, -1).min(dim=-1)[0].reshape(N, C, 1, 1)
    return F.relu(cam - mn - 1e-5) / (mx - mn + 1e-5)


def infer_multi_cam(sd, img, label, n_layers=12):
    """model.py:468-477: flip-TTA CAM, bilinear upsample (align_corners=False) x label, max_norm.
    Returns the normalised CAM [B,C,H,W] (the reference keeps [0] only because its batch is 1)."""
    B, _, H, W = img.shape
    C = label.shape[1]
    cam = mctformer_cam(sd, img, n_layers)
    cam = F.interpolate(cam, [H, W], mode="bilinear", align_corners=False) * label.reshape(B, C, 1, 1)
    cf = mctformer_cam(sd, torch.flip(img, (3,)), n_layers)
    cf = F.interpolate(cf, [H, W], mode="bilinear", align_corners=False) * label.reshape(B, C, 1, 1)
    return max_norm(cam + torch.flip(cf, (3,)))


def cam_dict_from(norm_cam, label):
    """model.py:479-483: keep channels whose label > 1e-5 (single image)."""
    return {i: norm_cam[i].numpy() for i in range(label.shape[0]) if label[i] > 1e-5}


def cam_argmax_labels(cam_dict, threshold, num_cls=21):
    """evaluation.py:42-50: tensor[0]=threshold, tensor[k+1]=cam_dict[k], argmax -> uint8."""
    h, w = next(iter(cam_dict.values())).shape
    t = np.zeros((num_cls, h, w), np.float32)
    for k, v in cam_dict.items():
        t[k + 1] = v
    t[0] = threshold
    return np.argmax(t, axis=0).astype(np.uint8)


def seg_counts(predicts, gts, num_cls=21):
    """evaluation.py:72-88 without the multiprocessing wrapper: P, T, TP per class as exact integers."""
    P, T, TP = (np.zeros(num_cls, np.int64) for _ in range(3))
    for predict, gt in zip(predicts, gts):
        cal = gt < 255
        mask = (predict == gt) * cal
        for i in range(num_cls):
            P[i] += np.sum((predict == i) * cal)
            T[i] += np.sum((gt == i) * cal)
            TP[i] += np.sum((gt == i) * mask)
    return P, T, TP


def seg_metrics(P, T, TP):
    """evaluation.py:96-122: IoU per class (percent), mIoU (percent), mean precision / recall."""
    iou = [TP[i] / (T[i] + P[i] - TP[i] + 1e-10) for i in range(len(P))]
    prec = [TP[i] / (P[i] + 1e-10) for i in range(len(P))]
    rec = [TP[i] / (T[i] + 1e-10) for i in range(len(P))]
    return {"IoU": [v * 100 for v in iou], "mIoU": float(np.mean(np.array(iou))) * 100, "mP": float(np.mean(np.array(prec))),
            "mR": float(np.mean(np.array(rec)))}


def cam_threshold_sweep(cam_dicts, gts, num_cls=21, lo=35, hi=70):
    """evaluation.py:159-175: sweep the background threshold, stop at the first non-improving value."""
    best = (0, 0, None, None)
    for i in range(lo, hi):
        t = i / 100.0
        preds = [cam_argmax_labels(d, t, num_cls) for d in cam_dicts]
        m = seg_metrics(*seg_counts(preds, gts, num_cls))
        if m["mIoU"] > best[0]:
            best = (m["mIoU"], t, m["mP"], m["mR"])
        else:
            break
    return best
