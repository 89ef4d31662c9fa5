"""Guidance-extraction glue of models/model_ECCV24_DiG.py:286-316 on the HIP path.

The reference runs six separate Unet passes (t = 0,1,2,3,4 and 100) per training batch.  The five low-t passes
share x0 and differ only in (a_t, s_t, noise, time embedding), so here they run as ONE batch of 5B images
(SURVEY.md 8a-15); denormforDiff + normalize are folded into the stem kernel's per-channel affine and
predict_start_from_v + unnormalize + normforCls into one elementwise kernel."""
import torch

from ._lib import call
from .diffusion import _mark_t_checked

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
T_LIST = (0, 1, 2, 3, 4, 100)  # model.py:291,296


def extract_guidance(diffusion, img, noises=None, t_list=T_LIST, sfi=100, batched=True):
    """img: ImageNet-normalised [B,3,H,W] on the HIP device.  noises: optional list (one [B,3,H,W] per entry of
    t_list; the reference draws them on device, ddp.py:865).  Returns (feat_diff_list, img_diff):
    len(t_list)-1 tensors [B,1024,H/16,W/16] fp32 (== torch.cat(feat, 1), model.py:313) and the t=sfi
    reconstruction normalised for the classifier (model.py:304-310)."""
    assert img.is_cuda, "dig_amd.guidance runs on the HIP device only"
    B = img.shape[0]
    dev = img.device
    if any(not (0 <= int(tv) < diffusion.num_timesteps) for tv in t_list):
        raise IndexError(f"timestep out of range: t_list {tuple(t_list)} must lie in [0, {diffusion.num_timesteps})")
    if noises is None:
        noises = [torch.randn_like(img) for _ in t_list]
    # x0 = ((img*std + mean)*2 - 1)  ->  pre_scale = 2*std, pre_shift = 2*mean - 1
    pre_scale = tuple(2.0 * s for s in IMAGENET_STD)
    pre_shift = tuple(2.0 * m - 1.0 for m in IMAGENET_MEAN)
    low = [j for j in range(len(t_list)) if t_list[j] != sfi]
    maps = {}
    if batched and len(low) > 1:
        xs = img.repeat(len(low), 1, 1, 1)
        nz = torch.cat([noises[j] for j in low], 0)
        tt = _mark_t_checked(torch.cat([torch.full((B,), t_list[j], dtype=torch.long, device=dev) for j in low]), diffusion.num_timesteps)
        diffusion.p_losses(xs, tt, noise=nz, pre_scale=pre_scale, pre_shift=pre_shift, want_pred=False)  # v of the low-t passes is discarded (model.py:302,312)
        allm = diffusion.model._last_maps
        for i, j in enumerate(low):
            maps[j] = allm[i * B:(i + 1) * B]
    else:
        for j in low:
            tt = _mark_t_checked(torch.full((B,), t_list[j], dtype=torch.long, device=dev), diffusion.num_timesteps)
            diffusion.p_losses(img, tt, noise=noises[j], pre_scale=pre_scale, pre_shift=pre_shift, want_pred=False)
            maps[j] = diffusion.model._last_maps
    img_diff = None
    for j in range(len(t_list)):
        if t_list[j] == sfi:
            tt = _mark_t_checked(torch.full((B,), sfi, dtype=torch.long, device=dev), diffusion.num_timesteps)
            x_t, v, _ = diffusion.p_losses(img, tt, noise=noises[j], pre_scale=pre_scale, pre_shift=pre_shift)
            # normforCls(unnormalize(predict_start_from_v)) = ((a*x_t - s*v + 1)/2 - mean)/std
            img_diff = diffusion._axpby(x_t, v, tt, -1.0, post_scale=tuple(0.5 / s for s in IMAGENET_STD),
                                        post_shift=tuple((0.5 - m) / s for m, s in zip(IMAGENET_MEAN, IMAGENET_STD)))
    return [maps[j] for j in low], img_diff


def update_forward(diffusion, net_trm, img, noises=None, n_layers=12):
    """The forward computations of model_WSSS.update (models/model_ECCV24_DiG.py:286-322,396-404), inference mode:
    guidance maps at t = 0..4 and the t = 100 reconstruction from the diffusion Unet, the MCTformer pass that consumes the
    maps through LFCA0/LFCA1 (`net_trm(img, feat_diff_list)`), and the second MCTformer pass on the reconstruction
    (`net_trm(img_diff)`).  The loss terms that follow (model.py:337-421) are dig_amd.losses.update_losses(outputs,
    outputs_diff, label, ...); the backward pass and the optimiser step (model.py:424-425) are outside this build.
    Returns (outputs, outputs_diff, feat_diff_list, img_diff)."""
    feat_diff_list, img_diff = extract_guidance(diffusion, img, noises)
    outputs = net_trm(img, feat_diff=feat_diff_list, n_layers=n_layers)
    outputs_diff = net_trm(img_diff, n_layers=n_layers)
    return outputs, outputs_diff, feat_diff_list, img_diff
