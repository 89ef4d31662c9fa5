"""Forward losses of model_WSSS.update (models/model_ECCV24_DiG.py:337-421) on the HIP kernels of csrc/losses.hip.

A first slice of the training step: the five loss terms and their weighted total computed from the outputs of the two
MCTformer passes (dig_amd.guidance.update_forward), as device scalars (no host synchronisation).  The backward pass and
the optimiser step that follow in the reference (model.py:424-425) are not built, and the reference computes these in
train mode (dropout / DropPath active) whereas dig_amd's MCTformer is forward-only (eval)."""
import torch

from ._lib import call

LOSS_NAMES = ("loss_cls", "loss_pred_diff", "loss_cam_consistency", "loss_cls_diff", "loss_rcams_consistency")
LFCA_EPOCH = 9  # model.py:281


def _f(x):
    return x.contiguous().float()


def _msm(x, y, out, scale, accumulate):
    B, C = x.shape
    call("msm_loss", x=_f(x), y=y, out=out, scale=float(scale), accumulate=int(accumulate), B=B, C=C)


def _kd(student, teacher, out, T, scale, accumulate):
    B, C = student.shape
    call("kd_bin_loss", student=_f(student), teacher=_f(teacher), out=out, T=float(T), scale=float(scale), accumulate=int(accumulate), B=B, C=C)


def _maxnorm_l1(a, b, label, out, scale, a_times_label):
    B, C, ha, wa = a.shape
    hb, wb = b.shape[2:]
    partial = torch.empty(B * C, dtype=torch.float32, device=a.device)
    call("maxnorm_l1", a=_f(a), b=_f(b), label=label, partial=partial, B=B, C=C, ha=ha, wa=wa, hb=hb, wb=wb, a_times_label=int(a_times_label))
    call("sum_scale", **{"in": partial}, out=out, scale=float(scale) / (B * C * ha * wa), accumulate=0, n=B * C)


def update_losses(outputs, outputs_diff, label, W=(1.0, 1.0, 1.0, 1.0, 10.0), T=5.0, A=0.2, epo=0):
    """model.py:337-421.  outputs = net_trm(img, feat_diff_list), outputs_diff = net_trm(img_diff) (or None when
    W[3] = W[4] = 0); label [B,C] multi-hot; W/T/A = args.W/args.T/args.A (train_trm.py:66-69 defaults); epo gates the two
    consistency terms (epo > 9).  -> dict of fp32 device tensors of shape [1]: the five LOSS_NAMES and 'loss_trm'."""
    if not label.is_cuda:
        raise RuntimeError("dig_amd.losses runs on the HIP device only (no CPU fallback)")
    dev = label.device
    label = _f(label)
    L = torch.zeros(len(LOSS_NAMES), dtype=torch.float32, device=dev)  # one slot per term; zero = the reference's torch.Tensor([0])
    s = [L[i:i + 1] for i in range(len(LOSS_NAMES))]
    have_diff = "diff_cls" in outputs
    # loss_cls (model.py:337-340)
    _msm(outputs["cls"], label, s[0], W[0], False)
    _msm(outputs["pcls"], label, s[0], W[0], True)
    if W[1] > 0:  # loss_pred_diff (model.py:346-363)
        if not have_diff:
            raise KeyError("W[1] > 0 needs the LFCA outputs (diff_cls / diff_pcls): run net_trm with feat_diff")
        _msm(outputs["diff_cls"], label, s[1], W[1] * A, False)
        _msm(outputs["diff_pcls"], label, s[1], W[1] * A, True)
        _kd(outputs["diff_cls"], outputs["cls"], s[1], T, W[1] * (1 - A) * (T * T) * 0.5, True)
        _kd(outputs["diff_pcls"], outputs["pcls"], s[1], T, W[1] * (1 - A) * (T * T) * 0.5, True)
    if W[2] > 0 and epo > LFCA_EPOCH:  # loss_cam_consistency (model.py:367-391)
        _maxnorm_l1(outputs["cam_diff"], outputs["cams"], label, s[2], W[2], True)
    if len(W) > 3 and (W[3] > 0 or W[4] > 0):
        if outputs_diff is None:
            raise KeyError("W[3] / W[4] > 0 need the second pass outputs_diff = net_trm(img_diff)")
        _msm(outputs_diff["cls"], label, s[3], W[3], False)  # loss_cls_diff (model.py:406-410)
        _msm(outputs_diff["pcls"], label, s[3], W[3], True)
        if W[4] > 0 and epo > LFCA_EPOCH:  # loss_rcams_consistency (model.py:414-419)
            _maxnorm_l1(outputs["rcams"], outputs_diff["rcams"], label, s[4], W[4], False)
    total = torch.empty(1, dtype=torch.float32, device=dev)
    call("sum_scale", **{"in": L}, out=total, scale=1.0, accumulate=0, n=len(LOSS_NAMES))
    out = {n: s[i] for i, n in enumerate(LOSS_NAMES)}
    out["loss_trm"] = total
    return out
