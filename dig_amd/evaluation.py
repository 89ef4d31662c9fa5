"""CAM evaluation on the device: the reference's evaluation.py (mIoU of CAM label maps against the segmentation ground
truth, threshold sweep) with the per-image numpy/multiprocessing loop replaced by two kernels - dig_cam_argmax
(evaluation.py:42-50) and dig_seg_hist (evaluation.py:72-88).  The on-disk cam_dict format is the reference's
(models/model_ECCV24_DiG.py:479-492): np.save of {class index -> float32 [H, W]}.
"""
import os

import numpy as np
import torch

from ._lib import call

CATEGORIES = ['background', 'aeroplane', 'bicycle', 'bird', 'boat', 'bottle', 'bus', 'car', 'cat', 'chair', 'cow',
              'diningtable', 'dog', 'horse', 'motorbike', 'person', 'pottedplant', 'sheep', 'sofa', 'train', 'tvmonitor']


def save_cam_dict(path, cam_dict, reference_format=False):
    """Write {class index -> float32 [H, W]}; returns the path written.  Default: an .npz archive of plain arrays (one entry
    per class index), which loads without executing anything; a trailing '.npy' of `path` is replaced by '.npz'.  reference_format=True writes what model.py:489-492 writes - np.save of the dict
    itself, i.e. a pickle inside the .npy - for interchange with the reference's evaluation.py."""
    d = {int(k): np.asarray(v, np.float32) for k, v in cam_dict.items()}
    path = str(path)
    if reference_format:
        np.save(path, d)
        return path if path.endswith(".npy") else path + ".npy"
    # the reference call site passes name + '.npy' (model.py:489-492): the archive takes the SAME stem, <name>.npz, which is
    # where do_python_eval looks first (never <name>.npy.npz)
    if path.endswith(".npy"):
        path = path[:-4]
    if not path.endswith(".npz"):
        path += ".npz"
    np.savez(path, **{str(k): v for k, v in d.items()})
    return path


def load_cam_dict(path, reference_format=False):
    """Read a cam_dict.  .npz files (save_cam_dict's default) are read with allow_pickle=False.  The reference's .npy files
    (evaluation.py:44: np.load(file, allow_pickle=True).item()) hold a pickled dict; unpickling executes code from the file,
    so that loader runs only when the caller opts in with reference_format=True - use it on files you wrote yourself."""
    path = str(path)
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return {int(k): z[k] for k in z.files}
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: no such cam_dict file (looked for the .npz archive of the same stem first)")
    if not reference_format:
        raise ValueError(f"{path}: a reference-format .npy cam_dict is a pickle; pass reference_format=True to load it "
                         "(only for files you trust), or save with save_cam_dict(..., reference_format=False)")
    return np.load(path, allow_pickle=True).item()


class SegCounter:
    """Accumulates P / T / TP per class over images (evaluation.py:30-36,78-88) in one uint64 device buffer."""

    def __init__(self, num_cls=21, device="cuda"):
        self.num_cls = num_cls
        self.counts = torch.zeros(3, num_cls, dtype=torch.int64, device=device)

    def add(self, predict, gt):
        """predict, gt: uint8 tensors of equal shape on the device (255 in gt is ignored)."""
        assert predict.is_cuda and predict.dtype == torch.uint8 and gt.dtype == torch.uint8 and predict.shape == gt.shape
        call("seg_hist", predict=predict.contiguous(), gt=gt.contiguous(), counts=self.counts, n=predict.numel(), num_cls=self.num_cls)

    def add_cam(self, norm_cam, label, gt, threshold):
        """One batch straight from the normalised CAMs [B,C,H,W] (zeroed where label <= 1e-5 by dig_cam_argmax)."""
        from . import cam
        self.add(cam.cam_labels(norm_cam, label, threshold), gt)

    def result(self):
        """evaluation.py:96-122: per-class IoU (percent), mIoU (percent), mean precision and recall."""
        P, T, TP = (self.counts[i].cpu().numpy().astype(np.float64) for i in range(3))
        return metrics_from_counts(P, T, TP)


def metrics_from_counts(P, T, TP, categories=None):
    iou = TP / (T + P - TP + 1e-10)
    precision = TP / (P + 1e-10)
    recall = TP / (T + 1e-10)
    names = categories if categories is not None else (CATEGORIES if len(iou) == len(CATEGORIES) else [str(i) for i in range(len(iou))])
    out = {names[i]: float(iou[i] * 100) for i in range(len(iou))}
    out["mIoU"] = float(np.mean(iou) * 100)
    out["mP"] = float(np.mean(precision))
    out["mR"] = float(np.mean(recall))
    return out


def do_python_eval(predict_folder, gt_folder, name_list, num_cls=21, task="cam", threshold=0.45, device="cuda",
                   reference_format=False):
    """Same arguments and result dict as evaluation.py:29-129 for task 'cam' and 'png'; files are read on the host, the
    argmax and the counting run on the device.  CAM dicts are looked up as <name>.npz first; the reference's pickled
    <name>.npy is read only with reference_format=True (see load_cam_dict)."""
    from PIL import Image
    assert task in ("cam", "png"), "only the CAM-dict and PNG tasks are on this path"
    ctr = SegCounter(num_cls, device)
    for name in name_list:
        gt = torch.from_numpy(np.array(Image.open(os.path.join(gt_folder, "%s.png" % name))).astype(np.uint8)).to(device)
        if task == "cam":
            npz = os.path.join(predict_folder, "%s.npz" % name)
            d = load_cam_dict(npz) if os.path.exists(npz) else load_cam_dict(os.path.join(predict_folder, "%s.npy" % name), reference_format)
            h, w = next(iter(d.values())).shape
            camt = torch.zeros(1, num_cls - 1, h, w, dtype=torch.float32, device=device)
            lab = torch.zeros(1, num_cls - 1, dtype=torch.float32, device=device)
            for k, v in d.items():
                camt[0, k] = torch.from_numpy(np.asarray(v, np.float32)).to(device)
                lab[0, k] = 1.0
            ctr.add_cam(camt, lab, gt.reshape(1, h, w), threshold)
        else:
            pred = torch.from_numpy(np.array(Image.open(os.path.join(predict_folder, "%s.png" % name))).astype(np.uint8)).to(device)
            ctr.add(pred, gt)
    return ctr.result()


def threshold_sweep(eval_fn, lo=35, hi=70):
    """evaluation.py:159-175: thresholds i/100 for i in [lo, hi), stop at the first one that does not improve mIoU.
    eval_fn(threshold) -> result dict.  Returns (max_miou, max_th, precision, recall)."""
    max_miou, max_th, precision, recall = 0, 0, None, None
    for i in range(lo, hi):
        t = i / 100.0
        r = eval_fn(t)
        if r["mIoU"] > max_miou:
            max_miou, max_th, precision, recall = r["mIoU"], t, r["mP"], r["mR"]
        else:
            break
    return max_miou, max_th, precision, recall
