"""TEST INFRASTRUCTURE ONLY - loader for the *reference* DiG python modules.

Only usable inside the build container (``/root/reference`` is mounted there and
nowhere else).  Used by ``oracle/gen_golden.py`` to (i) validate the CPU
restatement in ``oracle/dig_oracle.py`` and (ii) emit the golden fixtures that
are committed under ``tests/golden/``.  Nothing under ``dig_amd/`` may import
this file, and nothing here is shipped or timed.

The reference imports a number of packages that are absent from this image and
are *off* the hot path: torchvision, ema_pytorch,
pytorch_fid (only used by ``Dataset``/``Trainer``/``FIDEvaluation``,
denoising_diffusion_pytorch.py:929-1190), timm (only ``DropPath``,
``to_2tuple``, ``trunc_normal_``, ``register_model``, ImageNet mean/std
constants) and cv2 (a junk import in networks/resnet38d.py:3).  They are
replaced by inert stub modules so that the hot-path classes import unchanged.
"""
import importlib
import math
import os
import sys
import types

sys.dont_write_bytecode = True

REF_ROOT = os.environ.get("DIG_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "denoising_diffusion_pytorch"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _install_stubs():
    import torch
    import torch.nn as nn

    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.transforms = _stub("torchvision.transforms")
        tv.utils = _stub("torchvision.utils")
    if "ema_pytorch" not in sys.modules:
        _stub("ema_pytorch", EMA=object)
    if "pytorch_fid" not in sys.modules:
        pf = _stub("pytorch_fid")
        pf.fid_score = _stub("pytorch_fid.fid_score", calculate_frechet_distance=None)
        pf.inception = _stub("pytorch_fid.inception", InceptionV3=object)
    if "cv2" not in sys.modules:
        _stub("cv2", CAP_MSMF=0)
    if "timm" not in sys.modules:
        class DropPath(nn.Module):
            """Stand-in for timm.models.layers.DropPath (timm is absent from this image).  Identity in eval mode and at rate 0,
            as timm's.  In train mode it multiplies by per-sample factors the CALLER queued in .factors (each a [B] tensor of
            mask_b / keep_prob, timm's scale_by_keep form): the Bernoulli draw is not made here, so a fixture can hand the
            very same factors to the product - what the fixture pins is the reference's own block arithmetic around the
            call (vision_transformer.py:193-194, mctformer.py:163-164), not timm's random stream."""

            def __init__(self, p=0.0):
                super().__init__()
                self.p = p
                self.factors = []

            def forward(self, x):
                if not self.training or self.p == 0.0:
                    return x
                assert self.factors, "stub DropPath in train mode: queue the per-sample factors first"
                s = self.factors.pop(0)
                return x * s.view(-1, *([1] * (x.dim() - 1)))

        def to_2tuple(x):
            return tuple(x) if isinstance(x, (tuple, list)) else (x, x)

        def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
            return torch.nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

        timm = _stub("timm")
        timm.data = _stub("timm.data",
                          IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406),
                          IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
        timm.models = _stub("timm.models")
        timm.models.helpers = _stub("timm.models.helpers", load_pretrained=None)
        timm.models.layers = _stub("timm.models.layers", DropPath=DropPath,
                                   to_2tuple=to_2tuple, trunc_normal_=trunc_normal_)
        timm.models.registry = _stub("timm.models.registry", register_model=lambda f: f)


def load_ddp():
    """Return the reference ``denoising_diffusion_pytorch.denoising_diffusion_pytorch`` module
    without executing the package ``__init__`` (which pulls the unused vendored variants)."""
    _install_stubs()
    pkg = "denoising_diffusion_pytorch"
    if pkg not in sys.modules:
        p = types.ModuleType(pkg)
        p.__path__ = [os.path.join(REF_ROOT, pkg)]
        sys.modules[pkg] = p
    return importlib.import_module(pkg + ".denoising_diffusion_pytorch")


def load_mctformer():
    """Return the reference ``networks.mctformer`` module."""
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    return importlib.import_module("networks.mctformer")


def load_evaluation():
    """Return the reference ``evaluation`` module (evaluation.py at the reference root: numpy, pandas, PIL, multiprocessing)."""
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    return importlib.import_module("evaluation")
