"""Deterministic synthetic weights and inputs.

The reference's checkpoints (model-140-pascal.pt, deit_small_imagenet.pth) are not available
offline, so parity and benchmarks run on seeded synthetic parameters.  The generator is a
counter-based hash (splitmix64 of key-hash + element index), so values depend only on
(seed, parameter name, element index) - not on torch/numpy RNG stream versions.
"""
import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def uniform01(key: str, n: int, seed: int = 0, stream: int = 0) -> np.ndarray:
    """n doubles in [0,1), a pure function of (key, seed, stream, index)."""
    base = np.uint64((_fnv1a64(key) ^ (seed * 0x2545F4914F6CDD1D) ^ (stream * 0xD6E8FEB86659FD93)) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + base) & _M64
    h = _splitmix64(idx)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform_pm1(key, shape, seed=0):
    n = int(np.prod(shape))
    return torch.from_numpy((uniform01(key, n, seed) * 2.0 - 1.0).astype(np.float32).reshape(shape))


def normal(key, shape, seed=0):
    """Standard normal via Box-Muller on two hash streams."""
    n = int(np.prod(shape))
    u1 = uniform01(key, n, seed, stream=1)
    u2 = uniform01(key, n, seed, stream=2)
    z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    return torch.from_numpy(z.astype(np.float32).reshape(shape))


def synth_param(key: str, shape, seed: int = 0) -> torch.Tensor:
    """Synthetic value for one state-dict entry, chosen so activations stay O(1)."""
    shape = tuple(shape)
    u = uniform_pm1(key, shape, seed)
    leaf = key.rsplit(".", 1)[-1]
    if leaf in ("cls_token", "pos_embed"):
        return 0.05 * u
    if leaf == "g":  # RMSNorm gain [1,C,1,1]
        return 1.0 + 0.1 * u
    if leaf == "weight" and len(shape) == 1:  # GroupNorm / LayerNorm gain
        return 1.0 + 0.1 * u
    if leaf == "bias":
        return 0.1 * u
    if leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        return u * float(np.sqrt(3.0 / fan_in))
    return 0.1 * u


def fill_state_dict(module: torch.nn.Module, seed: int = 0, skip_prefixes=()):
    """Overwrite every floating-point parameter of `module` with synthetic values (buffers such as
    the diffusion schedule are left alone)."""
    with torch.no_grad():
        for name, p in module.named_parameters():
            if any(name.startswith(s) for s in skip_prefixes):
                continue
            p.copy_(synth_param(name, p.shape, seed).to(p.dtype))
    return module


def synth_images(B, H, W, seed=1234):
    """x0 ~ U(-1,1), one hash stream per image."""
    return torch.stack([uniform_pm1("x0", (3, H, W), seed + i) for i in range(B)])


def synth_noise(B, H, W, seed=1235, tag="noise"):
    return torch.stack([normal(tag, (3, H, W), seed + i) for i in range(B)])


def synth_imagenet_images(B, H, W, seed=4321):
    """ImageNet-normalised U(0,1) pixels, as the MCTformer sees them."""
    mean = torch.tensor([0.485, 0.456, 0.406]).reshape(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).reshape(1, 3, 1, 1)
    x = torch.stack([(uniform_pm1("img", (3, H, W), seed + i) + 1) * 0.5 for i in range(B)])
    return (x - mean) / std


def synth_labels(B, C, seed=99):
    """1-3 positive classes per image."""
    lab = torch.zeros(B, C)
    for i in range(B):
        u = uniform01("label", 4, seed + i)
        k = 1 + int(u[0] * 3)
        for j in range(k):
            lab[i, int(u[1 + j] * C) % C] = 1.0
    return lab


def sample_index(name, numel, n=256):
    """The fixed sample of a parameter-sized tensor that the backward fixture keeps (a full ViT-S gradient set is 88 MB):
    n flat indices from a generator seeded by the parameter's name."""
    import zlib
    gen = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    return torch.randint(0, numel, (min(n, numel),), generator=gen)
