"""MCTformerV2_DiG (DeiT-small with C class tokens) - parameter tree with the reference's state_dict keys
(networks/mctformer.py:171-192, networks/vision_transformer.py:54-103,177-196,226-278); the CAM forward
(return_att=True) runs on the HIP kernels, see dig_amd/cam.py."""
import torch
from torch import nn

__all__ = ["MCTformerV2_DiG", "create_model", "deit_small_MCTformerV2_DiG"]


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container; run MCTformerV2_DiG.forward on a HIP device")


class _Attn(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Mlp(_Holder):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _Block(_Holder):
    def __init__(self, dim, mlp_ratio):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attn(dim)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class _PatchEmbed(_Holder):
    def __init__(self, patch, in_chans, dim):
        super().__init__()
        self.patch_size = (patch, patch)
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch)


class _ResBlockYsh(_Holder):
    """Parameter holder with the reference's names (resnet101.py:31-38,195-207), groups = 1."""

    def __init__(self, cin, cout):
        super().__init__()
        self.convbnrelu1 = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1, bias=False), nn.GroupNorm(1, cout), nn.ReLU())
        self.convbn2 = nn.Sequential(nn.Conv2d(cout, cout, 3, padding=1, bias=False), nn.GroupNorm(1, cout))
        self.branch = nn.Conv2d(cin, cout, 1, bias=False)


class _LFCA(_Holder):
    """Parameter holder for LFCA(D, D_f, C) (mctformer.py:56-93); the forward is dig_amd/cam.py:lfca_forward."""

    def __init__(self, D, Df):
        super().__init__()
        self.diff_init = _ResBlockYsh(Df * 5, Df)
        self.norm_k, self.norm_v = nn.LayerNorm(D, eps=1e-6), nn.LayerNorm(Df, eps=1e-6)
        self.norm_p1, self.norm_c1 = nn.GroupNorm(1, Df, eps=1e-6), nn.LayerNorm(D, eps=1e-6)
        self.query = nn.Linear(D, D)
        self.key = nn.Conv2d(Df, D, 1)
        self.value = nn.Conv2d(Df, Df, 3, padding=1)
        self.mlp_c0 = nn.Linear(Df, D)
        self.mlp_p0 = nn.Conv2d(Df, Df, 3, padding=1)
        self.mlp_c1 = nn.Sequential(nn.Linear(D, D * 4), nn.GELU(), nn.Linear(D * 4, D))
        self.mlp_p1 = nn.Sequential(nn.Conv2d(Df, Df * 4, 1), nn.GELU(), nn.Conv2d(Df * 4, Df, 1))


class MCTformerV2_DiG(nn.Module):
    """Same parameter tree as the reference (mctformer.py:171-192): `lfca=True` (default) creates LFCA0 / LFCA1 /
    diff_head_p under the reference's names, so `load_state_dict(torch.load('{epo:03d}net_trm.pth'))` (model.py:214-218)
    works strictly, and forward(x, feat_diff=[5 x [B,1024,h,w]]) runs the LFCA consumer in eval mode.  `lfca=False` is an
    explicit opt-out that saves the 191 M LFCA parameters for CAM-only inference (then load checkpoints with strict=False).
    forward() is the inference product: in training mode with a drop rate set it raises instead of silently running without
    stochastic depth.  Training goes through dig_amd.train.update_step, which takes the DropPath factors of a step explicitly
    (train.draw_drop_path reads drop_path_rate from here); dropout (drop_rate, 0 in the reference's configuration) is not built."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=20, embed_dim=384, depth=12, num_heads=6,
                 mlp_ratio=4., lfca=True, drop_rate=0., drop_path_rate=0., **_ignored):
        super().__init__()
        self.num_classes, self.embed_dim, self.num_heads, self.depth = num_classes, embed_dim, num_heads, depth
        self.patch_embed = _PatchEmbed(patch_size, in_chans, embed_dim)
        self.num_patches = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, num_classes, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches + num_classes, embed_dim))
        self.blocks = nn.ModuleList([_Block(embed_dim, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)  # present in the checkpoint, unused by the DiG forward
        self.head = nn.Conv2d(embed_dim, num_classes, kernel_size=3, stride=1, padding=1)
        self.D_f, self.D = 512 + 512, 384  # mctformer.py:183-184
        if lfca:
            assert embed_dim == self.D, "LFCA hard-codes D = 384 (mctformer.py:138-146)"
            self.LFCA0, self.LFCA1 = _LFCA(self.D, self.D_f), _LFCA(self.D, self.D_f)
            # the reference hard-codes 20 output maps here whatever num_classes is (mctformer.py:189)
            self.diff_head_p = nn.Sequential(nn.Conv2d(self.D_f, self.D_f, 3, 1, 1), nn.ReLU(inplace=True), nn.Conv2d(self.D_f, 20, 1, 1, 0))
        self.has_lfca = bool(lfca)
        self.drop_rate, self.drop_path_rate = float(drop_rate or 0.), float(drop_path_rate or 0.)
        self.lfca_dtype = torch.float32  # storage type of the LFCA convolutions / wide GEMMs: float32 (parity) or bfloat16 (fast)
        self._packed = None
        self._packed_lfca = None

    def forward(self, x, feat_diff=None, return_att=False, n_layers=12, detach=True):
        from . import cam
        if self.training and (self.drop_path_rate > 0 or self.drop_rate > 0 or feat_diff is not None):
            # the reference trains with DropPath(drop_path_rate) in every block and DropPath(0.1) inside LFCA
            # (vision_transformer.py:184-186, mctformer.py:93,163-164); this build is forward-only
            raise NotImplementedError("dig_amd.MCTformerV2_DiG.forward is the inference path (no DropPath / dropout): call .eval() first; "
                                      "training steps go through dig_amd.train.update_step(..., drop_path=train.draw_drop_path(...))")
        if feat_diff is not None and not self.has_lfca:
            raise RuntimeError("this model was built with lfca=False: no LFCA0/LFCA1/diff_head_p parameters")
        outs = cam.mctformer_forward(self, x, n_layers, feat_diff=feat_diff)
        return outs["rcams"] if return_att else outs


def _set_lfca_dtype(self, dtype):
    assert dtype in (torch.float32, torch.bfloat16)
    self.lfca_dtype = dtype
    return self


MCTformerV2_DiG.set_lfca_dtype = _set_lfca_dtype


def deit_small_MCTformerV2_DiG(pretrained=False, **kwargs):
    if pretrained:
        raise RuntimeError("pretrained weights are downloaded by the reference; unavailable offline")
    kwargs.pop("drop_block_rate", None)
    return MCTformerV2_DiG(patch_size=16, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, **kwargs)


def create_model(name, pretrained=False, **kwargs):
    """Stand-in for timm.create_model at the reference call site (model.py:117-124)."""
    if name != "deit_small_MCTformerV2_DiG":
        raise ValueError(f"unknown model {name}")
    return deit_small_MCTformerV2_DiG(pretrained=pretrained, **kwargs)
