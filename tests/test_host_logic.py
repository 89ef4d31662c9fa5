"""CPU tier: host-side logic - the C-ABI library loads and exports every symbol include/dig_hip.h declares (no compute
calls without a GPU), argument structs mirror the header, product modules keep the reference's state_dict keys and
refuse to run off-device (no CPU fallback), synthetic generators are deterministic, sharding + all-gather (gloo, 2 ranks)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import pytest
import torch

from dig_amd import _lib, synth
from dig_amd.diffusion import GaussianDiffusion, Unet
from dig_amd.mctformer import create_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    with open(os.path.join(ROOT, "include", "dig_hip.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    declared = re.findall(r"\b(dig_[a-z0-9_]+)\s*\(", text)
    assert len(declared) >= 27 and set(declared) == set(_lib.FUNCTIONS)
    L = _lib.lib()
    for fn in declared:
        assert hasattr(L, fn), fn
    assert L.dig_version() >= 100
    assert L.dig_conv3x3_tiles(256, 256, 512, 0, 512, 8, 0) == 256 and L.dig_conv3x3_tiles(28, 28, 256, 256, 512, 8, 1) == 4  # bf16 wide: one partial per 16x16 pixels from either tile height
    assert L.dig_conv3x3_tiles(256, 256, 64, 0, 64, 8, 0) == 1024  # weight-stationary 64->64 variant: per-wave partials


def test_struct_mirror_sizes():
    # every pointer is 8 bytes, long long 8, int/float 4: ctypes must agree with the C compiler's layout
    S = _lib.STRUCTS["dig_gemm_args"]
    assert ctypes.sizeof(S) == 8 * 8 + 10 * 8 + 22 * 4 + 4 + 4  # 8 ptrs, 10 strides, 22 ints, alpha, tail pad
    assert ctypes.sizeof(_lib.STRUCTS["dig_conv3x3_args"]) == 7 * 8 + 8 + 9 * 4 + 4
    assert [f for f, _ in _lib.STRUCTS["dig_stem_args"]._fields_][-7:] == ["pre_scale", "pre_shift", "B", "H", "W", "Cout", "dtype"]


def test_invalid_arguments_are_rejected_without_a_gpu():
    L = _lib.lib()
    a = _lib.STRUCTS["dig_gemm_args"]()
    assert L.dig_gemm_nt(ctypes.byref(a), None) == -1 and b"null pointer" in L.dig_last_error()
    c = _lib.STRUCTS["dig_conv3x3_args"]()
    assert L.dig_conv3x3(ctypes.byref(c), None) == -1
    for fn in ("gemm_f32", "layernorm_bwd", "colsum", "gelu_bwd", "softmax_bwd", "msm_grad", "head_pool_bwd", "adamw", "head_sum", "maxnorm_l1_bwd",
               "cam_fuse_bwd"):
        st = _lib.STRUCTS[f"dig_{fn}_args"]()
        assert getattr(L, f"dig_{fn}")(ctypes.byref(st), None) == -1, fn
    assert L.dig_layernorm_bwd_blocks(ctypes.c_longlong(33)) == 5 and L.dig_colsum_chunks(ctypes.c_longlong(129)) == 2


def test_adamw_rule_and_decay_filter():
    """The optimiser restatements against torch.optim.AdamW itself (CPU): oracle formula = torch's update; the product's
    no-decay filter = the oracle's (timm's add_weight_decay rule: 1-D, .bias, pos_embed, cls_token)."""
    import dig_oracle as O
    from dig_amd import train
    torch.manual_seed(3)
    p0, g = torch.randn(7, 5), torch.randn(7, 5)
    q = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([q], lr=1e-3, weight_decay=0.05)
    m, v, p = torch.zeros_like(p0), torch.zeros_like(p0), p0.clone()
    for t in (1, 2, 3):
        q.grad = g * t
        opt.step()
        p, m, v = O.adamw_step(p, g * t, m, v, t, 1e-3, 0.05)
        assert float((p - q.detach()).abs().max()) <= 1e-6
    net = create_model("deit_small_MCTformerV2_DiG", pretrained=False, num_classes=20, lfca=False)
    for k, prm in net.named_parameters():
        assert train.no_weight_decay(k, prm) == O.adamw_no_decay(k, prm), k
    nd = {k for k, prm in net.named_parameters() if train.no_weight_decay(k, prm)}
    assert {"pos_embed", "cls_token", "head.bias", "blocks.0.norm1.weight", "blocks.3.attn.qkv.bias"} <= nd
    assert "blocks.0.attn.qkv.weight" not in nd and "head.weight" not in nd and "patch_embed.proj.weight" not in nd


def test_draw_drop_path_follows_the_reference_rates():
    """train.draw_drop_path: block i draws at linspace(0, drop_path_rate, depth)[i] (vision_transformer.py:265; block 0 has
    rate 0 = nn.Identity = no factors), two draws per block (:193-194), two per LFCA at 0.1 in the first pass only
    (mctformer.py:93,163-164); every factor is 0 or 1 / keep_prob (timm's scale_by_keep)."""
    from dig_amd import train
    m = create_model("deit_small_MCTformerV2_DiG", pretrained=False, num_classes=20, drop_path_rate=0.1, lfca=True)
    g = torch.Generator().manual_seed(3)
    dp = train.draw_drop_path(m, 64, "cpu", generator=g)
    assert set(dp) == {"pass1", "pass2"} and set(dp["pass1"]) == {"blocks", "LFCA0", "LFCA1"} and set(dp["pass2"]) == {"blocks"}
    rates = [float(r) for r in torch.linspace(0, 0.1, 12)]
    for ps in ("pass1", "pass2"):
        assert len(dp[ps]["blocks"]) == 12 and dp[ps]["blocks"][0] == (None, None)
        for r, (a, b) in list(zip(rates, dp[ps]["blocks"]))[1:]:
            for s_ in (a, b):
                assert s_.shape == (64,) and set(s_.tolist()) <= {0.0, float(torch.tensor(1.0) / (1.0 - r))}
    drops = sum(int((s_ == 0).sum()) for a, b in dp["pass1"]["blocks"][1:] for s_ in (a, b))
    assert 20 <= drops <= 140  # expectation 64 * 2 * sum(rates) = 70
    for k in ("LFCA0", "LFCA1"):
        assert all(set(s_.tolist()) <= {0.0, float(torch.tensor(1.0) / 0.9)} for s_ in dp["pass1"][k])
    m0 = create_model("deit_small_MCTformerV2_DiG", pretrained=False, num_classes=20, drop_path_rate=0.0, lfca=False)
    assert train.draw_drop_path(m0, 4, "cpu", lfca=False) is None


def test_state_dict_keys_and_no_cpu_fallback():
    u = Unet(dim=64, dim_mults=(1, 2, 4, 8), flash_attn=True)
    gd = GaussianDiffusion(u, image_size=224, timesteps=1000, sampling_timesteps=150).eval()
    sd = gd.state_dict()
    assert len(sd) == 287 and sum(p.numel() for p in u.parameters()) == 35710339
    for k in ("model.downs.0.2.norm.g", "model.downs.0.2.to_qkv.weight", "model.downs.0.2.to_out.0.weight", "model.downs.0.2.to_out.1.g",
              "model.downs.0.3.1.weight", "model.downs.3.3.weight", "model.mid_attn.to_out.bias", "model.ups.0.0.mlp.1.weight",
              "model.ups.2.3.1.bias", "model.final_res_block.res_conv.weight", "model.time_mlp.3.bias", "sqrt_alphas_cumprod", "loss_weight"):
        assert k in sd, k
    assert abs(float(gd.sqrt_alphas_cumprod[100]) - 0.97979611) < 1e-7
    with pytest.raises(RuntimeError, match="HIP device only"):
        gd.p_losses(torch.zeros(1, 3, 32, 32), torch.zeros(1, dtype=torch.long), noise=torch.zeros(1, 3, 32, 32))
    with pytest.raises(NotImplementedError):
        Unet(dim=64, self_condition=True)
    # the reference's own {key: shape} table (tests/golden/state_dict_keys.json, written by oracle/gen_golden.py from the
    # imported reference modules): set-equality, so model.py:140-142's strict load_state_dict works unchanged
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as f:
        tables = json.load(f)
    ref = tables["GaussianDiffusion(Unet(dim=64,dim_mults=(1,2,4,8)))"]
    assert {k: list(v.shape) for k, v in sd.items()} == ref


@pytest.mark.parametrize("C", [20, 80])
def test_mctformer_state_dict_matches_reference_call_site(C):
    """model.py:117-124 (create_model with the reference's keywords, nothing else) then model.py:214-218 (strict load):
    the default build carries LFCA0 / LFCA1 / diff_head_p, key- and shape-identical to the reference's table."""
    with open(os.path.join(ROOT, "tests", "golden", "state_dict_keys.json")) as f:
        ref = json.load(f)[f"deit_small_MCTformerV2_DiG(num_classes={C})"]
    m = create_model("deit_small_MCTformerV2_DiG", pretrained=False, num_classes=C, drop_rate=0.0, drop_path_rate=0.1, drop_block_rate=None)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == ref
    m.load_state_dict({k: torch.zeros(v) for k, v in ref.items()}, strict=True)
    with pytest.raises(RuntimeError, match="HIP device only"):
        m.eval()(torch.zeros(1, 3, 224, 224), return_att=True)
    with pytest.raises(NotImplementedError, match="inference path"):
        m.train()(torch.zeros(1, 3, 224, 224))  # the reference's update() runs DropPath 0.1 here: refuse instead of diverging silently
    small = create_model("deit_small_MCTformerV2_DiG", pretrained=False, num_classes=C, lfca=False)  # explicit CAM-only opt-out
    assert set(small.state_dict()) == {k for k in ref if not k.startswith(("LFCA", "diff_head_p"))}


class _FakeDeviceTensor(torch.Tensor):
    """A CPU tensor that claims to live on a HIP device: lets the CPU tier drive the product's guards up to (not into) a launch."""

    @staticmethod
    def __new__(cls, t):
        return torch.Tensor._make_subclass(cls, t)

    @property
    def is_cuda(self):
        return True

    @property
    def device(self):
        return torch.device("cuda", 0)


def test_cpu_resident_module_and_bad_timesteps_raise_before_any_launch(monkeypatch):
    """the guards must look at the parameters/buffers too, not only at the input, and t must be range-checked on
    the host (the kernels index the schedule tables with it).  No GPU here: any attempt to launch would hit _lib.call."""
    from dig_amd import _lib as L, diffusion as D

    def no_launch(fn, **kw):
        raise AssertionError(f"dig_{fn} would have been launched")

    monkeypatch.setattr(D, "call", no_launch)
    u = Unet(dim=16, dim_mults=(1, 2, 4, 8), flash_attn=True)
    gd = GaussianDiffusion(u, image_size=32, timesteps=1000, sampling_timesteps=150).eval()
    x = _FakeDeviceTensor(torch.zeros(2, 3, 32, 32))
    t = torch.zeros(2, dtype=torch.long)
    with pytest.raises(RuntimeError, match=r"lives on cpu but the input on cuda:0"):
        gd.p_losses(x, t, noise=x)
    with pytest.raises(RuntimeError, match=r"lives on cpu but the input on cuda:0"):
        u(x, t)
    with pytest.raises(RuntimeError, match="schedule buffers on cpu"):
        gd.q_sample(x, t, noise=x)
    with pytest.raises(IndexError, match="timestep out of range"):
        gd.p_losses(x, torch.tensor([0, 1000]), noise=x)
    with pytest.raises(IndexError, match="timestep out of range"):
        gd.p_losses(x, torch.tensor([-1, 5]), noise=x)
    m = create_model("deit_small_MCTformerV2_DiG", pretrained=False, num_classes=20, lfca=False).eval()
    with pytest.raises(RuntimeError, match=r"lives on cpu but the input on cuda:0"):
        m(_FakeDeviceTensor(torch.zeros(1, 3, 224, 224)), return_att=True)
    # the ctypes layer itself refuses host tensors (no pointer of a CPU tensor ever reaches a kernel)
    with pytest.raises(RuntimeError, match="HIP-device pointers only"):
        L.call("rmsnorm", x=torch.zeros(4, 8), g=torch.zeros(8), res=None, out=torch.zeros(4, 8), rows=4, C=8, dtype=1)
    a = L.STRUCTS["dig_sched_axpby_args"]()
    for f in ("x", "y", "t", "sqrt_ac", "sqrt_1mac", "out"):
        setattr(a, f, 16)
    assert L.lib().dig_sched_axpby(ctypes.byref(a), None) == -1 and b"bad shape" in L.lib().dig_last_error()


def test_cam_dict_formats(tmp_path):
    import numpy as np
    from dig_amd import evaluation as E
    d = {3: np.arange(12, dtype=np.float32).reshape(3, 4), 17: np.ones((3, 4), np.float32)}
    E.save_cam_dict(str(tmp_path / "a.npz"), d)
    back = E.load_cam_dict(str(tmp_path / "a.npz"))
    assert sorted(back) == [3, 17] and np.array_equal(back[3], d[3])
    E.save_cam_dict(str(tmp_path / "b.npy"), d, reference_format=True)  # model.py:489-492's format: a pickled dict
    with pytest.raises(ValueError, match="reference_format"):
        E.load_cam_dict(str(tmp_path / "b.npy"))
    assert np.array_equal(E.load_cam_dict(str(tmp_path / "b.npy"), reference_format=True)[17], d[17])
    # the reference call site's file name (name + '.npy', model.py:489-492) lands on <name>.npz - where do_python_eval looks
    # first - not on <name>.npy.npz; a missing file is reported as missing, not as a format problem
    assert E.save_cam_dict(str(tmp_path / "c.npy"), d) == str(tmp_path / "c.npz")
    assert (tmp_path / "c.npz").exists() and not (tmp_path / "c.npy.npz").exists() and not (tmp_path / "c.npy").exists()
    assert np.array_equal(E.load_cam_dict(str(tmp_path / "c.npz"))[3], d[3])
    with pytest.raises(FileNotFoundError):
        E.load_cam_dict(str(tmp_path / "missing.npy"))


def test_synth_is_deterministic_and_keyed():
    a, b = synth.synth_param("downs.0.0.block1.proj.weight", (64, 64, 3, 3)), synth.synth_param("downs.0.0.block1.proj.weight", (64, 64, 3, 3))
    assert torch.equal(a, b) and abs(float(a.var()) - 1 / 576) < 2e-4
    assert not torch.equal(a, synth.synth_param("downs.0.1.block1.proj.weight", (64, 64, 3, 3)))
    assert abs(float(synth.synth_param("x.norm.weight", (64,)).mean()) - 1.0) < 0.05
    n = synth.synth_noise(2, 64, 64)
    assert abs(float(n.mean())) < 0.02 and abs(float(n.std()) - 1.0) < 0.02
    assert float(a.flatten()[0]) == pytest.approx(float(synth.uniform_pm1("downs.0.0.block1.proj.weight", (1,))) * (3 / 576) ** 0.5)


_WORKER = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from dig_amd import dist as dd
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = int(sys.argv[2])
x = torch.arange(n * 6, dtype=torch.float32).reshape(n, 2, 3)
def fn(xs):  # stand-in for the per-image guidance forward: any per-sample function
    return xs * 2 + 1, xs.sum((1, 2))
a, b = dd.run_sharded(fn, x)
ra, rb = fn(x)
assert torch.equal(a, ra) and torch.equal(b, rb), (rank, a.shape)
lo, hi = dd.shard_bounds(n, world, rank)
assert hi - lo in (n // world, n // world + 1)
# pipelined gather (bench.py's N > 1 path): step i's collective is waited for at step i+1
pg = dd.PipelinedGather((3, 4), torch.float32, "cpu")
got = []
for i in range(5):
    prev = pg.submit(torch.full((3, 4), float(10 * i + rank)))
    if prev is not None:
        got.append(prev.clone())
got.append(pg.drain().clone())
assert pg.drain() is None and len(got) == 5
for i, g in enumerate(got):
    want = torch.cat([torch.full((3, 4), float(10 * i + r)) for r in range(world)])
    assert torch.equal(g, want), (rank, i)
# the buffer contract of submit(): the tensor it returns is one of three internal buffers - the NEXT submit leaves it alone,
# the one after that gathers a new step into the same storage; a consumer that needs it longer clones it
pg = dd.PipelinedGather((2,), torch.float32, "cpu")
step = lambda i: torch.full((2,), float(100 * i + rank))
whole = lambda i: torch.cat([torch.full((2,), float(100 * i + r)) for r in range(world)])
assert pg.submit(step(0)) is None
held = pg.submit(step(1))                  # result of step 0
assert torch.equal(held, whole(0))
nxt = pg.submit(step(2))                   # result of step 1; its own collective (step 2) goes to the third buffer
pg.pending[0].wait()                       # let the step-2 gather finish: it must not have touched `held`
assert torch.equal(held, whole(0)) and torch.equal(nxt, whole(1)) and nxt.data_ptr() != held.data_ptr()
pg.submit(step(3))                         # this one gathers step 3 into the buffer `held` points at
assert torch.equal(pg.drain(), whole(3))
assert torch.equal(held, whole(3))         # overwritten two submits later, as documented
dist.destroy_process_group()
print("ok", rank)
"""


@pytest.mark.parametrize("n", [8, 7, 1])
def test_sharded_all_gather_gloo_world2(tmp_path, n):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    procs = []
    port = 29500 + (os.getpid() % 500) + n
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(n)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in procs:
        out, _ = p.communicate(timeout=120)
        assert p.returncode == 0, out.decode()


def test_c_abi_from_plain_c(tmp_path):
    """include/dig_hip.h is consumable from C99 and libdig_hip.so links and runs from a C program (no GPU needed for the
    query / argument-rejection entry points)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "abi")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi_smoke.c"), "-o", exe, "-L", libdir, "-ldig_hip", f"-Wl,-rpath,{libdir}"], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert out.startswith("128 16 ") and "null pointer" in out
