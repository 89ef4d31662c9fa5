"""GPU tier: the allocation-free entry point dig_la_ctx_noalloc (include/dig_hip.h) against the existing entry points it is a
variant of (dig_la_kvctx + dig_la_merge with a caller-allocated workspace), through the C ABI.  Bit-exact (same kernels, same
split count), on two calls in a row, the second with a smaller input - the static scratch keeps the first call's contents and
must not leak them into the second result.  Named test_zz_* so that pytest collects it last (the round-end run uses -x: a failure here must not hide other tests)."""
import pytest
import torch

from dig_amd import _lib
from dig_amd._lib import call, dtype_code

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ctx_with_workspace(x, wkv, B, n, C, splits, dt):
    part = torch.empty(B, splits, 4, 32, 32, dtype=torch.float32, device=DEV)
    pmax = torch.empty(B, splits, 4, 32, dtype=torch.float32, device=DEV)
    psum = torch.empty(B, splits, 4, 32, dtype=torch.float32, device=DEV)
    call("la_kvctx", x=x, wkv=wkv, part=part, pmax=pmax, psum=psum, B=B, n=n, C=C, hid=128, splits=splits, dtype=dtype_code(dt))
    ctx = torch.empty(B * 4, 32, 32, dtype=dt, device=DEV)
    call("la_merge", part=part, pmax=pmax, psum=psum, ctx=ctx, B=B, splits=splits, dtype=dtype_code(dt))
    return ctx


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_la_ctx_noalloc_matches_workspace_entry_points(dt):
    g = torch.Generator().manual_seed(11)
    C = 64
    wkv = (torch.randn(256, C, generator=g) * 0.2).to(DEV).to(dt)
    for B, hw, splits in ((6, 64 * 64, 16), (2, 32 * 48, 5)):  # second call: smaller batch, fewer pixels, fewer splits
        x = torch.randn(B, hw, C, generator=g).to(DEV).to(dt)
        want = _ctx_with_workspace(x, wkv, B, hw, C, splits, dt)
        got = torch.full((B * 4, 32, 32), float("nan"), dtype=dt, device=DEV)
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated()
        call("la_ctx_noalloc", x=x, wkv=wkv, ctx=got, B=B, n=hw, C=C, hid=128, splits=splits, dtype=dtype_code(dt))
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before
        assert torch.isfinite(got.float()).all()
        assert torch.equal(got, want), float((got.float() - want.float()).abs().max())


def test_la_ctx_noalloc_rejects_oversize():
    L = _lib.lib()
    mx = L.dig_la_ctx_noalloc_max_slots()
    assert mx == 1024
    x = torch.zeros(1, 64, 64, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(256, 64, dtype=torch.bfloat16, device=DEV)
    ctx = torch.zeros(4, 32, 32, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="exceeds the static scratch"):
        call("la_ctx_noalloc", x=x, wkv=w, ctx=ctx, B=mx // 8 + 1, n=64, C=64, hid=128, splits=8, dtype=0)


def test_la_ctx_noalloc_rejects_unknown_dtype():
    x = torch.zeros(1, 64, 64, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(256, 64, dtype=torch.bfloat16, device=DEV)
    ctx = torch.zeros(4, 32, 32, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="unknown dtype"):
        call("la_ctx_noalloc", x=x, wkv=w, ctx=ctx, B=1, n=64, C=64, hid=128, splits=1, dtype=7)


@pytest.mark.parametrize("dt,C", [(torch.bfloat16, 64), (torch.float32, 256)])
def test_la_ctx_noalloc_under_graph_capture(dt, C):
    """The caller class the entry point exists for: after ONE warm-up call per device (which resolves the scratch symbol and,
    for the fp32 C = 256 instantiation, raises the kernel's LDS limit), the call is captured in a HIP graph - where hipMalloc
    is illegal - and every replay reproduces the eager result bit for bit, also after the input has changed."""
    g = torch.Generator().manual_seed(5)
    B, hw, splits = 3, 48 * 32, 6
    wkv = (torch.randn(256, C, generator=g) * 0.2).to(DEV).to(dt)
    x = torch.randn(B, hw, C, generator=g).to(DEV).to(dt)
    got = torch.zeros(B * 4, 32, 32, dtype=dt, device=DEV)
    kw = dict(x=x, wkv=wkv, ctx=got, B=B, n=hw, C=C, hid=128, splits=splits, dtype=dtype_code(dt))
    call("la_ctx_noalloc", **kw)  # warm-up outside capture
    want1 = _ctx_with_workspace(x, wkv, B, hw, C, splits, dt)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            call("la_ctx_noalloc", stream=s.cuda_stream, **kw)
    torch.cuda.current_stream().wait_stream(s)
    got.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want1)
    x.copy_(torch.randn(B, hw, C, generator=g).to(DEV).to(dt))  # same buffers, new contents
    want2 = _ctx_with_workspace(x, wkv, B, hw, C, splits, dt)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(got, want2) and not torch.equal(want1, want2)
