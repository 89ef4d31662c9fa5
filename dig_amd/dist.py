"""Data-parallel sharding of the guidance/CAM path: one process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" in CPU tests).  Images are independent (GroupNorm is per sample, attention is per image,
ViT tokens never mix across the batch), so the only communication is ONE all-gather of the per-rank results."""
import torch
import torch.distributed as dist


def shard_bounds(n_items, world, rank):
    """Contiguous, balanced split of n_items over world ranks -> (lo, hi) of `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_batch(local, n_items=None, group=None):
    """Gather per-rank batches [b_r, ...] into [sum b_r, ...] on every rank with a single collective.
    Equal shards use all_gather_into_tensor directly; ragged shards are padded to the largest shard first."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    local = local.contiguous()
    if n_items is None or n_items % world == 0:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    sizes = [shard_bounds(n_items, world, r) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * mx:r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)


def run_sharded(fn, *batched_inputs, group=None):
    """Split every input along dim 0 over the ranks, run fn on the local shard, all-gather each output.
    fn returns a tensor or a tuple/list of tensors whose dim 0 is the (local) batch."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = batched_inputs[0].shape[0]
    assert n > 0, "run_sharded needs at least one item"
    lo, hi = shard_bounds(n, world, rank)
    outs, single = None, None
    if hi > lo:  # the kernels reject B == 0: a rank whose shard is empty (n < world) skips the compute
        outs = fn(*[x[lo:hi] for x in batched_inputs])
        single = torch.is_tensor(outs)
        outs = [outs] if single else list(outs)
    if n < world:
        # ranks with an empty shard learn the output shapes from rank 0 (which always owns item 0) and contribute a
        # zero-length slice to the gather
        spec = [(single, [(tuple(o.shape[1:]), o.dtype) for o in outs])] if rank == 0 else [None]
        dist.broadcast_object_list(spec, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        single, shapes = spec[0]
        if outs is None:
            dev = batched_inputs[0].device
            outs = [torch.empty((0,) + shp, dtype=dt, device=dev) for shp, dt in shapes]
    res = [all_gather_batch(o, n, group) for o in outs]
    return res[0] if single else res


class PipelinedGather:
    """The all-gather of step i overlapped with the compute of step i+1: three output buffers in rotation, the collective is issued
    asynchronously (RCCL runs it on its own stream, behind the producing kernels of the current stream) and waited for
    one step later - a stream-level wait, the host never blocks.  `drain()` before reading the last result or stopping
    a timer.  With one rank it is the identity."""

    def __init__(self, shape, dtype, device, group=None, force=False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = self.world > 1 or (force and dist.is_initialized())  # force: run the collective even with one rank (rehearsal)
        self.bufs = [torch.empty((self.world * shape[0],) + tuple(shape[1:]), dtype=dtype, device=device) for _ in range(3)] if self.collective else None
        self.pending = None  # (work, output buffer, input kept alive)
        self.count = 0

    def submit(self, local):
        """Queue the gather of `local` ([b, ...], equal b on every rank); returns the PREVIOUS step's gathered tensor (or None).
        The returned tensor is one of the three internal buffers: the next submit() leaves it alone (its collective writes the
        third buffer), the one after that gathers into it - consume it within those two steps or .clone() it.  (With two
        buffers the very next submit's collective would already write into the tensor just handed out.)"""
        if not self.collective:
            prev, self.pending = (self.pending[1] if self.pending else None), (None, local, None)
            return prev
        prev = self.drain()
        out = self.bufs[self.count % 3]
        self.count += 1
        local = local.contiguous()
        work = dist.all_gather_into_tensor(out, local, group=self.group, async_op=True)
        self.pending = (work, out, local)
        return prev

    def drain(self):
        """Make the current stream wait for the outstanding gather; returns its output (or None)."""
        if self.pending is None:
            return None
        work, out, _ = self.pending
        self.pending = None
        if work is not None:
            work.wait()
        return out
