"""Data-parallel training across the GPUs of one node: one process per GPU, gradient all-reduce over RCCL/xGMI.

The reference has no collective of its own; with ``training.devices > 1`` Lightning wraps the module in DDP
(SURVEY.md section 5): per-step gradient AVERAGE over ranks, BatchNorm statistics stay rank-local (no SyncBN),
confusion matrices are summed at epoch end.  Here the whole gradient lives in one flat fp32 buffer
(6 629 233 floats = 26.5 MB) laid out [encoder | decoder | head], so the exchange is in-place ``all_reduce(SUM)`` on
slices of it -- no bucket copies, no extra pass -- and the 1/world average is folded into the fused Adam kernel
(``grad_scale``).  Two buckets: the decoder + head slice (17.7 MB, complete when the backward walk leaves the decoder)
is reduced asynchronously on RCCL's stream while the encoder's backward runs; the encoder slice (8.8 MB) follows when
the walk ends.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): few large messages let RCCL use all links at
once, which is why there are two buckets and not DDP's 25 MB / per-layer granularity.
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, world_size=None, group=None):
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)

    def _host_staged(self, t):
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def __call__(self, flat_grads: torch.Tensor) -> float:
        """Sum ``flat_grads`` over ranks in place; returns the scale (1/world) the optimiser must apply."""
        if self.world > 1 and flat_grads.numel():
            if self._host_staged(flat_grads):
                # functional path only (gloo has no device transport on ROCm): stage through the host
                host = flat_grads.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                flat_grads.copy_(host)
            else:
                dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=self.group)      # RCCL, in place, compute stream
        return 1.0 / self.world

    def begin(self, bucket: torch.Tensor):
        """Start summing ``bucket`` (a slice of the flat gradient buffer whose gradients are all queued on the current
        stream) over ranks, asynchronously; returns a handle for :meth:`finish` (None if nothing is in flight)."""
        if self.world <= 1 or bucket.numel() == 0:
            return None
        if self._host_staged(bucket):
            self(bucket)
            return None
        return dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self, rest: torch.Tensor, handles) -> float:
        """Sum ``rest`` (the remainder of the buffer), then make the current stream wait for the buckets in flight."""
        scale = self(rest)
        for h in handles:
            if h is not None:
                h.wait()
        return scale


def broadcast_parameters(network, src=0, group=None):
    """Identical initial weights and BatchNorm buffers on every rank (what DDP does at construction)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    dist.broadcast(network.flat_parameters(), src=src, group=group)
    for b in network.buffers():
        dist.broadcast(b, src=src, group=group)
    network.mark_parameters_changed()


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of ``n_items`` independent work items (tiles, mag1c column groups) for ``rank``."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sharded_map(fn, items: torch.Tensor, group=None, gather=True):
    """Independent work items (inference tiles, scenes, mag1c column blocks) partitioned over the ranks -- the
    "tile-sharded" multi-GPU mode of BASELINE configs[4].  ``items`` is a tensor whose first dimension indexes the work
    items (identical on every rank); ``fn(items[lo:hi])`` must return a tensor whose first dimension is ``hi - lo``.
    No collective on the data path; with ``gather`` the per-rank results are concatenated on every rank by one
    ``all_gather`` of equally padded slabs (RCCL on GPU tensors; gloo results are staged through the host)."""
    n = int(items.shape[0])
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(n, rank, world)
    out = fn(items[lo:hi])
    if world == 1 or not gather:
        return out
    slab = -(-n // world)
    pad = torch.zeros((slab,) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device)
    pad[:hi - lo] = out
    via_host = pad.is_cuda and dist.get_backend(group) == "gloo"
    send = pad.cpu() if via_host else pad
    recv = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(recv, send, group=group)
    parts = []
    for r in range(world):
        a, b = shard_range(n, r, world)
        parts.append(recv[r][:b - a])
    res = torch.cat(parts, 0)
    return res.to(out.device) if via_host else res
