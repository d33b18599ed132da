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
    """Sum of the flat gradient buffer over the ranks, in place.

    ``mode``:
      * ``"allreduce"`` (default): one ``all_reduce(SUM)`` per bucket (RCCL picks ring / tree);
      * ``"rs_ag"``: ``reduce_scatter_tensor`` + ``all_gather_into_tensor`` on the same memory (rank r reduces the r-th
        1/world chunk of the bucket, then the chunks are gathered back; both in place: the chunk IS a slice of the bucket).
        On 8 fully connected xGMI peers this is the direct algorithm of SURVEY.md section 5 -- every GPU exchanges 1/8 of the
        buffer with each peer on all 7 links at once instead of pushing the whole buffer round a ring.  The < world trailing
        elements that do not fill a chunk go through a (tiny) all-reduce.  ``STARCOP_GRADSYNC=rs_ag`` selects it globally.
    ``timing=True`` records, per step, the host-blocking time of each phase with device events (``last_timing``): how long the
    asynchronous tail bucket had been running when the backward walk ended, and how long the step then still waited for it and
    for the second bucket -- the EXPOSED (non-overlapped) communication time ``bench.py --gpus N`` prints per rank."""

    def __init__(self, world_size=None, group=None, mode=None, timing=False):
        import os
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.mode = mode or os.environ.get("STARCOP_GRADSYNC", "allreduce")
        if self.mode not in ("allreduce", "rs_ag"):
            raise ValueError(f"GradSync: mode must be 'allreduce' or 'rs_ag' (got {self.mode!r})")
        self.timing = timing
        self.last_timing = None
        self._ev = []

    def _host_staged(self, t):
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def _reduce(self, t, async_op=False):
        """in-place SUM of ``t`` over the ranks; returns the list of work handles when ``async_op``"""
        handles = []
        if self.mode == "rs_ag" and t.numel() >= self.world:
            rank = dist.get_rank(self.group)
            chunk = t.numel() // self.world
            main = t[:chunk * self.world]
            mine = main[rank * chunk:(rank + 1) * chunk]
            h1 = dist.reduce_scatter_tensor(mine, main, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
            h2 = dist.all_gather_into_tensor(main, mine, group=self.group, async_op=async_op)     # same stream: ordered after h1
            handles += [h1, h2]
            rest = t[chunk * self.world:]
            if rest.numel():
                handles.append(dist.all_reduce(rest, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op))
        else:
            handles.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op))
        return handles if async_op else None

    def __call__(self, flat_grads: torch.Tensor) -> float:
        """Sum ``flat_grads`` over ranks in place; returns the scale (1/world) the optimiser must apply."""
        if self.world > 1 and flat_grads.numel():
            if self._host_staged(flat_grads):
                # functional path only (gloo has no device transport on ROCm): stage through the host
                host = flat_grads.cpu()
                self._reduce(host)
                flat_grads.copy_(host)
            else:
                self._reduce(flat_grads)      # RCCL, in place, compute stream
        return 1.0 / self.world

    def _mark(self, t):
        if self.timing and t.is_cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._ev.append(e)

    def begin(self, bucket: torch.Tensor):
        """Start summing ``bucket`` (a slice of the flat gradient buffer whose gradients are all queued on the current
        stream) over ranks, asynchronously; returns a handle for :meth:`finish` (None if nothing is in flight)."""
        self._ev = []
        if self.world <= 1 or bucket.numel() == 0:
            return None
        self._mark(bucket)                      # event 0: tail bucket handed to the collective
        if self._host_staged(bucket):
            self(bucket)
            return None
        if self.mode == "rs_ag" and dist.get_backend(self.group) != "nccl":
            # only RCCL orders a process group's collectives on one stream; gloo's asynchronous ops run on independent threads,
            # so the all-gather could read the chunk before the reduce-scatter has written it
            self._reduce(bucket)
            return None
        return self._reduce(bucket, async_op=True)

    def finish(self, rest: torch.Tensor, handles) -> float:
        """Sum ``rest`` (the remainder of the buffer), then make the current stream wait for the buckets in flight."""
        self._mark(rest)                        # event 1: the backward walk has ended on this stream
        scale = self(rest)
        self._mark(rest)                        # event 2: second bucket reduced
        for h in handles:
            for w in (h if isinstance(h, (list, tuple)) else [h]):
                if w is not None:
                    w.wait()
        self._mark(rest)                        # event 3: first (asynchronous) bucket joined
        return scale

    def collect_timing(self):
        """ms of the last step: {'overlap_window': tail bucket start -> end of backward, 'bucket_rest': exposed time of the
        second bucket, 'bucket_tail_wait': exposed remainder of the asynchronous bucket}.  Synchronises the device."""
        if not self.timing or len(self._ev) < 4:
            return None
        torch.cuda.synchronize()
        e = self._ev
        self.last_timing = dict(overlap_window=e[0].elapsed_time(e[1]), bucket_rest=e[1].elapsed_time(e[2]),
                                bucket_tail_wait=e[2].elapsed_time(e[3]))
        return self.last_timing


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
