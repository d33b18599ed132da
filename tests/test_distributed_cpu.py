"""N>1 path on CPU: world_size-2 gloo processes exercise the data-parallel glue (gradient sum + 1/world scale,
parameter broadcast, confusion-matrix sum, partition of independent work)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from starcop_amd import metrics as M
        from starcop_amd.parallel import GradSync, shard_range
        # per-rank gradients g_r ; DDP semantics = mean over ranks = (sum) * (1/world) applied by the optimiser
        g = torch.arange(10, dtype=torch.float32) * (rank + 1)
        scale = GradSync(world)(g)
        avg = g * scale
        # two-bucket form used by fused_train_step: tail bucket asynchronously, the rest at the end
        g2 = torch.arange(10, dtype=torch.float32) * (rank + 1)
        gs = GradSync(world)
        h = gs.begin(g2[6:])
        assert gs.finish(g2[:6], [h]) == scale and torch.equal(g2, g)
        cm = M.BinaryConfusionMatrix()
        cm.update(torch.tensor([1, 0, 1, rank]), torch.tensor([1, 0, 0, 1]))
        cm.sync()
        lo, hi = shard_range(621, rank, world)
        # tile-sharded map: 7 "tiles" over 2 ranks (4 + 3), results gathered in order on every rank
        from starcop_amd.parallel import sharded_map
        tiles = torch.arange(7 * 3, dtype=torch.float32).reshape(7, 3)
        calls = []
        got = sharded_map(lambda t: (calls.append(int(t.shape[0])) or t * 2 + 1), tiles)
        assert torch.equal(got, tiles * 2 + 1) and calls == [4 - rank]
        q.put((rank, avg.tolist(), scale, cm.compute().tolist(), (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [(1.5 * i) for i in range(10)]                 # mean of g and 2g
    for rank, avg, scale, cm, span in res:
        assert avg == pytest.approx(want) and scale == 0.5
        assert sum(sum(r) for r in cm) == 8                # both ranks' 4 samples
    assert res[0][3] == res[1][3]
    assert res[0][4] == (0, 311) and res[1][4] == (311, 621)
