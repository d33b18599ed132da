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
        # direct reduce-scatter + all-gather on the same memory, sizes that do / do not divide by world,
        # a bucket shorter than world, and the asynchronous two-bucket form
        for n in (10, 11, 1):
            g3 = torch.arange(n, dtype=torch.float32) * (rank + 1)
            assert GradSync(world, mode="rs_ag")(g3) == scale and torch.equal(g3, torch.arange(n, dtype=torch.float32) * 3)
        g4 = torch.arange(11, dtype=torch.float32) * (rank + 1)
        gs = GradSync(world, mode="rs_ag")
        h = gs.begin(g4[5:])
        assert gs.finish(g4[:5], [h]) == scale and torch.equal(g4, torch.arange(11, dtype=torch.float32) * 3)
        # a rank that left the two-fp16-term mode on its own (inference on rank-local data) must not desynchronise the
        # collectives of the periodic range check: every rank issues the all-reduce, and the others follow the switch
        import warnings
        from starcop_amd.network import HyperStarcopUNet
        torch.manual_seed(0)
        net = HyperStarcopUNet(4, 1)
        net._ensure_flat()
        if rank == 1:
            net.precision, net._range_switched = "fp32-x3", True
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net.check_split_range(sync_ranks=True)
        assert net.precision == "fp32-x3", net.precision
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
