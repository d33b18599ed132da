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


def _worker8(rank, world, port, q):
    """the data-parallel exchange at the node's real width with the REAL bucket sizes: the flat gradient of HyperStarcopUNet is
    [encoder 2 224 160 | decoder + head 4 405 073] floats = 6 629 233; the tail bucket (and the whole buffer) leaves a remainder of 1
    at world 8, which rs_ag routes through its third collective (parallel.py `rest`)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from starcop_amd.parallel import GradSync, shard_range, sharded_map
        n_enc, n_tail = 2224160, 4405073
        n = n_enc + n_tail
        assert n % world == 1 and n_tail % world == 1 and n_enc % world == 0
        # integer-valued floats below 2^24: every summation order gives the same bits, so the expected sum is exact
        base = (torch.arange(n, dtype=torch.int64) % 1000).float()
        want = base * float(sum(range(1, world + 1)))
        out = {}
        for mode in ("allreduce", "rs_ag"):
            g = base * (rank + 1)
            gs = GradSync(world, mode=mode)
            h = gs.begin(g[n_enc:])                      # decoder + head bucket, asynchronously (where the backend orders it)
            scale = gs.finish(g[:n_enc], [h])            # encoder bucket, then the join
            assert scale == 1.0 / world
            assert torch.equal(g, want), (mode, float((g - want).abs().max()))
            g1 = base * (rank + 1)                       # the one-message form (a grad_sync object without begin())
            assert GradSync(world, mode=mode)(g1) == scale and torch.equal(g1, want)
            out[mode] = float(g.double().sum())
        # partition of independent work at world 8: EMIT's 621 column blocks, configs[4]'s 9 row strips (fewer items than 2 x ranks)
        lo, hi = shard_range(621, rank, world)
        tiles = torch.arange(9 * 2, dtype=torch.float32).reshape(9, 2)
        got = sharded_map(lambda t: t + 0.5, tiles)
        assert torch.equal(got, tiles + 0.5)
        few = sharded_map(lambda t: t * 3, torch.arange(5, dtype=torch.float32).reshape(5, 1))      # ranks 5-7 hold no item
        assert torch.equal(few, torch.arange(5, dtype=torch.float32).reshape(5, 1) * 3)
        q.put((rank, out, (lo, hi)))
    finally:
        dist.destroy_process_group()


def test_world8_gloo_real_bucket_sizes():
    """first-contact insurance for the 8-GPU run: world 8 on gloo, real bucket sizes, both exchange modes, the
    two-bucket asynchronous form, the remainder collective of rs_ag"""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [r[0] for r in res] == list(range(world))
    assert all(r[1] == res[0][1] for r in res) and res[0][1]["allreduce"] == res[0][1]["rs_ag"]
    spans = [r[2] for r in res]
    assert spans[0][0] == 0 and spans[-1][1] == 621 and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert sorted(b - a for a, b in spans) == [77] * 3 + [78] * 5
