"""The N>1 path of bench.py end to end on ONE GPU: two ranks share the device, gloo backend (GradSync stages the flat
gradient through the host).  Checks that both ranks walk the same sequence of collectives (a rank-0-only collective would
dead-lock or kill the job) and that rank 0 prints one well-formed JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_gpu(hip):
    env = dict(os.environ, STARCOP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2", "--tile", "256"]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
             for r in (1, 0)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two-rank bench timed out (collective mismatch between ranks?)")
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    line = [l for l in outs[1][0].splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 4 and rec["value"] > 0
    assert rec["roofline"]["frac"] > 0 and "cpu_baseline" not in rec
    assert not [l for l in outs[0][0].splitlines() if l.startswith("{")]          # rank 1 prints nothing
