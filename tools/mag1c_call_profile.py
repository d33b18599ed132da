"""cProfile of acrwl1mf_by_groups at the cfg3 shape (512 x 512 x 125 fp32, 512 column groups): where the host time of one call goes.
python tools/mag1c_call_profile.py [calls]"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from starcop_amd import mag1c
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
rng = np.random.default_rng(0)
S = 125
x = torch.from_numpy((rng.uniform(1, 6, size=S) * (1 + 0.05 * rng.standard_normal((512, 512, S)))).astype(np.float32)).cuda()
t = np.linspace(-1, -0.1, S)
groups = np.arange(1, 513)[None, :].repeat(512, 0)
for _ in range(3):
    mag1c.acrwl1mf_by_groups(x, t, groups)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    mag1c.acrwl1mf_by_groups(x, t, groups)
torch.cuda.synchronize()
print(f"wall {1e3 * (time.perf_counter() - t0) / n:.3f} ms / call")
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    mag1c.acrwl1mf_by_groups(x, t, groups)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
