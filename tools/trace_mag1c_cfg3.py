import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from starcop_amd import mag1c
rng = np.random.default_rng(0)
S = 125
base = rng.uniform(1, 6, size=S)
x = torch.from_numpy((base * (1 + 0.05 * rng.standard_normal((512, 512, S)))).astype(np.float32)).cuda()
t = rng.uniform(-1, 0, size=S)
groups = np.arange(1, 513)[None, :].repeat(512, 0)
for _ in range(5):
    mag1c.acrwl1mf_by_groups(x, t, groups)
torch.cuda.synchronize()
