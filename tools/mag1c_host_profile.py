"""Where the non-kernel time of acrwl1mf_by_groups goes (cfg3 shape): rocprof-free timing of the host-side torch ops."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from starcop_amd import mag1c
dev = "cuda"
rng = np.random.default_rng(0)
S = 125
x = torch.from_numpy((rng.uniform(1, 6, size=S) * (1 + 0.05 * rng.standard_normal((512, 512, S)))).astype(np.float32)).to(dev)
t = np.linspace(-1, -0.1, S)
groups = np.arange(1, 513)[None, :].repeat(512, 0)
gt = torch.as_tensor(groups).to(dev)
def T(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("total numpy groups :", T(lambda: mag1c.acrwl1mf_by_groups(x, t, groups)))
print("total device groups:", T(lambda: mag1c.acrwl1mf_by_groups(x, t, gt)))
g = gt.reshape(-1).long()
print("groups upload+long :", T(lambda: torch.as_tensor(np.asarray(groups)).to(dev).reshape(-1).long()))
m = torch.ones(512 * 512, dtype=torch.bool, device=dev)
print("nonzero            :", T(lambda: torch.nonzero(m).reshape(-1)))
vi = torch.nonzero(m).reshape(-1)
print("gather groups      :", T(lambda: g[vi]))
gv = g[vi]
print("argsort stable     :", T(lambda: torch.argsort(gv, stable=True)))
order = torch.argsort(gv, stable=True)
print("unique_consecutive :", T(lambda: torch.unique_consecutive(gv[order], return_counts=True)))
u, c = torch.unique_consecutive(gv[order], return_counts=True)
print("repeat_interleave  :", T(lambda: torch.repeat_interleave(c > 10, c)))
print("counts.cpu()       :", T(lambda: c[c > 10].cpu()))
print("full x2            :", T(lambda: (torch.full((512 * 512,), -9999.0, device=dev), torch.full((512 * 512,), -9999.0, device=dev))))
