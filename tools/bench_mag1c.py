"""mag1c throughput on the MI355X: BASELINE configs[2] (AVIRIS 125-band, 512 column groups x 512 px, fp32, alpha=0)
and the EMIT-like case (1280 x 1242 px, 49 bands, fp64, column_step=2, alpha=1e-4)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from starcop_amd import mag1c

dev = "cuda"
rng = np.random.default_rng(0)
g3 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g3_templates.npz"))


def cube(H, W, S, templ):
    base = rng.uniform(1, 6, size=S)
    c = base * (1 + 0.05 * rng.standard_normal((H, W, S)))
    conc = np.zeros((H, W)); conc[H // 3:H // 3 + 60, W // 4:W // 4 + 40] = 2000.0
    return (c * (1 + conc[..., None] * 1e-5 * templ)).astype(np.float32)


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


S = 125
t125 = np.interp(np.linspace(0, 72, S), np.arange(73), g3["aviris_template_kept"][:, 1])
x = torch.from_numpy(cube(512, 512, S, t125)).to(dev)
groups = np.arange(1, 513)[None, :].repeat(512, 0)
dt = timeit(lambda: mag1c.acrwl1mf_by_groups(x, t125, groups), 3)
print(f"cfg3 AVIRIS 512x512x{S} fp32, 512 column groups, 30 iters: {dt*1e3:.1f} ms/tile  {1/dt:.2f} tiles/s (incl. sort/pack/scatter)")

S73 = 73
x73 = torch.from_numpy(cube(512, 512, S73, g3["aviris_template_kept"][:, 1])).to(dev)
dt = timeit(lambda: mag1c.acrwl1mf_by_groups(x73, g3["aviris_template_kept"][:, 1], groups), 3)
print(f"AVIRIS-NG real grid 512x512x{S73} fp32: {dt*1e3:.1f} ms/tile  {1/dt:.2f} tiles/s")

te = g3["emit_template_kept"][:, 1]
raw = torch.from_numpy(cube(1280, 1242, te.size, te)).to(dev)
dt = timeit(lambda: mag1c.mag1c_columns(raw, te, -9999.0, column_step=2), 3)
print(f"EMIT 1280x1242x{te.size} fp64 column_step=2 (621 groups x 2560 px): {dt*1e3:.1f} ms/granule  {1280*1242/dt/1e6:.2f} Mpx/s  ({1280*1242/262144/dt:.2f} tile-eq/s)")

# setup cost vs iteration cost (cfg3 shape)
for ni in (0, 30):
    dt = timeit(lambda: mag1c.acrwl1mf_by_groups(x, t125, groups, num_iter=ni), 3)
    print(f"cfg3 num_iter={ni}: {dt*1e3:.2f} ms")
xb = x.reshape(512 * 512, S)[None]
