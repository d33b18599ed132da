"""tiled scene inference: exactness and cost of the tiling modes (configs[4]); python tools/bench_tiling.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from starcop_amd import model_module as mm, pipeline
dev = "cuda"
torch.manual_seed(3)
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).eval()
g = torch.Generator().manual_seed(4)
H, W = 1280, 1248
x = torch.cat([torch.randn(1, H, W, generator=g).abs() * 600, torch.rand(3, H, W, generator=g) * 100 + 5]).to(dev)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps


with torch.no_grad():
    whole = model(x[None])[0, 0]
    tw = timeit(lambda: model(x[None]))
    print(f"whole scene {H}x{W}: {tw*1e3:.2f} ms")
    for tile, halo, strips in ((512, 320, False), (512, 320, True), (640, 320, True), (512, 128, False), (512, 128, True), (512, 64, True), (512, 192, True), (1024, 128, False)):
        t = pipeline.tiled_logits(model, x, tile=tile, halo=halo, strips=strips)
        d = (t - whole).abs() / whole.abs().max()
        agree = float(((t >= 0) == (whole >= 0)).float().mean())
        dt = timeit(lambda: pipeline.tiled_logits(model, x, tile=tile, halo=halo, strips=strips), 3)
        print(f"tile {tile} halo {halo} strips {strips}: {dt*1e3:7.2f} ms ({dt/tw:.2f}x whole)  max rel err {float(d.max()):.2e}  mean {float(d.mean()):.2e}  mask agreement {agree:.6f}")
