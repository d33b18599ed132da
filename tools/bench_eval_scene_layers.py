"""per-op time of one eval forward on a 1 x 4 x 1280 x 1248 scene (events around every launch) + predict() split into upload / forward / download"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from starcop_amd import model_module as mm
m = mm.ModelModule(mm.default_settings(pos_weight=1)).to("cuda").eval()
x = bench.synth_batch(1, 1280, 1248, 77, "cuda")["input"]
net = m.network
with torch.no_grad():
    for _ in range(3): m(x)
    net.profile, net.profile_detail = {}, True
    for _ in range(5): m(x)
prof = net.collect_profile(); net.profile = None
tot = 0
for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"]):
    tot += v["ms"] / 5
    print(f"{v['ms'] / 5 * 1e3:8.1f} us  {k}")
print(f"total {tot:.3f} ms")
scene = np.random.default_rng(5).uniform(0, 100, size=(4, 1280, 1242)).astype(np.float32)
def T(fn, n=10):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"predict() {T(lambda: m.predict(scene)):.3f} ms; forward alone {T(lambda: m(x)):.3f} ms; H2D of the scene {T(lambda: torch.from_numpy(scene).cuda()):.3f} ms; D2H of the probability map {T(lambda: x[0, 0].cpu()):.3f} ms")
