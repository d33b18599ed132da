"""whole-scene inference (ModelModule.predict on a host (4, 1280, 1242) scene; device-side forward alone too) -- run under STARCOP_SP=0 / 1 / all"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from starcop_amd import model_module as mm

torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).eval()
scene = np.random.default_rng(5).uniform(0, 100, size=(4, 1280, 1242)).astype(np.float32)
xd = torch.from_numpy(np.pad(scene, ((0, 0), (0, 0), (3, 3)), "reflect"))[None].to(dev)


def timeit(fn, reps):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


with torch.no_grad():
    print(f"STARCOP_SP={os.environ.get('STARCOP_SP', '1')}: predict() {timeit(lambda: model.predict(scene), 20) * 1e3:.2f} ms, "
          f"device forward of the padded (1, 4, 1280, 1248) scene {timeit(lambda: model(xd), 20) * 1e3:.2f} ms")
