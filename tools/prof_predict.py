"""where ModelModule.predict() spends its time on a host (4, 1280, 1242) scene: python tools/prof_predict.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from starcop_amd import model_module as mm
from starcop_amd.padding import find_padding
m = mm.ModelModule(mm.default_settings(pos_weight=1)).to("cuda").eval()
scene = np.random.default_rng(5).uniform(0, 100, size=(4, 1280, 1242)).astype(np.float32)
def T(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"predict()                         {T(lambda: m.predict(scene)):.3f} ms")
print(f"eval() + train(was) toggles       {T(lambda: (m.eval(), m.train(False))):.3f} ms")
print(f"np.ascontiguousarray + as_tensor  {T(lambda: torch.as_tensor(np.ascontiguousarray(scene))):.3f} ms")
xh = torch.as_tensor(scene)
print(f"H2D pageable .to(device)          {T(lambda: xh.to('cuda')):.3f} ms")
xp = xh.pin_memory()
print(f"H2D pinned                        {T(lambda: xp.to('cuda', non_blocking=True)):.3f} ms")
x = xh.to("cuda")[None]
(l, r) = find_padding(1242, 32)
print(f"F.pad reflect                     {T(lambda: F.pad(x, (l, r, 0, 0), mode='reflect')):.3f} ms")
xpad = F.pad(x, (l, r, 0, 0), mode="reflect")
with torch.no_grad():
    print(f"model(x) (forward + clone)        {T(lambda: m(xpad)):.3f} ms")
    z = m(xpad)
    print(f"masks_from_logits                 {T(lambda: mm.masks_from_logits(z)):.3f} ms")
    pr = mm.masks_from_logits(z)["prediction"][0]
    print(f"crop view .cpu().numpy()          {T(lambda: pr[:, 0:1280, l:l + 1242].cpu().numpy()):.3f} ms")
    print(f"crop .contiguous() then .cpu()    {T(lambda: pr[:, 0:1280, l:l + 1242].contiguous().cpu().numpy()):.3f} ms")
    ph = torch.empty((1, 1280, 1242), dtype=torch.float32).pin_memory()
    def d2h():
        ph.copy_(pr[:, 0:1280, l:l + 1242].contiguous(), non_blocking=True); torch.cuda.synchronize()
    print(f"crop contiguous -> pinned host    {T(d2h):.3f} ms")
# ---- predict() step by step with a device synchronisation after each (what a single cold call pays)
from starcop_amd.padding import padded_predict
def step_times():
    ts = []
    def tick(): torch.cuda.synchronize(); ts.append(time.perf_counter())
    tick()
    a = np.asarray(scene, dtype=np.float32); xx = torch.as_tensor(np.ascontiguousarray(a)).to("cuda")[None]; tick()
    xx = F.pad(xx, (l, r, 0, 0), mode="reflect"); tick()
    with torch.no_grad():
        zz = m(xx); tick()
        pp = mm.masks_from_logits(zz)["prediction"][0]; tick()
        o = pp[:, 0:1280, l:l + 1242].cpu().numpy(); tick()
    return [1e3 * (b - a_) for a_, b in zip(ts, ts[1:])]
for _ in range(3): step_times()
acc = np.mean([step_times() for _ in range(10)], 0)
print("synchronised steps: upload %.3f  pad %.3f  forward %.3f  masks %.3f  crop + download %.3f  = %.3f ms" % (*acc, acc.sum()))
print(f"predict() after the toggle fix    {T(lambda: m.predict(scene)):.3f} ms")
