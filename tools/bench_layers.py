"""per-layer timing of one train step (serial launches, events around every kernel): python tools/bench_layers.py [pattern] [--batch 16]"""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ap = argparse.ArgumentParser(); ap.add_argument("pattern", nargs="?", default=""); ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--reps", type=int, default=5); ap.add_argument("--precision", default="fp32")
a = ap.parse_args()
from bench import synth_batch
from starcop_amd import model_module as mm
dev = torch.device("cuda:0"); torch.manual_seed(1234)
model = mm.ModelModule(mm.default_settings(pos_weight=1, precision=a.precision)).to(dev).train()
opt = model.configure_optimizers()["optimizer"]
batch = synth_batch(a.batch, 512, 512, 1234, dev)
net = model.network
for _ in range(3): model.fused_train_step(batch, opt)
torch.cuda.synchronize()
net.profile, net.profile_detail, net.overlap_wgrad = {}, True, False
for _ in range(a.reps): model.fused_train_step(batch, opt)
prof = net.collect_profile()
rows = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for v in prof.values()) / a.reps
print(f"total {tot:.3f} ms/step over {len(rows)} (op, family) entries")
sel = 0.0
for k, v in rows:
    if a.pattern and a.pattern not in k: continue
    ms = v["ms"] / a.reps; sel += ms
    tf = v["flop"] / v["ms"] / 1e9 if v["ms"] else 0; gb = v["bytes"] / v["ms"] / 1e6 if v["ms"] else 0
    print(f"{ms*1e3:8.1f} us  x{v['n']//a.reps:2d}  {tf:7.1f} TF/s  {gb:7.0f} GB/s  {k}")
print(f"selected {sel:.3f} ms/step")
