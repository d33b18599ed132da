"""Per-layer timing of the HIP train step (events on the launch stream): ms and achieved TFLOP/s per op."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from starcop_amd import model_module as mm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).train()
opt = model.configure_optimizers()["optimizer"]
batch = synth_batch(B, T, T, 1, dev)
net = model.network
net.overlap_wgrad = False     # serial launches: clean per-kernel timings
for _ in range(3):
    model.fused_train_step(batch, opt)
torch.cuda.synchronize()
net.profile, net.profile_detail = {}, True
R = 3
for _ in range(R):
    model.fused_train_step(batch, opt)
prof = net.collect_profile()
net.profile = None
rows = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for v in prof.values()) / R
print(f"total instrumented GPU time per step: {tot:.2f} ms (B={B}, {T}x{T})")
print(f"{'op|family':60s} {'ms/step':>9s} {'TFLOP/s':>9s}")
for k, v in rows:
    tf = v["flop"] / (v["ms"] * 1e-3) / 1e12 if v["flop"] else 0.0
    print(f"{k:60s} {v['ms']/R:9.3f} {tf:9.1f}")
