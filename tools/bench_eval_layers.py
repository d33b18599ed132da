"""per-op time of one eval forward (events around every launch): python tools/bench_eval_layers.py [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from starcop_amd import model_module as mm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = mm.ModelModule(mm.default_settings(pos_weight=1)).to("cuda").eval()
b = bench.synth_batch(B, 512, 512, 77, "cuda")
net = m.network
with torch.no_grad():
    for _ in range(3): m(b["input"])
    net.profile, net.profile_detail = {}, True
    for _ in range(5): m(b["input"])
prof = net.collect_profile()
tot = 0
for k, v in prof.items():
    tot += v["ms"] / 5
    print(f"{v['ms'] / 5 * 1e3:8.1f} us  {k}")
print(f"total {tot:.3f} ms")
