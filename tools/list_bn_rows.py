"""per BatchNorm of the training forward at a batch size: producer op type, statistics kind, partial rows, channels and the bytes a
finalizing work-group would read (rows x C x 8) -- input for sizing a producer-tail finalize.  python tools/list_bn_rows.py [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from starcop_amd import model_module as mm
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).train()
opt = model.configure_optimizers()["optimizer"]
b = synth_batch(B, 512, 512, 1, dev)
model.fused_train_step(b, opt)
net = model.network
plan = net._plans[(B, 512, 512)]
tot = small = 0
for op in net._ops:
    t = op["out"]
    if t.bn is None:
        continue
    rows = plan.srows[t.name]
    kb = rows * t.C * 8 / 1024
    tot += 1; small += kb <= 256
    print(f"{t.name:8s} {op['type']:6s} C {t.C:5d} @ {512 >> t.shift:4d}^2  rows {rows:6d}  {kb:9.1f} KB{'  <= 256 KB' if kb <= 256 else ''}")
print(f"{tot} BatchNorms, {small} with <= 256 KB of partial rows")
