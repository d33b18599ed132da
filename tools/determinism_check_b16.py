"""Run-to-run bit reproducibility of one training step at the benched shape (16 x 4 x 512 x 512): every parameter gradient of three
identical steps compared bitwise (round 5: 0 of 188 differ -- the sub-pixel, skip-tile, box-sum and 96-wide-tile kernels included)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from starcop_amd import model_module as mm
dev = torch.device("cuda", 0); torch.manual_seed(0)
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).train()
batch = synth_batch(16, 512, 512, 1, dev)
ref = None
for rep in range(3):
    model.zero_grad()
    loss = model.training_step(batch, 0); loss.backward(); torch.cuda.synchronize()
    g = [p.grad.clone() for p in model.network.parameters()]
    if ref is None: ref = g
    else: print("run", rep, "differing tensors:", sum(int(not torch.equal(a, b)) for a, b in zip(ref, g)), "of", len(g))
