"""sha256 of the logits and of every gradient of one training step on a fixed batch: run it under two builds of the library
(STARCOP_HIP_LIB=...) to show that a kernel change is bit-neutral.  python tools/step_digest.py [--size 256] [--batch 4]"""
import sys, os, argparse, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from starcop_amd import model_module as mm
ap = argparse.ArgumentParser(); ap.add_argument("--size", type=int, default=256); ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--precision", default="fp32")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = mm.ModelModule(mm.default_settings(pos_weight=1, precision=a.precision)).to(dev).train()
batch = synth_batch(a.batch, a.size, a.size, 1, dev)
net = model.network
model.zero_grad()
loss = model.training_step(batch, 0)
loss.backward()
h = hashlib.sha256()
h.update(net._plans[(a.batch, a.size, a.size)].buf["logits"].detach().cpu().numpy().tobytes())
print("logits", h.hexdigest()[:16], "loss", float(loss))
g = hashlib.sha256()
for k, p in sorted(net.named_parameters()):
    g.update(p.grad.detach().cpu().numpy().tobytes())
print("grads ", g.hexdigest()[:16])
