"""Debug aid: per-parameter gradient error of the HIP network vs the fp32 and fp64 CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import copy
import torch
import torch.nn.functional as F
from test_gpu_unet import make_pair, synth_batch, ref_normalize, to_dev
from hip_ops import relerr

B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
model, ref = make_pair(seed=3, pos_weight=1.0)
model.train(); ref.train()
ref64 = copy.deepcopy(ref).double()
batch = synth_batch(B, H, W, seed=5)


def run_ref(net, dt):
    x = ref_normalize(batch["input"]).to(dt)
    logits = net(x)
    loss = (F.binary_cross_entropy_with_logits(logits, batch["output"].to(dt), reduction="none") * batch["weight_loss"].to(dt)).mean()
    net.zero_grad(); loss.backward()
    return logits.detach(), {k: p.grad.clone() for k, p in net.named_parameters()}


l32, g32 = run_ref(ref, torch.float32)
l64, g64 = run_ref(ref64, torch.float64)
loss = model.training_step(to_dev(batch), 0)
loss.backward()
lh = model.network._plans[(B, H, W)].buf["logits"]
print("logits: hip-vs-64 %.2e   ref32-vs-64 %.2e" % (relerr(lh, l64), relerr(l32, l64)))
names = [k for k, _ in model.network.named_parameters()]
print("%-48s %10s %10s" % ("param (reverse order)", "hip-vs-64", "ref32-vs-64"))
for k in reversed(names):
    p = dict(model.network.named_parameters())[k]
    print("%-48s %10.2e %10.2e" % (k, relerr(p.grad, g64[k]), relerr(g32[k], g64[k])))
