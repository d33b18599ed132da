import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, copy
import torch.nn.functional as F
from test_gpu_unet import make_pair, synth_batch, to_dev, ref_normalize
from test_gpu_irt import _train_once
from hip_ops import relerr
model, ref = make_pair(seed=11)
batch_c = synth_batch(2, 128, 128, seed=12)
batch = to_dev(batch_c)
a = _train_once(model, batch, "0")
b = _train_once(model, batch, "all")
r64 = copy.deepcopy(ref).double().train()
lg = r64(ref_normalize(batch_c["input"]).double())
loss = (F.binary_cross_entropy_with_logits(lg, batch_c["output"].double(), reduction="none") * batch_c["weight_loss"].double()).mean()
loss.backward()
g64 = {n: p.grad for n, p in r64.named_parameters()}
rows = []
for n in a["grads"]:
    rows.append((relerr(b["grads"][n], a["grads"][n]), relerr(a["grads"][n], g64[n]), relerr(b["grads"][n], g64[n]), float(g64[n].abs().max()), n))
rows.sort(reverse=True)
for r in rows[:14]:
    print("fused-vs-sep %.1e  sep-vs-f64 %.1e  fused-vs-f64 %.1e  max|g| %.1e  %s" % r)
