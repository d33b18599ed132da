"""Run-to-run bit reproducibility of one training step (same weights, same batch): which gradients differ, if any."""
import sys, os, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from starcop_amd import model_module as mm
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).train()
batch = synth_batch(4, 256, 256, 1, dev)
net = model.network
grads = []
for rep in range(3):
    loss = model.training_step(batch, 0)
    net.zero_grad(set_to_none=False) if rep else None
    model.zero_grad()
    loss = model.training_step(batch, 0)
    loss.backward()
    grads.append({k: p.grad.detach().clone() for k, p in net.named_parameters()})
    logits = net._plans[(4, 256, 256)].buf["logits"].clone()
    if rep == 0:
        l0 = logits
    else:
        print("logits bit-identical to run 0:", torch.equal(l0, logits))
for rep in (1, 2):
    diff = [k for k in grads[0] if not torch.equal(grads[0][k], grads[rep][k])]
    worst = max((float((grads[0][k] - grads[rep][k]).abs().max() / grads[0][k].abs().max().clamp_min(1e-30)) for k in diff), default=0.0)
    print(f"run {rep}: {len(diff)} of {len(grads[0])} gradient tensors differ bitwise; worst rel diff {worst:.2e}; e.g. {diff[:4]}")
