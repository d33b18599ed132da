import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, g9_util
from starcop_amd.unet_simple import SimpleUNet
from oracle.unet_ref import simple_unet_ref
net = SimpleUNet(4, 1); net.load_state_dict(g9_util.full_unet_state()); net = net.to("cuda").train()
x, r = g9_util.full_unet_grad_case()
y = net(x.cuda()); (y * r.cuda()).sum().backward()
errs = g9_util.full_unet_grad_errs(g9_util.load(), {k: p.grad for k, p in net.named_parameters()})
for k in g9_util.FULL_KEYS[::-1]:
    if k.endswith("bias"): print(f"{k:28s} {errs[k]:.2e}")
    else: print(f"{k:28s} cisum {errs[k+'.cisum']:.2e} cosum {errs[k+'.cosum']:.2e}")
# full comparison against the CPU oracle's gradients (not only marginals)
sd = {k: v.clone().requires_grad_(True) for k, v in g9_util.full_unet_state().items()}
(simple_unet_ref(sd, x) * r).sum().backward()
for k, p in net.named_parameters():
    a, b = p.grad.cpu().double(), sd[k].grad.double()
    print(f"{k:28s} full rel err {float((a-b).abs().max()/b.abs().max()):.2e}   |diff|>1e-3*max: {int(((a-b).abs() > 1e-3*b.abs().max()).sum())} of {a.numel()}")
print("---- fp32 CPU oracle vs fp64 CPU oracle, and HIP vs fp64")
sd64 = {k: v.clone().double().requires_grad_(True) for k, v in g9_util.full_unet_state().items()}
(simple_unet_ref(sd64, x.double()) * r.double()).sum().backward()
for k, p in net.named_parameters():
    t = sd64[k].grad
    e32 = float((sd[k].grad.double() - t).abs().max() / t.abs().max())
    eh = float((p.grad.cpu().double() - t).abs().max() / t.abs().max())
    print(f"{k:28s} fp32-CPU vs fp64 {e32:.2e}   HIP vs fp64 {eh:.2e}")
