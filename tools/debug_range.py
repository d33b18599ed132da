import sys, torch, warnings
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_unet import make_pair, synth_batch, ref_normalize, to_dev
from hip_ops import relerr
batch = synth_batch(2, 64, 64, seed=8)
for prec in ("fp32-x3", "fp32"):
    model2, ref2 = make_pair(seed=9)
    sd = {k: v.clone() for k, v in model2.network.state_dict().items()}
    sd["decoder.blocks.2.conv1.1.running_var"][:] = 1e-12
    sd["decoder.blocks.2.conv1.1.running_mean"][:] = -200.0
    model2.network.load_state_dict(sd)
    model2.network.precision = prec
    model2.network.range_check_every = 0 if prec == "fp32-x3" else 200
    ref2.load_state_dict({k: v.cpu() for k, v in sd.items()})
    model2.eval(); ref2.eval()
    outs = {}
    mods = dict(ref2.named_modules())
    hs = [mods[f"decoder.blocks.{b}.conv{k}.0"].register_forward_hook(lambda m, i, o, n=f"d{b}{'ab'[k-1]}": outs.__setitem__(n, o)) for b in range(5) for k in (1, 2)]
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = ref2(ref_normalize(batch["input"]))
        got = model2(to_dev(batch)["input"])
    plan = list(model2.network._plans.values())[0]
    print(prec, "->", model2.network.precision, "logits", relerr(got, want))
    for n in ("d0a", "d0b", "d1a", "d1b", "d2a", "d2b", "d3a", "d3b", "d4a", "d4b"):
        print("   ", n, relerr(plan.buf[n], outs[n]), float(outs[n].abs().max()))
