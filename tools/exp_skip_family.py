"""elimination experiment on the whole step: replace a family of launches by no-ops (results wrong, timing valid) to see what the
family costs in the overlapped two-stream step:  python tools/exp_skip_family.py none|pwwgrad|wgrad3|bnfin|bnbfin|bnred"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from starcop_amd import model_module as mm, _lib
which = sys.argv[1] if len(sys.argv) > 1 else "none"
dev = torch.device("cuda:0"); torch.manual_seed(1234)
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).train()
opt = model.configure_optimizers()["optimizer"]
batch = synth_batch(16, 512, 512, 1234, dev)
lib = _lib.load()
class Wrap:
    def __init__(self, lib, skip): self._lib, self._skip = lib, skip
    def __getattr__(self, n):
        if n in self._skip: return lambda *a: 0
        return getattr(self._lib, n)
SK = {"none": [], "pwwgrad": ["sc_conv1x1_wgrad_pw3", "sc_conv2d_wgrad_mfma_deferred", "sc_wgrad_reduce_batch"],
      "wgrad3": ["sc_conv3x3_wgrad_bx3", "sc_conv3x3_wgrad_thin16"], "bnfin": ["sc_bn_finalize"], "bnbfin": ["sc_bn_bwd_finalize"],
      "bnred": ["sc_bn_bwd_reduce", "sc_bn_bwd_small"], "adds": ["sc_add_srcs_absmax"], "adam": ["sc_adam_step", "sc_adam_prepare"]}[which]
for _ in range(3): model.fused_train_step(batch, opt)
_lib._lib = Wrap(lib, SK)
for _ in range(3): model.fused_train_step(batch, opt)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): model.fused_train_step(batch, opt)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
print(f"skip {which}: {dt*1e3:.3f} ms/step")
