import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from starcop_amd import model_module as mm
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).train()
opt = model.configure_optimizers()["optimizer"]
batch = synth_batch(2, 128, 128, 1, dev)
for _ in range(2):
    model.fused_train_step(batch, opt)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        model.fused_train_step(batch, opt)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print("capture ok")
except Exception:
    traceback.print_exc()
