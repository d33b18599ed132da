import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from starcop_amd import model_module as mm
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).train()
opt = model.configure_optimizers()["optimizer"]
batch = synth_batch(16, 512, 512, 1, dev)
for _ in range(5):
    model.fused_train_step(batch, opt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    model.fused_train_step(batch, opt)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/20:.2f} ms/step (CPU), total {1e3*(t2-t0)/20:.2f} ms/step")
# tiny batch: GPU work negligible -> pure host cost per step
b2 = synth_batch(1, 64, 64, 1, dev)
for _ in range(5):
    model.fused_train_step(b2, opt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    model.fused_train_step(b2, opt)
torch.cuda.synchronize()
print(f"1x64x64 step: {1e3*(time.perf_counter()-t0)/20:.2f} ms/step (host + launch bound)")
