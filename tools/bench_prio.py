"""experiment: main chain on a HIGH-priority HIP stream (weight gradients stay on the normal-priority side stream)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from starcop_amd import model_module as mm
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).train()
opt = model.configure_optimizers()["optimizer"]
batch = synth_batch(16, 512, 512, 1234, dev)
def run(stream):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(5): model.fused_train_step(batch, opt)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(30): model.fused_train_step(batch, opt)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 30 * 1e3
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else None)
print("default stream      :", round(run(None), 3), "ms/step")
hp = torch.cuda.Stream(priority=-1)
print("high-priority main  :", round(run(hp), 3), "ms/step")
lp = torch.cuda.Stream(priority=0)
print("normal non-default  :", round(run(lp), 3), "ms/step")
print("default stream again:", round(run(None), 3), "ms/step")
