"""per-call wall time of the eval forward after the same sequence bench.py's extras run (train steps, Lightning-path steps): localises
one-off stalls inside bench.py's 10 timed repetitions.  python tools/debug_eval_time.py"""
import sys, os, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from starcop_amd import model_module as mm
dev = torch.device("cuda:0"); torch.manual_seed(1234)
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).train()
opt = model.configure_optimizers()["optimizer"]
b = synth_batch(16, 512, 512, 1234, dev)
for _ in range(12): model.fused_train_step(b, opt)
for _ in range(21):
    loss = model.training_step(b, 0); loss.backward(); opt.step(); opt.zero_grad()
for _ in range(21): model.fused_train_step(b, opt)
torch.cuda.synchronize()
model.eval()
b16 = synth_batch(16, 512, 512, 77, dev)
if os.environ.get("NOGC"): gc.disable()
with torch.no_grad():
    for r in range(4):
        ts = []
        for i in range(11):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            model(b16["input"]); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("round", r, " ".join(f"{t:.2f}" for t in ts), "| gc counts", gc.get_count())
