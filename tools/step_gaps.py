"""Does the device idle BETWEEN training steps when the bench loop runs eagerly?  Events before / after every step on the launch stream:
per-step device span, the device-side gap from one step's end to the next one's start, and the host's enqueue time per step.
usage: python tools/step_gaps.py [steps=40]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from starcop_amd import model_module as mm

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).train()
opt = model.configure_optimizers()["optimizer"]
batch = synth_batch(16, 512, 512, 1234, dev)
for _ in range(8):
    model.fused_train_step(batch, opt)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
host = []
t0 = time.perf_counter()
for a, b in ev:
    h0 = time.perf_counter()
    a.record()
    model.fused_train_step(batch, opt)
    b.record()
    host.append(time.perf_counter() - h0)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
spans = [a.elapsed_time(b) for a, b in ev]
gaps = [ev[i][1].elapsed_time(ev[i + 1][0]) for i in range(steps - 1)]
print(f"{steps} steps: wall {wall / steps * 1e3:.3f} ms/step; host enqueue {sum(host) / steps * 1e3:.3f} ms/step (all enqueued after {t_enq * 1e3:.1f} ms)")
print(f"device span per step: mean {sum(spans) / steps:.3f} ms (min {min(spans):.3f}, max {max(spans):.3f}); "
      f"gap between steps: mean {sum(gaps) / len(gaps) * 1e3:.1f} us (max {max(gaps) * 1e3:.1f})")
