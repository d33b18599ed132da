"""experiment: main stream and weight-gradient stream on DISJOINT CU sets (hipExtStreamCreateWithCUMask): python tools/bench_cumask.py <side share in 1/8ths>"""
import ctypes, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import synth_batch
from starcop_amd import model_module as mm
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = (ctypes.c_uint32 * 8)(*[sum(((bits >> (32 * w + b)) & 1) << b for b in range(32)) for w in range(8)])
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 2          # side gets CUs with (i % 8) < k
pattern = sys.argv[2] if len(sys.argv) > 2 else "mod"
side_bits = main_bits = 0
for i in range(256):
    on = (i % 8) < k if pattern == "mod" else i < 32 * k
    if on: side_bits |= 1 << i
    else: main_bits |= 1 << i
dev = torch.device("cuda:0"); torch.manual_seed(1234)
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).train()
opt = model.configure_optimizers()["optimizer"]
batch = synth_batch(16, 512, 512, 1234, dev)
net = model.network
def run(tag, main=None, side=None, steps=30):
    if side is not None: net._side_stream = side
    ctx = torch.cuda.stream(main) if main is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(5): model.fused_train_step(batch, opt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps): model.fused_train_step(batch, opt)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{tag}: {dt*1e3:.3f} ms/step  {16/dt:.1f} tiles/s")
run("default streams")
if k > 0:
    run(f"side masked to {k}/8 of the CUs ({pattern}), main unmasked", None, masked_stream(side_bits))
    run(f"side {k}/8, main {8-k}/8 (disjoint, {pattern})", masked_stream(main_bits), masked_stream(side_bits))
