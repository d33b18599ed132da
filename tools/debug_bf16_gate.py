"""loss trajectory of the bf16 gate (tests/test_gpu_unet512.py::test_bf16_mode_trains_like_fp32_at_512): python tools/debug_bf16_gate.py [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from starcop_amd import model_module as mm
B, T, steps = 16, 512, int(sys.argv[1]) if len(sys.argv) > 1 else 300
train = bench.synth_batch(B, T, T, 4321, "cuda")
for prec in ("fp32", "bf16"):
    torch.manual_seed(0)
    model = mm.ModelModule(mm.default_settings(pos_weight=1, lr=1e-3, precision=prec)).to("cuda").train()
    opt = model.configure_optimizers()["optimizer"]
    losses = [float(model.fused_train_step(train, opt).item()) / (B * T * T) for _ in range(steps)]
    print(prec, " ".join(f"{l:.4f}" for l in losses[::10]))
    big = [(i, round(l, 4)) for i, l in enumerate(losses) if i > 20 and l > 3 * min(losses[max(0, i - 20):i])]
    print("   spikes:", big[:12])
