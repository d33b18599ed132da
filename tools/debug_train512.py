import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
from starcop_amd import model_module as mm
DEV = "cuda"; B, T, steps = 16, 512, int(os.environ.get("STEPS", "400"))
train = bench.synth_batch(B, T, T, 4321, DEV)
for prec in sys.argv[1:] or ("fp32", "bf16", "fp32-x3"):
    torch.manual_seed(0)
    model = mm.ModelModule(mm.default_settings(pos_weight=1, lr=float(os.environ.get("LR", "1e-3")), precision=prec)).to(DEV).train()
    opt = model.configure_optimizers()["optimizer"]
    losses = []
    for i in range(steps):
        losses.append(float(model.fused_train_step(train, opt).item()) / (B * T * T))
    print(prec, "->", model.network.precision, " ".join(f"{l:.4f}" for l in losses[::25]))
    mx = int(np.argmax(np.array(losses[100:]))) + 100
    print("   max after step 100 at", mx, ":", " ".join(f"{l:.4f}" for l in losses[max(0, mx - 10):mx + 3]))
    model.eval()
    with torch.no_grad():
        pred = (model(train["input"]) >= 0).long()
    y = train["output"].long()
    tp = int(((pred == 1) & (y == 1)).sum()); fp = int(((pred == 1) & (y == 0)).sum()); fn = int(((pred == 0) & (y == 1)).sum())
    print("   F1", 2 * tp / max(2 * tp + fp + fn, 1))
