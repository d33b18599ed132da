"""Single-tile inference latency: eager launches vs replay of a captured hipGraph (eval-mode forward + masks)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from starcop_amd import model_module as mm
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).eval()
for B in (1, 4, 16):
    x = synth_batch(B, 512, 512, 1, dev)["input"]
    def T(fn, n=50):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    with torch.no_grad():
        eager = T(lambda: model(x))
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            model(x); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                out = model(x)
        torch.cuda.synchronize()
        ref = model(x).clone()
        g.replay(); torch.cuda.synchronize()
        ok = torch.equal(out, ref)
        graph = T(g.replay)
    print(f"B={B}: eager {eager:.3f} ms, hipGraph replay {graph:.3f} ms, identical={ok}")
