"""stage times of pipeline.emit_scene_predict on an EMIT-like cube: python tools/profile_emit_scene.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from starcop_amd import model_module as mm, pipeline, mag1c
from starcop_amd.features import emit_to_aviris_input, ratio_2c_match_c_from_sums_outlier
from starcop_amd.model_module import masks_from_logits
dev = "cuda"
g3 = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g3_templates.npz"))
te = g3["emit_template_kept"][:, 1]
wl = np.linspace(381.0, 2493.0, 285)
keep = np.nonzero((wl >= 2122.0) & (wl <= 2488.0))[0]
t = np.interp(wl[keep], np.linspace(2122.0, 2488.0, te.size), te)
gen = torch.Generator(device=dev).manual_seed(11)
raw = ((torch.rand(285, generator=gen, device=dev) * 5 + 1) * (1 + 0.05 * torch.randn(1280, 1242, 285, generator=gen, device=dev))).float().contiguous()
model = mm.ModelModule(mm.default_settings(pos_weight=1)).to(dev).eval()


def T(name, fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    torch.cuda.synchronize(); print(f"{name:28s} {(time.perf_counter() - t0) / reps * 1e3:7.3f} ms"); return r


sub = T("band slice .contiguous()", lambda: raw[..., int(keep[0]):int(keep[-1]) + 1].contiguous())
mf, alb = T("mag1c_columns", lambda: mag1c.mag1c_columns(sub, t, -9999.0, column_step=2))
rgb = T("rgb planes", lambda: raw[..., pipeline.nearest_bands(wl)].permute(2, 0, 1).contiguous())
x = T("emit_to_aviris_input", lambda: emit_to_aviris_input(mf, rgb))
ia, ir = pipeline.nearest_bands(wl, (2350, 2310))
pa = T("ratio planes", lambda: (raw[..., ia].contiguous(), raw[..., ir].contiguous()))
T("band ratio", lambda: ratio_2c_match_c_from_sums_outlier(pa[0], pa[1]))
with torch.no_grad():
    lg = T("model(x[None])", lambda: model(x[None]))
    T("masks_from_logits", lambda: masks_from_logits(lg.contiguous()))
    T("emit_scene_predict (all)", lambda: pipeline.emit_scene_predict(model, raw, wl, t, column_step=2, ratio_bands=(2350, 2310)))
