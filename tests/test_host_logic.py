"""Host-side mirror of the reference interface (no GPU): constructor/attribute/state_dict contract, padding
arithmetic, metrics, normaliser table, band masks, settings, sharding helpers -- against the golden vectors."""
import os

import numpy as np
import pytest
import torch

from oracle.unet_ref import UnetMobileNetV2
from starcop_amd import metrics as M
from starcop_amd import model_module as mm
from starcop_amd import padding
from starcop_amd.mag1c import generate_template_from_bands, get_mask_bad_bands
from starcop_amd.normalizer import BAND_NORMALIZATION, DataNormalizer
from starcop_amd.parallel import shard_range

G = os.path.join(os.path.dirname(__file__), "golden")


def test_model_module_surface_and_state_dict():
    model = mm.ModelModule(mm.default_settings())
    for a in ("network", "normalizer", "num_channels", "num_classes", "lr", "lr_decay", "lr_patience", "loss_name",
              "reduction", "pos_weight", "loss_function", "confusion_matrix", "classification_confusion_matrix"):
        assert hasattr(model, a), a
    for meth in ("forward", "training_step", "val_step", "validation_step", "test_step", "val_epoch_end",
                 "validation_epoch_end", "test_epoch_end", "configure_optimizers", "batch_with_preds",
                 "pred_classification", "log", "debug", "predict"):
        assert callable(getattr(model, meth)), meth
    assert model.num_channels == 4 and model.num_classes == 1 and model.reduction == "none"
    assert float(model.pos_weight) == 15.0 and not model.pos_weight.requires_grad
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    n_frozen = sum(p.numel() for p in model.parameters() if not p.requires_grad)
    assert (n_train, n_frozen) == (6_629_233, 17)                    # Lightning summary of the reference run
    sd = model.state_dict()
    ref = UnetMobileNetV2(4, 1)
    assert {k[len("network."):] for k in sd if k.startswith("network.")} == set(ref.state_dict())
    for k, v in ref.state_dict().items():
        assert sd["network." + k].shape == v.shape, k
    assert {k for k in sd if not k.startswith("network.")} == {
        "pos_weight", "loss_function.pos_weight", "normalizer.offsets_input", "normalizer.factors_input",
        "normalizer.clip_min_input", "normalizer.clip_max_input"}
    ref.load_state_dict(model.network.state_dict(), strict=True)       # checkpoints interchange
    with pytest.raises(Exception, match="No model implemented"):
        mm.configure_architecture("unet", 4, 1, mm.default_settings().model)
    with pytest.raises(ValueError):
        mm.load_weights("/nonexistent/model.pt")


def test_default_settings_match_config_yaml():
    s = mm.default_settings()
    assert s.dataset.input_products == ["mag1c", "TOA_AVIRIS_640nm", "TOA_AVIRIS_550nm", "TOA_AVIRIS_460nm"]
    assert (s.model.lr, s.model.lr_decay, s.model.lr_patience, s.model.pos_weight) == (1e-4, 0.5, 4, 15)
    assert "use_weight_loss" in s.dataset and s.model.semseg_backbone == "mobilenet_v2"


def test_normalizer_table_and_parameters():
    g = np.load(os.path.join(G, "g4_normalizer.npz"))
    n = DataNormalizer(mm.default_settings())
    assert str(n.factors_input.dtype) == str(g["cfg4_param_dtype"]) == "torch.int64"
    assert n.factors_input.reshape(-1).tolist() == [1750, 60, 60, 60] and n.clip_max_input.reshape(-1).tolist() == [2, 2, 2, 2]
    assert n.offsets_input.shape == (4, 1, 1) and n.factors_output is None
    c = n.consts("cpu")
    assert c.shape == (4, 8) and c[:, 1].tolist() == [1750.0, 60.0, 60.0, 60.0] and c[:, 3].tolist() == [2.0] * 4
    assert len(BAND_NORMALIZATION) == 58 and BAND_NORMALIZATION["ratio_wv3_B8_B8MLR_SanchezGarcia22_simplediv"]["offset"] == -0.5
    y = torch.from_numpy(g["cfg4_y"])
    assert torch.equal(n.normalize_y(y), y)
    with pytest.warns(UserWarning):
        DataNormalizer(mm.Settings(dataset=dict(input_products=["not_a_product"], output_products=["labelbinary"])))


def test_padding_arithmetic():
    g = np.load(os.path.join(G, "g6_padding.npz"))
    assert np.array_equal(np.array([padding.find_padding(int(v), 32) for v in g["v"]]), g["pad32"])
    assert np.array_equal(np.array([padding.find_padding(int(v)) for v in g["v"]]), g["pad8"])

    class Smooth(torch.nn.Module):
        def forward(self, t):
            return torch.nn.functional.avg_pool2d(t.sum(1, keepdim=True), 3, 1, 1, count_include_pad=False)
    out3 = padding.padded_predict(g["pp_x"], Smooth(), 32)
    out2 = padding.padded_predict(g["pp_x"], lambda t: Smooth()(t)[:, 0], 32)
    assert out3.shape == (1, 45, 70) and out2.shape == (45, 70)
    assert np.abs(out3 - g["pp_out3d"]).max() < 1e-6 and np.abs(out2 - g["pp_out2d"]).max() < 1e-6
    with pytest.raises(AssertionError):
        padding.padded_predict(np.zeros((2, 3, 4, 5), dtype=np.float32), Smooth())


def test_metrics_against_reference():
    g = np.load(os.path.join(G, "g7_metrics.npz"))
    for i, cm in enumerate(g["cm"]):
        t = torch.from_numpy(cm)
        for fn in M.METRICS_CONFUSION_MATRIX + [M.TP, M.TN, M.FP, M.FN, M.FPR]:
            assert float(fn(t)) == pytest.approx(float(g[fn.__name__][i]), rel=1e-6, abs=1e-7), fn.__name__
    cmobj = M.BinaryConfusionMatrix()
    cmobj.update(torch.tensor([1, 1, 0, 0, 1]), torch.tensor([1, 0, 0, 1, 1]))
    assert cmobj.compute().tolist() == [[1, 1], [1, 2]]          # [[TN, FP], [FN, TP]]
    cmobj.reset()
    assert int(cmobj.compute().sum()) == 0


def test_differences_and_band_mask():
    g5 = np.load(os.path.join(G, "g5_masks.npz"))
    assert np.array_equal(mm.differences(torch.from_numpy(g5["pb_64"]), torch.from_numpy(g5["gt_64"])).numpy(), g5["diff_64"])
    g3 = np.load(os.path.join(G, "g3_templates.npz"))
    assert np.array_equal(get_mask_bad_bands(g3["badband_wave"]), g3["badband_keep"])
    # the CH4 look-up table ships with the package (starcop_amd/data/, BSD-3 data file of mag1c upstream): no lut_dir needed
    t = generate_template_from_bands(g3["aviris_centers"], g3["aviris_fwhm"])
    assert np.abs(t[g3["aviris_keep"]] - g3["aviris_template_kept"]).max() < 1e-9 * np.abs(g3["aviris_template_kept"]).max()
    te = generate_template_from_bands(g3["emit_centers"], g3["emit_fwhm"])
    assert np.abs(te[g3["emit_keep"]] - g3["emit_template_kept"]).max() < 1e-9 * np.abs(g3["emit_template_kept"]).max()
    with pytest.raises(RuntimeError):
        generate_template_from_bands([2300.0, np.nan], [5.0, 5.0])
    with pytest.raises(RuntimeError):
        generate_template_from_bands([2300.0, 2310.0], [5.0])



def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 512, 621):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_scene_tiles_partition_and_stitch():
    """sliding-window tiler of BASELINE configs[4]: the cores partition the scene, the windows are the cores grown by the halo and
    clipped to the scene, everything stays on the 32-px grid of the encoder stride; stitch() inverts the cut"""
    from starcop_amd.pipeline import scene_tiles, stitch
    for H, W, tile, halo in ((1280, 1216, 512, 320), (512, 512, 512, 64), (96, 160, 64, 32), (1024, 1024, 256, 0)):
        r = scene_tiles(H, W, tile, halo)
        cover = torch.zeros(H, W, dtype=torch.int32)
        for y0, y1, x0, x1, wy0, wy1, wx0, wx1 in r.tolist():
            cover[y0:y1, x0:x1] += 1
            assert wy0 == max(0, y0 - halo) and wy1 == min(H, y1 + halo) and wx0 == max(0, x0 - halo) and wx1 == min(W, x1 + halo)
            assert all(v % 32 == 0 for v in (y0, y1, x0, x1, wy0, wy1, wx0, wx1))
            assert 0 < y1 - y0 <= tile and 0 < x1 - x0 <= tile
        assert int(cover.min()) == int(cover.max()) == 1
        scene = torch.arange(H * W, dtype=torch.float32).reshape(H, W)
        cores = torch.zeros(r.shape[0], tile, tile)
        for i, (y0, y1, x0, x1, *_) in enumerate(r.tolist()):
            cores[i, :y1 - y0, :x1 - x0] = scene[y0:y1, x0:x1]
        assert torch.equal(stitch(cores, r, H, W), scene)
    with pytest.raises(ValueError):
        scene_tiles(1000, 1024, 512, 64)


def test_scene_tiles_square_and_strips():
    """tile table of the sliding-window inference (pipeline.scene_tiles): cores partition the scene, windows = cores + halo clipped to
    the scene, everything on the 32-pixel stride grid; strips = full-width rows with a vertical halo only"""
    from starcop_amd import pipeline
    H, W = 1280, 1248
    for strips in (False, True):
        r = pipeline.scene_tiles(H, W, 512, 320, strips=strips).tolist()
        cover = np.zeros((H, W), dtype=np.int32)
        for y0, y1, x0, x1, wy0, wy1, wx0, wx1 in r:
            cover[y0:y1, x0:x1] += 1
            assert wy0 == max(0, y0 - 320) and wy1 == min(H, y1 + 320) and wx0 == max(0, x0 - 320) and wx1 == min(W, x1 + 320)
            assert all(v % 32 == 0 for v in (y0, x0, wy0, wx0)) and (wy1 % 32 == 0 and wx1 % 32 == 0)
            if strips:
                assert (x0, x1, wx0, wx1) == (0, W, 0, W)
        assert (cover == 1).all()
        assert len(r) == (3 if strips else 9)
    work = lambda rows: sum((wy1 - wy0) * (wx1 - wx0) for _, _, _, _, wy0, wy1, wx0, wx1 in rows) / (H * W)
    assert work(pipeline.scene_tiles(H, W, 512, 320, strips=True).tolist()) < 0.55 * work(pipeline.scene_tiles(H, W, 512, 320).tolist())      # 1.95x vs 3.8x the scene (clipped at the borders)
    with pytest.raises(ValueError):
        pipeline.scene_tiles(1000, W, 512, 320)


def test_pointwise_kernel_choice_rules():
    """which pointwise family a launch takes (network._use_pw3): pure host logic, measured per layer and pass at batch 4 / 16 / 64
    (profiles/r03a-b_layers_*, profiles/r04_layers_b*_pw3_*.txt)"""
    from starcop_amd import network as nw
    if nw._PW3 != "1":
        pytest.skip("STARCOP_PW3 overridden")
    u = nw._use_pw3
    # batch 16 (the benched configuration): 32^2 / 16^2 expansions and features.18 forward on the register-only family, the long
    # contractions at 16^2 and the 256^2 planes not; the projections' data gradients yes, the expansions' no
    assert u(0, 16, 1024, 64, 384) and u(0, 16, 256, 160, 960) and u(0, 16, 256, 320, 1280) and u(0, 16, 1024, 576, 96)
    assert not u(0, 16, 256, 960, 160) and not u(0, 16, 65536, 16, 96) and not u(0, 16, 256, 576, 160)
    assert u(1, 16, 1024, 384, 64) and not u(1, 16, 1024, 64, 384) and not u(1, 16, 256, 160, 960)
    assert u(2, 16, 1024, 64, 384) and not u(2, 16, 1024, 96, 576) and not u(2, 16, 4, 64, 384)          # H*W % 8
    # batch 4: a long contraction has too few (pixel block, output block) pairs: the K-splitting kernel keeps the 32^2 / 16^2 projections
    assert not u(0, 4, 1024, 576, 96) and not u(0, 4, 256, 960, 160) and not u(0, 4, 1024, 384, 96) and u(0, 4, 1024, 64, 384)
    assert u(1, 4, 1024, 384, 64) and not u(1, 4, 256, 160, 960)
    # batch 64: every 32^2 / 16^2 layer moves over, forward and backward-data (72 vs 103, 83 vs 130 us on features.15-17); the 64^2
    # projections' data gradients and the large-batch weight gradients go back to the LDS-staged kernels
    assert u(0, 64, 256, 960, 160) and u(0, 64, 1024, 576, 96) and u(1, 64, 256, 160, 960) and u(1, 64, 1024, 64, 384)
    assert not u(1, 64, 4096, 192, 32) and u(1, 64, 1024, 384, 64) and not u(2, 64, 1024, 64, 384) and u(2, 64, 256, 160, 960)
    # 64^2 planes with a short contraction (features.5-.7 expansions forward, features.4-.6 projections' data gradients): the streaming
    # kernel inside sc_conv2d_mfma since late round 6; the long contractions of those planes stay where they were
    if nw._PWS64:
        assert not u(0, 16, 4096, 32, 192) and not u(1, 16, 4096, 192, 32) and not u(1, 16, 4096, 144, 32)
        assert u(0, 16, 4096, 192, 32) and u(0, 16, 4096, 144, 32)


def test_fused_block_rule():
    """where the expansion + depthwise pair of an inverted-residual block trains without its 6x tensor (network._use_irt, measured per
    block and batch: DESIGN.md 14): stride-2 blocks on planes of at least 256^2 whose expanded tensor is at least 256 MB -- features.2
    at 512^2 tiles from batch 11 -- and nothing the kernels do not
    take (stride 1, Cin not a multiple of 8 or > 32, hidden > 192); pure host logic + the library's host-side sc_irt_supported"""
    from starcop_amd import network as nw
    if nw._IRT != "1":
        pytest.skip("STARCOP_IRT overridden")
    assert nw._use_irt(16, 16, 96, 256, 256, 2) and nw._use_irt(64, 16, 96, 256, 256, 2)      # features.2 at 512^2 tiles, batch 16 / 64
    assert not nw._use_irt(4, 16, 96, 256, 256, 2)                 # batch 4: e = 101 MB sits in the Infinity Cache (step -5 %)
    assert not nw._use_irt(16, 24, 144, 128, 128, 2)               # features.4: measured slower (414 vs 341 us)
    assert not nw._use_irt(64, 24, 144, 128, 128, 2)               # ... and a tie at batch 64
    assert not nw._use_irt(16, 16, 96, 64, 64, 2)                  # features.2 at 128^2 crops
    assert not nw._use_irt(16, 24, 144, 256, 256, 1)               # stride 1 is not built
    from starcop_amd import _lib
    lib = _lib.load()
    assert lib.sc_irt_supported(32, 192, 64, 64, 2) == 1 and lib.sc_irt_supported(32, 192, 64, 64, 1) == 0
    assert lib.sc_irt_supported(64, 384, 32, 32, 2) == 0 and lib.sc_irt_supported(20, 120, 32, 32, 2) == 0 and lib.sc_irt_supported(32, 224, 32, 32, 2) == 0
    # rows / workspace are pure functions of the shape
    assert lib.sc_irt_rows(1, 16, 256, 256, 2) == 16 * 32 * 8 and lib.sc_irt_bwd_rows(16, 96, 256, 256) == 512
    assert lib.sc_irt_bwd_workspace_floats(16, 16, 96, 256, 256) > 16 * 16 * 256 * 256


def test_inference_block_rule():
    """which inverted-residual blocks run as ONE launch in inference (network._use_irb, measured per block: DESIGN.md 16): features.5 /
    .6 / .8-.13 (stride 1, Cin <= 96) and the stride-2 features.7 at 512^2 tiles -- not features.3 / .4 (hidden = 144), not features.14
    (a tie), not features.15-.17 (Cin = 160: slower); the same blocks on the planes of a 1280 x 1248 scene; host logic + sc_irb_supported"""
    from starcop_amd import _lib, network as nw
    if nw._IRB != "1":
        pytest.skip("STARCOP_IRB overridden")
    net = nw.HyperStarcopUNet(4, 1)
    def fused(N, H, W):
        out = []
        for i_e, (i_dw, i_pr, stride) in net._ir_blocks.items():
            cv_e, cv_p, tin = net._ops[i_e]["conv"], net._ops[i_pr]["conv"], net._ops[i_e]["ins"][0]
            if nw._use_irb(N, cv_e.in_channels, cv_e.out_channels, cv_p.out_channels, H >> tin.shift, W >> tin.shift, stride):
                out.append(net._ops[i_pr]["out"].name)
        return out
    want = ["f5p", "f6p", "f7p", "f8p", "f9p", "f10p", "f11p", "f12p", "f13p"]
    assert fused(16, 512, 512) == want and fused(64, 512, 512) == want
    assert fused(1, 1280, 1248) == ["f7p", "f8p", "f9p", "f10p", "f11p", "f12p", "f13p"]      # (features.5 / .6 planes are 160 x 156 there: separate launches)
    lib = _lib.load()
    assert lib.sc_irb_supported(64, 384, 64, 32, 32, 1) == 1 and lib.sc_irb_supported(160, 960, 320, 16, 16, 1) == 1
    assert lib.sc_irb_supported(24, 144, 24, 128, 128, 1) == 0            # hidden % 32
    assert lib.sc_irb_supported(32, 192, 64, 64, 64, 2) == 1 and lib.sc_irb_supported(160, 960, 320, 16, 16, 2) == 0


def test_sub_pixel_rules():
    """which decoder conv1 launches run as sub-pixel convolutions (network._use_sp / _use_spd, measured per layer: DESIGN.md 15,
    profiles/r05_bench_sp_b16.txt): forward from 32 output channels and 128 work-groups, the data gradient of the up-sampled channels
    from 128 of them -- or, with <= 64 up-sampled and <= 16 skip channels, as ONE launch with the skip channels' gradient
    (decoder.blocks.3) -- and nothing at odd sizes; pure host logic + the library's host-side helpers"""
    from starcop_amd import _lib, network as nw
    if nw._SP != "1":
        pytest.skip("STARCOP_SP overridden")
    # forward at 16 x 512^2 tiles: decoder.blocks.0 (32^2 output, 256 couts) .. blocks.3 (256^2, 32 couts) yes; blocks.4 (16 couts) no
    assert nw._use_sp(16, 32, 32, 256) and nw._use_sp(16, 64, 64, 128) and nw._use_sp(16, 128, 128, 64) and nw._use_sp(16, 256, 256, 32)
    assert not nw._use_sp(16, 512, 512, 16)
    assert not nw._use_sp(4, 32, 32, 256)            # batch 4: 32 work-groups of 8 waves for decoder.blocks.0
    assert nw._use_sp(64, 32, 32, 256)               # batch 64: 512 (1131 -> 778 us)
    assert not nw._use_sp(16, 33, 64, 64)            # odd output size: no half-resolution source
    # data gradient: 1280 / 256 / 128 up-sampled channels yes, 64 only together with its <= 16 skip channels, 32 no
    assert nw._use_spd(16, 32, 32, 1280, 96) and nw._use_spd(16, 64, 64, 256, 32) and nw._use_spd(16, 128, 128, 128, 24)
    assert nw._use_spd(16, 256, 256, 64, 16) and not nw._use_spd(16, 256, 256, 64, 0) and not nw._use_spd(16, 512, 512, 32, 0)
    # weight gradient as the box-sum GEMM: only where the layer is matrix-bound (decoder.blocks.0: 1280 up-sampled channels)
    assert nw._use_spw(16, 32, 32, 256, 1280) and not nw._use_spw(16, 64, 64, 128, 256) and not nw._use_spw(16, 256, 256, 32, 64)
    lib = _lib.load()
    assert lib.sc_spd_vskip_ok(64, 16) == 1 and lib.sc_spd_vskip_ok(128, 16) == 0 and lib.sc_spd_vskip_ok(64, 24) == 0
    # statistics rows / packed sizes are pure functions of the shape
    assert lib.sc_sp_stat_rows(16, 64, 64, 128) == 16 * 1 * 4                  # 8 x 32 low-resolution tiles
    # 16-wide tiles: 16 rows, or 8 rows when the launch has at most 128 work-groups (decoder.blocks.0 at batch 16: 16 planes x 8 cout tiles)
    assert lib.sc_sp_stat_rows(16, 32, 32, 256) == 32 and lib.sc_sp_stat_rows(64, 32, 32, 256) == 64 and lib.sc_sp_stat_rows(16, 32, 32, 320) == 16
    assert lib.sc_packed_weight_floats_sp(128, 256, 32) == 4 * (16 + 4 * 2) * 2048 * 4
    assert lib.sc_packed_weight_floats_spd(128, 256, 0) == 2 * 4 * 8 * 2048 * 4 and lib.sc_packed_weight_floats_spd(128, 256, 33) == 4 * 4 * 8 * 2048 * 4
