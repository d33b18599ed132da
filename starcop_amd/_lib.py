"""ctypes binding of libstarcop_hip.so (the C ABI in include/starcop_hip.h).

This is the binding a maintainer of the reference would add.
There is NO fallback: if the shared library is missing, or the current device is
not gfx950, every call raises.  Nothing here imports ``oracle``.
"""
import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libstarcop_hip.so")
CSRC = os.path.join(_HERE, "csrc")

SC_CST = 8
STAT_CONV3, STAT_CONV1, STAT_DW, STAT_STEM, STAT_BNBWD, STAT_CONV1K, STAT_PW3 = 0, 1, 2, 3, 4, 5, 6

SRC_RAW, SRC_AFFINE, SRC_BNBWD, SRC_NORM = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2


class StarcopHipError(RuntimeError):
    pass


class sc_src(C.Structure):
    _fields_ = [("x", C.c_void_p), ("aux", C.c_void_p), ("cst", C.c_void_p),
                ("C", C.c_int32), ("mode", C.c_int32), ("act", C.c_int32), ("up", C.c_int32)]


class sc_conv_args(C.Structure):
    _fields_ = [("src", sc_src * 2), ("nsrc", C.c_int32), ("wpk", C.c_void_p),
                ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cout", C.c_int32),
                ("ks", C.c_int32), ("co_t", C.c_int32),
                ("out0", C.c_void_p), ("out1", C.c_void_p),
                ("csplit", C.c_int32), ("accum0", C.c_int32), ("accum1", C.c_int32),
                ("add0", C.c_void_p), ("add1", C.c_void_p), ("stats", C.c_void_p), ("terms", C.c_int32), ("down0", C.c_int32),
                ("absmax", C.c_void_p), ("xbound", C.c_void_p * 2), ("bnr", C.c_void_p)]


class sc_bn_tail(C.Structure):
    """producer-tail BatchNorm finalize (include/starcop_hip.h: sc_bn_tail)"""
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("momentum", C.c_float), ("eps", C.c_float), ("cst", C.c_void_p), ("act_bound", C.c_void_p), ("tickets", C.c_void_p)]


class sc_bnr_args(C.Structure):
    """BatchNorm-backward sums left by a data-gradient launch (include/starcop_hip.h: sc_bnr_args)"""
    _fields_ = [("y", C.c_void_p), ("cst", C.c_void_p), ("act", C.c_int32), ("rows", C.c_void_p), ("absmax", C.c_void_p)]


class sc_wgrad_args(C.Structure):
    _fields_ = [("dy", sc_src), ("src", sc_src * 2), ("nsrc", C.c_int32),
                ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("Cout", C.c_int32), ("Cin", C.c_int32), ("ks", C.c_int32),
                ("part", C.c_void_p), ("part_floats", C.c_size_t), ("dw", C.c_void_p), ("terms", C.c_int32),
                ("absmax", C.c_void_p), ("xbound", C.c_void_p * 2)]


class sc_irt_args(C.Structure):
    _fields_ = [("x", sc_src), ("w_expand", C.c_void_p), ("w_dw", C.c_void_p), ("cst_expand", C.c_void_p),
                ("N", C.c_int32), ("Cin", C.c_int32), ("hidden", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("stride", C.c_int32)]


class sc_irb_args(C.Structure):
    _fields_ = [("x", sc_src), ("wpk_expand", C.c_void_p), ("cst_expand", C.c_void_p), ("w_dw", C.c_void_p), ("cst_dw", C.c_void_p),
                ("wpk_project", C.c_void_p), ("cst_project", C.c_void_p), ("out", C.c_void_p), ("z_absmax", C.c_void_p),
                ("N", C.c_int32), ("Cin", C.c_int32), ("hidden", C.c_int32), ("Cout", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("stride", C.c_int32), ("residual", C.c_int32)]


class sc_wgrad_pending(C.Structure):
    _fields_ = [("part", C.c_void_p), ("dw", C.c_void_p), ("nparts", C.c_int32), ("taps", C.c_int32), ("Cout", C.c_int32),
                ("Cin", C.c_int32), ("CoP", C.c_int32), ("CiP", C.c_int32), ("total", C.c_uint64)]


class sc_mag1c_args(C.Structure):
    _fields_ = [("x", C.c_void_p), ("x_is_f64", C.c_int32), ("xoff", C.c_void_p),
                ("P", C.c_void_p), ("Ppad", C.c_void_p), ("poff", C.c_void_p), ("statmask", C.c_void_p),
                ("G", C.c_int32), ("S", C.c_int32), ("npix", C.c_int64),
                ("templ", C.c_void_p), ("num_iter", C.c_int32), ("alpha", C.c_double),
                ("cov_update_scaling", C.c_double),
                ("albedo_override", C.c_int32), ("zero_override", C.c_int32),
                ("sparse_override", C.c_int32), ("apply_scaling", C.c_int32),
                ("work", C.c_void_p), ("mf_out", C.c_void_p), ("albedo_out", C.c_void_p),
                ("status", C.c_void_p), ("energy", C.c_void_p), ("logdet", C.c_void_p),
                ("cube", C.c_void_p), ("S_total", C.c_int32), ("band0", C.c_int32), ("pix_index", C.c_void_p),
                ("scatter_mf", C.c_void_p), ("scatter_alb", C.c_void_p), ("scatter_is_f64", C.c_int32)]


_vp, _i, _f, _d, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); every symbol include/starcop_hip.h declares
SIGNATURES = {
    "sc_last_error": (C.c_char_p, []),
    "sc_version": (_i, []),
    "sc_device_check": (_i, []),
    "sc_pack_weights": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sc_packed_weight_floats": (_sz, [_i, _i, _i, _i, _i]),
    "sc_pack_work_items": (_sz, [_i, _i, _i, _i, _i, _i]),
    "sc_pack_weights_batch": (_i, [_vp, _vp, _i, C.c_uint32, _vp]),
    "sc_conv2d_mfma": (_i, [C.POINTER(sc_conv_args), _vp]),
    "sc_conv1x1_ksplit": (_i, [C.POINTER(sc_conv_args), _vp]),
    "sc_wgrad_bx3_workspace_floats": (_sz, [_i, _i, _i, _i, _i]),
    "sc_conv3x3_wgrad_bx3": (_i, [C.POINTER(sc_wgrad_args), _vp]),
    "sc_pack_weights_bx3": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sc_packed_weight_floats_bx3": (_sz, [_i, _i, _i, _i, _i]),
    "sc_conv3x3_bx3": (_i, [C.POINTER(sc_conv_args), _vp]),
    "sc_wgrad_workspace_floats": (_sz, [_i, _i, _i, _i, _i, _i]),
    "sc_conv2d_wgrad_mfma": (_i, [C.POINTER(sc_wgrad_args), _vp]),
    "sc_dwconv3x3_fwd": (_i, [C.POINTER(sc_src), _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "sc_dwconv3x3_fwd_bn": (_i, [C.POINTER(sc_src), _vp, _vp, _i, _i, _i, _i, _i, _vp, C.POINTER(sc_bn_tail), _vp]),
    "sc_dwconv3x3_dgrad": (_i, [C.POINTER(sc_src), _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "sc_dwconv3x3_wgrad": (_i, [C.POINTER(sc_src), C.POINTER(sc_src), _vp, _i, _i, _i, _i, _i, _vp]),
    "sc_cast_f64_f32": (_i, [_vp, _vp, _sz, _vp]),
    "sc_cast_f64_f32_batch": (_i, [_vp, _i, _vp]),
    "sc_stem_conv_fwd": (_i, [C.POINTER(sc_src), _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sc_stem_wgrad_workspace_floats": (_sz, [_i, _i, _i, _i]),
    "sc_stem_conv_wgrad": (_i, [C.POINTER(sc_src), C.POINTER(sc_src), _vp, _sz, _vp, _i, _i, _i, _i, _vp]),
    "sc_head_conv_fwd": (_i, [C.POINTER(sc_src), _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sc_head_conv_dgrad": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sc_head_wgrad_workspace_floats": (_sz, [_i, _i, _i, _i]),
    "sc_head_conv_wgrad": (_i, [_vp, C.POINTER(sc_src), _vp, _sz, _vp, _vp, _i, _i, _i, _i, _vp]),
    "sc_head_conv_bwd": (_i, [_vp, C.POINTER(sc_src), _vp, _vp, _vp, _sz, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "sc_head_bwd_bn_rows": (_i, [_i, _i, _i]),
    "sc_stat_rows": (_i, [_i, _i, _i, _i]),
    "sc_bn_finalize": (_i, [_vp, _i, _d, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _i, _vp, _vp, _vp]),
    "sc_bn_bwd_reduce": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "sc_bn_bwd_small": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sc_bn_bwd_finalize": (_i, [_vp, _i, _d, _vp, _vp, _vp, _vp, _i, _vp]),
    "sc_bn_bwd_finalize_rows32": (_i, [_vp, _i, _d, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "sc_add_srcs": (_i, [C.POINTER(sc_src), C.POINTER(sc_src), _vp, _i, _i, _i, _vp]),
    "sc_stream_wait_stream": (_i, [_vp, _vp]),
    "sc_add_srcs_absmax": (_i, [C.POINTER(sc_src), C.POINTER(sc_src), _vp, _i, _i, _i, _vp, _vp]),
    "sc_apply_src": (_i, [C.POINTER(sc_src), _vp, _i, _i, _i, _vp]),
    "sc_downsum2x2": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sc_fill_f64": (_i, [_vp, _d, _sz, _vp]),
    "sc_bce_logits_weighted": (_i, [_vp, _vp, _vp, _f, _sz, _vp, _vp, _vp, _vp]),
    "sc_adam_step": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _f, _f, _f, _f, _vp, _vp]),
    "sc_adam_prepare": (_i, [_vp, _vp, _f, _f, _vp, _vp]),
    "sc_threshold_masks": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "sc_pred_classification": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sc_mag1c_workspace_doubles": (_sz, [_i, _i, C.c_int64]),
    "sc_mag1c_groups": (_i, [C.POINTER(sc_mag1c_args), _vp]),
    "sc_mag1c_pack": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "sc_valid_mask": (_i, [_vp, _i, _i, _i, _i, _d, C.c_int64, _vp, _vp]),
    "sc_scatter": (_i, [_vp, _i, _vp, _sz, _vp, _i, _vp]),
    "sc_scatter_n": (_i, [_vp, _i, _vp, _vp, _sz, _vp, _i, _vp]),
    "sc_valid_mask_ne": (_i, [_vp, _i, _i, _i, _i, _d, C.c_int64, _vp, _vp]),
    "sc_mag1c_layout_columns": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sc_mag1c_layout_ids_workspace_ints": (_sz, [C.c_int64, _i]),
    "sc_mag1c_layout_ids": (_i, [_vp, _vp, C.c_int64, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sc_trimmed_sum_workspace_bytes": (_sz, [_i]),
    "sc_trimmed_sums": (_i, [_vp, _i, _sz, _d, _vp, _vp, _sz, _vp]),
    "sc_band_ratio": (_i, [_vp, _vp, _vp, _i, _sz, _vp, _vp, _f, _f, _vp]),
    "sc_clip_scale": (_i, [_vp, _vp, _sz, _f, _f, _f, _f, _i, _vp]),
    "sc_packed_weight_floats_thin16": (_sz, [_i, _i, _i]),
    "sc_pack_weights_thin16": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "sc_conv3x3_thin16": (_i, [C.POINTER(sc_conv_args), _vp]),
    "sc_packed_weight_floats_sp": (_sz, [_i, _i, _i]),
    "sc_pack_weights_sp": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sc_sp_stat_rows": (_i, [_i, _i, _i, _i]),
    "sc_conv3x3_sp": (_i, [C.POINTER(sc_conv_args), _vp]),
    "sc_packed_weight_floats_spd": (_sz, [_i, _i, _i]),
    "sc_pack_weights_spd": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sc_spd_vskip_ok": (_i, [_i, _i]),
    "sc_conv3x3_sp_dgrad": (_i, [C.POINTER(sc_conv_args), _vp]),
    "sc_sp_wgrad_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "sc_conv3x3_sp_wgrad": (_i, [C.POINTER(sc_wgrad_args), _vp, _sz, _vp]),
    "sc_wgrad_scatter_cols": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sc_binary_opening": (_i, [_vp, _f, _i, _vp, _vp, _i, _i, _i, _vp]),
    "sc_threshold_confusion": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, _i, _vp]),
    "sc_gather_augment": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sc_dwconv3x3_bwd_fused": (_i, [C.POINTER(sc_src), C.POINTER(sc_src), _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sc_wgrad_thin16_workspace_floats": (_sz, [_i, _i, _i, _i, _i]),
    "sc_conv3x3_wgrad_thin16": (_i, [C.POINTER(sc_wgrad_args), _vp]),
    "sc_conv2d_wgrad_mfma_deferred": (_i, [C.POINTER(sc_wgrad_args), C.POINTER(sc_wgrad_pending), _vp]),
    "sc_wgrad_reduce_batch": (_i, [_vp, _vp, _i, C.c_uint32, _vp]),
    "sc_packed_weight_floats_pw3": (_sz, [_i, _i, _i]),
    "sc_conv1x1_pw3": (_i, [C.POINTER(sc_conv_args), _vp]),
    "sc_wgrad_pw3_workspace_floats": (_sz, [_i, _i, _i, _i, _i]),
    "sc_conv1x1_wgrad_pw3": (_i, [C.POINTER(sc_wgrad_args), C.POINTER(sc_wgrad_pending), _vp]),
    "sc_irb_supported": (_i, [_i, _i, _i, _i, _i, _i]),
    "sc_irb_eval": (_i, [C.POINTER(sc_irb_args), _vp]),
    "sc_irt_supported": (_i, [_i, _i, _i, _i, _i]),
    "sc_irt_rows": (_i, [_i, _i, _i, _i, _i]),
    "sc_irt_bwd_rows": (_i, [_i, _i, _i, _i]),
    "sc_irt_bwd_workspace_floats": (_sz, [_i, _i, _i, _i, _i]),
    "sc_irt_expand_stats": (_i, [C.POINTER(sc_irt_args), _vp, _vp]),
    "sc_irt_fwd": (_i, [C.POINTER(sc_irt_args), _vp, _vp, _vp]),
    "sc_irt_bwd": (_i, [C.POINTER(sc_irt_args), C.POINTER(sc_src), _vp, _vp, _vp, _vp]),
    "sc_irt_xmoments": (_i, [C.POINTER(sc_irt_args), _vp, _vp]),
    "sc_irt_bwd_fix": (_i, [C.POINTER(sc_irt_args), _vp, _vp, _vp, _vp, _i, _vp]),
    "sc_irt_wgrad_finalize": (_i, [C.POINTER(sc_irt_args), _vp, _vp, _vp, _vp]),
    "sc_maxpool2x2": (_i, [C.POINTER(sc_src), _vp, _i, _i, _i, _i, _vp]),
    "sc_upsample_bilinear2x": (_i, [C.POINTER(sc_src), _vp, _i, _i, _i, _i, _vp]),
    "sc_maxpool2x2_bwd": (_i, [C.POINTER(sc_src), _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "sc_upsample_bilinear2x_bwd": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "sc_tiff_lzw_decode": (_i, [_vp, _sz, _vp, _sz, C.POINTER(C.c_size_t)]),
    "sc_tiff_unpredict": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp]),
}
PACK_SPD = 8           # ... and of its data gradient sc_conv3x3_sp_dgrad
PACK_SP = 7            # sc_pack_desc.bx3 code of the phase-filter layout of sc_conv3x3_sp
PACK_PW3 = 6           # sc_pack_desc.bx3 code of the pointwise layout of sc_conv1x1_pw3
PACK_THIN16 = 5        # sc_pack_desc.bx3 code of the register layout of sc_conv3x3_thin16
TERMS_F16X2 = 4        # `terms` code of the two-fp16-term kernels (include/starcop_hip.h SC_TERMS_F16X2)
SE_CROSS = 0xBA        # the 3x3 cross of starcop/baselines.py:39-41 as sc_binary_opening's se_bits

_lib = None


def build(force=False):
    """Compile every HIP source for gfx950 into starcop_amd/libstarcop_hip.so (hipcc; no GPU needed)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run(["make", "-C", CSRC, "-j", "4"], capture_output=True, text=True)
    if r.returncode != 0:
        raise StarcopHipError("building libstarcop_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


def load():
    """dlopen the library and declare every entry point.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("STARCOP_HIP_LIB", LIB_PATH)        # development knob: an experiment build of the same sources
    if not os.path.exists(path):
        raise StarcopHipError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C starcop_amd/csrc`).  starcop_amd has no CPU/torch fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_EXC = {-1: ValueError, -2: RuntimeError, -3: torch.linalg.LinAlgError, -4: RuntimeError}


def check(rc):
    """Map negative C status codes to the exception types the reference raises."""
    if rc != 0:
        msg = load().sc_last_error().decode("utf-8", "replace")
        raise _EXC.get(rc, StarcopHipError)(f"libstarcop_hip: {msg} (status {rc})")


_device_ok = {}


def require_device(t=None):
    """The product path runs on a gfx950 GPU only; anything else is a loud error."""
    if not torch.cuda.is_available():
        raise StarcopHipError("starcop_amd needs a ROCm GPU (gfx950); torch.cuda.is_available() is False")
    if t is not None and not t.is_cuda:
        raise StarcopHipError(f"starcop_amd kernels need device tensors, got a tensor on {t.device}")
    dev = torch.cuda.current_device()
    if dev not in _device_ok:
        check(load().sc_device_check())
        _device_ok[dev] = True


def tiff_lzw_decode(buf: bytes, n_out: int) -> bytes:
    """host: TIFF LZW stream -> at most n_out bytes (sc_tiff_lzw_decode)"""
    out = C.create_string_buffer(n_out)
    written = C.c_size_t(0)
    src = (C.c_char * len(buf)).from_buffer_copy(buf)
    if load().sc_tiff_lzw_decode(C.cast(src, C.c_void_p), len(buf), C.cast(out, C.c_void_p), n_out, C.byref(written)) != 0:
        raise ValueError("corrupt TIFF LZW stream")
    return out.raw[:written.value]


def tiff_unpredict(a, predictor, rows, cols, spp, bps, big_endian):
    """host: undo TIFF predictor 2 / 3 on a decoded block (uint8 numpy array) -> uint8 array of little-endian samples"""
    import numpy as np
    a = np.ascontiguousarray(a, dtype=np.uint8)
    out = np.empty_like(a)
    if load().sc_tiff_unpredict(a.ctypes.data_as(C.c_void_p), predictor, rows, cols, spp, bps, int(bool(big_endian)),
                                out.ctypes.data_as(C.c_void_p)) != 0:
        raise ValueError(f"TIFF predictor {predictor} with {bps}-byte samples is not supported")
    return out


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def make_src(x, C_, mode=SRC_RAW, act=ACT_NONE, up=0, cst=None, aux=None):
    s = sc_src()
    s.x = x.data_ptr() if x is not None else None
    s.aux = aux.data_ptr() if aux is not None else None
    s.cst = cst.data_ptr() if cst is not None else None
    s.C, s.mode, s.act, s.up = int(C_), int(mode), int(act), int(up)
    return s
