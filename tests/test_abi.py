"""The C-ABI library loads on a CPU-only box and exports every symbol include/starcop_hip.h declares
(no compute calls here -- those are the -m gpu tests)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "starcop_hip.h")


def declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sc_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    from starcop_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_header_symbols_are_exported(lib):
    from starcop_amd import _lib
    names = declared_functions()
    assert len(names) >= 35
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in starcop_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in starcop_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)


def test_pure_host_entry_points(lib):
    assert lib.sc_version() >= 100
    assert lib.sc_packed_weight_floats(256, 1376, 3, 64, 0) == 4 * 1376 * 9 * 64
    assert lib.sc_packed_weight_floats(256, 1376, 3, 64, 1) == 22 * 256 * 9 * 64      # dgrad: 1376 -> 22 tiles of 64
    assert lib.sc_wgrad_workspace_floats(16, 512, 512, 16, 32, 3) > 0
    assert lib.sc_mag1c_workspace_doubles(4, 125, 2048) == 4 * 125 * 125 + 3 * 2048
    # thin-layer packs: 16-byte entries x 64 lanes x 2 terms per K step; the forward filter of a 32-channel layer carries the 16 phase
    # steps of its sub-pixel form behind the nine 3x3 ones, the transposed pack of that layer is the 16 entries of its half-resolution
    # data gradient (8 K steps x 2 row blocks)
    assert lib.sc_packed_weight_floats_thin16(16, 16, 0) == 5 * 512 and lib.sc_packed_weight_floats_thin16(16, 16, 1) == 5 * 512
    assert lib.sc_packed_weight_floats_thin16(16, 32, 0) == (9 + 16) * 512 and lib.sc_packed_weight_floats_thin16(16, 32, 1) == 16 * 512
    assert lib.sc_pack_work_items(16, 32, 3, 16, 0, 5) == (9 + 16) * 512 and lib.sc_pack_work_items(16, 32, 3, 16, 1, 5) == 16 * 512


def test_struct_layouts_match_header():
    from starcop_amd import _lib
    assert ctypes.sizeof(_lib.sc_src) == 40
    assert ctypes.sizeof(_lib.sc_conv_args) == 2 * 40 + 8 + 8 + 6 * 4 + 2 * 8 + 3 * 4 + 4 + 3 * 8 + 8 + 8 + 2 * 8 + 8     # ..., add0, add1, stats, terms, down0, absmax, xbound[2], bnr
    assert ctypes.sizeof(_lib.sc_bnr_args) == 2 * 8 + 8 + 2 * 8     # y, cst, act (+ padding), rows, absmax


def test_struct_sizes_match_the_c_compiler(tmp_path):
    """sizeof / offsetof of every struct in include/starcop_hip.h as gcc lays them out == the ctypes mirrors in _lib.py"""
    import shutil
    import subprocess
    from starcop_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "starcop_hip.h"\nint main(void){'
                   'printf("%zu %zu %zu %zu %zu %zu %zu\\n", sizeof(sc_src), sizeof(sc_conv_args), offsetof(sc_conv_args, terms), '
                   'sizeof(sc_wgrad_args), offsetof(sc_wgrad_args, terms), sizeof(sc_pack_desc), offsetof(sc_pack_desc, total));'
                   'printf("%zu %zu\\n", sizeof(sc_wgrad_pending), offsetof(sc_wgrad_pending, total));'
                   'printf("%zu %zu %zu %zu %zu\\n", sizeof(sc_irb_args), offsetof(sc_irb_args, out), offsetof(sc_irb_args, N), offsetof(sc_irb_args, residual), '
                   'sizeof(sc_irt_args));return 0;}\n')
    exe = tmp_path / "sz"
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.run(["gcc", "-I", inc, str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    want = [ctypes.sizeof(_lib.sc_src), ctypes.sizeof(_lib.sc_conv_args), _lib.sc_conv_args.terms.offset,
            ctypes.sizeof(_lib.sc_wgrad_args), _lib.sc_wgrad_args.terms.offset, 48, 40,
            ctypes.sizeof(_lib.sc_wgrad_pending), _lib.sc_wgrad_pending.total.offset,
            ctypes.sizeof(_lib.sc_irb_args), _lib.sc_irb_args.out.offset, _lib.sc_irb_args.N.offset, _lib.sc_irb_args.residual.offset,
            ctypes.sizeof(_lib.sc_irt_args)]
    assert got == want


def test_no_cpu_fallback():
    """The product path fails loudly without a gfx950 device; it never routes to torch CPU ops or the oracle."""
    import torch
    from starcop_amd import _lib, model_module as mm
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    model = mm.ModelModule(mm.default_settings())
    with pytest.raises(_lib.StarcopHipError):
        model(torch.zeros(1, 4, 64, 64))
    with pytest.raises(_lib.StarcopHipError):
        model.normalizer.normalize_x(torch.zeros(1, 4, 8, 8))
    import starcop_amd.mag1c as m1
    with pytest.raises(_lib.StarcopHipError):
        m1.acrwl1mf(torch.zeros(1, 32, 8), torch.ones(8))
    src = "".join(open(os.path.join(ROOT, "starcop_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "starcop_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src
