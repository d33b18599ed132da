"""Random-shape sweep of the split 3x3 kernels (forward with statistics, backward-data, weight gradient, thin weight gradient; single
source or upsampled + raw concat; ragged sizes) against the fp32-MFMA kernels behind the same C ABI -- once with the host's kernel
choice, once with every two-fp16-term launch routed through the wave-specialised kernel."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("force_ws", [False, True])
def test_split_kernels_random_shapes(force_ws):
    env = dict(os.environ)
    if force_ws:
        env["STARCOP_BX3_WS_MINCHUNKS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_split_kernels.py"), "60", "21" if force_ws else "20"],
                       env=env, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0 and "\n0 problems in" in "\n" + r.stdout, tail
