"""Run the whole-network GPU parity tests with the split-bf16 kernels switched off (fp32-MFMA kernels, up_tmp + downsum path)."""
import sys, pytest
sys.path.insert(0, ".")
from starcop_amd.network import HyperStarcopUNet
HyperStarcopUNet.split_bf16 = False
sys.exit(pytest.main(["tests/test_gpu_unet.py", "-m", "gpu", "-x", "-q", "-k", "eval or train_forward or fused_train or predict"]))
