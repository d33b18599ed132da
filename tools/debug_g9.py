"""debug: unet_up1 block through the fp32-MFMA kernels, every intermediate vs torch CPU"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, torch.nn.functional as F
import g9_util
import test_gpu_g9 as T
from hip_ops import dev, relerr, conv_mfma, pack, wgrad_mfma
from starcop_amd._lib import *
name = sys.argv[1] if len(sys.argv) > 1 else "unet_up1"
params, x, r = g9_util.case_tensors(name)
(w1, b1), (w2, b2) = params
y1 = F.conv2d(x, w1, padding=1); a1 = F.relu(y1 + b1[None, :, None, None]); y2 = F.conv2d(a1, w2, padding=1)
N, _, H, W = x.shape
srcs = [make_src(dev(x[:, :128]), 128, SRC_RAW), make_src(dev(x[:, 128:]), 64, SRC_RAW)] if name == "unet_up1" else [make_src(dev(x), x.shape[1], SRC_RAW)]
hy1 = T._conv(srcs, dev(w1), N, H, W, 0, 0)
print("y1", relerr(hy1, y1))
c1 = T._bias_relu_cst(b1, ACT_RELU)
hy2 = T._conv([make_src(hy1, 64, SRC_AFFINE, act=ACT_RELU, cst=c1)], dev(w2), N, H, W, 0, 0)
print("y2", relerr(hy2, y2), "max abs diff", float((hy2.cpu() - y2).abs().max()))
d = (hy2.cpu() - y2).abs()
print("y2 bad elements", int((d > 1e-4).sum()), "rows", sorted(set(torch.nonzero(d > 1e-4)[:, 2].tolist()))[:40])
c2 = T._bias_relu_cst(b2, ACT_RELU)
gb, amax = T._bias_grad_and_absmax(dev(r), hy2, c2, ACT_RELU, N, 64, H * W)
gb_ref = (r * ((y2 + b2[None, :, None, None]) > 0)).sum((0, 2, 3))
print("gb1", relerr(gb, gb_ref))
gb_b, _ = T._bias_grad_and_absmax(dev(r), dev(y2), c2, ACT_RELU, N, 64, H * W)
print("gb1 from the CPU y2", relerr(gb_b, gb_ref))
hb = dev(b2)
m_h = (hy2 + hb[None, :, None, None]) > 0
m_c = ((y2 + b2[None, :, None, None]) > 0)
print("mask flips between hy2 and y2:", int((m_h.cpu() != m_c).sum()))
print("gb from torch-on-GPU with hy2:", relerr((dev(r) * m_h).sum((0, 2, 3)), gb_ref))
gb_c, _ = T._bias_grad_and_absmax(dev(r), hy2.clone(), c2, ACT_RELU, N, 64, H * W)
print("gb1 kernel on hy2.clone()", relerr(gb_c, gb_ref), "ptr%4096", hy2.data_ptr() % 4096, hy2.clone().data_ptr() % 4096)
gb_d, _ = T._bias_grad_and_absmax(dev(r), hy2, c2, ACT_RELU, N, 64, H * W)
print("gb1 kernel on hy2 again", relerr(gb_d, gb_ref))
print("per-channel err", ((gb_d.cpu() - gb_ref).abs() / gb_ref.abs().max()).topk(5))
print("hy2 finite", bool(torch.isfinite(hy2).all()), "dtype", hy2.dtype, hy2.shape, hy2.is_contiguous())
