"""``predict()`` surface: whole-image inference with reflect padding to a multiple of 32.

Behavioural contract follows /root/reference/starcop/models/utils/padding.py:5-50 (``find_padding``:
pad_1 = total // 2, pad_2 = total - pad_1; ``padded_predict``: reflect pad, forward under no_grad, crop,
return numpy).  Here the pad and the crop happen on the device that runs the network (one H2D copy of the
raw tile, one D2H copy of the cropped prediction) instead of in numpy on the host.
"""
import numpy as np
import torch
import torch.nn.functional as F


def find_padding(v, divisor=8):
    """(before, after) padding so that ``v`` becomes a non-zero multiple of ``divisor``."""
    target = max(divisor, -(-int(v) // divisor) * divisor)
    before = (target - int(v)) // 2
    return before, target - int(v) - before


def padded_predict(tensor, model, divisor=32, device=torch.device("cpu")):
    """tensor: array-like (C, H, W) -> np.ndarray (K, H, W) or (H, W), as the reference function."""
    if len(tensor.shape) != 3:
        raise AssertionError(f"Expected 3D tensor, found {len(tensor.shape)}D tensor")
    rows, cols = tensor.shape[-2], tensor.shape[-1]
    (top, bottom), (left, right) = find_padding(rows, divisor), find_padding(cols, divisor)
    x = torch.as_tensor(np.ascontiguousarray(tensor)).to(device)[None]      # a strided host view makes the H2D copy 10x slower
    with torch.no_grad():
        if top or bottom or left or right:
            if max(top, bottom) >= rows or max(left, right) >= cols:
                raise ValueError("padded_predict: reflect padding needs the image to be larger than the pad")
            x = F.pad(x, (left, right, top, bottom), mode="reflect")
        out = model(x)[0]
        if out.dim() == 3:
            out = out[:, top:top + rows, left:left + cols]
        elif out.dim() == 2:
            out = out[top:top + rows, left:left + cols]
        else:
            raise NotImplementedError(f"Don't know how to slice the tensor of shape {out.shape}")
    return out.cpu().numpy()
