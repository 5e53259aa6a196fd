"""Oracle: image canonicalisation + letterbox.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates src/vla_fastvlm/model/fastvlm_adapter.py:
  resize_with_pad           :36-55   (ratio=max(W/S,H/S); int() truncation; bilinear align_corners=False;
                                      pad LEFT and TOP with pad_value)
  _normalize_channels       :444-449 (gray -> 3ch repeat, >3ch -> first 3)
  _resize_image             :451-461
  _prepare_images_tensor    :479-488
  _maybe_normalize_imagenet :463-477 (after the letterbox: pad pixels are normalised too)

The bilinear resampling is written out explicitly (no F.interpolate) so that it is an independent statement of
``torch.nn.functional.interpolate(mode="bilinear", align_corners=False, antialias=False)``:
  src = (dst + 0.5) * (in/out) - 0.5, clamped at 0; i0 = floor(src); i1 = min(i0+1, in-1); w1 = src - i0.
"""
from __future__ import annotations

import torch


def letterbox_geometry(h_in: int, w_in: int, size: int):
    """Return (resized_h, resized_w, pad_top, pad_left) exactly as fastvlm_adapter.py:44-51 computes them."""
    ratio = max(w_in / size, h_in / size)
    rh = int(h_in / ratio)
    rw = int(w_in / ratio)
    return rh, rw, max(0, int(size - rh)), max(0, int(size - rw))


def _axis_taps(n_in: int, n_out: int, dtype=torch.float32):
    # ATen area_pixel_compute_source_index(scale=in/out, align_corners=False, cubic=False)
    scale = n_in / n_out
    dst = torch.arange(n_out, dtype=dtype)
    src = (dst + 0.5) * scale - 0.5
    src = torch.clamp(src, min=0.0)
    i0 = torch.floor(src).to(torch.int64)
    i0 = torch.clamp(i0, max=n_in - 1)
    i1 = torch.clamp(i0 + 1, max=n_in - 1)
    w1 = src - i0.to(dtype)
    return i0, i1, w1


def bilinear_resize(x: torch.Tensor, out_h: int, out_w: int) -> torch.Tensor:
    """(B,C,H,W) fp32 -> (B,C,out_h,out_w), align_corners=False, no antialias."""
    _, _, h, w = x.shape
    y0, y1, wy = _axis_taps(h, out_h, x.dtype)
    x0, x1, wx = _axis_taps(w, out_w, x.dtype)
    top = x[:, :, y0, :]
    bot = x[:, :, y1, :]
    wy = wy.view(1, 1, -1, 1)
    wx = wx.view(1, 1, 1, -1)
    # same association as ATen: h0*(w0*p00 + w1*p01) + h1*(w0*p10 + w1*p11)
    t = top[:, :, :, x0] * (1 - wx) + top[:, :, :, x1] * wx
    b = bot[:, :, :, x0] * (1 - wx) + bot[:, :, :, x1] * wx
    return t * (1 - wy) + b * wy


def normalize_channels(x: torch.Tensor) -> torch.Tensor:
    if x.shape[1] == 1:
        return x.repeat(1, 3, 1, 1)
    if x.shape[1] > 3:
        return x[:, :3]
    return x


def letterbox(x: torch.Tensor, size: int, pad_value: float = 0.0, resize_with_padding: bool = True) -> torch.Tensor:
    """(B,C,H,W) any float -> (B,3,size,size) fp32, the tensor the reference hands to the VLM."""
    if x.ndim != 4:
        raise ValueError(f"(B,C,H,W) expected, but got shape {tuple(x.shape)}")
    x = normalize_channels(x.to(torch.float32))
    _, _, h, w = x.shape
    if not resize_with_padding:
        if (h, w) != (size, size):
            x = bilinear_resize(x, size, size)
        return x
    rh, rw, pt, pl = letterbox_geometry(h, w, size)
    r = bilinear_resize(x, rh, rw)
    out = torch.full((x.shape[0], 3, max(size, rh), max(size, rw)), float(pad_value), dtype=torch.float32)
    out[:, :, pt:pt + rh, pl:pl + rw] = r
    return out


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def normalize_imagenet(x: torch.Tensor, torchvision_branch: bool = True) -> torch.Tensor:
    """fastvlm_adapter.py:463-477 on the letterboxed (B,3,S,S) fp32 tensor.  torchvision_branch=True (:471-477, what an installed reference runs --
    torchvision is a declared dependency, pyproject.toml:33): `if x.max() > 1.5: x = x / 255.0` over the WHOLE batch tensor, then TF.normalize =
    x.sub_(mean[:, None, None]).div_(std[:, None, None]) in the tensor's dtype.  False (:466-470, the branch without torchvision, the one the golden
    vectors of tests/golden/g1_normalize.npz were produced by): (x - mean) / std alone."""
    x = x.to(torch.float32)
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32).view(1, 3, 1, 1)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32).view(1, 3, 1, 1)
    if torchvision_branch and float(x.max()) > 1.5:
        x = x / 255.0
    return (x - mean) / std


def prepare_images(x: torch.Tensor, size: int, pad_value: float = 0.0, resize_with_padding: bool = True, normalize: bool = False,
                   torchvision_branch: bool = True) -> torch.Tensor:
    """_prepare_images_tensor :479-488: letterbox, then the optional normalisation."""
    y = letterbox(x, size, pad_value, resize_with_padding)
    return normalize_imagenet(y, torchvision_branch) if normalize else y
