"""Oracle: the UNFOLDED (training-form, eval-mode) forward of the re-parameterisable FastViT blocks.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED ([UNVENDORED] apple ml-fastvlm mobileclip/mci.py; modules MobileOneBlock, RepMixer, RepCPE,
ReparamLargeKernelConv -- their `forward` in training form, with BatchNorm in eval mode).  The product folds these forms into
single convolutions on the host (vla_fastvlm/model/reparam.py); tests/test_reparam.py checks that fold against these."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _bn(x, p, pre, eps):
    return F.batch_norm(x, p[pre + "running_mean"], p[pre + "running_var"], p[pre + "weight"], p[pre + "bias"], training=False, eps=eps)


def mobileone_train(p, pre, x, stride=1, groups=1, eps=1e-5):
    """sum of the k x k conv+BN branches, the 1 x 1 scale branch and the BN-only skip (no activation)."""
    out, i = 0, 0
    while pre + f"rbr_conv.{i}.conv.weight" in p:
        w = p[pre + f"rbr_conv.{i}.conv.weight"]
        out = out + _bn(F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2, groups=groups), p, pre + f"rbr_conv.{i}.bn.", eps)
        i += 1
    if pre + "rbr_scale.conv.weight" in p:
        out = out + _bn(F.conv2d(x, p[pre + "rbr_scale.conv.weight"], None, stride=stride, padding=0, groups=groups), p, pre + "rbr_scale.bn.", eps)
    if pre + "rbr_skip.weight" in p:
        out = out + _bn(x, p, pre + "rbr_skip.", eps)
    return out


def repmixer_train(p, pre, x, eps=1e-5):
    """x + layer_scale * (mixer(x) - norm(x)); pre = '...token_mixer.'"""
    c = x.shape[1]
    return x + p[pre + "layer_scale"].view(1, -1, 1, 1) * (mobileone_train(p, pre + "mixer.", x, groups=c, eps=eps) - _bn(x, p, pre + "norm.rbr_skip.", eps))


def repcpe_train(p, pre, x):
    w = p[pre + "pe.weight"]
    return F.conv2d(x, w, p[pre + "pe.bias"], padding=w.shape[-1] // 2, groups=x.shape[1]) + x


def lkb_train(p, pre, x, stride=2, eps=1e-5):
    w = p[pre + "lkb_origin.conv.weight"]
    g = x.shape[1]
    out = _bn(F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2, groups=g), p, pre + "lkb_origin.bn.", eps)
    if pre + "small_conv.conv.weight" in p:
        sw = p[pre + "small_conv.conv.weight"]
        out = out + _bn(F.conv2d(x, sw, None, stride=stride, padding=sw.shape[-1] // 2, groups=g), p, pre + "small_conv.bn.", eps)
    return out
