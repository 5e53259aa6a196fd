"""Train-form -> inference-form folding of a FastViT-HD (mci.py `fastvithd`) state dict.

Apple ships the stage-2/3 checkpoints (reference scripts/download_fastvlm.sh:14-22) with the vision tower either already
re-parameterised (`...reparam_conv.weight`, `...lkb_reparam.weight`: what libfastvla_hip.so packs) or in TRAINING form:
multi-branch MobileOne blocks with one BatchNorm per branch, RepMixer as mixer - norm, RepCPE as conv + identity, and the
large-kernel patch embedding as a 7x7 and a 3x3 conv + BN pair.  The reference never sees the difference -- the HF remote code
builds whichever modules the checkpoint has and runs them ([UNVENDORED] mci.py MobileOneBlock / RepMixer / RepCPE /
ReparamLargeKernelConv `.reparameterize()`); this path has one set of kernels, so the algebra of those `reparameterize()`
methods is done here, once, on the host, before fv_load_weights.  Every rule is linear algebra on eval-mode BatchNorm:

    conv + BN        w' = w * g / sqrt(v + eps) (per out channel),  b' = beta - mean * g / sqrt(v + eps)  (+ conv bias scaled)
    MobileOneBlock   sum over rbr_conv.i (k x k) + rbr_scale (1 x 1, zero-padded to k x k) + rbr_skip (BN of the identity kernel)
    RepMixer         w = I + ls * (mixer - norm),  b = ls * (b_mixer - b_norm)        (both sides MobileOne blocks, folded first)
    RepCPE           w = pe.weight + I,  b = pe.bias
    large-kernel     lkb_origin (7x7 conv + BN) + small_conv (3x3 conv + BN, zero-padded to 7x7)

Keys that are already in inference form pass through untouched; ConvFFN's `conv.conv` + `conv.bn` pair stays a pair (the
library folds it when it packs the depthwise weights).  oracle/reparam.py holds the UNFOLDED forward of each block; the
tests check fold(train form) against it (tests/test_reparam.py).
"""
from __future__ import annotations

import re
from typing import Dict, Tuple

import torch

Tensor = torch.Tensor


def _bn_scale_shift(sd: Dict[str, Tensor], pre: str, eps: float) -> Tuple[Tensor, Tensor]:
    g, b = sd[pre + "weight"].float(), sd[pre + "bias"].float()
    m, v = sd[pre + "running_mean"].float(), sd[pre + "running_var"].float()
    s = g / torch.sqrt(v + eps)
    return s, b - m * s


def fuse_conv_bn(w: Tensor, sd: Dict[str, Tensor], bn_pre: str, eps: float, conv_bias: Tensor | None = None) -> Tuple[Tensor, Tensor]:
    s, t = _bn_scale_shift(sd, bn_pre, eps)
    b = t if conv_bias is None else t + conv_bias.float() * s
    return w.float() * s.view(-1, 1, 1, 1), b


def _identity_kernel(out_ch: int, in_per_group: int, k: int) -> Tensor:
    """the k x k kernel of the identity map of a conv with `in_per_group` inputs per group (mci.py MobileOneBlock._fuse_bn_tensor)"""
    w = torch.zeros(out_ch, in_per_group, k, k)
    for i in range(out_ch):
        w[i, i % in_per_group, k // 2, k // 2] = 1.0
    return w


def fold_mobileone(sd: Dict[str, Tensor], pre: str, eps: float = 1e-5) -> Tuple[Tensor, Tensor]:
    """Training-form MobileOneBlock under `pre` -> (kernel, bias) of its single `reparam_conv`."""
    kernel, bias = None, None
    i = 0
    while pre + f"rbr_conv.{i}.conv.weight" in sd:
        w, b = fuse_conv_bn(sd[pre + f"rbr_conv.{i}.conv.weight"], sd, pre + f"rbr_conv.{i}.bn.", eps)
        kernel, bias = (w, b) if kernel is None else (kernel + w, bias + b)
        i += 1
    if pre + "rbr_scale.conv.weight" in sd:
        w, b = fuse_conv_bn(sd[pre + "rbr_scale.conv.weight"], sd, pre + "rbr_scale.bn.", eps)
        if kernel is not None:
            pad = (kernel.shape[-1] - w.shape[-1]) // 2
            w = torch.nn.functional.pad(w, [pad, pad, pad, pad])
        kernel, bias = (w, b) if kernel is None else (kernel + w, bias + b)
    if pre + "rbr_skip.weight" in sd:
        ch = sd[pre + "rbr_skip.weight"].numel()
        if kernel is None:   # a skip-only block (RepMixer's `norm`): depthwise identity; the caller gives it the mixer's size
            raise KeyError(pre + "rbr_skip without a conv branch: fold through fold_repmixer")
        ident = _identity_kernel(ch, kernel.shape[1], kernel.shape[-1])
        s, t = _bn_scale_shift(sd, pre + "rbr_skip.", eps)
        kernel, bias = kernel + ident * s.view(-1, 1, 1, 1), bias + t
    if kernel is None:
        raise KeyError(f"no MobileOne branches under '{pre}'")
    return kernel, bias


def fold_repmixer(sd: Dict[str, Tensor], pre: str, eps: float = 1e-5) -> Tuple[Tensor, Tensor]:
    """RepMixer `x + ls * (mixer(x) - norm(x))` (pre = '...token_mixer.') -> depthwise (kernel, bias)."""
    mw, mb = fold_mobileone(sd, pre + "mixer.", eps)
    ch, k = mw.shape[0], mw.shape[-1]
    ident = _identity_kernel(ch, 1, k)
    s, t = _bn_scale_shift(sd, pre + "norm.rbr_skip.", eps)       # `norm` is a skip-only MobileOne block
    nw, nb = ident * s.view(-1, 1, 1, 1), t
    ls = sd[pre + "layer_scale"].float().reshape(-1)
    return ident + ls.view(-1, 1, 1, 1) * (mw - nw), ls * (mb - nb)


_TRAIN_MARKERS = ("rbr_conv.", "rbr_scale.", "rbr_skip.", ".pe.weight", ".pe.bias", "lkb_origin.", "small_conv.", "token_mixer.mixer.", "token_mixer.norm.")


def is_train_form(sd: Dict[str, Tensor]) -> bool:
    return any(any(m in k for m in _TRAIN_MARKERS) for k in sd)


def fold_train_form(sd: Dict[str, Tensor], eps: float = 1e-5) -> Dict[str, Tensor]:
    """Any mix of training-form and inference-form tower keys -> inference-form keys only (new dict; non-tower keys untouched)."""
    out = {k: v for k, v in sd.items() if not any(m in k for m in _TRAIN_MARKERS) and not k.endswith("token_mixer.layer_scale")}
    done = set()
    for k in sd:
        m = re.match(r"(.*token_mixer\.)(mixer|norm)\.", k)
        if m and m.group(1) not in done:                       # RepMixer (its layer_scale is consumed by the fold)
            pre = m.group(1)
            done.add(pre)
            w, b = fold_repmixer(sd, pre, eps)
            out[pre + "reparam_conv.weight"], out[pre + "reparam_conv.bias"] = w, b
            continue
        m = re.match(r"(.*)\.pe\.weight$", k)
        if m:                                                  # RepCPE: conv + identity
            pre = m.group(1) + "."
            w = sd[k].float()
            out[pre + "reparam_conv.weight"] = w + _identity_kernel(w.shape[0], w.shape[1], w.shape[-1])
            out[pre + "reparam_conv.bias"] = sd[pre + "pe.bias"].float()
            continue
        m = re.match(r"(.*)lkb_origin\.conv\.weight$", k)
        if m:                                                  # large-kernel conv + its small-kernel companion
            pre = m.group(1)
            w, b = fuse_conv_bn(sd[k], sd, pre + "lkb_origin.bn.", eps)
            if pre + "small_conv.conv.weight" in sd:
                sw, sb = fuse_conv_bn(sd[pre + "small_conv.conv.weight"], sd, pre + "small_conv.bn.", eps)
                pad = (w.shape[-1] - sw.shape[-1]) // 2
                w, b = w + torch.nn.functional.pad(sw, [pad, pad, pad, pad]), b + sb
            out[pre + "lkb_reparam.weight"], out[pre + "lkb_reparam.bias"] = w, b
            continue
        m = re.match(r"(.*?)(rbr_conv\.0\.conv\.weight|rbr_scale\.conv\.weight)$", k)
        if m and "token_mixer." not in m.group(1) and m.group(1) not in done:   # a plain MobileOneBlock
            pre = m.group(1)
            done.add(pre)
            w, b = fold_mobileone(sd, pre, eps)
            out[pre + "reparam_conv.weight"], out[pre + "reparam_conv.bias"] = w, b
    # a RepMixer block's own `layer_scale` (the ConvFFN residual scale) sits one level up and must survive
    for k, v in sd.items():
        if k.endswith(".layer_scale") and not k.endswith("token_mixer.layer_scale"):
            out[k] = v
    return out
