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


def tower_forward_train_form(p, x, cfg=None, prefix="model.vision_tower.vision_tower.model."):
    """The WHOLE FastViT-HD tower evaluated on a TRAINING-form checkpoint dict (multi-branch MobileOne blocks, RepMixer with its own
    layer scale and norm branch, RepCPE as conv + identity, PatchEmbed as large-kernel + small-kernel branches, every BatchNorm in
    eval mode) -> image embeddings (B, tokens, out_dim), fp32.  Nothing is folded: this is the graph mci.py runs before
    `reparameterize_model`, and the -m gpu checkpoint-interop test holds the engine (which loads the same directory through
    vla_fastvlm/model/reparam.py's fold) against it.  Attention blocks, ConvFFN, SE and the token order are the inference graph's
    (oracle/fastvit_hd.py): they have no training form."""
    from . import fastvit_hd as fv
    cfg = cfg or fv.TowerCfg()
    q = {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}
    c0 = cfg.dims[0]
    x = fv.gelu(mobileone_train(q, "patch_embed.0.", x, stride=2, eps=cfg.bn_eps))
    x = fv.gelu(mobileone_train(q, "patch_embed.1.", x, stride=2, groups=c0, eps=cfg.bn_eps))
    x = fv.gelu(mobileone_train(q, "patch_embed.2.", x, eps=cfg.bn_eps))
    for idx, (kind, i) in enumerate(fv.network_index_map(cfg)):
        c = cfg.dims[i]
        if kind == "cpe":
            x = repcpe_train(q, f"network.{idx}.", x)
        elif kind == "down":
            x = fv.gelu(lkb_train(q, f"network.{idx}.proj.0.", x, stride=2, eps=cfg.bn_eps))
            x = fv.gelu(mobileone_train(q, f"network.{idx}.proj.1.", x, eps=cfg.bn_eps))
        else:
            for j in range(cfg.layers[i]):
                pre = f"network.{idx}.{j}."
                if i in cfg.attn_stages:
                    y = fv._layernorm_channel(x, q[pre + "norm.weight"], q[pre + "norm.bias"], cfg.ln_eps)
                    x = x + q[pre + "layer_scale_1"].view(1, -1, 1, 1) * fv._mhsa(y, q, pre + "token_mixer.", cfg)
                    x = x + q[pre + "layer_scale_2"].view(1, -1, 1, 1) * fv._convffn(x, q, pre + "convffn.", cfg)
                else:
                    x = repmixer_train(q, pre + "token_mixer.", x, eps=cfg.bn_eps)
                    x = x + q[pre + "layer_scale"].view(1, -1, 1, 1) * fv._convffn(x, q, pre + "convffn.", cfg)
    c = cfg.dims[-1]
    x = mobileone_train(q, "conv_exp.", x, groups=c, eps=cfg.bn_eps)
    s = x.mean(dim=(2, 3), keepdim=True)
    s = F.relu(F.conv2d(s, q["conv_exp.se.reduce.weight"], q["conv_exp.se.reduce.bias"]))
    s = torch.sigmoid(F.conv2d(s, q["conv_exp.se.expand.weight"], q["conv_exp.se.expand.bias"]))
    x = fv.gelu(x * s)
    return x.flatten(2).transpose(1, 2).contiguous()
