"""Oracle: FastViT-HD vision tower + mlp2x_gelu projector, inference-mode graph.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/__init__.py): the reference delegates this arithmetic to HF remote code
(`apple/FastVLM-0.5B:llava_qwen.py` -> mobileclip/mci.py `fastvithd`), reached from
src/vla_fastvlm/model/fastvlm_adapter.py:183-191 (load) and :533 (call).  This file restates the published
FastViT / FastViT-HD architecture in its re-parameterised (inference) form:

  stem          3x MobileOneBlock: conv3x3 s2 (3->C0) + GELU ; dw3x3 s2 + GELU ; conv1x1 + GELU
  stage i       RepMixerBlock (i<3):  x = dw3x3(x)                       (RepMixer, identity+BN folded)
                                      x = x + ls * fc2(GELU(fc1(BN(dw7x7(x)))))      (ConvFFN)
                AttentionBlock (i>=3): x = x + ls1 * proj(MHSA(LayerNormChannel(x)))   head_dim 32, no qkv bias
                                      x = x + ls2 * ConvFFN(x)
  PatchEmbed    grouped 7x7 s2 (groups=Cin, Cin->2Cin, o <- o//2) + GELU ; conv1x1 + GELU
  RepCPE        dw7x7 (+bias, identity folded) in front of the two attention stages
  conv_exp      dw3x3 (groups=C4, C4->2*C4) -> SE(rd 1/16: avgpool, 1x1+ReLU, 1x1+sigmoid, scale) -> GELU
  tokens        (B, 2*C4, S/64, S/64) -> (B, (S/64)^2, 2*C4)
  projector     Linear(2*C4 -> H) + GELU + Linear(H -> H), with bias   ([site] fast_vlm/modeling_fast_vlm.py:39-56)

Parameter names follow the Apple checkpoint convention under the prefix VT ("...vision_tower.model.").
BatchNorm of ConvFFN is kept UNFOLDED here (eval-mode running statistics); the product folds it at pack time, so the
comparison also checks the folding.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

VT = "model.vision_tower.vision_tower.model."
PROJ = "model.mm_projector."


@dataclass(frozen=True)
class TowerCfg:
    layers: Tuple[int, ...] = (2, 12, 24, 4, 2)
    dims: Tuple[int, ...] = (96, 192, 384, 768, 1536)
    mlp_ratio: int = 4
    head_dim: int = 32
    attn_stages: Tuple[int, ...] = (3, 4)
    se_ratio: float = 0.0625
    cls_ratio: float = 2.0
    ln_eps: float = 1e-5
    bn_eps: float = 1e-5

    @property
    def out_dim(self) -> int:
        return int(self.dims[-1] * self.cls_ratio)


def gelu(x):
    return F.gelu(x)  # exact erf form (nn.GELU default)


def _conv(x, p, name, stride=1, padding=0, groups=1):
    return F.conv2d(x, p[name + ".weight"], p.get(name + ".bias"), stride=stride, padding=padding, groups=groups)


def _convffn(x, p, pre, cfg: TowerCfg):
    c = x.shape[1]
    y = F.conv2d(x, p[pre + "conv.conv.weight"], None, padding=3, groups=c)
    y = F.batch_norm(y, p[pre + "conv.bn.running_mean"], p[pre + "conv.bn.running_var"],
                     p[pre + "conv.bn.weight"], p[pre + "conv.bn.bias"], training=False, eps=cfg.bn_eps)
    y = gelu(_conv(y, p, pre + "fc1"))
    return _conv(y, p, pre + "fc2")


def _layernorm_channel(x, w, b, eps):
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[None, :, None, None] * x + b[None, :, None, None]


def _mhsa(x, p, pre, cfg: TowerCfg):
    b, c, h, w = x.shape
    n = h * w
    nh = c // cfg.head_dim
    t = x.flatten(2).transpose(1, 2)  # (B,N,C)
    qkv = F.linear(t, p[pre + "qkv.weight"]).reshape(b, n, 3, nh, cfg.head_dim).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q * cfg.head_dim ** -0.5) @ k.transpose(-2, -1)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(b, n, c)
    o = F.linear(o, p[pre + "proj.weight"], p[pre + "proj.bias"])
    return o.transpose(1, 2).reshape(b, c, h, w)


def network_index_map(cfg: TowerCfg):
    """mci.py FastViT.__init__ ordering: [RepCPE?] stage [PatchEmbed] ...  -> list of (kind, stage)."""
    out = []
    n = len(cfg.layers)
    for i in range(n):
        if i in cfg.attn_stages:
            out.append(("cpe", i))
        out.append(("stage", i))
        if i < n - 1:
            out.append(("down", i))
    return out


def tower_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, cfg: TowerCfg = TowerCfg(), prefix: str = VT,
                  taps: dict | None = None) -> torch.Tensor:
    """x: (B,3,S,S) fp32, S % 64 == 0  ->  image embeddings (B, (S/64)^2, out_dim)."""
    q = {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}
    c0 = cfg.dims[0]
    x = gelu(_conv(x, q, "patch_embed.0.reparam_conv", stride=2, padding=1))
    if taps is not None:
        taps["stem0"] = x
    x = gelu(_conv(x, q, "patch_embed.1.reparam_conv", stride=2, padding=1, groups=c0))
    x = gelu(_conv(x, q, "patch_embed.2.reparam_conv"))
    if taps is not None:
        taps["stem"] = x
    for idx, (kind, i) in enumerate(network_index_map(cfg)):
        c = cfg.dims[i]
        if kind == "cpe":
            x = _conv(x, q, f"network.{idx}.reparam_conv", padding=3, groups=c)
        elif kind == "down":
            x = gelu(_conv(x, q, f"network.{idx}.proj.0.lkb_reparam", stride=2, padding=3, groups=c))
            x = gelu(_conv(x, q, f"network.{idx}.proj.1.reparam_conv"))
        else:
            for j in range(cfg.layers[i]):
                pre = f"network.{idx}.{j}."
                if i in cfg.attn_stages:
                    y = _layernorm_channel(x, q[pre + "norm.weight"], q[pre + "norm.bias"], cfg.ln_eps)
                    x = x + q[pre + "layer_scale_1"].view(1, -1, 1, 1) * _mhsa(y, q, pre + "token_mixer.", cfg)
                    x = x + q[pre + "layer_scale_2"].view(1, -1, 1, 1) * _convffn(x, q, pre + "convffn.", cfg)
                else:
                    x = _conv(x, q, pre + "token_mixer.reparam_conv", padding=1, groups=c)
                    x = x + q[pre + "layer_scale"].view(1, -1, 1, 1) * _convffn(x, q, pre + "convffn.", cfg)
            if taps is not None:
                taps[f"stage{i}"] = x
    c = cfg.dims[-1]
    x = _conv(x, q, "conv_exp.reparam_conv", padding=1, groups=c)
    s = x.mean(dim=(2, 3), keepdim=True)
    s = F.relu(_conv(s, q, "conv_exp.se.reduce"))
    s = torch.sigmoid(_conv(s, q, "conv_exp.se.expand"))
    x = gelu(x * s)
    return x.flatten(2).transpose(1, 2).contiguous()


def projector_forward(p: Dict[str, torch.Tensor], tokens: torch.Tensor, prefix: str = PROJ) -> torch.Tensor:
    h = gelu(F.linear(tokens, p[prefix + "0.weight"], p[prefix + "0.bias"]))
    return F.linear(h, p[prefix + "2.weight"], p[prefix + "2.bias"])


def tower_flops_per_image(cfg: TowerCfg, size: int) -> dict:
    """Algorithmic multiply-accumulate FLOPs (2/MAC), split into MFMA-shaped (dense contractions) and VALU-shaped
    (depthwise/grouped taps).  Used by bench.py for the roofline numerator."""
    dense = 0
    dw = 0
    h = size // 2
    c0 = cfg.dims[0]
    dense += 2 * h * h * 27 * c0
    h //= 2
    dw += 2 * h * h * 9 * c0
    dense += 2 * h * h * c0 * c0
    n = len(cfg.layers)
    for i in range(n):
        c = cfg.dims[i]
        px = h * h
        if i in cfg.attn_stages:
            dw += 2 * px * 49 * c
        for _ in range(cfg.layers[i]):
            if i in cfg.attn_stages:
                dense += 2 * px * c * 3 * c + 2 * px * c * c + 4 * px * px * c
            else:
                dw += 2 * px * 9 * c
            dw += 2 * px * 49 * c
            dense += 2 * 2 * px * c * c * cfg.mlp_ratio
        if i < n - 1:
            h //= 2
            c2 = cfg.dims[i + 1]
            dw += 2 * h * h * 49 * c2
            dense += 2 * h * h * c2 * c2
    c = cfg.dims[-1]
    co = cfg.out_dim
    dw += 2 * h * h * 9 * co
    rd = int(co * cfg.se_ratio)
    dense += 2 * 2 * co * rd
    return {"dense": dense, "depthwise": dw, "tokens": h * h}
