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
    if pre + "conv.folded.weight" in p:   # the TRAINABLE inference form (oracle/train_tower.py fold_tower): BatchNorm already inside the 7x7
        y = F.conv2d(x, p[pre + "conv.folded.weight"], p[pre + "conv.folded.bias"], padding=3, groups=c)
        return _conv(gelu(_conv(y, p, pre + "fc1")), p, pre + "fc2")
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


# ---------------------------------------------------------------------------------------------------------------------
# bf16-faithful mode.  The product keeps tower activations in bf16 (fp32 accumulation inside every kernel) and rounds the
# depthwise weights of its matrix-core depthwise kernels to bf16; the reference under its default
# TrainingConfig.mixed_precision="bf16" (training/trainer.py:31) rounds at least as often.  Against the fp32 graph above
# that is 2-3 % rel-L2 after 44 blocks at 1024^2 -- too loose to notice a wrong tap in one of 49 taps.  `emulate_bf16=True`
# restates the SAME graph with a round-to-bf16 at exactly the points where the product's kernels round
# (csrc/engine.hip tower_pass; one `_r` per tensor that reaches HBM or an MFMA operand), so the comparison isolates the
# kernels' arithmetic from the precision policy.  Everything else (fp32 accumulation, exact-erf GELU, softmax) stays as is.
class _RoundBF16(torch.autograd.Function):
    """round-to-bf16 with a STRAIGHT-THROUGH gradient.  (Plain `x.to(bfloat16).to(float32)` is differentiable too, but its backward casts the incoming
    gradient to bf16 on the way: oracle/train_tower.py differentiates this graph, and the gradient must stay fp32.)"""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def _r(x):
    return _RoundBF16.apply(x) if x.requires_grad else x.to(torch.bfloat16).to(torch.float32)


def _dw_mfma(map_w: int, c: int) -> bool:
    """stride-1 depthwise layers the product runs on MFMA with bf16 Toeplitz tables (csrc/tower_kernels.hip
    dwconv_mfma_supported): their weights are rounded to bf16; smaller maps use the fp32-weight VALU kernel."""
    return map_w >= 16 and c % 32 == 0   # (W >= 16 since round 4: the last stage's 16 x 16 maps run on the MFMA kernels too)


def _convffn_bf16(x, p, pre, cfg: TowerCfg):
    """dw7x7 with the BatchNorm folded into weights/bias the way fv_load_weights folds it (csrc/engine.hip load_ffn), the
    folded weights rounded to bf16 where the MFMA depthwise kernel serves the layer; t, the GELU'd hidden and nothing else
    rounded (the second product's fp32 accumulator goes straight into the layer-scale + residual epilogue)."""
    c, w = x.shape[1], x.shape[-1]
    if pre + "conv.folded.weight" in p:   # the trainable inference form (oracle/train_tower.py fold_tower)
        wf, bf = p[pre + "conv.folded.weight"], p[pre + "conv.folded.bias"]
    else:
        sc = p[pre + "conv.bn.weight"] / torch.sqrt(p[pre + "conv.bn.running_var"] + cfg.bn_eps)
        wf = p[pre + "conv.conv.weight"] * sc.view(-1, 1, 1, 1)
        bf = p[pre + "conv.bn.bias"] - p[pre + "conv.bn.running_mean"] * sc
    t = _r(F.conv2d(x, _r(wf) if _dw_mfma(w, c) else wf, bf, padding=3, groups=c))
    h = _r(gelu(_conv(t, p, pre + "fc1")))
    return _conv(h, p, pre + "fc2")


def _mhsa_bf16(x, p, pre, cfg: TowerCfg):
    """attention32_kernel's rounding points: q/k/v bf16 (the qkv GEMM's output), scores and softmax in fp32, the
    exponentiated P rounded to bf16 as the second product's operand with the row sum taken over the ROUNDED P, output bf16."""
    b, c, h, w = x.shape
    n = h * w
    nh = c // cfg.head_dim
    t = x.flatten(2).transpose(1, 2)
    qkv = _r(F.linear(t, p[pre + "qkv.weight"])).reshape(b, n, 3, nh, cfg.head_dim).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    s = (q @ k.transpose(-2, -1)) * cfg.head_dim ** -0.5
    pr = _r(torch.exp(s - s.amax(dim=-1, keepdim=True)))
    o = _r((pr @ v) / pr.sum(dim=-1, keepdim=True)).transpose(1, 2).reshape(b, n, c)
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


def tower_units(cfg: TowerCfg):
    """The tower as a list of sequential UNITS, in execution order (the product's fv_vision_unit_info enumerates the same list):
    ("stem", 0, None, None), then per stage [("cpe", i, idx, None)], ("block", i, idx, j) ..., [("down", i, idx, None)] with idx the
    position in mci.py's `network` ModuleList.  conv_exp + SE and the projector follow the last unit."""
    out = [("stem", 0, None, None)]
    for idx, (kind, i) in enumerate(network_index_map(cfg)):
        if kind == "stage":
            out.extend(("block", i, idx, j) for j in range(cfg.layers[i]))
        else:
            out.append((kind, i, idx, None))
    return out


def unit_forward(q: Dict[str, torch.Tensor], x: torch.Tensor, unit, cfg: TowerCfg = TowerCfg(), emulate_bf16: bool = False,
                 taps: dict | None = None) -> torch.Tensor:
    """One tower unit on NCHW fp32 input (q = checkpoint dict with the tower prefix stripped).  A per-unit parity test feeds it
    the product's own input of that unit (teacher forcing) -- same arithmetic as tower_forward, which is the fold over units."""
    kind, i, idx, j = unit
    R = _r if emulate_bf16 else (lambda t: t)
    if kind == "stem":
        c0 = cfg.dims[0]
        w0 = q["patch_embed.0.reparam_conv.weight"]
        if emulate_bf16 and c0 % 16 == 0 and c0 <= 128:   # the implicit-GEMM stem packs its weights as bf16 MFMA operands
            w0 = _r(w0)
        x = R(gelu(F.conv2d(x, w0, q["patch_embed.0.reparam_conv.bias"], stride=2, padding=1)))
        if taps is not None:
            taps["stem0"] = x
        x = R(gelu(_conv(x, q, "patch_embed.1.reparam_conv", stride=2, padding=1, groups=c0)))
        return R(gelu(_conv(x, q, "patch_embed.2.reparam_conv")))
    c = cfg.dims[i]
    mw = x.shape[-1]
    if kind == "cpe":
        wc = q[f"network.{idx}.reparam_conv.weight"]
        return R(F.conv2d(x, _r(wc) if emulate_bf16 and _dw_mfma(mw, c) else wc, q[f"network.{idx}.reparam_conv.bias"], padding=3, groups=c))
    if kind == "down":
        wl = q[f"network.{idx}.proj.0.lkb_reparam.weight"]
        if emulate_bf16 and c % 32 == 0 and mw % 2 == 0 and mw >= 16:   # dwconv_s2_mfma_supported
            wl = _r(wl)
        x = R(gelu(F.conv2d(x, wl, q[f"network.{idx}.proj.0.lkb_reparam.bias"], stride=2, padding=3, groups=c)))
        return R(gelu(_conv(x, q, f"network.{idx}.proj.1.reparam_conv")))
    pre = f"network.{idx}.{j}."
    ffn = _convffn_bf16 if emulate_bf16 else _convffn
    if i in cfg.attn_stages:
        y = R(_layernorm_channel(x, q[pre + "norm.weight"], q[pre + "norm.bias"], cfg.ln_eps))
        att = (_mhsa_bf16 if emulate_bf16 else _mhsa)(y, q, pre + "token_mixer.", cfg)
        x = R(x + q[pre + "layer_scale_1"].view(1, -1, 1, 1) * att)
        return R(x + q[pre + "layer_scale_2"].view(1, -1, 1, 1) * ffn(x, q, pre + "convffn.", cfg))
    wm = q[pre + "token_mixer.reparam_conv.weight"]
    x = R(F.conv2d(x, _r(wm) if emulate_bf16 and _dw_mfma(mw, c) else wm, q[pre + "token_mixer.reparam_conv.bias"], padding=1, groups=c))
    return R(x + q[pre + "layer_scale"].view(1, -1, 1, 1) * ffn(x, q, pre + "convffn.", cfg))


def tower_head_forward(q: Dict[str, torch.Tensor], x: torch.Tensor, cfg: TowerCfg = TowerCfg(), emulate_bf16: bool = False) -> torch.Tensor:
    """conv_exp (dw3x3, channel multiplier 2) + SE + GELU on the last stage's map -> (B, tokens, out_dim)."""
    R = _r if emulate_bf16 else (lambda t: t)
    c = cfg.dims[-1]
    x = R(_conv(x, q, "conv_exp.reparam_conv", padding=1, groups=c))
    s = x.mean(dim=(2, 3), keepdim=True)
    s = F.relu(_conv(s, q, "conv_exp.se.reduce"))
    s = torch.sigmoid(_conv(s, q, "conv_exp.se.expand"))
    x = R(gelu(x * s))
    return x.flatten(2).transpose(1, 2).contiguous()


def strip_prefix(p: Dict[str, torch.Tensor], prefix: str = VT) -> Dict[str, torch.Tensor]:
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


def tower_forward(p: Dict[str, torch.Tensor], x: torch.Tensor, cfg: TowerCfg = TowerCfg(), prefix: str = VT,
                  taps: dict | None = None, emulate_bf16: bool = False) -> torch.Tensor:
    """x: (B,3,S,S) fp32, S % 64 == 0  ->  image embeddings (B, (S/64)^2, out_dim).
    emulate_bf16: the same graph with the product's bf16 rounding points (see the block comment above)."""
    q = strip_prefix(p, prefix)
    units = tower_units(cfg)
    for n, unit in enumerate(units):
        x = unit_forward(q, x, unit, cfg, emulate_bf16, taps)
        if taps is not None:
            if unit[0] == "stem":
                taps["stem"] = x
            elif unit[0] == "block" and (n + 1 == len(units) or units[n + 1][0] != "block" or units[n + 1][1] != unit[1]):
                taps[f"stage{unit[1]}"] = x
    return tower_head_forward(q, x, cfg, emulate_bf16)


def projector_forward(p: Dict[str, torch.Tensor], tokens: torch.Tensor, prefix: str = PROJ, emulate_bf16: bool = False) -> torch.Tensor:
    h = gelu(F.linear(tokens, p[prefix + "0.weight"], p[prefix + "0.bias"]))
    if emulate_bf16:  # the hidden is the second GEMM's bf16 operand; the output stays fp32 (FV_EPI_F32)
        h = _r(h)
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
