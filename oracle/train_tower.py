"""Oracle: gradients of the UNFROZEN FastViT-HD tower (the tower half).  TEST INFRASTRUCTURE ONLY.

torch.autograd over the fp32 inference-form graph oracle/fastvit_hd.py states (parity unpinned like that file: the reference delegates the tower to HF remote code,
model/fastvlm_adapter.py:183-191, call site :533), i.e. what `loss.backward()` (training/trainer.py:175) would compute for the tower if model/fastvlm_adapter.py:501
did not wrap the backbone in no_grad and `freeze_backbone` (fastvla/configuration_fastvla.py:23, applied at model/fastvlm_adapter.py:170-173) were off.

What is trained is the re-parameterised form the product's kernels consume: every ConvFFN's 7x7 carries its (eval-mode) BatchNorm folded in -- `fold_tower`
replaces `convffn.conv.conv.weight` + `convffn.conv.bn.*` by `convffn.conv.folded.weight / .bias` (the same fold csrc/engine.hip load_ffn does); every other tensor
keeps its checkpoint key.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import fastvit_hd, head, qwen2
from .fastvit_hd import VT, TowerCfg


def fold_tower(p: Dict[str, torch.Tensor], cfg: TowerCfg, prefix: str = VT) -> Dict[str, torch.Tensor]:
    """checkpoint dict -> the same dict with every ConvFFN conv + BatchNorm replaced by the folded weight / bias (fp32)"""
    out = {}
    for k, v in p.items():
        if k.startswith(prefix) and ".convffn.conv." in k:
            continue
        out[k] = v
    for k in p:
        if k.startswith(prefix) and k.endswith(".convffn.conv.conv.weight"):
            pre = k[: -len("conv.weight")]            # "...convffn.conv."
            g, b = p[pre + "bn.weight"].float(), p[pre + "bn.bias"].float()
            m, v = p[pre + "bn.running_mean"].float(), p[pre + "bn.running_var"].float()
            sc = g / torch.sqrt(v + cfg.bn_eps)
            out[pre + "folded.weight"] = p[k].float() * sc.view(-1, 1, 1, 1)
            out[pre + "folded.bias"] = b - m * sc
    return out


def tower_keys(pf: Dict[str, torch.Tensor], prefix: str = VT):
    return [k for k in pf if k.startswith(prefix)]


def unit_prefix(unit) -> str:
    kind, i, idx, j = unit
    if kind == "stem":
        return "patch_embed."
    if kind == "head":
        return "conv_exp."
    return f"network.{idx}.{j}." if kind == "block" else f"network.{idx}."


def unit_backward(qf: Dict[str, torch.Tensor], x: torch.Tensor, unit, g_out: torch.Tensor, cfg: TowerCfg, emulate_bf16: bool = False):
    """One tower unit (fastvit_hd.tower_units entry, or ("head", ...) for conv_exp + SE) on NCHW fp32 input x with upstream gradient g_out (the output's shape):
    -> (y, dL/dx, {key (prefix stripped) -> gradient}).
    emulate_bf16: differentiate the forward the PRODUCT computes -- the same graph with a round-to-bf16 wherever its kernels round an activation or a depthwise
    weight (fastvit_hd.py's bf16-faithful mode; the casts are straight-through for autograd) -- instead of the all-fp32 graph: against the latter a single unit's
    gradient already carries the precision policy's 3e-3 .. 6e-3, which would hide a wrong tap in one of 49."""
    pre = unit_prefix(unit)
    leaf = {k: v.detach().clone().float().requires_grad_(True) for k, v in qf.items() if k.startswith(pre)}
    q = dict(qf)
    q.update(leaf)
    xx = x.detach().clone().float().requires_grad_(True)
    if unit[0] == "head":
        y = fastvit_hd.tower_head_forward(q, xx, cfg, emulate_bf16)          # (B, tokens, out_dim)
    else:
        y = fastvit_hd.unit_forward(q, xx, unit, cfg, emulate_bf16)
    y.backward(g_out)
    return y.detach(), xx.grad, {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}


def forward_backward(pf: Dict[str, torch.Tensor], head_p: Dict[str, torch.Tensor], pixels: torch.Tensor, input_ids: torch.Tensor, attention_mask: torch.Tensor,
                     states: torch.Tensor, targets: torch.Tensor, tcfg: TowerCfg, lcfg: qwen2.Qwen2Cfg, drop_mask: Optional[torch.Tensor] = None, drop_p: float = 0.0,
                     emulate_bf16: bool = False, tower_out_value: Optional[torch.Tensor] = None, unit_values=None):
    """The whole spliced policy with EVERYTHING trainable: pixels (B, 3, S, S) fp32 (already letterboxed) -> tower -> projector -> decoder -> head -> MSE.
    pf = fold_tower(checkpoint).  -> dict(loss, pred, tower_out, grads={key -> gradient} for every model.* tensor and `head.<key>`)."""
    keys = [k for k in pf if k.startswith("model.")]
    q = {k: (v.detach().clone().float().requires_grad_(True) if k in keys else v) for k, v in pf.items()}
    hp = {k: v.detach().clone().float().requires_grad_(True) for k, v in head_p.items()}
    # (the bf16-faithful tower: see unit_backward; projector / decoder / head stay fp32)
    # unit_values (the engine's own unit outputs, NCHW): VALUE teacher-forcing at every unit boundary -- each unit's forward value is replaced by the engine's while
    # the gradient flows through this graph, so two bf16 executions cannot drift apart (free-running they sit ~1e-2 apart after 17 units) before being differentiated
    qs = fastvit_hd.strip_prefix(q)
    xx = pixels.float()
    for n, unit in enumerate(fastvit_hd.tower_units(tcfg)):
        xx = fastvit_hd.unit_forward(qs, xx, unit, tcfg, emulate_bf16)
        if unit_values is not None:
            xx = xx + (unit_values[n].float() - xx).detach()
    emb = fastvit_hd.tower_head_forward(qs, xx, tcfg, emulate_bf16)
    if tower_out_value is not None:
        # VALUE teacher-forcing at the tower's output: everything downstream sees the embeddings the engine computed (so dL/dpred = 2 (pred - target) / n does not
        # amplify the bf16 tower's forward noise by |pred| / |pred - target|), while the gradient still flows through THIS graph's tower
        emb = emb + (tower_out_value.float() - emb).detach()
    tok = fastvit_hd.projector_forward(q, emb)
    pooled = qwen2.llm_pooled(q, input_ids, attention_mask, lcfg, image_tokens=tok, splice=True)
    pred = head.head_forward(hp, pooled, states, drop_mask=drop_mask, drop_p=drop_p)
    loss = F.mse_loss(pred, targets)
    loss.backward()
    grads = {k: (q[k].grad if q[k].grad is not None else torch.zeros_like(q[k])) for k in keys}
    grads.update({"head." + k: hp[k].grad for k in hp})
    return {"loss": loss.detach(), "pred": pred.detach(), "tower_out": emb.detach(), "grads": grads}
