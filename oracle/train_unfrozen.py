"""Oracle: gradients of the UNFROZEN decoder + projector + action expert.  TEST INFRASTRUCTURE ONLY.

torch.autograd over the fp32 forward oracle/ already states -- `fastvit_hd.projector_forward` ([site] fast_vlm/modeling_fast_vlm.py:39-56),
`qwen2.llm_pooled(splice=True)` ([site] qwen2/modeling_qwen2.py; reference call site model/fastvlm_adapter.py:533, pooling :551-559),
`head.head_forward` + `F.mse_loss` (fastvla/fastvlm_with_expert.py:50-54, fastvla/modeling_fastvla.py:56) -- i.e. exactly what
`loss.backward()` (training/trainer.py:175) would compute if model/fastvlm_adapter.py:501 did not wrap the backbone in no_grad, followed
by the reference's step body (clip_grad_norm_ + AdamW, training/trainer.py:60-66,178-180) over ALL trainable tensors.  The vision tower
stays frozen: its embeddings (B, Ni, tower_out_dim) are an input.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import fastvit_hd, head, qwen2


def trainable_backbone_keys(p: Dict[str, torch.Tensor]):
    return [k for k in p if k.startswith("model.") and not k.startswith("model.vision_tower.")]


def forward_backward(p: Dict[str, torch.Tensor], head_p: Dict[str, torch.Tensor], tower_out: torch.Tensor, input_ids: torch.Tensor,
                     attention_mask: torch.Tensor, states: torch.Tensor, targets: torch.Tensor, cfg: qwen2.Qwen2Cfg,
                     drop_mask: Optional[torch.Tensor] = None, drop_p: float = 0.0):
    """-> dict(loss, pred, grads={canonical key -> fp32 gradient} for every decoder / projector tensor and `head.<key>` for the 12 head
    tensors).  tower_out: (B, Ni, C) fp32 embeddings of the frozen tower."""
    bk = trainable_backbone_keys(p)
    q = {k: (v.detach().clone().float().requires_grad_(True) if k in bk else v) for k, v in p.items()}
    hp = {k: v.detach().clone().float().requires_grad_(True) for k, v in head_p.items()}
    tok = fastvit_hd.projector_forward(q, tower_out.float())
    pooled = qwen2.llm_pooled(q, input_ids, attention_mask, cfg, image_tokens=tok, splice=True)
    pred = head.head_forward(hp, pooled, states, drop_mask=drop_mask, drop_p=drop_p)
    loss = F.mse_loss(pred, targets)
    loss.backward()
    grads = {k: (q[k].grad if q[k].grad is not None else torch.zeros_like(q[k])) for k in bk}
    grads.update({"head." + k: hp[k].grad for k in hp})
    return {"loss": loss.detach(), "pred": pred.detach(), "pooled": pooled.detach(), "grads": grads}


def adamw_clip_step(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], *, lr: float, betas=(0.9, 0.95), eps: float = 1e-8,
                    weight_decay: float = 1e-4, max_grad_norm: float = 1.0):
    """FIRST optimiser step (m = v = 0) of torch.optim.AdamW after clip_grad_norm_ over all tensors -> (new params, global grad norm)."""
    norm = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_grad_norm / (norm + 1e-6), max=1.0) if max_grad_norm and max_grad_norm > 0 else torch.tensor(1.0)
    out = {}
    b1, b2 = betas
    for k, w in params.items():
        g = grads[k] * coef
        m = (1 - b1) * g
        v = (1 - b2) * g * g
        mh, vh = m / (1 - b1), v / (1 - b2)
        out[k] = w * (1 - lr * weight_decay) - lr * mh / (vh.sqrt() + eps)
    return out, norm
