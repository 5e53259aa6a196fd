"""Oracle: action-expert head, MSE, hand-derived backward, clip + AdamW, LR schedules.  TEST INFRASTRUCTURE ONLY.

Restates
  src/vla_fastvlm/fastvla/fastvlm_with_expert.py:23-38 (modules) and :50-54 (forward):
      s = SiLU(Linear(LayerNorm(state)))                       state_projection.{0,1}
      f = cat([feat, s]);  f = Linear(f); f = LayerNorm(f); f = SiLU(f); f = Dropout(f)   fusion.{0,1,2,3}
      f = SiLU(Linear(f))                                      fusion.{4,5}
      a = Linear(f)                                            action_head
  src/vla_fastvlm/fastvla/modeling_fastvla.py:56 and lerobot_fastvla/modeling_fastvla.py:132: F.mse_loss (mean over B*A)
  src/vla_fastvlm/training/trainer.py:60-66,171-182 (AdamW, clip_grad_norm_, step) and :233-244 (linear warmup/decay)
  src/vla_fastvlm/lerobot_fastvla/configuration_fastvla.py:51-59 (LeRobot AdamW preset; cosine-with-warmup preset,
      [UNVENDORED] lerobot CosineDecayWithWarmupSchedulerConfig formula restated below)

The backward is written out by hand (no autograd) so it is an independent statement; tests pin it against
autograd through the imported reference head (tests/golden/head_*.npz).
Parameter keys are the reference's state-dict keys under "model." (SURVEY.md section 5).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

HEAD_KEYS = (
    "state_projection.0.weight", "state_projection.0.bias",
    "state_projection.1.weight", "state_projection.1.bias",
    "fusion.0.weight", "fusion.0.bias",
    "fusion.1.weight", "fusion.1.bias",
    "fusion.4.weight", "fusion.4.bias",
    "action_head.weight", "action_head.bias",
)
LN_EPS = 1e-5


def head_shapes(feat_dim: int, state_dim: int, action_dim: int, hidden_dim: int, fusion_dim: int):
    return {
        "state_projection.0.weight": (state_dim,), "state_projection.0.bias": (state_dim,),
        "state_projection.1.weight": (hidden_dim, state_dim), "state_projection.1.bias": (hidden_dim,),
        "fusion.0.weight": (fusion_dim, feat_dim + hidden_dim), "fusion.0.bias": (fusion_dim,),
        "fusion.1.weight": (fusion_dim,), "fusion.1.bias": (fusion_dim,),
        "fusion.4.weight": (fusion_dim, fusion_dim), "fusion.4.bias": (fusion_dim,),
        "action_head.weight": (action_dim, fusion_dim), "action_head.bias": (action_dim,),
    }


def _ln_fwd(x, w, b):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + LN_EPS)
    xh = (x - mu) * rstd
    return xh * w + b, (xh, rstd)


def _ln_bwd(dy, w, cache):
    xh, rstd = cache
    dw = (dy * xh).sum(0)
    db = dy.sum(0)
    g = dy * w
    dx = rstd * (g - g.mean(-1, keepdim=True) - xh * (g * xh).mean(-1, keepdim=True))
    return dx, dw, db


def _silu_bwd(dy, x):
    s = torch.sigmoid(x)
    return dy * s * (1 + x * (1 - s))


def head_forward(p: Dict[str, torch.Tensor], feat: torch.Tensor, state: torch.Tensor,
                 drop_mask: Optional[torch.Tensor] = None, drop_p: float = 0.0, keep_cache: bool = False):
    """feat (B,H) fp32, state (B,Ds) fp32 -> actions (B,A).  drop_mask: (B,fusion) of {0,1} keep flags
    (training); None = eval / dropout 0."""
    n0, c0 = _ln_fwd(state, p["state_projection.0.weight"], p["state_projection.0.bias"])
    z1 = F.linear(n0, p["state_projection.1.weight"], p["state_projection.1.bias"])
    s = F.silu(z1)
    cat = torch.cat([feat, s], dim=-1)
    z2 = F.linear(cat, p["fusion.0.weight"], p["fusion.0.bias"])
    n2, c2 = _ln_fwd(z2, p["fusion.1.weight"], p["fusion.1.bias"])
    a2 = F.silu(n2)
    if drop_mask is not None and drop_p > 0.0:
        d2 = a2 * drop_mask / (1.0 - drop_p)
    else:
        d2 = a2
    z3 = F.linear(d2, p["fusion.4.weight"], p["fusion.4.bias"])
    a3 = F.silu(z3)
    act = F.linear(a3, p["action_head.weight"], p["action_head.bias"])
    if keep_cache:
        return act, dict(n0=n0, c0=c0, z1=z1, cat=cat, c2=c2, n2=n2, d2=d2, z3=z3, a3=a3,
                         drop_mask=drop_mask, drop_p=drop_p, feat_dim=feat.shape[1])
    return act


def mse(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return ((pred - target) ** 2).mean()


def head_mse_backward(p: Dict[str, torch.Tensor], cache: dict, pred: torch.Tensor, target: torch.Tensor
                      ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """-> (loss, grads keyed like HEAD_KEYS).  d(mean((pred-t)^2)) / d pred = 2 (pred-t) / (B*A)."""
    g: Dict[str, torch.Tensor] = {}
    loss = mse(pred, target)
    da = 2.0 * (pred - target) / pred.numel()
    g["action_head.weight"] = da.t() @ cache["a3"]
    g["action_head.bias"] = da.sum(0)
    da3 = da @ p["action_head.weight"]
    dz3 = _silu_bwd(da3, cache["z3"])
    g["fusion.4.weight"] = dz3.t() @ cache["d2"]
    g["fusion.4.bias"] = dz3.sum(0)
    dd2 = dz3 @ p["fusion.4.weight"]
    if cache["drop_mask"] is not None and cache["drop_p"] > 0.0:
        da2 = dd2 * cache["drop_mask"] / (1.0 - cache["drop_p"])
    else:
        da2 = dd2
    dn2 = _silu_bwd(da2, cache["n2"])
    dz2, g["fusion.1.weight"], g["fusion.1.bias"] = _ln_bwd(dn2, p["fusion.1.weight"], cache["c2"])
    g["fusion.0.weight"] = dz2.t() @ cache["cat"]
    g["fusion.0.bias"] = dz2.sum(0)
    dcat = dz2 @ p["fusion.0.weight"]
    ds = dcat[:, cache["feat_dim"]:]
    dz1 = _silu_bwd(ds, cache["z1"])
    g["state_projection.1.weight"] = dz1.t() @ cache["n0"]
    g["state_projection.1.bias"] = dz1.sum(0)
    dn0 = dz1 @ p["state_projection.1.weight"]
    _, g["state_projection.0.weight"], g["state_projection.0.bias"] = _ln_bwd(dn0, p["state_projection.0.weight"], cache["c0"])
    return loss, g


def clip_grad_norm(grads: Dict[str, torch.Tensor], max_norm: float):
    """torch.nn.utils.clip_grad_norm_ semantics: total L2 norm; coef = clamp(max_norm / (norm + 1e-6), max=1)."""
    total = torch.sqrt(sum((v.double() ** 2).sum() for v in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return {k: v * coef for k, v in grads.items()}, total


def adamw_step(p, g, m, v, step: int, lr: float, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 1e-4):
    """torch.optim.AdamW (decoupled decay, bias correction, amsgrad off).  step is 1-based.  Returns new (p,m,v)."""
    b1, b2 = betas
    out_p, out_m, out_v = {}, {}, {}
    for k in p:
        pk = p[k] * (1.0 - lr * weight_decay)
        mk = m[k] * b1 + g[k] * (1.0 - b1)
        vk = v[k] * b2 + g[k] * g[k] * (1.0 - b2)
        bc1 = 1.0 - b1 ** step
        bc2 = 1.0 - b2 ** step
        denom = vk.sqrt() / math.sqrt(bc2) + eps
        out_p[k] = pk - (lr / bc1) * mk / denom
        out_m[k], out_v[k] = mk, vk
    return out_p, out_m, out_v


def trainer_lr_lambda(step: int, total_steps: int, warmup_ratio: float) -> float:
    """training/trainer.py:233-244: linear warmup for int(total*ratio) steps then linear decay to 0."""
    warm = int(total_steps * warmup_ratio)
    if step < warm:
        return float(step) / float(max(1, warm))
    return max(0.0, float(total_steps - step) / float(max(1, total_steps - warm)))


def lerobot_cosine_lr_lambda(step: int, peak_lr: float = 1e-4, decay_lr: float = 2.5e-6, warmup: int = 500,
                             decay_steps: int = 20_000) -> float:
    """[UNVENDORED] lerobot CosineDecayWithWarmupSchedulerConfig.build lr_lambda (multiplier on peak_lr):
    warmup: step<=0 -> 1/(warmup+1); else frac = 1 - step/warmup; 1/(warmup+1) + (1 - 1/(warmup+1)) * (1-frac)
    decay : step = min(step, decay_steps); cos = 0.5 (1 + cos(pi step / decay_steps)); alpha = decay_lr/peak_lr;
            (1-alpha) cos + alpha."""
    if step < warmup:
        if step <= 0:
            return 1.0 / (warmup + 1)
        frac = 1.0 - step / warmup
        return (1.0 / (warmup + 1) - 1.0) * frac + 1.0
    s = min(step, decay_steps)
    cos = 0.5 * (1.0 + math.cos(math.pi * s / decay_steps))
    alpha = decay_lr / peak_lr
    return (1.0 - alpha) * cos + alpha
