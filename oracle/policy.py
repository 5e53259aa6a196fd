"""Oracle: the whole policy step (img + prompt + state -> action; loss; head grads; one optimiser step).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the composition in
  src/vla_fastvlm/fastvla/fastvlm_with_expert.py:40-54      backbone -> state_projection -> cat -> fusion -> action_head
  src/vla_fastvlm/model/fastvlm_adapter.py:501-560           letterbox -> VLM -> hidden_states[-1] -> pool
  src/vla_fastvlm/fastvla/processor_fastvla.py:23-43         task broadcast + trailing newline, last-timestep slicing
  src/vla_fastvlm/lerobot_fastvla/modeling_fastvla.py:81-133 first camera, last timestep, gt[:,0], action deque
"""
from __future__ import annotations

from collections import deque
from typing import Dict, List, Optional, Sequence

import torch

from . import fastvit_hd, head, preprocess, qwen2


def normalize_tasks(tasks, batch_size: int, add_trailing_newline: bool = True) -> List[str]:
    """fastvla/processor_fastvla.py:23-30."""
    if isinstance(tasks, str):
        tasks = [tasks]
    tasks = list(tasks)
    if len(tasks) == 1 and batch_size > 1:
        tasks = [tasks[0]] * batch_size
    if add_trailing_newline:
        tasks = [t if t.endswith("\n") else t + "\n" for t in tasks]
    return tasks


def lerobot_tasks(task, batch_size: int, add_trailing_newline: bool = True) -> List[str]:
    """lerobot_fastvla/modeling_fastvla.py:91-105."""
    if task is None:
        tasks = [""] * batch_size
    elif isinstance(task, str):
        tasks = [task] * batch_size
    elif isinstance(task, (list, tuple)):
        tasks = [str(t) for t in task]
        if len(tasks) == 1 and batch_size > 1:
            tasks = tasks * batch_size
    else:
        tasks = [str(task)] * batch_size
    if add_trailing_newline:
        tasks = [t if t.endswith("\n") else t + "\n" for t in tasks]
    return tasks


def last_timestep(x: torch.Tensor, base_ndim: int) -> torch.Tensor:
    """images (B,T,C,H,W)->(B,C,H,W) / states (B,T,D)->(B,D): `[:, -1]` (processor_fastvla.py:32-40)."""
    return x[:, -1] if x.ndim == base_ndim + 1 else x


def backbone_features(params: Dict[str, torch.Tensor], images: torch.Tensor, input_ids: torch.Tensor,
                      attention_mask: torch.Tensor, *, image_size: int, llm_cfg: qwen2.Qwen2Cfg,
                      tower_cfg: fastvit_hd.TowerCfg = fastvit_hd.TowerCfg(), splice: bool = False,
                      pad_value: float = 0.0, run_tower: bool = True, pool: str = "last_token",
                      emulate_bf16_tower: bool = False):
    """-> (pooled (B,H), image_tokens or None).  With splice=False the tower/projector output is computed and then
    dropped, exactly what the literal reference does."""
    img_tok = None
    if run_tower or splice:
        pix = preprocess.letterbox(images, image_size, pad_value)
        if emulate_bf16_tower:  # the product's precision policy for the tower (oracle/fastvit_hd.py, bf16-faithful mode)
            pix = fastvit_hd._r(pix)
        emb = fastvit_hd.tower_forward(params, pix, tower_cfg, emulate_bf16=emulate_bf16_tower)
        img_tok = fastvit_hd.projector_forward(params, emb, emulate_bf16=emulate_bf16_tower)
    pooled = qwen2.llm_pooled(params, input_ids, attention_mask, llm_cfg, img_tok, splice=splice, pool=pool)
    return pooled, img_tok


def policy_forward(params, head_params, images, states, input_ids, attention_mask, **kw):
    pooled, _ = backbone_features(params, images, input_ids, attention_mask, **kw)
    return head.head_forward(head_params, pooled, states.float())


def train_step(params, head_params, opt_m, opt_v, step: int, images, states, targets, input_ids, attention_mask, *,
               lr: float, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-4, max_grad_norm: Optional[float] = 1.0,
               drop_mask=None, drop_p: float = 0.0, **kw):
    """One optimiser step exactly as trainer.py:171-182 orders it: loss -> backward -> clip -> AdamW."""
    pooled, _ = backbone_features(params, images, input_ids, attention_mask, **kw)
    pred, cache = head.head_forward(head_params, pooled, states.float(), drop_mask, drop_p, keep_cache=True)
    loss, grads = head.head_mse_backward(head_params, cache, pred, targets.float())
    gnorm = None
    if max_grad_norm is not None:
        grads, gnorm = head.clip_grad_norm(grads, max_grad_norm)
    new_p, new_m, new_v = head.adamw_step(head_params, grads, opt_m, opt_v, step, lr, betas, eps, weight_decay)
    return dict(loss=loss, pred=pred, grads=grads, grad_norm=gnorm, params=new_p, m=new_m, v=new_v)


class ActionQueue:
    """lerobot_fastvla/modeling_fastvla.py:78-79,119-125: deque(maxlen=n_action_steps) refilled from a [B,1,A] chunk."""

    def __init__(self, n_action_steps: int = 1):
        self.n = n_action_steps
        self.q: deque = deque([], maxlen=n_action_steps)

    def select(self, predict_chunk):
        if len(self.q) == 0:
            chunk = predict_chunk()[:, : self.n]
            self.q.extend(chunk.transpose(0, 1))
        return self.q.popleft()
