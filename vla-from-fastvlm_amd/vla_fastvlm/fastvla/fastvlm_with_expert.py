"""FastVLM backbone + action-expert head (reference: src/vla_fastvlm/fastvla/fastvlm_with_expert.py:12-54).

The module tree (`state_projection`, `fusion`, `action_head`) and therefore the state-dict keys are the reference's;
the 12 trainable tensors are views into ONE flat fp32 device buffer so the HIP head kernels, the fused AdamW and the
RCCL all-reduce each see a single contiguous array, while `parameters()`, `state_dict()` and any torch optimizer keep
working on the same nn.Parameters.  forward/backward of the head run in libfastvla_hip.so via one autograd.Function.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from fastvla_hip import HEAD_KEYS

from ..model.fastvlm_adapter import FastVLMBackbone
from .configuration_fastvla import FastVLAConfig


class _HeadFunction(torch.autograd.Function):
    """actions = head(pooled, states); backward hands dL/dactions to fv_head_backward and returns views of the flat
    gradient buffer for the 12 parameters (the frozen backbone and the states get no gradient)."""

    @staticmethod
    def forward(ctx, owner, pooled, states, training, differentiable, *params):
        eng, flat = owner._engine(), owner._flat
        p = float(owner.config.dropout) if training else 0.0
        owner._drop_calls += 1
        # `differentiable` (a gradient may be asked of these actions, in train OR eval mode): keep them in normalised space --
        # fv_head_backward differentiates the head, not the folded `* action_std + action_mean` behind it
        actions, saved = eng.head_forward(flat, pooled, states, training=bool(training and p > 0.0), dropout_p=p,
                                          seed=owner._drop_seed, offset=owner._drop_calls,
                                          normalized_actions=bool(training or differentiable))
        ctx.owner, ctx.saved, ctx.p = owner, saved, p
        return actions

    @staticmethod
    def backward(ctx, grad_actions):
        owner = ctx.owner
        grads = owner._engine().head_backward_from_grad(owner._flat, grad_actions, ctx.saved, ctx.p)
        views = owner._engine().head_views(grads)
        return (None, None, None, None, None) + tuple(views[k] for k in HEAD_KEYS)


class _HeadLossFunction(torch.autograd.Function):
    """(loss, actions) = (mse(head(pooled, states), targets), head(...)) with the head forward, the MSE and the head
    backward all inside libfastvla_hip.so (fv_head_forward + fv_head_mse_backward): what the reference spells as
    `F.mse_loss(pred, target)` + autograd (fastvla/modeling_fastvla.py:56, lerobot_fastvla/modeling_fastvla.py:132).
    backward() only scales the flat gradient by the incoming dL/dloss (fv_grad_scale) and hands out views of it."""

    @staticmethod
    def forward(ctx, owner, pooled, states, targets, training, *params):
        eng, flat = owner._engine(), owner._flat
        p = float(owner.config.dropout) if training else 0.0
        owner._drop_calls += 1
        actions, saved = eng.head_forward(flat, pooled, states, training=bool(training and p > 0.0), dropout_p=p,
                                          seed=owner._drop_seed, offset=owner._drop_calls, normalized_actions=True)  # a loss: normalised space
        loss, grads = eng.head_backward(flat, actions, targets, saved, dropout_p=p)
        ctx.owner, ctx.grads = owner, grads
        ctx.mark_non_differentiable(actions)
        return loss[0], actions

    @staticmethod
    def backward(ctx, grad_loss, _grad_actions):
        eng = ctx.owner._engine()
        eng.grad_scale(ctx.grads, grad_loss.reshape(1).to(ctx.grads.device, torch.float32).contiguous())
        views = eng.head_views(ctx.grads)
        return (None, None, None, None, None) + tuple(views[k] for k in HEAD_KEYS)


class FastVLMWithExpert(nn.Module):
    def __init__(self, config: FastVLAConfig) -> None:
        super().__init__()
        self.config = config
        self.backbone = FastVLMBackbone(config.to_backbone_config())
        self.backbone.configure_head(state_dim=config.state_dim, action_dim=config.action_dim,
                                     hidden_dim=config.hidden_dim, fusion_dim=config.fusion_dim)
        # same module tree / init as the reference so checkpoints and seeds line up
        self.state_projection = nn.Sequential(nn.LayerNorm(config.state_dim), nn.Linear(config.state_dim, config.hidden_dim), nn.SiLU())
        self.fusion = nn.Sequential(
            nn.Linear(self.backbone.output_dim + config.hidden_dim, config.fusion_dim), nn.LayerNorm(config.fusion_dim),
            nn.SiLU(), nn.Dropout(config.dropout), nn.Linear(config.fusion_dim, config.fusion_dim), nn.SiLU())
        self.action_head = nn.Linear(config.fusion_dim, config.action_dim)
        self._flat: Optional[torch.Tensor] = None
        self._drop_seed = int(torch.initial_seed() & 0x7FFFFFFFFFFFFFFF)
        self._drop_calls = 0

    # ------------------------------------------------------------------ flat parameter storage
    def head_parameters(self):
        named = dict(self.named_parameters())
        return [named[k] for k in HEAD_KEYS]

    def _engine(self):
        return self.backbone.engine()

    def materialize(self, device: torch.device | None = None) -> torch.Tensor:
        """Move the 12 head tensors into one flat device buffer (idempotent) and re-point the Parameters at it."""
        eng = self.backbone.engine(device)
        params = self.head_parameters()
        views = None if self._flat is None else eng.head_views(self._flat)
        if views is not None and all(p.data_ptr() == views[k].data_ptr() for p, k in zip(params, HEAD_KEYS)):
            return self._flat
        flat = torch.zeros(eng.head_numel(), dtype=torch.float32, device=eng.device)
        views = eng.head_views(flat)
        with torch.no_grad():
            for p, k in zip(params, HEAD_KEYS):
                views[k].copy_(p.detach().to(eng.device, torch.float32))
                p.data = views[k]
        self._flat = flat
        return flat

    def flat_grads(self) -> Optional[torch.Tensor]:
        """The flat gradient buffer if every .grad is a view of one (true after a backward through _HeadFunction)."""
        params = self.head_parameters()
        if any(p.grad is None for p in params):
            return None
        base = params[0].grad._base if params[0].grad._base is not None else None
        if base is None or any(p.grad._base is not base for p in params):
            return None
        return base

    # ------------------------------------------------------------------ forward
    def features(self, images, tasks: List[str], device=None) -> torch.Tensor:
        return self.backbone(images, tasks, device=device)

    def head(self, pooled: torch.Tensor, states: torch.Tensor) -> torch.Tensor:
        self.materialize(pooled.device)
        states = states.to(pooled.device, torch.float32)
        if states.ndim != 2 or states.shape[1] != self.config.state_dim:
            raise ValueError(f"states must be (B,{self.config.state_dim}), got {tuple(states.shape)}")
        params = self.head_parameters()
        differentiable = torch.is_grad_enabled() and any(p.requires_grad for p in params)
        actions = _HeadFunction.apply(self, pooled, states, self.training, differentiable, *params)
        io = self.backbone._io_norm
        if differentiable and not self.training and io is not None:
            # eval mode with autograd on: the library kept the actions in normalised space (its backward differentiates the head, not
            # the folded statistics); finish `* action_std + action_mean` here, in torch, so that predict() returns the SAME space with
            # and without torch.no_grad() -- and the result stays differentiable
            std = torch.as_tensor(io["action_std"], dtype=torch.float32, device=actions.device).reshape(1, -1)
            mean = torch.as_tensor(io["action_mean"], dtype=torch.float32, device=actions.device).reshape(1, -1)
            actions = actions * std + mean
        return actions

    def head_loss(self, pooled: torch.Tensor, states: torch.Tensor, targets: torch.Tensor):
        """-> (loss 0-dim, actions): MSE of the head's prediction against `targets`, differentiable w.r.t. the 12 head
        tensors, with no torch operator between the pooled feature and the loss."""
        self.materialize(pooled.device)
        states = states.to(pooled.device, torch.float32)
        targets = targets.to(pooled.device, torch.float32).contiguous()
        if states.ndim != 2 or states.shape[1] != self.config.state_dim:
            raise ValueError(f"states must be (B,{self.config.state_dim}), got {tuple(states.shape)}")
        if targets.shape != (pooled.shape[0], self.config.action_dim):
            raise ValueError(f"targets must be (B,{self.config.action_dim}), got {tuple(targets.shape)}")
        return _HeadLossFunction.apply(self, pooled, states, targets, self.training, *self.head_parameters())

    def forward(self, images: torch.Tensor, states: torch.Tensor, tasks: List[str], device: torch.device | None = None) -> torch.Tensor:
        if device is None:
            device = images.device
        return self.head(self.features(images, tasks, device=device), states)

    def forward_loss(self, images, states, tasks: List[str], targets: torch.Tensor, device: torch.device | None = None):
        """-> (loss, actions) for `compute_loss` / the LeRobot `forward(batch)`."""
        if device is None:
            device = images.device
        return self.head_loss(self.features(images, tasks, device=device), states, targets)
