"""Standalone FastVLA policy (reference: src/vla_fastvlm/fastvla/modeling_fastvla.py:14-77): forward / compute_loss /
select_action / reset with the reference's signatures, plus `fused_train_step`, the native train step
(forward -> MSE -> head backward -> [all-reduce] -> clip -> AdamW in libfastvla_hip.so) that
vla_fastvlm.training.Trainer and bench.py drive."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import nn
from torch.nn import functional as F

from .configuration_fastvla import FastVLAConfig
from .fastvlm_with_expert import FastVLMWithExpert
from .processor_fastvla import FastVLAProcessor


class FastVLAPolicy(nn.Module):
    config_class = FastVLAConfig
    name = "fastvla"

    def __init__(self, config: FastVLAConfig | None = None) -> None:
        super().__init__()
        self.config = config or FastVLAConfig()
        self.model = FastVLMWithExpert(self.config)
        self.processor = FastVLAProcessor(self.config, self.model.backbone)
        self._opt_state = None

    def forward(self, images: torch.Tensor, states: torch.Tensor, tasks: List[str] | str,
                device: torch.device | None = None) -> torch.Tensor:
        if device is None:
            device = images.device
        images = self.processor.prepare_images(images, device)
        states = self.processor.prepare_states(states, device)
        tasks = self.processor.prepare_tasks(tasks, batch_size=images.shape[0])
        return self.model(images, states, tasks, device=device)

    def compute_loss(self, batch: Dict[str, torch.Tensor | List[str]]) -> Dict[str, torch.Tensor]:
        pred = self.forward(batch["images"], batch["states"], batch["tasks"])
        mse = F.mse_loss(pred, batch["actions"].to(pred.device, pred.dtype))
        return {"loss": mse, "mse": mse.detach()}

    @torch.inference_mode()
    def select_action(self, image: torch.Tensor, state: torch.Tensor, task: str, device: torch.device) -> torch.Tensor:
        self.eval()
        tasks = self.processor.prepare_tasks(task, batch_size=1)
        action = self.forward(image.unsqueeze(0).to(device), state.unsqueeze(0).to(device), tasks, device=device)
        return action.squeeze(0)

    def reset(self) -> None:
        return

    # ------------------------------------------------------------------ native train step
    def fused_train_step(self, batch: Dict[str, torch.Tensor | List[str]], *, lr: float, betas=(0.9, 0.95), eps: float = 1e-8,
                         weight_decay: float = 1e-4, max_grad_norm: Optional[float] = 1.0,
                         process_group=None) -> Dict[str, torch.Tensor]:
        """One optimiser step with the ordering of reference training/trainer.py:171-182 (loss -> backward -> clip ->
        AdamW), entirely on the HIP path.  Under torch.distributed the flat head gradient is summed across ranks with
        ONE all-reduce on a side stream and the 1/world scale is folded into the optimiser kernel."""
        from ..training.dp import allreduce_flat_grads
        m = self.model
        dev = m.backbone.engine().device
        images = self.processor.prepare_images(batch["images"], dev)
        states = self.processor.prepare_states(batch["states"], dev).float()
        tasks = self.processor.prepare_tasks(batch["tasks"], batch_size=images.shape[0])
        targets = batch["actions"].to(dev, torch.float32)
        if targets.ndim == 3:
            targets = targets[:, 0]
        eng, flat = m._engine(), m.materialize(dev)
        with torch.no_grad():
            pooled = m.features(images, tasks, device=dev)
        if self._opt_state is None or self._opt_state["m"].data_ptr() == 0 or self._opt_state["flat"] is not flat:
            self._opt_state = dict(m=torch.zeros_like(flat), v=torch.zeros_like(flat), g=torch.zeros_like(flat), step=0,
                                   flat=flat, comm=torch.cuda.Stream(device=dev), norm=torch.zeros(1, device=dev))
        st = self._opt_state
        st["step"] += 1
        p = float(self.config.dropout) if self.training else 0.0
        m._drop_calls += 1
        actions, saved = eng.head_forward(flat, pooled, states, training=p > 0.0, dropout_p=p, seed=m._drop_seed,
                                          offset=m._drop_calls)
        loss, grads = eng.head_backward(flat, actions, targets, saved, dropout_p=p, flat_grads=st["g"])
        scale = allreduce_flat_grads(grads, st["comm"], process_group)
        eng.adamw_step(flat, grads, st["m"], st["v"], st["step"], lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                       max_grad_norm=max_grad_norm or 0.0, grad_scale=scale, grad_norm_out=st["norm"])
        return {"loss": loss[0], "mse": loss[0].detach(), "grad_norm": st["norm"][0], "actions": actions}
