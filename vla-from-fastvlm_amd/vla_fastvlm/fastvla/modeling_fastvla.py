"""Standalone FastVLA policy (reference: src/vla_fastvlm/fastvla/modeling_fastvla.py:14-77): forward / compute_loss /
select_action / reset with the reference's signatures, plus `fused_train_step`, the native train step
(forward -> MSE -> head backward -> [all-reduce] -> clip -> AdamW in libfastvla_hip.so) that
vla_fastvlm.training.Trainer and bench.py drive."""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
from torch import nn

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
        self._unfrozen = None   # training/unfrozen.py UnfrozenState once enable_backbone_training() ran

    def enable_backbone_training(self, bucket_min_numel: int = 1 << 22, tower: Optional[bool] = None):
        """Extension of this build: fine-tune the Qwen2 decoder + mm_projector together with the action expert
        (image tokens spliced); tower=True (or FASTVLA_TRAIN_TOWER=1) trains the FastViT-HD tower too, in its inference form, otherwise it stays
        frozen.  Explicit on purpose: the reference's `freeze_backbone=False` trains nothing but the head either (model/fastvlm_adapter.py:501),
        so the config flag alone must not change what a step computes."""
        if tower is None:
            tower = os.environ.get("FASTVLA_TRAIN_TOWER", "0") == "1"
        if self._unfrozen is not None and bool(tower) and not self._unfrozen.train_tower:
            raise RuntimeError("backbone training is already running with the tower frozen: ask for tower=True on the first call")
        if self._unfrozen is None:
            from ..training.unfrozen import UnfrozenState
            self._unfrozen = UnfrozenState(self, bucket_min_numel=bucket_min_numel, train_tower=bool(tower))
        return self._unfrozen

    def forward(self, images: torch.Tensor, states: torch.Tensor, tasks: List[str] | str,
                device: torch.device | None = None) -> torch.Tensor:
        if device is None:
            device = images.device
        images = self.processor.prepare_images(images, device)
        states = self.processor.prepare_states(states, device)
        tasks = self.processor.prepare_tasks(tasks, batch_size=images.shape[0])
        return self.model(images, states, tasks, device=device)

    def compute_loss(self, batch: Dict[str, torch.Tensor | List[str]]) -> Dict[str, torch.Tensor]:
        """reference fastvla/modeling_fastvla.py:52-57: {"loss": mse (differentiable w.r.t. the head), "mse": detached}.
        Head forward, MSE and the head gradients come from one pass through the library (no torch operator in between)."""
        images, states, tasks = batch["images"], batch["states"], batch["tasks"]
        device = images.device
        images = self.processor.prepare_images(images, device)
        states = self.processor.prepare_states(states, device)
        tasks = self.processor.prepare_tasks(tasks, batch_size=images.shape[0])
        targets = batch["actions"]
        if targets.ndim == 3:
            targets = targets[:, 0]
        mse, _pred = self.model.forward_loss(images, states, tasks, targets, device=device)
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
    def prepare_batch(self, batch: Dict[str, torch.Tensor | List[str]]) -> Dict[str, torch.Tensor]:
        """Everything of a train step that does NOT depend on the trainable parameters: image prep, tokenisation and the
        frozen backbone forward (reference model/fastvlm_adapter.py:501: no_grad, frozen) -> pooled features.  Enqueued on
        the current stream; a pipelined loop calls it for batch k+1 while batch k's gradient all-reduce is in flight."""
        m = self.model
        dev = m.backbone.engine().device
        images = self.processor.prepare_images(batch["images"], dev)
        states = self.processor.prepare_states(batch["states"], dev).float()
        tasks = self.processor.prepare_tasks(batch["tasks"], batch_size=images.shape[0])
        targets = batch["actions"].to(dev, torch.float32)
        if targets.ndim == 3:
            targets = targets[:, 0]
        with torch.no_grad():
            pooled = m.features(images, tasks, device=dev)
        return {"pooled": pooled, "states": states, "targets": targets.contiguous()}

    def _optimizer_state(self, flat: torch.Tensor) -> Dict:
        st = self._opt_state
        if st is None or st.get("flat") is not flat:
            from ..training.dp import GradExchange
            dev = flat.device
            keep = st or {}
            st = dict(m=torch.zeros_like(flat), v=torch.zeros_like(flat), g=torch.zeros_like(flat), acc=None, step=0, micro=0,
                      flat=flat, exchange=GradExchange(dev), norm=torch.zeros(1, device=dev), loss=torch.zeros(1, device=dev))
            if "resume" in keep:  # load_optimizer_state() ran before the flat buffer existed
                r = keep["resume"]
                st["m"].copy_(r["m"].to(dev))
                st["v"].copy_(r["v"].to(dev))
                st["step"] = int(r["step"])
            self._opt_state = st
        return st

    def load_optimizer_state(self, m: torch.Tensor, v: torch.Tensor, step: int, flat: Optional[torch.Tensor] = None,
                             train_tower: Optional[bool] = None) -> None:
        """Restore AdamW moments and the bias-correction step (Trainer._load_checkpoint; reference trainer.py:257-262
        restores them through accelerator.load_state).  train_tower: what optimizer.pt recorded about the run being resumed (None: a round-5 file, which
        did not record it -- FASTVLA_TRAIN_TOWER decides then, as before)."""
        head_numel = sum(p.numel() for p in self.model.head_parameters())
        if self._unfrozen is None and (train_tower is not None or m.numel() > 2 * head_numel):
            # moments of a whole-backbone run (training/unfrozen.py writes one flat m / v over every trainable tensor): the run resumes unfrozen,
            # training what the checkpointed run trained
            self.enable_backbone_training(tower=train_tower)
        if self._unfrozen is not None:
            u = self._unfrozen
            if m.numel() != u.m.numel():
                raise ValueError(f"optimizer state has {m.numel()} elements, the trainable tensors of this run {u.m.numel()} (tower trained in one run and frozen in the other?)")
            u.m.copy_(m.to(u.m.device))
            u.v.copy_(v.to(u.v.device))
            u.step_count = int(step)
            self._opt_state["step"] = int(step)
            if flat is not None:      # the fp32 master of the run being resumed (Trainer._save_checkpoint): parameters continue bit for bit
                if flat.numel() != u.flat.numel():
                    raise ValueError(f"checkpointed master has {flat.numel()} elements, this run's {u.flat.numel()}")
                u.flat.copy_(flat.to(u.flat.device))
                u.eng.train_commit(u.flat)
                bb = self.model.backbone
                bb.clear_prefix_cache()
                bb.clear_prompt_cache()
            return
        flat = self.model._flat
        if flat is None:
            self._opt_state = {"resume": {"m": m, "v": v, "step": int(step)}}
            return
        st = self._optimizer_state(flat)
        st["m"].copy_(m.to(flat.device))
        st["v"].copy_(v.to(flat.device))
        st["step"] = int(step)

    def fused_train_step(self, batch: Optional[Dict] = None, *, lr: float, betas=(0.9, 0.95), eps: float = 1e-8,
                         weight_decay: float = 1e-4, max_grad_norm: Optional[float] = 1.0, process_group=None,
                         prepared: Optional[Dict] = None, next_batch: Optional[Dict] = None,
                         grad_accum_steps: int = 1, force_sync: bool = False) -> Dict[str, torch.Tensor]:
        """One pass of the step body of reference training/trainer.py:171-182 (loss -> backward -> [accumulate] -> clip ->
        AdamW), entirely on the HIP path.

        * grad_accum_steps = k: the flat gradient is summed over k calls (fv_grad_accumulate); the exchange, the clip and
          the optimiser run on every k-th call -- or when `force_sync` says the loader is exhausted -- with the 1/k of
          accelerate's `accumulate` folded into the optimiser's grad_scale (trainer.py:96,171).
        * under torch.distributed the accumulated gradient is summed across ranks with ONE all-reduce on a side stream
          (1/world folded into grad_scale as well).  When `next_batch` is given, ITS frozen backbone forward is enqueued
          between the start of the all-reduce and the optimiser kernel, so the collective runs underneath it; the
          prepared batch comes back under "next" and is passed as `prepared=` to the following call.
        """
        if self._unfrozen is None and not self.config.freeze_backbone and os.environ.get("FASTVLA_TRAIN_BACKBONE", "0") == "1":
            self.enable_backbone_training()
        if self._unfrozen is not None:
            out = self._unfrozen.step(batch, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, max_grad_norm=max_grad_norm,
                                      process_group=process_group, prepared=prepared, grad_accum_steps=grad_accum_steps, force_sync=force_sync)
            if next_batch is not None:
                # Trainer's one batch of look-ahead (training/trainer.py _train_one_epoch): an unfrozen forward depends on the update, so only the
                # parameter-INDEPENDENT half of the next batch is prepared here, AFTER the commit (image prep, tokenisation, the tower while it is frozen)
                out["next"] = self._unfrozen.prepare(next_batch)
            return out
        m = self.model
        prep = prepared if prepared is not None else self.prepare_batch(batch)
        dev = prep["pooled"].device
        eng, flat = m._engine(), m.materialize(dev)
        st = self._optimizer_state(flat)
        k = max(1, int(grad_accum_steps))
        st["micro"] += 1
        sync = force_sync or st["micro"] % k == 0
        p = float(self.config.dropout) if self.training else 0.0
        m._drop_calls += 1
        actions, saved = eng.head_forward(flat, prep["pooled"], prep["states"], training=p > 0.0, dropout_p=p,
                                          seed=m._drop_seed, offset=m._drop_calls, normalized_actions=True)
        if k > 1 and st["acc"] is None:
            st["acc"] = torch.zeros_like(flat)
        first = st["micro"] == 1  # first micro-batch of an accumulation window: the backward writes the window's buffer
        target_buf = st["g"] if k == 1 else (st["acc"] if first else st["g"])
        loss, grads = eng.head_backward(flat, actions, prep["targets"], saved, dropout_p=p, flat_grads=target_buf)
        if k > 1 and not first:
            eng.grad_accumulate(st["acc"], grads)
        total = st["g"] if k == 1 else st["acc"]
        out = {"loss": loss[0], "mse": loss[0].detach(), "actions": actions, "synced": sync, "next": None}
        scale = 1.0
        if sync:
            st["exchange"].group = process_group
            scale = st["exchange"].start(total) / k       # all-reduce launched on the side stream
        if next_batch is not None:
            out["next"] = self.prepare_batch(next_batch)  # frozen forward of batch k+1, underneath the collective
        if sync:
            st["exchange"].finish(dev)
            st["step"] += 1
            st["micro"] = 0
            eng.adamw_step(flat, total, st["m"], st["v"], st["step"], lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                           max_grad_norm=max_grad_norm or 0.0, grad_scale=scale, grad_norm_out=st["norm"])
        out["grad_norm"] = st["norm"][0]
        return out
