"""Unfrozen-backbone training on the HIP path: decoder + mm_projector + action expert trainable, FastViT-HD
tower frozen, image tokens spliced in front of the text.

The reference exposes `freeze_backbone` (fastvla/configuration_fastvla.py:23, applied at model/fastvlm_adapter.py:170-173) but wraps the
backbone forward in an unconditional `@torch.no_grad()` (model/fastvlm_adapter.py:501), so its own loop (training/trainer.py:171-182)
only ever trains the head.  Because of that, `freeze_backbone=False` ALONE keeps the reference's behaviour here too (frozen backbone, head
training); the VLM is fine-tuned only after an explicit `policy.enable_backbone_training()` (or FASTVLA_TRAIN_BACKBONE=1 together with
`freeze_backbone=False`).  The step body is the reference's -- loss -> backward -> clip_grad_norm_(1.0) over ALL parameters -> AdamW ->
schedule -- with every kernel in libfastvla_hip.so (fv_train_forward_backward, fv_adamw_clip_step, fv_train_commit).

All trainable tensors live in ONE flat fp32 buffer (fv_train_layout: [head | projector | embedding | layers | final norm], matrices in the
library's packed layout); gradients, Adam's m and v mirror it.  Under torch.distributed the gradient is exchanged per BUCKET while the
backward pass is still running (training/dp.py BucketedGradExchange, driven by the library's fv_bucket_cb).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from fastvla_hip import HEAD_KEYS

from .dp import BucketedGradExchange, GradExchange


class UnfrozenState:
    def __init__(self, policy, bucket_min_numel: int = 1 << 22, train_tower: bool = False):
        m = policy.model
        bb = m.backbone
        self.policy = policy
        eng = bb.engine()
        if eng.llm_precision != 1:
            raise RuntimeError("backbone training needs the split-bf16 decoder policy (llm_precision=1): its bf16 weight copies are refreshed "
                               f"from the fp32 master after every step; this engine runs llm_precision={eng.llm_precision}")
        eng.train_begin()
        self.train_tower = bool(train_tower)
        if self.train_tower:
            eng.train_tower_begin()        # the FastViT-HD tower's inference-form tensors join the flat master (fv_train_tower_*)
        self._tws: Dict[int, torch.Tensor] = {}
        self._dto: Dict[int, torch.Tensor] = {}
        self.eng = eng
        self.tensors, self.total, self.n_buckets = eng.train_layout()
        dev = eng.device
        self.flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        eng.train_export_params(self.flat)
        # the 12 head tensors move into the front of the flat buffer and the nn.Parameters are re-pointed at it, so state_dict(), a torch
        # optimiser or LeRobot's checkpointing keep seeing the live values
        old = m.materialize(dev)
        hn = eng.head_numel()
        self.flat[:hn].copy_(old)
        views = eng.head_views(self.flat[:hn])
        with torch.no_grad():
            for p, k in zip(m.head_parameters(), HEAD_KEYS):
                p.data = views[k]
        m._flat = self.flat[:hn]
        self.g = torch.zeros_like(self.flat)
        self.acc: Optional[torch.Tensor] = None
        self.m, self.v = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self.step_count, self.micro = 0, 0
        self.norm = torch.zeros(1, device=dev)
        self.loss_scale_log2 = 12
        self.saturation_check_every = 50
        self.bucketed = BucketedGradExchange(dev, min_numel=bucket_min_numel)
        self.whole = GradExchange(dev)
        self._ws: Dict[tuple, torch.Tensor] = {}
        bb.splice_image_tokens = True          # the model being trained IS the spliced one: inference must run the same graph
        bb._trained_tensors = self.named_backbone_tensors   # checkpoint export reads the master, not the original weight source
        pending = policy._opt_state if isinstance(policy._opt_state, dict) and "resume" in policy._opt_state else None
        policy._opt_state = {"m": self.m, "v": self.v, "step": 0, "flat": self.flat, "norm": self.norm}
        if pending is not None and pending["resume"]["m"].numel() == self.m.numel():   # load_optimizer_state() ran before this state existed
            r = pending["resume"]
            self.m.copy_(r["m"].to(dev))
            self.v.copy_(r["v"].to(dev))
            self.step_count = int(r["step"])
            policy._opt_state["step"] = self.step_count

    # ------------------------------------------------------------------ views / export
    def named_backbone_tensors(self) -> Dict[str, torch.Tensor]:
        """canonical checkpoint key -> fp32 copy of every trained decoder / projector tensor (q / k / v and gate / up unpacked)"""
        out = {k: v for k, v in self.eng.train_named_tensors(self.flat).items() if not k.startswith("head.")}
        # a trained tower lives in the master in its inference form (every ConvFFN's BatchNorm folded into its 7x7): written back under the
        # checkpoint's own keys as that conv + an identity BatchNorm, which is what a loader (this build's or the reference's) folds to the same tensor
        bn_eps = float(self.eng.model.tower.bn_eps)
        for k in [k for k in out if k.endswith(".convffn.conv.folded.weight")]:
            pre = k[: -len("folded.weight")]
            w = out.pop(k)
            c = w.shape[0]
            out[pre + "conv.weight"] = w
            out[pre + "bn.weight"] = torch.ones(c, device=w.device)
            out[pre + "bn.bias"] = out.pop(pre + "folded.bias")
            out[pre + "bn.running_mean"] = torch.zeros(c, device=w.device)
            out[pre + "bn.running_var"] = torch.full((c,), 1.0 - bn_eps, device=w.device)
        return out

    def _workspace(self, B: int, T: int) -> torch.Tensor:
        key = (B, T)
        if key not in self._ws:
            self._ws.clear()                   # one shape at a time: the stash is ~0.5 GB per sample at FastVLM-0.5B
            self._ws[key] = self.eng.train_workspace(B, T)
        return self._ws[key]

    # ------------------------------------------------------------------ one step
    def prepare(self, batch: Dict) -> Dict:
        """everything that does not depend on the trainable parameters: image prep + the FROZEN tower, tokenisation"""
        pol, bb, eng = self.policy, self.policy.model.backbone, self.eng
        dev = eng.device
        images = pol.processor.prepare_images(batch["images"], dev)
        states = pol.processor.prepare_states(batch["states"], dev).float()
        tasks = pol.processor.prepare_tasks(batch["tasks"], batch_size=images.shape[0])
        targets = batch["actions"].to(dev, torch.float32)
        if targets.ndim == 3:
            targets = targets[:, 0]
        pix = bb._prepare_images_tensor(images, dev)
        tower_out = None
        if not self.train_tower:           # a trainable tower's forward depends on the parameters: it belongs to step()
            with torch.no_grad():
                _, tower_out = eng.vision_forward(pix, return_tower_out=True)
        text = bb._prep_text(tasks, dev)
        ids, mask = text["input_ids"], text["attention_mask"]
        T = ids.shape[1]
        Tp = (T + 7) // 8 * 8                  # the training kernels want whole 16-byte rows of ids: right-pad, masked
        if Tp != T:
            ids = torch.nn.functional.pad(ids, (0, Tp - T))
            mask = torch.nn.functional.pad(mask, (0, Tp - T))
        return {"tower_out": tower_out, "pix": pix if self.train_tower else None, "ids": ids, "lens": mask.to(torch.int32).sum(1).to(torch.int32), "states": states,
                "targets": targets.contiguous()}

    def step(self, batch: Optional[Dict] = None, *, lr: float, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 1e-4,
             max_grad_norm: Optional[float] = 1.0, process_group=None, prepared: Optional[Dict] = None, grad_accum_steps: int = 1,
             force_sync: bool = False) -> Dict[str, torch.Tensor]:
        pol, eng = self.policy, self.eng
        prep = prepared if prepared is not None else self.prepare(batch)
        B, T = prep["ids"].shape
        ws = self._workspace(B, T)
        k = max(1, int(grad_accum_steps))
        self.micro += 1
        sync = force_sync or self.micro % k == 0
        p = float(pol.config.dropout) if pol.training else 0.0
        m = pol.model
        m._drop_calls += 1
        overlap = k == 1 and sync              # per-bucket all-reduce under the backward pass; with accumulation the sum is exchanged once
        if overlap:
            self.bucketed.group = process_group
            self.bucketed.begin(self.g)
        tower_out, tws, dto = prep["tower_out"], None, None
        if self.train_tower:
            if B not in self._tws:
                self._tws.clear(); self._dto.clear()
                self._tws[B] = eng.train_tower_workspace(B)
                t = eng.model.tower
                self._dto[B] = torch.zeros(B, t.num_tokens, t.out_dim, dtype=torch.float16, device=eng.device)
            tws, dto = self._tws[B], self._dto[B]
            eng.train_set_tower_grad(dto)
            tower_out = eng.train_tower_forward(prep["pix"], tws)
        actions, loss, _ = eng.train_forward_backward(self.flat, tower_out, prep["ids"], prep["lens"], prep["states"], prep["targets"], ws,
                                                      training=pol.training, dropout_p=p, seed=m._drop_seed, offset=m._drop_calls, flat_grads=self.g,
                                                      bucket_cb=self.bucketed.bucket_ready if overlap else None)
        if self.train_tower:
            eng.train_tower_backward(prep["pix"], dto, tws, self.g, bucket_cb=self.bucketed.bucket_ready if overlap else None)
        total = self.g
        if k > 1:
            if self.acc is None:
                self.acc = torch.zeros_like(self.flat)
            if self.micro == 1:
                self.acc.copy_(self.g)
            else:
                eng.grad_accumulate(self.acc, self.g)
            total = self.acc
        out = {"loss": loss[0], "mse": loss[0].detach(), "actions": actions, "synced": sync, "next": None}
        if sync:
            if overlap:
                scale = self.bucketed.finish(eng.device)
            else:
                self.whole.group = process_group
                scale = self.whole.start(total) / k
                self.whole.finish(eng.device)
            self.step_count += 1
            self.micro = 0
            scale /= eng.train_loss_scale()      # every gradient of fv_train_forward_backward carries the loss scale (2^12 by default)
            eng.adamw_step(self.flat, total, self.m, self.v, self.step_count, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay,
                           max_grad_norm=max_grad_norm or 0.0, grad_scale=scale, grad_norm_out=self.norm)
            eng.train_commit(self.flat)       # bf16 operand copies (and their transposes) follow the master
            bb = pol.model.backbone
            bb.clear_prefix_cache()           # per-image decoder prefixes / per-prompt features computed with the OLD weights must not serve an eval between steps
            bb.clear_prompt_cache()
            pol._opt_state["step"] = self.step_count
            if self.step_count % self.saturation_check_every == 0:
                # the backward's fp16 operands carry the gradient x 2^loss_scale: a clamp means the scale is too large for this model / loss
                # -- say so and halve it (every 50 optimiser steps: the read synchronises)
                n = eng.fp16_saturations(reset=True)
                if n and self.loss_scale_log2 > 0:
                    import warnings
                    self.loss_scale_log2 -= 1
                    eng.train_set_options(loss_scale_log2=self.loss_scale_log2, keep=True)
                    warnings.warn(f"{n} fp16 gradient-operand groups saturated in the last {self.saturation_check_every} steps: loss scale lowered to 2^{self.loss_scale_log2}")
        out["grad_norm"] = self.norm[0]
        return out
