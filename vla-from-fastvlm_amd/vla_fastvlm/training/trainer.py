"""Native trainer for the FastVLA head on the HIP path.

Keeps the reference's `TrainingConfig` fields and `Trainer(model, train_dl, eval_dl, config).fit()` surface
(src/vla_fastvlm/training/trainer.py:20-51,145-166) but not its machinery: there is no `accelerate`; one process per
GPU under torch.distributed (RCCL), batches sharded round-robin by rank, and the step body of trainer.py:171-182
(loss -> backward -> clip_grad_norm_ -> AdamW.step -> LambdaLR.step) is ONE call into libfastvla_hip.so
(`FastVLAPolicy.fused_train_step`): head forward, MSE, head backward, [gradient accumulation,] all-reduce of the flat
12 MB gradient on a side stream, fused clip + AdamW.  The frozen backbone never sees a gradient (reference
fastvlm_adapter.py:501 wraps it in no_grad unconditionally), so this IS the whole trainable path -- and because it is
frozen, batch k+1's backbone forward is enqueued before the optimiser waits on batch k's all-reduce (the loop looks one
batch ahead), which is how the collective overlaps with compute here.

What accelerate/DDP did implicitly for the reference and is explicit here: rank 0's head parameters (and, on resume, the
AdamW moments) are broadcast at start-up; every rank runs the same number of steps (dp.shard_batches).
"""
from __future__ import annotations

import json
import logging
import os
from dataclasses import asdict, dataclass, field
from pathlib import Path
from typing import Dict, Iterable, Optional

import torch

from ..device import move_batch_to_device
from .dp import broadcast_flat, shard_batches

logger = logging.getLogger(__name__)


@dataclass
class TrainingConfig:
    """Fields and defaults of reference training/trainer.py:20-39.  With `gradient_accumulation_steps` = k > 1 the step
    counters keep the reference's meaning: its LambdaLR is NOT passed through accelerator.prepare, so the schedule advances
    once per MICRO-batch (LR index = global_step, trainer.py:180,182), and `max_steps` / `num_training_steps` are compared
    with global_step too (trainer.py:160,203); only the optimiser update itself waits for the k-th micro-batch."""
    output_dir: str = "outputs/train"
    num_epochs: int = 10
    max_steps: int | None = None
    gradient_accumulation_steps: int = 1
    learning_rate: float = 3e-4
    weight_decay: float = 0.01
    betas: tuple[float, float] = (0.9, 0.95)
    eps: float = 1e-8
    warmup_ratio: float = 0.03
    max_grad_norm: float = 1.0
    mixed_precision: str | None = "bf16"  # the HIP path is always bf16 MFMA / fp32 accumulate; kept for CLI parity
    logging_steps: int = 50
    eval_steps: int = 500
    save_steps: int = 1000
    seed: int = 42
    resume_from: str | None = None
    gradient_checkpointing: bool = False
    report_to: list[str] = field(default_factory=lambda: ["tensorboard"])


def linear_warmup_decay(step: int, total_steps: int, warmup_ratio: float) -> float:
    """LR multiplier of reference trainer.py:233-244."""
    warm = int(total_steps * warmup_ratio)
    if step < warm:
        return step / max(1, warm)
    return max(0.0, (total_steps - step) / max(1, total_steps - warm))


class Trainer:
    def __init__(self, model: torch.nn.Module, train_dataloader: Iterable[Dict], eval_dataloader: Optional[Iterable[Dict]] = None,
                 config: TrainingConfig | None = None) -> None:
        import torch.distributed as dist
        self.config = config or TrainingConfig()
        if self.config.gradient_accumulation_steps < 1:
            raise ValueError("gradient_accumulation_steps must be >= 1")
        torch.manual_seed(self.config.seed)
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.cuda.set_device(self.local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
        self.device = torch.device("cuda", self.local_rank)
        self.model = model
        self.train_dataloader = train_dataloader
        self.eval_dataloader = eval_dataloader
        self.num_training_steps = self._compute_total_training_steps()
        self.global_step = 0     # micro-batches seen: the reference's only counter (trainer.py:182) -- LR index, max_steps, logging
        self.update_step = 0     # optimiser updates actually applied (Adam's bias-correction index)
        self.epoch = 0
        self.last_lr = 0.0
        self._resume_opt = None

    @property
    def is_main_process(self) -> bool:
        return self.rank == 0

    def _shard(self, loader):
        return shard_batches(loader, self.rank, self.world)

    def _compute_total_training_steps(self) -> int:
        if self.config.max_steps:
            return self.config.max_steps
        if hasattr(self.train_dataloader, "__len__") and len(self.train_dataloader) > 0:
            per_epoch = max(len(self.train_dataloader) // self.world // self.config.gradient_accumulation_steps, 1)
            return per_epoch * self.config.num_epochs  # optimiser updates, as reference trainer.py:223-231
        raise ValueError("Unable to infer total training steps from dataloader; please set max_steps.")

    def fit(self) -> None:
        out = Path(self.config.output_dir)
        if self.is_main_process:
            (out / "checkpoints").mkdir(parents=True, exist_ok=True)
            (out / "logs").mkdir(exist_ok=True)
            (out / "training_config.json").write_text(json.dumps(asdict(self.config), indent=2))
        if self.config.resume_from:
            self._load_checkpoint(self.config.resume_from)
        self._sync_replicas()
        for epoch in range(self.config.num_epochs):
            self.epoch = epoch
            self._train_one_epoch()
            if self.global_step >= self.num_training_steps:   # reference trainer.py:160
                break

    def _sync_replicas(self) -> None:
        """Put the head in its flat device buffer, restore the optimiser state of a resumed run, and make every rank start
        from rank 0's parameters / moments (the caller builds the model before any seed is set, so replicas may differ)."""
        core = getattr(self.model, "model", None)
        if core is None or not hasattr(core, "materialize"):
            return
        flat = core.materialize(self.device)
        if self._resume_opt is not None and hasattr(self.model, "load_optimizer_state"):
            self.model.load_optimizer_state(self._resume_opt["m"], self._resume_opt["v"], int(self._resume_opt["step"]), flat=self._resume_opt.get("flat"),
                                            train_tower=self._resume_opt.get("train_tower"))
            self._resume_opt = None
        un = getattr(self.model, "_unfrozen", None)
        if un is not None:
            flat = un.flat        # an unfrozen run: the replicas share the WHOLE master (head + projector + decoder [+ tower]), not the head alone
        if self.world > 1:
            broadcast_flat(flat)
            st = getattr(self.model, "_opt_state", None)
            if st and "m" in st:
                broadcast_flat(st["m"])
                broadcast_flat(st["v"])

    def _train_one_epoch(self) -> None:
        cfg = self.config
        k = cfg.gradient_accumulation_steps
        self.model.train()
        it = iter(self._shard(self.train_dataloader))
        nxt = next(it, None)
        prepared = None
        while nxt is not None:
            batch = move_batch_to_device(nxt, self.device) if prepared is None else None
            nxt = next(it, None)  # one batch of look-ahead: "is this the last one" and the overlap partner of the all-reduce
            stop_after = bool(cfg.max_steps and self.global_step + 1 >= cfg.max_steps)   # reference trainer.py:203: micro-batches
            # the reference's LambdaLR steps every micro-batch (it is not wrapped by accelerate): index = global_step
            self.last_lr = cfg.learning_rate * linear_warmup_decay(self.global_step, self.num_training_steps, cfg.warmup_ratio)
            out = self.model.fused_train_step(batch, prepared=prepared, lr=self.last_lr, betas=cfg.betas, eps=cfg.eps,
                                              weight_decay=cfg.weight_decay, max_grad_norm=cfg.max_grad_norm,
                                              grad_accum_steps=k, force_sync=nxt is None,  # accelerate syncs at the end of the loader
                                              next_batch=move_batch_to_device(nxt, self.device) if nxt is not None and not stop_after else None)
            prepared = out["next"]
            self.global_step += 1
            if out["synced"]:
                self.update_step += 1
                gn = out["grad_norm"]   # a device scalar on the HIP path (kept as one: no host sync here); a norm exists only for micro-batches that closed an update
                self._last_grad_norm = gn.detach().clone() if torch.is_tensor(gn) else float(gn)
            if self.is_main_process and self.global_step % cfg.logging_steps == 0:
                rec = {"train/loss": float(out["loss"]), "train/mse": float(out["mse"]), "train/lr": self.last_lr, "train/epoch": self.epoch}
                if getattr(self, "_last_grad_norm", None) is not None:   # the LAST closed update's norm, also on logging steps that fall between two updates
                    rec["train/grad_norm"] = float(self._last_grad_norm)
                self._log(rec)
            if self.global_step % cfg.eval_steps == 0 and self.eval_dataloader is not None:
                metrics = self.evaluate()
                if self.is_main_process:
                    self._log(metrics)
            if self.global_step % cfg.save_steps == 0:
                self._save_checkpoint(f"step-{self.global_step}")
            if stop_after:
                break

    def _log(self, metrics: Dict[str, float]) -> None:
        rec = {"step": self.global_step, **metrics}
        logger.info(json.dumps(rec))
        with open(Path(self.config.output_dir) / "logs" / "metrics.jsonl", "a", encoding="utf-8") as f:
            f.write(json.dumps(rec) + "\n")

    @torch.no_grad()
    def evaluate(self) -> Dict[str, float]:
        if self.eval_dataloader is None:
            return {}
        self.model.eval()
        total, count = 0.0, 0
        for batch in self.eval_dataloader:
            batch = move_batch_to_device(batch, self.device)
            n = batch["actions"].shape[0]
            total += float(self.model.compute_loss(batch)["mse"]) * n
            count += n
        self.model.train()
        return {"eval/mse": total / max(count, 1)}

    # Extension of this build (an attribute, not a TrainingConfig field: the dataclass is the reference's contract): True writes the frozen
    # VLM tensors into policy_state_dict.pt under `model.backbone.model.*` the way the reference's state_dict() does (trainer.py:255), so
    # its own strict loader accepts the file; False (default, or FASTVLA_SAVE_BACKBONE unset) keeps checkpoints at the head's 12 MB.
    save_backbone_weights: bool = os.environ.get("FASTVLA_SAVE_BACKBONE", "0") == "1"

    def _save_checkpoint(self, suffix: str) -> None:
        """Same files as reference trainer.py:246-255: policy_config.json + policy_state_dict.pt (head tensors under the
        reference's `model.*` keys; the frozen backbone lives in the library and is not duplicated)."""
        if not self.is_main_process:
            return
        from ..utils.checkpoint import save_policy_checkpoint
        # an unfrozen run's checkpoint without the VLM would lose what was trained: the backbone always travels then
        d = save_policy_checkpoint(self.model, Path(self.config.output_dir) / "checkpoints" / suffix,
                                   include_backbone=self.save_backbone_weights or getattr(self.model, "_unfrozen", None) is not None)
        st = getattr(self.model, "_opt_state", None)
        if st is not None:
            rec = {"m": st["m"].cpu(), "v": st["v"].cpu(), "step": st["step"], "global_step": self.global_step, "update_step": self.update_step}
            un = getattr(self.model, "_unfrozen", None)
            if un is not None:
                # the fp32 MASTER of an unfrozen run: the VLM tensors of policy_state_dict.pt come back through the engine's bf16 operand copies, and a master
                # rebuilt from those has lost the low bits every later update (~1e-3 of a bf16 ulp) lives in
                rec["flat"] = un.flat.cpu()
                rec["train_backbone"], rec["train_tower"] = True, bool(un.train_tower)   # what the run trains comes back from the checkpoint, not from the environment
            torch.save(rec, d / "optimizer.pt")

    def _load_checkpoint(self, path: str) -> None:
        p = Path(path)
        if not p.exists():
            raise FileNotFoundError(f"Checkpoint path {path} does not exist.")
        state = torch.load(p / "policy_state_dict.pt", map_location="cpu")
        from ..utils.checkpoint import BACKBONE_PREFIX, EXTRA_STATE_MARKS
        vlm = {k[len(BACKBONE_PREFIX):]: v for k, v in state.items() if k.startswith(BACKBONE_PREFIX)}
        if vlm:
            # the VLM the checkpoint carries (always, for an unfrozen run) replaces the one `vlm_model_name` resolves to; a running unfrozen state
            # mirrors the OLD weights in its master, so it is rebuilt from the restored ones
            un = getattr(self.model, "_unfrozen", None)
            self.model._unfrozen = None
            self.model.model.backbone.load_backbone_state(vlm)
            if un is not None:
                self.model.enable_backbone_training(tower=un.train_tower)
        own = self.model.state_dict()
        # `.io_norm.` / splice-mode keys exist in state_dict() only while they are on, so a freshly built model does not list them -- let them
        # through (FastVLMBackbone._load_from_state_dict re-applies them)
        self.model.load_state_dict({k: v for k, v in state.items() if k in own or any(m in k for m in EXTRA_STATE_MARKS)}, strict=False)
        from ..utils.checkpoint import read_extras
        ex = read_extras(p)
        if "splice_image_tokens" in ex:
            self.model.model.backbone.splice_image_tokens = bool(ex["splice_image_tokens"])
        if (p / "optimizer.pt").is_file():
            self._resume_opt = torch.load(p / "optimizer.pt", map_location="cpu")  # applied by _sync_replicas()
            self.global_step = int(self._resume_opt.get("global_step", 0))
            self.update_step = int(self._resume_opt.get("update_step", self._resume_opt.get("step", 0)))
