"""LeRobot policy wrapper around FastVLMWithExpert (reference:
src/vla_fastvlm/lerobot_fastvla/modeling_fastvla.py:19-133): first VISUAL key, last timestep, task -> list[str] (+"\\n"),
forward(batch) -> (loss, {"loss","mse"}), predict_action_chunk -> [B,1,A], select_action with the action deque."""
from __future__ import annotations

from collections import deque
from typing import Any

import torch
from torch import Tensor

from ..fastvla.configuration_fastvla import FastVLAConfig as CoreFastVLAConfig
from ..fastvla.fastvlm_with_expert import FastVLMWithExpert
from ._lerobot_compat import ACTION, FeatureType, PreTrainedPolicy
from .configuration_fastvla import FastVLAConfig

_CORE_FIELDS = ("vlm_model_name", "bootstrap_model_name", "state_dim", "action_dim", "hidden_dim", "fusion_dim", "dropout",
                "freeze_backbone", "tokenizer_max_length", "tokenizer_padding_side", "pad_to_max_length",
                "resize_with_padding", "image_size", "pad_value", "add_trailing_newline")


class FastVLAPolicy(PreTrainedPolicy):
    config_class = FastVLAConfig
    name = "fastvla"

    def __init__(self, config: FastVLAConfig, **kwargs: Any):
        super().__init__(config)
        config.validate_features()
        self.config = config
        self._state_key, self._image_keys = self._resolve_input_keys()
        self._infer_io_dims_from_features()
        self.model = FastVLMWithExpert(CoreFastVLAConfig(**{k: getattr(config, k) for k in _CORE_FIELDS}))
        self.reset()

    def _resolve_input_keys(self) -> tuple[str, list[str]]:
        feats = self.config.input_features
        if not feats:
            raise ValueError("FastVLA requires input_features to be set.")
        states = [k for k, ft in feats.items() if ft.type is FeatureType.STATE]
        images = [k for k, ft in feats.items() if ft.type is FeatureType.VISUAL]
        if not states:
            raise ValueError("No state feature found in input_features.")
        if not images:
            raise ValueError("No visual feature found in input_features.")
        return states[0], images

    def _infer_io_dims_from_features(self) -> None:
        feats = self.config.input_features
        if feats and self._state_key in feats:
            self.config.state_dim = feats[self._state_key].shape[0]
        if self.config.action_feature is not None:
            self.config.action_dim = self.config.action_feature.shape[0]

    def get_optim_params(self):
        return self.parameters()

    def reset(self):
        self._action_queue: deque[Tensor] = deque([], maxlen=self.config.n_action_steps)

    def fold_dataset_stats(self, dataset_stats) -> None:
        """Take over the STATE / ACTION MEAN_STD normalisation LeRobot's processors would apply around the policy
        (reference lerobot_fastvla/processor_fastvla.py:34-48 with the map of configuration_fastvla.py:21-27): raw states go
        in, un-normalised actions come out of `select_action`, with the arithmetic folded into the head kernels.
        `forward(batch)` still expects NORMALISED action targets (the loss lives in normalised space).  None switches it off."""
        if dataset_stats is None:
            self.model.backbone.set_io_normalization()
            return
        act_key = next((k for k, ft in (self.config.output_features or {}).items() if ft.type is FeatureType.ACTION), ACTION)
        st, ac = dataset_stats[self._state_key], dataset_stats[act_key]
        self.model.backbone.set_io_normalization(state_mean=st["mean"], state_std=st["std"], action_mean=ac["mean"], action_std=ac["std"])

    def _prepare_inputs(self, batch: dict[str, Tensor]) -> tuple[Tensor, Tensor, list[str]]:
        images = batch[self._image_keys[0]]  # only the first camera feeds the backbone
        if images.ndim == 5:
            images = images[:, -1]
        states = batch[self._state_key]
        if states.ndim == 3:
            states = states[:, -1]
        n = images.shape[0]
        task = batch.get("task")
        if task is None:
            tasks = [""] * n
        elif isinstance(task, (list, tuple)):
            tasks = [str(t) for t in task]
            tasks = tasks * n if len(tasks) == 1 and n > 1 else tasks
        else:
            tasks = [str(task)] * n
        if self.config.add_trailing_newline:
            tasks = [t if t.endswith("\n") else t + "\n" for t in tasks]
        return images, states, tasks

    def _predict_actions(self, batch: dict[str, Tensor]) -> Tensor:
        images, states, tasks = self._prepare_inputs(batch)
        return self.model(images, states, tasks, device=images.device)

    @torch.no_grad()
    def predict_action_chunk(self, batch: dict[str, Tensor]) -> Tensor:
        self.eval()
        return self._predict_actions(batch).unsqueeze(1)  # [B, chunk=1, A]

    @torch.no_grad()
    def select_action(self, batch: dict[str, Tensor]) -> Tensor:
        self.eval()
        if not self._action_queue:
            chunk = self.predict_action_chunk(batch)[:, : self.config.n_action_steps]
            self._action_queue.extend(chunk.transpose(0, 1))
        return self._action_queue.popleft()

    def forward(self, batch: dict[str, Tensor]) -> tuple[Tensor, dict]:
        images, states, tasks = self._prepare_inputs(batch)
        gt = batch[ACTION]
        if gt.ndim == 3:
            gt = gt[:, 0]
        # head forward + MSE (+ the head gradients autograd will ask for) in one pass through the library
        loss, _pred = self.model.forward_loss(images, states, tasks, gt, device=images.device)
        val = loss.item()
        return loss, {"loss": val, "mse": val}
