"""Input normalisation in front of the policy (reference: src/vla_fastvlm/fastvla/processor_fastvla.py:11-43):
task broadcast + trailing newline, last-timestep slicing of time-major stacks, image prep delegated to the backbone
(which runs it on the HIP device and returns an already-prepared pixel tensor)."""
from __future__ import annotations

from typing import List, Sequence, Union

import torch

from .configuration_fastvla import FastVLAConfig


class FastVLAProcessor:
    def __init__(self, config: FastVLAConfig, backbone) -> None:
        self.config = config
        self.backbone = backbone

    def normalize_tasks(self, tasks: Union[Sequence[str], str], batch_size: int) -> List[str]:
        items = [tasks] if isinstance(tasks, str) else list(tasks)
        if len(items) == 1 and batch_size > 1:
            items = items * batch_size
        if self.config.add_trailing_newline:
            items = [t if t.endswith("\n") else t + "\n" for t in items]
        return items

    prepare_tasks = normalize_tasks

    def prepare_images(self, images: torch.Tensor, device: torch.device) -> torch.Tensor:
        if images.ndim == 5:  # (B,T,C,H,W): keep the most recent frame
            images = images[:, -1]
        return self.backbone._prepare_images_tensor(images, device)

    def prepare_states(self, states: torch.Tensor, device: torch.device) -> torch.Tensor:
        if states.ndim == 3:  # (B,T,D)
            states = states[:, -1]
        return states.to(device)
