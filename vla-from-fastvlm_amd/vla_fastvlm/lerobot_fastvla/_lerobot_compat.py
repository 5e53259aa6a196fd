"""Bindings to LeRobot, with minimal stand-ins when `lerobot` is not installed (it is absent from the build image).

With LeRobot present (README.md:35 of the reference pins 0.4.4) the real classes are used and the plugin registers
itself as policy.type=fastvla exactly like the reference does.  Without it, the stand-ins below keep the plugin
importable and unit-testable; they implement only what FastVLAPolicy / FastVLAConfig touch.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

from torch import nn

try:  # pragma: no cover - exercised only where lerobot exists
    from lerobot.configs.policies import PreTrainedConfig
    from lerobot.configs.types import FeatureType, NormalizationMode, PolicyFeature
    from lerobot.optim.optimizers import AdamWConfig
    from lerobot.optim.schedulers import CosineDecayWithWarmupSchedulerConfig
    from lerobot.policies.pretrained import PreTrainedPolicy
    from lerobot.utils.constants import ACTION
    HAVE_LEROBOT = True
except Exception:
    HAVE_LEROBOT = False
    ACTION = "action"

    class FeatureType(str, enum.Enum):
        STATE = "STATE"
        VISUAL = "VISUAL"
        ENV = "ENV"
        ACTION = "ACTION"

    class NormalizationMode(str, enum.Enum):
        MIN_MAX = "MIN_MAX"
        MEAN_STD = "MEAN_STD"
        IDENTITY = "IDENTITY"

    @dataclass
    class PolicyFeature:
        type: FeatureType
        shape: tuple

    @dataclass
    class AdamWConfig:
        lr: float = 1e-3
        betas: tuple = (0.9, 0.999)
        eps: float = 1e-8
        weight_decay: float = 1e-2
        grad_clip_norm: float = 10.0

    @dataclass
    class CosineDecayWithWarmupSchedulerConfig:
        peak_lr: float = 1e-4
        decay_lr: float = 2.5e-6
        num_warmup_steps: int = 500
        num_decay_steps: int = 20_000

    @dataclass
    class PreTrainedConfig:
        n_obs_steps: int = 1
        input_features: Dict[str, PolicyFeature] = field(default_factory=dict)
        output_features: Dict[str, PolicyFeature] = field(default_factory=dict)
        device: Optional[str] = None
        _registry = {}

        def __post_init__(self):
            pass

        @classmethod
        def register_subclass(cls, name: str):
            def deco(sub):
                cls._registry[name] = sub
                return sub
            return deco

        @property
        def action_feature(self):
            for ft in (self.output_features or {}).values():
                if ft.type is FeatureType.ACTION:
                    return ft
            return None

    class PreTrainedPolicy(nn.Module):
        config_class: Any = None
        name: str = ""

        def __init__(self, config, *args, **kwargs):
            super().__init__()
            self.config = config
