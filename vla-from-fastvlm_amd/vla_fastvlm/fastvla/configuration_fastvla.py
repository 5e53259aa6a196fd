"""Core policy configuration.  Field names and defaults are the public contract of the reference
(src/vla_fastvlm/fastvla/configuration_fastvla.py:9-46); `dataclasses.asdict(cfg)` must keep producing the same
`policy_config.json` (reference training/trainer.py:250-253, utils/checkpoint.py:29-35)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

from ..model.fastvlm_adapter import FastVLMBackboneConfig


@dataclass
class FastVLAConfig:
    # backbone
    vlm_model_name: str = "apple/FastVLM-0.5B"
    bootstrap_model_name: str = "apple/FastVLM-0.5B"
    # action expert
    state_dim: int = 14
    action_dim: int = 14
    hidden_dim: int = 1024
    fusion_dim: int = 1024
    dropout: float = 0.1
    freeze_backbone: bool = True
    # text / image preparation
    tokenizer_max_length: int = 64
    tokenizer_padding_side: str = "right"
    pad_to_max_length: bool = False
    resize_with_padding: bool = True
    image_size: Optional[int] = None
    pad_value: float = 0.0
    add_trailing_newline: bool = True

    def to_backbone_config(self) -> FastVLMBackboneConfig:
        return FastVLMBackboneConfig(
            model_id=self.vlm_model_name, bootstrap_model_id=self.bootstrap_model_name,
            freeze_backbone=self.freeze_backbone, force_image_size=self.image_size,
            resize_with_padding=self.resize_with_padding, pad_value=self.pad_value,
            tokenizer_max_length=self.tokenizer_max_length, tokenizer_padding_side=self.tokenizer_padding_side,
            pad_to_max_length=self.pad_to_max_length)
