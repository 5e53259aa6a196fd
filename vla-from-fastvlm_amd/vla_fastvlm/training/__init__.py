"""Training loop (reference: src/vla_fastvlm/training/__init__.py)."""
from .trainer import Trainer, TrainingConfig

__all__ = ["Trainer", "TrainingConfig"]
