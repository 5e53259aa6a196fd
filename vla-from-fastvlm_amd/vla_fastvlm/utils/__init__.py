"""Small host utilities (reference: src/vla_fastvlm/utils/__init__.py)."""
from .checkpoint import load_policy_from_checkpoint, save_policy_checkpoint
from .logging import configure_logging

__all__ = ["configure_logging", "load_policy_from_checkpoint", "save_policy_checkpoint"]
