"""Standalone FastVLA policy surface (reference: src/vla_fastvlm/fastvla/__init__.py:1-7)."""
from .configuration_fastvla import FastVLAConfig
from .modeling_fastvla import FastVLAPolicy

__all__ = ["FastVLAConfig", "FastVLAPolicy"]
