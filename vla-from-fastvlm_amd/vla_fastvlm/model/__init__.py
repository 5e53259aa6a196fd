"""Backbone adapter (reference: src/vla_fastvlm/model/__init__.py)."""
from .fastvlm_adapter import FastVLMBackbone, FastVLMBackboneConfig, resize_with_pad

__all__ = ["FastVLMBackbone", "FastVLMBackboneConfig", "resize_with_pad"]
