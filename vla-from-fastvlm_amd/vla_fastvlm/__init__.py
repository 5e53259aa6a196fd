"""vla_fastvlm -- drop-in host-side mirror of syun88/VLA-from-FastVLM's package for the MI355X-native path.

Same import surface as the reference (`vla_fastvlm.fastvla`, `vla_fastvlm.lerobot_fastvla`,
`vla_fastvlm.model.fastvlm_adapter`, `vla_fastvlm.training`), same config fields, state-dict keys and error
behaviour; the arithmetic runs in libfastvla_hip.so (hand-written HIP for gfx950) through `fastvla_hip`.
Put `vla-from-fastvlm_amd/` on PYTHONPATH ahead of the reference's `src/` and `scripts/train.py` /
`lerobot-train --policy.type=fastvla --policy.discover_packages_path=vla_fastvlm.lerobot_fastvla` pick this up unchanged.
(reference: src/vla_fastvlm/__init__.py:9-20)
"""
from .device import get_best_device, is_cuda_available, is_mps_available
from .fastvla import FastVLAConfig, FastVLAPolicy

__all__ = ["get_best_device", "is_cuda_available", "is_mps_available", "FastVLAConfig", "FastVLAPolicy"]
