"""Device selection helpers (reference: src/vla_fastvlm/device.py:9-56).  ROCm shows up as `cuda` in torch."""
from __future__ import annotations

import os
from typing import Any, Dict, Optional

import torch


def _forced_cpu() -> bool:
    return os.environ.get("FASTVLM_FORCE_DEVICE", "").lower() == "cpu"


def is_cuda_available() -> bool:
    return (not _forced_cpu()) and torch.cuda.is_available()


def is_mps_available() -> bool:
    return (not _forced_cpu()) and torch.backends.mps.is_available()


def get_best_device(preferred: Optional[str] = None) -> torch.device:
    want = preferred.lower() if preferred else None
    ranked = [("cuda", is_cuda_available), ("mps", is_mps_available)]
    for name, ok in ranked:
        if want == name and ok():
            return torch.device(name)
    for name, ok in ranked:
        if ok():
            return torch.device(name)
    return torch.device("cpu")


def move_batch_to_device(batch: Dict[str, Any], device: torch.device) -> Dict[str, Any]:
    moved: Dict[str, Any] = {}
    for key, val in batch.items():
        if torch.is_tensor(val):
            moved[key] = val.to(device)
        elif isinstance(val, dict):
            moved[key] = move_batch_to_device(val, device)
        else:
            moved[key] = val
    return moved
