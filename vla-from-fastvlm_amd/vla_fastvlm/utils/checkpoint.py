"""Load `policy_config.json` + `policy_state_dict.pt` (reference: src/vla_fastvlm/utils/checkpoint.py:14-47).
Only the 12 head tensors (`model.state_projection.*`, `model.fusion.*`, `model.action_head.*`) are consumed; a
reference checkpoint's `model.backbone.model.*` entries are ignored because the frozen backbone lives in the library."""
from __future__ import annotations

import json
from pathlib import Path

import torch

from ..fastvla import FastVLAConfig, FastVLAPolicy


def load_policy_from_checkpoint(checkpoint_dir: str, device: torch.device | None = None) -> FastVLAPolicy:
    root = Path(checkpoint_dir)
    cfg_path, sd_path = root / "policy_config.json", root / "policy_state_dict.pt"
    if not cfg_path.is_file() or not sd_path.is_file():
        raise FileNotFoundError(f"{checkpoint_dir} must contain policy_config.json and policy_state_dict.pt")
    payload = json.loads(cfg_path.read_text())
    if "vlm_model_name" not in payload:
        raise ValueError("legacy FastVLMPolicy checkpoints (nested backbone config) are not supported by the HIP path")
    policy = FastVLAPolicy(FastVLAConfig(**payload))
    state = torch.load(sd_path, map_location="cpu")
    own = policy.state_dict()
    missing = [k for k in own if k not in state]
    if missing:
        raise KeyError(f"checkpoint lacks head tensors: {missing}")
    policy.load_state_dict({k: state[k] for k in own})
    policy.eval()
    return policy
