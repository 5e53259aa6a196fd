"""Load / write `policy_config.json` + `policy_state_dict.pt` (reference: src/vla_fastvlm/utils/checkpoint.py:14-47 reads them,
training/trainer.py:246-255 writes them).

Read side.  The 12 head tensors (`model.state_projection.*`, `model.fusion.*`, `model.action_head.*`) always come from the file.  A
checkpoint written by the REFERENCE also carries the whole VLM under `model.backbone.model.*` (the backbone is a submodule there, so
`state_dict()` includes it, and `load_state_dict` puts it back over whatever `from_pretrained` loaded): those tensors are handed to the
backbone (`FastVLMBackbone.load_backbone_state`: canonical keys = the reference's keys minus the prefix, training-form towers folded,
`lm_head.*` dropped) and packed into the engine instead of the ones `vlm_model_name` resolves to.  A checkpoint without them (what this
build writes by default: the frozen backbone is not duplicated next to every head) keeps the `vlm_model_name` weights.

Write side.  `save_policy_checkpoint(policy, dir, include_backbone=True)` writes the file the reference's own
`load_policy_from_checkpoint(strict=True)` accepts: head tensors + every VLM tensor under `model.backbone.model.*` (+ the tied
`lm_head.weight`)."""
from __future__ import annotations

import dataclasses
import json
from pathlib import Path

import torch

from ..fastvla import FastVLAConfig, FastVLAPolicy

BACKBONE_PREFIX = "model.backbone.model."
# run properties of THIS build that are not tensors of the reference's module tree: written beside the two reference files, never into them, so that the
# reference's strict `load_state_dict` / `FastVLAConfig(**payload)` accept a checkpoint of an unfrozen run as it is.  Keys: "splice_image_tokens"
# (the decoder was trained on [image tokens | text]: inference must run the same graph), "train_backbone", "train_tower".
EXTRAS_FILE = "hip_extras.json"
SPLICE_KEY_MARK = ".splice_image_tokens"


def read_extras(checkpoint_dir) -> dict:
    f = Path(checkpoint_dir) / EXTRAS_FILE
    return json.loads(f.read_text()) if f.is_file() else {}
# state-dict keys of this build that exist only while their feature is on (so a default policy keeps exactly the reference's keys): folded
# dataset statistics (FastVLMBackbone._save_to_state_dict), and -- in files written by round 5 -- the spliced-sequence mode of an unfrozen run (now in EXTRAS_FILE)
EXTRA_STATE_MARKS = (".io_norm.", ".splice_image_tokens")


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
    vlm = {k[len(BACKBONE_PREFIX):]: v for k, v in state.items() if k.startswith(BACKBONE_PREFIX)}
    if vlm:
        policy.model.backbone.load_backbone_state(vlm)
    own = [k for k in policy.state_dict() if not any(m in k for m in EXTRA_STATE_MARKS)]
    missing = [k for k in own if k not in state]
    if missing:
        raise KeyError(f"checkpoint lacks head tensors: {missing}")
    extra = {k: v for k, v in state.items() if any(m in k for m in EXTRA_STATE_MARKS)}   # folded dataset statistics travel with the state dict (round-5 files: the splice mode too)
    policy.load_state_dict({**{k: state[k] for k in own}, **extra})
    ex = read_extras(root)
    if "splice_image_tokens" in ex:
        policy.model.backbone.splice_image_tokens = bool(ex["splice_image_tokens"])
    policy.eval()
    return policy


def save_policy_checkpoint(policy: FastVLAPolicy, checkpoint_dir: str, include_backbone: bool = False) -> Path:
    """policy_config.json + policy_state_dict.pt under the reference's key names (trainer.py:246-255).  include_backbone=True adds
    the VLM tensors the engine was packed from as `model.backbone.model.<canonical key>` and the tied `lm_head.weight`, which is what
    makes the file loadable by the reference's strict `load_state_dict`."""
    d = Path(checkpoint_dir)
    d.mkdir(parents=True, exist_ok=True)
    (d / "policy_config.json").write_text(json.dumps(dataclasses.asdict(policy.config), indent=2))
    # (the splice mode goes to the side file: policy_state_dict.pt keeps the reference's keys only, + `io_norm.*` while dataset statistics are folded)
    state = {k: v.detach().cpu().clone() for k, v in policy.state_dict().items() if SPLICE_KEY_MARK not in k}
    un = getattr(policy, "_unfrozen", None)
    extras = {"splice_image_tokens": bool(policy.model.backbone.splice_image_tokens), "train_backbone": un is not None,
              "train_tower": bool(un is not None and un.train_tower)}
    if any(extras.values()):
        (d / EXTRAS_FILE).write_text(json.dumps(extras, indent=2))
    if include_backbone:
        embed = None
        for name, t in policy.model.backbone.source_tensors():
            state[BACKBONE_PREFIX + name] = t.detach().cpu()
            if name == "model.embed_tokens.weight":
                embed = state[BACKBONE_PREFIX + name]
        if embed is not None and BACKBONE_PREFIX + "lm_head.weight" not in state:
            state[BACKBONE_PREFIX + "lm_head.weight"] = embed    # tied embeddings: the HF state dict lists both names
    torch.save(state, d / "policy_state_dict.pt")
    return d
