"""FastVLM backbone on the HIP engine: image canonicalisation, tokenisation, VLM call, pooling.

Mirror of the reference's `vla_fastvlm.model.fastvlm_adapter` (src/vla_fastvlm/model/fastvlm_adapter.py):
  FastVLMBackboneConfig            :58-80    same fields / defaults
  FastVLMBackbone.expected_size    :143, resolution order :245-278, tower-name parsing :300-335, too-small guard :145-154
  FastVLMBackbone.output_dim       :107
  FastVLMBackbone._prepare_images_tensor :479-488  -> here ON DEVICE (fv_preprocess), no D2H/CPU-resize/H2D round trip
  FastVLMBackbone._prep_text       :361-380
  FastVLMBackbone.forward          :501-560  -> fv_vision_forward + fv_llm_forward_pooled (pooling fused, :337-359)
  resize_with_pad                  :36-55
Errors keep the reference's types: ValueError (shapes, too-small image_size), RuntimeError (tokenizer missing),
OSError (weights unreachable).  There is no CPU execution path: forward() needs a HIP device.
"""
from __future__ import annotations

import os
import re
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import torch
from torch import nn

from fastvla_hip import FastVLAEngine, FastVLAHipError, arch as fv_arch, weights as fv_weights

from ..tokenization import SyntheticTokenizer

Tensor = torch.Tensor

# Hub ids of the checkpoints the reference documents (README.md:27-40, scripts/download_fastvlm.sh:14-22) -> preset
_KNOWN_MODELS = {
    "apple/fastvlm-0.5b": "fastvlm-0.5b", "llava-fastvithd_0.5b_stage3": "fastvlm-0.5b", "llava-fastvithd_0.5b_stage2": "fastvlm-0.5b",
    "apple/fastvlm-1.5b": "fastvlm-1.5b", "llava-fastvithd_1.5b_stage3": "fastvlm-1.5b", "llava-fastvithd_1.5b_stage2": "fastvlm-1.5b",
    "apple/fastvlm-7b": "fastvlm-7b", "llava-fastvithd_7b_stage3": "fastvlm-7b", "llava-fastvithd_7b_stage2": "fastvlm-7b",
}


@dataclass
class FastVLMBackboneConfig:
    model_id: str = "apple/FastVLM-0.5B"
    bootstrap_model_id: str = "apple/FastVLM-0.5B"  # kept for config compatibility; the native loader needs no bootstrap
    freeze_backbone: bool = True
    image_feature_pool: str = "last_token"  # or "mean_pool"
    fallback_image_size: int = 512
    force_image_size: Optional[int] = None
    normalize_imagenet: bool = False
    resize_with_padding: bool = True
    pad_value: float = 0.0
    tokenizer_max_length: int = 64
    pad_to_max_length: bool = False
    tokenizer_padding_side: str = "right"
    image_key_order: Tuple[str, ...] = ("images", "pixel_values", "pixel_values_vit")


class PreparedPixels(torch.Tensor):
    """(B,S,S,4) bf16 NHWC pixels already letterboxed on device; `_prepare_images_tensor` is idempotent on it, which
    is what makes the reference's double image prep (fastvla/processor_fastvla.py:35 then fastvlm_adapter.py:513)
    a no-op here."""


def infer_size_from_tower_name(tower_name: Any) -> Optional[int]:
    """'mobileclip_l_1024' -> 1024, '...patch14-384' -> 384, 'fastvithd' -> None (reference :300-335)."""
    if not isinstance(tower_name, str):
        return None
    name = tower_name.lower()
    plausible = lambda v: 64 <= v <= 4096  # noqa: E731
    for pat in (r"(?:^|[_-])(\d{2,4})$", r"patch\d+[-_](\d{2,4})(?:$|[_-])"):
        hit = re.search(pat, name)
        if hit and plausible(int(hit.group(1))):
            return int(hit.group(1))
    last = None
    for hit in re.finditer(r"(\d{2,4})", name):
        if plausible(int(hit.group(1))) and name[hit.end():hit.end() + 1] not in ("m", "b"):
            last = int(hit.group(1))  # skip model-scale suffixes such as "so400m"
    return last


def canonical_bchw(images) -> Tensor:
    """PIL / numpy / tensor in BCHW, BHWC, CHW, HWC, HW or a list of those -> (B,C,H,W), float32 (uint8 batches stay
    uint8: the kernel converts), values untouched (reference `_as_bchw` :384-442)."""
    def one(x) -> Tensor:
        if not torch.is_tensor(x):
            import numpy as np
            arr = np.asarray(x)
            if arr.dtype == object or arr.ndim not in (2, 3):
                raise TypeError(f"Unsupported image type: {type(x)}")
            x = torch.from_numpy(arr)
        if x.ndim == 2:
            return x.unsqueeze(0).float()
        if x.ndim != 3:
            raise ValueError(f"Unsupported tensor shape: {tuple(x.shape)}")
        return (x if x.shape[0] in (1, 3) else x.permute(2, 0, 1)).float()

    if torch.is_tensor(images) and images.ndim == 4:
        x = images
        if x.shape[-1] in (1, 3) and x.shape[1] not in (1, 3):
            x = x.permute(0, 3, 1, 2)
        return x if x.dtype == torch.uint8 else x.float()
    if isinstance(images, (list, tuple)):
        return torch.stack([one(i) for i in images], dim=0)
    return one(images).unsqueeze(0)


_LB_ENGINES: Dict[Tuple[int, int], FastVLAEngine] = {}


def _letterbox_engine(device: torch.device, size: int) -> FastVLAEngine:
    """fv_preprocess only reads image_size from its handle: a minimal weight-less engine serves the bare function."""
    key = (device.index or 0, size)
    if key not in _LB_ENGINES:
        model = fv_arch.ModelConfig("letterbox", fv_arch.LLMConfig(hidden=64, layers=1, heads=2, kv_heads=1, head_dim=32, inter=64, vocab=64),
                                    fv_arch.TowerConfig(layers=(1,), dims=(32,), attn_stages=(), image_size=size))
        _LB_ENGINES[key] = FastVLAEngine(model, hidden_dim=32, fusion_dim=32, device=device, max_batch=1, max_text_tokens=8)
    return _LB_ENGINES[key]


def resize_with_pad(img: Tensor, width: int, height: int, pad_value: float = 0.0) -> Tensor:
    """Aspect-preserving bilinear resize + top/left padding (reference :36-55) by the HIP letterbox kernel.
    Square targets (the tower is square); img must live on a HIP device.  Returns (B,3,H,W) float32 holding the bf16
    pixel values the tower consumes."""
    if img.ndim != 4:
        raise ValueError(f"(B,C,H,W) expected, but got shape {tuple(img.shape)}")
    if width != height or width % 4:
        raise ValueError("the HIP letterbox kernel produces square outputs with a side that is a multiple of 4")
    if img.device.type != "cuda":
        raise FastVLAHipError("resize_with_pad runs on the HIP device; move the tensor to cuda first (no CPU path)")
    pix = _letterbox_engine(img.device, int(width)).preprocess(img, pad_value, True)
    return pix[..., :3].permute(0, 3, 1, 2).float()


class FastVLMBackbone(nn.Module):
    """Frozen FastVLM feature extractor `(images, tasks) -> (B, hidden)` running on libfastvla_hip.so."""

    def __init__(self, config: FastVLMBackboneConfig | None = None) -> None:
        super().__init__()
        self.config = config or FastVLMBackboneConfig()
        self.arch, self._weights_source = self._resolve_model(self.config.model_id)
        self.output_dim = int(self.arch.llm.hidden)
        self.expected_size = self._resolve_expected_image_size()
        declared, tower = self._resolve_declared_tower_size()
        if declared is not None and self.config.force_image_size is not None and int(self.expected_size) < int(declared):
            raise ValueError(
                "Configured image_size is too small for this FastVLM vision tower. "
                f"force_image_size={self.expected_size}, tower={tower}, required>={declared}. "
                "Set image_size to the declared tower size (e.g. 1024) or leave it unset (None) for auto-detection.")
        gran = 4 << (len(self.arch.tower.layers) - 1)
        if self.expected_size % gran:
            raise ValueError(f"image size {self.expected_size} must be a multiple of {gran} for the FastViT-HD tower")
        if self.config.tokenizer_padding_side != "right":
            raise ValueError("the HIP decoder masks RIGHT padding; tokenizer_padding_side must be 'right'")
        if self.config.image_feature_pool not in ("last_token", "mean_pool"):
            raise ValueError(f"unknown image_feature_pool '{self.config.image_feature_pool}'")
        self.tokenizer = self._load_tokenizer()
        self.processor = None
        self.image_processor = None
        self.splice_image_tokens = os.environ.get("FASTVLA_SPLICE", "0") == "1"
        # Extensions of this build, attributes rather than config fields because the config dataclass
        # is the reference's public contract; both OFF by default so that a step does exactly the reference's work.  In
        # literal (non-splice) mode the pooled feature depends on the prompt alone, so a policy that is asked the same task
        # string every env step can keep it (LRU keyed by the prompt's token ids) and skip the decoder ...
        self.cache_prompt_features = os.environ.get("FASTVLA_PROMPT_CACHE", "0") == "1"
        self.prompt_cache_size = 256
        # ... and the tower, whose output the literal reference computes and never consumes
        self.skip_unused_tower = os.environ.get("FASTVLA_SKIP_UNUSED_TOWER", "0") == "1"
        # SPLICE mode (image tokens in front of the text): the decoder's keys / values of the image positions depend on the image
        # alone, so they are kept per image (LRU keyed by a 128-bit hash of the image tensor, computed on the device): a repeated
        # frame -- or further prompts on the same frame -- skips letterbox, tower, projector and the 256-token prefix pass, and
        # even a new frame runs its text positions against the prefix instead of one joint 320-token prefill
        self.cache_image_prefix = os.environ.get("FASTVLA_PREFIX_CACHE", "0") == "1"
        self.prefix_cache_size = 64
        self._engine: Optional[FastVLAEngine] = None
        self._weights_override: Optional[Dict[str, Tensor]] = None   # load_backbone_state(): a checkpoint's own VLM tensors
        self._io_norm: Optional[dict] = None   # dataset statistics folded into the head kernels (set_io_normalization)
        self._head_dims = dict(state_dim=14, action_dim=14, hidden_dim=1024, fusion_dim=1024)
        self._max_batch = int(os.environ.get("FASTVLA_MAX_BATCH", "64"))
        print(f"[FastVLMBackbone] expected (S,S) = ({self.expected_size},{self.expected_size})")

    # ------------------------------------------------------------------ model / weights resolution
    @staticmethod
    def _resolve_model(model_id: str):
        mid = str(model_id)
        if mid.startswith("synthetic:"):
            parts = mid.split(":")
            seed = int(parts[2]) if len(parts) > 2 else 1234
            return fv_arch.preset(parts[1]), ("synthetic", seed)
        path = Path(mid)
        key = (path.name if path.is_dir() else mid).lower()
        if path.is_dir():
            for fn in ("fastvla_hip_weights.pt", "fastvla_hip_weights.safetensors"):
                if (path / fn).is_file() and key in _KNOWN_MODELS:
                    return fv_arch.preset(_KNOWN_MODELS[key]), ("file", str(path / fn))
            # a local llava_qwen2 checkpoint directory as Apple ships it (scripts/download_fastvlm.sh of the reference):
            # config.json + *.safetensors whose keys are the canonical inference-form names this library packs
            if (path / "config.json").is_file() and list(path.glob("*.safetensors")):
                return arch_from_hf_config(path / "config.json"), ("hf_dir", str(path))
        if key in _KNOWN_MODELS:
            if os.environ.get("FASTVLA_SYNTHETIC_WEIGHTS", "0") == "1":
                return fv_arch.preset(_KNOWN_MODELS[key]), ("synthetic", 1234)
            raise OSError(
                f"FastVLM checkpoint '{model_id}' is not reachable (offline, nothing cached) and no re-parameterised "
                "weight file (fastvla_hip_weights.pt) was found.  Set FASTVLA_SYNTHETIC_WEIGHTS=1 to run the same "
                "architecture with seeded random weights, or use model_id='synthetic:fastvlm-0.5b'.")
        raise OSError(f"unknown FastVLM model id '{model_id}' (known: {sorted(set(_KNOWN_MODELS))}, or 'synthetic:<preset>')")

    def _load_tokenizer(self):
        """Real checkpoints need their real tokenizer: a missing one raises RuntimeError like the reference
        (model/fastvlm_adapter.py:366-367) instead of silently hashing bytes to ids.  The byte-hash SyntheticTokenizer is
        for seeded synthetic weights only (or, with a warning, under FASTVLA_SYNTHETIC_TOKENIZER=1)."""
        if self._weights_source[0] == "synthetic":
            return SyntheticTokenizer(self.arch.llm.vocab, padding_side=self.config.tokenizer_padding_side)
        src = self._weights_source[1] if self._weights_source[0] == "hf_dir" else self.config.model_id
        try:
            from transformers import AutoTokenizer
            tok = AutoTokenizer.from_pretrained(src, trust_remote_code=False, local_files_only=True)
            tok.padding_side = self.config.tokenizer_padding_side
            return tok
        except Exception as exc:
            if os.environ.get("FASTVLA_SYNTHETIC_TOKENIZER", "0") == "1":
                import warnings
                warnings.warn(f"tokenizer of '{src}' could not be loaded ({exc}); FASTVLA_SYNTHETIC_TOKENIZER=1: using the "
                              "byte-hash SyntheticTokenizer -- input ids are NOT the checkpoint's vocabulary")
                return SyntheticTokenizer(self.arch.llm.vocab, padding_side=self.config.tokenizer_padding_side)
            raise RuntimeError(f"Tokenizer is missing for checkpoint '{src}': {exc}.  Put the tokenizer files next to the "
                               "weights, or set FASTVLA_SYNTHETIC_TOKENIZER=1 to run with hashed ids (smoke tests only).") from exc

    def _resolve_expected_image_size(self) -> int:
        if self.config.force_image_size is not None:
            return int(self.config.force_image_size)
        declared, _ = self._resolve_declared_tower_size()
        if declared is not None:
            return int(declared)
        return int(self.config.fallback_image_size)

    def _resolve_declared_tower_size(self):
        name = self.arch.tower.name
        size = infer_size_from_tower_name(name)
        return (size, name) if size is not None else (None, None)

    _infer_size_from_tower_name = staticmethod(infer_size_from_tower_name)

    # ------------------------------------------------------------------ engine
    def configure_head(self, *, state_dim: int, action_dim: int, hidden_dim: int, fusion_dim: int) -> None:
        self._head_dims = dict(state_dim=state_dim, action_dim=action_dim, hidden_dim=hidden_dim, fusion_dim=fusion_dim)
        self._engine = None

    def engine(self, device: torch.device | None = None) -> FastVLAEngine:
        """Create (once) the HIP engine on `device` and pack the frozen weights into it."""
        if self._engine is None:
            dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
            if dev.type != "cuda":
                raise FastVLAHipError(f"FastVLMBackbone runs on a HIP device only, got '{dev}' (there is no CPU fallback)")
            if dev.index is None:
                dev = torch.device("cuda", torch.cuda.current_device())
            tower = fv_arch.TowerConfig(**{**self.arch.tower.__dict__, "image_size": int(self.expected_size)})
            model = fv_arch.ModelConfig(self.arch.name, self.arch.llm, tower)
            # decoder arithmetic (include/fastvla_hip.h fv_model_desc.llm_precision): fastvla_hip.arch.default_llm_precision (1: split-bf16
            # everywhere, the only policy that holds 1e-3 on every row); FASTVLA_LLM_PRECISION=0 / 1 / 2 / 5 overrides it.  A checkpoint whose gate/up/down weights leave the fp16 range is refused by the library in the fp16 modes
            # (FV_ERR_UNSUPPORTED): warn and fall back to 1, which has no range limit.
            env_prec = os.environ.get("FASTVLA_LLM_PRECISION")
            kind, arg = self._weights_source
            if self._weights_override is not None:
                kind, arg = "state", None    # a policy checkpoint's own VLM tensors (reference utils/checkpoint.py:41: load_state_dict overwrites the backbone)
            prec = int(env_prec) if env_prec is not None else fv_arch.default_llm_precision(self.arch, kind)
            if prec == 2:
                import warnings
                warnings.warn("FASTVLA_LLM_PRECISION=2 (one fp16 pass for gate/up and down): the WORST ROW of the actions sits at 1.0e-3 .. 1.1e-3 from the fp32 "
                              "reference on C1 / C2 (tests/test_gpu_fullsize.py) -- outside the 1e-3 bar; the default (1) holds 2e-5, FASTVLA_LLM_PRECISION=5 holds 6e-4")
            llm = self.arch.llm
            big = 3 * llm.hidden * llm.inter * llm.layers > 2e9   # 7B: 7.6 G parameters = 30 GB as an fp32 host dict

            def build(p: int) -> FastVLAEngine:
                e = FastVLAEngine(model, device=dev, max_batch=self._max_batch, max_text_tokens=self.config.tokenizer_max_length,
                                  tower_microbatch=int(os.environ.get("FASTVLA_TOWER_MICROBATCH", "0")), llm_precision=p, **self._head_dims)
                try:
                    if kind == "state":
                        e.load_weights(self._weights_override)
                    elif kind == "synthetic" and big:
                        # every decoder tensor is drawn on the device in bf16 when the packer asks for it (fv_load_weights_cb)
                        e.load_weights_streaming(fv_weights.stream_backbone(self.arch, seed=arg, device=dev))
                    elif kind == "synthetic":
                        e.load_weights(fv_weights.init_backbone(self.arch, seed=arg))
                    elif kind == "hf_dir":
                        prov = hf_checkpoint_provider(arg)   # one tensor alive at a time, bf16 stays bf16
                        try:
                            e.load_weights_streaming(prov)
                        finally:
                            prov.close()
                    else:
                        e.load_weights(torch.load(arg, map_location="cpu") if arg.endswith(".pt") else _load_safetensors(arg))
                except Exception:
                    e.close()
                    raise
                return e

            try:
                eng = build(prec)
            except FastVLAHipError as exc:
                if prec < 2 or exc.status != -5:   # FV_ERR_UNSUPPORTED from the fp16 weight-range check
                    raise
                import warnings
                warnings.warn(f"llm_precision={prec} refused for '{self.config.model_id}' ({exc}); falling back to llm_precision=1 "
                              "(split-bf16 operands: no range limit, 2x the decoder's MFMA work)")
                eng = build(1)
            if self._io_norm is not None:
                eng.set_io_norm(**self._io_norm)
            self._engine = eng
        return self._engine

    def load_backbone_state(self, state: Dict[str, Tensor]) -> None:
        """Use THESE VLM tensors (canonical checkpoint keys: `model.layers.N...`, `model.vision_tower...`, `model.mm_projector...`)
        instead of the ones `model_id` resolves to -- what the reference's `policy.load_state_dict(state_dict)` does to
        `model.backbone.model.*` (utils/checkpoint.py:41; trainer.py:255 writes them).  A training-form tower is folded;
        `lm_head.*` is dropped (the path computes no logits).  The engine is (re)built on the next use."""
        from .reparam import fold_train_form, is_train_form
        st = {k: v for k, v in state.items() if not k.startswith("lm_head.")}
        if is_train_form(st):
            st = fold_train_form(st)
        self._weights_override = st
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def source_tensors(self):
        """(name, tensor) of every canonical inference-form tensor this backbone packs its engine from, one at a time -- the write
        side of the checkpoint interop (vla_fastvlm/utils/checkpoint.py save_policy_checkpoint(include_backbone=True))."""
        trained = getattr(self, "_trained_tensors", None)
        if trained is not None:
            # an unfrozen run (training/unfrozen.py): decoder + projector come from the fp32 master, the frozen tower from its source
            new = {k: v.detach().cpu() for k, v in trained().items()}
            self._trained_tensors = None
            try:
                for name, t in self.source_tensors():
                    yield name, new.get(name, t).reshape(t.shape)
            finally:
                self._trained_tensors = trained
            return
        if self._weights_override is not None:
            yield from self._weights_override.items()
            return
        kind, arg = self._weights_source
        if kind == "synthetic":
            llm = self.arch.llm
            if 3 * llm.hidden * llm.inter * llm.layers > 2e9:
                prov = fv_weights.stream_backbone(self.arch, seed=arg, device="cpu")
                names = list(fv_weights.init_tower(self.arch.tower, llm.hidden, torch.Generator().manual_seed(arg)))
                names += _llm_tensor_names(llm)
                for n in names:
                    yield n, prov(n)
            else:
                yield from fv_weights.init_backbone(self.arch, seed=arg).items()
        elif kind == "hf_dir":
            yield from load_hf_checkpoint_dir(arg).items()
        else:
            yield from (torch.load(arg, map_location="cpu") if arg.endswith(".pt") else _load_safetensors(arg)).items()

    def set_io_normalization(self, state_mean=None, state_std=None, action_mean=None, action_std=None, eps: float = 1e-8) -> None:
        """Fold the dataset's MEAN_STD statistics of the state input and the action output into the head kernels
        (replaces LeRobot's Normalizer/Unnormalizer steps for those features, reference
        lerobot_fastvla/processor_fastvla.py:34-48).  All None switches the folding off."""
        off = all(v is None for v in (state_mean, state_std, action_mean, action_std))
        self._io_norm = None if off else dict(state_mean=state_mean, state_std=state_std, action_mean=action_mean,
                                              action_std=action_std, eps=eps)
        if self._engine is not None:
            self._engine.set_io_norm(**(self._io_norm or {}))

    # The folded statistics are STATE of the policy: they travel with state_dict() (only while the folding is on, so a policy
    # that never folds keeps exactly the reference's keys) and come back through load_state_dict(), which re-applies them.
    _IO_KEYS = ("state_mean", "state_std", "action_mean", "action_std", "eps")

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self._io_norm is not None:
            for k in self._IO_KEYS:
                destination[prefix + "io_norm." + k] = torch.as_tensor(self._io_norm[k], dtype=torch.float32).detach().cpu().clone().reshape(-1)
        if self.splice_image_tokens:
            # the decoder of an unfrozen run was trained on [image tokens | text] (training/unfrozen.py): a reload must run the same graph
            destination[prefix + "splice_image_tokens"] = torch.ones(1)

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        if prefix + "splice_image_tokens" in state_dict:
            self.splice_image_tokens = bool(float(state_dict.pop(prefix + "splice_image_tokens").reshape(-1)[0]) != 0.0)
        found = {k: state_dict.pop(prefix + "io_norm." + k) for k in self._IO_KEYS if prefix + "io_norm." + k in state_dict}
        if found and len(found) != len(self._IO_KEYS):
            error_msgs.append(f"incomplete folded normalisation statistics under '{prefix}io_norm.': have {sorted(found)}")
        elif found:
            self.set_io_normalization(**{k: found[k].float().cpu() for k in self._IO_KEYS[:4]}, eps=float(found["eps"].reshape(-1)[0]))
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs)

    # ------------------------------------------------------------------ preprocessing
    def _prepare_images_tensor(self, images, device: torch.device) -> Tensor:
        if isinstance(images, PreparedPixels):
            return images
        x = canonical_bchw(images)
        if x.shape[1] not in (1, 3) and x.shape[1] <= 3:
            raise ValueError(f"unsupported channel count {x.shape[1]}")
        if x.shape[1] > 4:
            x = x[:, :3]  # reference _normalize_channels :447-448
        eng = self.engine(device if torch.device(device).type == "cuda" else None)
        # normalize_imagenet (reference _maybe_normalize_imagenet :463-477, after the letterbox :486-487): folded into the letterbox kernel, with the
        # value-range test of the reference's torchvision branch (what an installed reference runs) decided on the device
        pix = eng.preprocess(x.to(eng.device), self.config.pad_value, self.config.resize_with_padding,
                             normalize_imagenet=bool(self.config.normalize_imagenet))
        return pix.as_subclass(PreparedPixels)

    def _prep_text(self, tasks: List[str], device: torch.device) -> Dict[str, Tensor]:
        if self.tokenizer is None:
            raise RuntimeError("Tokenizer is missing; ensure AutoTokenizer/AutoProcessor is available.")
        try:
            self.tokenizer.padding_side = self.config.tokenizer_padding_side
        except Exception:
            pass
        # A policy is asked the same task strings step after step (one env, one instruction): the tokenised batch is kept on the
        # device in a small LRU keyed by the strings themselves, so a repeated prompt costs neither the host tokenizer nor the
        # H2D copy of its ids (tokenisation is a pure function of the strings and these settings).
        tasks = list(tasks)
        key = (tuple(tasks), bool(self.config.pad_to_max_length), int(self.config.tokenizer_max_length), str(self.config.tokenizer_padding_side), str(device))
        lru = self.__dict__.setdefault("_text_lru", {})
        hit = lru.pop(key, None)
        if hit is None:
            tok = self.tokenizer(tasks, padding="max_length" if self.config.pad_to_max_length else "longest",
                                 truncation=True, max_length=self.config.tokenizer_max_length, return_tensors="pt")
            hit = {k: v.to(device) for k, v in tok.items()}
        lru[key] = hit                      # most recent last
        while len(lru) > 32:
            lru.pop(next(iter(lru)))
        return dict(hit)

    @staticmethod
    def _pool_hidden(hidden: Tensor, attention_mask: Optional[Tensor], mode: str) -> Tensor:
        """Torch statement of the pooling the decoder kernel fuses (kept for API parity; reference :337-359)."""
        if mode == "mean_pool":
            if attention_mask is None:
                return hidden.mean(dim=1)
            m = attention_mask.to(hidden.dtype).unsqueeze(-1)
            return (hidden * m).sum(dim=1) / m.sum(dim=1).clamp_min(1e-6)
        if attention_mask is None:
            return hidden[:, -1, :]
        idx = (attention_mask.long().sum(dim=1) - 1).clamp_min(0)
        return hidden[torch.arange(hidden.shape[0], device=hidden.device), idx]

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, images, tasks: List[str], device: torch.device | None = None) -> Tensor:
        eng = self.engine(device if device is not None and torch.device(device).type == "cuda" else None)
        pix = self._prepare_images_tensor(images, eng.device)
        text = self._prep_text(tasks, eng.device)
        return self.forward_ids(pix, text["input_ids"], text["attention_mask"])

    @torch.no_grad()
    def forward_ids(self, images, input_ids: Tensor, attention_mask: Tensor) -> Tensor:
        """Pre-tokenised entry (the benchmark bypasses the host tokenizer): -> pooled (B, hidden) f32."""
        eng = self.engine()
        lens = attention_mask.to(torch.int32).sum(dim=1).to(torch.int32)
        mode = 0 if self.config.image_feature_pool == "last_token" else 1
        literal = not self.splice_image_tokens
        eng.tokens_consumed(not literal)   # spliced tokens feed the decoder: the tower then keeps one set of kernel forms at every batch size (an observation's
                                           # action must not depend on how many observations were evaluated with it); literal mode drops them and stays fast
        if (not literal and self.cache_image_prefix and mode == 0 and eng.llm_precision >= 1 and eng.model.llm.head_dim >= 64
                and torch.is_tensor(images) and images.ndim == 4):
            return self._pooled_through_prefix_cache(eng, images, input_ids, lens)
        pix = self._prepare_images_tensor(images, eng.device)
        if pix.shape[0] != input_ids.shape[0]:
            raise ValueError(f"batch mismatch: {pix.shape[0]} images vs {input_ids.shape[0]} prompts")
        if literal and not self.skip_unused_tower and not self.cache_prompt_features:
            # the reference-literal step: tower + projector run (their output is dropped) BESIDE the decoder on
            # a second HIP stream -- the same schedule bench.py times at the engine level
            return eng.backbone(None, input_ids, lens, pool_mode=mode, pix=pix)
        tok = None
        if not (literal and self.skip_unused_tower):
            tok = eng.vision_forward(pix)  # computed even when not spliced: the literal reference runs the tower too
        if literal and self.cache_prompt_features:
            return self._pooled_through_cache(eng, input_ids, lens, mode)
        return eng.llm_pooled(input_ids, lens, tok if self.splice_image_tokens else None, pool_mode=mode)

    def _pooled_through_cache(self, eng, input_ids: Tensor, lens: Tensor, mode: int) -> Tensor:
        """Literal mode only: pooled rows by prompt.  Keys are the valid token ids of each row (one small D2H copy per call);
        the decoder runs once per call over the rows that missed, and not at all when every prompt is known."""
        cache = self.__dict__.setdefault("_prompt_cache", {})
        ids_h, lens_h = input_ids.detach().cpu(), lens.detach().cpu()
        keys = [(mode, ids_h[b, : int(lens_h[b])].numpy().tobytes()) for b in range(ids_h.shape[0])]
        miss = [b for b, k in enumerate(keys) if k not in cache]
        if miss:
            sel = torch.as_tensor(miss, device=input_ids.device)
            rows = eng.llm_pooled(input_ids.index_select(0, sel).contiguous(), lens.index_select(0, sel).contiguous(), None, pool_mode=mode)
            for j, b in enumerate(miss):
                cache[keys[b]] = rows[j].clone()
            while len(cache) > max(int(self.prompt_cache_size), len(keys)):
                cache.pop(next(iter(cache)))   # dicts keep insertion order: drop the oldest
        return torch.stack([cache[k] for k in keys], dim=0)

    def _image_keys(self, images: Tensor) -> List[tuple]:
        """One 128-bit key per image: two wrapping int64 dot products of the image's raw words with fixed odd multipliers, on the
        device (one small D2H copy per call).  Equal tensors give equal keys; a collision needs both 64-bit sums to agree.  The
        products are taken over bounded column chunks (2^22 words of the whole batch at a time): the transient is ~100 MB whatever
        the frame size, not an int64 copy of the batch."""
        x = images.contiguous()
        words = x.view(x.shape[0], -1).view(torch.int16 if x.element_size() == 2 else torch.int32 if x.element_size() == 4 else torch.uint8)
        B, n = words.shape
        step = max(4096, (1 << 22) // max(B, 1))
        mult = self.__dict__.setdefault("_hash_mult", {})
        mk = (min(step, n), str(x.device))
        if mk not in mult:
            g = torch.Generator().manual_seed(0x5eed)
            mult[mk] = (torch.randint(-2 ** 62, 2 ** 62, (2, mk[0]), generator=g, dtype=torch.int64) | 1).to(x.device)
        m = mult[mk]
        h = torch.zeros(B, 2, dtype=torch.int64, device=x.device)
        for ci, c0 in enumerate(range(0, n, step)):
            w64 = words[:, c0:c0 + step].to(torch.int64)
            k = w64.shape[1]
            # the chunk index enters through an odd per-chunk factor, so equal chunks at different offsets do not cancel or commute
            f = 2 * ci + 1
            h[:, 0] += (w64 * m[0, :k]).sum(1) * f
            h[:, 1] += (w64 * m[1, :k]).sum(1) * (f * f + 2)
        h = h.cpu()
        return [(str(x.dtype), tuple(x.shape[1:]), int(h[b, 0]), int(h[b, 1])) for b in range(B)]

    def _pooled_through_prefix_cache(self, eng, images: Tensor, input_ids: Tensor, lens: Tensor) -> Tensor:
        """Splice mode: per-image decoder prefixes (FastVLAEngine.llm_prefix) from the LRU, tower + prefix pass only for the images
        that miss, then ONE suffix pass over the text positions of the whole batch (llm_pooled_prefixed)."""
        images = images.to(eng.device)
        if images.shape[0] != input_ids.shape[0]:
            raise ValueError(f"batch mismatch: {images.shape[0]} images vs {input_ids.shape[0]} prompts")
        cache = self.__dict__.setdefault("_prefix_cache", {})
        keys = self._image_keys(images)
        miss = [b for b, k in enumerate(keys) if k not in cache]
        first = {}
        for b in miss:
            first.setdefault(keys[b], b)           # the same frame twice in one batch is computed once
        todo = list(first.values())
        if todo:
            sel = torch.as_tensor(todo, device=images.device)
            pix = self._prepare_images_tensor(images.index_select(0, sel), eng.device)
            kv = eng.llm_prefix(eng.vision_forward(pix))
            for j, b in enumerate(todo):
                cache[keys[b]] = kv[:, j].clone()
        for k in keys:                              # refresh recency, oldest first in the dict
            cache[k] = cache.pop(k)
        kvb = torch.stack([cache[k] for k in keys], dim=1)
        while len(cache) > max(int(self.prefix_cache_size), len(set(keys))):
            cache.pop(next(iter(cache)))
        self._prefix_stats = {"images": len(keys), "tower_runs": len(todo)}
        return eng.llm_pooled_prefixed(input_ids, lens, kvb)

    def clear_prefix_cache(self) -> None:
        self.__dict__.get("_prefix_cache", {}).clear()

    def clear_prompt_cache(self) -> None:
        self.__dict__.get("_prompt_cache", {}).clear()

    def backbone(self, images, tasks, device: Optional[torch.device] = None, **kwargs):
        return self.forward(images, tasks, device=device)


def _llm_tensor_names(llm) -> List[str]:
    out = ["model.embed_tokens.weight"]
    for i in range(llm.layers):
        pre = f"model.layers.{i}."
        out += [pre + "input_layernorm.weight", pre + "post_attention_layernorm.weight"]
        out += [pre + f"self_attn.{n}_proj.{k}" for n in ("q", "k", "v") for k in ("weight", "bias")]
        out += [pre + "self_attn.o_proj.weight", pre + "mlp.gate_proj.weight", pre + "mlp.up_proj.weight", pre + "mlp.down_proj.weight"]
    return out + ["model.norm.weight"]


def _load_safetensors(path: str):
    from safetensors.torch import load_file
    return load_file(path)


def arch_from_hf_config(config_json) -> "fv_arch.ModelConfig":
    """Model geometry from a llava_qwen2 `config.json` (Qwen2 fields + `mm_vision_tower`, e.g. "mobileclip_l_1024").
    Only the FastViT-HD tower is supported; its constants are architectural (fastvla_hip.arch.TowerConfig)."""
    import json
    cfg = json.loads(Path(config_json).read_text())
    if cfg.get("model_type") not in ("llava_qwen2", "qwen2", None):
        raise ValueError(f"unsupported model_type '{cfg.get('model_type')}' (expected llava_qwen2)")
    heads = int(cfg["num_attention_heads"])
    hidden = int(cfg["hidden_size"])
    llm = fv_arch.LLMConfig(hidden=hidden, layers=int(cfg["num_hidden_layers"]), heads=heads,
                            kv_heads=int(cfg.get("num_key_value_heads", heads)), head_dim=int(cfg.get("head_dim", hidden // heads)),
                            inter=int(cfg["intermediate_size"]), vocab=int(cfg["vocab_size"]),
                            rope_theta=float(cfg.get("rope_theta", 1e6)), rms_eps=float(cfg.get("rms_norm_eps", 1e-6)))
    tower_name = str(cfg.get("mm_vision_tower", "mobileclip_l_1024"))
    size = infer_size_from_tower_name(tower_name) or 1024
    tower = fv_arch.TowerConfig(image_size=int(size), name=tower_name)
    geo = cfg.get("fastvla_tower")   # extension of THIS build: a reduced FastViT-HD geometry (test checkpoints); Apple's configs have no such key
    if isinstance(geo, dict):
        tower = fv_arch.TowerConfig(layers=tuple(int(v) for v in geo["layers"]), dims=tuple(int(v) for v in geo["dims"]),
                                    attn_stages=tuple(int(v) for v in geo.get("attn_stages", tower.attn_stages)), image_size=int(size), name=tower_name)
    return fv_arch.ModelConfig(Path(config_json).parent.name or "hf-checkpoint", llm, tower)


def hf_checkpoint_provider(path):
    """provider(name) -> tensor for FastVLAEngine.load_weights_streaming over the *.safetensors shards of a llava_qwen2
    checkpoint directory (reference scripts/download_fastvlm.sh:14-22, utils/checkpoint.py:29-42): the decoder's tensors are read
    from their shard one at a time when the packer asks for them (a 7B checkpoint never exists as one host dict and bf16 stays
    bf16); the vision tower + projector (0.25 GB) are read up front because a TRAINING-form tower has to be folded as a whole
    (vla_fastvlm/model/reparam.py).  `lm_head.*` is never read: the path computes no logits."""
    from safetensors import safe_open
    where: Dict[str, str] = {}
    for shard in sorted(Path(path).glob("*.safetensors")):
        with safe_open(str(shard), framework="pt", device="cpu") as f:
            for k in f.keys():
                if not k.startswith("lm_head."):
                    where[k] = str(shard)
    tower_keys = [k for k in where if k.startswith("model.vision_tower.") or k.startswith("model.mm_projector.")]
    tower: Dict[str, Tensor] = {}
    for shard in sorted({where[k] for k in tower_keys}):
        with safe_open(shard, framework="pt", device="cpu") as f:
            for k in tower_keys:
                if where[k] == shard:
                    tower[k] = f.get_tensor(k)
    from .reparam import fold_train_form, is_train_form
    if is_train_form(tower):
        tower = fold_train_form(tower)
    handles: Dict[str, Any] = {}

    def provider(name: str):
        if name in tower:
            return tower[name]
        shard = where.get(name)
        if shard is None:
            return None
        if shard not in handles:
            handles[shard] = safe_open(shard, framework="pt", device="cpu").__enter__()
        return handles[shard].get_tensor(name)

    def close() -> None:
        """release the shard handles (mmap'd files) once fv_load_weights_cb has returned"""
        for hnd in handles.values():
            try:
                hnd.__exit__(None, None, None)
            except Exception:
                pass
        handles.clear()

    provider.close = close
    return provider


def load_hf_checkpoint_dir(path) -> Dict[str, Tensor]:
    """All tensors of every *.safetensors shard in the directory; `lm_head.*` is dropped (the path never computes logits) and a
    vision tower in TRAINING form (multi-branch MobileOne blocks, RepMixer, RepCPE, large-kernel + small-kernel convs with their
    BatchNorms) is folded to the inference form the library packs (vla_fastvlm/model/reparam.py).  Missing or mis-shaped
    tensors are reported by fv_load_weights with the key name."""
    from safetensors.torch import load_file
    state: Dict[str, Tensor] = {}
    for shard in sorted(Path(path).glob("*.safetensors")):
        for k, v in load_file(str(shard)).items():
            if not k.startswith("lm_head."):
                state[k] = v
    from .reparam import fold_train_form, is_train_form
    if is_train_form(state):   # a checkpoint saved before re-parameterisation: fold its branches / BatchNorms on the host, once
        state = fold_train_form(state)
    return state
