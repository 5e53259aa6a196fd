"""Architecture constants of the models the path serves (shapes only; no arithmetic here).

Qwen2 constants: public Qwen2-0.5B / 7B configs, corroborated by [site] transformers fast_vlm/configuration_fast_vlm.py:82-89.
FastViT-HD constants: [UNVENDORED] apple ml-fastvlm mobileclip/mci.py `fastvithd` (layers [2,12,24,4,2],
dims [96,192,384,768,1536], RepMixer x3 + attention x2, mlp_ratio 4, head_dim 32, conv_exp x2 + SE 1/16), running at the
tower's declared size (1024 for `mobileclip_l_1024`; reference model/fastvlm_adapter.py:245-278).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Tuple


@dataclass(frozen=True)
class LLMConfig:
    hidden: int = 896
    layers: int = 24
    heads: int = 14
    kv_heads: int = 2
    head_dim: int = 64
    inter: int = 4864
    vocab: int = 151936
    rope_theta: float = 1e6
    rms_eps: float = 1e-6


@dataclass(frozen=True)
class TowerConfig:
    layers: Tuple[int, ...] = (2, 12, 24, 4, 2)
    dims: Tuple[int, ...] = (96, 192, 384, 768, 1536)
    attn_stages: Tuple[int, ...] = (3, 4)
    mlp_ratio: int = 4
    head_dim: int = 32
    se_ratio: float = 0.0625
    cls_ratio: float = 2.0
    ln_eps: float = 1e-5
    bn_eps: float = 1e-5
    image_size: int = 1024
    name: str = "mobileclip_l_1024"

    @property
    def out_dim(self) -> int:
        return int(self.dims[-1] * self.cls_ratio)

    @property
    def se_rd(self) -> int:
        return int(self.out_dim * self.se_ratio)

    @property
    def tokens_side(self) -> int:
        return self.image_size >> (len(self.layers) + 1)

    @property
    def num_tokens(self) -> int:
        return self.tokens_side ** 2


@dataclass(frozen=True)
class ModelConfig:
    name: str
    llm: LLMConfig = field(default_factory=LLMConfig)
    tower: TowerConfig = field(default_factory=TowerConfig)


PRESETS = {
    "fastvlm-0.5b": ModelConfig("fastvlm-0.5b"),
    # apple/FastVLM-1.5B: the Qwen2-1.5B decoder behind the same FastViT-HD tower (projector 3072 -> 1536)
    "fastvlm-1.5b": ModelConfig("fastvlm-1.5b", LLMConfig(hidden=1536, layers=28, heads=12, kv_heads=2, head_dim=128,
                                                        inter=8960, vocab=151936)),
    "fastvlm-7b": ModelConfig("fastvlm-7b", LLMConfig(hidden=3584, layers=28, heads=28, kv_heads=4, head_dim=128,
                                                    inter=18944, vocab=152064)),
    # reduced shapes for parity tests the CPU oracle finishes in seconds (same graph, same kernels)
    "tiny": ModelConfig("tiny", LLMConfig(hidden=128, layers=2, heads=4, kv_heads=2, head_dim=32, inter=256, vocab=512),
                        TowerConfig(layers=(1, 2, 2, 1, 1), dims=(32, 64, 128, 256, 512), image_size=256, name="tiny_256")),
    "small": ModelConfig("small", LLMConfig(hidden=256, layers=3, heads=4, kv_heads=2, head_dim=64, inter=640, vocab=1024),
                         TowerConfig(layers=(1, 2, 3, 2, 1), dims=(32, 64, 128, 256, 512), image_size=384, name="small_384")),
}


def default_llm_precision(model: ModelConfig, weights_source: str = "synthetic") -> int:
    """The decoder arithmetic a backbone gets when nothing asks for another (fv_model_desc.llm_precision): 1 for EVERY model and weight
    source since round 4.

    1 = split-bf16 operands on every projection (16 significant bits, fp32 attention): actions 1e-5 from the fp32 oracle at 0.5B, pooled
    feature <= 3e-4 on the WHOLE 7B decoder (28 layers, tests/test_gpu_fullsize.py::test_7b_full_depth_against_layer_streamed_oracle), no
    range limit on weights or activations.
    2 = split-bf16 qkv / o + ONE fp16 pass for gate/up and down (0.56x the MFMA work, -2.5 ms of the 0.5B bs=64 step) was round 3's default
    for the 0.5B decoder on the strength of a batch-level rel-L2 of 4.4e-4 .. 6.4e-4.  Round 4 measured it ROW BY ROW: the worst sample of
    C1's four rows sits at 1.1e-3 (tests/test_gpu_fullsize.py::test_c1_train_step[policy2-optin-literal]) -- outside north_star's 1e-3 for
    that env's action -- and a real checkpoint's outlier channels have never been measured.  It therefore stays an OPT-IN
    (FASTVLA_LLM_PRECISION=2, bench.py --llm-precision 2): weights outside the fp16 range are refused at load time and the backbone
    falls back to 1; activations saturate and are counted (FastVLAEngine.fp16_saturations)."""
    return 1


def preset(name: str) -> ModelConfig:
    key = name.lower()
    if key not in PRESETS:
        raise ValueError(f"unknown model preset '{name}' (have {sorted(PRESETS)})")
    return PRESETS[key]
