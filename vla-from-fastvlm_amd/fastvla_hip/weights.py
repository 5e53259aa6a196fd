"""Seeded synthetic weights in canonical (checkpoint-key) form.

No FastVLM checkpoint is reachable offline, so parity and benchmarks run on seeded random weights of the exact
architecture.  Keys follow the Apple / HF checkpoint naming in inference (re-parameterised) form, so that a real
checkpoint loader only has to produce the same dict ("checkpoint interop").

Every tensor the library stores as bf16 is rounded to a bf16-representable value HERE, so the fp32 oracle and the HIP
path consume identical weights (the real checkpoints are bf16 too).  Tower weights use fan-in scaling instead of the
flat N(0, 0.02) of BASELINE.md so activations stay O(1) through 44 blocks and parity stays sensitive.
"""
from __future__ import annotations

import math
from typing import Dict

import torch

from .arch import ModelConfig

VT = "model.vision_tower.vision_tower.model."
PROJ = "model.mm_projector."
LLM = "model."


def _bf(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def network_index_map(cfg):
    out = []
    n = len(cfg.layers)
    for i in range(n):
        if i in cfg.attn_stages:
            out.append(("cpe", i))
        out.append(("stage", i))
        if i < n - 1:
            out.append(("down", i))
    return out


def init_tower(cfg, hidden: int, gen: torch.Generator) -> Dict[str, torch.Tensor]:
    p: Dict[str, torch.Tensor] = {}

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=gen) * std

    def ru(*shape, lo=0.0, hi=1.0):
        return torch.rand(*shape, generator=gen) * (hi - lo) + lo

    def dw_identity(c, k, noise):
        w = rn(c, 1, k, k, std=noise)
        w[:, 0, k // 2, k // 2] += 1.0
        return w

    c0 = cfg.dims[0]
    p[VT + "patch_embed.0.reparam_conv.weight"] = rn(c0, 3, 3, 3, std=1.5 / math.sqrt(27))
    p[VT + "patch_embed.0.reparam_conv.bias"] = rn(c0, std=0.1)
    p[VT + "patch_embed.1.reparam_conv.weight"] = rn(c0, 1, 3, 3, std=1.5 / 3)
    p[VT + "patch_embed.1.reparam_conv.bias"] = rn(c0, std=0.1)
    p[VT + "patch_embed.2.reparam_conv.weight"] = _bf(rn(c0, c0, 1, 1, std=1.5 / math.sqrt(c0)))
    p[VT + "patch_embed.2.reparam_conv.bias"] = rn(c0, std=0.1)

    def ffn(pre, c, ls_key):
        h = c * cfg.mlp_ratio
        p[pre + "convffn.conv.conv.weight"] = rn(c, 1, 7, 7, std=1.0 / 7)
        p[pre + "convffn.conv.bn.weight"] = ru(c, lo=0.8, hi=1.2)
        p[pre + "convffn.conv.bn.bias"] = rn(c, std=0.05)
        p[pre + "convffn.conv.bn.running_mean"] = rn(c, std=0.1)
        p[pre + "convffn.conv.bn.running_var"] = ru(c, lo=0.5, hi=1.5)
        p[pre + "convffn.fc1.weight"] = _bf(rn(h, c, 1, 1, std=1.0 / math.sqrt(c)))
        p[pre + "convffn.fc1.bias"] = rn(h, std=0.05)
        p[pre + "convffn.fc2.weight"] = _bf(rn(c, h, 1, 1, std=1.0 / math.sqrt(h)))
        p[pre + "convffn.fc2.bias"] = rn(c, std=0.05)
        p[pre + ls_key] = ru(c, 1, 1, lo=0.05, hi=0.25)

    for idx, (kind, i) in enumerate(network_index_map(cfg)):
        c = cfg.dims[i]
        if kind == "cpe":
            p[VT + f"network.{idx}.reparam_conv.weight"] = dw_identity(c, 7, 0.03)
            p[VT + f"network.{idx}.reparam_conv.bias"] = rn(c, std=0.02)
        elif kind == "down":
            c2 = cfg.dims[i + 1]
            p[VT + f"network.{idx}.proj.0.lkb_reparam.weight"] = rn(c2, 1, 7, 7, std=1.5 / 7)
            p[VT + f"network.{idx}.proj.0.lkb_reparam.bias"] = rn(c2, std=0.05)
            p[VT + f"network.{idx}.proj.1.reparam_conv.weight"] = _bf(rn(c2, c2, 1, 1, std=1.5 / math.sqrt(c2)))
            p[VT + f"network.{idx}.proj.1.reparam_conv.bias"] = rn(c2, std=0.05)
        else:
            for j in range(cfg.layers[i]):
                pre = VT + f"network.{idx}.{j}."
                if i in cfg.attn_stages:
                    p[pre + "norm.weight"] = 1.0 + rn(c, std=0.1)
                    p[pre + "norm.bias"] = rn(c, std=0.05)
                    p[pre + "token_mixer.qkv.weight"] = _bf(rn(3 * c, c, std=1.5 / math.sqrt(c)))
                    p[pre + "token_mixer.proj.weight"] = _bf(rn(c, c, std=1.0 / math.sqrt(c)))
                    p[pre + "token_mixer.proj.bias"] = rn(c, std=0.05)
                    p[pre + "layer_scale_1"] = ru(c, 1, 1, lo=0.05, hi=0.25)
                    ffn(pre, c, "layer_scale_2")
                else:
                    p[pre + "token_mixer.reparam_conv.weight"] = dw_identity(c, 3, 0.08)
                    p[pre + "token_mixer.reparam_conv.bias"] = rn(c, std=0.02)
                    ffn(pre, c, "layer_scale")
    co, rd = cfg.out_dim, cfg.se_rd
    p[VT + "conv_exp.reparam_conv.weight"] = rn(co, 1, 3, 3, std=1.0 / 3)
    p[VT + "conv_exp.reparam_conv.bias"] = rn(co, std=0.05)
    p[VT + "conv_exp.se.reduce.weight"] = rn(rd, co, 1, 1, std=1.0 / math.sqrt(co))
    p[VT + "conv_exp.se.reduce.bias"] = rn(rd, std=0.1)
    p[VT + "conv_exp.se.expand.weight"] = rn(co, rd, 1, 1, std=1.0 / math.sqrt(rd))
    p[VT + "conv_exp.se.expand.bias"] = rn(co, std=0.1)
    p[PROJ + "0.weight"] = _bf(rn(hidden, co, std=1.0 / math.sqrt(co)))
    p[PROJ + "0.bias"] = rn(hidden, std=0.02)
    p[PROJ + "2.weight"] = _bf(rn(hidden, hidden, std=0.5 / math.sqrt(hidden)))
    p[PROJ + "2.bias"] = rn(hidden, std=0.02)
    return p


def init_llm(cfg, gen: torch.Generator, std: float = 0.02) -> Dict[str, torch.Tensor]:
    p: Dict[str, torch.Tensor] = {}

    def rn(*shape, s=std):
        return torch.randn(*shape, generator=gen) * s

    h, d = cfg.hidden, cfg.head_dim
    p[LLM + "embed_tokens.weight"] = _bf(rn(cfg.vocab, h))
    for i in range(cfg.layers):
        pre = f"{LLM}layers.{i}."
        p[pre + "input_layernorm.weight"] = 1.0 + rn(h, s=0.05)
        p[pre + "post_attention_layernorm.weight"] = 1.0 + rn(h, s=0.05)
        p[pre + "self_attn.q_proj.weight"] = _bf(rn(cfg.heads * d, h))
        p[pre + "self_attn.k_proj.weight"] = _bf(rn(cfg.kv_heads * d, h))
        p[pre + "self_attn.v_proj.weight"] = _bf(rn(cfg.kv_heads * d, h))
        p[pre + "self_attn.q_proj.bias"] = rn(cfg.heads * d)
        p[pre + "self_attn.k_proj.bias"] = rn(cfg.kv_heads * d)
        p[pre + "self_attn.v_proj.bias"] = rn(cfg.kv_heads * d)
        p[pre + "self_attn.o_proj.weight"] = _bf(rn(h, cfg.heads * d))
        p[pre + "mlp.gate_proj.weight"] = _bf(rn(cfg.inter, h))
        p[pre + "mlp.up_proj.weight"] = _bf(rn(cfg.inter, h))
        p[pre + "mlp.down_proj.weight"] = _bf(rn(h, cfg.inter))
    p[LLM + "norm.weight"] = 1.0 + rn(h, s=0.05)
    return p


def init_backbone(model: ModelConfig, seed: int = 1234) -> Dict[str, torch.Tensor]:
    gen = torch.Generator().manual_seed(seed)
    p = init_tower(model.tower, model.llm.hidden, gen)
    p.update(init_llm(model.llm, gen))
    return p


# ---------------------------------------------------------------------------------------------------------------
# Streaming form: one tensor at a time, generated where it is needed (FastVLAEngine.load_weights_streaming).
# A 7B decoder is 7.6 G parameters -- 30 GB as an fp32 dict on the host; here each tensor is drawn from its OWN generator
# (seeded by the run seed and the tensor's name) on the target device, in bf16 where the library stores bf16, so the
# biggest live object is one weight matrix and any single tensor can be regenerated for the oracle by name.
def _name_seed(seed: int, name: str) -> int:
    import zlib
    return (seed * 1_000_003 + zlib.crc32(name.encode())) & 0x7FFFFFFFFFFFFFFF


def llm_tensor(cfg, name: str, seed: int = 1234, device="cpu", std: float = 0.02):
    """One decoder tensor by checkpoint key (same distributions as init_llm); None for keys the decoder does not have."""
    if not name.startswith(LLM):
        return None
    key = name[len(LLM):]
    h, d = cfg.hidden, cfg.head_dim
    shapes = {"embed_tokens.weight": (cfg.vocab, h), "norm.weight": (h,)}
    if key.startswith("layers."):
        _, idx, rest = key.split(".", 2)
        if not idx.isdigit() or int(idx) >= cfg.layers:
            return None
        shapes = {"input_layernorm.weight": (h,), "post_attention_layernorm.weight": (h,),
                  "self_attn.q_proj.weight": (cfg.heads * d, h), "self_attn.k_proj.weight": (cfg.kv_heads * d, h),
                  "self_attn.v_proj.weight": (cfg.kv_heads * d, h), "self_attn.q_proj.bias": (cfg.heads * d,),
                  "self_attn.k_proj.bias": (cfg.kv_heads * d,), "self_attn.v_proj.bias": (cfg.kv_heads * d,),
                  "self_attn.o_proj.weight": (h, cfg.heads * d), "mlp.gate_proj.weight": (cfg.inter, h),
                  "mlp.up_proj.weight": (cfg.inter, h), "mlp.down_proj.weight": (h, cfg.inter)}
        key = rest
    if key not in shapes:
        return None
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(_name_seed(seed, name))
    shape = shapes[key]
    if key.endswith("norm.weight") or key.endswith("layernorm.weight"):
        return 1.0 + torch.randn(shape, generator=g, device=dev) * 0.05
    t = torch.randn(shape, generator=g, device=dev) * std
    return t.to(torch.bfloat16) if len(shape) == 2 else t


def stream_backbone(model: ModelConfig, seed: int = 1234, device="cpu"):
    """provider(name) -> tensor for FastVLAEngine.load_weights_streaming: the tower + projector (125 M parameters) from
    init_tower on the host, every decoder tensor drawn on `device` when asked for."""
    tower = init_tower(model.tower, model.llm.hidden, torch.Generator().manual_seed(seed))

    def provider(name: str):
        if name in tower:
            return tower[name]
        return llm_tensor(model.llm, name, seed, device)

    return provider
