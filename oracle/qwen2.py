"""Oracle: Qwen2 decoder prefill -> final-norm hidden states, plus the reference's pooling.  TEST INFRASTRUCTURE ONLY.

Restates [site] transformers/models/qwen2/modeling_qwen2.py (installed 5.15.0), which is what the reference reaches
through `AutoModelForCausalLM` at src/vla_fastvlm/model/fastvlm_adapter.py:533:
  MLP  down(silu(gate(x)) * up(x))                                   :35-48
  RoPE inv_freq = theta^(-2i/d), emb = cat(freqs, freqs), rotate_half :105-135
  attention  softmax(QK^T * d^-0.5 + causal/pad mask) V, GQA repeat_kv, q/k/v bias, no o bias   :150-172,195-234
  RMSNorm  w * x * rsqrt(mean(x^2) + eps) in fp32                      :247-252
  layer    x + attn(norm(x)); x + mlp(norm(x))                        :269-298
and the reference's own pooling, src/vla_fastvlm/model/fastvlm_adapter.py:337-359 (`_pool_hidden`).

The multimodal splice restates upstream LLaVA `prepare_inputs_labels_for_multimodal` [UNVENDORED, PARITY UNPINNED]:
the reference never inserts an image placeholder (fastvlm_adapter.py:361-380), so the literal behaviour is
`cat([text_embeds, image_features[0:0]])`, i.e. a TEXT-ONLY sequence (splice=False).  splice=True puts the projected
image tokens in front of the text (what a prompt "<image>\\n{task}" would produce).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F

LLM = "model."


@dataclass(frozen=True)
class Qwen2Cfg:
    hidden: int = 896
    layers: int = 24
    heads: int = 14
    kv_heads: int = 2
    head_dim: int = 64
    inter: int = 4864
    vocab: int = 151936
    rope_theta: float = 1e6
    rms_eps: float = 1e-6


QWEN2_0_5B = Qwen2Cfg()
QWEN2_7B = Qwen2Cfg(hidden=3584, layers=28, heads=28, kv_heads=4, head_dim=128, inter=18944, vocab=152064)


def rmsnorm(x, w, eps):
    v = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(v + eps))


def rope_tables(cfg: Qwen2Cfg, positions: torch.Tensor):
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, cfg.head_dim, 2, dtype=torch.float32) / cfg.head_dim))
    fr = positions.float()[..., None] * inv  # (..., d/2)
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def decoder_forward(p: Dict[str, torch.Tensor], embeds: torch.Tensor, lengths: torch.Tensor, cfg: Qwen2Cfg,
                    prefix: str = LLM, taps: dict | None = None) -> torch.Tensor:
    """embeds (B,T,H) fp32 right-padded, lengths (B,) valid counts  ->  post-final-norm hidden (B,T,H).

    Positions are 0..T-1 per row (transformers default when no position_ids are given); keys >= length are masked,
    which is what the HF causal+padding mask does for right padding.
    """
    b, t, h = embeds.shape
    pos = torch.arange(t)
    cos, sin = rope_tables(cfg, pos)  # (T,d)
    key_ok = pos[None, :] < lengths[:, None]  # (B,T)
    causal = pos[None, :] <= pos[:, None]  # (Tq,Tk)
    mask = causal[None, None] & key_ok[:, None, None, :]
    x = embeds
    g = cfg.heads // cfg.kv_heads
    for i in range(cfg.layers):
        pre = f"{prefix}layers.{i}."
        y = rmsnorm(x, p[pre + "input_layernorm.weight"], cfg.rms_eps)
        q = F.linear(y, p[pre + "self_attn.q_proj.weight"], p[pre + "self_attn.q_proj.bias"])
        k = F.linear(y, p[pre + "self_attn.k_proj.weight"], p[pre + "self_attn.k_proj.bias"])
        v = F.linear(y, p[pre + "self_attn.v_proj.weight"], p[pre + "self_attn.v_proj.bias"])
        q = q.view(b, t, cfg.heads, cfg.head_dim).transpose(1, 2)
        k = k.view(b, t, cfg.kv_heads, cfg.head_dim).transpose(1, 2)
        v = v.view(b, t, cfg.kv_heads, cfg.head_dim).transpose(1, 2)
        q = q * cos + _rotate_half(q) * sin
        k = k * cos + _rotate_half(k) * sin
        k = k.repeat_interleave(g, dim=1)
        v = v.repeat_interleave(g, dim=1)
        s = (q @ k.transpose(-1, -2)) * cfg.head_dim ** -0.5
        s = s.masked_fill(~mask, torch.finfo(torch.float32).min)
        a = torch.softmax(s, dim=-1, dtype=torch.float32)
        o = (a @ v).transpose(1, 2).reshape(b, t, cfg.heads * cfg.head_dim)
        x = x + F.linear(o, p[pre + "self_attn.o_proj.weight"])
        y = rmsnorm(x, p[pre + "post_attention_layernorm.weight"], cfg.rms_eps)
        m = F.silu(F.linear(y, p[pre + "mlp.gate_proj.weight"])) * F.linear(y, p[pre + "mlp.up_proj.weight"])
        x = x + F.linear(m, p[pre + "mlp.down_proj.weight"])
        if taps is not None:
            taps[f"layer{i}"] = x
    return rmsnorm(x, p[prefix + "norm.weight"], cfg.rms_eps)


def pool_hidden(hidden: torch.Tensor, attention_mask: Optional[torch.Tensor], mode: str = "last_token") -> torch.Tensor:
    """fastvlm_adapter.py:337-359."""
    if mode == "mean_pool":
        if attention_mask is None:
            return hidden.mean(dim=1)
        m = attention_mask.float().unsqueeze(-1)
        return (hidden * m).sum(dim=1) / m.sum(dim=1).clamp_min(1e-6)
    if attention_mask is None:
        return hidden[:, -1, :]
    idx = (attention_mask.long().sum(dim=1) - 1).clamp_min(0)
    return hidden[torch.arange(hidden.shape[0]), idx]


def llm_pooled(p: Dict[str, torch.Tensor], input_ids: torch.Tensor, attention_mask: torch.Tensor, cfg: Qwen2Cfg,
               image_tokens: Optional[torch.Tensor] = None, splice: bool = False, pool: str = "last_token",
               prefix: str = LLM) -> torch.Tensor:
    """input_ids/attention_mask (B,T) right-padded; image_tokens (B,Ni,H) projected features or None -> (B,H).

    Pooling follows fastvlm_adapter.py:558-559: the index comes from the TEXT attention mask.  With splice=True the
    text positions are shifted by Ni, so the pooled row is (Ni + len - 1).
    """
    emb = F.embedding(input_ids, p[prefix + "embed_tokens.weight"])
    lengths = attention_mask.long().sum(dim=1)
    if splice and image_tokens is not None:
        ni = image_tokens.shape[1]
        emb = torch.cat([image_tokens.to(emb.dtype), emb], dim=1)
        hid = decoder_forward(p, emb, lengths + ni, cfg, prefix)
        if pool == "mean_pool":
            full = torch.cat([torch.ones(emb.shape[0], ni, dtype=attention_mask.dtype), attention_mask], dim=1)
            return pool_hidden(hid, full, pool)
        idx = (lengths - 1).clamp_min(0) + ni
        return hid[torch.arange(hid.shape[0]), idx]
    hid = decoder_forward(p, emb, lengths, cfg, prefix)
    return pool_hidden(hid, attention_mask, pool)


def decoder_flops_per_token(cfg: Qwen2Cfg) -> int:
    """GEMM FLOPs per token, lm_head excluded."""
    qkv = cfg.hidden * (cfg.heads + 2 * cfg.kv_heads) * cfg.head_dim
    o = cfg.heads * cfg.head_dim * cfg.hidden
    mlp = 3 * cfg.hidden * cfg.inter
    return 2 * cfg.layers * (qkv + o + mlp)
