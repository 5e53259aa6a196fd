"""Per-GEMM precision budget of the Qwen2 decoder (VERDICT r2 #4c) -- analysis script, CPU, uses the oracle (test infrastructure).

The engine's parity mode (llm_precision = 1) carries EVERY GEMM's activation operand as split bf16 (hi + lo, 16 significant
bits) at twice the MFMA work; plain bf16 operands (8 bits) everywhere give actions 7.9e-3 from the fp32 reference, above
north_star's 1e-3.  This script rounds the activation operand of each projection family {qkv, o, gate_up, down} to either 8 or
16 significant bits inside the fp32 oracle decoder (weights are exact bf16 on both sides, accumulation fp32, attention fp32) and
reports the rel-L2 of the pooled feature and of the actions for all 16 subsets:  python tests/precision_budget.py [0.5b|7b4]
"""
import itertools
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "vla-from-fastvlm_amd")]
from fastvla_hip import arch, weights  # noqa: E402
from oracle import head, qwen2  # noqa: E402


def r8(x):
    return x.to(torch.bfloat16).float()


def r16(x):
    hi = r8(x)
    return hi + r8(x - hi)


def r11(x):
    return x.to(torch.float16).float()


def decoder(p, emb, lengths, cfg, split, base=None):
    """oracle/qwen2.py decoder_forward with the activation operand of each projection rounded as the engine would."""
    b, t, _ = emb.shape
    pos = torch.arange(t)
    cos, sin = qwen2.rope_tables(cfg, pos)
    mask = (pos[None, :] <= pos[:, None])[None, None] & (pos[None, :] < lengths[:, None])[:, None, None, :]
    rnd = {k: (r16 if k in split else (base or r8)) for k in ("qkv", "o", "gate_up", "down")}
    x, g = emb, cfg.heads // cfg.kv_heads
    for i in range(cfg.layers):
        pre = f"model.layers.{i}."
        y = rnd["qkv"](qwen2.rmsnorm(x, p[pre + "input_layernorm.weight"], cfg.rms_eps))
        q = F.linear(y, p[pre + "self_attn.q_proj.weight"], p[pre + "self_attn.q_proj.bias"]).view(b, t, cfg.heads, cfg.head_dim).transpose(1, 2)
        k = F.linear(y, p[pre + "self_attn.k_proj.weight"], p[pre + "self_attn.k_proj.bias"]).view(b, t, cfg.kv_heads, cfg.head_dim).transpose(1, 2)
        v = F.linear(y, p[pre + "self_attn.v_proj.weight"], p[pre + "self_attn.v_proj.bias"]).view(b, t, cfg.kv_heads, cfg.head_dim).transpose(1, 2)
        q = q * cos + qwen2._rotate_half(q) * sin
        k = k * cos + qwen2._rotate_half(k) * sin
        s = (q @ k.repeat_interleave(g, 1).transpose(-1, -2)) * cfg.head_dim ** -0.5
        a = torch.softmax(s.masked_fill(~mask, torch.finfo(torch.float32).min), dim=-1)
        o = rnd["o"]((a @ v.repeat_interleave(g, 1)).transpose(1, 2).reshape(b, t, -1))
        x = x + F.linear(o, p[pre + "self_attn.o_proj.weight"])
        y = rnd["gate_up"](qwen2.rmsnorm(x, p[pre + "post_attention_layernorm.weight"], cfg.rms_eps))
        m = rnd["down"](F.silu(F.linear(y, p[pre + "mlp.gate_proj.weight"])) * F.linear(y, p[pre + "mlp.up_proj.weight"]))
        x = x + F.linear(m, p[pre + "mlp.down_proj.weight"])
    return qwen2.rmsnorm(x, p["model.norm.weight"], cfg.rms_eps)


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "0.5b"
    torch.manual_seed(0)
    if which == "0.5b":
        m = arch.preset("fastvlm-0.5b")
        p = weights.init_llm(m.llm, torch.Generator().manual_seed(2024))
        llm = m.llm
    else:   # the 7B width on 4 layers (tests/test_gpu_fullsize.py::test_7b_decoder_full_width_four_layers)
        llm = arch.LLMConfig(hidden=3584, layers=4, heads=28, kv_heads=4, head_dim=128, inter=18944, vocab=8192)
        p = weights.init_llm(llm, torch.Generator().manual_seed(5))
    cfg = qwen2.Qwen2Cfg(hidden=llm.hidden, layers=llm.layers, heads=llm.heads, kv_heads=llm.kv_heads, head_dim=llm.head_dim,
                         inter=llm.inter, vocab=llm.vocab)
    B, T = 8, 64
    ids = torch.randint(0, min(151643, llm.vocab), (B, T))
    lengths = torch.full((B,), T)
    lengths[1] = 23
    emb = F.embedding(ids, p["model.embed_tokens.weight"])
    shapes = head.head_shapes(llm.hidden, 14, 14, 1024, 1024)
    g = torch.Generator().manual_seed(22)
    hp = {k: (torch.randn(*s, generator=g) / (s[-1] ** 0.5 if len(s) > 1 else 10.0)) + (1.0 if k in ("state_projection.0.weight", "fusion.1.weight") else 0.0)
          for k, s in shapes.items()}
    states = torch.randn(B, 14)
    idx = (lengths - 1).clamp_min(0)

    base = r11 if "f16" in sys.argv else None

    def run(split):
        with torch.no_grad():
            hid = decoder(p, emb, lengths, cfg, split, base)
            pooled = hid[torch.arange(B), idx]
            return pooled, head.head_forward(hp, pooled, states)

    with torch.no_grad():
        ref_p = qwen2.decoder_forward(p, emb, lengths, cfg)[torch.arange(B), idx]
        ref_a = head.head_forward(hp, ref_p, states)
    fams = ("qkv", "o", "gate_up", "down")
    cost = {"qkv": llm.hidden * (llm.heads + 2 * llm.kv_heads) * llm.head_dim, "o": llm.hidden * llm.heads * llm.head_dim,
            "gate_up": 2 * llm.hidden * llm.inter, "down": llm.hidden * llm.inter}
    tot = sum(cost.values())
    rows = []
    for n in range(len(fams) + 1):
        for sub in itertools.combinations(fams, n):
            pl, ac = run(set(sub))
            rp = float((pl - ref_p).norm() / ref_p.norm())
            ra = float((ac - ref_a).norm() / ref_a.norm())
            rows.append((ra, rp, sub, 1 + sum(cost[f] for f in sub) / tot))
    print(f"{which}: B={B} T={T}; split-bf16 (16-bit) operand on the listed families, {'fp16 (11-bit)' if base else 'plain bf16 (8-bit)'} on the rest")
    for ra, rp, sub, c in sorted(rows, key=lambda r: r[3]):
        print(f"  MFMA work x{c:.2f}  actions {ra:.2e}  pooled {rp:.2e}  split: {', '.join(sub) or '-'}")


if __name__ == "__main__":
    main()
