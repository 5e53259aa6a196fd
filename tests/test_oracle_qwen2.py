"""CPU: the Qwen2 restatement against the installed third-party transformers Qwen2 (random weights).
This is the only live oracle for the decoder: the reference holds no fixture for it."""
import pytest
import torch

from oracle import qwen2

transformers = pytest.importorskip("transformers")


def _hf_model(cfg: qwen2.Qwen2Cfg, seed: int):
    from transformers import Qwen2Config, Qwen2ForCausalLM
    hf = Qwen2Config(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter,
                     num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_heads,
                     max_position_embeddings=512, rms_norm_eps=cfg.rms_eps, rope_theta=cfg.rope_theta,
                     tie_word_embeddings=True, attn_implementation="eager")
    torch.manual_seed(seed)
    m = Qwen2ForCausalLM(hf).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
            if n.endswith("bias"):
                p.normal_(0, 0.05)
    return m


@pytest.mark.parametrize("cfg", [
    qwen2.Qwen2Cfg(hidden=64, layers=3, heads=4, kv_heads=2, head_dim=16, inter=160, vocab=211),
    qwen2.Qwen2Cfg(hidden=128, layers=2, heads=2, kv_heads=1, head_dim=64, inter=96, vocab=97, rope_theta=1e6),
])
def test_decoder_matches_transformers(cfg):
    m = _hf_model(cfg, 7)
    p = {k: v.detach() for k, v in m.state_dict().items()}
    torch.manual_seed(8)
    b, t = 3, 9
    ids = torch.randint(0, cfg.vocab, (b, t))
    mask = torch.ones(b, t, dtype=torch.long)
    mask[1, 5:] = 0
    mask[2, 1:] = 0
    with torch.no_grad():
        out = m(input_ids=ids, attention_mask=mask, output_hidden_states=True, return_dict=True)
    assert not hasattr(out, "last_hidden_state") or out.last_hidden_state is None  # -> hidden_states[-1] branch
    ref = out.hidden_states[-1]
    hid = qwen2.decoder_forward(p, torch.nn.functional.embedding(ids, p["model.embed_tokens.weight"]),
                                mask.sum(1), cfg)
    valid = mask.bool()
    assert float((hid[valid] - ref[valid]).abs().max()) < 2e-5
    pooled = qwen2.llm_pooled(p, ids, mask, cfg)
    ref_pooled = qwen2.pool_hidden(ref, mask)
    assert float((pooled - ref_pooled).abs().max()) < 2e-5


def test_right_padding_does_not_change_valid_rows():
    cfg = qwen2.Qwen2Cfg(hidden=64, layers=2, heads=4, kv_heads=2, head_dim=16, inter=96, vocab=50)
    m = _hf_model(cfg, 3)
    p = {k: v.detach() for k, v in m.state_dict().items()}
    ids = torch.randint(0, 50, (1, 6))
    a = qwen2.llm_pooled(p, ids, torch.ones(1, 6, dtype=torch.long), cfg)
    ids2 = torch.cat([ids, torch.randint(0, 50, (1, 4))], dim=1)
    m2 = torch.cat([torch.ones(1, 6, dtype=torch.long), torch.zeros(1, 4, dtype=torch.long)], dim=1)
    b = qwen2.llm_pooled(p, ids2, m2, cfg)
    assert float((a - b).abs().max()) < 1e-6


def test_flops_per_token_matches_survey():
    assert abs(qwen2.decoder_flops_per_token(qwen2.QWEN2_0_5B) / 1e9 - 0.716) < 0.002
    assert abs(qwen2.decoder_flops_per_token(qwen2.QWEN2_7B) / 1e9 - 13.05) < 0.02
