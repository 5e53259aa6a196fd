"""CPU: the oracle's mm-projector and image-token SPLICE against the installed third-party `transformers.models.fast_vlm`.

VERDICT r3 missing #4: the FastViT-HD tower has no importable oracle here (timm is absent), but the two steps BEHIND it can be
pinned the way the Qwen2 decoder is (tests/test_oracle_qwen2.py):
  * `FastVlmMultiModalProjector` ([site] transformers/models/fast_vlm/modeling_fast_vlm.py:39-56): Linear + exact-erf GELU + Linear
    with bias -> `oracle.fastvit_hd.projector_forward`;
  * the token order of `get_image_features` (:128-131: `last_hidden_state.flatten(2).permute(0, 2, 1)`, i.e. row-major over the
    16 x 16 map) -> the tail of `oracle.fastvit_hd.tower_head_forward`;
  * the splice itself (:204: `inputs_embeds.masked_scatter(special_image_mask, image_features)`) followed by the Qwen2 decoder
    -> `oracle.qwen2.llm_pooled(..., image_tokens=..., splice=True)`, which puts the projected tokens in FRONT of the text --
    what the prompt "<image> x Ni + task" produces in the HF model.
The vision tower of the HF model is replaced by a stub that returns a given (B, C, h, w) map (its arithmetic is the part that
stays parity-unpinned: SURVEY.md section 8c).  The reference reaches this code path through AutoModelForCausalLM at
src/vla_fastvlm/model/fastvlm_adapter.py:183-191,533.
"""
import types

import pytest
import torch

from oracle import fastvit_hd, qwen2

transformers = pytest.importorskip("transformers")
fast_vlm = pytest.importorskip("transformers.models.fast_vlm.modeling_fast_vlm")

IMG_ID = 90


def _hf_model(cfg: qwen2.Qwen2Cfg, vis_hidden: int, seed: int):
    from transformers.models.fast_vlm.configuration_fast_vlm import FastVlmConfig
    text = dict(model_type="qwen2", vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.inter, num_hidden_layers=cfg.layers,
                num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_heads, max_position_embeddings=512, rms_norm_eps=cfg.rms_eps,
                rope_theta=cfg.rope_theta, tie_word_embeddings=True, attn_implementation="eager")
    # any constructible vision config: the tower module is swapped for a stub below (timm, which the real one wraps, is not installed)
    vis = dict(model_type="clip_vision_model", hidden_size=vis_hidden, intermediate_size=32, num_hidden_layers=1, num_attention_heads=2,
               image_size=32, patch_size=16)
    conf = FastVlmConfig(vision_config=vis, text_config=text, image_token_index=IMG_ID)
    conf._attn_implementation = "eager"
    torch.manual_seed(seed)
    m = fast_vlm.FastVlmModel(conf).eval()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.startswith("vision_tower"):
                continue
            if "norm" in n:
                p.add_(0.1 * torch.randn_like(p))
            if n.endswith("bias"):
                p.normal_(0, 0.05)
    return m


class _StubTower(torch.nn.Module):
    """what `get_image_features` needs of the vision tower: `.last_hidden_state` = the (B, C, h, w) map of the last stage"""

    def __init__(self, fmap):
        super().__init__()
        self.fmap = fmap

    def forward(self, pixel_values, return_dict=True, **kw):
        return types.SimpleNamespace(last_hidden_state=self.fmap[: pixel_values.shape[0]], pooler_output=None)


def _oracle_weights(m):
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    p = {"model." + k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}
    p[fastvit_hd.PROJ + "0.weight"], p[fastvit_hd.PROJ + "0.bias"] = sd["multi_modal_projector.linear_1.weight"], sd["multi_modal_projector.linear_1.bias"]
    p[fastvit_hd.PROJ + "2.weight"], p[fastvit_hd.PROJ + "2.bias"] = sd["multi_modal_projector.linear_2.weight"], sd["multi_modal_projector.linear_2.bias"]
    return p


def test_projector_matches_transformers_fast_vlm():
    cfg = qwen2.Qwen2Cfg(hidden=48, layers=1, heads=4, kv_heads=2, head_dim=12, inter=64, vocab=101)
    m = _hf_model(cfg, vis_hidden=40, seed=3)
    p = _oracle_weights(m)
    torch.manual_seed(4)
    x = torch.randn(3, 7, 40) * 2.0      # wide enough that tanh-GELU and erf-GELU would differ by > 1e-4
    with torch.no_grad():
        ref = m.multi_modal_projector(x)
        got = fastvit_hd.projector_forward(p, x)
    assert float((got - ref).abs().max()) < 2e-6
    assert m.config.projector_hidden_act == "gelu"   # exact-erf GELU: what the real FastVLM configs carry


@pytest.mark.parametrize("cfg,side", [
    (qwen2.Qwen2Cfg(hidden=64, layers=2, heads=4, kv_heads=2, head_dim=16, inter=96, vocab=97), 3),
    (qwen2.Qwen2Cfg(hidden=128, layers=2, heads=2, kv_heads=1, head_dim=64, inter=160, vocab=127, rope_theta=1e6), 4),
])
def test_splice_order_and_spliced_prefill_match_transformers_fast_vlm(cfg, side):
    vis_hidden = 56
    m = _hf_model(cfg, vis_hidden, seed=5)
    p = _oracle_weights(m)
    torch.manual_seed(6)
    B, T, Ni = 3, 7, side * side
    fmap = torch.randn(B, vis_hidden, side, side)               # what the tower's last stage hands over (NCHW)
    m.vision_tower = _StubTower(fmap)
    text = torch.randint(0, IMG_ID, (B, T))                     # ids below the placeholder id
    mask = torch.ones(B, T, dtype=torch.long)
    mask[1, 4:] = 0
    mask[2, 1:] = 0
    ids = torch.cat([torch.full((B, Ni), IMG_ID), text], dim=1)
    full_mask = torch.cat([torch.ones(B, Ni, dtype=torch.long), mask], dim=1)
    with torch.no_grad():
        out = m(input_ids=ids, pixel_values=torch.zeros(B, 3, 8, 8), attention_mask=full_mask, return_dict=True)
        # oracle: tokens in the order tower_head_forward emits them, projected, spliced in front of the text
        tokens = fmap.flatten(2).transpose(1, 2).contiguous()
        img_tok = fastvit_hd.projector_forward(p, tokens)
        pooled = qwen2.llm_pooled(p, text, mask, cfg, image_tokens=img_tok, splice=True)
    # the projected features the HF model scattered, in its order
    hf_feats = out.image_hidden_states.view(B, Ni, cfg.hidden)
    assert float((hf_feats - img_tok).abs().max()) < 2e-6
    lens = mask.sum(1)
    ref_pooled = out.last_hidden_state[torch.arange(B), Ni + lens - 1]
    assert float((pooled - ref_pooled).abs().max()) < 3e-5
    # the order matters: transposing the map (column-major tokens) must NOT reproduce the HF result
    wrong = fastvit_hd.projector_forward(p, fmap.transpose(2, 3).flatten(2).transpose(1, 2).contiguous())
    with torch.no_grad():
        pooled_wrong = qwen2.llm_pooled(p, text, mask, cfg, image_tokens=wrong, splice=True)
    assert float((pooled_wrong - ref_pooled).abs().max()) > 1e-3
