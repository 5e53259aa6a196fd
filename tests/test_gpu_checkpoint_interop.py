"""-m gpu: checkpoint interop END TO END (VERDICT r3 missing #3 / next #1-v).

Read side: a llava_qwen2 checkpoint DIRECTORY as Apple ships it (reference scripts/download_fastvlm.sh:14-22: config.json +
*.safetensors shards) whose vision tower is in TRAINING form (multi-branch MobileOne blocks, RepMixer, RepCPE, large-kernel + small-
kernel PatchEmbeds, BatchNorms) -> FastVLMBackbone(model_id=<dir>) -> hf_checkpoint_provider -> fold -> fv_load_weights_cb -> the
engine; its image tokens are held against oracle/reparam.py's UNFOLDED forward (nothing folded on the oracle side), its pooled feature
and actions against the fp32 oracle at north_star's 1e-3.  Real checkpoints get llm_precision = 1 by default and fall back to it when
an fp16 policy is asked for weights outside the fp16 range.

Write side: vla_fastvlm.utils.save_policy_checkpoint(include_backbone=True) writes `policy_state_dict.pt` with the reference's key
names (`model.state_projection.*` ... AND the whole VLM under `model.backbone.model.*`, reference training/trainer.py:246-255);
load_policy_from_checkpoint (reference utils/checkpoint.py:29-42) packs the engine from THOSE tensors, not from `vlm_model_name`.
"""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import DEV, check_close, rel_l2  # noqa: E402
from fastvla_hip import arch, weights  # noqa: E402
from oracle import fastvit_hd, head, preprocess, qwen2  # noqa: E402
from oracle import reparam as oref  # noqa: E402
from test_reparam import _unfold  # noqa: E402


def _write_checkpoint_dir(d, m, w_tower_train, w_llm, extra=None):
    from safetensors.torch import save_file
    d.mkdir(parents=True, exist_ok=True)
    cfg = dict(model_type="llava_qwen2", hidden_size=m.llm.hidden, num_hidden_layers=m.llm.layers, num_attention_heads=m.llm.heads,
               num_key_value_heads=m.llm.kv_heads, intermediate_size=m.llm.inter, vocab_size=m.llm.vocab, rope_theta=m.llm.rope_theta,
               rms_norm_eps=m.llm.rms_eps, head_dim=m.llm.head_dim, mm_vision_tower=f"mobileclip_l_{m.tower.image_size}", torch_dtype="bfloat16",
               fastvla_tower=dict(layers=list(m.tower.layers), dims=list(m.tower.dims), attn_stages=list(m.tower.attn_stages)))
    (d / "config.json").write_text(json.dumps(cfg))
    # the decoder's matrices as bf16 (what the real checkpoints hold), everything else fp32; two shards; an lm_head nobody reads
    llm = {k: (v.to(torch.bfloat16) if v.ndim == 2 else v).contiguous() for k, v in w_llm.items()}
    llm["lm_head.weight"] = llm["model.embed_tokens.weight"].clone()
    llm.update(extra or {})
    save_file({k: v.contiguous() for k, v in w_tower_train.items()}, str(d / "model-00001-of-00002.safetensors"))
    save_file(llm, str(d / "model-00002-of-00002.safetensors"))


@pytest.fixture(scope="module")
def ckpt(tmp_path_factory):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device")
    m = arch.preset("small")
    g = torch.Generator().manual_seed(31)
    w_tower = weights.init_tower(m.tower, m.llm.hidden, g)
    w_llm = weights.init_llm(m.llm, g)
    train = _unfold(w_tower, g)
    assert any(".rbr_conv." in k for k in train) and any("lkb_origin" in k for k in train)
    d = tmp_path_factory.mktemp("ckpt") / "llava-fastvithd_small_stage3"
    _write_checkpoint_dir(d, m, train, w_llm)
    return m, d, w_tower, w_llm, train


def _backbone(d, monkeypatch, **env):
    from vla_fastvlm.model.fastvlm_adapter import FastVLMBackbone, FastVLMBackboneConfig
    monkeypatch.setenv("FASTVLA_SYNTHETIC_TOKENIZER", "1")   # the directory ships no tokenizer files; ids are given explicitly below
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    with pytest.warns(UserWarning):
        bb = FastVLMBackbone(FastVLMBackboneConfig(model_id=str(d)))
    bb.configure_head(state_dim=14, action_dim=14, hidden_dim=64, fusion_dim=64)
    return bb


def test_train_form_checkpoint_directory_through_the_engine(ckpt, monkeypatch):
    m, d, w_tower, w_llm, train = ckpt
    bb = _backbone(d, monkeypatch)
    assert bb._weights_source[0] == "hf_dir" and bb.arch.tower.layers == m.tower.layers and bb.arch.llm == m.llm
    eng = bb.engine(torch.device(DEV))
    assert eng.llm_precision == 1, "a real checkpoint directory gets the split-bf16 policy unless it opts in"
    tc = fastvit_hd.TowerCfg(layers=m.tower.layers, dims=m.tower.dims, mlp_ratio=m.tower.mlp_ratio, head_dim=m.tower.head_dim, attn_stages=m.tower.attn_stages)
    lc = qwen2.Qwen2Cfg(hidden=m.llm.hidden, layers=m.llm.layers, heads=m.llm.heads, kv_heads=m.llm.kv_heads, head_dim=m.llm.head_dim, inter=m.llm.inter,
                        vocab=m.llm.vocab, rope_theta=m.llm.rope_theta, rms_eps=m.llm.rms_eps)
    torch.manual_seed(32)
    B, T = 3, 12
    img = torch.rand(B, 3, 120, 160)
    pix = eng.preprocess(img.to(DEV))
    tok, tout = eng.vision_forward(pix, return_tower_out=True)
    torch.cuda.synchronize()
    x = pix.float().cpu()[..., :3].permute(0, 3, 1, 2).contiguous()
    check_close(x, preprocess.letterbox(img, m.tower.image_size), rel=3e-3, amax=5e-3, what="letterbox")
    with torch.no_grad():
        emb = oref.tower_forward_train_form(train, x, tc)             # the UNFOLDED graph on the training-form tensors
        ref_tok = fastvit_hd.projector_forward(train, emb)
        emb_folded = fastvit_hd.tower_forward(w_tower, x, tc)         # the folded oracle: same thing to fp32 rounding
    assert rel_l2(emb, emb_folded) < 1e-4
    r1, _ = check_close(tout.float().cpu(), emb, rel=1e-2, amax=5e-2, what="tower embeddings vs the unfolded oracle")
    r2, _ = check_close(tok.cpu(), ref_tok, rel=1e-2, amax=5e-2, what="image tokens vs the unfolded oracle")
    # the decoder + pooling + head on the directory's bf16 matrices, literal mode, through the backbone's own entry point
    ids = torch.randint(0, m.llm.vocab, (B, T))
    mask = torch.ones(B, T, dtype=torch.long)
    mask[1, 7:] = 0
    pooled = bb.forward_ids(img.to(DEV), ids.to(DEV), mask.to(DEV))
    wl = {k: v.to(torch.bfloat16).float() if v.ndim == 2 else v for k, v in w_llm.items()}
    with torch.no_grad():
        ref_pooled = qwen2.llm_pooled(wl, ids, mask, lc)
    shapes = head.head_shapes(lc.hidden, 14, 14, 64, 64)
    gh = torch.Generator().manual_seed(33)
    p = {k: torch.randn(*s, generator=gh) * 0.1 + (1.0 if k in ("state_projection.0.weight", "fusion.1.weight") else 0.0) for k, s in shapes.items()}
    flat = torch.zeros(eng.head_numel(), device=DEV)
    for k, v in eng.head_views(flat).items():
        v.copy_(p[k])
    states = torch.randn(B, 14, generator=gh)
    act, _ = eng.head_forward(flat, pooled, states.to(DEV))
    torch.cuda.synchronize()
    rp = rel_l2(pooled.cpu(), ref_pooled)
    ra = rel_l2(act.cpu(), head.head_forward(p, ref_pooled, states))
    print(f"[f3 train-form dir -> engine] tower {r1:.2e}, tokens {r2:.2e} vs the UNFOLDED oracle; pooled {rp:.2e}, actions {ra:.2e} vs fp32 oracle")
    assert rp <= 3e-4 and ra <= 1e-3
    eng.close()


def test_fp16_policy_falls_back_on_a_checkpoint_outside_the_fp16_range(ckpt, tmp_path, monkeypatch):
    """FASTVLA_LLM_PRECISION=2 on a directory whose down_proj holds a 5000 (x16 = 80000 > 65504): the library refuses the load
    (FV_ERR_UNSUPPORTED), the backbone warns and rebuilds with llm_precision=1 -- no inf, no silent clamp."""
    m, _, w_tower, w_llm, train = ckpt
    bad = dict(w_llm)
    t = bad["model.layers.1.mlp.down_proj.weight"].clone()
    t[3, 5] = 4992.0
    bad["model.layers.1.mlp.down_proj.weight"] = t
    d = tmp_path / "llava-fastvithd_outlier_stage3"
    _write_checkpoint_dir(d, m, train, bad)
    bb = _backbone(d, monkeypatch, FASTVLA_LLM_PRECISION="2")
    with pytest.warns(UserWarning, match="falling back to llm_precision=1"):
        eng = bb.engine(torch.device(DEV))
    assert eng.llm_precision == 1
    ids = torch.randint(0, m.llm.vocab, (2, 6))
    pooled = eng.llm_pooled(ids, torch.tensor([6, 4]))
    torch.cuda.synchronize()
    assert torch.isfinite(pooled).all()
    eng.close()
    # the same request on the healthy directory is honoured
    bb2 = _backbone(ckpt[1], monkeypatch, FASTVLA_LLM_PRECISION="2")
    e2 = bb2.engine(torch.device(DEV))
    assert e2.llm_precision == 2 and torch.isfinite(e2.llm_pooled(ids, torch.tensor([6, 4]))).all() and e2.fp16_saturations() == 0
    e2.close()


def test_policy_state_dict_round_trip_with_the_reference_key_names(ckpt, tmp_path, monkeypatch):
    from vla_fastvlm.fastvla import FastVLAConfig, FastVLAPolicy
    from vla_fastvlm.utils import load_policy_from_checkpoint, save_policy_checkpoint
    from vla_fastvlm.utils.checkpoint import BACKBONE_PREFIX
    m, d, w_tower, w_llm, train = ckpt
    monkeypatch.setenv("FASTVLA_SYNTHETIC_TOKENIZER", "1")
    torch.manual_seed(34)
    with pytest.warns(UserWarning):
        pol = FastVLAPolicy(FastVLAConfig(vlm_model_name=str(d), hidden_dim=64, fusion_dim=64)).to(DEV)
    pol.eval()
    img, st = torch.rand(2, 3, 96, 96, device=DEV), torch.randn(2, 14, device=DEV)
    tasks = ["pick up the cube", "open the drawer"]
    with torch.no_grad():
        a0 = pol(img, st, tasks).clone()
    out = save_policy_checkpoint(pol, tmp_path / "step-1", include_backbone=True)
    sd = torch.load(out / "policy_state_dict.pt", map_location="cpu")
    # the reference's names: the head under model.*, the VLM under model.backbone.model.* in canonical (inference-form) keys + tied lm_head
    for k in ("model.state_projection.0.weight", "model.fusion.0.weight", "model.fusion.4.bias", "model.action_head.weight",
              BACKBONE_PREFIX + "model.embed_tokens.weight", BACKBONE_PREFIX + "lm_head.weight", BACKBONE_PREFIX + "model.layers.0.self_attn.q_proj.weight",
              BACKBONE_PREFIX + "model.mm_projector.0.weight", BACKBONE_PREFIX + "model.vision_tower.vision_tower.model.patch_embed.0.reparam_conv.weight"):
        assert k in sd, k
    assert json.loads((out / "policy_config.json").read_text())["vlm_model_name"] == str(d)
    # (a) read back: same actions, bit for bit (the engine is packed from the file's own tensors: same values)
    with pytest.warns(UserWarning):
        again = load_policy_from_checkpoint(str(out)).to(DEV)
    assert again.model.backbone._weights_override is not None
    with torch.no_grad():
        a1 = again(img, st, tasks)
    torch.cuda.synchronize()
    assert torch.equal(a0, a1)
    # (b) the file's VLM tensors are what is used, not the directory's: a checkpoint with another final norm gives other actions that
    # match the oracle evaluated on THAT tensor
    sd2 = dict(sd)
    sd2[BACKBONE_PREFIX + "model.norm.weight"] = sd[BACKBONE_PREFIX + "model.norm.weight"] * 1.5
    d2 = tmp_path / "step-2"
    d2.mkdir()
    (d2 / "policy_config.json").write_text((out / "policy_config.json").read_text())
    torch.save(sd2, d2 / "policy_state_dict.pt")
    with pytest.warns(UserWarning):
        other = load_policy_from_checkpoint(str(d2)).to(DEV)
    text = other.model.backbone._prep_text([t + "\n" for t in tasks], torch.device(DEV))
    with torch.no_grad():
        pooled = other.model.backbone.forward_ids(img, text["input_ids"], text["attention_mask"])
    lc = qwen2.Qwen2Cfg(hidden=m.llm.hidden, layers=m.llm.layers, heads=m.llm.heads, kv_heads=m.llm.kv_heads, head_dim=m.llm.head_dim, inter=m.llm.inter,
                        vocab=m.llm.vocab, rope_theta=m.llm.rope_theta, rms_eps=m.llm.rms_eps)
    wl = {k[len(BACKBONE_PREFIX):]: v.float() for k, v in sd2.items() if k.startswith(BACKBONE_PREFIX + "model.") and "vision_tower" not in k and "mm_projector" not in k}
    with torch.no_grad():
        ref = qwen2.llm_pooled(wl, text["input_ids"].cpu().long(), text["attention_mask"].cpu().long(), lc)
    torch.cuda.synchronize()
    assert rel_l2(pooled.cpu(), ref) <= 3e-4
    for pl in (pol, again, other):
        pl.model.backbone.engine().close()
