"""CPU: the host-side mirror keeps the reference's public contract (goldens generated from the imported reference)."""
import dataclasses
import json

import numpy as np
import pytest
import torch

import fastvla_hip
from vla_fastvlm.fastvla import FastVLAConfig, FastVLAPolicy
from vla_fastvlm.fastvla.processor_fastvla import FastVLAProcessor
from vla_fastvlm.model.fastvlm_adapter import (FastVLMBackbone, FastVLMBackboneConfig, canonical_bchw,
                                               infer_size_from_tower_name)
from vla_fastvlm.tokenization import SyntheticTokenizer
from vla_fastvlm.training import TrainingConfig
from vla_fastvlm.training.trainer import linear_warmup_decay


def test_config_contract(golden_dir):
    g = json.loads((golden_dir / "g7_config_contract.json").read_text())

    def norm(d):
        return json.loads(json.dumps(d, default=list))
    assert norm(dataclasses.asdict(FastVLAConfig())) == g["FastVLAConfig"]
    assert norm(dataclasses.asdict(FastVLMBackboneConfig())) == g["FastVLMBackboneConfig"]
    assert norm(dataclasses.asdict(TrainingConfig())) == g["TrainingConfig"]
    bc = FastVLAConfig(vlm_model_name="m", image_size=1024, pad_value=0.5, tokenizer_max_length=48).to_backbone_config()
    assert norm(dataclasses.asdict(bc)) == g["to_backbone_config"]


def test_tower_name_table(golden_dir):
    for row in json.loads((golden_dir / "g5_tower_names.json").read_text()):
        assert infer_size_from_tower_name(row["name"]) == row["size"], row


def test_task_table(golden_dir):
    a, b = FastVLAProcessor(FastVLAConfig(), None), FastVLAProcessor(FastVLAConfig(add_trailing_newline=False), None)
    for row in json.loads((golden_dir / "g4_tasks.json").read_text()):
        assert a.normalize_tasks(row["tasks"], row["batch"]) == row["out"]
        assert b.normalize_tasks(row["tasks"], row["batch"]) == row["out_nonewline"]


def test_pool_hidden_api(golden_dir):
    g = np.load(golden_dir / "g2_pool.npz")
    hid, mask = torch.from_numpy(g["hidden"]), torch.from_numpy(g["mask"])
    for mode, key in (("last_token", "last"), ("mean_pool", "mean")):
        np.testing.assert_allclose(FastVLMBackbone._pool_hidden(hid, mask, mode).numpy(), g[key], atol=1e-7)
        np.testing.assert_allclose(FastVLMBackbone._pool_hidden(hid, None, mode).numpy(), g[key + "_nomask"], atol=1e-7)


def test_schedule(golden_dir):
    for key, row in json.loads((golden_dir / "g6_trainer_schedule.json").read_text()).items():
        total, ratio = key.split("_")
        for s, v in zip(row["steps"], row["values"]):
            assert linear_warmup_decay(s, int(total), float(ratio)) == pytest.approx(v, abs=1e-12)


def test_policy_surface_and_state_dict_keys():
    pol = FastVLAPolicy(FastVLAConfig(vlm_model_name="synthetic:tiny", hidden_dim=32, fusion_dim=48, state_dim=6, action_dim=5))
    keys = set(pol.state_dict())
    assert keys == {f"model.{k}" for k in fastvla_hip.HEAD_KEYS}
    shapes = {k: tuple(v.shape) for k, v in pol.state_dict().items()}
    assert shapes["model.fusion.0.weight"] == (48, 128 + 32) and shapes["model.action_head.weight"] == (5, 48)
    assert pol.model.backbone.output_dim == 128 and pol.model.backbone.expected_size == 256
    assert pol.name == "fastvla" and pol.config_class is FastVLAConfig
    assert sum(p.numel() for p in pol.parameters()) == sum(v.numel() for v in pol.state_dict().values())
    pol.reset()
    for name in ("forward", "compute_loss", "select_action", "fused_train_step"):
        assert callable(getattr(pol, name))
    # full-size head: 3,048,490 trainable parameters
    full = FastVLAPolicy(FastVLAConfig(vlm_model_name="synthetic:fastvlm-0.5b"))
    assert sum(p.numel() for p in full.parameters()) == 3_048_490


def test_errors_match_reference_types(monkeypatch):
    with pytest.raises(ValueError, match="too small"):
        FastVLMBackbone(FastVLMBackboneConfig(model_id="synthetic:fastvlm-0.5b", force_image_size=336))
    with pytest.raises(OSError):
        monkeypatch.delenv("FASTVLA_SYNTHETIC_WEIGHTS", raising=False)
        FastVLMBackbone(FastVLMBackboneConfig(model_id="apple/FastVLM-0.5B"))
    monkeypatch.setenv("FASTVLA_SYNTHETIC_WEIGHTS", "1")
    bb = FastVLMBackbone(FastVLMBackboneConfig(model_id="apple/FastVLM-0.5B"))
    assert bb.expected_size == 1024 and bb.output_dim == 896
    bb7 = FastVLMBackbone(FastVLMBackboneConfig(model_id="apple/FastVLM-7B", force_image_size=1024))
    assert bb7.output_dim == 3584
    if not torch.cuda.is_available():  # the product path must fail loudly without a HIP device
        with pytest.raises(fastvla_hip.FastVLAHipError):
            bb.forward(torch.zeros(1, 3, 8, 8), ["x\n"])


def test_canonical_bchw_shapes():
    assert canonical_bchw(torch.zeros(2, 5, 7, 3)).shape == (2, 3, 5, 7)  # BHWC
    assert canonical_bchw(torch.zeros(3, 5, 7)).shape == (1, 3, 5, 7)      # CHW
    assert canonical_bchw(torch.zeros(5, 7, 3)).shape == (1, 3, 5, 7)      # HWC
    assert canonical_bchw(torch.zeros(5, 7)).shape == (1, 1, 5, 7)
    assert canonical_bchw([np.zeros((5, 7, 3), dtype=np.uint8)] * 2).shape == (2, 3, 5, 7)
    assert canonical_bchw(torch.zeros(2, 3, 4, 4, dtype=torch.uint8)).dtype == torch.uint8
    with pytest.raises(ValueError):
        canonical_bchw([torch.zeros(1, 2, 3, 4, 5)])


def test_synthetic_tokenizer_contract():
    tok = SyntheticTokenizer(512)
    out = tok(["pick up\n", "a\n"], padding="longest", truncation=True, max_length=5, return_tensors="pt")
    assert out["input_ids"].shape == (2, 5) and out["attention_mask"].tolist() == [[1] * 5, [1, 1, 0, 0, 0]]
    assert torch.equal(tok(["a\n"])["input_ids"], tok(["a\n"])["input_ids"])
    out = tok(["ab"], padding="max_length", max_length=4)
    assert out["input_ids"].shape == (1, 4) and out["attention_mask"].tolist() == [[1, 1, 0, 0]]
    assert int(out["input_ids"].max()) < 512


def test_lerobot_plugin_semantics():
    from vla_fastvlm.lerobot_fastvla import FastVLAConfig as LRConfig, FastVLAPolicy as LRPolicy
    from vla_fastvlm.lerobot_fastvla._lerobot_compat import HAVE_LEROBOT, FeatureType, PolicyFeature
    if HAVE_LEROBOT:
        pytest.skip("stand-in semantics are only exercised without lerobot")
    with pytest.raises(ValueError):
        LRConfig(chunk_size=1, n_action_steps=2)
    feats = {"observation.images.top": PolicyFeature(FeatureType.VISUAL, (3, 96, 96)),
             "observation.images.wrist": PolicyFeature(FeatureType.VISUAL, (3, 96, 96)),
             "observation.state": PolicyFeature(FeatureType.STATE, (6,))}
    cfg = LRConfig(vlm_model_name="synthetic:tiny", hidden_dim=16, fusion_dim=16, input_features=feats,
                   output_features={"action": PolicyFeature(FeatureType.ACTION, (5,))})
    assert cfg.observation_delta_indices == [0] and cfg.action_delta_indices == [0] and cfg.reward_delta_indices is None
    assert cfg.get_optimizer_preset().lr == 1e-4 and cfg.get_scheduler_preset().num_decay_steps == 20_000
    pol = LRPolicy(cfg)
    assert cfg.state_dim == 6 and cfg.action_dim == 5 and pol.name == "fastvla"
    batch = {"observation.images.top": torch.zeros(2, 4, 3, 8, 8) + torch.arange(4).view(1, 4, 1, 1, 1),
             "observation.images.wrist": torch.ones(2, 3, 8, 8), "observation.state": torch.arange(2 * 3 * 6).view(2, 3, 6).float(),
             "task": ["lift"]}
    images, states, tasks = pol._prepare_inputs(batch)
    assert images.shape == (2, 3, 8, 8) and float(images.mean()) == 3.0  # FIRST camera, LAST timestep
    assert torch.equal(states, batch["observation.state"][:, -1]) and tasks == ["lift\n", "lift\n"]
    assert pol._prepare_inputs({**batch, "task": None})[2] == ["\n", "\n"]
    with pytest.raises(ValueError):
        LRPolicy(LRConfig(vlm_model_name="synthetic:tiny", input_features={"s": PolicyFeature(FeatureType.STATE, (6,))}))
    # action deque: one prediction per env step with n_action_steps = 1
    calls = []
    pol._predict_actions = lambda b: calls.append(1) or torch.full((2, 5), float(len(calls)))
    a1, a2 = pol.select_action(batch), pol.select_action(batch)
    assert a1.shape == (2, 5) and float(a1[0, 0]) == 1.0 and float(a2[0, 0]) == 2.0
    pol.reset()
    assert len(pol._action_queue) == 0


def test_checkpoint_roundtrip(tmp_path):
    from vla_fastvlm.utils import load_policy_from_checkpoint
    pol = FastVLAPolicy(FastVLAConfig(vlm_model_name="synthetic:tiny", hidden_dim=16, fusion_dim=16))
    (tmp_path / "policy_config.json").write_text(json.dumps(dataclasses.asdict(pol.config)))
    sd = {k: v.clone() for k, v in pol.state_dict().items()}
    sd["model.backbone.model.lm_head.weight"] = torch.zeros(2, 2)  # reference checkpoints carry the frozen VLM: ignored
    torch.save(sd, tmp_path / "policy_state_dict.pt")
    again = load_policy_from_checkpoint(str(tmp_path))
    for k, v in pol.state_dict().items():
        assert torch.equal(v, again.state_dict()[k])


def test_hf_checkpoint_directory_is_resolved(tmp_path, monkeypatch):
    """A local llava_qwen2 directory (config.json + safetensors with canonical inference-form keys) -> architecture and
    tensors for fv_load_weights; lm_head is dropped."""
    from safetensors.torch import save_file
    from fastvla_hip import arch, weights
    from vla_fastvlm.model.fastvlm_adapter import arch_from_hf_config, load_hf_checkpoint_dir
    m = arch.preset("tiny")
    cfg = dict(model_type="llava_qwen2", hidden_size=m.llm.hidden, num_hidden_layers=m.llm.layers, num_attention_heads=m.llm.heads,
               num_key_value_heads=m.llm.kv_heads, intermediate_size=m.llm.inter, vocab_size=m.llm.vocab, rope_theta=1e6,
               rms_norm_eps=1e-6, mm_vision_tower="mobileclip_l_1024", torch_dtype="bfloat16")
    d = tmp_path / "llava-fastvithd_tiny"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(cfg))
    state = weights.init_backbone(m, seed=1)
    state["lm_head.weight"] = torch.zeros(4, 4)
    keys = sorted(state)
    save_file({k: state[k].contiguous() for k in keys[: len(keys) // 2]}, str(d / "model-00001-of-00002.safetensors"))
    save_file({k: state[k].contiguous() for k in keys[len(keys) // 2:]}, str(d / "model-00002-of-00002.safetensors"))
    got = arch_from_hf_config(d / "config.json")
    assert got.llm == m.llm and got.tower.image_size == 1024 and got.tower.layers == (2, 12, 24, 4, 2)
    loaded = load_hf_checkpoint_dir(d)
    assert "lm_head.weight" not in loaded and set(loaded) == set(state) - {"lm_head.weight"}
    assert torch.equal(loaded["model.norm.weight"], state["model.norm.weight"])
    monkeypatch.setenv("FASTVLA_SYNTHETIC_TOKENIZER", "1")  # this directory ships no tokenizer files (next test: that raises)
    with pytest.warns(UserWarning):
        bb = FastVLMBackbone(FastVLMBackboneConfig(model_id=str(d)))
    assert bb.output_dim == m.llm.hidden and bb.expected_size == 1024 and bb._weights_source[0] == "hf_dir"


def test_real_checkpoint_without_tokenizer_raises(tmp_path, monkeypatch):
    """a real checkpoint whose tokenizer files are missing must raise (reference fastvlm_adapter.py:366-367),
    not run on hashed ids; only synthetic weights (or an explicit opt-in) get the SyntheticTokenizer."""
    import json
    from vla_fastvlm.model.fastvlm_adapter import FastVLMBackbone, FastVLMBackboneConfig
    d = tmp_path / "llava-fastvithd_0.5b_stage3"
    d.mkdir()
    (d / "config.json").write_text(json.dumps(dict(model_type="llava_qwen2", hidden_size=896, num_hidden_layers=24, num_attention_heads=14,
                                                   num_key_value_heads=2, intermediate_size=4864, vocab_size=151936,
                                                   mm_vision_tower="mobileclip_l_1024")))
    (d / "model.safetensors").write_bytes(b"")
    monkeypatch.delenv("FASTVLA_SYNTHETIC_TOKENIZER", raising=False)
    with pytest.raises(RuntimeError, match="Tokenizer is missing"):
        FastVLMBackbone(FastVLMBackboneConfig(model_id=str(d)))
    monkeypatch.setenv("FASTVLA_SYNTHETIC_TOKENIZER", "1")
    with pytest.warns(UserWarning, match="SyntheticTokenizer"):
        bb = FastVLMBackbone(FastVLMBackboneConfig(model_id=str(d)))
    assert isinstance(bb.tokenizer, SyntheticTokenizer)
    assert isinstance(FastVLMBackbone(FastVLMBackboneConfig(model_id="synthetic:tiny")).tokenizer, SyntheticTokenizer)


def test_hf_checkpoint_provider_streams_one_tensor_at_a_time(tmp_path):
    """The streaming form of the HF-directory loader (fv_load_weights_cb's provider): every key of every shard by name, bf16 kept
    as bf16, `lm_head.*` never read, unknown keys -> None (the packer reports them as missing)."""
    from safetensors.torch import save_file
    from fastvla_hip import arch, weights
    from vla_fastvlm.model.fastvlm_adapter import hf_checkpoint_provider
    m = arch.preset("tiny")
    state = weights.init_backbone(m, seed=2)
    state["model.layers.0.mlp.down_proj.weight"] = state["model.layers.0.mlp.down_proj.weight"].to(torch.bfloat16)
    state["lm_head.weight"] = torch.zeros(4, 4)
    keys = sorted(state)
    d = tmp_path / "ckpt"
    d.mkdir()
    save_file({k: state[k].contiguous() for k in keys[::2]}, str(d / "model-00001-of-00002.safetensors"))
    save_file({k: state[k].contiguous() for k in keys[1::2]}, str(d / "model-00002-of-00002.safetensors"))
    prov = hf_checkpoint_provider(d)
    for k in keys:
        if k == "lm_head.weight":
            assert prov(k) is None
        else:
            t = prov(k)
            assert t.dtype == state[k].dtype and torch.equal(t, state[k]), k
    assert prov("model.layers.99.mlp.up_proj.weight") is None


def test_folded_statistics_travel_with_the_state_dict():
    """STATE / ACTION statistics folded into the head are state of the policy -- state_dict() carries them while the
    folding is on (and only then: an unfolded policy keeps exactly the reference's keys), load_state_dict() re-applies them."""
    cfg = FastVLAConfig(vlm_model_name="synthetic:tiny", hidden_dim=16, fusion_dim=16, state_dim=6, action_dim=5)
    pol = FastVLAPolicy(cfg)
    plain = set(pol.state_dict())
    assert not any("io_norm" in k for k in plain)
    stats = dict(state_mean=torch.arange(6.0), state_std=torch.ones(6) * 2, action_mean=-torch.arange(5.0), action_std=torch.ones(5) * 3)
    pol.model.backbone.set_io_normalization(**stats)
    sd = pol.state_dict()
    extra = set(sd) - plain
    assert extra == {f"model.backbone.io_norm.{k}" for k in ("state_mean", "state_std", "action_mean", "action_std", "eps")}
    again = FastVLAPolicy(cfg)
    assert again.model.backbone._io_norm is None
    again.load_state_dict(sd)   # strict: the io_norm.* keys are consumed, not "unexpected"
    got = again.model.backbone._io_norm
    assert got is not None and abs(got["eps"] - 1e-8) < 1e-12
    for k, v in stats.items():
        assert torch.equal(got[k], v)
    # an unfolded checkpoint leaves a folded policy's statistics alone and loads strictly
    again.load_state_dict({k: v for k, v in sd.items() if "io_norm" not in k})
    del sd["model.backbone.io_norm.eps"]
    with pytest.raises(RuntimeError, match="incomplete folded normalisation"):
        FastVLAPolicy(cfg).load_state_dict(sd)


def test_trainer_schedule_counts_micro_batches_like_the_reference(tmp_path):
    """with gradient accumulation the reference's LambdaLR (not wrapped by accelerate) advances once per MICRO-batch and
    max_steps counts micro-batches (training/trainer.py:180-182,203): LR index = global_step, only the update waits for the k-th."""
    from vla_fastvlm.training import Trainer, TrainingConfig
    from vla_fastvlm.training.trainer import linear_warmup_decay

    class Fake(torch.nn.Module):
        config = FastVLAConfig(vlm_model_name="synthetic:tiny")

        def __init__(self):
            super().__init__()
            self.calls, self.micro = [], 0

        def fused_train_step(self, batch, *, prepared=None, lr, grad_accum_steps=1, force_sync=False, next_batch=None, **kw):
            self.micro += 1
            synced = self.micro % grad_accum_steps == 0 or force_sync
            if synced:
                self.micro = 0
            self.calls.append((lr, synced))
            return {"loss": 0.0, "mse": 0.0, "grad_norm": 1.0 if synced else float("nan"), "synced": synced, "next": None}

    data = [{"i": i} for i in range(6)]
    fake = Fake()
    tr = Trainer(fake, data, None, TrainingConfig(output_dir=str(tmp_path), num_epochs=3, gradient_accumulation_steps=2, learning_rate=1.0,
                                                   warmup_ratio=0.25, max_steps=8, logging_steps=1, save_steps=10 ** 6, eval_steps=10 ** 6))
    tr._sync_replicas = lambda: None
    tr.fit()
    assert tr.global_step == 8 and tr.update_step == 4 and len(fake.calls) == 8
    assert [lr for lr, _ in fake.calls] == [linear_warmup_decay(g, 8, 0.25) for g in range(8)]
    assert [s for _, s in fake.calls] == [False, True] * 4
    logged = [json.loads(l) for l in (tmp_path / "logs" / "metrics.jsonl").read_text().splitlines()]
    # the norm of the LAST closed update is logged on every logging step once one exists (k = 2: from the second micro-batch on),
    # not only on the steps that happen to close an update
    assert all(("train/grad_norm" in r) == (i >= 1) for i, r in enumerate(logged))
