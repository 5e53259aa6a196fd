#!/usr/bin/env python
"""Generate the golden fixtures in this directory by IMPORTING the reference (CPU, build container only).

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference/src python tests/golden/make_golden.py

Only inputs and expected outputs are stored (small .npz / .json); nothing of the reference's source travels.
What is driven:
  G1 letterbox        vla_fastvlm.model.fastvlm_adapter.resize_with_pad / _prepare_images_tensor
  G1b normalisation   FastVLMBackbone._maybe_normalize_imagenet through _prepare_images_tensor (the branch the container's reference takes)
  G2 pooling          FastVLMBackbone._pool_hidden (both modes, incl. an all-pad row)
  G3 head             FastVLMWithExpert (stub backbone) + FastVLAPolicy.compute_loss -> actions, loss, 12 grads
  G4 task table       FastVLAProcessor.normalize_tasks
  G5 tower-name table FastVLMBackbone._infer_size_from_tower_name
  G7 config contract  dataclasses.asdict of FastVLAConfig / FastVLMBackboneConfig / TrainingConfig defaults
  G6 train step       clip_grad_norm_ + torch.optim.AdamW exactly as training/trainer.py:60-66,171-182 issues them;
                      Trainer._build_scheduler_lambda
"""
from __future__ import annotations

import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.dont_write_bytecode = True

import vla_fastvlm.fastvla.fastvlm_with_expert as fwe  # noqa: E402
from vla_fastvlm.fastvla.configuration_fastvla import FastVLAConfig  # noqa: E402
from vla_fastvlm.fastvla.processor_fastvla import FastVLAProcessor  # noqa: E402
from vla_fastvlm.model.fastvlm_adapter import FastVLMBackbone, FastVLMBackboneConfig, resize_with_pad  # noqa: E402
from vla_fastvlm.training.trainer import Trainer  # noqa: E402


def npz(name, **arrs):
    np.savez_compressed(HERE / name, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                        for k, v in arrs.items()})


def g1_letterbox():
    torch.manual_seed(101)
    a = torch.rand(2, 3, 30, 40)
    b = torch.rand(1, 3, 21, 21)
    c = torch.rand(1, 3, 50, 20) * 255.0
    bb = FastVLMBackbone.__new__(FastVLMBackbone)
    torch.nn.Module.__init__(bb)
    bb.config = FastVLMBackboneConfig(pad_value=0.25)
    bb.expected_size = 64
    gray = torch.rand(2, 1, 17, 33)
    rgba = torch.rand(1, 4, 40, 24)
    bhwc = torch.rand(2, 20, 28, 3)
    out = dict(
        a=a, a_out=resize_with_pad(a, 64, 64), b=b, b_out=resize_with_pad(b, 64, 64),
        c=c, c_out=resize_with_pad(c, 48, 48, pad_value=-1.0),
        gray=gray, gray_out=bb._prepare_images_tensor(gray, torch.device("cpu")),
        rgba=rgba, rgba_out=bb._prepare_images_tensor(rgba, torch.device("cpu")),
        bhwc=bhwc, bhwc_out=bb._prepare_images_tensor(bhwc, torch.device("cpu")),
    )
    # headline size 336^2 -> 1024^2: per-channel sums + 64 sampled pixels
    torch.manual_seed(102)
    big = torch.rand(1, 3, 336, 336)
    big_out = resize_with_pad(big, 1024, 1024)
    g = torch.Generator().manual_seed(103)
    ys = torch.randint(0, 1024, (64,), generator=g)
    xs = torch.randint(0, 1024, (64,), generator=g)
    out.update(big_seed=102, big_sums=big_out.double().sum(dim=(0, 2, 3)), big_ys=ys, big_xs=xs,
               big_samples=big_out[0][:, ys, xs])
    # non-square headline-like: 240x320 -> 1024
    torch.manual_seed(104)
    ns = torch.rand(1, 3, 240, 320)
    ns_out = resize_with_pad(ns, 1024, 1024)
    out.update(ns_seed=104, ns_sums=ns_out.double().sum(dim=(0, 2, 3)), ns_samples=ns_out[0][:, ys, xs],
               ns_first_row=int((ns_out[0, 0].abs().sum(dim=1) > 0).nonzero()[0]))
    npz("g1_letterbox.npz", **out)


def g1_normalize():
    """_maybe_normalize_imagenet (fastvlm_adapter.py:463-477) through _prepare_images_tensor, normalize_imagenet=True.  torchvision is absent from the build
    container, so the imported reference takes its :466-470 branch ((x - mean) / std, no value-range test); the file records which branch ran.  The
    torchvision branch (:471-477) is the same arithmetic once x.max() <= 1.5 -- the 0..1 cases below pin it too -- and divides by 255 first otherwise
    (restated in oracle/preprocess.py, not drivable here)."""
    import vla_fastvlm.model.fastvlm_adapter as fa
    torch.manual_seed(111)
    bb = FastVLMBackbone.__new__(FastVLMBackbone)
    torch.nn.Module.__init__(bb)
    bb.config = FastVLMBackboneConfig(pad_value=0.25, normalize_imagenet=True)
    bb.expected_size = 64
    unit = torch.rand(2, 3, 30, 40)
    gray = torch.rand(1, 1, 17, 33)
    wide = torch.rand(1, 3, 50, 20) * 255.0
    npz("g1_normalize.npz", has_torchvision=int(fa._HAS_TV),
        unit=unit, unit_out=bb._prepare_images_tensor(unit, torch.device("cpu")),
        gray=gray, gray_out=bb._prepare_images_tensor(gray, torch.device("cpu")),
        wide=wide, wide_out=bb._prepare_images_tensor(wide, torch.device("cpu")))


def g2_pool():
    torch.manual_seed(201)
    hid = torch.randn(3, 5, 8)
    mask = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1], [0, 0, 0, 0, 0]])
    npz("g2_pool.npz", hidden=hid, mask=mask,
        last=FastVLMBackbone._pool_hidden(hid, mask, "last_token"),
        mean=FastVLMBackbone._pool_hidden(hid, mask, "mean_pool"),
        last_nomask=FastVLMBackbone._pool_hidden(hid, None, "last_token"),
        mean_nomask=FastVLMBackbone._pool_hidden(hid, None, "mean_pool"))


class _StubBackbone(torch.nn.Module):
    """Stands in for FastVLMBackbone (cannot be built offline): returns a fixed feature matrix."""

    def __init__(self, cfg):
        super().__init__()
        self.output_dim = _StubBackbone.dim
        self.feats = None

    def forward(self, images, tasks, device=None):
        return self.feats

    def _prepare_images_tensor(self, images, device):
        return images


def _head_case(name, feat_dim, hidden, fusion, ds, da, batch, seed, full):
    _StubBackbone.dim = feat_dim
    orig = fwe.FastVLMBackbone
    fwe.FastVLMBackbone = _StubBackbone
    try:
        from vla_fastvlm.fastvla.modeling_fastvla import FastVLAPolicy
        torch.manual_seed(seed)
        cfg = FastVLAConfig(state_dim=ds, action_dim=da, hidden_dim=hidden, fusion_dim=fusion, dropout=0.0)
        pol = FastVLAPolicy(cfg)
        with torch.no_grad():  # non-trivial LayerNorm affine so their grads are exercised
            for n, p in pol.named_parameters():
                if n.endswith("0.weight") and p.ndim == 1 or n.endswith("fusion.1.weight"):
                    p.add_(0.1 * torch.randn_like(p))
                if p.ndim == 1 and "bias" in n:
                    p.add_(0.05 * torch.randn_like(p))
        feats = torch.randn(batch, feat_dim)
        states = torch.randn(batch, ds) * 2.0 + 0.3
        targets = torch.randn(batch, da)
        pol.model.backbone.feats = feats
        pol.train()
        out = pol.compute_loss(dict(images=torch.zeros(batch, 3, 8, 8), states=states, actions=targets,
                                    tasks=["t"] * batch))
        out["loss"].backward()
        params = {n[len("model."):]: p.detach().clone() for n, p in pol.named_parameters()}
        grads = {n[len("model."):]: p.grad.detach().clone() for n, p in pol.named_parameters()}
        pol.eval()
        with torch.no_grad():
            actions = pol.forward(torch.zeros(batch, 3, 8, 8), states, ["t"])
            sel = pol.select_action(torch.zeros(3, 8, 8), states[0], "t", torch.device("cpu")) if batch == 1 else None
        # one optimiser step exactly as trainer.py does it (both presets)
        steps = {}
        for tag, (lr, wd) in {"lerobot": (1e-4, 1e-4), "trainer": (3e-4, 0.01)}.items():
            ps = [p.detach().clone().requires_grad_(True) for p in pol.parameters()]
            for q, p in zip(ps, pol.parameters()):
                q.grad = p.grad.detach().clone()
            opt = torch.optim.AdamW(ps, lr=lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=wd)
            norm = torch.nn.utils.clip_grad_norm_(ps, 1.0)
            opt.step()
            steps[tag] = (norm, {n[len("model."):]: q.detach() for (n, _), q in zip(pol.named_parameters(), ps)})
        arrs = dict(feats=feats, states=states, targets=targets, loss=out["loss"].detach(), actions=actions,
                    dims=np.array([feat_dim, hidden, fusion, ds, da, batch]))
        if full:
            for k, v in params.items():
                arrs["p." + k] = v
            for k, v in grads.items():
                arrs["g." + k] = v
            for tag, (norm, ps) in steps.items():
                arrs[f"norm.{tag}"] = norm
                for k, v in ps.items():
                    arrs[f"step.{tag}." + k] = v
        else:  # full-size config: checksums only (weights are re-derivable from the seed by the same torch init)
            for k, v in grads.items():
                arrs["gsum." + k] = v.double().sum()
                arrs["gabs." + k] = v.double().abs().sum()
            arrs["seed"] = seed
            for tag, (norm, ps) in steps.items():
                arrs[f"norm.{tag}"] = norm
        npz(name, **arrs)
    finally:
        fwe.FastVLMBackbone = orig


def g3_head():
    _head_case("g3_head_small.npz", feat_dim=32, hidden=64, fusion=64, ds=14, da=14, batch=5, seed=301, full=True)
    _head_case("g3_head_metaworld.npz", feat_dim=48, hidden=32, fusion=40, ds=4, da=4, batch=3, seed=302, full=True)
    _head_case("g3_head_b1.npz", feat_dim=16, hidden=24, fusion=24, ds=6, da=5, batch=1, seed=303, full=True)


def g4_tasks():
    proc = FastVLAProcessor(FastVLAConfig(), backbone=None)
    proc_nonl = FastVLAProcessor(FastVLAConfig(add_trailing_newline=False), backbone=None)
    cases = [("pick", 3), (["pick"], 3), (["a", "b\n", "c"], 3), ("done\n", 1), (["x"], 1), ("", 2)]
    table = []
    for tasks, bs in cases:
        table.append(dict(tasks=tasks, batch=bs, out=proc.normalize_tasks(tasks, bs),
                          out_nonewline=proc_nonl.normalize_tasks(tasks, bs)))
    (HERE / "g4_tasks.json").write_text(json.dumps(table, indent=1))


def g5_tower_names():
    names = ["mobileclip_l_1024", "google/siglip-so400m-patch14-384", "openai/clip-vit-large-patch14-336",
             "fastvithd", "mobileclip_l_384", "vit_b_16", "foo-512-bar", "model_8192", "so400m", None, 17]
    table = [dict(name=n, size=FastVLMBackbone._infer_size_from_tower_name(n)) for n in names]
    (HERE / "g5_tower_names.json").write_text(json.dumps(table, indent=1))


def g6_schedules():
    tr = Trainer.__new__(Trainer)
    out = {}
    for total, ratio in [(1000, 0.03), (10, 0.03), (100000, 0.03), (50, 0.5)]:
        lam = tr._build_scheduler_lambda(total, ratio)
        steps = sorted({0, 1, int(total * ratio), max(int(total * ratio) - 1, 0), total // 2, total - 1, total, total + 5})
        out[f"{total}_{ratio}"] = dict(steps=steps, values=[lam(s) for s in steps])
    (HERE / "g6_trainer_schedule.json").write_text(json.dumps(out, indent=1))


def g7_config_contract():
    """Public config contract: field names + defaults of the dataclasses the mirror must keep verbatim."""
    import dataclasses
    from vla_fastvlm.training.trainer import TrainingConfig
    out = {
        "FastVLAConfig": dataclasses.asdict(FastVLAConfig()),
        "FastVLMBackboneConfig": dataclasses.asdict(FastVLMBackboneConfig()),
        "TrainingConfig": dataclasses.asdict(TrainingConfig()),
        "to_backbone_config": dataclasses.asdict(FastVLAConfig(vlm_model_name="m", image_size=1024, pad_value=0.5,
                                                               tokenizer_max_length=48).to_backbone_config()),
    }
    (HERE / "g7_config_contract.json").write_text(json.dumps(out, indent=1, default=list))


if __name__ == "__main__":
    os.environ.setdefault("HF_HUB_OFFLINE", "1")
    torch.set_num_threads(4)
    g1_letterbox(); g1_normalize(); g2_pool(); g3_head(); g4_tasks(); g5_tower_names(); g6_schedules(); g7_config_contract()
    print("golden fixtures written to", HERE)
