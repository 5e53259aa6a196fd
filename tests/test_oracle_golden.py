"""CPU: the oracle against the fixtures generated from the imported reference (tests/golden/make_golden.py)."""
import json

import numpy as np
import pytest
import torch

from oracle import head, policy, preprocess, qwen2


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_letterbox_small_cases(golden_dir):
    g = np.load(golden_dir / "g1_letterbox.npz")
    for k, size, pad in (("a", 64, 0.0), ("b", 64, 0.0), ("c", 48, -1.0), ("gray", 64, 0.25), ("rgba", 64, 0.25)):
        out = preprocess.letterbox(_t(g[k]), size, pad)
        ref = _t(g[k + "_out"])
        assert out.shape == ref.shape
        scale = max(1.0, float(ref.abs().max()))
        assert float((out - ref).abs().max()) <= 2e-6 * scale, k
    # BHWC input is permuted by _as_bchw before the letterbox (fastvlm_adapter.py:423-428)
    out = preprocess.letterbox(_t(g["bhwc"]).permute(0, 3, 1, 2), 64, 0.25)
    assert float((out - _t(g["bhwc_out"])).abs().max()) <= 2e-6


def test_letterbox_headline_size(golden_dir):
    g = np.load(golden_dir / "g1_letterbox.npz")
    ys, xs = _t(g["big_ys"]), _t(g["big_xs"])
    torch.manual_seed(int(g["big_seed"]))
    out = preprocess.letterbox(torch.rand(1, 3, 336, 336), 1024)
    assert out.shape == (1, 3, 1024, 1024)
    np.testing.assert_allclose(out.double().sum(dim=(0, 2, 3)).numpy(), g["big_sums"], rtol=1e-7)
    np.testing.assert_allclose(out[0][:, ys, xs].numpy(), g["big_samples"], atol=2e-6)
    torch.manual_seed(int(g["ns_seed"]))
    out = preprocess.letterbox(torch.rand(1, 3, 240, 320), 1024)
    np.testing.assert_allclose(out.double().sum(dim=(0, 2, 3)).numpy(), g["ns_sums"], rtol=1e-7)
    np.testing.assert_allclose(out[0][:, ys, xs].numpy(), g["ns_samples"], atol=2e-6)
    first = int((out[0, 0].abs().sum(dim=1) > 0).nonzero()[0])
    assert first == int(g["ns_first_row"]) == 1024 - 768  # padding goes on TOP


def test_normalize_imagenet_golden(golden_dir):
    """_maybe_normalize_imagenet (reference model/fastvlm_adapter.py:463-477) after the letterbox: the oracle against what the imported reference produced
    (the branch recorded in the file: no torchvision in the build container -> :466-470).  Where the batch maximum is <= 1.5 the torchvision branch is the
    same arithmetic, so those cases pin it as well; beyond 1.5 it divides by 255 first (restatement only)."""
    g = np.load(golden_dir / "g1_normalize.npz")
    tv = bool(int(g["has_torchvision"]))
    for k in ("unit", "gray", "wide"):
        out = preprocess.prepare_images(_t(g[k]), 64, 0.25, True, normalize=True, torchvision_branch=tv)
        ref = _t(g[k + "_out"])
        assert out.shape == ref.shape
        assert float((out - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max())), k
    for k in ("unit", "gray"):   # max <= 1.5 (values in [0, 1), pad 0.25): both branches agree bit for bit
        a = preprocess.prepare_images(_t(g[k]), 64, 0.25, True, normalize=True, torchvision_branch=True)
        b = preprocess.prepare_images(_t(g[k]), 64, 0.25, True, normalize=True, torchvision_branch=False)
        assert torch.equal(a, b)
    w = preprocess.prepare_images(_t(g["wide"]), 64, 0.25, True, normalize=True, torchvision_branch=True)
    lb = preprocess.letterbox(_t(g["wide"]), 64, 0.25)
    assert float(lb.max()) > 1.5 and torch.equal(w, preprocess.normalize_imagenet(lb / 255.0, torchvision_branch=False))


def test_letterbox_rejects_non_4d():
    with pytest.raises(ValueError):
        preprocess.letterbox(torch.zeros(3, 8, 8), 16)


def test_pool(golden_dir):
    g = np.load(golden_dir / "g2_pool.npz")
    hid, mask = _t(g["hidden"]), _t(g["mask"])
    np.testing.assert_allclose(qwen2.pool_hidden(hid, mask, "last_token").numpy(), g["last"], atol=0)
    np.testing.assert_allclose(qwen2.pool_hidden(hid, mask, "mean_pool").numpy(), g["mean"], atol=1e-7)
    np.testing.assert_allclose(qwen2.pool_hidden(hid, None, "last_token").numpy(), g["last_nomask"], atol=0)
    np.testing.assert_allclose(qwen2.pool_hidden(hid, None, "mean_pool").numpy(), g["mean_nomask"], atol=1e-7)


@pytest.mark.parametrize("name", ["g3_head_small.npz", "g3_head_metaworld.npz", "g3_head_b1.npz"])
def test_head_forward_backward_step(golden_dir, name):
    g = np.load(golden_dir / name)
    p = {k[2:]: _t(g[k]) for k in g.files if k.startswith("p.")}
    assert set(p) == set(head.HEAD_KEYS)
    feats, states, targets = _t(g["feats"]), _t(g["states"]), _t(g["targets"])
    act = head.head_forward(p, feats, states)
    np.testing.assert_allclose(act.numpy(), g["actions"], rtol=2e-5, atol=2e-6)
    pred, cache = head.head_forward(p, feats, states, keep_cache=True)
    loss, grads = head.head_mse_backward(p, cache, pred, targets)
    np.testing.assert_allclose(float(loss), float(g["loss"]), rtol=1e-6)
    for k in head.HEAD_KEYS:
        ref = g["g." + k]
        tol = 1e-5 * max(1e-3, float(np.abs(ref).max()))
        assert float((grads[k] - _t(ref)).abs().max()) <= tol, k
    for tag, (lr, wd) in {"lerobot": (1e-4, 1e-4), "trainer": (3e-4, 0.01)}.items():
        clipped, norm = head.clip_grad_norm({k: _t(g["g." + k]) for k in head.HEAD_KEYS}, 1.0)
        np.testing.assert_allclose(float(norm), float(g[f"norm.{tag}"]), rtol=1e-5)
        zeros = {k: torch.zeros_like(v) for k, v in p.items()}
        newp, _, _ = head.adamw_step(p, clipped, zeros, zeros, 1, lr, (0.9, 0.95), 1e-8, wd)
        for k in head.HEAD_KEYS:
            np.testing.assert_allclose(newp[k].numpy(), g[f"step.{tag}." + k], rtol=0, atol=2e-7, err_msg=k)


def test_task_tables(golden_dir):
    for row in json.loads((golden_dir / "g4_tasks.json").read_text()):
        assert policy.normalize_tasks(row["tasks"], row["batch"], True) == row["out"]
        assert policy.normalize_tasks(row["tasks"], row["batch"], False) == row["out_nonewline"]


def test_trainer_schedule(golden_dir):
    for key, row in json.loads((golden_dir / "g6_trainer_schedule.json").read_text()).items():
        total, ratio = key.split("_")
        for s, v in zip(row["steps"], row["values"]):
            assert head.trainer_lr_lambda(s, int(total), float(ratio)) == pytest.approx(v, abs=1e-12)


def test_lerobot_semantics():
    # lerobot_fastvla/modeling_fastvla.py:82-105,119-125 (cannot be imported: lerobot missing) -- restated behaviour
    assert policy.lerobot_tasks(None, 2) == ["\n", "\n"]
    assert policy.lerobot_tasks("go", 2) == ["go\n", "go\n"]
    assert policy.lerobot_tasks(["a"], 3) == ["a\n"] * 3
    assert policy.lerobot_tasks(7, 1) == ["7\n"]
    x = torch.arange(2 * 3 * 4).view(2, 3, 4)
    assert torch.equal(policy.last_timestep(x, 2), x[:, -1])
    q = policy.ActionQueue(1)
    calls = []

    def chunk():
        calls.append(1)
        return torch.full((2, 1, 3), float(len(calls)))

    a1 = q.select(chunk)
    a2 = q.select(chunk)
    assert a1.shape == (2, 3) and float(a1[0, 0]) == 1.0 and float(a2[0, 0]) == 2.0 and len(calls) == 2
