"""CPU: train-form -> inference-form folding (vla_fastvlm/model/reparam.py) against the unfolded oracle
(oracle/reparam.py): every rule on random eval-mode statistics, then a whole tiny tower converted to training form and back."""
import torch
import torch.nn.functional as F

from fastvla_hip import arch, weights
from oracle import fastvit_hd, reparam as oref
from vla_fastvlm.model import reparam


def _bn(c, g, pre, p):
    p[pre + "weight"] = torch.rand(c, generator=g) + 0.5
    p[pre + "bias"] = torch.randn(c, generator=g) * 0.2
    p[pre + "running_mean"] = torch.randn(c, generator=g) * 0.3
    p[pre + "running_var"] = torch.rand(c, generator=g) + 0.4


def _close(a, b, tol=2e-5):
    assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


def test_mobileone_branches_fold_to_one_conv():
    g = torch.Generator().manual_seed(0)
    for (cin, cout, k, stride, groups, branches, skip) in [(8, 8, 3, 1, 8, 1, True), (3, 16, 3, 2, 1, 2, False), (16, 16, 1, 1, 1, 1, True), (8, 16, 3, 1, 8, 1, False)]:
        p, pre = {}, "blk."
        for i in range(branches):
            p[pre + f"rbr_conv.{i}.conv.weight"] = torch.randn(cout, cin // groups, k, k, generator=g) * 0.3
            _bn(cout, g, pre + f"rbr_conv.{i}.bn.", p)
        if k > 1:
            p[pre + "rbr_scale.conv.weight"] = torch.randn(cout, cin // groups, 1, 1, generator=g) * 0.3
            _bn(cout, g, pre + "rbr_scale.bn.", p)
        if skip:
            _bn(cout, g, pre + "rbr_skip.", p)
        x = torch.randn(2, cin, 9, 10, generator=g)
        ref = oref.mobileone_train(p, pre, x, stride=stride, groups=groups)
        out = reparam.fold_train_form(p)
        assert set(out) == {"blk.reparam_conv.weight", "blk.reparam_conv.bias"}
        _close(F.conv2d(x, out["blk.reparam_conv.weight"], out["blk.reparam_conv.bias"], stride=stride, padding=k // 2, groups=groups), ref)


def test_repmixer_repcpe_and_large_kernel_fold():
    g = torch.Generator().manual_seed(1)
    c = 12
    p, pre = {}, "network.0.0.token_mixer."
    p[pre + "mixer.rbr_conv.0.conv.weight"] = torch.randn(c, 1, 3, 3, generator=g) * 0.3
    _bn(c, g, pre + "mixer.rbr_conv.0.bn.", p)
    p[pre + "mixer.rbr_scale.conv.weight"] = torch.randn(c, 1, 1, 1, generator=g) * 0.3
    _bn(c, g, pre + "mixer.rbr_scale.bn.", p)
    _bn(c, g, pre + "mixer.rbr_skip.", p)
    _bn(c, g, pre + "norm.rbr_skip.", p)
    p[pre + "layer_scale"] = torch.rand(c, 1, 1, generator=g) * 0.2
    p["network.0.0.layer_scale"] = torch.rand(c, 1, 1, generator=g)          # the block's own scale: must pass through
    p["network.3.pe.weight"] = torch.randn(c, 1, 7, 7, generator=g) * 0.05
    p["network.3.pe.bias"] = torch.randn(c, generator=g) * 0.1
    p["network.1.proj.0.lkb_origin.conv.weight"] = torch.randn(2 * c, 1, 7, 7, generator=g) * 0.1
    _bn(2 * c, g, "network.1.proj.0.lkb_origin.bn.", p)
    p["network.1.proj.0.small_conv.conv.weight"] = torch.randn(2 * c, 1, 3, 3, generator=g) * 0.2
    _bn(2 * c, g, "network.1.proj.0.small_conv.bn.", p)
    x = torch.randn(2, c, 10, 12, generator=g)
    out = reparam.fold_train_form(p)
    assert set(out) == {pre + "reparam_conv.weight", pre + "reparam_conv.bias", "network.0.0.layer_scale", "network.3.reparam_conv.weight",
                        "network.3.reparam_conv.bias", "network.1.proj.0.lkb_reparam.weight", "network.1.proj.0.lkb_reparam.bias"}
    _close(F.conv2d(x, out[pre + "reparam_conv.weight"], out[pre + "reparam_conv.bias"], padding=1, groups=c), oref.repmixer_train(p, pre, x))
    _close(F.conv2d(x, out["network.3.reparam_conv.weight"], out["network.3.reparam_conv.bias"], padding=3, groups=c), oref.repcpe_train(p, "network.3.", x))
    _close(F.conv2d(x, out["network.1.proj.0.lkb_reparam.weight"], out["network.1.proj.0.lkb_reparam.bias"], stride=2, padding=3, groups=c),
           oref.lkb_train(p, "network.1.proj.0.", x))
    assert torch.equal(out["network.0.0.layer_scale"], p["network.0.0.layer_scale"])


def _unfold(w, g):
    """inference-form tiny tower -> an equivalent TRAINING-form dict: every reparam_conv becomes conv+BN (+ random scale branch
    that the k x k branch compensates), RepMixer / RepCPE / lkb get their own training forms."""
    VT = fastvit_hd.VT
    out = {}
    for k, v in w.items():
        m = None
        if k.endswith("token_mixer.reparam_conv.weight"):
            pre = k[: -len("reparam_conv.weight")]
            c = v.shape[0]
            ls = torch.rand(c, generator=g) * 0.2 + 0.05
            # target: I + ls * (mixer - norm) == v  with norm = BN-identity (scale s_n, shift t_n), mixer = one 3x3 conv + BN
            _bn(c, g, pre + "norm.rbr_skip.", out)
            s_n = out[pre + "norm.rbr_skip.weight"] / torch.sqrt(out[pre + "norm.rbr_skip.running_var"] + 1e-5)
            t_n = out[pre + "norm.rbr_skip.bias"] - out[pre + "norm.rbr_skip.running_mean"] * s_n
            ident = torch.zeros_like(v)
            ident[:, 0, 1, 1] = 1.0
            mix_w = (v - ident) / ls.view(-1, 1, 1, 1) + ident * s_n.view(-1, 1, 1, 1)
            mix_b = w[pre + "reparam_conv.bias"] / ls + t_n
            _bn(c, g, pre + "mixer.rbr_conv.0.bn.", out)
            s = out[pre + "mixer.rbr_conv.0.bn.weight"] / torch.sqrt(out[pre + "mixer.rbr_conv.0.bn.running_var"] + 1e-5)
            out[pre + "mixer.rbr_conv.0.conv.weight"] = mix_w / s.view(-1, 1, 1, 1)
            out[pre + "mixer.rbr_conv.0.bn.bias"] = mix_b + out[pre + "mixer.rbr_conv.0.bn.running_mean"] * s
            out[pre + "layer_scale"] = ls.view(-1, 1, 1)
        elif k.endswith("token_mixer.reparam_conv.bias"):
            continue
        elif re_match(k, r"network\.\d+\.reparam_conv\.weight$"):          # RepCPE
            pre = k[: -len("reparam_conv.weight")]
            ident = torch.zeros_like(v)
            ident[:, 0, 3, 3] = 1.0
            out[pre + "pe.weight"], out[pre + "pe.bias"] = v - ident, w[pre + "reparam_conv.bias"]
        elif re_match(k, r"network\.\d+\.reparam_conv\.bias$"):
            continue
        elif k.endswith("lkb_reparam.weight"):
            pre = k[: -len("lkb_reparam.weight")]
            c = v.shape[0]
            small = torch.randn(c, 1, 3, 3, generator=g) * 0.1
            _bn(c, g, pre + "small_conv.bn.", out)
            out[pre + "small_conv.conv.weight"] = small
            ss = out[pre + "small_conv.bn.weight"] / torch.sqrt(out[pre + "small_conv.bn.running_var"] + 1e-5)
            sb = out[pre + "small_conv.bn.bias"] - out[pre + "small_conv.bn.running_mean"] * ss
            big = v - F.pad(small * ss.view(-1, 1, 1, 1), [2, 2, 2, 2])
            _bn(c, g, pre + "lkb_origin.bn.", out)
            s = out[pre + "lkb_origin.bn.weight"] / torch.sqrt(out[pre + "lkb_origin.bn.running_var"] + 1e-5)
            out[pre + "lkb_origin.conv.weight"] = big / s.view(-1, 1, 1, 1)
            out[pre + "lkb_origin.bn.bias"] = (w[pre + "lkb_reparam.bias"] - sb) + out[pre + "lkb_origin.bn.running_mean"] * s
        elif k.endswith("lkb_reparam.bias"):
            continue
        elif k.endswith("reparam_conv.weight"):                            # plain MobileOne block: one conv + BN branch
            pre = k[: -len("reparam_conv.weight")]
            c = v.shape[0]
            _bn(c, g, pre + "rbr_conv.0.bn.", out)
            s = out[pre + "rbr_conv.0.bn.weight"] / torch.sqrt(out[pre + "rbr_conv.0.bn.running_var"] + 1e-5)
            out[pre + "rbr_conv.0.conv.weight"] = v / s.view(-1, 1, 1, 1)
            out[pre + "rbr_conv.0.bn.bias"] = w[pre + "reparam_conv.bias"] + out[pre + "rbr_conv.0.bn.running_mean"] * s
        elif k.endswith("reparam_conv.bias"):
            continue
        else:
            out[k] = v
    return out


def re_match(k, pat):
    import re
    return re.search(pat, k) is not None


def test_whole_tower_round_trip():
    """a tiny inference-form tower -> training form -> fold_train_form: the same keys and (to fp32 rounding) the same tensors, and
    the same image embeddings through the tower oracle"""
    m = arch.preset("tiny")
    g = torch.Generator().manual_seed(5)
    w = weights.init_tower(m.tower, m.llm.hidden, g)
    train = _unfold(w, g)
    assert reparam.is_train_form(train) and not reparam.is_train_form(w)
    back = reparam.fold_train_form(train)
    assert set(back) == set(w)
    for k in w:
        _close(back[k].reshape(w[k].shape), w[k], tol=5e-5)
    x = torch.rand(1, 3, 256, 256, generator=g)
    tc = fastvit_hd.TowerCfg(layers=m.tower.layers, dims=m.tower.dims)
    with torch.no_grad():
        a = fastvit_hd.tower_forward(w, x, tc)
        b = fastvit_hd.tower_forward({k: v.reshape(w[k].shape) for k, v in back.items()}, x, tc)
    _close(b, a, tol=2e-4)


def test_unfolded_whole_tower_oracle_equals_the_folded_one():
    """oracle/reparam.py tower_forward_train_form (every branch and BatchNorm evaluated as mci.py's training form does) against
    oracle/fastvit_hd.py tower_forward on the folded weights: the two statements of the tower agree to fp32 rounding, so the -m gpu
    checkpoint-interop test may hold the engine against the unfolded one."""
    m = arch.preset("small")
    g = torch.Generator().manual_seed(9)
    w = weights.init_tower(m.tower, m.llm.hidden, g)
    train = _unfold(w, g)
    x = torch.rand(2, 3, 128, 192, generator=g)
    tc = fastvit_hd.TowerCfg(layers=m.tower.layers, dims=m.tower.dims)
    with torch.no_grad():
        a = fastvit_hd.tower_forward(w, x, tc)
        b = oref.tower_forward_train_form(train, x, tc)
    _close(b, a, tol=2e-5)


def test_checkpoint_directory_in_training_form_is_folded_on_load(tmp_path):
    from safetensors.torch import save_file
    from vla_fastvlm.model.fastvlm_adapter import load_hf_checkpoint_dir
    m = arch.preset("tiny")
    g = torch.Generator().manual_seed(6)
    w = weights.init_tower(m.tower, m.llm.hidden, g)
    train = _unfold(w, g)
    train["lm_head.weight"] = torch.zeros(2, 2)
    save_file({k: v.contiguous() for k, v in train.items()}, str(tmp_path / "model.safetensors"))
    got = load_hf_checkpoint_dir(tmp_path)
    assert set(got) == set(w)
    for k in w:
        _close(got[k].reshape(w[k].shape), w[k], tol=5e-5)
