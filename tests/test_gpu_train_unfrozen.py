"""-m gpu: the UNFROZEN decoder + projector training slice (fv_train_* in include/fastvla_hip.h).

The reference's own loop would run this if model/fastvlm_adapter.py:501 were not an unconditional no_grad (`freeze_backbone`,
fastvla/configuration_fastvla.py:23; step body training/trainer.py:171-182).  Oracle = torch.autograd over the fp32 forward that oracle/
already states (oracle/train_unfrozen.py), on the same seeded weights, the same frozen-tower embeddings and the same inputs:
  * op level: RMSNorm backward and the RoPE + causal-GQA attention backward (head_dim 64 / 128, ragged lengths) against autograd;
  * the `small` preset (3 layers, spliced 36 + 16 tokens): loss, actions and EVERY tensor's gradient (<= 2e-3 rel-L2 each), the gradient
    buckets' order and coverage, bit-identical repeats, one clip + AdamW step over all 12 + 4 + 1 + 7 L + 1 tensors, and the committed
    bf16 operand copies afterwards;
  * four FULL-WIDTH FastVLM-0.5B decoder layers (hidden 896, 14 q / 2 kv heads of 64, inter 4864) the same way.
"""
import ctypes as C
import json
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import DEV, call, lib, rel_l2, stream  # noqa: E402
from fastvla_hip import FastVLAEngine, arch, weights  # noqa: E402
from oracle import head, qwen2, train_unfrozen  # noqa: E402

GRAD_TOL = 2e-3


def test_rmsnorm_backward_matches_autograd():
    torch.manual_seed(1)
    for rows, H in ((37, 256), (300, 896), (5, 3584)):
        x = torch.randn(rows, H, requires_grad=True)
        w = (1 + 0.1 * torch.randn(H)).requires_grad_(True)
        dy, dres = torch.randn(rows, H), torch.randn(rows, H)
        y = qwen2.rmsnorm(x, w, 1e-6)
        y.backward(dy)
        waves = (rows + 15) // 16
        scratch = torch.empty(((waves + 3) // 4 * 4 + 64) * H, device=DEV)
        dx, dw = torch.full((rows, H), float("nan"), device=DEV), torch.full((H,), float("nan"), device=DEV)
        xd, wd, dyd, dresd = x.detach().to(DEV), w.detach().to(DEV), dy.to(DEV), dres.to(DEV)
        call(lib().fv_op_rmsnorm_bwd(xd.data_ptr(), wd.data_ptr(), dyd.data_ptr(), dresd.data_ptr(), dx.data_ptr(), dw.data_ptr(), scratch.data_ptr(), rows, H, 1e-6, stream()),
             "fv_op_rmsnorm_bwd")
        torch.cuda.synchronize()
        assert rel_l2(dx.cpu(), x.grad + dres) <= 2e-6 and rel_l2(dw.cpu(), w.grad) <= 2e-6, (rows, H)


def _attention_autograd(qkv, B, T, heads, kv, D, lens, theta):
    cfg = qwen2.Qwen2Cfg(hidden=heads * D, layers=1, heads=heads, kv_heads=kv, head_dim=D, rope_theta=theta)
    qd, kd = heads * D, kv * D
    x = qkv.view(B, T, qd + 2 * kd)
    q = x[..., :qd].reshape(B, T, heads, D).transpose(1, 2)
    k = x[..., qd:qd + kd].reshape(B, T, kv, D).transpose(1, 2)
    v = x[..., qd + kd:].reshape(B, T, kv, D).transpose(1, 2)
    pos = torch.arange(T)
    cos, sin = qwen2.rope_tables(cfg, pos)
    q = q * cos + qwen2._rotate_half(q) * sin
    k = k * cos + qwen2._rotate_half(k) * sin
    k, v = k.repeat_interleave(heads // kv, dim=1), v.repeat_interleave(heads // kv, dim=1)
    mask = (pos[None, :] <= pos[:, None])[None, None] & (pos[None, :] < lens[:, None])[:, None, None, :]
    s = (q @ k.transpose(-1, -2)) * D ** -0.5
    s = s.masked_fill(~mask, torch.finfo(torch.float32).min)
    return (torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(B * T, qd)


@pytest.mark.parametrize("B,T,heads,kv,D", [(2, 52, 4, 2, 64), (3, 96, 14, 2, 64), (2, 70, 6, 2, 128), (1, 320, 14, 2, 64),
                                            # no grouping (one block per head) and a group wider than a block holds (9 q heads per kv head): the per-head kernels
                                            (2, 200, 2, 2, 64), (1, 130, 9, 1, 64)])
def test_attention_backward_matches_autograd(B, T, heads, kv, D):
    torch.manual_seed(B * 100 + T)
    qd, kd = heads * D, kv * D
    ld = qd + 2 * kd
    qkv = (torch.randn(B * T, ld) * 0.8).requires_grad_(True)
    lens = torch.tensor([T, max(1, T - 17), max(1, T // 3)][:B])
    dO = torch.randn(B * T, qd)
    for b in range(B):
        dO.view(B, T, qd)[b, int(lens[b]):] = 0   # rows past a prompt's end receive no gradient in the decoder (their outputs feed nothing)
    out = _attention_autograd(qkv, B, T, heads, kv, D, lens, 1e6)
    out.backward(dO)
    ref = qkv.grad.view(B, T, ld).clone()
    for b in range(B):
        ref[b, int(lens[b]):] = 0
    qd_, dod = qkv.detach().to(DEV), dO.to(DEV)
    dq = torch.full((B * T, ld), float("nan"), device=DEV)
    osc = torch.empty(B * T, 2 * qd, dtype=torch.bfloat16, device=DEV)
    st = torch.empty(2 * B * heads * T, device=DEV)
    ld_ = lens.to(torch.int32).to(DEV)
    call(lib().fv_op_attention_bwd(qd_.data_ptr(), ld, dod.data_ptr(), dq.data_ptr(), osc.data_ptr(), st.data_ptr(), B, T, heads, kv, D, ld_.data_ptr(), 1e6, stream()),
         "fv_op_attention_bwd")
    torch.cuda.synchronize()
    got = dq.cpu().view(B, T, ld)
    # the forward the backward started from
    o = (osc[:, :qd].float() + osc[:, qd:].float()).cpu().view(B, T, qd)
    for b in range(B):
        n = int(lens[b])
        assert rel_l2(o[b, :n], out.detach().view(B, T, qd)[b, :n]) <= 1e-5
        assert torch.isfinite(got[b, :n]).all()
        for name, c0, c1 in (("dq", 0, qd), ("dk", qd, qd + kd), ("dv", qd + kd, ld)):
            r = rel_l2(got[b, :n, c0:c1], ref[b, :n, c0:c1])
            assert r <= 2e-5, (name, b, r)
        assert float(got[b, n:, qd:].abs().max() if n < T else 0.0) == 0.0    # masked keys: exactly zero dk / dv
    dq2 = torch.empty_like(dq)
    call(lib().fv_op_attention_bwd(qd_.data_ptr(), ld, dod.data_ptr(), dq2.data_ptr(), osc.data_ptr(), st.data_ptr(), B, T, heads, kv, D, ld_.data_ptr(), 1e6, stream()),
         "fv_op_attention_bwd again")
    torch.cuda.synchronize()
    assert torch.equal(dq2.cpu().view(B, T, ld)[0, : int(lens[0])], got[0, : int(lens[0])])   # fixed summation order


def _rig(model, seed, head_dims, B, T):
    w = weights.init_backbone(model, seed=seed)
    eng = FastVLAEngine(model, state_dim=14, action_dim=14, hidden_dim=head_dims, fusion_dim=head_dims, max_batch=B, max_text_tokens=T, llm_precision=1)
    eng.load_weights(w)
    eng.train_begin()
    tensors, total, nb = eng.train_layout()
    flat = torch.zeros(total, dtype=torch.float32, device=DEV)
    eng.train_export_params(flat)
    lc = qwen2.Qwen2Cfg(hidden=model.llm.hidden, layers=model.llm.layers, heads=model.llm.heads, kv_heads=model.llm.kv_heads, head_dim=model.llm.head_dim,
                        inter=model.llm.inter, vocab=model.llm.vocab, rope_theta=model.llm.rope_theta, rms_eps=model.llm.rms_eps)
    shapes = head.head_shapes(lc.hidden, 14, 14, head_dims, head_dims)
    g = torch.Generator().manual_seed(seed + 1)
    hp = {k: (torch.randn(*s, generator=g) / (s[-1] ** 0.5 if len(s) > 1 else 10.0)) + (1.0 if k in ("state_projection.0.weight", "fusion.1.weight") else 0.0)
          for k, s in shapes.items()}
    for k, v in eng.head_views(flat).items():
        v.copy_(hp[k])
    return w, eng, tensors, total, nb, flat, lc, hp


def _inputs(model, B, T, seed):
    g = torch.Generator().manual_seed(seed)
    tower_out = (torch.randn(B, model.tower.num_tokens, model.tower.out_dim, generator=g) * 0.7).to(torch.bfloat16)
    ids = torch.randint(0, model.llm.vocab, (B, T), generator=g)
    ids[0, 3] = ids[0, 1]      # a repeated id inside a row and across rows: the embedding gradient must sum them
    ids[1, 0] = ids[0, 1]
    mask = torch.ones(B, T, dtype=torch.long)
    mask[1, T // 2 + 1:] = 0
    if B > 2:
        mask[2, 1:] = 0
    states, targets = torch.randn(B, 14, generator=g), torch.randn(B, 14, generator=g)
    return tower_out, ids, mask, states, targets


def _check_grads(eng, grads_flat, ref, tol=GRAD_TOL):
    got = eng.train_named_tensors(grads_flat / eng.train_loss_scale())    # every gradient carries the loss scale (a power of two: the division is exact)
    assert set(got) == set(ref["grads"]), sorted(set(got) ^ set(ref["grads"]))[:8]
    worst = ("", 0.0)
    for k, g in got.items():
        r = ref["grads"][k]
        assert g.shape == r.shape, (k, g.shape, r.shape)
        denom = float(r.norm())
        if denom < 1e-12:
            assert float(g.abs().max()) == 0.0, k
            continue
        e = rel_l2(g.cpu(), r)
        if e > worst[1]:
            worst = (k, e)
        assert e <= tol, f"gradient of {k}: rel_l2 {e:.3e} > {tol}"
    return worst


@pytest.mark.parametrize("name,llm,tower,B,T,hd", [
    ("small", None, "small", 3, 16, 64),
    ("0.5b-width-4-layers", arch.LLMConfig(hidden=896, layers=4, heads=14, kv_heads=2, head_dim=64, inter=4864, vocab=8192), "tiny", 4, 32, 128),
    # the WHOLE FastVLM-0.5B decoder (24 layers, 151936-row embedding) behind the real tower's 256 image tokens: the gradient error through the full depth
    ("0.5b-full-depth", arch.preset("fastvlm-0.5b").llm, "full", 2, 8, 256),
    # FastVLM-7B's decoder geometry at full width (C5's model): head_dim 128 through the attention backward, GQA group 7, K = 18944 contractions
    ("7b-width-2-layers", arch.LLMConfig(hidden=3584, layers=2, heads=28, kv_heads=4, head_dim=128, inter=18944, vocab=4096), "tiny", 2, 16, 128),
])
def test_unfrozen_step_matches_autograd(name, llm, tower, B, T, hd):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device")
    model = arch.preset("small") if llm is None else arch.ModelConfig(name, llm, arch.preset("fastvlm-0.5b" if tower == "full" else tower).tower)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    w, eng, tensors, total, nb, flat, lc, hp = _rig(model, 41, hd, B, T)
    tower_out, ids, mask, states, targets = _inputs(model, B, T, 42)
    # the exported master equals the weights the engine packed (bf16 matrices widen exactly)
    named = eng.train_named_tensors(flat)
    for k in train_unfrozen.trainable_backbone_keys(w):
        assert torch.equal(named[k].cpu(), w[k].reshape(named[k].shape).float()), k
    ws = eng.train_workspace(B, T)
    order = []
    act, loss, grads = eng.train_forward_backward(flat, tower_out.to(DEV), ids, mask.sum(1), states, targets, ws, training=False,
                                                  flat_grads=torch.zeros_like(flat), bucket_cb=lambda b, off, n: order.append((b, off, n)))
    torch.cuda.synchronize()
    ref = train_unfrozen.forward_backward(w, hp, tower_out.float(), ids, mask, states, targets, lc)
    ra, rl = rel_l2(act.cpu(), ref["pred"]), abs(float(loss) - float(ref["loss"])) / float(ref["loss"])
    worst = _check_grads(eng, grads, ref, tol=1e-3)    # the bar is 2e-3; the default arithmetic holds it with >= 2x margin at every size tested
    print(f"[unfrozen {name}] B={B} T={T} actions rel_l2={ra:.2e} loss rel={rl:.2e} worst gradient: {worst[0]} {worst[1]:.2e} ({len(ref['grads'])} tensors)")
    if name in ("0.5b-full-depth", "7b-width-2-layers"):   # the split-bf16 dgrad (two passes) through the full depth / at the 7B width, for the record
        eng.train_set_options(grad_split=1)
        _, _, g1s = eng.train_forward_backward(flat, tower_out.to(DEV), ids, mask.sum(1), states, targets, ws, training=False, flat_grads=torch.zeros_like(flat))
        torch.cuda.synchronize()
        w1s = _check_grads(eng, g1s, ref, tol=GRAD_TOL)
        print(f"[unfrozen {name}] split-bf16 dgrad operands: worst gradient {w1s[0]} {w1s[1]:.2e}")
        eng.train_set_options(grad_split=1, wgrad_f16=False)      # ... and the other shipped option at full width: two-pass split-bf16 weight gradients
        _, _, g2p = eng.train_forward_backward(flat, tower_out.to(DEV), ids, mask.sum(1), states, targets, ws, training=False, flat_grads=torch.zeros_like(flat))
        torch.cuda.synchronize()
        w2p = _check_grads(eng, g2p, ref, tol=2.5e-3)              # (the activation operand's 8 bits: 1.8e-3 on `small`; why it is not the default)
        print(f"[unfrozen {name}] + two-pass split-bf16 wgrad: worst gradient {w2p[0]} {w2p[1]:.2e}")
        eng.train_set_options()
    assert ra <= 1e-3 and rl <= 1e-3
    # the one-pass fp16 TRAINING forward (fv_train_set_forward_f16: half the forward's MFMA work): loss, actions and every gradient against the same fp32 oracle.
    # MEASURED OUTSIDE the 1e-3 bar at full depth (actions 1.0e-3, 231 of 306 gradients between 1e-3 and 2.1e-3): an opt-in speed knob, never the default;
    # what is asserted is its own envelope (actions 2e-3, gradients 2.5e-3)
    eng.train_set_forward_f16(True)
    act16, loss16, g16 = eng.train_forward_backward(flat, tower_out.to(DEV), ids, mask.sum(1), states, targets, ws, training=False, flat_grads=torch.zeros_like(flat))
    torch.cuda.synchronize()
    ra16, rl16 = rel_l2(act16.cpu(), ref["pred"]), abs(float(loss16) - float(ref["loss"])) / float(ref["loss"])
    got16 = eng.train_named_tensors(g16 / eng.train_loss_scale())
    e16 = sorted(((rel_l2(v.cpu(), ref["grads"][k]), k) for k, v in got16.items() if float(ref["grads"][k].norm()) > 1e-12), reverse=True)
    print(f"[unfrozen {name}] fp16 training forward: actions rel_l2={ra16:.2e} loss rel={rl16:.2e}; worst gradients: " + "; ".join(f"{k} {e:.2e}" for e, k in e16[:5])
          + f"; tensors above 1e-3: {sum(1 for e, _ in e16 if e > 1e-3)} of {len(e16)} ({eng.fp16_saturations()} saturations)")
    assert ra16 <= 2e-3 and rl16 <= 1e-3
    for e, k in e16:
        assert e <= 2.5e-3, f"fp16 training forward: gradient of {k}: rel_l2 {e:.3e}"
    eng.train_set_forward_f16(False)
    act1b, loss1b, g1b = eng.train_forward_backward(flat, tower_out.to(DEV), ids, mask.sum(1), states, targets, ws, training=False, flat_grads=torch.zeros_like(flat))
    torch.cuda.synchronize()
    assert torch.equal(g1b, grads) and torch.equal(act1b, act)      # switching back restores the split-bf16 forward bit for bit
    # buckets: head first, then final norm, layers last to first, embedding, projector; together they tile the flat buffer exactly once
    L = model.llm.layers
    assert [b for b, _, _ in order] == [0, 3 + L] + [3 + l for l in range(L - 1, -1, -1)] + [2, 1]
    spans = sorted((off, off + n) for _, off, n in order)
    assert spans[0][0] == 0 and spans[-1][1] == total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    # bit-identical repeat (no float atomics anywhere in the backward)
    act2, loss2, grads2 = eng.train_forward_backward(flat, tower_out.to(DEV), ids, mask.sum(1), states, targets, ws, training=False, flat_grads=torch.zeros_like(flat))
    torch.cuda.synchronize()
    assert torch.equal(grads, grads2) and torch.equal(act, act2) and torch.equal(loss, loss2)
    # one clip + AdamW step over ALL tensors (reference training/trainer.py:60-66,178-180), then the operand copies follow the master
    m, v, norm = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros(1, device=DEV)
    new = flat.clone()
    eng.adamw_step(new, grads, m, v, 1, lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0, grad_norm_out=norm, grad_scale=1.0 / eng.train_loss_scale())
    torch.cuda.synchronize()
    params = {k: w[k].float() for k in train_unfrozen.trainable_backbone_keys(w)}
    params.update({"head." + k: t for k, t in hp.items()})
    ref_new, ref_norm = train_unfrozen.adamw_clip_step(params, ref["grads"], lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0)
    assert abs(float(norm) - float(ref_norm)) <= 2e-3 * float(ref_norm)
    got_new = eng.train_named_tensors(new)
    coef = min(1.0, 1.0 / (float(ref_norm) + 1e-6))
    for k, r in ref_new.items():
        du, dr = got_new[k].cpu() - params[k].reshape(got_new[k].shape), r - params[k]
        dr = dr.reshape(du.shape)
        # Adam's first step is lr * g / (|g| + eps): ~lr * sign(g) wherever |g| >> eps = 1e-8, whatever the 1e-3-class noise on g; entries whose
        # clipped gradient is within 100 eps of zero (a key bias under a nearly position-independent score, unused embedding rows) move by
        # lr * noise / (|noise| + eps) and are compared by |update| <= lr only
        big = (ref["grads"][k].reshape(du.shape) * coef).abs() > 1e-6
        assert float(du.abs().max()) <= 1.0001e-3 + 1e-2 * 1e-3 * float(params[k].abs().max()), k
        if big.any():
            bad = float(((du - dr).abs()[big] > 0.05 * 1e-3 + 1e-2 * dr.abs()[big]).float().mean())
            assert bad <= 5e-3, (k, bad)
    eng.train_commit(new)
    back = torch.zeros_like(flat)
    eng.train_export_params(back)
    torch.cuda.synchronize()
    for t in tensors:
        if t["bucket"] == 0:
            continue
        a, b = back[t["offset"]: t["offset"] + t["numel"]], new[t["offset"]: t["offset"] + t["numel"]]
        if t["rows"] > 1:
            assert torch.equal(a, b.to(torch.bfloat16).float()), t["name"]      # matrices: the master rounded to bf16 (RNE)
        else:
            assert torch.equal(a, b), t["name"]                                  # norms / biases stay fp32
    # ... and the frozen-path entry points now run on the updated weights: spliced prefill vs the oracle on the bf16-rounded master
    w2 = dict(w)
    for k, tnew in got_new.items():
        if k.startswith("head."):
            continue
        w2[k] = (tnew.cpu().to(torch.bfloat16).float() if tnew.ndim == 2 else tnew.cpu()).reshape(w[k].shape)
    from oracle import fastvit_hd
    with torch.no_grad():
        tok_ref = fastvit_hd.projector_forward(w2, tower_out.float())
        pooled_ref = qwen2.llm_pooled(w2, ids, mask, lc, image_tokens=tok_ref, splice=True)
    pooled = eng.llm_pooled(ids, mask.sum(1), tok_ref.to(DEV))
    torch.cuda.synchronize()
    assert rel_l2(pooled.cpu(), pooled_ref) <= 3e-4
    if name == "small":
        # the one-launch commit also rewrote the dgrad's transposed fp16 weight copies: a backward on them must equal, bit for bit, a backward on
        # copies rebuilt from the library's weights by the standalone transpose kernels (what a switch of the dgrad form and back does)
        _, _, g_commit = eng.train_forward_backward(new, tower_out.to(DEV), ids, mask.sum(1), states, targets, ws, training=False, flat_grads=torch.zeros_like(flat))
        eng.train_set_options(grad_split=1)
        eng.train_set_options()
        _, _, g_rebuilt = eng.train_forward_backward(new, tower_out.to(DEV), ids, mask.sum(1), states, targets, ws, training=False, flat_grads=torch.zeros_like(flat))
        torch.cuda.synchronize()
        assert torch.equal(g_commit, g_rebuilt)
        assert float((g_commit - grads).abs().max()) > 0      # (and the parameters did move)
    eng.close()


def test_backward_arithmetic_options():
    """fv_train_set_options: the default (dgrad and wgrad each ONE fp16 pass, loss scale 2^12) against the two-pass
    split-bf16 wgrad it replaced and against the plain-bf16 dgrad speed knob.  Same loss and actions in every mode (the forward is untouched);
    the worst per-tensor gradient error of each mode is printed: the default must hold the 2e-3 bar WITH margin (<= 1e-3), the legacy wgrad holds
    it barely (the activation operand's 8 bits), plain-bf16 dgrad operands do not (bounded at 8e-3: an option, never the default).  Another loss
    scale gives the same gradients up to the fp16 rounding of the scaled operand, and no cast saturates."""
    model = arch.preset("small")
    B, T = 3, 16
    w, eng, tensors, total, nb, flat, lc, hp = _rig(model, 41, 64, B, T)
    tower_out, ids, mask, states, targets = _inputs(model, B, T, 42)
    ws = eng.train_workspace(B, T)
    ref = train_unfrozen.forward_backward(w, hp, tower_out.float(), ids, mask, states, targets, lc)

    def run():
        a, l, g = eng.train_forward_backward(flat, tower_out.to(DEV), ids, mask.sum(1), states, targets, ws, training=False, flat_grads=torch.zeros_like(flat))
        torch.cuda.synchronize()
        return a, l, g

    assert eng.train_loss_scale() == 4096.0
    a1, l1, g1 = run()                                     # default: dgrad and wgrad each ONE fp16 pass
    worst_default = _check_grads(eng, g1, ref, tol=1e-3)
    from fastvla_hip import FastVLAHipError
    for gone in (dict(grad_split=0), dict(wgrad_f16=2)):   # round 5: the two options outside the bar / slower than the default are no longer in the product
        with pytest.raises(FastVLAHipError):
            eng.train_set_options(**gone)
    eng.train_set_options(grad_split=1, wgrad_f16=True)    # split-bf16 dgrad operands (two passes): the most exact form
    a6, l6, g6 = run()
    worst_split = _check_grads(eng, g6, ref, tol=1e-3)
    assert torch.equal(a6, a1) and torch.equal(l6, l1)
    eng.train_set_options(grad_split=1, wgrad_f16=False)   # round 4's first form: split-bf16 gradient x bf16 activation for the weight gradients
    a2, l2, g2 = run()
    worst_legacy = _check_grads(eng, g2, ref, tol=GRAD_TOL)
    eng.train_set_options(loss_scale_log2=8)
    a4, l4, g4 = run()
    assert eng.train_loss_scale() == 256.0
    worst_ls8 = _check_grads(eng, g4, ref, tol=1e-3)
    print(f"[unfrozen small, backward arithmetic] worst gradient -- default (fp16 dgrad + fp16 wgrad, one pass each): {worst_default[0]} {worst_default[1]:.2e}; "
          f"split-bf16 dgrad: {worst_split[1]:.2e}; + two-pass split-bf16 wgrad: {worst_legacy[1]:.2e}; "
          f"loss scale 2^8: {worst_ls8[1]:.2e}")
    for a, l in ((a2, l2), (a4, l4)):
        assert torch.equal(a, a1) and torch.equal(l, l1)
    assert eng.fp16_saturations() == 0
    eng.train_set_options()
    _, _, g5 = run()
    assert torch.equal(g5, g1)
    eng.close()


def test_unfrozen_training_needs_split_bf16_weights():
    from fastvla_hip import FastVLAHipError
    m = arch.preset("small")
    eng = FastVLAEngine(m, hidden_dim=32, fusion_dim=32, max_batch=2, max_text_tokens=8, llm_precision=2)
    eng.load_weights(weights.init_backbone(m, seed=3))
    with pytest.raises(FastVLAHipError, match="llm_precision = 1"):
        eng.train_begin()
    eng.close()


def test_policy_level_unfrozen_training_overfits_one_batch_and_exports_the_trained_backbone(tmp_path):
    """The surface a trainer drives: FastVLAPolicy.enable_backbone_training() + fused_train_step (reference step body
    training/trainer.py:171-182 over ALL parameters).  `freeze_backbone=False` alone keeps the reference's behaviour (head only);
    after the explicit opt-in a fixed batch is overfitted, the loss of step 1 equals the oracle's, select_action runs the UPDATED spliced
    model, gradient accumulation over two half batches gives the full batch's update, and the checkpoint written with the reference's key
    names carries the trained decoder (model.backbone.model.*) and loads back to the same actions."""
    from vla_fastvlm.fastvla import FastVLAConfig, FastVLAPolicy
    from vla_fastvlm.utils import load_policy_from_checkpoint, save_policy_checkpoint
    torch.manual_seed(5)
    cfg = FastVLAConfig(vlm_model_name="synthetic:small:41", hidden_dim=64, fusion_dim=64, dropout=0.0, freeze_backbone=False)
    pol = FastVLAPolicy(cfg).to(DEV)
    pol.train()
    g = torch.Generator().manual_seed(6)
    B = 4
    batch = {"images": torch.rand(B, 3, 96, 128, generator=g).to(DEV), "states": torch.randn(B, 14, generator=g).to(DEV),
             "actions": torch.randn(B, 14, generator=g).to(DEV), "tasks": ["pick up the red cube", "open the drawer", "push", "pick up the red cube"]}
    # the config flag alone: a head-only step, exactly as the reference would run it
    out0 = pol.fused_train_step(batch, lr=1e-3)
    assert pol._unfrozen is None and pol.model.backbone.splice_image_tokens is False
    pol2 = FastVLAPolicy(cfg).to(DEV)
    pol2.train()
    st = pol2.enable_backbone_training()
    assert pol2.model.backbone.splice_image_tokens is True
    before = {k: v.clone() for k, v in st.named_backbone_tensors().items()}
    losses = []
    for i in range(16):
        out = pol2.fused_train_step(batch, lr=2e-3, weight_decay=0.0)
        losses.append(float(out["loss"]))
    torch.cuda.synchronize()
    print("[unfrozen policy] loss over 16 steps on one batch:", " ".join(f"{x:.4f}" for x in losses))
    assert all(map(math.isfinite, losses)) and losses[-1] < 0.6 * losses[0] and losses[-1] < losses[7] < losses[0]
    after = st.named_backbone_tensors()
    moved = [k for k in before if not torch.equal(before[k], after[k])]
    assert len(moved) >= len(before) - 1       # everything but (at most) untouched tensors moved: embedding rows count as one tensor
    assert any("mm_projector" in k for k in moved) and any("embed_tokens" in k for k in moved)
    # inference now runs the updated, spliced model: actions move towards the targets
    pol2.eval()
    with torch.no_grad():
        a = pol2(batch["images"], batch["states"], batch["tasks"])
    mse = float(((a - batch["actions"]) ** 2).mean())
    assert mse < 0.7 * losses[0]
    # the checkpoint carries the TRAINED decoder under the reference's names and loads back to the same actions
    out_dir = save_policy_checkpoint(pol2, tmp_path / "step-16", include_backbone=True)
    sd = torch.load(out_dir / "policy_state_dict.pt", map_location="cpu")
    k0 = "model.backbone.model.model.layers.0.mlp.down_proj.weight"
    assert torch.equal(sd[k0], after["model.layers.0.mlp.down_proj.weight"].cpu()) and not torch.equal(sd[k0], before["model.layers.0.mlp.down_proj.weight"].cpu())
    again = load_policy_from_checkpoint(str(out_dir)).to(DEV)
    # the mode the decoder was trained in travels BESIDE the reference's two files (hip_extras.json), never inside them: policy_state_dict.pt keeps the reference's keys
    assert again.model.backbone.splice_image_tokens is True and "model.backbone.splice_image_tokens" not in sd
    assert json.loads((out_dir / "hip_extras.json").read_text())["splice_image_tokens"] is True
    with torch.no_grad():
        a2 = again(batch["images"], batch["states"], batch["tasks"])
    torch.cuda.synchronize()
    assert rel_l2(a2.cpu(), a.cpu()) <= 1e-5       # the file holds the fp32 master; both engines round it to the same bf16 operands
    # gradient accumulation: two half batches == the full batch (reference trainer.py:96,171: accelerate sums micro-batch gradients)
    pa, pb = FastVLAPolicy(cfg).to(DEV), FastVLAPolicy(cfg).to(DEV)
    for p_ in (pa, pb):
        p_.train()
        p_.load_state_dict(pol.state_dict(), strict=False)
    sa, sb = pa.enable_backbone_training(), pb.enable_backbone_training()
    sb.flat.copy_(sa.flat)
    pa.fused_train_step(batch, lr=1e-3)
    half = lambda lo, hi: {k: (v[lo:hi] if torch.is_tensor(v) else v[lo:hi]) for k, v in batch.items()}   # noqa: E731
    pb.fused_train_step(half(0, 2), lr=1e-3, grad_accum_steps=2)
    pb.fused_train_step(half(2, 4), lr=1e-3, grad_accum_steps=2)
    torch.cuda.synchronize()
    ga, gb = sa.g, sb.acc / 2
    assert rel_l2(gb.cpu(), ga.cpu()) <= 2e-3      # other tile shapes at other row counts: the operand roundings differ, the sums agree
    for p_ in (pol, pol2, again, pa, pb):
        p_.model.backbone.engine().close()


def test_two_rank_unfrozen_step_equals_the_full_batch_step(tmp_path):
    """Data-parallel unfrozen training on DEVICE tensors: two rank processes (gloo: this pool gives one GPU, so RCCL cannot form a
    communicator; under RCCL the same worker is one rank per GPU) each take half of one fixed batch, differentiate it, and let the gradient
    leave bucket by bucket -- fv_bucket_cb -> BucketedGradExchange: event on the compute stream, all-reduce of that slice on the side
    stream -- under the rest of the backward pass.  The reduced, averaged gradient must equal the single-process full-batch gradient (the
    MSE is a mean over samples: reference fastvla/modeling_fastvla.py:56; accelerate/DDP averaging: training/trainer.py:68-78,175), both
    ranks must end with bit-identical parameters, and the clipped update must match the full-batch one."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    worker = str(root / "tools" / "unfrozen_dp_worker.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    one = tmp_path / "w1"
    one.mkdir()
    r = subprocess.run([sys.executable, worker, "--out", str(one)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    two = tmp_path / "w2"
    two.mkdir()
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, worker, "--out", str(two)], env=dict(env, RANK=str(k), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                                                                                   MASTER_PORT=str(port)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for k in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-1500:] for o in outs]
    full = torch.load(one / "rank0.pt")
    r0, r1 = torch.load(two / "rank0.pt"), torch.load(two / "rank1.pt")
    assert r0["world"] == 2 and len(r0["collectives"]) >= 3          # head, final norm + layers (coalesced), embedding, projector ...
    spans = sorted(r0["collectives"])
    assert spans[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:])) and spans[-1][0] + spans[-1][1] == full["grads"].numel()
    assert torch.equal(r0["grads"], r1["grads"]) and torch.equal(r0["flat"], r1["flat"])      # replicas stay identical
    e = rel_l2(r0["grads"], full["grads"])
    print(f"[unfrozen dp2 vs full batch] reduced gradient rel_l2 {e:.2e}; grad norm {r0['grad_norm']:.4f} vs {full['grad_norm']:.4f}; "
          f"{len(r0['collectives'])} collectives per step")
    assert e <= 2e-3 and abs(r0["grad_norm"] - full["grad_norm"]) <= 2e-3 * full["grad_norm"]
    # the half-batch losses average to the full-batch loss
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - full["loss"]) <= 1e-4 * abs(full["loss"])


@pytest.mark.parametrize("tower", [False, True])
def test_trainer_fit_with_the_backbone_unfrozen_saves_and_resumes(tmp_path, tower):
    """the documented route -- `Trainer.fit()` (reference training/trainer.py:145-206) over a loader with several batches while the backbone is
    unfrozen.  The loop's one batch of look-ahead (`next_batch=`) must not break the unfrozen step; its checkpoints must carry the TRAINED VLM, the splice mode
    and the whole-master AdamW moments; a resumed run must continue bit for bit where the uninterrupted one is.  tower=True: FastViT-HD trainable as well."""
    from vla_fastvlm.fastvla import FastVLAConfig, FastVLAPolicy
    from vla_fastvlm.training import Trainer, TrainingConfig
    g = torch.Generator().manual_seed(8)

    def mk(B):
        return {"images": torch.rand(B, 3, 96, 128, generator=g), "states": torch.randn(B, 14, generator=g), "actions": torch.randn(B, 14, generator=g),
                "tasks": ["pick up the red cube", "open the drawer", "push"][:B]}

    data = [mk(3), mk(3), mk(3), mk(3)]
    cfg = FastVLAConfig(vlm_model_name="synthetic:small:43", hidden_dim=64, fusion_dim=64, dropout=0.0, freeze_backbone=False)
    tkw = dict(num_epochs=1, learning_rate=1e-3, warmup_ratio=0.5, logging_steps=1000, eval_steps=1000, seed=1)

    def fresh():
        torch.manual_seed(7)
        p = FastVLAPolicy(cfg).to(DEV)
        p.enable_backbone_training(tower=tower)
        return p

    a = fresh()
    Trainer(a, data, None, TrainingConfig(output_dir=str(tmp_path / "a"), save_steps=1000, max_steps=4, **tkw)).fit()
    b = fresh()
    tb = Trainer(b, data[:3], None, TrainingConfig(output_dir=str(tmp_path / "b"), save_steps=3, max_steps=4, **tkw))
    tb.num_training_steps = 4
    tb.fit()
    ck = tmp_path / "b" / "checkpoints" / "step-3"
    sd = torch.load(ck / "policy_state_dict.pt", map_location="cpu")
    opt = torch.load(ck / "optimizer.pt", map_location="cpu")
    assert any(k.startswith("model.backbone.model.model.layers.") for k in sd) and "model.backbone.splice_image_tokens" not in sd    # the trained VLM travels by default
    assert json.loads((ck / "hip_extras.json").read_text()) == {"splice_image_tokens": True, "train_backbone": True, "train_tower": bool(tower)}
    assert opt["train_tower"] is bool(tower) and opt["train_backbone"] is True
    assert opt["m"].numel() == b._unfrozen.flat.numel() and int(opt["step"]) == 3
    if tower:
        assert any(".vision_tower." in k and k.endswith("convffn.conv.bn.running_var") for k in sd)
    c = fresh()
    tc = Trainer(c, data[3:], None, TrainingConfig(output_dir=str(tmp_path / "c"), save_steps=1000, max_steps=4, resume_from=str(ck), **tkw))
    tc.num_training_steps = 4
    tc.fit()
    torch.cuda.synchronize()
    assert tc.global_step == 4 and c._unfrozen.step_count == 4
    fa, fc = a._unfrozen.flat, c._unfrozen.flat
    # optimizer.pt carries the fp32 master itself beside m / v (the VLM tensors of policy_state_dict.pt go through the engine's bf16 operand copies: a master rebuilt
    # from THEM would lose the residue that the next updates, 1e-3 of a bf16 ulp each, live in): the resumed run IS the uninterrupted one
    e = rel_l2(fc.cpu(), fa.cpu())
    print(f"[Trainer.fit unfrozen, tower={tower}] resumed vs uninterrupted master after 4 steps: rel_l2 {e:.2e}")
    assert "flat" in opt and torch.equal(fc, fa)
    for p_ in (a, b, c):
        p_.model.backbone.engine().close()
