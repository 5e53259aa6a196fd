"""Unfrozen decoder + projector training step alone, for rocprofv3 --kernel-trace --stats and A/B runs:
    python tools/train_unfrozen_bench.py [--batch 32] [--tokens 64] [--steps 6] [--model fastvlm-0.5b]
One step = letterbox + frozen tower + fv_train_forward_backward + fv_adamw_clip_step + fv_train_commit (what bench.py's train_unfrozen leg times)."""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "vla-from-fastvlm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
from fastvla_hip import FastVLAEngine, arch, weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="fastvlm-0.5b")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--tokens", type=int, default=64)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-tower", action="store_true", help="reuse one tower output (times the trainable part alone)")
    ap.add_argument("--train-tower", action="store_true", help="the FastViT-HD tower trainable too (fv_train_tower_*): forward with stash + backward of the tower inside the step")
    ap.add_argument("--tower-only", action="store_true", help="with --train-tower: time the tower's forward + backward alone (a fixed dL/d tower_out)")
    ap.add_argument("--dgrad-split", action="store_true", help="fv_train_set_options(grad_split=1): split-bf16 dgrad operands (two passes) instead of ONE fp16 pass")
    ap.add_argument("--wgrad-bf16", action="store_true", help="fv_train_set_options(wgrad_f16=0): weight gradients as split-bf16 gradient x bf16 activation (two passes)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    model = arch.preset(args.model)
    B, T = args.batch, args.tokens
    eng = FastVLAEngine(model, max_batch=B, max_text_tokens=T, llm_precision=1)
    if 3 * model.llm.hidden * model.llm.inter * model.llm.layers > 2e9:   # 7B: 30 GB as an fp32 host dict -- stream it, drawn on the device
        eng.load_weights_streaming(weights.stream_backbone(model, seed=1234, device=dev))
    else:
        eng.load_weights(weights.init_backbone(model, seed=1234))
    eng.train_begin()
    if args.train_tower:
        eng.train_tower_begin()
    eng.train_set_options(grad_split=1 if args.dgrad_split else 2, wgrad_f16=0 if args.wgrad_bf16 else 1)
    _, total, nb = eng.train_layout()
    flat = torch.zeros(total, device=dev)
    eng.train_export_params(flat)
    g = torch.Generator().manual_seed(1)
    hv = eng.head_views(flat[: eng.head_numel()])
    for k, v in hv.items():
        v.copy_(torch.randn(v.shape, generator=g) * 0.02 + (1.0 if k in ("state_projection.0.weight", "fusion.1.weight") else 0.0))
    grads, m, v = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(flat)
    ws = eng.train_workspace(B, T)
    images = torch.rand(B, 3, 336, 336, generator=g).to(dev)
    ids = torch.randint(0, 151643, (B, T), generator=g)
    lens = torch.full((B,), T)
    states, targets = torch.randn(B, 14, generator=g).to(dev), torch.randn(B, 14, generator=g).to(dev)
    st = {"n": 0, "tower_out": None}
    if args.train_tower:
        tws = eng.train_tower_workspace(B)
        dto = (torch.randn(B, model.tower.num_tokens, model.tower.out_dim, generator=g) * 1e-2).to(torch.float16).to(dev)
        if not args.tower_only:
            eng.train_set_tower_grad(dto)

    def step():
        st["n"] += 1
        if args.train_tower:
            pix = eng.preprocess(images)
            tower_out = eng.train_tower_forward(pix, tws)
            loss = torch.zeros(1)
            if not args.tower_only:
                _, loss, _ = eng.train_forward_backward(flat, tower_out, ids, lens, states, targets, ws, training=True, dropout_p=0.1, seed=7, offset=st["n"], flat_grads=grads)
            eng.train_tower_backward(pix, dto, tws, grads)
            if not args.tower_only:
                eng.adamw_step(flat, grads, m, v, st["n"], lr=1e-5, weight_decay=1e-4, max_grad_norm=1.0, grad_scale=1.0 / eng.train_loss_scale())
                eng.train_commit(flat)
            return loss
        if st["tower_out"] is None or not args.no_tower:
            _, st["tower_out"] = eng.vision_forward(eng.preprocess(images), return_tower_out=True)
        _, loss, _ = eng.train_forward_backward(flat, st["tower_out"], ids, lens, states, targets, ws, training=True, dropout_p=0.1, seed=7, offset=st["n"], flat_grads=grads)
        eng.adamw_step(flat, grads, m, v, st["n"], lr=1e-5, weight_decay=1e-4, max_grad_norm=1.0, grad_scale=1.0 / eng.train_loss_scale())
        eng.train_commit(flat)
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    print(json.dumps({"model": args.model, "batch": B, "tokens": model.tower.num_tokens + T, "ms_per_step": round(ms, 2), "loss": float(loss), "trainable_params": total,
                      "buckets": nb, "tower_in_step": not args.no_tower, "dgrad_operands": "split bf16 (hi + lo)" if args.dgrad_split else "fp16, one pass", "wgrad": "split-bf16 x bf16, two passes" if args.wgrad_bf16 else "one fp16 pass"}))


if __name__ == "__main__":
    main()
