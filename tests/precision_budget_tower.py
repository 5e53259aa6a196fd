"""Precision budget of the SPLICE mode -- analysis script, CPU, uses the oracle (test infrastructure).

In splice mode the 256 projected image tokens enter the decoder, so the tower's bf16 activations (rounded once per tensor that reaches
HBM, 44 blocks deep) reach the actions: 1.7e-2 against the all-fp32 oracle at full size, where the literal (text-only) mode sits at 1e-5.
This script switches the bf16-faithful mode of oracle/fastvit_hd.py on PER GROUP of tower units -- stem, each stage's blocks, the RepCPEs, the
PatchEmbeds, conv_exp + SE, the projector's hidden -- on the full-size FastVLM-0.5B graph (1024^2, seeded weights), feeds the resulting image
tokens through the fp32 decoder + head, and reports the spliced actions' rel-L2 against the all-fp32 run:
  * each group ALONE rounded (what that group contributes),
  * everything rounded EXCEPT one group (what keeping that group in fp32 / split-bf16 would buy),
  * everything rounded from stage k on in fp32 (a cumulative cut).
    python tests/precision_budget_tower.py [B]
Reference call site: src/vla_fastvlm/model/fastvlm_adapter.py:519-536 (the VLM call with images; fp32 in the reference)."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "vla-from-fastvlm_amd")]
from fastvla_hip import arch, weights  # noqa: E402
from oracle import fastvit_hd, head, preprocess, qwen2  # noqa: E402


def group_of(unit):
    kind, i, _, _ = unit
    return "stem" if kind == "stem" else (f"stage{i}" if kind == "block" else kind)   # "cpe", "down"


def tokens(w, x, tc, rounded):
    q = fastvit_hd.strip_prefix(w, fastvit_hd.VT)
    if "pixels" in rounded:
        x = fastvit_hd._r(x)
    for unit in fastvit_hd.tower_units(tc):
        x = fastvit_hd.unit_forward(q, x, unit, tc, emulate_bf16=group_of(unit) in rounded)
    emb = fastvit_hd.tower_head_forward(q, x, tc, emulate_bf16="head" in rounded)
    return fastvit_hd.projector_forward(w, emb, emulate_bf16="projector" in rounded)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    torch.set_num_threads(8)
    m = arch.preset("fastvlm-0.5b")
    w = weights.init_backbone(m, seed=2024)
    tc = fastvit_hd.TowerCfg(layers=m.tower.layers, dims=m.tower.dims, mlp_ratio=m.tower.mlp_ratio, head_dim=m.tower.head_dim, attn_stages=m.tower.attn_stages)
    lc = qwen2.Qwen2Cfg(hidden=m.llm.hidden, layers=m.llm.layers, heads=m.llm.heads, kv_heads=m.llm.kv_heads, head_dim=m.llm.head_dim, inter=m.llm.inter,
                        vocab=m.llm.vocab, rope_theta=m.llm.rope_theta, rms_eps=m.llm.rms_eps)
    g = torch.Generator().manual_seed(21)
    img = torch.rand(B, 3, 336, 336, generator=g)
    ids = torch.randint(0, 151643, (B, 32), generator=g)
    mask = torch.ones(B, 32, dtype=torch.long)
    states = torch.randn(B, 14, generator=g)
    shapes = head.head_shapes(lc.hidden, 14, 14, 1024, 1024)
    hp = {k: (torch.randn(*s, generator=g) / (s[-1] ** 0.5 if len(s) > 1 else 10.0)) + (1.0 if k in ("state_projection.0.weight", "fusion.1.weight") else 0.0)
          for k, s in shapes.items()}
    x = preprocess.letterbox(img, m.tower.image_size)
    groups = ["pixels", "stem", "stage0", "stage1", "stage2", "down", "cpe", "stage3", "stage4", "head", "projector"]

    def actions(rounded):
        with torch.no_grad():
            tok = tokens(w, x, tc, set(rounded))
            pooled = qwen2.llm_pooled(w, ids, mask, lc, image_tokens=tok, splice=True)
            return head.head_forward(hp, pooled, states), tok

    t0 = time.time()
    ref, tok_ref = actions([])
    print(f"all-fp32 reference: {time.time() - t0:.1f} s per configuration (B = {B})")

    def err(rounded):
        a, tok = actions(rounded)
        return float((a - ref).norm() / ref.norm()), float((tok - tok_ref).norm() / tok_ref.norm())

    ea, et = err(groups)
    print(f"{'everything rounded (the product)':<44} actions {ea:.2e}   image tokens {et:.2e}")
    print("-- one group alone rounded:")
    for grp in groups:
        ea, et = err([grp])
        print(f"   only {grp:<37} actions {ea:.2e}   image tokens {et:.2e}")
    print("-- everything rounded EXCEPT one group:")
    for grp in groups:
        ea, et = err([x_ for x_ in groups if x_ != grp])
        print(f"   all but {grp:<34} actions {ea:.2e}   image tokens {et:.2e}")
    print("-- fp32 from a cut onwards (everything before it rounded):")
    order = ["pixels", "stem", "stage0", "down", "stage1", "stage2", "cpe", "stage3", "stage4", "head", "projector"]
    for cut in ("stage2", "stage3", "stage4", "head"):
        keep = order[: order.index(cut)]
        if cut in ("stage3", "stage4", "head"):
            keep = [k for k in keep]      # 'down' and 'cpe' are single groups: PatchEmbeds before the cut stay rounded with the early stages
        ea, et = err(keep)
        print(f"   rounded up to (not incl.) {cut:<16} actions {ea:.2e}   image tokens {et:.2e}")


if __name__ == "__main__":
    main()
