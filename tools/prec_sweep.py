"""tools/prec_sweep.py [model] [B] -- (needs the TOOLS build: make AB=1, FASTVLA_HIP_LIB=tools/bin/libfastvla_hip_ab.so: policies 3 / 4 are not in the product library) -- decoder precision policies (fv_model_desc.llm_precision 1..4) on one seeded synthetic model: the
pooled feature of each policy against policy 1 (split-bf16 everywhere, 1e-5 of the fp32 oracle at 0.5B), its batch-invariance
(rows 3 and 5 alone against the same rows of the batch) and the time of the decoder call.  GPU box only; prints a table."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vla-from-fastvlm_amd"))
from fastvla_hip import FastVLAEngine, arch, weights  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "fastvlm-7b"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
T = 64
dev = torch.device("cuda", 0)
model = arch.preset(name)


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


ref = {}
for seed in (1234, 77):
    for prec in (1, 2, 3, 4):
        eng = FastVLAEngine(model, state_dim=14, action_dim=14, hidden_dim=1024, fusion_dim=1024, device=dev, max_batch=B, max_text_tokens=T,
                            llm_precision=prec)
        eng.load_weights_streaming(weights.stream_backbone(model, seed=seed, device=dev))
        g = torch.Generator().manual_seed(seed + 5)
        ids = torch.randint(0, 151643, (B, T), generator=g)
        lens = torch.full((B,), T)
        lens[3] = 40
        pooled = eng.llm_pooled(ids, lens).clone()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(5):
            eng.llm_pooled(ids, lens)
        torch.cuda.synchronize()
        ms = (time.time() - t0) / 5 * 1e3
        sub = torch.tensor([3, 5])
        alone = eng.llm_pooled(ids[sub], lens[sub]).clone()
        torch.cuda.synchronize()
        if prec == 1:
            ref[seed] = pooled.cpu()
        rows = [rel(pooled[i].cpu(), ref[seed][i]) for i in range(B)]
        print(f"{name} seed {seed} llm_precision {prec}: decoder {ms:7.2f} ms  vs policy 1: all rows {rel(pooled.cpu(), ref[seed]):.2e}  worst row {max(rows):.2e}  "
              f"rows alone vs in batch {rel(alone.cpu(), pooled[sub].cpu()):.2e}", flush=True)
        eng.close()
        del eng
        torch.cuda.empty_cache()
