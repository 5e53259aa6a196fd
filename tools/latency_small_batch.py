"""tools/latency_small_batch.py [model] -- the control-loop view of the path: ONE observation in, one action out, host-synchronised every
step (what `select_action` costs a robot loop).  Eager launches against the whole step replayed as one hipGraph
(FastVLAEngine.capture_policy_step), B = 1, 2, 4, literal and splice mode.  GPU box only."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vla-from-fastvlm_amd"))
from fastvla_hip import FastVLAEngine, arch, weights  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "fastvlm-0.5b"
dev = torch.device("cuda", 0)
model = arch.preset(name)
T = 64
eng = FastVLAEngine(model, state_dim=14, action_dim=14, hidden_dim=1024, fusion_dim=1024, device=dev, max_batch=4, max_text_tokens=T,
                    llm_precision=arch.default_llm_precision(model))
eng.load_weights_streaming(weights.stream_backbone(model, seed=1234, device=dev))
flat = torch.zeros(eng.head_numel(), dtype=torch.float32, device=dev).normal_(0, 0.02)
for k, v in eng.head_views(flat).items():
    if k in ("state_projection.0.weight", "fusion.1.weight"):
        v.fill_(1.0)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
        torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return 1e3 * ts[len(ts) // 2], 1e3 * ts[0]


for splice in (False, True):
    for B in (1, 2, 4):
        g = torch.Generator().manual_seed(B)
        img = torch.rand(B, 3, 336, 336, generator=g).to(dev)
        ids = torch.randint(0, 151643, (B, T), generator=g)
        lens = torch.full((B,), T)
        st = torch.randn(B, 14, generator=g).to(dev)

        def eager():
            pooled = eng.backbone(img, ids, lens, splice=splice)
            return eng.head_forward(flat, pooled, st)[0]

        a_eager = eager().clone()
        e_med, e_min = timeit(eager)
        replay, act = eng.capture_policy_step(img, ids, lens, flat, st, splice=splice)
        replay()
        torch.cuda.synchronize()
        same = bool(torch.equal(act, a_eager))
        g_med, g_min = timeit(replay)
        print(f"{name} {'splice' if splice else 'literal'} B={B}: eager {e_med:.2f} ms (min {e_min:.2f})  graph {g_med:.2f} ms (min {g_min:.2f})  "
              f"bit-identical {same}", flush=True)
