import sys, torch
sys.path.insert(0, 'vla-from-fastvlm_amd')
from fastvla_hip import FastVLAEngine, arch, weights
dev = torch.device('cuda', 0)
model = arch.preset('fastvlm-0.5b')
eng = FastVLAEngine(model, state_dim=14, action_dim=14, hidden_dim=1024, fusion_dim=1024, device=dev, max_batch=16, max_text_tokens=64, llm_precision=1)
eng.load_weights_streaming(weights.stream_backbone(model, seed=1234, device=dev))
g = torch.Generator().manual_seed(3)
img = torch.rand(16, 3, 336, 336, generator=g).to(dev)
worst = 0.0
for T in (8, 24, 40, 64):
    ids = torch.randint(0, 151643, (16, T), generator=g)
    lens = torch.randint(1, T + 1, (16,), generator=g)
    for splice in (False, True):
        big = eng.backbone(img, ids, lens, splice=splice).clone()
        for B in (1, 2, 3, 4, 5, 7):
            small = eng.backbone(img[:B].contiguous(), ids[:B], lens[:B], splice=splice)
            torch.cuda.synchronize()
            assert torch.isfinite(small).all(), (T, splice, B)
            r = float((small - big[:B]).norm() / big[:B].norm())
            worst = max(worst, r if not splice else 0.0)
            tol = 2e-4 if not splice else 5e-2   # spliced: the tower's rounding-level batch dependence reaches the pooled feature
            assert r <= tol, (T, splice, B, r)
            print(f"T={T} splice={splice} B={B}: pooled rel_l2 vs the rows of the B=16 run {r:.2e}")
print("worst literal", worst)
