"""tools/dwpair_check.py -- fv_op_dwconv_pair at the tower's full shapes against torch's own grouped convolutions on the GPU (fp32,
x' rounded to bf16 in between); prints where the mismatches sit.  FASTVLA_DWPAIR_GEO=0|1|2 selects the geometry."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vla-from-fastvlm_amd"))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import fastvla_hip  # noqa: E402
from test_gpu_ops import _toeplitz  # noqa: E402

lib = fastvla_hip._lib.load_testops()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
for C, H, B in ((384, 64, 8), (192, 128, 4), (96, 256, 2)):
    x = torch.randn(B, H, H, C, device=dev).bfloat16()
    w3, w7 = (torch.randn(C, 1, 3, 3) / 3).bfloat16().float(), (torch.randn(C, 1, 7, 7) / 7).bfloat16().float()
    b3, b7 = (torch.randn(C) * 0.1).to(dev), (torch.randn(C) * 0.1).to(dev)
    t3, t7 = _toeplitz(w3, 3).bfloat16().to(dev), _toeplitz(w7, 7).bfloat16().to(dev)
    xc = x.float().permute(0, 3, 1, 2)
    r1 = F.conv2d(xc, w3.to(dev), b3, padding=1, groups=C).bfloat16().float()
    r2 = F.conv2d(r1, w7.to(dev), b7, padding=3, groups=C)
    for rep in range(3):
        y1 = torch.full_like(x, float("nan"))
        y2 = torch.full_like(x, float("nan"))
        assert lib.fv_op_dwconv_pair(x.data_ptr(), t3.data_ptr(), b3.data_ptr(), t7.data_ptr(), b7.data_ptr(), y1.data_ptr(), y2.data_ptr(), B, H, H, C, st) == 0
        torch.cuda.synchronize()
        for name, y, r in (("x'", y1, r1), ("t", y2, r2)):
            d = (y.float().permute(0, 3, 1, 2) - r).abs()
            bad = (d > 0.05 + 0.02 * r.abs()) | ~torch.isfinite(d)
            n = int(bad.sum())
            msg = f"C={C} H={H} rep {rep} {name}: max err {float(d[torch.isfinite(d)].max()):.3g}, bad {n}"
            if n:
                idx = bad.nonzero()
                msg += f"  images {sorted(set(idx[:, 0].tolist()))[:8]} ch {sorted(set(idx[:, 1].tolist()))[:12]} rows {sorted(set(idx[:, 2].tolist()))[:24]} cols {sorted(set(idx[:, 3].tolist()))[:24]}"
            print(msg)
