"""tools/power_by_kernel.py -- each hot kernel family looped back to back for ~4 s on random data at the tower's real shapes, with
wall-clock marks; tools/power_by_kernel.sh samples rocm-smi beside it and joins the two: board power and reported sclk per family
(is THIS kernel at the 1400 W cap, or is it slow for another reason?)."""
import math
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vla-from-fastvlm_amd"))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import fastvla_hip  # noqa: E402
from test_gpu_ops import _pack_wq, _toeplitz  # noqa: E402

lib = fastvla_hip._lib.load_testops()
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 else None   # family names to run (default: all)
TAG = sys.argv[3] if len(sys.argv) > 3 else ""


def loop(name, fn, unit, per_call):
    if ONLY and name not in ONLY:
        return
    name = name + TAG
    for _ in range(3):
        assert fn() == 0
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < SECS:
        for _ in range(20):
            assert fn() == 0
        torch.cuda.synchronize()
        n += 20
    t1 = time.time()
    print(f"MARK {name} {t0:.3f} {t1:.3f} {(t1 - t0) / n * 1e6:.0f} us/launch {per_call / ((t1 - t0) / n) / 1e12:.2f} {unit}", flush=True)
    time.sleep(1.5)


B = 64
for C, H in ((384, 64), (192, 128)):
    M, Hd = B * H * H, 4 * C
    x, res = torch.randn(M, C, device=dev).bfloat16(), torch.randn(M, C, device=dev).bfloat16()
    w1, w2 = (torch.randn(Hd, C) / math.sqrt(C)).bfloat16(), (torch.randn(C, Hd) / math.sqrt(Hd)).bfloat16()
    b1, b2, ls = torch.randn(Hd, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) * 0.3
    wq = _pack_wq(w1.float(), w2.float()).bfloat16().to(dev)
    out = torch.empty_like(x)
    loop(f"convffn32_C{C}", lambda: lib.fv_op_convffn32(x.data_ptr(), wq.data_ptr(), b1.data_ptr(), b2.data_ptr(), ls.data_ptr(), res.data_ptr(), out.data_ptr(), M, C, st),
         "PF", 16.0 * M * C * C / 1e3)
    x4 = x.view(B, H, H, C)
    w3, w7 = (torch.randn(C, 1, 3, 3) / 3).bfloat16().float(), (torch.randn(C, 1, 7, 7) / 7).bfloat16().float()
    t3, t7 = _toeplitz(w3, 3).bfloat16().to(dev), _toeplitz(w7, 7).bfloat16().to(dev)
    b3, b7 = torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
    y1, y2 = torch.empty_like(x4), torch.empty_like(x4)
    loop(f"dwpair_C{C}", lambda: lib.fv_op_dwconv_pair(x4.data_ptr(), t3.data_ptr(), b3.data_ptr(), t7.data_ptr(), b7.data_ptr(), y1.data_ptr(), y2.data_ptr(), B, H, H, C, st),
         "TB/s", 3.0 * x.numel() * 2)
    del x, res, out, y1, y2, x4
# the tower's fc1-shaped GEMM (stage 4: 65536 x 3072 x 768) and fc2 (65536 x 768 x 3072), bf16, bias epilogue
for (Mg, N, K) in ((65536, 3072, 768), (65536, 768, 3072)):
    A = torch.randn(Mg, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, device=dev) * 0.1
    O = torch.empty(Mg, N, device=dev, dtype=torch.bfloat16)
    loop(f"gemm256_{Mg}x{N}x{K}", lambda: lib.fv_op_gemm(A.data_ptr(), K, W.data_ptr(), Mg, N, K, bias.data_ptr(), None, None, 0, O.data_ptr(), N, 0, st), "PF", 2.0 * Mg * N * K / 1e3)
    del A, W, O
# a plain HBM stream for scale: torch copy of 1 GB
src = torch.empty(1 << 29, device=dev, dtype=torch.bfloat16).normal_()
dst = torch.empty_like(src)


def cp():
    dst.copy_(src)
    return 0


loop("copy_1GiB", cp, "TB/s", 2.0 * src.numel() * 2)
