"""tools/dwpair_bench.py -- times fv_op_dwconv_pair (RepMixer 3x3 + ConvFFN 7x7 march) at the tower's three real shapes on random
data, interleaved rounds; prints us per launch and the algorithmic HBM rate (3 tensor passes).  python tools/dwpair_bench.py [rounds]"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vla-from-fastvlm_amd"))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import fastvla_hip  # noqa: E402
from test_gpu_ops import _toeplitz  # noqa: E402

lib = fastvla_hip._lib.load_testops()
dev = "cuda:0"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
st = torch.cuda.current_stream().cuda_stream
B = 64
out = []
for C, H in ((384, 64), (192, 128), (96, 256)):
    x = torch.randn(B, H, H, C, device=dev).bfloat16()
    w3, w7 = (torch.randn(C, 1, 3, 3) / 3).bfloat16().float(), (torch.randn(C, 1, 7, 7) / 7).bfloat16().float()
    t3, t7 = _toeplitz(w3, 3).bfloat16().to(dev), _toeplitz(w7, 7).bfloat16().to(dev)
    b3, b7 = torch.randn(C, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    ts = []
    for r in range(rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            assert lib.fv_op_dwconv_pair(x.data_ptr(), t3.data_ptr(), b3.data_ptr(), t7.data_ptr(), b7.data_ptr(), y1.data_ptr(), y2.data_ptr(), B, H, H, C, st) == 0
        e1.record()
        e1.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / 4)
    by = 3.0 * x.numel() * 2
    out.append(f"C={C}: {min(ts)*1e3:.0f} us = {by/min(ts)/1e9:.2f} TB/s")
    if hasattr(lib, "fv_dbg_dp_stamps"):
        import ctypes, statistics
        buf = (ctypes.c_ulonglong * (512 * 16))()
        lib.fv_dbg_dp_stamps.argtypes = [ctypes.c_void_p]
        torch.cuda.synchronize()
        assert lib.fv_dbg_dp_stamps(buf) == 0
        rows = [[buf[b_ * 16 + z] for z in range(13)] for b_ in range(512)]
        rows = [r for r in rows if r[12]]
        names = ["bar1", "3x3+x'->lds", "bar2", "wait+x->lds", "x loads", "x' emit", "7x7", "bar3", "t->lds", "bar4", "t emit", "loop"]
        med = [statistics.median(r[z] / r[12] for r in rows) for z in range(12)]
        chb = 64 if C % 64 == 0 else 32       # the library's one geometry per shape (the FASTVLA_DWPAIR_GEO switch is gone: round 4)
        tw = 16 if chb == 64 else 32
        nblk = B * ((H + tw - 1) // tw) * (C // chb)
        occ = 3 if (chb, tw) == (32, 16) else 2
        per_block_s = min(ts) * 1e-3 / max(1.0, nblk / (256.0 * occ))          # a block's lifetime if the rounds were equal
        ghz = sum(med) * rows[0][12] / per_block_s / 1e9
        print(f"C={C} clk per step (thread 0, median of {len(rows)} blocks): " + "  ".join(f"{n} {m:.0f}" for n, m in zip(names, med)) + f"  | sum {sum(med):.0f}  ~{ghz:.2f} GHz")
    del x, y1, y2
print("  ".join(out))
