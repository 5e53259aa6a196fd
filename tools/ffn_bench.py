"""tools/ffn_bench.py -- A/B of the two fused ConvFFN kernels (16x16x32 vs 32x32x16 MFMA) at the tower's real shapes, interleaved
rounds in ONE process on random data (cdna_hip_programming.md rules 24, 25).  python tools/ffn_bench.py [rounds]"""
import math
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "vla-from-fastvlm_amd"))
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT))
import fastvla_hip  # noqa: E402
from test_gpu_ops import _pack_w2, _pack_wq  # noqa: E402

lib = fastvla_hip._lib.load_testops()
dev = "cuda:0"
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
st = torch.cuda.current_stream().cuda_stream
for C, M in ((384, 64 * 64 * 64), (192, 64 * 128 * 128), (96, 64 * 256 * 256)):
    Hd = 4 * C
    x = torch.randn(M, C, device=dev).bfloat16()
    res = torch.randn(M, C, device=dev).bfloat16()
    w1 = (torch.randn(Hd, C) / math.sqrt(C)).bfloat16()
    w2 = (torch.randn(C, Hd) / math.sqrt(Hd)).bfloat16()
    b1, b2, ls = torch.randn(Hd, device=dev) * 0.1, torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) * 0.3
    w1d, w2p, wq = w1.to(dev), _pack_w2(w2.float()).bfloat16().to(dev), _pack_wq(w1.float(), w2.float()).bfloat16().to(dev)
    out = torch.empty_like(x)
    fns = {"16x16x32": lambda: lib.fv_op_convffn(x.data_ptr(), w1d.data_ptr(), b1.data_ptr(), w2p.data_ptr(), b2.data_ptr(), ls.data_ptr(), res.data_ptr(), out.data_ptr(), M, C, st),
           "32x32x16": lambda: lib.fv_op_convffn32(x.data_ptr(), wq.data_ptr(), b1.data_ptr(), b2.data_ptr(), ls.data_ptr(), res.data_ptr(), out.data_ptr(), M, C, st)}
    times = {k: [] for k in fns}
    for r in range(rounds + 1):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                assert f() == 0
            e1.record()
            e1.synchronize()
            if r:
                times[k].append(e0.elapsed_time(e1) / 4)
    fl = 16.0 * M * C * C
    print(f"C={C} M={M}: " + "  ".join(f"{k}: med {sorted(v)[len(v)//2]*1e3:.0f} us min {min(v)*1e3:.0f} us = {fl/min(v)/1e9:.0f} TF ({fl/min(v)/1e9/2500:.3f})" for k, v in times.items()))

    if hasattr(lib, "fv_dbg_ffn32_stamps"):
        import ctypes
        buf = (ctypes.c_ulonglong * (256 * 8))()
        lib.fv_dbg_ffn32_stamps.argtypes = [ctypes.c_void_p]
        torch.cuda.synchronize()
        assert lib.fv_dbg_ffn32_stamps(buf) == 0
        import statistics
        rows = [[buf[b * 8 + z] for z in range(5)] for b in range(256)]
        rows = [r for r in rows if r[4]]
        med = [statistics.median(r[z] / r[4] for r in rows) for z in range(4)]
        print(f"   stamps per tile (wave 0, median over blocks, clk): chunk loop {med[0]:.0f}  epilogue head {med[1]:.0f}  passes {med[2]:.0f}  (residual wait {med[3]:.0f})  tiles/block {rows[0][4]}")
