"""tools/gemm_stamps.py -- where a K-tile of gemm256_kernel<8,4> goes (needs a -DG2_STAMPS build: tools/variants.sh gemm_bf16.hip
stamps:-DG2_STAMPS; FASTVLA_HIP_LIB=tools/bin/libv_stamps.so python tools/gemm_stamps.py)"""
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vla-from-fastvlm_amd"))
from fastvla_hip import _lib  # noqa: E402

L = _lib.load_testops()
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
B = 64
for name, M, N, K, epi in (("tower s4 fc1", B * 256, 3072, 768, _lib.EPI_BIAS_GELU), ("tower s4 fc2", B * 256, 768, 3072, _lib.EPI_LS_RES),
                           ("tower s4 qkv", B * 256, 2304, 768, _lib.EPI_BIAS), ("square 8192", 8192, 8192, 8192, _lib.EPI_BIAS)):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias, scale = torch.randn(N, device=dev), torch.rand(N, device=dev)
    res = torch.randn(M, N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ts = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            assert L.fv_op_gemm(A.data_ptr(), K, W.data_ptr(), M, N, K, bias.data_ptr(), scale.data_ptr(), res.data_ptr(), N, out.data_ptr(), N, epi, s) == 0
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) / 4)
    t = min(ts[1:])
    print(f"{name}: {M}x{N}x{K}: {t*1e3:.0f} us = {2.0*M*N*K/t/1e9:.0f} TF")
    if hasattr(L, "fv_dbg_g2_stamps"):
        buf = (ctypes.c_ulonglong * (256 * 8))()
        L.fv_dbg_g2_stamps.argtypes = [ctypes.c_void_p]
        torch.cuda.synchronize()
        assert L.fv_dbg_g2_stamps(buf) == 0
        rows = [[buf[b * 8 + z] for z in range(8)] for b in range(256)]
        rows = [r for r in rows if r[6]]
        nm = ["stage issue", "reads+64 MFMA", "vmcnt(0) wait", "barrier", "loop top", "epilogue(/tile)"]
        med = [statistics.median(r[z] / r[6] for r in rows) for z in range(6)]
        tot = statistics.median(sum(r[:6]) for r in rows)
        print("   clk per K-tile (wave 0): " + "  ".join(f"{n} {m:.0f}" for n, m in zip(nm, med)) + f"  | sum {sum(med):.0f}; MFMA pipe 1024/wave; kernel clk {tot:.0f} -> {tot/t/1e6:.2f} GHz")
