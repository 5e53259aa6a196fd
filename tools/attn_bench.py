"""Kernel-level timing rig for the decoder attention forward + backward at the unfrozen training shape (run under rocprofv3 --kernel-trace --stats;
tools/attn_ablate.sh loops it over FASTVLA_ATTN_ABL with the A/B build of the library)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vla-from-fastvlm_amd"))
from fastvla_hip import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--tokens", type=int, default=320)
ap.add_argument("--heads", type=int, default=14)
ap.add_argument("--kv-heads", type=int, default=2)
ap.add_argument("--head-dim", type=int, default=64)
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
B, T, H, KV, D = a.batch, a.tokens, a.heads, a.kv_heads, a.head_dim
qd, kd = H * D, KV * D
ld = qd + 2 * kd
torch.manual_seed(0)
qkv = torch.randn(B * T, ld, device=dev) * 0.8
dO = torch.randn(B * T, qd, device=dev)
dq = torch.empty(B * T, ld, device=dev)
osc = torch.empty(B * T, 2 * qd, dtype=torch.bfloat16, device=dev)
st = torch.empty(2 * B * H * T, device=dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
lib = _lib.load_testops()
for _ in range(a.iters):
    _lib.check(lib.fv_op_attention_bwd(qkv.data_ptr(), ld, dO.data_ptr(), dq.data_ptr(), osc.data_ptr(), st.data_ptr(), B, T, H, KV, D, lens.data_ptr(), 1e6,
                                       torch.cuda.current_stream().cuda_stream), "fv_op_attention_bwd")
torch.cuda.synchronize()
print("ok", float(dq.abs().mean()))
