"""Helpers shared by the -m gpu parity tests (all calls go through the C ABI of libfastvla_hip.so)."""
import ctypes as C

import torch

import fastvla_hip
from fastvla_hip import _lib

DEV = "cuda:0"


def lib():
    """product entry points + the TEST-ONLY fv_op_* ones (tests/_native/libfastvla_hip_testops.so, include/fastvla_hip_testops.h)"""
    return fastvla_hip._lib.load_testops()


def stream():
    return torch.cuda.current_stream().cuda_stream


def bf(t):
    """round to bf16-representable fp32 (CPU)"""
    return t.to(torch.bfloat16).to(torch.float32)


def dev_bf16(t):
    return t.to(torch.bfloat16).to(DEV).contiguous()


def dev_f32(t):
    return t.to(torch.float32).to(DEV).contiguous()


def rel_l2(out, ref):
    out, ref = out.double().flatten(), ref.double().flatten()
    return float((out - ref).norm() / ref.norm().clamp_min(1e-30))


def worst_row(out, ref):
    """(B, D) tensors -> the largest per-sample relative error ||out_b - ref_b|| / max(||ref_b||, rms_b' ||ref_b'|| / 4): the bound a
    single env's action has to meet, not the batch average (a row whose own norm is unusually small is measured against a quarter of
    the batch's typical row norm instead of blowing the ratio up)."""
    out, ref = out.double(), ref.double()
    rn = ref.norm(dim=1)
    floor = 0.25 * float(rn.pow(2).mean().sqrt())
    return float(((out - ref).norm(dim=1) / rn.clamp_min(max(floor, 1e-30))).max())


def check_close(out, ref, *, rel=4e-3, amax=2e-2, what=""):
    """out/ref CPU fp32.  rel: relative L2 error bound; amax: max-abs error bound relative to max|ref|."""
    assert out.shape == ref.shape, (what, out.shape, ref.shape)
    assert torch.isfinite(out).all(), f"{what}: non-finite output"
    r = rel_l2(out, ref)
    m = float((out - ref).abs().max() / ref.abs().max().clamp_min(1e-30))
    assert r <= rel and m <= amax, f"{what}: rel_l2={r:.3e} (<= {rel}), max_abs/max_ref={m:.3e} (<= {amax})"
    return r, m


def call(rc, what):
    _lib.check(rc, what)
