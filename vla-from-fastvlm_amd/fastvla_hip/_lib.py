"""ctypes binding of libfastvla_hip.so (include/fastvla_hip.h).  No torch types cross this boundary: pointers are
plain integers (``tensor.data_ptr()``), the stream is ``torch.cuda.current_stream().cuda_stream``.

The product path has NO CPU fallback: if the shared library is missing or a call fails, ``FastVLAHipError`` is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

FV_MAX_STAGES = 8
FV_F32, FV_BF16, FV_U8, FV_I32 = 0, 1, 2, 3
EPI_BIAS, EPI_BIAS_GELU, EPI_LS_RES, EPI_RES_F32, EPI_SWIGLU, EPI_F32 = range(6)
EPI_SWIGLU_SPLIT = 7
EPI_SWIGLU_F16 = 8


class FastVLAHipError(RuntimeError):
    status = 0   # the fv_status the library returned (include/fastvla_hip.h), 0 when raised by the host side


class ModelDesc(C.Structure):
    _fields_ = [
        ("llm_hidden", C.c_int32), ("llm_layers", C.c_int32), ("llm_heads", C.c_int32), ("llm_kv_heads", C.c_int32),
        ("llm_head_dim", C.c_int32), ("llm_inter", C.c_int32), ("llm_vocab", C.c_int32),
        ("rope_theta", C.c_float), ("rms_eps", C.c_float),
        ("tower_stages", C.c_int32),
        ("tower_layers", C.c_int32 * FV_MAX_STAGES), ("tower_dims", C.c_int32 * FV_MAX_STAGES),
        ("tower_is_attn", C.c_int32 * FV_MAX_STAGES),
        ("tower_mlp_ratio", C.c_int32), ("tower_head_dim", C.c_int32), ("tower_se_rd", C.c_int32),
        ("tower_out_dim", C.c_int32),
        ("ln_eps", C.c_float), ("bn_eps", C.c_float),
        ("image_size", C.c_int32),
        ("state_dim", C.c_int32), ("action_dim", C.c_int32), ("hidden_dim", C.c_int32), ("fusion_dim", C.c_int32),
        ("max_batch", C.c_int32), ("max_text_tokens", C.c_int32), ("tower_microbatch", C.c_int32),
        ("llm_precision", C.c_int32),
    ]


class TensorDesc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4), ("device", C.c_int32), ("reserved", C.c_int32)]


class RcclId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


TENSOR_PROVIDER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(TensorDesc))


class AdamWHParams(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("max_grad_norm", C.c_float), ("grad_scale", C.c_float)]


class ProfileEntry(C.Structure):
    _fields_ = [("ms", C.c_double), ("flops", C.c_double), ("bytes", C.c_double), ("launches", C.c_int64)]


class GemmProfile(C.Structure):
    _fields_ = [("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("epi", C.c_int32), ("ms", C.c_double),
                ("launches", C.c_int64)]


class TrainTensor(C.Structure):
    _fields_ = [("name", C.c_char * 104), ("offset", C.c_int64), ("numel", C.c_int64), ("rows", C.c_int32), ("cols", C.c_int32),
                ("bucket", C.c_int32), ("packing", C.c_int32)]


BUCKET_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int64, C.c_int64)

FAMILIES = ("gemm", "dwconv", "stem", "attention", "norm", "elementwise", "head")

_vp, _i, _f, _u64, _i64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_int64

# name -> (restype, argtypes); this table is also what tests/test_abi.py checks against include/fastvla_hip.h
SIGNATURES = {
    "fv_create": (_i, [C.POINTER(ModelDesc), _i, C.POINTER(_vp)]),
    "fv_load_weights": (_i, [_vp, C.POINTER(TensorDesc), _i]),
    "fv_load_weights_cb": (_i, [_vp, TENSOR_PROVIDER, _vp]),
    "fv_destroy": (None, [_vp]),
    "fv_last_error": (C.c_char_p, [_vp]),
    "fv_version": (C.c_char_p, []),
    "fv_workspace_bytes": (_i, [_vp, _i, _i, _i, C.POINTER(C.c_size_t)]),
    "fv_bind_workspace": (_i, [_vp, _vp, C.c_size_t]),
    "fv_preprocess": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp]),
    "fv_preprocess_normalized": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _i, _vp, _vp]),
    "fv_vision_forward": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "fv_vision_forward_images": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp]),
    "fv_vision_forward_taps": (_i, [_vp, _vp, _i, _vp, _vp, C.POINTER(_vp), _i, _vp]),
    "fv_vision_unit_info": (_i, [_vp, _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "fv_vision_forward_unit_taps": (_i, [_vp, _vp, _i, _vp, _vp, C.POINTER(_vp), _i, _vp]),
    "fv_llm_forward_pooled": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "fv_llm_fp16_saturations": (_i, [_vp, C.POINTER(_u64), _i]),
    "fv_set_batch_invariant": (_i, [_vp, _i]),
    "fv_llm_prefix_bytes": (_i, [_vp, _i, _i, C.POINTER(C.c_size_t)]),
    "fv_llm_prefix": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "fv_llm_forward_pooled_prefixed": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "fv_head_layout": (_i, [_vp, C.POINTER(_i64 * 13)]),
    "fv_head_saved_bytes": (_i, [_vp, _i, C.POINTER(C.c_size_t)]),
    "fv_head_forward": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _u64, _u64, _vp, _vp, _vp]),
    "fv_head_set_io_norm": (_i, [_vp, _vp, _vp, _vp, _vp, _f]),
    "fv_head_mse_backward": (_i, [_vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _vp]),
    "fv_head_backward": (_i, [_vp, _vp, _vp, _i, _f, _vp, _vp, _vp]),
    "fv_grad_accumulate": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "fv_grad_scale": (_i, [_vp, _vp, _i64, _vp, _vp]),
    "fv_comm_unique_id": (_i, [_vp, C.POINTER(RcclId)]),
    "fv_comm_init": (_i, [_vp, C.POINTER(RcclId), _i, _i, C.POINTER(_vp)]),
    "fv_comm_destroy": (_i, [_vp, _vp]),
    "fv_allreduce_grads": (_i, [_vp, _vp, _vp, _i64, _vp]),
    "fv_train_layout": (_i, [_vp, C.POINTER(TrainTensor), _i, C.POINTER(_i), C.POINTER(_i64), C.POINTER(_i)]),
    "fv_train_begin": (_i, [_vp]),
    "fv_train_export_params": (_i, [_vp, _vp, _vp]),
    "fv_train_commit": (_i, [_vp, _vp, _vp]),
    "fv_train_set_options": (_i, [_vp, _i, _i, _i]),
    "fv_train_loss_scale": (_i, [_vp, C.POINTER(_f)]),
    "fv_train_set_forward_f16": (_i, [_vp, _i]),
    "fv_train_workspace_bytes": (_i, [_vp, _i, _i, C.POINTER(C.c_size_t)]),
    "fv_train_forward_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _u64, _u64, _vp, C.c_size_t, _vp, _vp, _vp, BUCKET_CB, _vp, _vp]),
    "fv_train_tower_begin": (_i, [_vp]),
    "fv_train_tower_workspace_bytes": (_i, [_vp, _i, C.POINTER(C.c_size_t)]),
    "fv_train_tower_forward": (_i, [_vp, _vp, _i, _vp, C.c_size_t, _vp, _vp]),
    "fv_train_set_tower_grad": (_i, [_vp, _vp]),
    "fv_train_tower_backward": (_i, [_vp, _vp, _vp, _i, _vp, C.c_size_t, _vp, BUCKET_CB, _vp, _vp]),
    "fv_train_tower_read_unit": (_i, [_vp, _i, _i, _vp, C.c_size_t, _vp, _vp]),
    "fv_train_tower_unit": (_i, [_vp, _i, _vp, _vp, _f, _i, _vp, C.c_size_t, _vp, _vp, _vp, _vp]),
    "fv_profile": (_i, [_vp, _i]),
    "fv_profile_read": (_i, [_vp, C.POINTER(ProfileEntry), C.POINTER(GemmProfile), _i, C.POINTER(_i)]),
    "fv_adamw_clip_step": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(AdamWHParams), _i64, _vp, _vp]),
}

# TEST-ONLY op-level entry points (include/fastvla_hip_testops.h, tests/_native/libfastvla_hip_testops.so): not part of the product library
OPS_SIGNATURES = {
    "fv_op_gemm": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "fv_op_gemm_f16": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _i, _i, _vp, C.c_size_t, _vp]),
    "fv_op_gemm_lo8": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _i, _i, _vp, C.c_size_t, _vp]),
    "fv_op_lo8_pack": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    "fv_op_gemm_ksplit": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _i, _i, _vp]),
    "fv_op_gemm_tn": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, C.c_size_t, _vp]),
    "fv_op_gemm_splitk": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _i, _i, _i, _vp, C.c_size_t, _vp]),
    "fv_op_dwconv": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "fv_op_stem_conv": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "fv_op_stem_mfma": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "fv_op_stem_fused": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "fv_op_stem_fused_images": (_i, [_vp, _i, _i, _i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "fv_op_layernorm_rows": (_i, [_vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "fv_op_attention": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _f, _vp]),
    "fv_op_rmsnorm": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp]),
    "fv_op_rope": (_i, [_vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "fv_op_dwconv_mfma": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fv_op_dwconv_s2_mfma": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "fv_op_dwconv_pair": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "fv_op_convffn": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "fv_op_convffn32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "fv_op_convffn32_stash": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "fv_op_gemm_f16_gelup": (_i, [_vp, _i, _vp, _i, _i, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "fv_op_dw_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, C.c_size_t, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "fv_op_convffn32_split": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, C.c_size_t, _vp]),
    "fv_op_attention_bwd": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _f, _vp]),
    "fv_op_rmsnorm_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp]),
    "fv_op_se_gelu": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
}

_LIB = None
_OPS = None


def library_path() -> Path:
    env = os.environ.get("FASTVLA_HIP_LIB")
    return Path(env) if env else Path(__file__).resolve().parent / "libfastvla_hip.so"


def load():
    """dlopen the library and attach prototypes.  Raises FastVLAHipError if it cannot be loaded."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not path.is_file():
        raise FastVLAHipError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C vla-from-fastvlm_amd/csrc).  There is no CPU fallback.")
    try:
        lib = C.CDLL(str(path))
    except OSError as exc:  # e.g. libamdhip64 missing
        raise FastVLAHipError(f"cannot load {path}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def testops_path() -> Path:
    """The test-only op library that goes with library_path(): FASTVLA_HIP_TESTOPS_LIB, or `<name>_testops.so` beside a FASTVLA_HIP_LIB override (the tools'
    A/B build), or tests/_native/libfastvla_hip_testops.so of this checkout."""
    env = os.environ.get("FASTVLA_HIP_TESTOPS_LIB")
    if env:
        return Path(env)
    if os.environ.get("FASTVLA_HIP_LIB"):
        lp = library_path()
        return lp.with_name(lp.stem + "_testops.so")
    return Path(__file__).resolve().parents[2] / "tests" / "_native" / "libfastvla_hip_testops.so"


class _WithTestOps:
    """What the parity tests and tools/ call through: product entry points from libfastvla_hip.so, fv_op_* from the test-only library."""

    def __init__(self, product, ops):
        self._product, self._ops = product, ops

    def __getattr__(self, name):
        try:
            return getattr(self._product, name)
        except AttributeError:
            return getattr(self._ops, name)


def load_testops():
    """TESTS / tools only: libfastvla_hip.so (re-opened RTLD_GLOBAL so that the op library's fv::launch_* references bind to it) + the op library."""
    global _OPS
    if _OPS is not None:
        return _OPS
    product = load()
    if hasattr(product, "fv_op_gemm"):      # a tools/ variant build that links the op entry points into the one library (tools/dw_variants.sh, ffn32_variants.sh)
        for name, (res, args) in OPS_SIGNATURES.items():
            fn = getattr(product, name)
            fn.restype = res
            fn.argtypes = args
        _OPS = product
        return _OPS
    path = testops_path()
    if not path.is_file():
        raise FastVLAHipError(f"{path} not found: `make -C vla-from-fastvlm_amd/csrc` builds it beside the product library (test-only entry points)")
    try:
        C.CDLL(str(library_path()), mode=C.RTLD_GLOBAL)
        ops = C.CDLL(str(path))
    except OSError as exc:
        raise FastVLAHipError(f"cannot load {path}: {exc}") from exc
    for name, (res, args) in OPS_SIGNATURES.items():
        fn = getattr(ops, name)
        fn.restype = res
        fn.argtypes = args
    _OPS = _WithTestOps(product, ops)
    return _OPS


def check(rc: int, what: str = "", handle=None) -> None:
    if rc != 0:
        msg = load().fv_last_error(handle)
        err = FastVLAHipError(f"{what or 'libfastvla_hip'} failed (status {rc}): {msg.decode() if msg else '?'}")
        err.status = int(rc)
        raise err
