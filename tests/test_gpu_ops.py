"""-m gpu: every HIP kernel of the path, called through the C ABI, against the fp32 oracle of the same op.

Inputs are bf16-representable, so the only differences are fp32 summation order and the final bf16 rounding of the
kernel's output (relative 2^-9): tolerances are rel-L2 <= 4e-3 and max-abs <= 2e-2 of max|ref| for bf16 outputs,
1e-5-class for fp32 outputs.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from gpu_util import DEV, bf, call, check_close, dev_bf16, dev_f32, lib, stream  # noqa: E402
from fastvla_hip import _lib  # noqa: E402
from oracle import fastvit_hd, preprocess, qwen2  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device (no CPU fallback in the product path)")


def _gemm(A, W, epi, bias=None, scale=None, res=None, lda=None, out_f32=False, ncols=None):
    M, K = A.shape
    N = W.shape[0]
    ncols = ncols or N
    a = dev_bf16(A)
    if lda:
        buf = torch.zeros(M, lda, dtype=torch.bfloat16, device=DEV)
        buf[:, :K] = a
        a = buf
    w = dev_bf16(W)
    out = torch.full((M, ncols), float("nan"), dtype=torch.float32 if out_f32 else torch.bfloat16, device=DEV)
    b = dev_f32(bias) if bias is not None else None
    s = dev_f32(scale) if scale is not None else None
    r = None
    if res is not None:
        r = dev_f32(res) if out_f32 else dev_bf16(res)
    p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
    call(lib().fv_op_gemm(a.data_ptr(), lda or K, w.data_ptr(), M, N, K, p(b), p(s), p(r), N if r is not None else 0,
                          out.data_ptr(), ncols, epi, stream()), "fv_op_gemm")
    torch.cuda.synchronize()
    return out.float().cpu()


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 96, 96), (200, 384, 96), (77, 1152, 896), (1024, 896, 4864),
                                   (130, 3072, 1536), (4096, 96, 384)])
def test_gemm_bias_and_gelu(M, N, K):
    torch.manual_seed(M + N + K)
    A, W, b = bf(torch.randn(M, K)), bf(torch.randn(N, K) / math.sqrt(K)), torch.randn(N) * 0.1
    ref = A.double() @ W.double().t() + b.double()
    check_close(_gemm(A, W, _lib.EPI_BIAS, bias=b), ref.float(), what=f"gemm bias {M}x{N}x{K}")
    check_close(_gemm(A, W, _lib.EPI_BIAS_GELU, bias=b), F.gelu(ref.float()), what=f"gemm gelu {M}x{N}x{K}")
    check_close(_gemm(A, W, _lib.EPI_BIAS), (A.double() @ W.double().t()).float(), what="gemm nobias")


def test_gemm_is_not_transposed():
    # A = I with an ASYMMETRIC W catches a swapped row/col map in the C write
    n = 128
    W = bf(torch.arange(n * n, dtype=torch.float32).view(n, n) % 251 - 100.0)
    out = _gemm(torch.eye(n), W, _lib.EPI_BIAS)
    assert torch.equal(out, W.t().contiguous())


def test_gemm_layerscale_residual_and_strided_a():
    torch.manual_seed(3)
    M, N, K = 300, 192, 768
    A, W = bf(torch.randn(M, K)), bf(torch.randn(N, K) / math.sqrt(K))
    b, ls, res = torch.randn(N) * 0.1, torch.rand(N) * 0.3, bf(torch.randn(M, N))
    ref = res.double() + ls.double() * (A.double() @ W.double().t() + b.double())
    check_close(_gemm(A, W, _lib.EPI_LS_RES, bias=b, scale=ls, res=res, lda=K + 64), ref.float(), what="ls_res")


def test_gemm_f32_residual_and_f32_out():
    torch.manual_seed(4)
    M, N, K = 260, 896, 4864
    A, W, res = bf(torch.randn(M, K)), bf(torch.randn(N, K) * 0.02), torch.randn(M, N)
    ref = (res.double() + A.double() @ W.double().t()).float()
    check_close(_gemm(A, W, _lib.EPI_RES_F32, res=res, out_f32=True), ref, rel=2e-5, amax=2e-5, what="res_f32")
    b = torch.randn(N)
    ref = (A.double() @ W.double().t() + b.double()).float()
    check_close(_gemm(A, W, _lib.EPI_F32, bias=b, out_f32=True), ref, rel=2e-5, amax=2e-5, what="f32 out")


def test_gemm_swiglu_interleaved():
    torch.manual_seed(5)
    M, I, K = 150, 256, 128
    A, G, U = bf(torch.randn(M, K)), bf(torch.randn(I, K) / math.sqrt(K)), bf(torch.randn(I, K) / math.sqrt(K))
    Wi = torch.empty(2 * I, K)
    j = torch.arange(I)
    Wi[(j // 8) * 16 + j % 8] = G
    Wi[(j // 8) * 16 + 8 + j % 8] = U
    ref = F.silu(A.double() @ G.double().t()) * (A.double() @ U.double().t())
    check_close(_gemm(A, Wi, _lib.EPI_SWIGLU, ncols=I), ref.float(), what="swiglu")


def test_gemm_rejects_bad_shapes():
    a = torch.zeros(8, 8, dtype=torch.bfloat16, device=DEV)
    rc = lib().fv_op_gemm(a.data_ptr(), 8, a.data_ptr(), 8, 8, 4, None, None, None, 0, a.data_ptr(), 8, 0, stream())
    assert rc == -1 and b"multiples of 8" in lib().fv_last_error(None)
    rc = lib().fv_op_gemm(None, 8, a.data_ptr(), 8, 8, 8, None, None, None, 0, a.data_ptr(), 8, 0, stream())
    assert rc == -1


@pytest.mark.parametrize("k,stride,mult,gelu,C,H,W", [(3, 1, 1, 0, 96, 20, 24), (7, 1, 1, 0, 64, 19, 33),
                                                       (3, 2, 1, 1, 32, 32, 32), (7, 2, 2, 1, 64, 24, 24),
                                                       (3, 1, 2, 0, 512, 4, 4), (7, 1, 1, 0, 768, 8, 8),
                                                       (7, 1, 1, 0, 96, 5, 3), (7, 1, 1, 0, 96, 40, 64), (3, 1, 1, 0, 32, 33, 47),
                                                       (7, 1, 1, 1, 192, 16, 32)])
def test_dwconv(k, stride, mult, gelu, C, H, W):
    torch.manual_seed(k * 100 + C)
    B = 2
    x = bf(torch.randn(B, C, H, W))
    w = torch.randn(C * mult, 1, k, k) / k
    b = torch.randn(C * mult) * 0.1
    ref = F.conv2d(x, w, b, stride=stride, padding=k // 2, groups=C)
    if gelu:
        ref = F.gelu(ref)
    Ho, Wo = ref.shape[2:]
    xd = dev_bf16(x.permute(0, 2, 3, 1))
    wd = dev_f32(w.view(C * mult, k * k).t())  # [k*k][Cout]
    bd = dev_f32(b)  # keep every device operand alive in a named variable until the sync
    y = torch.full((B, Ho, Wo, C * mult), float("nan"), dtype=torch.bfloat16, device=DEV)
    call(lib().fv_op_dwconv(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, H, W, C, k, stride, mult,
                            gelu, stream()), "fv_op_dwconv")
    torch.cuda.synchronize()
    check_close(y.float().cpu().permute(0, 3, 1, 2), ref, what=f"dwconv k{k} s{stride} m{mult}")


def test_stem_conv_and_letterbox():
    torch.manual_seed(11)
    B, S, C0 = 2, 64, 32
    img = torch.rand(B, 3, 21, 30)
    ref_pix = preprocess.letterbox(img, S, pad_value=0.25)
    pix = torch.empty(B, S, S, 4, dtype=torch.bfloat16, device=DEV)
    # fv_preprocess needs a handle for image_size only: use the letterbox through a tiny engine-free path
    from fastvla_hip import FastVLAEngine, arch
    m = arch.ModelConfig("lb", arch.LLMConfig(hidden=64, layers=1, heads=2, kv_heads=1, head_dim=32, inter=64, vocab=64),
                         arch.TowerConfig(layers=(1, 1, 1, 1, 1), dims=(32, 64, 128, 256, 512), image_size=S))
    eng = FastVLAEngine(m, hidden_dim=32, fusion_dim=32)
    pix = eng.preprocess(img.to(DEV), pad_value=0.25)
    torch.cuda.synchronize()
    got = pix.float().cpu()
    assert float(got[..., 3].abs().max()) == 0.0
    check_close(got[..., :3].permute(0, 3, 1, 2), ref_pix, rel=3e-3, amax=5e-3, what="letterbox")
    u8 = (torch.rand(1, 1, 40, 17) * 255).to(torch.uint8)
    ref_u8 = preprocess.letterbox(u8, S)
    got = eng.preprocess(u8.to(DEV)).float().cpu()[..., :3].permute(0, 3, 1, 2)
    check_close(got, ref_u8, rel=3e-3, amax=5e-3, what="letterbox u8 gray")
    got = eng.preprocess(img.to(DEV), resize_with_padding=False).float().cpu()[..., :3].permute(0, 3, 1, 2)
    check_close(got, preprocess.letterbox(img, S, resize_with_padding=False), rel=3e-3, amax=5e-3, what="stretch")
    with pytest.raises(ValueError):
        eng.preprocess(torch.zeros(3, 8, 8))
    # ---- normalize_imagenet (reference fastvlm_adapter.py:463-477) folded into the letterbox: against the golden vectors the imported reference produced
    # (its branch without the value-range test) and against the oracle's torchvision branch, whose `max > 1.5 -> / 255` is decided on the device
    import numpy as np
    from pathlib import Path
    g = np.load(Path(__file__).parent / "golden" / "g1_normalize.npz")
    for k in ("unit", "gray", "wide"):
        src, ref = torch.from_numpy(g[k]), torch.from_numpy(g[k + "_out"])
        got = eng.preprocess(src.to(DEV), pad_value=0.25, normalize_imagenet=True, range_heuristic=bool(int(g["has_torchvision"])))
        torch.cuda.synchronize()
        got = got.float().cpu()
        assert float(got[..., 3].abs().max()) == 0.0
        check_close(got[..., :3].permute(0, 3, 1, 2), ref, rel=3e-3, amax=5e-3 * max(1.0, float(ref.abs().max())), what=f"normalize_imagenet golden {k}")
    for name, src in (("0..255 f32", torch.rand(2, 3, 33, 20) * 255.0), ("u8", (torch.rand(1, 3, 19, 40) * 255).to(torch.uint8)), ("0..1 f32", torch.rand(2, 3, 33, 20)),
                      ("0..1 with a pad of 2", torch.rand(1, 3, 10, 30))):
        pad = 2.0 if "pad of 2" in name else 0.25    # the pad pixels count in the maximum the reference tests (x.max() of the letterboxed tensor)
        ref = preprocess.prepare_images(src.float(), S, pad, True, normalize=True, torchvision_branch=True)
        got = eng.preprocess(src.to(DEV), pad_value=pad, normalize_imagenet=True).float().cpu()[..., :3].permute(0, 3, 1, 2)
        check_close(got, ref, rel=3e-3, amax=5e-3 * max(1.0, float(ref.abs().max())), what=f"normalize_imagenet {name}")
        # one rounding only: the result is the bf16 image of the fp32 oracle value up to the last fp32 bit of the interpolation
        assert float((got - ref.bfloat16().float()).abs().max()) <= 2.0 ** -7 * max(1.0, float(ref.abs().max()))
    # stem conv on the bf16 pixels
    x = pix.float().cpu()[..., :3].permute(0, 3, 1, 2)
    w = torch.randn(C0, 3, 3, 3) / 5
    b = torch.randn(C0) * 0.1
    ref = F.gelu(F.conv2d(x, w, b, stride=2, padding=1))
    wd = dev_f32(w.permute(2, 3, 1, 0).reshape(27, C0))  # [(ky,kx,ci)][co]
    bd = dev_f32(b)
    y = torch.full((B, S // 2, S // 2, C0), float("nan"), dtype=torch.bfloat16, device=DEV)
    call(lib().fv_op_stem_conv(pix.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, S, C0, stream()),
         "fv_op_stem_conv")
    torch.cuda.synchronize()
    check_close(y.float().cpu().permute(0, 3, 1, 2), ref, what="stem conv")
    eng.close()


def test_layernorm_rows_and_rmsnorm():
    torch.manual_seed(12)
    for rows, C in [(37, 768), (5, 1536), (64, 256)]:
        x = bf(torch.randn(rows, C) * 2 + 0.5)
        w, b = 1 + 0.1 * torch.randn(C), 0.1 * torch.randn(C)
        ref = F.layer_norm(x, (C,), w, b, 1e-5)
        y = torch.empty(rows, C, dtype=torch.bfloat16, device=DEV)
        xd, wd, bd = dev_bf16(x), dev_f32(w), dev_f32(b)
        call(lib().fv_op_layernorm_rows(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(),
                                        rows, C, 1e-5, stream()), "ln rows")
        torch.cuda.synchronize()
        check_close(y.float().cpu(), ref, what="layernorm rows")
    for rows, H in [(70, 896), (3, 3584), (9, 128)]:
        x = torch.randn(rows, H) * 3
        w = 1 + 0.1 * torch.randn(H)
        y = torch.empty(rows, H, dtype=torch.bfloat16, device=DEV)
        xd, wd = dev_f32(x), dev_f32(w)
        call(lib().fv_op_rmsnorm(xd.data_ptr(), wd.data_ptr(), y.data_ptr(), rows, H, 1e-6, stream()), "rmsnorm")
        torch.cuda.synchronize()
        check_close(y.float().cpu(), qwen2.rmsnorm(x, w, 1e-6), what="rmsnorm")


def test_rope_matches_rotate_half():
    torch.manual_seed(13)
    B, T, heads, kv, D = 2, 9, 4, 2, 64
    ld = (heads + 2 * kv) * D
    qkv = bf(torch.randn(B * T, ld))
    d = dev_bf16(qkv)
    call(lib().fv_op_rope(d.data_ptr(), ld, B * T, T, heads, kv, D, 1e6, stream()), "rope")
    torch.cuda.synchronize()
    cfg = qwen2.Qwen2Cfg(head_dim=D, rope_theta=1e6)
    cos, sin = qwen2.rope_tables(cfg, torch.arange(T))
    x = qkv.view(B, T, heads + 2 * kv, D)
    rot = x * cos[None, :, None, :] + qwen2._rotate_half(x) * sin[None, :, None, :]
    ref = x.clone()
    ref[:, :, : heads + kv] = rot[:, :, : heads + kv]  # v untouched
    check_close(d.float().cpu().view(B, T, heads + 2 * kv, D), ref, what="rope")


def _attn_ref(q, k, v, causal, lens, scale):
    B, T, Hh, D = q.shape
    g = Hh // k.shape[2]
    kk, vv = k.repeat_interleave(g, dim=2), v.repeat_interleave(g, dim=2)
    s = torch.einsum("bqhd,bkhd->bhqk", q.double(), kk.double()) * scale
    pos = torch.arange(T)
    mask = torch.ones(B, 1, T, T, dtype=torch.bool)
    if causal:
        mask = mask & (pos[None, :] <= pos[:, None])[None, None]
    if lens is not None:
        mask = mask & (pos[None, :] < lens[:, None])[:, None, None, :]
    s = s.masked_fill(~mask, -1e300)
    return torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, dim=-1), vv.double()).float()


@pytest.mark.parametrize("B,T,heads,kv,D,causal", [(2, 256, 4, 4, 32, 0), (1, 1024, 2, 2, 32, 0), (3, 64, 14, 2, 64, 1),
                                                   (1, 64, 1, 1, 32, 0), (2, 128, 3, 3, 32, 0), (1, 96, 2, 2, 32, 0),
                                                   (2, 77, 4, 2, 64, 1), (2, 40, 4, 1, 128, 1), (1, 320, 14, 2, 64, 1)])
def test_attention(B, T, heads, kv, D, causal):
    torch.manual_seed(T + D)
    ld = (heads + 2 * kv) * D
    qkv = bf(torch.randn(B, T, ld))
    q = qkv[..., : heads * D].reshape(B, T, heads, D)
    k = qkv[..., heads * D: (heads + kv) * D].reshape(B, T, kv, D)
    v = qkv[..., (heads + kv) * D:].reshape(B, T, kv, D)
    lens = None
    if causal:
        lens = torch.tensor([T, max(1, T // 2), 1][:B], dtype=torch.int32)
    d = dev_bf16(qkv)
    out = torch.full((B, T, heads * D), float("nan"), dtype=torch.bfloat16, device=DEV)
    ld_dev = None if lens is None else lens.to(DEV)
    base = d.data_ptr()
    call(lib().fv_op_attention(base, base + heads * D * 2, base + (heads + kv) * D * 2, ld, ld, ld, out.data_ptr(), heads * D,
                               B, T, heads, kv, D, causal, None if ld_dev is None else ld_dev.data_ptr(), D ** -0.5, stream()),
         "attention")
    torch.cuda.synchronize()
    ref = _attn_ref(q, k, v, causal, None if lens is None else lens.long(), D ** -0.5)
    got = out.float().cpu().view(B, T, heads, D)
    if lens is not None:  # rows past len are padding: never consumed, compare the valid ones
        for b in range(B):
            check_close(got[b, : int(lens[b])], ref[b, : int(lens[b])], rel=6e-3, what=f"attention b{b}")
    else:
        check_close(got, ref, rel=6e-3, what="attention")


def test_se_gelu_tail():
    torch.manual_seed(17)
    B, P, Cc, R = 3, 16, 1024, 64
    x = bf(torch.randn(B, P, Cc))
    w1, b1 = torch.randn(R, Cc) / 32, torch.randn(R) * 0.1
    w2, b2 = torch.randn(Cc, R) / 8, torch.randn(Cc) * 0.1
    s = torch.sigmoid(F.linear(F.relu(F.linear(x.mean(1), w1, b1)), w2, b2))
    ref = F.gelu(x * s[:, None, :])
    y = torch.empty(B, P, Cc, dtype=torch.bfloat16, device=DEV)
    scratch = torch.empty(B * (2 * Cc + R), dtype=torch.float32, device=DEV)
    xd, w1d, b1d, w2d, b2d = dev_bf16(x), dev_f32(w1), dev_f32(b1), dev_f32(w2), dev_f32(b2)
    call(lib().fv_op_se_gelu(xd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), w2d.data_ptr(),
                             b2d.data_ptr(), y.data_ptr(), scratch.data_ptr(), B, P, Cc, R, stream()), "se_gelu")
    torch.cuda.synchronize()
    check_close(y.float().cpu(), ref, what="se+gelu")


def _pack_w2(w2):
    """fc2 weight (C, 4C) -> [4C/32][C][32], slot 8g+j of each 32-block = hidden 16*(j>>2) + 4*g + (j&3) (fastvla_hip.h)."""
    Cc, Hd = w2.shape
    idx = torch.tensor([16 * (j >> 2) + 4 * g + (j & 3) for g in range(4) for j in range(8)])
    return w2.view(Cc, Hd // 32, 32)[:, :, idx].permute(1, 0, 2).contiguous()


# the last two cases give every persistent block (one per CU) two to three row tiles: the weight stream, the fragment ring
# and the x prefetch then run across tile boundaries, which the small cases never reach
@pytest.mark.parametrize("Cc,M", [(96, 1000), (192, 300), (384, 130), (32, 520), (64, 77), (128, 256), (96, 150013), (384, 70001)])
def test_fused_convffn(Cc, M):
    torch.manual_seed(Cc + M)
    Hd = 4 * Cc
    x, res = bf(torch.randn(M, Cc)), bf(torch.randn(M, Cc))
    w1, w2 = bf(torch.randn(Hd, Cc) / math.sqrt(Cc)), bf(torch.randn(Cc, Hd) / math.sqrt(Hd))
    b1, b2, ls = torch.randn(Hd) * 0.1, torch.randn(Cc) * 0.1, torch.rand(Cc) * 0.3 + 0.05
    wide = torch.float64 if M < 5000 else torch.float32   # the big cases keep the host reference to seconds
    hid = bf(F.gelu(x.to(wide) @ w1.to(wide).t() + b1.to(wide)).float())  # the kernel rounds the hidden to bf16 too
    ref = (res.to(wide) + ls.to(wide) * (hid.to(wide) @ w2.to(wide).t() + b2.to(wide))).float()
    xd, rd, w1d, w2d = dev_bf16(x), dev_bf16(res), dev_bf16(w1), dev_bf16(_pack_w2(w2))
    b1d, b2d, lsd = dev_f32(b1), dev_f32(b2), dev_f32(ls)
    out = torch.full((M, Cc), float("nan"), dtype=torch.bfloat16, device=DEV)
    call(lib().fv_op_convffn(xd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), lsd.data_ptr(),
                             rd.data_ptr(), out.data_ptr(), M, Cc, stream()), "fv_op_convffn")
    torch.cuda.synchronize()
    check_close(out.float().cpu(), ref, what=f"fused convffn C={Cc}")
    # in place on the residual buffer (how the engine calls it)
    call(lib().fv_op_convffn(xd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), lsd.data_ptr(),
                             rd.data_ptr(), rd.data_ptr(), M, Cc, stream()), "fv_op_convffn in place")
    torch.cuda.synchronize()
    assert torch.equal(rd, out)


def _pack_wq(w1, w2):
    """fc1 (4C,C) + fc2 (C,4C) -> the convffn32 weight stream [4C/32][64 C] (fastvla_hip.h, fv_op_convffn32)."""
    Cc, Hd = w2.shape
    nch = Hd // 32
    sh, mask = {384: (0, 15), 192: (1, 7), 96: (2, 3)}[Cc]
    r = torch.arange(32)
    c = torch.arange(Cc // 8)
    phys1 = c[None, :] ^ ((r[:, None] >> sh) & mask)                      # (32, C/8): physical chunk of logical chunk c in row r
    t1 = torch.empty(nch, 32, Cc // 8, 8)
    t1[:, r[:, None], phys1] = 0.25 * w1.view(nch, 32, Cc // 8, 8)           # W1 / 4 and 4 W2: exact (the GELU runs in y = x / 4)
    n = torch.arange(Cc)
    if (int(os.environ.get("FFN32_S16_MASK", "0")) >> {384: 0, 192: 1, 96: 2}[Cc]) & 1:   # 16x16x32 form of the kernel (experiment builds)
        hid = torch.tensor([[4 * g + j if j < 4 else 16 + 4 * g + j - 4 for j in range(8)] for g in range(4)])
        phys2 = torch.arange(4)[None, :] ^ ((4 - ((n[:, None] >> 2) & 3)) & 3)
    else:
        hid = torch.tensor([[16 * s + 8 * (j >> 2) + 4 * h + (j & 3) for j in range(8)] for s in range(2) for h in range(2)])  # (4 chunks, 8)
        phys2 = torch.arange(4)[None, :] ^ ((n[:, None] >> 2) & 3)              # (C, 4)
    w2c = 4.0 * w2.view(Cc, nch, 32)[:, :, hid]                       # (C, nch, 4, 8): logical chunks
    t2 = torch.empty(nch, Cc, 4, 8)
    t2[:, n[:, None], phys2] = w2c.permute(1, 0, 2, 3)
    T = torch.cat([t1.reshape(nch, -1), t2.reshape(nch, -1)], dim=1)        # (nch, 64 C) slot images
    G = T.view(nch, -1, 4, 64, 2).permute(0, 1, 3, 2, 4)                    # [kb][d][l][b] -> [kb][l][d][b]
    return G.reshape(nch, 64 * Cc).contiguous()


# the 32x32x16 variant the engine runs at the real tower widths; the big cases give every persistent block several row tiles
@pytest.mark.parametrize("Cc,M", [(96, 1000), (192, 300), (384, 130), (384, 128), (96, 150013), (192, 131072 + 77), (384, 70001),
                                  (384, 1), (192, 5), (96, 31)])   # the last three: fewer rows than one wave's tile (rows past M are clamped / dropped)
def test_fused_convffn32(Cc, M):
    torch.manual_seed(Cc + M + 1)
    Hd = 4 * Cc
    x, res = bf(torch.randn(M, Cc)), bf(torch.randn(M, Cc))
    w1, w2 = bf(torch.randn(Hd, Cc) / math.sqrt(Cc)), bf(torch.randn(Cc, Hd) / math.sqrt(Hd))
    b1, b2, ls = torch.randn(Hd) * 0.1, torch.randn(Cc) * 0.1, torch.rand(Cc) * 0.3 + 0.05
    wide = torch.float64 if M < 5000 else torch.float32
    hid = bf(F.gelu(x.to(wide) @ w1.to(wide).t() + b1.to(wide)).float())  # the kernel rounds the hidden to bf16 too
    ref = (res.to(wide) + ls.to(wide) * (hid.to(wide) @ w2.to(wide).t() + b2.to(wide))).float()
    xd, rd, w1d, wqd = dev_bf16(x), dev_bf16(res), dev_bf16(w1), dev_bf16(_pack_wq(w1, w2))
    b1d, b2d, lsd = dev_f32(b1), dev_f32(b2), dev_f32(ls)
    out = torch.full((M, Cc), float("nan"), dtype=torch.bfloat16, device=DEV)
    call(lib().fv_op_convffn32(xd.data_ptr(), wqd.data_ptr(), b1d.data_ptr(), b2d.data_ptr(), lsd.data_ptr(),
                               rd.data_ptr(), out.data_ptr(), M, Cc, stream()), "fv_op_convffn32")
    torch.cuda.synchronize()
    check_close(out.float().cpu(), ref, what=f"fused convffn32 C={Cc}")
    # run to run bit-identical (a stale MFMA operand behind an unpadded hazard shows as last-bit flicker, not as a wrong tile)
    out2 = torch.full((M, Cc), float("nan"), dtype=torch.bfloat16, device=DEV)
    call(lib().fv_op_convffn32(xd.data_ptr(), wqd.data_ptr(), b1d.data_ptr(), b2d.data_ptr(), lsd.data_ptr(),
                               rd.data_ptr(), out2.data_ptr(), M, Cc, stream()), "fv_op_convffn32 again")
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    # the two kernels compute the same thing: against each other the difference is accumulation order only
    out16 = torch.empty_like(out)
    call(lib().fv_op_convffn(xd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), dev_bf16(_pack_w2(w2)).data_ptr(), b2d.data_ptr(), lsd.data_ptr(),
                             rd.data_ptr(), out16.data_ptr(), M, Cc, stream()), "fv_op_convffn")
    torch.cuda.synchronize()
    check_close(out.float().cpu(), out16.float().cpu(), rel=2e-3, amax=2e-2, what=f"convffn32 vs convffn16 C={Cc}")
    # in place on the residual buffer (how the engine calls it)
    call(lib().fv_op_convffn32(xd.data_ptr(), wqd.data_ptr(), b1d.data_ptr(), b2d.data_ptr(), lsd.data_ptr(),
                               rd.data_ptr(), rd.data_ptr(), M, Cc, stream()), "fv_op_convffn32 in place")
    torch.cuda.synchronize()
    assert torch.equal(rd, out)


@pytest.mark.parametrize("Cc,M", [(384, 4096), (384, 1000), (192, 8192 + 77), (96, 16384), (96, 700)])
def test_fused_convffn32_stash_and_gelup_epilogue(Cc, M):
    """Round 6, the tower's TRAINING forward: the fused ConvFFN leaves its pre-activation behind (convffn32.hip STASH: stash_y = (fc1(x) + b1) / 4 as fp16, natural column
    order) with the SAME output bits as the inference instance; and the fc2 input gradient's epilogue FV_EPI_MUL_GELUP turns it into both things the backward needs from
    ONE read -- out = acc * gelu'(4 aux) and gelu(4 aux) rounded to a bf16 value (the fc2 weight gradient's operand: what the second product of the forward consumed) --
    against torch ([UNVENDORED] mci.py ConvFFN; the backward the reference would run through autograd, training/trainer.py:175)."""
    torch.manual_seed(Cc + M + 9)
    Hd = 4 * Cc
    x, res = bf(torch.randn(M, Cc)), bf(torch.randn(M, Cc))
    w1, w2 = bf(torch.randn(Hd, Cc) / math.sqrt(Cc)), bf(torch.randn(Cc, Hd) / math.sqrt(Hd))
    b1, b2, ls = torch.randn(Hd) * 0.1, torch.randn(Cc) * 0.1, torch.rand(Cc) * 0.3 + 0.05
    a = x.double() @ w1.double().t() + b1.double()
    xd, rd, wqd = dev_bf16(x), dev_bf16(res), dev_bf16(_pack_wq(w1, w2))
    b1d, b2d, lsd = dev_f32(b1), dev_f32(b2), dev_f32(ls)
    out, ref_out = (torch.full((M, Cc), float("nan"), dtype=torch.bfloat16, device=DEV) for _ in range(2))
    guard = 64
    sy = torch.full((M + guard, Hd), float("nan"), dtype=torch.float16, device=DEV)
    call(lib().fv_op_convffn32_stash(xd.data_ptr(), wqd.data_ptr(), b1d.data_ptr(), b2d.data_ptr(), lsd.data_ptr(), rd.data_ptr(), out.data_ptr(), M, Cc,
                                     sy.data_ptr(), stream()), "fv_op_convffn32_stash")
    call(lib().fv_op_convffn32(xd.data_ptr(), wqd.data_ptr(), b1d.data_ptr(), b2d.data_ptr(), lsd.data_ptr(), rd.data_ptr(), ref_out.data_ptr(), M, Cc, stream()),
         "fv_op_convffn32")
    torch.cuda.synchronize()
    assert torch.equal(out, ref_out)                                            # the stash changes nothing about the output
    assert torch.isnan(sy[M:].float()).all()                                    # rows past M are not written
    y = sy[:M].float().cpu()
    check_close(y, (a / 4).float(), rel=1e-3, amax=2e-3, what=f"stashed pre-activation / 4, C={Cc}")
    # FV_EPI_MUL_GELUP through the fp16 GEMM: dA = (G . W2s) * gelu'(a), h = gelu(a), a = 4 * stash_y
    Mg = min(M, 2048) // 8 * 8
    G = (torch.randn(Mg, Cc) * 0.05).half()
    W2s = (torch.randn(Hd, Cc) / math.sqrt(Cc)).half()                          # [N = 4C][K = C]: the transposed, layer-scaled fc2 copy of the backward
    Gd, Wd = G.to(DEV), W2s.to(DEV)
    dA = torch.full((Mg, Hd), float("nan"), dtype=torch.float16, device=DEV)
    hO = torch.full((Mg, Hd), float("nan"), dtype=torch.float16, device=DEV)
    call(lib().fv_op_gemm_f16_gelup(Gd.data_ptr(), Cc, Wd.data_ptr(), Mg, Hd, Cc, sy.data_ptr(), Hd, dA.data_ptr(), Hd, hO.data_ptr(), stream()), "fv_op_gemm_f16_gelup")
    torch.cuda.synchronize()
    av = (4.0 * y[:Mg].double()).requires_grad_(True)
    F.gelu(av).sum().backward()
    ref = (G.double() @ W2s.double().t()) * av.grad
    check_close(dA.float().cpu(), ref.float(), rel=2e-3, amax=4e-3, what=f"MUL_GELUP epilogue C={Cc}")
    h = hO.float().cpu()
    assert torch.equal(h, bf(h))                                                # bf16 values: the operand form the forward's second product consumed
    check_close(h, F.gelu(av.detach()).float(), rel=4e-3, amax=8e-3, what=f"MUL_GELUP's gelu(a), C={Cc}")
    check_close(h, F.gelu(a[:Mg]).float(), rel=5e-3, amax=2e-2, what=f"... against the unrounded pre-activation, C={Cc}")
    # without the second output the first is unchanged
    dA2 = torch.full((Mg, Hd), float("nan"), dtype=torch.float16, device=DEV)
    call(lib().fv_op_gemm_f16_gelup(Gd.data_ptr(), Cc, Wd.data_ptr(), Mg, Hd, Cc, sy.data_ptr(), Hd, dA2.data_ptr(), Hd, None, stream()), "fv_op_gemm_f16_gelup (no h)")
    torch.cuda.synchronize()
    assert torch.equal(dA, dA2)


@pytest.mark.parametrize("Cc,M", [(384, 4096), (384, 8192), (384, 16384 - 37), (192, 16384), (192, 65536), (96, 65536), (96, 131072), (384, 100)])
def test_fused_convffn32_hidden_ranges(Cc, M):
    """One to four observations give the fused ConvFFN 32 .. 128 row tiles for 256 CUs; the launcher then cuts the hidden units into 2 / 4 / 8 ranges
    (block = row tile x range, fp32 partial sums, one reduce pass with the one-launch epilogue's arithmetic).  Against the oracle like the one-launch
    kernel, against that kernel within one rounding of the output (the ranges change the fp32 summation order only), bit-repeatable, in place."""
    torch.manual_seed(Cc + M + 5)
    Hd = 4 * Cc
    x, res = bf(torch.randn(M, Cc)), bf(torch.randn(M, Cc))
    w1, w2 = bf(torch.randn(Hd, Cc) / math.sqrt(Cc)), bf(torch.randn(Cc, Hd) / math.sqrt(Hd))
    b1, b2, ls = torch.randn(Hd) * 0.1, torch.randn(Cc) * 0.1, torch.rand(Cc) * 0.3 + 0.05
    hid = bf(F.gelu(x @ w1.t() + b1))
    ref = res + ls * (hid @ w2.t() + b2)
    xd, rd, wqd = dev_bf16(x), dev_bf16(res), dev_bf16(_pack_wq(w1, w2))
    b1d, b2d, lsd = dev_f32(b1), dev_f32(b2), dev_f32(ls)
    part = torch.full((8 * M * Cc + 64,), float("nan"), dtype=torch.float32, device=DEV)
    outs = []
    for rep in range(2):
        out = torch.full((M, Cc), float("nan"), dtype=torch.bfloat16, device=DEV)
        call(lib().fv_op_convffn32_split(xd.data_ptr(), wqd.data_ptr(), b1d.data_ptr(), b2d.data_ptr(), lsd.data_ptr(), rd.data_ptr(), out.data_ptr(),
                                         M, Cc, part.data_ptr(), part.numel() * 4, stream()), "fv_op_convffn32_split")
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    check_close(outs[0].float().cpu(), ref, what=f"convffn32 hidden ranges C={Cc} M={M}")
    one = torch.empty_like(outs[0])
    call(lib().fv_op_convffn32(xd.data_ptr(), wqd.data_ptr(), b1d.data_ptr(), b2d.data_ptr(), lsd.data_ptr(), rd.data_ptr(), one.data_ptr(), M, Cc,
                               stream()), "fv_op_convffn32")
    torch.cuda.synchronize()
    d = (outs[0].float() - one.float()).abs()
    ulp = one.float().abs().clamp_min(2.0 ** -6) * 2.0 ** -7
    assert bool((d <= ulp).all()), f"more than one bf16 step from the one-launch kernel: {float((d / ulp).max()):.2f}"
    if M >= 4096:
        assert float((d > 0).float().mean()) < 0.05
    call(lib().fv_op_convffn32_split(xd.data_ptr(), wqd.data_ptr(), b1d.data_ptr(), b2d.data_ptr(), lsd.data_ptr(), rd.data_ptr(), rd.data_ptr(),
                                     M, Cc, part.data_ptr(), part.numel() * 4, stream()), "fv_op_convffn32_split in place")
    torch.cuda.synchronize()
    assert torch.equal(rd, outs[0])


@pytest.mark.parametrize("k,C,H,W,B", [(7, 64, 16, 32, 2), (7, 384, 64, 64, 2), (7, 96, 40, 72, 1), (3, 192, 24, 100, 2), (3, 32, 17, 33, 3), (7, 32, 19, 50, 2),
                                        (7, 1536, 16, 16, 2), (3, 96, 12, 16, 1)])   # the last two: maps narrower than a strip -> the VALU form
def test_dw_wgrad_matches_autograd(k, C, H, W, B):
    """Tap + bias gradients of the stride-1 depthwise convs of the tower's backward (round 6: the 7x7 on the matrix cores for C % 32 == 0, W >= 32, H >= 16 -- one
    v_mfma_f32_4x4x4_16B_f16 block per channel, x rows against shifted windows of a dy row, 32-column strips marching down the map -- the 3x3 and narrow maps on the VALU form)
    against torch.autograd over F.conv2d(groups = C) in fp64 ([UNVENDORED] mci.py RepMixer / ConvFFN depthwise convs; training/trainer.py:175 loss.backward())."""
    torch.manual_seed(k * 100 + C + W)
    x = bf(torch.randn(B, C, H, W))
    dy = (torch.randn(B, C, H, W) * 0.25).half()
    w = torch.zeros(C, 1, k, k, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(C, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double(), w, bias, padding=k // 2, groups=C)
    y.backward(dy.double())
    ref_w = w.grad.reshape(C, k * k).t().contiguous().float()      # tap-major [k*k][C]
    ref_b = bias.grad.float()
    xd = x.permute(0, 2, 3, 1).contiguous().bfloat16().to(DEV)
    dyd = dy.permute(0, 2, 3, 1).contiguous().to(DEV)
    nfl = max(512, B * ((W + 31) // 32)) * (k * k + 1) * C + 1024
    scr = torch.full((nfl,), float("nan"), dtype=torch.float32, device=DEV)
    outs = []
    for rep in range(2):
        dw = torch.full((k * k, C), float("nan"), dtype=torch.float32, device=DEV)
        db = torch.full((C,), float("nan"), dtype=torch.float32, device=DEV)
        call(lib().fv_op_dw_wgrad(xd.data_ptr(), dyd.data_ptr(), dw.data_ptr(), db.data_ptr(), scr.data_ptr(), nfl, B, H, W, C, k, 1, 1, stream()), "fv_op_dw_wgrad")
        torch.cuda.synchronize()
        outs.append((dw.cpu(), db.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])      # fixed-order partial sums: bit-repeatable
    check_close(outs[0][0], ref_w, rel=2e-5, amax=1e-4, what=f"dw{k}x{k} tap gradients C={C} {H}x{W}")
    check_close(outs[0][1], ref_b, rel=2e-5, amax=1e-4, what=f"dw{k}x{k} bias gradient C={C} {H}x{W}")


def _toeplitz(w, k):
    """depthwise weights (C,1,k,k) -> bf16 table [C/16][k][NM][16][4 i][4 kk] = w[ky][4m + kk - i] (fastvla_hip.h)."""
    Cc = w.shape[0]
    nm = (k + 6) // 4
    t = torch.zeros(Cc // 16, k, nm, 16, 4, 4)
    wv = w.view(Cc // 16, 16, k, k)
    for m in range(nm):
        for i in range(4):
            for kk in range(4):
                kx = 4 * m + kk - i
                if 0 <= kx < k:
                    t[:, :, m, :, i, kk] = wv[:, :, :, kx].permute(0, 2, 1)
    return t


@pytest.mark.parametrize("k,C,H,W,gelu", [(7, 32, 8, 32, 0), (7, 96, 40, 64, 0), (3, 64, 33, 47, 0), (7, 64, 19, 33, 1),
                                          (3, 192, 16, 32, 0), (7, 384, 24, 40, 0),
                                          # maps narrower than a 32-column strip (the last stage's 16 x 16, and a ragged 20): half-masked strips
                                          (7, 1536, 16, 16, 0), (7, 64, 16, 20, 1), (3, 96, 12, 16, 0)])
def test_dwconv_mfma(k, C, H, W, gelu):
    torch.manual_seed(k * 1000 + C + W)
    B = 2
    x = bf(torch.randn(B, C, H, W))
    w = bf(torch.randn(C, 1, k, k) / k)  # the table stores the weights in bf16
    b = torch.randn(C) * 0.1
    ref = F.conv2d(x, w, b, padding=k // 2, groups=C)
    if gelu:
        ref = F.gelu(ref)
    xd, td, bd = dev_bf16(x.permute(0, 2, 3, 1)), dev_bf16(_toeplitz(w, k)), dev_f32(b)
    y = torch.full((B, H, W, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    call(lib().fv_op_dwconv_mfma(xd.data_ptr(), td.data_ptr(), bd.data_ptr(), y.data_ptr(), B, H, W, C, k, gelu, stream()),
         "fv_op_dwconv_mfma")
    torch.cuda.synchronize()
    check_close(y.float().cpu().permute(0, 3, 1, 2), ref, what=f"dwconv mfma k{k} C{C} {H}x{W}")


def _toeplitz_s2(w):
    """[2C,1,7,7] -> the table of fv_op_dwconv_s2_mfma: [C/16][e][ky][m][16 ch][4 i][4 kk] = w[2(16g+ch)+e][ky][4m+kk-2i]."""
    C = w.shape[0] // 2
    wv = w.view(C // 16, 16, 2, 7, 7)                       # [g][ch][e][ky][kx]
    t = torch.zeros(C // 16, 2, 7, 4, 16, 4, 4)
    for m in range(4):
        for i in range(4):
            for kk in range(4):
                kx = 4 * m + kk - 2 * i
                if 0 <= kx < 7:
                    t[:, :, :, m, :, i, kk] = wv[:, :, :, :, kx].permute(0, 2, 3, 1)
    return t


@pytest.mark.parametrize("C,H,W,gelu", [(32, 16, 32, 1), (96, 40, 64, 1), (64, 18, 34, 0), (192, 32, 32, 1), (32, 2, 16, 0)])
def test_dwconv_s2_mfma(C, H, W, gelu):
    """PatchEmbed large-kernel conv (7x7, stride 2, groups = C, 2C outputs) on the matrix core against conv2d; ragged tiles
    (Ho % 8, Wo % 16 != 0) and a map smaller than one tile included."""
    torch.manual_seed(C + H + W)
    B = 2
    x = bf(torch.randn(B, C, H, W))
    w = bf(torch.randn(2 * C, 1, 7, 7) / 7)  # the table stores the weights in bf16
    b = torch.randn(2 * C) * 0.1
    ref = F.conv2d(x, w, b, stride=2, padding=3, groups=C)
    if gelu:
        ref = F.gelu(ref)
    xd, td, bd = dev_bf16(x.permute(0, 2, 3, 1)), dev_bf16(_toeplitz_s2(w)), dev_f32(b)
    y = torch.full((B, H // 2, W // 2, 2 * C), float("nan"), dtype=torch.bfloat16, device=DEV)
    call(lib().fv_op_dwconv_s2_mfma(xd.data_ptr(), td.data_ptr(), bd.data_ptr(), y.data_ptr(), B, H, W, C, gelu, stream()),
         "fv_op_dwconv_s2_mfma")
    torch.cuda.synchronize()
    check_close(y.float().cpu().permute(0, 3, 1, 2), ref, what=f"dwconv s2 mfma C{C} {H}x{W}")


@pytest.mark.parametrize("S,C0", [(64, 32), (96, 96), (40, 16)])
def test_stem_mfma(S, C0):
    torch.manual_seed(S + C0)
    B = 2
    x = bf(torch.rand(B, 3, S, S))
    w = bf(torch.randn(C0, 3, 3, 3) / 5)  # the MFMA image stores the stem weights in bf16
    b = torch.randn(C0) * 0.1
    ref = F.gelu(F.conv2d(x, w, b, stride=2, padding=1))
    pix = torch.zeros(B, S, S, 4, dtype=torch.bfloat16, device=DEV)
    pix[..., :3] = dev_bf16(x.permute(0, 2, 3, 1))
    wp = torch.zeros(C0, 64)
    for ks in range(2):
        for g in range(4):
            for e in range(8):
                ky, kx, ch = 2 * ks + (g >> 1), 2 * (g & 1) + (e >> 2), e & 3
                if ky < 3 and kx < 3 and ch < 3:
                    wp[:, ks * 32 + g * 8 + e] = w[:, ch, ky, kx]
    wd, bd = dev_bf16(wp), dev_f32(b)
    y = torch.full((B, S // 2, S // 2, C0), float("nan"), dtype=torch.bfloat16, device=DEV)
    call(lib().fv_op_stem_mfma(pix.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, S, C0, stream()), "fv_op_stem_mfma")
    torch.cuda.synchronize()
    check_close(y.float().cpu().permute(0, 3, 1, 2), ref, what=f"stem mfma S={S} C0={C0}")


@pytest.mark.parametrize("S,B", [(64, 2), (136, 1), (1024, 1)])
def test_stem_fused(S, B):
    """conv 3x3 s2 + GELU -> (bf16) -> depthwise 3x3 s2 + GELU in one kernel vs the two convolutions on the host."""
    torch.manual_seed(S)
    C0 = 96
    x = bf(torch.rand(B, 3, S, S))
    w1 = bf(torch.randn(C0, 3, 3, 3) / 5)
    b1 = torch.randn(C0) * 0.1
    w2 = torch.randn(C0, 1, 3, 3) / 3
    b2 = torch.randn(C0) * 0.1
    mid = bf(F.gelu(F.conv2d(x, w1, b1, stride=2, padding=1)))          # the kernel rounds the half-resolution map to bf16 too
    ref = F.gelu(F.conv2d(mid, w2, b2, stride=2, padding=1, groups=C0))
    pix = torch.zeros(B, S, S, 4, dtype=torch.bfloat16, device=DEV)
    pix[..., :3] = dev_bf16(x.permute(0, 2, 3, 1))
    wp = torch.zeros(C0, 64)
    for ks in range(2):
        for g in range(4):
            for e in range(8):
                ky, kx, ch = 2 * ks + (g >> 1), 2 * (g & 1) + (e >> 2), e & 3
                if ky < 3 and kx < 3 and ch < 3:
                    wp[:, ks * 32 + g * 8 + e] = w1[:, ch, ky, kx]
    wd, b1d, b2d = dev_bf16(wp), dev_f32(b1), dev_f32(b2)
    w2d = dev_f32(w2.view(C0, 9).t().contiguous())                       # tap-major [9][C0]
    y = torch.full((B, S // 4, S // 4, C0), float("nan"), dtype=torch.bfloat16, device=DEV)
    call(lib().fv_op_stem_fused(pix.data_ptr(), wd.data_ptr(), b1d.data_ptr(), w2d.data_ptr(), b2d.data_ptr(), y.data_ptr(), B, S, C0,
                                stream()), "fv_op_stem_fused")
    torch.cuda.synchronize()
    check_close(y.float().cpu().permute(0, 3, 1, 2), ref, what=f"fused stem S={S}")


@pytest.mark.parametrize("C,H,W", [(32, 16, 32), (96, 40, 64), (64, 19, 37), (192, 24, 96), (128, 33, 50), (384, 16, 32), (64, 72, 64)])
def test_dwconv_pair(C, H, W):
    """x' = dw3x3(x) and t = dw7x7(x') from one marching kernel vs the two convolutions on the host (x' rounded to bf16
    in between, as both the kernel's LDS ring and the unfused pair's HBM round trip do).  C % 64 == 0 takes the 16-column x
    64-channel geometry (full cache lines), the rest the 32 x 32 one; ragged H and W exercise both sets of ring slots."""
    torch.manual_seed(C + H + W)
    B = 2
    x = bf(torch.randn(B, C, H, W))
    w3, w7 = bf(torch.randn(C, 1, 3, 3) / 3), bf(torch.randn(C, 1, 7, 7) / 7)
    b3, b7 = torch.randn(C) * 0.1, torch.randn(C) * 0.1
    ref1 = bf(F.conv2d(x, w3, b3, padding=1, groups=C))
    ref2 = F.conv2d(ref1, w7, b7, padding=3, groups=C)
    xd = dev_bf16(x.permute(0, 2, 3, 1))
    t3, t7 = dev_bf16(_toeplitz(w3, 3)), dev_bf16(_toeplitz(w7, 7))
    b3d, b7d = dev_f32(b3), dev_f32(b7)
    y1 = torch.full((B, H, W, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    y2 = torch.full((B, H, W, C), float("nan"), dtype=torch.bfloat16, device=DEV)
    call(lib().fv_op_dwconv_pair(xd.data_ptr(), t3.data_ptr(), b3d.data_ptr(), t7.data_ptr(), b7d.data_ptr(), y1.data_ptr(), y2.data_ptr(),
                                 B, H, W, C, stream()), "fv_op_dwconv_pair")
    torch.cuda.synchronize()
    check_close(y1.float().cpu().permute(0, 3, 1, 2), ref1, what=f"dw pair 3x3 C{C} {H}x{W}")
    check_close(y2.float().cpu().permute(0, 3, 1, 2), ref2, what=f"dw pair 7x7 C{C} {H}x{W}")


@pytest.mark.parametrize("C,H", [(192, 128), (384, 64)])
def test_dwconv_pair_tower_shape_repeats(C, H):
    """The tower's own shapes, four launches: each must match the host convolutions and the launches must agree bit for bit.  (A
    write-after-write race between two prologue ring writes of different waves once corrupted x row 0 of about one block in 10^4:
    only a shape with thousands of blocks, repeated, sees that.)"""
    torch.manual_seed(C)
    B = 3
    x = bf(torch.randn(B, C, H, H))
    w3, w7 = bf(torch.randn(C, 1, 3, 3) / 3), bf(torch.randn(C, 1, 7, 7) / 7)
    b3, b7 = torch.randn(C) * 0.1, torch.randn(C) * 0.1
    ref1 = bf(F.conv2d(x, w3, b3, padding=1, groups=C))
    ref2 = F.conv2d(ref1, w7, b7, padding=3, groups=C)
    xd = dev_bf16(x.permute(0, 2, 3, 1))
    t3, t7 = dev_bf16(_toeplitz(w3, 3)), dev_bf16(_toeplitz(w7, 7))
    b3d, b7d = dev_f32(b3), dev_f32(b7)
    first = None
    for rep in range(4):
        y1 = torch.full((B, H, H, C), float("nan"), dtype=torch.bfloat16, device=DEV)
        y2 = torch.full((B, H, H, C), float("nan"), dtype=torch.bfloat16, device=DEV)
        call(lib().fv_op_dwconv_pair(xd.data_ptr(), t3.data_ptr(), b3d.data_ptr(), t7.data_ptr(), b7d.data_ptr(), y1.data_ptr(), y2.data_ptr(),
                                     B, H, H, C, stream()), "fv_op_dwconv_pair")
        torch.cuda.synchronize()
        if first is None:
            first = (y1, y2)
            check_close(y1.float().cpu().permute(0, 3, 1, 2), ref1, what=f"dw pair 3x3 C{C} {H}x{H}")
            check_close(y2.float().cpu().permute(0, 3, 1, 2), ref2, what=f"dw pair 7x7 C{C} {H}x{H}")
            assert float((y2.float().cpu().permute(0, 3, 1, 2) - ref2).abs().max()) < 0.1   # a corrupted row is O(1), rounding is 0.03
        else:
            assert torch.equal(y1, first[0]) and torch.equal(y2, first[1]), f"launch {rep} differs from launch 0"


@pytest.mark.parametrize("C,H,W", [(192, 128, 128), (384, 64, 64), (96, 256, 256), (64, 19, 37), (192, 40, 96)])
def test_dwconv_pair_row_segments_equal_the_uncut_march(C, H, W):
    """One observation gives the pair 24 strips for 256 CUs, so the launcher cuts the march into row segments (nstrips < 256); a batch of 24 is
    walked uncut.  The same image must come out bit for bit the same either way (every output row is computed by the same arithmetic in exactly
    one block), at the tower's three shapes and at ragged ones."""
    torch.manual_seed(7 * C + H)
    Bbig = max(2, -(-256 // (-(-W // 16) * max(C // 64, 1))) + 1) if C % 64 == 0 else max(2, -(-256 // (-(-W // 32) * (C // 32))) + 1)
    x1 = bf(torch.randn(1, C, H, W))
    xb = torch.cat([x1, bf(torch.randn(Bbig - 1, C, H, W))], 0)
    w3, w7 = bf(torch.randn(C, 1, 3, 3) / 3), bf(torch.randn(C, 1, 7, 7) / 7)
    t3, t7 = dev_bf16(_toeplitz(w3, 3)), dev_bf16(_toeplitz(w7, 7))
    b3d, b7d = dev_f32(torch.randn(C) * 0.1), dev_f32(torch.randn(C) * 0.1)
    outs = []
    for x in (x1, xb):
        B = x.shape[0]
        xd = dev_bf16(x.permute(0, 2, 3, 1))
        y1 = torch.full((B, H, W, C), float("nan"), dtype=torch.bfloat16, device=DEV)
        y2 = torch.full((B, H, W, C), float("nan"), dtype=torch.bfloat16, device=DEV)
        call(lib().fv_op_dwconv_pair(xd.data_ptr(), t3.data_ptr(), b3d.data_ptr(), t7.data_ptr(), b7d.data_ptr(), y1.data_ptr(), y2.data_ptr(),
                                     B, H, W, C, stream()), "fv_op_dwconv_pair")
        torch.cuda.synchronize()
        outs.append((y1[0].clone(), y2[0].clone()))
    assert not torch.isnan(outs[0][0].float()).any() and not torch.isnan(outs[0][1].float()).any()
    assert torch.equal(outs[0][0], outs[1][0]), "x' of the segmented march differs from the uncut one"
    assert torch.equal(outs[0][1], outs[1][1]), "t of the segmented march differs from the uncut one"


@pytest.mark.parametrize("M,C", [(128 * 3, 192), (128 * 800, 96), (128 * 300, 192)])
def test_gemm_pointwise_square(M, C):
    """K = N = C in {96, 192}, M % 128 == 0, contiguous rows: the persistent pointwise-conv kernel (weight resident in LDS,
    rows streamed straight into MFMA operands).  The larger cases have more row tiles than the grid has blocks."""
    torch.manual_seed(M + C)
    A, W, b = bf(torch.randn(M, C)), bf(torch.randn(C, C) / math.sqrt(C)), torch.randn(C) * 0.1
    W[5, 7] += 1.0                                     # asymmetric: a transposed weight cannot pass
    W = bf(W)
    ref = A @ W.t() + b
    check_close(_gemm(A, W, _lib.EPI_BIAS, bias=b), ref, what=f"pwconv bias {M}x{C}")
    check_close(_gemm(A, W, _lib.EPI_BIAS_GELU, bias=b), F.gelu(ref), what=f"pwconv gelu {M}x{C}")


@pytest.mark.parametrize("M,N,K", [(16384, 2048, 192), (8192, 4096, 448), (4096, 5632, 192), (1024, 9216, 192), (512, 20480, 128),
                                   # ragged edge tiles under the bf16 epilogues: a half-filled last column tile (N = 896), a last row tile of 8 rows
                                   (8192, 896, 256), (8200, 1024, 192), (32768, 384, 384)])
def test_gemm_256_tile_variant(M, N, K):
    """Shapes the 256 x 256 LDS-DMA kernel takes (M, N multiples of 256, K of 64, >= 320 tiles; from 128 tiles up when M <= 2048): all
    three of its epilogues, plus the asymmetric-operand check that a swapped row/column map cannot pass.  The third shape has 352 tiles: a
    ragged second round of the persistent loop; the last two (4 x 36 and 2 x 80 tiles) are walked column-major (few rows, many weight
    columns: the decoder at small batch)."""
    torch.manual_seed(M + N + K)
    A, W = bf(torch.randn(M, K)), bf(torch.randn(N, K) / math.sqrt(K))
    b, ls, res = torch.randn(N) * 0.1, torch.rand(N) * 0.3 + 0.05, bf(torch.randn(M, N))
    ref = A @ W.t() + b                       # fp32 on the host keeps the reference to seconds at this size
    check_close(_gemm(A, W, _lib.EPI_BIAS, bias=b), ref, what=f"gemm256 bias {M}x{N}x{K}")
    check_close(_gemm(A, W, _lib.EPI_BIAS_GELU, bias=b), F.gelu(ref), what=f"gemm256 gelu {M}x{N}x{K}")
    check_close(_gemm(A, W, _lib.EPI_LS_RES, bias=b, scale=ls, res=res), res + ls * ref, what=f"gemm256 ls_res {M}x{N}x{K}")
    check_close(_gemm(A, W, _lib.EPI_BIAS, bias=b, lda=K + 64), ref, what="gemm256 strided A")


@pytest.mark.parametrize("M,I,K", [(300, 256, 128), (4096, 5120, 192)])
def test_gemm_ksplit_swiglu_split(M, I, K):
    """The parity-mode decoder's gate/up GEMM: split-bf16 activations [hi | lo] against bf16 weights in one launch, SwiGLU in
    the epilogue, output again split [hi | lo].  The second shape is taken by the 256 x 256 kernel, the first by the 128."""
    torch.manual_seed(M + I + K)
    A = torch.randn(M, K)
    Ah = bf(A)
    Al = bf(A - Ah)
    G, U = bf(torch.randn(I, K) / math.sqrt(K)), bf(torch.randn(I, K) / math.sqrt(K))
    Wi = torch.empty(2 * I, K)
    j = torch.arange(I)
    Wi[(j // 8) * 16 + j % 8] = G
    Wi[(j // 8) * 16 + 8 + j % 8] = U
    a16 = (Ah + Al).double()                                 # what the kernel multiplies: 16 significant bits
    ref = (F.silu(a16 @ G.double().t()) * (a16 @ U.double().t())).float()
    a = torch.cat([dev_bf16(Ah), dev_bf16(Al)], dim=1).contiguous()
    w = dev_bf16(Wi)
    out = torch.full((M, 2 * I), float("nan"), dtype=torch.bfloat16, device=DEV)
    call(lib().fv_op_gemm_ksplit(a.data_ptr(), 2 * K, w.data_ptr(), M, 2 * I, K, None, None, 0, out.data_ptr(), 2 * I,
                                 _lib.EPI_SWIGLU_SPLIT, stream()), "fv_op_gemm_ksplit")
    torch.cuda.synchronize()
    o = out.float().cpu()
    check_close(o[:, :I] + o[:, I:], ref, rel=2e-5, amax=2e-4, what=f"ksplit swiglu split {M}x{I}x{K}")
    check_close(o[:, :I], ref, what="hi half alone is the bf16 rounding of the result")


def test_gemm_glds_128_tile_fp32_epilogues():
    """The decoder's projections at M = 4096 (N = 896: too few 256-tiles) go to the 128 x 128 LDS-DMA variant: fp32 residual
    stream, fp32 output with bias, and the split-bf16 form of both."""
    torch.manual_seed(11)
    M, N, K = 4096, 896, 576
    A = torch.randn(M, K)
    Ah = bf(A)
    Al = bf(A - Ah)
    W, res, b = bf(torch.randn(N, K) * 0.05), torch.randn(M, N), torch.randn(N)
    ref = (res.double() + Ah.double() @ W.double().t()).float()
    check_close(_gemm(Ah, W, _lib.EPI_RES_F32, res=res, out_f32=True), ref, rel=2e-5, amax=2e-5, what="glds128 res_f32")
    ref = (Ah.double() @ W.double().t() + b.double()).float()
    check_close(_gemm(Ah, W, _lib.EPI_F32, bias=b, out_f32=True), ref, rel=2e-5, amax=2e-5, what="glds128 f32 out")
    a = torch.cat([dev_bf16(Ah), dev_bf16(Al)], dim=1).contiguous()
    w, r = dev_bf16(W), dev_f32(res)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    call(lib().fv_op_gemm_ksplit(a.data_ptr(), 2 * K, w.data_ptr(), M, N, K, None, r.data_ptr(), N, out.data_ptr(), N, _lib.EPI_RES_F32,
                                 stream()), "fv_op_gemm_ksplit")
    torch.cuda.synchronize()
    ref = (res.double() + (Ah + Al).double() @ W.double().t()).float()
    check_close(out.cpu(), ref, rel=2e-5, amax=2e-5, what="glds128 ksplit res_f32")


def test_gemm_splitk_down_projection_shape():
    """M = 4096, N = 896: 64 output tiles of 256 x 256 (the last column tile padded) cut into K ranges, one unit per CU, summed
    by the reduce kernel -- plain and split-bf16 operands, against the same GEMM without the scratch buffer."""
    torch.manual_seed(12)
    M, N, K = 4096, 896, 2432
    A = torch.randn(M, K)
    Ah = bf(A)
    Al = bf(A - Ah)
    W, res = bf(torch.randn(N, K) * 0.03), torch.randn(M, N)
    w, r = dev_bf16(W), dev_f32(res)
    ws = torch.empty(8 * M * 1024, dtype=torch.float32, device=DEV)
    for ksplit, a, ref in ((0, dev_bf16(Ah), res.double() + Ah.double() @ W.double().t()),
                           (1, torch.cat([dev_bf16(Ah), dev_bf16(Al)], dim=1).contiguous(), res.double() + (Ah + Al).double() @ W.double().t())):
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
        call(lib().fv_op_gemm_splitk(a.data_ptr(), a.shape[1], w.data_ptr(), M, N, K, None, r.data_ptr(), N, out.data_ptr(), N,
                                     _lib.EPI_RES_F32, ksplit, ws.data_ptr(), ws.numel() * 4, stream()), "fv_op_gemm_splitk")
        torch.cuda.synchronize()
        check_close(out.cpu(), ref.float(), rel=2e-5, amax=2e-5, what=f"split-K res_f32 ksplit={ksplit}")
    # in place on the residual stream, as the decoder calls it
    x = r.clone()
    call(lib().fv_op_gemm_splitk(a.data_ptr(), a.shape[1], w.data_ptr(), M, N, K, None, x.data_ptr(), N, x.data_ptr(), N,
                                 _lib.EPI_RES_F32, 1, ws.data_ptr(), ws.numel() * 4, stream()), "fv_op_gemm_splitk in place")
    torch.cuda.synchronize()
    assert torch.equal(x, out)


@pytest.mark.parametrize("ws_splits", [8, 3])
def test_gemm_splitk_few_rows_many_splits(ws_splits):
    """The 7B decoder at C5's rank shape (M = 512): 2 x 4 tiles of 256 x 256 over a long K -- up to eight K ranges per tile, and as
    many as the scratch buffer holds when it is smaller (here: room for 8 or for 3), column-major tile walk (M <= 2048), bias in the
    reducer; against fp64."""
    torch.manual_seed(21 + ws_splits)
    M, N, K = 512, 1024, 8192
    A, W, b = bf(torch.randn(M, K)), bf(torch.randn(N, K) * 0.02), torch.randn(N)
    ref = (A.double() @ W.double().t() + b.double()).float()
    a, w, bd = dev_bf16(A), dev_bf16(W), dev_f32(b)
    ws = torch.empty(ws_splits * M * 1024, dtype=torch.float32, device=DEV)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    call(lib().fv_op_gemm_splitk(a.data_ptr(), K, w.data_ptr(), M, N, K, bd.data_ptr(), None, 0, out.data_ptr(), N,
                                 _lib.EPI_F32, 0, ws.data_ptr(), ws.numel() * 4, stream()), "fv_op_gemm_splitk few rows")
    torch.cuda.synchronize()
    check_close(out.cpu(), ref, rel=2e-5, amax=2e-4, what=f"split-K few rows (scratch for {ws_splits})")


@pytest.mark.parametrize("M,N,K,epi", [(1024, 37888, 1024, "swiglu"), (4096, 9728, 1024, "swiglu"), (1024, 18944 + 256 * 10, 1024, "res_f32")])
def test_gemm_tail_round_as_k_ranges(M, N, K, epi):
    """2.3 rounds of 256-tiles cost three rounds of time: with scratch, launch_gemm runs the full rounds' column panels as they are and the remaining panels
    as a K-range problem of their own (7B gate/up at M = 1024: 592 tiles; 0.5B gate/up at M = 4096: 608; an fp32-residual shape with bias).  Every column
    against the same call without scratch (fp32 summation order only), a sample of columns on both sides of the cut against fp64, twice for repeatability."""
    torch.manual_seed(M + N + K)
    xh = bf(torch.randn(M, K))
    xl = bf(torch.randn(M, K) * 2.0 ** -9)
    a = torch.cat([dev_bf16(xh), dev_bf16(xl)], dim=1).contiguous()
    W = bf(torch.randn(N, K) / math.sqrt(K))
    w = dev_bf16(W)
    ws = torch.full((8 * M * 4608,), float("nan"), dtype=torch.float32, device=DEV)
    a16 = (xh + xl).double()
    if epi == "swiglu":
        outs = []
        for sc in (True, True, False):
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            call(lib().fv_op_gemm_splitk(a.data_ptr(), 2 * K, w.data_ptr(), M, N, K, None, None, 0, out.data_ptr(), N, _lib.EPI_SWIGLU_SPLIT, 1,
                                         ws.data_ptr() if sc else None, ws.numel() * 4 if sc else 0, stream()), "fv_op_gemm_splitk")
            torch.cuda.synchronize()
            outs.append(out)
        assert torch.equal(outs[0], outs[1]) and not torch.isnan(outs[0].float()).any()
        o, o1 = outs[0].float().cpu(), outs[2].float().cpu()
        I = N // 2
        check_close(o[:, :I] + o[:, I:], o1[:, :I] + o1[:, I:], rel=2e-6, amax=2e-5, what="tail form against the one-launch form")
        cols = torch.cat([torch.arange(0, 64), torch.arange(I - 64, I), torch.arange(I // 2 - 32, I // 2 + 32)])   # outputs at both ends and in the middle
        rows16 = ((cols // 8) * 16)[:, None]
        G = W[(rows16 + (cols % 8)[:, None]).flatten()].double()
        U = W[(rows16 + 8 + (cols % 8)[:, None]).flatten()].double()
        ref = (F.silu(a16 @ G.t()) * (a16 @ U.t())).float()
        check_close((o[:, :I] + o[:, I:])[:, cols], ref, rel=2e-5, amax=2e-4, what=f"tail form {M}x{N}x{K} against fp64")
    else:
        b, res = torch.randn(N), torch.randn(M, N)
        bd = dev_f32(b)
        xs = []
        for sc in (True, True, False):
            x = dev_f32(res).clone()
            call(lib().fv_op_gemm_splitk(a.data_ptr(), 2 * K, w.data_ptr(), M, N, K, bd.data_ptr(), x.data_ptr(), N, x.data_ptr(), N, _lib.EPI_RES_F32, 1,
                                         ws.data_ptr() if sc else None, ws.numel() * 4 if sc else 0, stream()), "fv_op_gemm_splitk")
            torch.cuda.synchronize()
            xs.append(x)
        assert torch.equal(xs[0], xs[1])
        check_close(xs[0].cpu(), xs[2].cpu(), rel=2e-6, amax=2e-5, what="tail form against the one-launch form")
        cols = torch.cat([torch.arange(0, 128), torch.arange(N - 128, N)])
        ref = (res[:, cols].double() + b[cols].double() + a16 @ W[cols].double().t()).float()
        check_close(xs[0].cpu()[:, cols], ref, rel=2e-5, amax=2e-5, what=f"tail form {M}x{N}x{K} res_f32 against fp64")


@pytest.mark.parametrize("M", [64, 128, 200, 256])
def test_gemm_few_rows_k_ranges(M):
    """The control loop's decoder (one to four observations x 64 tokens): 64-row tiles cut into K ranges until the chip is covered, partial sums finished by
    the reducers -- qkv (bias, fp32), down (fp32 residual, 152 K-tiles of the split-bf16 operand) and gate/up (SwiGLU + hi | lo split; at M = 256 its 304
    tiles keep the one-launch form).  Against fp64, against the same call without scratch (fp32 summation order only), and twice for bit-repeatability."""
    torch.manual_seed(900 + M)
    Hd, I, QK = 896, 4864, 1152
    ws = torch.full((40 * M * 1280,), float("nan"), dtype=torch.float32, device=DEV)

    def split(x):
        h = bf(x)
        return h, bf(x - h)

    def run(a, lda, w, N, K, bias, res, out, epi, scratch):
        call(lib().fv_op_gemm_splitk(a.data_ptr(), lda, w.data_ptr(), M, N, K, None if bias is None else bias.data_ptr(), None if res is None else res.data_ptr(),
                                     0 if res is None else N, out.data_ptr(), out.shape[1], epi, 1, ws.data_ptr() if scratch else None, ws.numel() * 4 if scratch else 0,
                                     stream()), "fv_op_gemm_splitk")
        torch.cuda.synchronize()
        return out

    # qkv: bias, fp32 out
    xh, xl = split(torch.randn(M, Hd))
    Wq, bq = bf(torch.randn(QK, Hd) * 0.03), torch.randn(QK)
    a = torch.cat([dev_bf16(xh), dev_bf16(xl)], dim=1).contiguous()
    outs = [run(a, 2 * Hd, dev_bf16(Wq), QK, Hd, dev_f32(bq), None, torch.full((M, QK), float("nan"), device=DEV), _lib.EPI_F32, sc) for sc in (True, True, False)]
    assert torch.equal(outs[0], outs[1])
    check_close(outs[0].cpu(), ((xh + xl).double() @ Wq.double().t() + bq.double()).float(), rel=2e-5, amax=2e-5, what=f"qkv K ranges M={M}")
    check_close(outs[0].cpu(), outs[2].cpu(), rel=2e-6, amax=2e-5, what="against the one-launch form")
    # down: residual stream in place, K = 2 x 4864
    ch, cl = split(torch.randn(M, I))
    Wd, res = bf(torch.randn(Hd, I) * 0.02), torch.randn(M, Hd)
    a = torch.cat([dev_bf16(ch), dev_bf16(cl)], dim=1).contiguous()
    ref = (res.double() + (ch + cl).double() @ Wd.double().t()).float()
    xs = []
    for sc in (True, True, False):
        x = dev_f32(res).clone()
        run(a, 2 * I, dev_bf16(Wd), Hd, I, None, x, x, _lib.EPI_RES_F32, sc)
        xs.append(x)
    assert torch.equal(xs[0], xs[1])
    check_close(xs[0].cpu(), ref, rel=2e-5, amax=2e-5, what=f"down K ranges M={M}")
    check_close(xs[0].cpu(), xs[2].cpu(), rel=2e-6, amax=2e-5, what="against the one-launch form")
    # gate / up: SwiGLU on the summed ranges, hi | lo out
    G, U = bf(torch.randn(I, Hd) / math.sqrt(Hd)), bf(torch.randn(I, Hd) / math.sqrt(Hd))
    Wi = torch.empty(2 * I, Hd)
    j = torch.arange(I)
    Wi[(j // 8) * 16 + j % 8] = G
    Wi[(j // 8) * 16 + 8 + j % 8] = U
    a16 = (xh + xl).double()
    ref = (F.silu(a16 @ G.double().t()) * (a16 @ U.double().t())).float()
    a = torch.cat([dev_bf16(xh), dev_bf16(xl)], dim=1).contiguous()
    outs = [run(a, 2 * Hd, dev_bf16(Wi), 2 * I, Hd, None, None, torch.full((M, 2 * I), float("nan"), dtype=torch.bfloat16, device=DEV), _lib.EPI_SWIGLU_SPLIT, sc)
            for sc in (True, True, False)]
    assert torch.equal(outs[0], outs[1])
    o = outs[0].float().cpu()
    check_close(o[:, :I] + o[:, I:], ref, rel=2e-5, amax=2e-4, what=f"gate/up K ranges M={M}")
    o1 = outs[2].float().cpu()
    check_close(o[:, :I] + o[:, I:], o1[:, :I] + o1[:, I:], rel=2e-6, amax=2e-5, what="against the one-launch form")


@pytest.mark.parametrize("M,N,K,ksplit,ws", [(8200, 896, 256, 0, False), (10240, 1152, 192, 1, False), (9728, 904, 128, 1, False),
                                              (896, 4864, 2048, 1, True), (900, 904, 4096, 0, True), (1152, 896, 2048, 1, True),
                                              # 11 x 12 tiles, ragged on both edges: the grouped tile walk (groups of 4 tile rows) with a last group of 3
                                              (2660, 2920, 128, 0, False), (2660, 2920, 1024, 1, True)])
def test_gemm_256_tile_ragged_edges_fp32_epilogues(M, N, K, ksplit, ws):
    """Round 4: fp32-epilogue problems whose M or N is not a multiple of 256 (the decoder's N = 896 / 1152 projections at a training
    batch, every dgrad / wgrad of the unfrozen path: M = 896, N = 4864 ...) on the 256-tile LDS-DMA kernel with ragged edge tiles --
    plain (>= 128 tiles) and cut along K (few tiles, long K, scratch given) -- F32 with bias and RES_F32 in place, against fp64.
    Rows / columns past the edge must not be written (the outputs are embedded in NaN-guarded buffers)."""
    torch.manual_seed(M + N + K)
    Kt = (2 if ksplit else 1) * K
    A = bf(torch.randn(M, Kt) * (0.5 if not ksplit else 1.0))
    if ksplit:
        A[:, K:] = bf(A[:, K:] * 2 ** -8)     # a lo half: hi + lo is the operand
    W, b, res = bf(torch.randn(N, K) / math.sqrt(K)), torch.randn(N), torch.randn(M, N)
    Aeff = (A[:, :K] + A[:, K:]) if ksplit else A
    ref = Aeff.double() @ W.double().t()
    a, w, bd = dev_bf16(A), dev_bf16(W), dev_f32(b)
    wsb = torch.empty(8 * 1024 * 1024, dtype=torch.float32, device=DEV) if ws else None
    guard = 8
    out = torch.full((M + guard, N + guard), float("nan"), dtype=torch.float32, device=DEV)
    call(lib().fv_op_gemm_splitk(a.data_ptr(), Kt, w.data_ptr(), M, N, K, bd.data_ptr(), None, 0, out.data_ptr(), N + guard, _lib.EPI_F32, ksplit,
                                 wsb.data_ptr() if ws else None, wsb.numel() * 4 if ws else 0, stream()), "ragged F32")
    torch.cuda.synchronize()
    o = out.cpu()
    check_close(o[:M, :N], (ref + b.double()).float(), rel=2e-5, amax=2e-4, what=f"ragged gemm F32 {M}x{N}x{K}")
    assert torch.isnan(o[M:]).all() and torch.isnan(o[:, N:]).all(), "wrote past the edge"
    x = torch.full((M + guard, N + guard), float("nan"), dtype=torch.float32, device=DEV)
    x[:M, :N] = res.to(DEV)
    call(lib().fv_op_gemm_splitk(a.data_ptr(), Kt, w.data_ptr(), M, N, K, None, x.data_ptr(), N + guard, x.data_ptr(), N + guard, _lib.EPI_RES_F32, ksplit,
                                 wsb.data_ptr() if ws else None, wsb.numel() * 4 if ws else 0, stream()), "ragged RES_F32 in place")
    torch.cuda.synchronize()
    o = x.cpu()
    check_close(o[:M, :N], (ref + res.double()).float(), rel=2e-5, amax=2e-4, what=f"ragged gemm RES_F32 {M}x{N}x{K}")
    assert torch.isnan(o[M:]).all() and torch.isnan(o[:, N:]).all(), "wrote past the edge"


@pytest.mark.parametrize("M,N,K,f16,ws", [(896, 4864, 1024, 1, True), (2304, 904, 512, 0, False), (9728, 896, 2048, 1, True), (520, 264, 192, 1, False)])
def test_gemm_tn_row_major_operands(M, N, K, f16, ws):
    """Round 4: the TN instance of the 256-tile kernel -- out[m][n] = sum_k A[k][m] W[k][n] with both operands row-major over the contraction
    (a weight gradient's dY and X as they are produced; LDS image in [k/8][n/16] blocks, fragments by ds_read_b64_tr_b16) -- against fp64, bf16
    and fp16 operands, ragged edges, with and without K ranges; embedded in NaN guards, operands with padded row strides."""
    torch.manual_seed(M + N + K + f16)
    lda, ldw = M + 8, N + 16
    A, W, b = torch.randn(K, M) * 0.5, torch.randn(K, N) / math.sqrt(K), torch.randn(N)
    cast = (lambda t: t.half()) if f16 else (lambda t: t.bfloat16())
    Ad, Wd = torch.zeros(K, lda, dtype=torch.float16 if f16 else torch.bfloat16, device=DEV), torch.zeros(K, ldw, dtype=torch.float16 if f16 else torch.bfloat16, device=DEV)
    Ad[:, :M], Wd[:, :N] = cast(A).to(DEV), cast(W).to(DEV)
    ref = (cast(A).double().t() @ cast(W).double() + b.double()).float()
    bd = dev_f32(b)
    wsb = torch.empty(8 * 1024 * 1024, dtype=torch.float32, device=DEV) if ws else None
    guard = 8
    out = torch.full((M + guard, N + guard), float("nan"), dtype=torch.float32, device=DEV)
    call(lib().fv_op_gemm_tn(Ad.data_ptr(), lda, Wd.data_ptr(), ldw, M, N, K, f16, bd.data_ptr(), out.data_ptr(), N + guard,
                             wsb.data_ptr() if ws else None, wsb.numel() * 4 if ws else 0, stream()), "fv_op_gemm_tn")
    torch.cuda.synchronize()
    o = out.cpu()
    check_close(o[:M, :N], ref, rel=2e-5, amax=2e-4, what=f"TN gemm {M}x{N}x{K} f16={f16}")
    assert torch.isnan(o[M:]).all() and torch.isnan(o[:, N:]).all(), "wrote past the edge"


# ------------------------------------------------------------------------------------------------ round 4: "hi + lo8" operands
def _e4m3(t):
    return t.to(torch.float8_e4m3fn).float()


@pytest.mark.parametrize("M,N,K,epi,ws", [(100, 256, 896, "f32", False), (4096, 1152, 896, "f32", False), (512, 896, 4864, "res", True),
                                            (8200, 896, 1152, "res", False), (4096, 2560, 256, "swiglu", False), (96, 128, 896, "swiglu", False),
                                            (256, 17920, 1536, "swiglu", True)])   # the 1.5B gate/up at 256 rows WITH scratch: the K-range forms must not take it
def test_gemm_hi_lo8_operands(M, N, K, epi, ws):
    """llm_precision = 5's projections (round 4): A = bf16 hi + ONE fp8 e4m3 byte of remainder (x 2^8) per element, the lo product on
    v_mfma_scale_f32_16x16x128_f8f6f4 against the weights' fp8 copy (x 2^6), through the register-staged kernel, the 256-tile kernel
    (plain, ragged, split-K) and the SwiGLU-split epilogue (whose output leaves in the same form).  The reference applies the SAME
    roundings in float64 (torch.float8_e4m3fn), so what is left is fp32 summation order; the distance to the UNROUNDED product is the
    policy's error and is printed ([site] transformers/models/qwen2/modeling_qwen2.py Linear layers via oracle/qwen2.py)."""
    torch.manual_seed(M + N + K)
    x = torch.randn(M, K)
    W = bf(torch.randn(N, K) / math.sqrt(K))
    hi = bf(x)
    lo8 = _e4m3((x - hi) * 256.0) / 256.0
    W8 = _e4m3(W * 64.0) / 64.0
    ref = hi.double() @ W.double().t() + lo8.double() @ W8.double().t()
    exact = x.double() @ W.double().t()
    xd, wd = dev_f32(x), dev_bf16(W)
    a = torch.zeros(M, 2 * K, dtype=torch.bfloat16, device=DEV)
    w8 = torch.zeros(N, 2 * K, dtype=torch.uint8, device=DEV)
    call(lib().fv_op_lo8_pack(xd.data_ptr(), a.data_ptr(), 2 * K, wd.data_ptr(), w8.data_ptr(), M, K, N, stream()), "fv_op_lo8_pack")
    torch.cuda.synchronize()
    assert torch.equal(a[:, :K].float().cpu(), hi)
    got_lo = a.view(torch.uint8).view(M, 4 * K)[:, 2 * K:3 * K].view(torch.float8_e4m3fn).float().cpu() / 256.0
    assert torch.equal(got_lo, lo8) and torch.equal(w8[:, :K].view(torch.float8_e4m3fn).float().cpu() / 64.0, W8)
    wsb = torch.empty(16 * 1024 * 1024, dtype=torch.float32, device=DEV) if ws else None   # two ranges of 256 x 17920 fp32 partials fit
    wsp, wsn = (wsb.data_ptr(), wsb.numel() * 4) if ws else (None, 0)
    if epi == "swiglu":
        I = N // 2
        Wi = torch.stack([W[:I].view(I // 8, 8, K), W[I:].view(I // 8, 8, K)], dim=1).reshape(N, K).contiguous()   # rows [8 gate | 8 up]
        call(lib().fv_op_lo8_pack(None, None, 0, dev_bf16(Wi).data_ptr(), w8.data_ptr(), M, K, N, stream()), "fv_op_lo8_pack W")
        g, u = ref[:, :I], ref[:, I:]
        act = torch.nn.functional.silu(g) * u
        out = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)      # [hi I | lo8 bytes ...]
        call(lib().fv_op_gemm_lo8(a.data_ptr(), 2 * K, dev_bf16(Wi).data_ptr(), w8.data_ptr(), M, N, K, None, None, 0, out.data_ptr(), N, 7, wsp, wsn, stream()),
             "fv_op_gemm_lo8 swiglu-split")
        torch.cuda.synchronize()
        oh = out[:, :I].float().cpu()
        ol = out.view(torch.uint8).view(M, 2 * N)[:, 2 * I:3 * I].view(torch.float8_e4m3fn).float().cpu() / 256.0
        check_close(oh + ol, act.float(), rel=2e-4, amax=2e-3, what=f"hi + lo8 SwiGLU output {M}x{N}x{K}")     # the output's own lo8 rounding: 2^-13
        return
    b, res = torch.randn(N), torch.randn(M, N)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    if epi == "f32":
        call(lib().fv_op_gemm_lo8(a.data_ptr(), 2 * K, wd.data_ptr(), w8.data_ptr(), M, N, K, dev_f32(b).data_ptr(), None, 0, out.data_ptr(), N, _lib.EPI_F32, wsp, wsn, stream()),
             "fv_op_gemm_lo8 f32")
        want = ref + b.double()
    else:
        r = dev_f32(res)
        call(lib().fv_op_gemm_lo8(a.data_ptr(), 2 * K, wd.data_ptr(), w8.data_ptr(), M, N, K, None, r.data_ptr(), N, out.data_ptr(), N, _lib.EPI_RES_F32, wsp, wsn, stream()),
             "fv_op_gemm_lo8 res_f32")
        want = ref + res.double()
    torch.cuda.synchronize()
    check_close(out.cpu(), want.float(), rel=2e-5, amax=2e-4, what=f"hi + lo8 gemm {M}x{N}x{K} {epi}")
    pol = float((ref - exact).norm() / exact.norm())
    print(f"[hi + lo8 {M}x{N}x{K}] policy error vs the unrounded product: {pol:.2e}")
    assert pol <= 1.2e-4


# ------------------------------------------------------------------------------------------------ round 3: fp16-operand GEMMs
@pytest.mark.parametrize("M,N,K", [(100, 256, 896), (512, 1024, 896), (4096, 5120, 256), (256, 896, 4864)])
def test_gemm_f16_operands(M, N, K):
    """llm_precision = 2's projections: A and W hold fp16 bits, fp32 accumulation on v_mfma_f32_16x16x32_f16 -- through the
    register-staged kernel (64- and 128-row tiles) and the 256-tile LDS-DMA kernel, plain and split-K (decoder down projection:
    [site] transformers/models/qwen2/modeling_qwen2.py Qwen2MLP.down_proj via oracle/qwen2.py).  Operands are fp16-exact, so the
    only difference to the float64 reference is the fp32 accumulation order."""
    torch.manual_seed(M + N + K)
    A = (torch.randn(M, K) * 0.7).half()
    W = (torch.randn(N, K) / math.sqrt(K)).half()
    res = torch.randn(M, N)
    ref = A.double() @ W.double().t()
    a, w = A.to(DEV).contiguous(), W.to(DEV).contiguous()
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    call(lib().fv_op_gemm_f16(a.data_ptr(), K, w.data_ptr(), M, N, K, None, None, 0, out.data_ptr(), N, _lib.EPI_F32, None, 0, stream()), "fv_op_gemm_f16")
    torch.cuda.synchronize()
    check_close(out.cpu(), ref.float(), rel=2e-5, amax=2e-4, what=f"f16 gemm f32 out {M}x{N}x{K}")
    ws = torch.empty(8 * M * ((N + 255) // 256 * 256), dtype=torch.float32, device=DEV)
    r = res.to(DEV)
    out2 = torch.full((M, N), float("nan"), dtype=torch.float32, device=DEV)
    call(lib().fv_op_gemm_f16(a.data_ptr(), K, w.data_ptr(), M, N, K, None, r.data_ptr(), N, out2.data_ptr(), N, _lib.EPI_RES_F32,
                              ws.data_ptr(), ws.numel() * 4, stream()), "fv_op_gemm_f16 res_f32 (+ split-K scratch)")
    torch.cuda.synchronize()
    check_close(out2.cpu(), (res.double() + ref).float(), rel=2e-5, amax=2e-4, what=f"f16 gemm res_f32 {M}x{N}x{K}")


@pytest.mark.parametrize("M,I,K", [(96, 128, 896), (4096, 2560, 256)])
def test_gemm_f16_swiglu(M, I, K):
    """FV_EPI_SWIGLU_F16: gate / up rows interleaved by 8, out = silu(gate) * up / 16 as fp16 (include/fastvla_hip.h)."""
    torch.manual_seed(M + I)
    A = (torch.randn(M, K) * 0.7).half()
    Wg, Wu = (torch.randn(I, K) / math.sqrt(K)).half(), (torch.randn(I, K) / math.sqrt(K)).half()
    Wi = torch.stack([Wg.view(I // 8, 8, K), Wu.view(I // 8, 8, K)], dim=1).reshape(2 * I, K).contiguous()
    g, u = A.double() @ Wg.double().t(), A.double() @ Wu.double().t()
    ref = (torch.nn.functional.silu(g) * u / 16).float()
    a, w = A.to(DEV).contiguous(), Wi.to(DEV)
    out = torch.full((M, I), float("nan"), dtype=torch.float16, device=DEV)
    call(lib().fv_op_gemm_f16(a.data_ptr(), K, w.data_ptr(), M, 2 * I, K, None, None, 0, out.data_ptr(), I, _lib.EPI_SWIGLU_F16, None, 0, stream()),
         "fv_op_gemm_f16 swiglu")
    torch.cuda.synchronize()
    check_close(out.float().cpu(), ref, rel=6e-4, amax=2e-3, what=f"f16 swiglu {M}x{I}x{K}")   # fp16 output rounding: 2^-11


@pytest.mark.parametrize("M,I,K", [(96, 128, 896), (4096, 2560, 256)])
def test_gemm_f16_swiglu_saturates_instead_of_overflowing(M, I, K):
    """VERDICT r3 #1-ii / ADVICE r3: FV_EPI_SWIGLU_F16 converts with SATURATING casts.  Operands scaled so that silu(gate) * up / 16
    leaves the binary16 range on most entries (an outlier channel of a real checkpoint): every output must be finite, the overflowing
    ones exactly +-65504, the others as the unsaturated arithmetic gives them.  Both GEMM kernels (register-staged, 256-tile)."""
    torch.manual_seed(M + I + 1)
    A = (torch.randn(M, K) * 400).half()     # gate, up ~ N(0, 1600^2): silu(gate) * up / 16 reaches ~1e5 .. 1e6
    Wg, Wu = (torch.randn(I, K) * 4 / math.sqrt(K)).half(), (torch.randn(I, K) * 4 / math.sqrt(K)).half()
    Wi = torch.stack([Wg.view(I // 8, 8, K), Wu.view(I // 8, 8, K)], dim=1).reshape(2 * I, K).contiguous()
    g, u = A.double() @ Wg.double().t(), A.double() @ Wu.double().t()
    ref = (torch.nn.functional.silu(g) * u / 16)
    a, w = A.to(DEV).contiguous(), Wi.to(DEV)
    out = torch.full((M, I), float("nan"), dtype=torch.float16, device=DEV)
    call(lib().fv_op_gemm_f16(a.data_ptr(), K, w.data_ptr(), M, 2 * I, K, None, None, 0, out.data_ptr(), I, _lib.EPI_SWIGLU_F16, None, 0, stream()),
         "fv_op_gemm_f16 swiglu (overflowing)")
    torch.cuda.synchronize()
    o = out.float().cpu().double()
    big = ref.abs() > 66000
    assert float(big.float().mean()) > 0.01, "the test must actually overflow"
    assert torch.isfinite(o).all()
    assert torch.equal(o[big], torch.sign(ref[big]) * 65504.0)
    ok = ref.abs() < 65000
    check_close(o[ok].float(), ref[ok].float(), rel=6e-4, amax=2e-3, what="entries inside the fp16 range")


@pytest.mark.parametrize("dtype,Cin,Hin,Win", [("f32", 3, 84, 84), ("u8", 3, 60, 100), ("f32", 1, 97, 41)])
def test_stem_fused_from_source_images(dtype, Cin, Hin, Win):
    """the stem that samples the SOURCE image through the letterbox arithmetic (fv_op_stem_fused_images, the kernel
    behind fv_vision_forward_images) against fv_preprocess + fv_op_stem_fused: bit-identical (reference: resize_with_pad,
    model/fastvlm_adapter.py:36-55, ahead of the tower at :533) -- float and uint8 sources, RGB and gray, square and not."""
    from fastvla_hip import FastVLAEngine, arch
    S, B, C0 = 256, 2, 96
    m = arch.ModelConfig("lb", arch.LLMConfig(hidden=64, layers=1, heads=2, kv_heads=1, head_dim=32, inter=64, vocab=64),
                         arch.TowerConfig(layers=(1,), dims=(32,), attn_stages=(), image_size=S))
    eng = FastVLAEngine(m, hidden_dim=32, fusion_dim=32, max_batch=B, max_text_tokens=8)
    g = torch.Generator().manual_seed(Hin + Win)
    img = (torch.rand(B, Cin, Hin, Win, generator=g) if dtype == "f32" else torch.randint(0, 256, (B, Cin, Hin, Win), generator=g, dtype=torch.uint8)).to(DEV)
    pix = eng.preprocess(img, 0.25, True)
    wp, b1 = dev_bf16(torch.randn(C0, 64, generator=g) / 5), dev_f32(torch.randn(C0, generator=g) * 0.1)
    w2, b2 = dev_f32(torch.randn(9, C0, generator=g) / 3), dev_f32(torch.randn(C0, generator=g) * 0.1)
    ya = torch.full((B, S // 4, S // 4, C0), float("nan"), dtype=torch.bfloat16, device=DEV)
    yb = torch.full_like(ya, float("nan"))
    call(lib().fv_op_stem_fused(pix.data_ptr(), wp.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), ya.data_ptr(), B, S, C0, stream()), "fv_op_stem_fused")
    call(lib().fv_op_stem_fused_images(img.data_ptr(), _lib.FV_F32 if dtype == "f32" else _lib.FV_U8, B, Cin, Hin, Win, 0.25, 1, wp.data_ptr(), b1.data_ptr(),
                                       w2.data_ptr(), b2.data_ptr(), yb.data_ptr(), S, C0, stream()), "fv_op_stem_fused_images")
    torch.cuda.synchronize()
    assert torch.isfinite(yb.float()).all() and torch.equal(ya, yb)
    eng.close()
