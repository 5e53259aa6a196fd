// ops_api.hip -- op-level C entry points (include/fastvla_hip_testops.h: TEST-ONLY, built into tests/_native/libfastvla_hip_testops.so, not into the product library): one kernel each, so the
// parity tests can check every kernel of the path against the oracle on its own.  No engine state is involved.
#include <cmath>
#include <vector>

#include "kernels.h"

extern "C" {

int fv_op_gemm(const void* A, int lda, const void* W, int M, int N, int K, const float* bias, const float* scale,
               const void* res, int ldr, void* out, int ldo, int epilogue, fv_stream s) {
  fv::GemmArgs g{static_cast<const bf16_t*>(A), lda, static_cast<const bf16_t*>(W), M, N, K, bias, scale, res, ldr, out, ldo, epilogue};
  return fv::launch_gemm(g, static_cast<hipStream_t>(s));
}
int fv_op_gemm_ksplit(const void* A, int lda, const void* W, int M, int N, int K, const float* bias, const void* res, int ldr,
                      void* out, int ldo, int epilogue, fv_stream s) {
  fv::GemmArgs g{static_cast<const bf16_t*>(A), lda, static_cast<const bf16_t*>(W), M, N, K, bias, nullptr, res, ldr, out, ldo, epilogue};
  g.ksplit = true;
  return fv::launch_gemm(g, static_cast<hipStream_t>(s));
}
int fv_op_gemm_splitk(const void* A, int lda, const void* W, int M, int N, int K, const float* bias, const void* res, int ldr,
                      void* out, int ldo, int epilogue, int ksplit, void* ws, size_t ws_bytes, fv_stream s) {
  fv::GemmArgs g{static_cast<const bf16_t*>(A), lda, static_cast<const bf16_t*>(W), M, N, K, bias, nullptr, res, ldr, out, ldo, epilogue};
  g.ksplit = ksplit != 0;
  g.splitk_ws = static_cast<float*>(ws);
  g.splitk_bytes = ws_bytes;
  return fv::launch_gemm(g, static_cast<hipStream_t>(s));
}

int fv_op_gemm_tn(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int f16, const float* bias, void* out, int ldo, void* ws,
                  size_t ws_bytes, fv_stream s) {
  fv::GemmArgs g{static_cast<const bf16_t*>(A), lda, static_cast<const bf16_t*>(W), M, N, K, bias, nullptr, nullptr, 0, out, ldo, FV_EPI_F32};
  g.tn = 1; g.ldw = ldw; g.f16 = f16 != 0;
  g.splitk_ws = static_cast<float*>(ws);
  g.splitk_bytes = ws_bytes;
  return fv::launch_gemm(g, static_cast<hipStream_t>(s));
}

int fv_op_gemm_lo8(const void* A, int lda, const void* W, const void* W8, int M, int N, int K, const float* bias, const void* res, int ldr, void* out,
                   int ldo, int epilogue, void* ws, size_t ws_bytes, fv_stream s) {
  fv::GemmArgs g{static_cast<const bf16_t*>(A), lda, static_cast<const bf16_t*>(W), M, N, K, bias, nullptr, res, ldr, out, ldo, epilogue};
  g.ksplit = 2;
  g.W8 = W8;
  g.splitk_ws = static_cast<float*>(ws);
  g.splitk_bytes = ws_bytes;
  return fv::launch_gemm(g, static_cast<hipStream_t>(s));
}
int fv_op_lo8_pack(const float* x, void* a_out, int lda, const void* W, void* w8_out, int M, int K, int N, fv_stream s) {
  // x (M, K) f32 -> the hi + lo8 operand rows (lda bf16 units each: K bf16, then at byte 2K the K fp8 remainders); W (N, K) bf16 -> W8
  int rc = FV_OK;
  if (x && a_out) {
    // RMSNorm with unit weights would rescale: use the plain split of the training glue for hi, then its remainders as fp8
    rc = fv::launch_split_rows(x, K, static_cast<bf16_t*>(a_out), lda, 0, M, K, static_cast<hipStream_t>(s));
    if (rc == FV_OK) rc = fv::launch_lo8_rows(x, K, static_cast<bf16_t*>(a_out), lda, M, K, static_cast<hipStream_t>(s));
  }
  if (rc == FV_OK && W && w8_out) rc = fv::launch_bf16_to_w8(static_cast<const bf16_t*>(W), w8_out, (size_t)N, K, static_cast<hipStream_t>(s));
  return rc;
}

int fv_op_gemm_f16(const void* A, int lda, const void* W, int M, int N, int K, const float* bias, const void* res, int ldr, void* out,
                   int ldo, int epilogue, void* ws, size_t ws_bytes, fv_stream s) {
  fv::GemmArgs g{static_cast<const bf16_t*>(A), lda, static_cast<const bf16_t*>(W), M, N, K, bias, nullptr, res, ldr, out, ldo, epilogue};
  g.f16 = 1;
  g.splitk_ws = static_cast<float*>(ws);
  g.splitk_bytes = ws_bytes;
  return fv::launch_gemm(g, static_cast<hipStream_t>(s));
}

int fv_op_dwconv(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int k,
                 int stride, int mult, int gelu, fv_stream s) {
  return fv::launch_dwconv(static_cast<const bf16_t*>(x), w, bias, static_cast<bf16_t*>(y), B, H, W, C, k, stride, mult, gelu, static_cast<hipStream_t>(s));
}

int fv_op_stem_conv(const void* pix, const float* w, const float* bias, void* y, int B, int S, int Cout, fv_stream s) {
  return fv::launch_stem_conv(static_cast<const bf16_t*>(pix), w, bias, static_cast<bf16_t*>(y), B, S, Cout, static_cast<hipStream_t>(s));
}

int fv_op_stem_mfma(const void* pix, const void* wp, const float* bias, void* y, int B, int S, int Cout, fv_stream s) {
  return fv::launch_stem_mfma(static_cast<const bf16_t*>(pix), static_cast<const bf16_t*>(wp), bias, static_cast<bf16_t*>(y), B, S, Cout, static_cast<hipStream_t>(s));
}
int fv_op_stem_fused(const void* pix, const void* wp, const float* b1, const float* w2, const float* b2, void* y, int B, int S, int Cout,
                     fv_stream s) {
  return fv::launch_stem_fused(static_cast<const bf16_t*>(pix), static_cast<const bf16_t*>(wp), b1, w2, b2, static_cast<bf16_t*>(y), B, S, Cout,
                               static_cast<hipStream_t>(s));
}

int fv_op_stem_fused_images(const void* img, int dtype, int B, int C, int Hin, int Win, float pad_value, int resize_with_padding, const void* wp,
                            const float* b1, const float* w2, const float* b2, void* y, int S, int Cout, fv_stream s) {
  return fv::launch_stem_fused_lb(img, dtype, B, C, Hin, Win, pad_value, resize_with_padding, static_cast<const bf16_t*>(wp), b1, w2, b2,
                                  static_cast<bf16_t*>(y), S, Cout, static_cast<hipStream_t>(s));
}

int fv_op_layernorm_rows(const void* x, const float* w, const float* b, void* y, int rows, int C, float eps, fv_stream s) {
  return fv::launch_layernorm_rows(static_cast<const bf16_t*>(x), w, b, static_cast<bf16_t*>(y), rows, C, eps, static_cast<hipStream_t>(s));
}

int fv_op_attention(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, void* out, int ldo, int B,
                    int T, int heads, int kv_heads, int D, int causal, const int32_t* lens, float scale, fv_stream s) {
  return fv::launch_attention(static_cast<const bf16_t*>(q), static_cast<const bf16_t*>(k), static_cast<const bf16_t*>(v), ldq, ldk,
                              ldv, static_cast<bf16_t*>(out), ldo, B, T, heads, kv_heads, D, causal, lens, 0, scale, static_cast<hipStream_t>(s));
}

int fv_op_rmsnorm(const float* x, const float* w, void* y_bf16, int rows, int H, float eps, fv_stream s) {
  return fv::launch_rmsnorm(x, w, static_cast<bf16_t*>(y_bf16), nullptr, H, rows, H, eps, static_cast<hipStream_t>(s));
}

int fv_op_rope(void* qkv, int ld, int rows, int T, int heads, int kv_heads, int D, float theta, fv_stream s) {
  if (T <= 0 || D <= 0 || D % 16) return fv_fail(FV_ERR_ARG, "rope: bad T/D");
  std::vector<float> cs((size_t)T * D);
  fv::rope_table_host(cs.data(), T, D, theta);
  float2* tab = nullptr;
  FV_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&tab), cs.size() * 4));
  hipError_t e = hipMemcpy(tab, cs.data(), cs.size() * 4, hipMemcpyHostToDevice);
  int rc = e == hipSuccess ? fv::launch_rope(static_cast<bf16_t*>(qkv), tab, ld, rows, T, heads, kv_heads, D, static_cast<hipStream_t>(s))
                           : fv_hip_fail(e, "hipMemcpy(rope table)");
  (void)hipStreamSynchronize(static_cast<hipStream_t>(s));
  (void)hipFree(tab);
  return rc;
}

int fv_op_attention_bwd(const float* qkv, int ld, const float* dO, float* dqkv, void* out_bf16_scratch, float* stat_scratch, int B, int T,
                        int heads, int kv_heads, int D, const int32_t* lens, float theta, fv_stream st) {
  if (!qkv || !dO || !dqkv || !out_bf16_scratch || !stat_scratch) return fv_fail(FV_ERR_ARG, "attention_bwd: null pointer");
  if (T <= 0 || (D != 64 && D != 128)) return fv_fail(FV_ERR_UNSUPPORTED, "attention_bwd: head_dim must be 64 or 128");
  hipStream_t s = static_cast<hipStream_t>(st);
  std::vector<float> cs((size_t)T * D);
  fv::rope_table_host(cs.data(), T, D, theta);
  float2* tab = nullptr;
  FV_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&tab), cs.size() * 4));
  hipError_t e = hipMemcpy(tab, cs.data(), cs.size() * 4, hipMemcpyHostToDevice);
  int rc = e == hipSuccess ? FV_OK : fv_hip_fail(e, "hipMemcpy(rope table)");
  const int qd = heads * D;
  bf16_t* o = static_cast<bf16_t*>(out_bf16_scratch);
  float* lse = stat_scratch;
  float* delta = stat_scratch + (size_t)B * heads * T;
  const float scale = 1.0f / sqrtf((float)D);
  void* scr = nullptr;   // the split-bf16 kernels' K / V records (the training path keeps this in its workspace)
  if (rc == FV_OK) { e = hipMalloc(&scr, fv::attention_split_scratch_bytes(B, T, kv_heads, D)); if (e != hipSuccess) rc = fv_hip_fail(e, "hipMalloc(attention scratch)"); }
  if (rc == FV_OK) rc = fv::launch_attention_f32(const_cast<float*>(qkv), ld, o, o + qd, 2 * qd, B, T, heads, kv_heads, D, lens, 0, scale, s, tab, nullptr, 0, 0, lse, 0, scr);
  if (rc == FV_OK) rc = fv::launch_attention_bwd(qkv, ld, o, o + qd, 2 * qd, dO, qd, lse, delta, dqkv, B, T, heads, kv_heads, D, lens, 0, scale, tab, s, nullptr, scr);
  (void)hipStreamSynchronize(s);
  (void)hipFree(tab);
  if (scr) (void)hipFree(scr);
  return rc;
}

int fv_op_rmsnorm_bwd(const float* x, const float* w, const float* dy, const float* dres, float* dx, float* dw, float* scratch, int rows,
                      int H, float eps, fv_stream s) {
  return fv::launch_rmsnorm_bwd(x, w, dy, dres, dx, dw, scratch, rows, H, eps, static_cast<hipStream_t>(s));
}

int fv_op_se_gelu(const void* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y,
                  float* scratch, int B, int P, int C, int R, fv_stream s) {
  return fv::launch_se_gelu(static_cast<const bf16_t*>(x), w1, b1, w2, b2, static_cast<bf16_t*>(y), scratch, B, P, C, R, static_cast<hipStream_t>(s));
}

int fv_op_dwconv_mfma(const void* x, const void* ttab, const float* bias, void* y, int B, int H, int W, int C, int k, int gelu,
                      fv_stream s) {
  return fv::launch_dwconv_mfma(static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(ttab), bias, static_cast<bf16_t*>(y), B, H, W, C, k,
                                gelu, static_cast<hipStream_t>(s));
}
int fv_op_dwconv_s2_mfma(const void* x, const void* ttab, const float* bias, void* y, int B, int H, int W, int C, int gelu, fv_stream s) {
  return fv::launch_dwconv_s2_mfma(static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(ttab), bias, static_cast<bf16_t*>(y), B, H, W, C,
                                   gelu, static_cast<hipStream_t>(s));
}
int fv_op_dwconv_pair(const void* x, const void* t3, const float* b3, const void* t7, const float* b7, void* y1, void* y2, int B,
                      int H, int W, int C, fv_stream s) {
  return fv::launch_dwconv_pair(static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(t3), b3, static_cast<const bf16_t*>(t7), b7,
                                static_cast<bf16_t*>(y1), static_cast<bf16_t*>(y2), B, H, W, C, static_cast<hipStream_t>(s));
}

int fv_op_convffn(const void* x, const void* w1, const float* b1, const void* w2p, const float* b2, const float* ls,
                  const void* res, void* out, int M, int C, fv_stream s) {
  return fv::launch_convffn(static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(w1), b1, static_cast<const bf16_t*>(w2p), b2, ls,
                            static_cast<const bf16_t*>(res), static_cast<bf16_t*>(out), M, C, 4 * C, static_cast<hipStream_t>(s));
}

int fv_op_convffn32(const void* x, const void* wq, const float* b1, const float* b2, const float* ls, const void* res, void* out,
                    int M, int C, fv_stream s) {
  return fv::launch_convffn32(static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(wq), b1, b2, ls, static_cast<const bf16_t*>(res),
                              static_cast<bf16_t*>(out), M, C, 4 * C, static_cast<hipStream_t>(s));
}

int fv_op_convffn32_stash(const void* x, const void* wq, const float* b1, const float* b2, const float* ls, const void* res, void* out,
                          int M, int C, void* stash_y, fv_stream s) {
  if (!stash_y) return fv_fail(FV_ERR_ARG, "fv_op_convffn32_stash: null stash");
  return fv::launch_convffn32(static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(wq), b1, b2, ls, static_cast<const bf16_t*>(res),
                              static_cast<bf16_t*>(out), M, C, 4 * C, static_cast<hipStream_t>(s), nullptr, 0, static_cast<bf16_t*>(stash_y));
}
// fv_op_gemm_f16 with FV_EPI_MUL_GELUP and its second output: out f16 = acc * gelu'(4 aux), h_out f16 = gelu(4 aux) rounded to bf16
int fv_op_gemm_f16_gelup(const void* A, int lda, const void* W, int M, int N, int K, const void* aux, int ldaux, void* out, int ldo, void* h_out, fv_stream s) {
  fv::GemmArgs g{static_cast<const bf16_t*>(A), lda, static_cast<const bf16_t*>(W), M, N, K, nullptr, nullptr, aux, ldaux, out, ldo, FV_EPI_MUL_GELUP};
  g.f16 = 1;
  g.stash = h_out;
  return fv::launch_gemm(g, static_cast<hipStream_t>(s));
}

// tap + bias gradients of a depthwise conv (the kernels behind the tower's backward): x bf16 (B,H,W,C), dy fp16 (B,Ho,Wo,C*mult) -> dw f32 tap-major [k*k][C*mult], db [C*mult]
int fv_op_dw_wgrad(const void* x, const void* dy, float* dw, float* db, float* scratch, size_t scratch_floats, int B, int H, int W, int C, int k, int stride, int mult, fv_stream s) {
  const int pad = k / 2, Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (scratch_floats < fv::dw_bwd_scratch_floats(B, Ho, Wo, C * mult, k)) return fv_fail(FV_ERR_ARG, "fv_op_dw_wgrad: scratch too small (%zu floats needed)", fv::dw_bwd_scratch_floats(B, Ho, Wo, C * mult, k));
  return fv::launch_dw_wgrad(static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(dy), dw, db, scratch, B, H, W, C, k, stride, mult, static_cast<hipStream_t>(s));
}

int fv_op_convffn32_split(const void* x, const void* wq, const float* b1, const float* b2, const float* ls, const void* res, void* out,
                          int M, int C, float* part, size_t part_bytes, fv_stream s) {
  return fv::launch_convffn32(static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(wq), b1, b2, ls, static_cast<const bf16_t*>(res),
                              static_cast<bf16_t*>(out), M, C, 4 * C, static_cast<hipStream_t>(s), part, part_bytes);
}

}  // extern "C"
