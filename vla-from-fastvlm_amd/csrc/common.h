// common.h -- shared device helpers for libfastvla_hip (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define FV_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
// plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving) on gfx950
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
// two RNE conversions in ONE v_cvt_pk_bf16_f32 (the scalar casts above cost two of them plus a shift and an or)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  const bf16x2 r = __builtin_convertvector((f32x2){lo, hi}, bf16x2);
  return __builtin_bit_cast(uint32_t, r);
}
// fp16 (11 significant bits) packing for the decoder's single-pass operand mode (llm_precision = 2): round-to-nearest-even casts
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
// SATURATING: values beyond +-65504 (the largest finite binary16) clamp instead of becoming inf -- a real checkpoint's outlier channel
// must not turn the policy's actions into NaN; the kernels that convert activations count every clamp (fv_llm_fp16_saturations)
#define FV_F16_MAX 65504.0f
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
  lo = __builtin_amdgcn_fmed3f(lo, -FV_F16_MAX, FV_F16_MAX);
  hi = __builtin_amdgcn_fmed3f(hi, -FV_F16_MAX, FV_F16_MAX);
  const f16x2 r = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(uint32_t, r);
}
// ---- "hi + lo8" operands (llm_precision = 5): x ~= bf16(x) + fp8_e4m3((x - bf16(x)) * 2^8) * 2^-8.  The lo half costs ONE byte per
// element and its product runs on v_mfma_scale_f32_16x16x128_f8f6f4 (twice the MACs per clock of the bf16 MFMA, the 2^-14 of the two
// operand scales applied by the instruction's E8M0 scale operand): 1.5 passes instead of split-bf16's 2, 13 significant bits instead
// of 16 (5.8e-5 per GEMM against 2.5e-6; one fp16 pass: 2.1e-4).  v_cvt_pk_fp8_f32 (OCP e4m3fn on gfx950) does NOT saturate
// (|v| > 448 -> NaN): clamp first.
#define FV_LO8_SCALE 256.0f   // 2^8 on the remainder
#define FV_W8_SCALE 64.0f     // 2^6 on the weights' fp8 copy
#define FV_LO8_MFMA_SCALE 113 // E8M0 of 2^-(8 + 6)
#define FV_F8_MAX 448.0f
__device__ __forceinline__ uint32_t pack_f8x4(float a, float b, float c, float d) {
  a = __builtin_amdgcn_fmed3f(a, -FV_F8_MAX, FV_F8_MAX); b = __builtin_amdgcn_fmed3f(b, -FV_F8_MAX, FV_F8_MAX);
  c = __builtin_amdgcn_fmed3f(c, -FV_F8_MAX, FV_F8_MAX); d = __builtin_amdgcn_fmed3f(d, -FV_F8_MAX, FV_F8_MAX);
  int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
  return (uint32_t)w;
}
// remainders l[0..7] (already x - bf16(x)) -> 8 fp8 bytes
__device__ __forceinline__ uint2 pack_lo8(const float* l) {
  uint2 u;
  u.x = pack_f8x4(l[0] * FV_LO8_SCALE, l[1] * FV_LO8_SCALE, l[2] * FV_LO8_SCALE, l[3] * FV_LO8_SCALE);
  u.y = pack_f8x4(l[4] * FV_LO8_SCALE, l[5] * FV_LO8_SCALE, l[6] * FV_LO8_SCALE, l[7] * FV_LO8_SCALE);
  return u;
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
  f[0] = bf_lo(u.x); f[1] = bf_hi(u.x); f[2] = bf_lo(u.y); f[3] = bf_hi(u.y);
  f[4] = bf_lo(u.z); f[5] = bf_hi(u.z); f[6] = bf_lo(u.w); f[7] = bf_hi(u.w);
}
__device__ __forceinline__ uint4 pack8_h(const float* f) {
  uint4 u;
  u.x = pack_h2(f[0], f[1]); u.y = pack_h2(f[2], f[3]); u.z = pack_h2(f[4], f[5]); u.w = pack_h2(f[6], f[7]);
  return u;
}
__device__ __forceinline__ void unpack8_h(const uint4& u, float* f) {
  const f16x8 h = __builtin_bit_cast(f16x8, u);
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)h[e];
}
// how many of the 8 values pack8_h will clamp (0 in a healthy model: the callers add it to the handle's saturation counter)
__device__ __forceinline__ void count_f16_sat8(const float* f, unsigned* counter) {
  float m = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(f[e]));
  if (counter && !(m <= FV_F16_MAX)) atomicAdd(counter, 1u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 u;
  u.x = pack_bf2(f[0], f[1]); u.y = pack_bf2(f[2], f[3]); u.z = pack_bf2(f[4], f[5]); u.w = pack_bf2(f[6], f[7]);
  return u;
}

// GELU 0.5 x (1 + erf(x/sqrt2)) = x * Phi(x), with Phi(x) ~= sigmoid(x (a + b x^2 + c x^4)): a minimax fit of the exact
// (erf) form, max |error| 2.5e-5 over all x (the tanh form is 4.7e-4) -- 20x below the bf16 resolution of the outputs it
// feeds -- at 8 VALU issues (1 exp, 1 rcp) instead of ~22 for an erf polynomial: GELU is applied to 23 G hidden
// activations per step and the fp32 VALU, not MFMA, is what it competes for.  Coefficients carry -log2(e).
__device__ __forceinline__ float gelu_f(float x) {
  const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);  // beyond |x| = 8 the result is x or 0 to fp32 precision
  const float x2 = xc * xc;
  float t = 1.0142648e-3f;                 //  0.0007030350668716528 * log2(e)
  t = t * x2 - 1.0677576e-1f;              // -0.07401130190482874  * log2(e)
  t = t * x2 - 2.3011213f;                 // -1.595015756858567    * log2(e)
  const float e = __builtin_amdgcn_exp2f(xc * t);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
#define FV_GELU_C0 0.398636314f
#define FV_GELU_C1 -0.0658484117f
#define FV_GELU_C2 0.00950338221f
#define FV_GELU_C3 -0.00101413109f
#define FV_GELU_C4 7.77420515e-05f
#define FV_GELU_C5 -4.11762334e-06f
#define FV_GELU_C6 1.4164305e-07f
#define FV_GELU_C7 -2.82720812e-09f
#define FV_GELU_C8 2.47331455e-11f
// Two GELUs at once with no transcendental: x * clamp(0.5 + xc P(xc^2), 0, 1), xc = clamp(x, -R, R), P a degree-8 minimax fit
// (LP over [0, R] with the constraint 0.5 + R P(R^2) >= 1 so the clamp makes the tail exact: tools/gelu_fit.py); max |error|
// 5.1e-5 over all x in fp32.  12 VALU issues per PAIR, 10 of them packed fp32 (v_pk_mul/fma_f32, full rate on gfx950), where
// gelu_f spends 7 + two quarter-rate transcendentals per element: ~24 cycles per element instead of ~60.  The fused ConvFFN
// applies it to 23 G hidden activations per step between its two products, with nothing else to issue meanwhile.
// N pairs in lockstep (coefficient-major): one wave per SIMD cannot hide the latency of a dependent VALU chain, so the
// N independent Horner chains are what keeps the VALU issuing every cycle.
template <int N>
__device__ __forceinline__ void gelu2_n(f32x2 (&x)[N]) {
  const float R = 4.625f;
  f32x2 xc[N], x2[N], p[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    xc[n].x = __builtin_amdgcn_fmed3f(x[n].x, -R, R);
    xc[n].y = __builtin_amdgcn_fmed3f(x[n].y, -R, R);
  }
#pragma unroll
  for (int n = 0; n < N; ++n) x2[n] = xc[n] * xc[n];
#pragma unroll
  for (int n = 0; n < N; ++n) p[n] = __builtin_elementwise_fma(x2[n], (f32x2){FV_GELU_C8, FV_GELU_C8}, (f32x2){FV_GELU_C7, FV_GELU_C7});
#define FV_HORNER(c)                                                                        \
  _Pragma("unroll") for (int n = 0; n < N; ++n) p[n] = __builtin_elementwise_fma(p[n], x2[n], (f32x2){c, c});
  FV_HORNER(FV_GELU_C6) FV_HORNER(FV_GELU_C5) FV_HORNER(FV_GELU_C4) FV_HORNER(FV_GELU_C3)
  FV_HORNER(FV_GELU_C2) FV_HORNER(FV_GELU_C1) FV_HORNER(FV_GELU_C0)
#undef FV_HORNER
  // The VOP3P clamp bit saturates both halves to [0, 1] for free (the compiler spends two v_max on it).  gfx950 wants one
  // wait state between a packed VALU write and a dependent VALU read; hipcc pads its own packed ops with s_nop 0 but
  // cannot see into asm, so the statement carries its own: one in front and one behind the N clamped fmas (inside the
  // run the producers and consumers are at least one instruction apart already).
  static_assert(N == 1 || N == 2 || N == 4, "gelu2_n chains");
  if constexpr (N == 1) {  // phi overwrites xc in place
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %0, %1, 0.5 op_sel_hi:[1,1,0] clamp\n\ts_nop 0" : "+v"(xc[0]) : "v"(p[0]));
  } else if constexpr (N == 2) {
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %0, %2, 0.5 op_sel_hi:[1,1,0] clamp\n\tv_pk_fma_f32 %1, %1, %3, 0.5 op_sel_hi:[1,1,0] clamp\n\ts_nop 0"
        : "+v"(xc[0]), "+v"(xc[1]) : "v"(p[0]), "v"(p[1]));
  } else {
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %0, %4, 0.5 op_sel_hi:[1,1,0] clamp\n\tv_pk_fma_f32 %1, %1, %5, 0.5 op_sel_hi:[1,1,0] clamp\n\t"
        "v_pk_fma_f32 %2, %2, %6, 0.5 op_sel_hi:[1,1,0] clamp\n\tv_pk_fma_f32 %3, %3, %7, 0.5 op_sel_hi:[1,1,0] clamp\n\ts_nop 0"
        : "+v"(xc[0]), "+v"(xc[1]), "+v"(xc[2]), "+v"(xc[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]));
  }
#pragma unroll
  for (int n = 0; n < N; ++n) x[n] = x[n] * xc[n];
}
// The same form with a degree-6 polynomial in x^2 (7 coefficients, R = 3.9375): max |error| 1.8e-4 over all x -- below the bf16
// rounding of the value it produces for |gelu| > 0.1 (bf16 half-ulp = 2^-9 relative) -- for two packed fmas less per pair.
// For consumers that round the result to bf16 as an MFMA operand and are short of VALU issue slots (convffn32_kernel).
template <int N>
__device__ __forceinline__ void gelu2_n7(f32x2 (&x)[N]) {
  const float R = 3.9375f;
  f32x2 xc[N], x2[N], p[N];
#pragma unroll
  for (int n = 0; n < N; ++n) {
    xc[n].x = __builtin_amdgcn_fmed3f(x[n].x, -R, R);
    xc[n].y = __builtin_amdgcn_fmed3f(x[n].y, -R, R);
  }
#pragma unroll
  for (int n = 0; n < N; ++n) x2[n] = xc[n] * xc[n];
#pragma unroll
  for (int n = 0; n < N; ++n) p[n] = __builtin_elementwise_fma(x2[n], (f32x2){2.60001215e-08f, 2.60001215e-08f}, (f32x2){-1.76554451e-06f, -1.76554451e-06f});
#define FV_HORNER7(c)                                                                       \
  _Pragma("unroll") for (int n = 0; n < N; ++n) p[n] = __builtin_elementwise_fma(p[n], x2[n], (f32x2){c, c});
  FV_HORNER7(5.13497647e-05f) FV_HORNER7(-0.000848164123f) FV_HORNER7(0.00894825811f) FV_HORNER7(-0.0650006125f) FV_HORNER7(0.398250524f)
#undef FV_HORNER7
  static_assert(N == 1 || N == 2 || N == 4, "gelu2_n7 chains");
  if constexpr (N == 1) {
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %0, %1, 0.5 op_sel_hi:[1,1,0] clamp\n\ts_nop 0" : "+v"(xc[0]) : "v"(p[0]));
  } else if constexpr (N == 2) {
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %0, %2, 0.5 op_sel_hi:[1,1,0] clamp\n\tv_pk_fma_f32 %1, %1, %3, 0.5 op_sel_hi:[1,1,0] clamp\n\ts_nop 0"
        : "+v"(xc[0]), "+v"(xc[1]) : "v"(p[0]), "v"(p[1]));
  } else {
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %0, %4, 0.5 op_sel_hi:[1,1,0] clamp\n\tv_pk_fma_f32 %1, %1, %5, 0.5 op_sel_hi:[1,1,0] clamp\n\t"
        "v_pk_fma_f32 %2, %2, %6, 0.5 op_sel_hi:[1,1,0] clamp\n\tv_pk_fma_f32 %3, %3, %7, 0.5 op_sel_hi:[1,1,0] clamp\n\ts_nop 0"
        : "+v"(xc[0]), "+v"(xc[1]), "+v"(xc[2]), "+v"(xc[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]));
  }
#pragma unroll
  for (int n = 0; n < N; ++n) x[n] = x[n] * xc[n];
}
// GELU in the SCALED variable y = x / 4 (convffn32_kernel: W1, b1 carry the 1/4, W2 the 4 -- exact powers of two):
//     u = clamp(y y, 0, 1);  phi = clamp(1/2 + y Q(u), 0, 1);  g = y phi = GELU(4 y) / 4
// Q of degree 6 with 1/2 + Q(1) >= 1 (tools/gelu_fit_scaled.py), so both tails are exact without clamping y itself: max |error|
// 1.9e-4 in x units.  10 VALU issues per PAIR, rounding to bf16 included (gelu2_n7: 13 with the bias add it no longer needs).
template <int N>
__device__ __forceinline__ void gelu2s_n(f32x2 (&y)[N]) {
  static_assert(N == 1 || N == 2 || N == 4, "gelu2s_n chains");
  f32x2 u[N], p[N];
  if constexpr (N == 1) {
    asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %1 clamp\n\ts_nop 0" : "=v"(u[0]) : "v"(y[0]));
  } else if constexpr (N == 2) {
    asm("s_nop 0\n\tv_pk_mul_f32 %0, %2, %2 clamp\n\tv_pk_mul_f32 %1, %3, %3 clamp\n\ts_nop 0" : "=&v"(u[0]), "=&v"(u[1]) : "v"(y[0]), "v"(y[1]));
  } else {
    asm("s_nop 0\n\tv_pk_mul_f32 %0, %4, %4 clamp\n\tv_pk_mul_f32 %1, %5, %5 clamp\n\tv_pk_mul_f32 %2, %6, %6 clamp\n\tv_pk_mul_f32 %3, %7, %7 clamp\n\ts_nop 0"
        : "=&v"(u[0]), "=&v"(u[1]), "=&v"(u[2]), "=&v"(u[3]) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]));
  }
#pragma unroll
  for (int n = 0; n < N; ++n) p[n] = __builtin_elementwise_fma(u[n], (f32x2){1.52883381f, 1.52883381f}, (f32x2){-6.70501111f, -6.70501111f});
#define FV_HORNERS(c)                                                                       \
  _Pragma("unroll") for (int n = 0; n < N; ++n) p[n] = __builtin_elementwise_fma(p[n], u[n], (f32x2){c, c});
  FV_HORNERS(12.571231f) FV_HORNERS(-13.3368161f) FV_HORNERS(8.98292293f) FV_HORNERS(-4.13267958f) FV_HORNERS(1.59153356f)
#undef FV_HORNERS
  if constexpr (N == 1) {
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %0, 0.5 op_sel_hi:[1,1,0] clamp\n\ts_nop 0" : "+v"(p[0]) : "v"(y[0]));
  } else if constexpr (N == 2) {
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %2, %0, 0.5 op_sel_hi:[1,1,0] clamp\n\tv_pk_fma_f32 %1, %3, %1, 0.5 op_sel_hi:[1,1,0] clamp\n\ts_nop 0"
        : "+v"(p[0]), "+v"(p[1]) : "v"(y[0]), "v"(y[1]));
  } else {
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %4, %0, 0.5 op_sel_hi:[1,1,0] clamp\n\tv_pk_fma_f32 %1, %5, %1, 0.5 op_sel_hi:[1,1,0] clamp\n\t"
        "v_pk_fma_f32 %2, %6, %2, 0.5 op_sel_hi:[1,1,0] clamp\n\tv_pk_fma_f32 %3, %7, %3, 0.5 op_sel_hi:[1,1,0] clamp\n\ts_nop 0"
        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]) : "v"(y[0]), "v"(y[1]), "v"(y[2]), "v"(y[3]));
  }
#pragma unroll
  for (int n = 0; n < N; ++n) y[n] = y[n] * p[n];
}
__device__ __forceinline__ f32x2 gelu2_f(f32x2 x) {
  f32x2 v[1] = {x};
  gelu2_n<1>(v);
  return v[0];
}
// exact (erf) GELU and its derivative, for the tower backward's recomputed hidden (FV_EPI_GELU_GRAD, gelu_grad_mul_kernel):
// Phi(x) = (1 + erf(x / sqrt 2)) / 2, gelu = x Phi, gelu' = Phi + x phi
__device__ __forceinline__ void gelu_and_grad(float x, float& g, float& dg) {
  const float Phi = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
  const float phi = 0.3989422804014327f * __expf(-0.5f * x * x);
  g = x * Phi;
  dg = Phi + x * phi;
}
// The same pair for the tower backward's hot places (the recompute GEMM's epilogue touches 12 G hidden activations per B = 32 step): Phi from gelu2_n's packed
// minimax polynomial (max |error| 5e-5, both tails exact through the clamp), phi from ONE v_exp_f32 -- ~14 VALU issues per element instead of ~60 for erff + expf.
__device__ __forceinline__ void gelu_and_grad2(const f32x2 x, f32x2& g, f32x2& dg) {
  const float R = 4.625f;
  const f32x2 xc = {__builtin_amdgcn_fmed3f(x.x, -R, R), __builtin_amdgcn_fmed3f(x.y, -R, R)};
  const f32x2 x2 = xc * xc;
  f32x2 p = __builtin_elementwise_fma(x2, (f32x2){FV_GELU_C8, FV_GELU_C8}, (f32x2){FV_GELU_C7, FV_GELU_C7});
  p = __builtin_elementwise_fma(p, x2, (f32x2){FV_GELU_C6, FV_GELU_C6});
  p = __builtin_elementwise_fma(p, x2, (f32x2){FV_GELU_C5, FV_GELU_C5});
  p = __builtin_elementwise_fma(p, x2, (f32x2){FV_GELU_C4, FV_GELU_C4});
  p = __builtin_elementwise_fma(p, x2, (f32x2){FV_GELU_C3, FV_GELU_C3});
  p = __builtin_elementwise_fma(p, x2, (f32x2){FV_GELU_C2, FV_GELU_C2});
  p = __builtin_elementwise_fma(p, x2, (f32x2){FV_GELU_C1, FV_GELU_C1});
  p = __builtin_elementwise_fma(p, x2, (f32x2){FV_GELU_C0, FV_GELU_C0});
  f32x2 Phi = __builtin_elementwise_fma(xc, p, (f32x2){0.5f, 0.5f});
  Phi.x = __builtin_amdgcn_fmed3f(Phi.x, 0.f, 1.f);
  Phi.y = __builtin_amdgcn_fmed3f(Phi.y, 0.f, 1.f);
  const f32x2 xx = x * x;
  const f32x2 phi = {0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * xx.x), 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.7213475204444817f * xx.y)};
  g = x * Phi;
  dg = __builtin_elementwise_fma(x, phi, Phi);
}
// v[8] *= gelu'(4 y[8]), h[8] = gelu(4 y[8]): the fc2 input gradient's epilogue against the pre-activation / 4 the training forward stashed (FV_EPI_MUL_GELUP)
__device__ __forceinline__ void mul_gelu_grad8(float* v, const float* y, float* h) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    f32x2 g, dg;
    gelu_and_grad2((f32x2){4.0f * y[e], 4.0f * y[e + 1]}, g, dg);
    v[e] *= dg.x;
    v[e + 1] *= dg.y;
    h[e] = g.x;
    h[e + 1] = g.y;
  }
}
__device__ __forceinline__ void gelu_and_grad8(float* v, float* d) {   // v[8] -> gelu in place, d[8] = gelu'
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    f32x2 g, dg;
    gelu_and_grad2((f32x2){v[e], v[e + 1]}, g, dg);
    v[e] = g.x; v[e + 1] = g.y; d[e] = dg.x; d[e + 1] = dg.y;
  }
}
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// XCD-aware bijective remap of a 1-D grid: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a
// contiguous run of logical ids (neighbouring tiles then share one L2).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = b & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}

// A/B switches (FASTVLA_NO_GEMM256, FASTVLA_NO_FFN32, ...) exist ONLY in the tools build (`make AB=1` -> -DFASTVLA_AB_SWITCHES ->
// tools/bin/libfastvla_hip_ab.so, used by tools/*.sh): in the product library every shape has exactly ONE dispatch path, so no
// environment variable can route a user onto a kernel the -m gpu suite does not cover.
#include <stdlib.h>
inline const char* fv_ab_env(const char* name) {
#ifdef FASTVLA_AB_SWITCHES
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

#define FV_HIP_CHECK(expr)                                  \
  do {                                                      \
    hipError_t _e = (expr);                                 \
    if (_e != hipSuccess) return fv_hip_fail(_e, #expr);    \
  } while (0)

int fv_hip_fail(hipError_t e, const char* what);  // records message, returns FV_ERR_HIP
int fv_fail(int code, const char* fmt, ...);       // records message, returns code
