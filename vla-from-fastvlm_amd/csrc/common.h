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
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}
__device__ __forceinline__ float bf_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
  f[0] = bf_lo(u.x); f[1] = bf_hi(u.x); f[2] = bf_lo(u.y); f[3] = bf_hi(u.y);
  f[4] = bf_lo(u.z); f[5] = bf_hi(u.z); f[6] = bf_lo(u.w); f[7] = bf_hi(u.w);
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

#define FV_HIP_CHECK(expr)                                  \
  do {                                                      \
    hipError_t _e = (expr);                                 \
    if (_e != hipSuccess) return fv_hip_fail(_e, #expr);    \
  } while (0)

int fv_hip_fail(hipError_t e, const char* what);  // records message, returns FV_ERR_HIP
int fv_fail(int code, const char* fmt, ...);       // records message, returns code
