// head_kernels.hip -- the only trainable part of the path, all fp32 (the reference keeps it fp32 too):
//   action expert forward     reference fastvla/fastvlm_with_expert.py:23-38,50-54
//   MSE + hand-derived backward for the 12 head tensors   fastvla/modeling_fastvla.py:56, trainer.py:175
//   fused global-norm clip + AdamW on one flat buffer      trainer.py:60-66,178-180; lerobot configuration_fastvla.py:51-55
// 3.05 M parameters / ~18 MFLOP per sample: launch-latency bound, so the kernels are simple wave-per-output fp32
// dot products with coalesced weight reads; parameters and gradients are views into flat caller-owned buffers.
#include "kernels.h"

namespace fv {
namespace {

constexpr float LN_EPS = 1e-5f;

// ---- Philox4x32-10 (counter-based; (seed, offset) explicit so a step is reproducible) ----
__device__ __forceinline__ uint4 philox(uint4 c, uint2 k) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
  }
  return c;
}

// one wave per row: y = LN(x) * w + b, saves xhat and rstd
// pre_mean / pre_istd (may be null): the dataset normalisation of the input folded in, x <- (x - pre_mean) * pre_istd
// (LeRobot NormalizerProcessorStep, MEAN_STD; reference lerobot_fastvla/processor_fastvla.py:34-39)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                      const float* __restrict__ b, float* __restrict__ y,
                                                      float* __restrict__ xhat, float* __restrict__ rstd, int rows, int n,
                                                      const float* __restrict__ pre_mean, const float* __restrict__ pre_istd) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ldx;
  auto in = [&](int i) { return pre_mean ? (xr[i] - pre_mean[i]) * pre_istd[i] : xr[i]; };
  float s = 0.f;
  for (int i = lane; i < n; i += 64) s += in(i);
  const float mean = wave_sum(s) / (float)n;
  float q = 0.f;
  for (int i = lane; i < n; i += 64) { const float d = in(i) - mean; q += d * d; }
  const float rs = rsqrtf(wave_sum(q) / (float)n + LN_EPS);
  for (int i = lane; i < n; i += 64) {
    const float h = (in(i) - mean) * rs;
    xhat[(size_t)row * n + i] = h;
    y[(size_t)row * n + i] = h * w[i] + b[i];
  }
  if (lane == 0) rstd[row] = rs;
}

// y[b][n] = sum_k x[b][k] W[n][k] + bias[n]; one wave per n, 8 batch rows per pass.  act 1: z = pre-activation,
// y = silu(z).  y may be a strided view (ldy) -- the state branch writes straight into cat[:, feat:].
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ W, const float* __restrict__ bias,
                                                          float* __restrict__ y, int ldy, float* __restrict__ z, int B,
                                                          int N, int K, int act, const float* __restrict__ post_scale,
                                                          const float* __restrict__ post_shift) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b0 = blockIdx.y * 8;
  if (n >= N) return;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float* wr = W + (size_t)n * K;
  for (int k = lane; k < K; k += 64) {
    const float wv = wr[k];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += x[(size_t)min(b0 + j, B - 1) * ldx + k] * wv;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = wave_sum(acc[j]);
  if (lane == 0) {
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (b0 + j >= B) break;
      const float v = acc[j] + bv;
      if (act) { z[(size_t)(b0 + j) * N + n] = v; y[(size_t)(b0 + j) * ldy + n] = silu_f(v); }
      else y[(size_t)(b0 + j) * ldy + n] = post_scale ? v * post_scale[n] + post_shift[n] : v;   // action un-normalisation folded in
    }
  }
}

// dx[b][k] = sum_n dy[b][n] W[n][k].  Block = 32 k x 8 batch rows; its 256 threads are 8 n-lanes x 32 k: lane j sums n = j, j + 8, ... (N / 8
// steps instead of N: the first version's 32 blocks of N-long dependent loops took 196 us per call at B = 32, N = 1024, K = 1920 -- 3 % of
// the C3 rank-shape train step for 126 MFLOP), the eight partial sums are folded through LDS in a fixed order (bit-repeatable).
__global__ __launch_bounds__(256) void linear_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ W,
                                                             float* __restrict__ dx, int lddx, int B, int N, int K) {
  __shared__ float part[8][8][32];   // [n-lane][batch row][k]
  const int tid = threadIdx.x, kl = tid & 31, nl = tid >> 5;
  const int k = blockIdx.x * 32 + kl, kc = min(k, K - 1);
  const int b0 = blockIdx.y * 8;
  const float* dyr[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) dyr[j] = dy + (size_t)min(b0 + j, B - 1) * N;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
  for (int n = nl; n < N; n += 8) {
    const float wv = W[(size_t)n * K + kc];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] += dyr[j][n] * wv;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[nl][j][kl] = acc[j];
  __syncthreads();
  const int j = tid >> 5;   // one (batch row, k) per thread
  float sum = 0.f;
#pragma unroll
  for (int l = 0; l < 8; ++l) sum += part[l][j][kl];
  if (k < K && b0 + j < B) dx[(size_t)(b0 + j) * lddx + k] = sum;
}

// dW[n][k] = sum_b dy[b][n] x[b][k]; block = one n, 256 k
__global__ __launch_bounds__(256) void linear_bwd_dw_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                             int ldx, float* __restrict__ dW, int B, int N, int K) {
  const int k = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
  if (k >= K) return;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) acc += dy[(size_t)b * N + n] * x[(size_t)b * ldx + k];
  dW[(size_t)n * K + k] = acc;
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, float* __restrict__ db, int B, int N) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) acc += dy[(size_t)b * N + n];
  db[n] = acc;
}

// g <- silu'(z) * g * (mul ? mul : 1); src may be a strided view (ds = dcat[:, feat:])
__global__ __launch_bounds__(256) void silu_bwd_kernel(const float* __restrict__ gin, int ldg, const float* __restrict__ z,
                                                        const float* __restrict__ mul, float* __restrict__ gout, int B, int N) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * N) return;
  const int b = (int)(i / N), n = (int)(i % N);
  float g = gin[(size_t)b * ldg + n];
  if (mul) g *= mul[i];
  const float zz = z[i], sg = sigmoid_f(zz);
  gout[i] = g * sg * (1.0f + zz * (1.0f - sg));
}

// dropout after SiLU(LayerNorm): d2 = silu(n2) * keep/(1-p); mask holds the multiplier
__global__ __launch_bounds__(256) void silu_dropout_kernel(const float* __restrict__ n2, float* __restrict__ d2,
                                                            float* __restrict__ mask, long total, int training, float p,
                                                            uint64_t seed, uint64_t offset) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float mlt = 1.0f;
  if (training && p > 0.f) {
    const uint64_t ctr = (uint64_t)(i >> 2);
    const uint4 r = philox(make_uint4((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)),
                           make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const uint32_t rv[4] = {r.x, r.y, r.z, r.w};
    const float u = (float)(rv[i & 3] >> 8) * (1.0f / 16777216.0f);
    mlt = u >= p ? 1.0f / (1.0f - p) : 0.f;
  }
  mask[i] = mlt;
  d2[i] = silu_f(n2[i]) * mlt;
}

__global__ __launch_bounds__(256) void copy_cols_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst,
                                                         int ldd, int B, int N) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * N) return;
  const int b = (int)(i / N), n = (int)(i % N);
  dst[(size_t)b * ldd + n] = src[(size_t)b * lds + n];
}

// single block: loss = mean((a-t)^2), g = 2 (a-t) / (B*A)
__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ a, const float* __restrict__ t,
                                                   float* __restrict__ loss, float* __restrict__ g, int n, float loss_scale) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = a[i] - t[i];
    s += d * d;
    g[i] = loss_scale * 2.0f * d / (float)n;   // loss_scale (a power of two, 1 by default) scales every gradient downstream; the loss itself is not scaled
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *loss = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

// LN backward: wave per row -> dx (may be null); column sums dw, db done by ln_bwd_cols_kernel
__global__ __launch_bounds__(256) void ln_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                           const float* __restrict__ xhat, const float* __restrict__ rstd,
                                                           float* __restrict__ dx, int rows, int n) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s1 = 0.f, s2 = 0.f;
  for (int i = lane; i < n; i += 64) {
    const float g = dy[(size_t)row * n + i] * w[i];
    s1 += g;
    s2 += g * xhat[(size_t)row * n + i];
  }
  s1 = wave_sum(s1) / (float)n;
  s2 = wave_sum(s2) / (float)n;
  const float rs = rstd[row];
  for (int i = lane; i < n; i += 64) {
    const float g = dy[(size_t)row * n + i] * w[i];
    dx[(size_t)row * n + i] = rs * (g - s1 - xhat[(size_t)row * n + i] * s2);
  }
}

__global__ __launch_bounds__(256) void ln_bwd_cols_kernel(const float* __restrict__ dy, const float* __restrict__ xhat,
                                                           float* __restrict__ dw, float* __restrict__ db, int rows, int n) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  float a = 0.f, c = 0.f;
  for (int r = 0; r < rows; ++r) {
    const float g = dy[(size_t)r * n + j];
    a += g * xhat[(size_t)r * n + j];
    c += g;
  }
  dw[j] = a;
  db[j] = c;
}

// ---- AdamW + clip ----
// Global gradient norm in a FIXED summation order (per-block partials, then one block folds them): float atomics would make
// the norm -- and through the clip coefficient every parameter -- depend on arrival order, so two data-parallel replicas
// (or two runs) would drift apart by an ulp per step.
constexpr int SUMSQ_BLOCKS = 1024;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
  __shared__ float red[4];
  // 16-byte loads, four independent sums per thread (the unfrozen path's 2 GB gradient: 0.84 -> ~0.45 ms); a fixed order either way
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = g4[i];
    s0 += v.x * v.x; s1 += v.y * v.y; s2 += v.z * v.z; s3 += v.w * v.w;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; s0 += v * v; }
  float s = wave_sum((s0 + s1) + (s2 + s3));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[1 + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void sumsq_fold_kernel(float* __restrict__ partial, int nb) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) s += partial[1 + i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                     float* __restrict__ m, float* __restrict__ v, long n,
                                                     fv_adamw_hparams hp, float bc1, float bc2_sqrt,
                                                     const float* __restrict__ sumsq, float* __restrict__ norm_out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  // torch clip_grad_norm_: coef = clamp(max_norm / (norm + 1e-6), max = 1)
  const float norm = sqrtf(*sumsq) * hp.grad_scale;
  float coef = hp.grad_scale;
  if (hp.max_grad_norm > 0.f) coef *= fminf(hp.max_grad_norm / (norm + 1e-6f), 1.0f);
  if (i == 0 && norm_out) *norm_out = norm;
  if (i >= n) return;
  const float gi = g[i] * coef;
  const float pi = p[i] * (1.0f - hp.lr * hp.weight_decay);
  const float mi = m[i] * hp.beta1 + gi * (1.0f - hp.beta1);
  const float vi = v[i] * hp.beta2 + gi * gi * (1.0f - hp.beta2);
  const float denom = sqrtf(vi) / bc2_sqrt + hp.eps;
  p[i] = pi - (hp.lr / bc1) * mi / denom;
  m[i] = mi;
  v[i] = vi;
}

// y += x (x != null) or y *= *scale (x == null): gradient accumulation over micro-batches / an upstream loss gradient
__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, long n4,
                                                    const float* __restrict__ scale) {
  const float sc = scale ? *scale : 1.0f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 a = reinterpret_cast<float4*>(y)[i];
    if (x) {
      const float4 b = reinterpret_cast<const float4*>(x)[i];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    } else {
      a.x *= sc; a.y *= sc; a.z *= sc; a.w *= sc;
    }
    reinterpret_cast<float4*>(y)[i] = a;
  }
}

inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

}  // namespace

HeadOffsets head_offsets(const HeadDims& d) {
  HeadOffsets r;
  const int64_t sz[12] = {d.ds, d.ds, (int64_t)d.hid * d.ds, d.hid, (int64_t)d.fus * (d.feat + d.hid), d.fus,
                          d.fus, d.fus, (int64_t)d.fus * d.fus, d.fus, (int64_t)d.da * d.fus, d.da};
  int64_t o = 0;
  for (int i = 0; i < 12; ++i) {
    r.o[i] = o;
    o += (sz[i] + 3) / 4 * 4;  // keep every tensor 16-byte aligned inside the flat buffer
  }
  r.o[12] = o;
  return r;
}

namespace {
struct Saved {
  float *n0, *xh0, *rstd0, *z1, *cat, *z2, *xh2, *rstd2, *n2, *d2, *mask, *z3, *a3;
  size_t total;
};
Saved carve(const HeadDims& d, int B, float* base) {
  Saved s;
  size_t o = 0;
  auto take = [&](size_t n) { float* p = base ? base + o : nullptr; o += (n + 3) / 4 * 4; return p; };
  s.n0 = take((size_t)B * d.ds); s.xh0 = take((size_t)B * d.ds); s.rstd0 = take(B);
  s.z1 = take((size_t)B * d.hid); s.cat = take((size_t)B * (d.feat + d.hid)); s.z2 = take((size_t)B * d.fus);
  s.xh2 = take((size_t)B * d.fus); s.rstd2 = take(B); s.n2 = take((size_t)B * d.fus); s.d2 = take((size_t)B * d.fus);
  s.mask = take((size_t)B * d.fus); s.z3 = take((size_t)B * d.fus); s.a3 = take((size_t)B * d.fus);
  s.total = o;
  return s;
}
}  // namespace

size_t adamw_scratch_bytes() { return (size_t)(SUMSQ_BLOCKS + 4) * sizeof(float); }

size_t head_saved_bytes(const HeadDims& d, int B) { return carve(d, B, nullptr).total * sizeof(float); }

size_t head_bwd_scratch_bytes(const HeadDims& d, int B) {
  const size_t wmax = (size_t)(d.feat + d.hid > d.fus ? d.feat + d.hid : d.fus);
  return ((size_t)B * d.da + 2 * (size_t)B * wmax + 64) * sizeof(float);
}

int launch_head_forward(const HeadDims& d, const float* P, const float* pooled, const float* states, int B,
                        int training, float drop_p, uint64_t seed, uint64_t offset, float* actions, float* saved,
                        hipStream_t s, const HeadIoNorm* io) {
  if (!P || !pooled || !states || !actions || !saved) return fv_fail(FV_ERR_ARG, "head_forward: null pointer");
  if (B <= 0) return fv_fail(FV_ERR_ARG, "head_forward: B must be positive");
  if (drop_p < 0.f || drop_p >= 1.f) return fv_fail(FV_ERR_ARG, "head_forward: dropout p out of range");
  const HeadOffsets ho = head_offsets(d);
  const Saved sv = carve(d, B, saved);
  const int cw = d.feat + d.hid;
  const dim3 blk(256);
  const float* sm = io ? io->state_mean : nullptr;
  const float* si = io ? io->state_istd : nullptr;
  // the action un-normalisation belongs to inference only: training computes its loss against NORMALISED targets
  const float* as = io && !training ? io->action_std : nullptr;
  const float* am = io && !training ? io->action_mean : nullptr;
  training &= 1;   // bit 0: dropout active; any non-zero value: actions stay in normalised space
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(cdiv(B, 4)), blk, 0, s, states, d.ds, P + ho.o[0], P + ho.o[1], sv.n0, sv.xh0, sv.rstd0, B, d.ds, sm, si);
  hipLaunchKernelGGL(copy_cols_kernel, dim3(cdiv((long)B * d.feat, 256)), blk, 0, s, pooled, d.feat, sv.cat, cw, B, d.feat);
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(cdiv(d.hid, 4), cdiv(B, 8)), blk, 0, s, sv.n0, d.ds, P + ho.o[2], P + ho.o[3], sv.cat + d.feat, cw, sv.z1, B, d.hid, d.ds, 1, (const float*)nullptr, (const float*)nullptr);
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(cdiv(d.fus, 4), cdiv(B, 8)), blk, 0, s, sv.cat, cw, P + ho.o[4], P + ho.o[5], sv.z2, d.fus, (float*)nullptr, B, d.fus, cw, 0, (const float*)nullptr, (const float*)nullptr);
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(cdiv(B, 4)), blk, 0, s, sv.z2, d.fus, P + ho.o[6], P + ho.o[7], sv.n2, sv.xh2, sv.rstd2, B, d.fus,
                     (const float*)nullptr, (const float*)nullptr);
  hipLaunchKernelGGL(silu_dropout_kernel, dim3(cdiv((long)B * d.fus, 256)), blk, 0, s, sv.n2, sv.d2, sv.mask, (long)B * d.fus, training, drop_p, seed, offset);
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(cdiv(d.fus, 4), cdiv(B, 8)), blk, 0, s, sv.d2, d.fus, P + ho.o[8], P + ho.o[9], sv.a3, d.fus, sv.z3, B, d.fus, d.fus, 1, (const float*)nullptr, (const float*)nullptr);
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(cdiv(d.da, 4), cdiv(B, 8)), blk, 0, s, sv.a3, d.fus, P + ho.o[10], P + ho.o[11], actions, d.da, (float*)nullptr, B, d.da, d.fus, 0, as, am);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_head_backward(const HeadDims& d, const float* P, const float* grad_actions, const float* actions,
                         const float* targets, int B, float drop_p, const float* saved, float* loss, float* G,
                         float* scratch, hipStream_t s, float* d_pooled, float loss_scale) {
  if (!P || !saved || !G || !scratch) return fv_fail(FV_ERR_ARG, "head_backward: null pointer");
  if (!grad_actions && (!actions || !targets || !loss)) return fv_fail(FV_ERR_ARG, "head_backward: need grad_actions or (actions, targets, loss)");
  if (B <= 0) return fv_fail(FV_ERR_ARG, "head_backward: B must be positive");
  (void)drop_p;  // the multiplier keep/(1-p) is stored in saved.mask
  const HeadOffsets ho = head_offsets(d);
  const Saved sv = carve(d, B, const_cast<float*>(saved));
  const int cw = d.feat + d.hid;
  const size_t wmax = (size_t)(cw > d.fus ? cw : d.fus);
  float* ga_own = scratch;
  float* g1 = ga_own + ((size_t)B * d.da + 3) / 4 * 4;
  float* g2 = g1 + (size_t)B * wmax;
  const dim3 blk(256);
  const unsigned b8 = cdiv(B, 8);
  const float* ga = grad_actions;
  if (!ga) {
    hipLaunchKernelGGL(mse_kernel, dim3(1), blk, 0, s, actions, targets, loss, ga_own, B * d.da, loss_scale);
    ga = ga_own;
  }
  // action_head
  hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3(cdiv(d.fus, 256), d.da), blk, 0, s, ga, sv.a3, d.fus, G + ho.o[10], B, d.da, d.fus);
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(d.da, 256)), blk, 0, s, ga, G + ho.o[11], B, d.da);
  hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(cdiv(d.fus, 32), b8), blk, 0, s, ga, P + ho.o[10], g1, d.fus, B, d.da, d.fus);
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(cdiv((long)B * d.fus, 256)), blk, 0, s, g1, d.fus, sv.z3, (const float*)nullptr, g1, B, d.fus);
  // fusion.4
  hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3(cdiv(d.fus, 256), d.fus), blk, 0, s, g1, sv.d2, d.fus, G + ho.o[8], B, d.fus, d.fus);
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(d.fus, 256)), blk, 0, s, g1, G + ho.o[9], B, d.fus);
  hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(cdiv(d.fus, 32), b8), blk, 0, s, g1, P + ho.o[8], g2, d.fus, B, d.fus, d.fus);
  // dropout multiplier, SiLU, LayerNorm (fusion.1)
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(cdiv((long)B * d.fus, 256)), blk, 0, s, g2, d.fus, sv.n2, sv.mask, g2, B, d.fus);
  hipLaunchKernelGGL(ln_bwd_cols_kernel, dim3(cdiv(d.fus, 256)), blk, 0, s, g2, sv.xh2, G + ho.o[6], G + ho.o[7], B, d.fus);
  hipLaunchKernelGGL(ln_bwd_rows_kernel, dim3(cdiv(B, 4)), blk, 0, s, g2, P + ho.o[6], sv.xh2, sv.rstd2, g1, B, d.fus);
  // fusion.0
  hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3(cdiv(cw, 256), d.fus), blk, 0, s, g1, sv.cat, cw, G + ho.o[4], B, d.fus, cw);
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(d.fus, 256)), blk, 0, s, g1, G + ho.o[5], B, d.fus);
  hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(cdiv(cw, 32), b8), blk, 0, s, g1, P + ho.o[4], g2, cw, B, d.fus, cw);
  // dcat = [d pooled | d state branch]: the first `feat` columns are what an unfrozen backbone's backward starts from
  if (d_pooled) hipLaunchKernelGGL(copy_cols_kernel, dim3(cdiv((long)B * d.feat, 256)), blk, 0, s, g2, cw, d_pooled, d.feat, B, d.feat);
  // state branch: ds = dcat[:, feat:], SiLU, Linear, LayerNorm
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(cdiv((long)B * d.hid, 256)), blk, 0, s, g2 + d.feat, cw, sv.z1, (const float*)nullptr, g1, B, d.hid);
  hipLaunchKernelGGL(linear_bwd_dw_kernel, dim3(cdiv(d.ds, 256), d.hid), blk, 0, s, g1, sv.n0, d.ds, G + ho.o[2], B, d.hid, d.ds);
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(d.hid, 256)), blk, 0, s, g1, G + ho.o[3], B, d.hid);
  hipLaunchKernelGGL(linear_bwd_dx_kernel, dim3(cdiv(d.ds, 32), b8), blk, 0, s, g1, P + ho.o[2], g2, d.ds, B, d.hid, d.ds);
  hipLaunchKernelGGL(ln_bwd_cols_kernel, dim3(cdiv(d.ds, 256)), blk, 0, s, g2, sv.xh0, G + ho.o[0], G + ho.o[1], B, d.ds);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_axpy(float* y, const float* x, int64_t n, const float* scale_dev, hipStream_t s) {
  // the flat head buffers keep every tensor 16-byte aligned and their total a multiple of 4 elements (head_offsets)
  if (n % 4 || ((uintptr_t)y & 15) || ((uintptr_t)x & 15)) return fv_fail(FV_ERR_ARG, "axpy: buffers must be 16-byte aligned, n %% 4 == 0");
  const unsigned nb = cdiv(n / 4, 256);
  hipLaunchKernelGGL(axpy_kernel, dim3(nb < 1024 ? nb : 1024), dim3(256), 0, s, y, x, (long)(n / 4), scale_dev);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_adamw_clip(float* p, const float* g, float* m, float* v, int64_t n, const fv_adamw_hparams& hp,
                      int64_t step, float* norm_scratch, float* grad_norm_out, hipStream_t s) {
  if (!p || !g || !m || !v || !norm_scratch) return fv_fail(FV_ERR_ARG, "adamw: null pointer");
  if (n <= 0 || step < 1) return fv_fail(FV_ERR_ARG, "adamw: n and step must be positive");
  const unsigned nb = (unsigned)((n + 255) / 256);
  const unsigned nb4 = (unsigned)((n / 4 + 255) / 256 > 0 ? (n / 4 + 255) / 256 : 1);
  const unsigned nsb = nb4 < (unsigned)SUMSQ_BLOCKS ? nb4 : (unsigned)SUMSQ_BLOCKS;
  hipLaunchKernelGGL(sumsq_kernel, dim3(nsb), dim3(256), 0, s, g, (long)n, norm_scratch);
  hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, s, norm_scratch, (int)nsb);
  const float bc1 = (float)(1.0 - pow((double)hp.beta1, (double)step));
  const float bc2 = (float)sqrt(1.0 - pow((double)hp.beta2, (double)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(nb), dim3(256), 0, s, p, g, m, v, (long)n, hp, bc1, bc2, norm_scratch, grad_norm_out);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

}  // namespace fv
