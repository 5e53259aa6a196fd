// train_kernels.hip -- kernels of the UNFROZEN-backbone training slice (decoder + projector trainable, tower
// frozen, image tokens spliced in front of the text).  The reference makes this path unreachable (model/fastvlm_adapter.py:501 wraps
// the backbone in no_grad whatever `freeze_backbone` says, fastvla/configuration_fastvla.py:23); the arithmetic restated here is
// torch autograd's over [site] transformers/models/qwen2/modeling_qwen2.py (RMSNorm :247-252, RoPE :105-135, attention :150-172,
// SwiGLU MLP :35-48) and the mm_projector ([site] fast_vlm/modeling_fast_vlm.py:39-56), pinned by tests against oracle/qwen2.py
// under torch.autograd.
//
// Every contraction of the backward pass runs on the SAME MFMA GEMM kernels as the forward (gemm_bf16.hip): out = A . W^T needs
//   dgrad  dX[M,K]  = dY[M,N] . W[N,K]        ->  A = dY (split bf16, [hi | lo]),  "W" = W^T  [K][N]   (a transposed weight copy)
//   wgrad  dW[N,K]  = dY^T[N,M] . X[M,K]      ->  A = dY^T (split bf16, [N][2 Mp]), "W" = X^T  [K][Mp]  (contraction over the rows)
// so what this file provides is the HBM-bound glue around them -- fp32 -> split-bf16 operand forms (row-major and transposed), the
// backward of RMSNorm / SwiGLU / GELU / RoPE+attention / embedding / pooling -- with fp32 everywhere outside the MFMA operands and a
// fixed summation order (no float atomics: gradients are bit-reproducible run to run, which the pipelined-equals-serial and
// bucketed-equals-unbucketed tests rely on).
#include "kernels.h"

#ifndef FV_TRY_RC
#define FV_TRY_RC(expr) do { const int rc_ = (expr); if (rc_ != FV_OK) return rc_; } while (0)
#endif

namespace fv {
namespace {

constexpr int TP = 64;   // transpose tile

__device__ __forceinline__ void load8(const float* p, float* v) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(float* p, const float* v) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// fp32 [R][C] (row stride ldi) -> bf16 [R][ldo]: the bf16 value at column c and, lo_off > 0, its bf16 remainder at lo_off + c
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ in, int ldi, bf16_t* __restrict__ out, int ldo, int lo_off,
                                                          long R, int C) {
  const int c8 = C >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < R * c8; i += (long)gridDim.x * 256) {
    const long r = i / c8;
    const int c = (int)(i % c8) * 8;
    float v[8], h[8], l[8];
    load8(in + r * ldi + c, v);
    const uint4 hv = pack8(v);
    *reinterpret_cast<uint4*>(out + r * ldo + c) = hv;
    if (lo_off) {
      unpack8(hv, h);
#pragma unroll
      for (int e = 0; e < 8; ++e) l[e] = v[e] - h[e];
      *reinterpret_cast<uint4*>(out + r * ldo + lo_off + c) = pack8(l);
    }
  }
}

// fp32 [R][C] -> the fp8 remainder bytes of the "hi + lo8" operand form: out row r (ldo bf16 units) gets, at byte offset 2C, fp8((x - bf16(x)) * 2^8)
__global__ __launch_bounds__(256) void lo8_rows_kernel(const float* __restrict__ in, int ldi, bf16_t* __restrict__ out, int ldo, long R, int C) {
  const int c8 = C >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < R * c8; i += (long)gridDim.x * 256) {
    const long r = i / c8;
    const int c = (int)(i % c8) * 8;
    float v[8], h[8], l[8];
    load8(in + r * ldi + c, v);
    unpack8(pack8(v), h);
#pragma unroll
    for (int e = 0; e < 8; ++e) l[e] = v[e] - h[e];
    *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(out + r * ldo) + 2 * (size_t)C + c) = pack_lo8(l);
  }
}

__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    float v[8];
    unpack8(reinterpret_cast<const uint4*>(in)[i], v);
    store8(out + i * 8, v);
  }
}

// in [R][C] (row stride ldi; fp32 or bf16) -> out bf16 [C][ldo] with out[c][r] = bf16(in[r][c]), rows r in [R, Rp) zero, and
// (lo_off > 0, fp32 input) out[c][lo_off + r] = the bf16 remainder.  64 x 64 tiles through LDS; a wave writes 8 columns x 128 bytes.
template <typename TIn>
__global__ __launch_bounds__(256) void transpose_kernel(const TIn* __restrict__ in, int ldi, bf16_t* __restrict__ out, int ldo, int lo_off,
                                                         int R, int Rp, int C) {
  __shared__ float tile[TP][TP + 1];
  const int r0 = blockIdx.x * TP, c0 = blockIdx.y * TP, tid = threadIdx.x;
  // load: thread -> (row = tid / 8 + 32 k, 8 columns at (tid % 8) * 8)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int r = (tid >> 3) + 32 * k, c = (tid & 7) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r0 + r < R && c0 + c < C) {
      if constexpr (sizeof(TIn) == 4) load8(reinterpret_cast<const float*>(in) + (size_t)(r0 + r) * ldi + c0 + c, v);
      else unpack8(*reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(in) + (size_t)(r0 + r) * ldi + c0 + c), v);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[r][c + e] = v[e];
  }
  __syncthreads();
  // store: thread -> (column = tid / 8 + 32 k, 8 rows at (tid % 8) * 8): 8 lanes cover one column's 64 rows = 128 contiguous bytes
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = (tid >> 3) + 32 * k, r = (tid & 7) * 8;
    if (c0 + c >= C || r0 + r >= Rp) continue;
    float v[8], h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[r + e][c];
    const uint4 hv = pack8(v);
    *reinterpret_cast<uint4*>(out + (size_t)(c0 + c) * ldo + r0 + r) = hv;
    if (lo_off) {
      unpack8(hv, h);
#pragma unroll
      for (int e = 0; e < 8; ++e) l[e] = v[e] - h[e];
      *reinterpret_cast<uint4*>(out + (size_t)(c0 + c) * ldo + lo_off + r0 + r) = pack8(l);
    }
  }
}

// the same tile transpose with fp16 output (11 significant bits: the wgrad operands): KIND 0 = fp32 input, 1 = bf16, 2 = split bf16 whose halves
// (lo_in columns apart) are summed before the one rounding; saturating casts, clamps counted
// (rows_out != null: the same values also as fp16 ROWS [R][ldro] -- a gradient's dgrad and wgrad operands from one read)
template <int KIND>
__global__ __launch_bounds__(256) void transpose_f16_kernel(const void* __restrict__ in_v, int ldi, int lo_in, bf16_t* __restrict__ out, int ldo, int R, int Rp, int C,
                                                             unsigned* __restrict__ sat, bf16_t* __restrict__ rows_out, int ldro) {
  __shared__ float tile[TP][TP + 1];
  const int r0 = blockIdx.x * TP, c0 = blockIdx.y * TP, tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int r = (tid >> 3) + 32 * k, c = (tid & 7) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r0 + r < R && c0 + c < C) {
      if constexpr (KIND == 0) load8(static_cast<const float*>(in_v) + (size_t)(r0 + r) * ldi + c0 + c, v);
      else {
        const bf16_t* p = static_cast<const bf16_t*>(in_v) + (size_t)(r0 + r) * ldi + c0 + c;
        if constexpr (KIND == 3) unpack8_h(*reinterpret_cast<const uint4*>(p), v);   // fp16 rows (the one-pass fp16 training forward's normed operands)
        else unpack8(*reinterpret_cast<const uint4*>(p), v);
        if constexpr (KIND == 2) {
          float l[8];
          unpack8(*reinterpret_cast<const uint4*>(p + lo_in), l);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += l[e];
        }
      }
      if (rows_out) *reinterpret_cast<uint4*>(rows_out + (size_t)(r0 + r) * ldro + c0 + c) = pack8_h(v);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[r][c + e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = (tid >> 3) + 32 * k, r = (tid & 7) * 8;
    if (c0 + c >= C || r0 + r >= Rp) continue;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[r + e][c];
    count_f16_sat8(v, sat);
    *reinterpret_cast<uint4*>(out + (size_t)(c0 + c) * ldo + r0 + r) = pack8_h(v);
  }
}

// rows -> fp16 rows (the gradient operand of the one-pass fp16 dgrad -- and, rows [R, Rp) written as zeros, either operand of the TN wgrad, whose
// contraction runs over whole 64-row K-tiles): KIND as transpose_f16_kernel; saturating, clamps counted
template <int KIND>
__global__ __launch_bounds__(256) void rows_f16_kernel(const void* __restrict__ in_v, int ldi, int lo_in, bf16_t* __restrict__ out, int ldo, long R, long Rp, int C,
                                                        unsigned* __restrict__ sat) {
  const int c8 = C >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < Rp * c8; i += (long)gridDim.x * 256) {
    const long r = i / c8;
    const int c = (int)(i % c8) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < R) {
      if constexpr (KIND == 0) load8(static_cast<const float*>(in_v) + r * ldi + c, v);
      else {
        const bf16_t* p = static_cast<const bf16_t*>(in_v) + r * ldi + c;
        unpack8(*reinterpret_cast<const uint4*>(p), v);
        if constexpr (KIND == 2) {
          float l[8];
          unpack8(*reinterpret_cast<const uint4*>(p + lo_in), l);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += l[e];
        }
      }
      count_f16_sat8(v, sat);
    }
    *reinterpret_cast<uint4*>(out + r * ldo + c) = pack8_h(v);
  }
}

// fv_train_commit as one launch (CommitDesc, kernels.h): the 99 f32->bf16 casts, 72 weight transposes and 167 vector copies of one optimiser step were
// ~340 launches of 3-10 us and read the bf16 weights back for the transposes; here the fp32 master is read once
__global__ __launch_bounds__(256) void commit_kernel(const CommitDesc* __restrict__ desc, int ndesc, const float* __restrict__ flat, int f16t,
                                                      unsigned* __restrict__ sat) {
  __shared__ float tile[TP][TP + 1];
  const int tid = threadIdx.x, bt = blockIdx.x;
  int lo = 0, hi = ndesc - 1;          // the last descriptor whose tile0 <= bt
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].tile0 <= bt) lo = mid; else hi = mid - 1;
  }
  const CommitDesc d = desc[lo];
  const int lt = bt - d.tile0;
  const float* src = flat + d.src_off;
  if (!d.is_mat) {
    const long base = (long)lt * (TP * TP);
    float* dst = static_cast<float*>(d.dst);
    for (int i = tid; i < TP * TP; i += 256)
      if (base + i < d.cols) dst[base + i] = src[base + i];
    return;
  }
  const int tcols = (d.cols + TP - 1) / TP;
  const int r0 = (lt / tcols) * TP, c0 = (lt % tcols) * TP;
  bf16_t* dst = static_cast<bf16_t*>(d.dst);
  const bool tr = d.dstT16 != nullptr;   // (both transposed forms exist or neither)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int r = (tid >> 3) + 32 * k, c = (tid & 7) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r0 + r < d.rows && c0 + c < d.cols) {
      load8(src + (size_t)(r0 + r) * d.cols + c0 + c, v);
      const uint4 hv = pack8(v);
      *reinterpret_cast<uint4*>(dst + (size_t)(r0 + r) * d.cols + c0 + c) = hv;
      unpack8(hv, v);   // the transposed copy holds the ROUNDED weight (what the forward multiplies by)
      if (d.dst16) {    // ... and so does the fp16 row copy of the one-pass fp16 training forward (down carries 2^4: FV_EPI_SWIGLU_F16 gives it up)
        float w16[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w16[e] = v[e] * d.scale16;
        count_f16_sat8(w16, sat);
        *reinterpret_cast<uint4*>(static_cast<bf16_t*>(d.dst16) + (size_t)(r0 + r) * d.cols + c0 + c) = pack8_h(w16);
      }
    }
    if (tr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) tile[r][c + e] = v[e];
    }
  }
  if (!tr) return;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = (tid >> 3) + 32 * k, r = (tid & 7) * 8;
    if (c0 + c >= d.cols || r0 + r >= d.rows) continue;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[r + e][c];
    if (f16t) {
      count_f16_sat8(v, sat);
      *reinterpret_cast<uint4*>(static_cast<bf16_t*>(d.dstT16) + (size_t)(c0 + c) * d.rows + r0 + r) = pack8_h(v);
    } else {
      *reinterpret_cast<uint4*>(static_cast<bf16_t*>(d.dstTb) + (size_t)(c0 + c) * d.rows + r0 + r) = pack8(v);
    }
  }
}

// (SwiGLU forward: in the gate/up GEMM's epilogue, FV_EPI_SWIGLU_SPLIT + GemmArgs::stash.)  gu [rows][2I] has its columns in the packed weight's
// order: 16-column groups [8 gate | 8 up] of the outputs 8j .. 8j+7.
// dgu (same interleaved layout) from dact [rows][I]:  dgate = dact * up * silu'(gate), dup = dact * silu(gate).  out_split == null: fp32,
// written over gu; else split bf16 [rows][4I] = [hi 2I | lo 2I]: the dgrad GEMM's operand as it is, and (transposed) the wgrad's
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(float* __restrict__ gu, const float* __restrict__ dact, long rows, int I, bf16_t* __restrict__ out_split) {
  const int j8 = I >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < rows * j8; i += (long)gridDim.x * 256) {
    const long r = i / j8;
    const int j = (int)(i % j8);
    float g[8], u[8], d[8], dg[8], du[8];
    float* gp = gu + r * 2 * I + 16 * j;
    load8(gp, g);
    load8(gp + 8, u);
    load8(dact + r * I + 8 * j, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float sg = 1.0f / (1.0f + __expf(-g[e]));
      dg[e] = d[e] * u[e] * sg * (1.0f + g[e] * (1.0f - sg));
      du[e] = d[e] * g[e] * sg;
    }
    if (out_split) {
      bf16_t* op = out_split + r * 4 * I + 16 * j;
      float h[8], l[8];
      uint4 hv = pack8(dg);
      *reinterpret_cast<uint4*>(op) = hv;
      unpack8(hv, h);
#pragma unroll
      for (int e = 0; e < 8; ++e) l[e] = dg[e] - h[e];
      *reinterpret_cast<uint4*>(op + 2 * I) = pack8(l);
      hv = pack8(du);
      *reinterpret_cast<uint4*>(op + 8) = hv;
      unpack8(hv, h);
#pragma unroll
      for (int e = 0; e < 8; ++e) l[e] = du[e] - h[e];
      *reinterpret_cast<uint4*>(op + 2 * I + 8) = pack8(l);
    } else {
      store8(gp, dg);
      store8(gp + 8, du);
    }
  }
}

// The same backward with BOTH of its consumers' operands written directly (the one-pass fp16 arithmetic, fv_train_set_options defaults): dgu as fp16
// rows [rows][2I] (the dgrad's A operand) and as fp16 columns outT [2I][Rp] (the wgrad's, rows in [rows, Rp) zero) -- no fp32 or split copy of
// dgu, no separate rounding and transposing passes.  A block owns 64 rows x 64 dgu columns (4 [8 gate | 8 up] groups); thread -> (row tid / 4,
// group tid % 4) on the way in, (column tid / 8 + 32 k, 8 rows) on the way out through a 64 x 64 fp32 LDS tile.
template <bool G16>   // G16: the kept gate/up accumulators are fp16 (GemmArgs::stash_f16)
__global__ __launch_bounds__(256) void swiglu_bwd_f16_kernel(const void* __restrict__ gu_v, const float* __restrict__ dact, int rows, int Rp, int I,
                                                              bf16_t* __restrict__ out_rows, bf16_t* __restrict__ outT, unsigned* __restrict__ sat,
                                                              bf16_t* __restrict__ actT) {
  // actT != null: also act = silu(gate) * up of the same tile as fp16 columns [I][Rp] -- the down projection's wgrad operand, recomputed here
  // from the accumulators this kernel reads anyway instead of transposed from a kept copy by a pass of its own
  __shared__ float tile[TP][TP + 1];
  __shared__ float tact[TP][TP / 2 + 1];
  const int r0 = blockIdx.x * TP, c0 = blockIdx.y * TP, tid = threadIdx.x, C = 2 * I;
  {
    const int r = tid >> 2, c = (tid & 3) * 16;
    float dg[8], du[8], av[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dg[e] = du[e] = av[e] = 0.f;
    if (r0 + r < rows && c0 + c < C) {
      float g[8], u[8], d[8];
      if constexpr (G16) {
        const bf16_t* gp = static_cast<const bf16_t*>(gu_v) + (size_t)(r0 + r) * C + c0 + c;
        unpack8_h(*reinterpret_cast<const uint4*>(gp), g);
        unpack8_h(*reinterpret_cast<const uint4*>(gp + 8), u);
      } else {
        const float* gp = static_cast<const float*>(gu_v) + (size_t)(r0 + r) * C + c0 + c;
        load8(gp, g);
        load8(gp + 8, u);
      }
      load8(dact + (size_t)(r0 + r) * I + ((c0 + c) >> 1), d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sg = 1.0f / (1.0f + __expf(-g[e]));
        dg[e] = d[e] * u[e] * sg * (1.0f + g[e] * (1.0f - sg));
        du[e] = d[e] * g[e] * sg;
        av[e] = silu_f(g[e]) * u[e];   // exactly the forward's value (the epilogue's silu_f on the same accumulators)
      }
      count_f16_sat8(dg, sat);
      count_f16_sat8(du, sat);
      bf16_t* op = out_rows + (size_t)(r0 + r) * C + c0 + c;
      *reinterpret_cast<uint4*>(op) = pack8_h(dg);
      *reinterpret_cast<uint4*>(op + 8) = pack8_h(du);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { tile[r][c + e] = dg[e]; tile[r][c + 8 + e] = du[e]; }
    if (actT) {
#pragma unroll
      for (int e = 0; e < 8; ++e) tact[r][(c >> 1) + e] = av[e];
    }
  }
  __syncthreads();
  if (actT) {   // 32 act columns x 64 rows: thread -> (column tid / 8, 8 rows)
    const int c = tid >> 3, r = (tid & 7) * 8, ac = (c0 >> 1) + c;
    if (ac < I && r0 + r < Rp) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tact[r + e][c];
      count_f16_sat8(v, sat);
      *reinterpret_cast<uint4*>(actT + (size_t)ac * Rp + r0 + r) = pack8_h(v);
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c = (tid >> 3) + 32 * k, r = (tid & 7) * 8;
    if (c0 + c >= C || r0 + r >= Rp) continue;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[r + e][c];
    *reinterpret_cast<uint4*>(outT + (size_t)(c0 + c) * Rp + r0 + r) = pack8_h(v);
  }
}

// The form the TN wgrad wants: d gate/up AND act = silu(gate) * up as fp16 ROWS only ([Rp][2I] / [Rp][I], rows [rows, Rp) zero -- the TN
// contraction walks whole 64-row K-tiles): no LDS, no transposed outputs; one thread = one 8-output group of one row
template <bool G16>
__global__ __launch_bounds__(256) void swiglu_bwd_rows_kernel(const void* __restrict__ gu_v, const float* __restrict__ dact, long rows, long Rp, int I,
                                                               bf16_t* __restrict__ dgu_rows, bf16_t* __restrict__ act_rows, unsigned* __restrict__ sat) {
  const int j8 = I >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < Rp * j8; i += (long)gridDim.x * 256) {
    const long r = i / j8;
    const int j = (int)(i % j8);
    float dg[8], du[8], av[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) dg[e] = du[e] = av[e] = 0.f;
    if (r < rows) {
      float g[8], u[8], d[8];
      if constexpr (G16) {
        const bf16_t* gp = static_cast<const bf16_t*>(gu_v) + r * 2 * I + 16 * j;
        unpack8_h(*reinterpret_cast<const uint4*>(gp), g);
        unpack8_h(*reinterpret_cast<const uint4*>(gp + 8), u);
      } else {
        const float* gp = static_cast<const float*>(gu_v) + r * 2 * I + 16 * j;
        load8(gp, g);
        load8(gp + 8, u);
      }
      load8(dact + r * I + 8 * j, d);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sg = 1.0f / (1.0f + __expf(-g[e]));
        dg[e] = d[e] * u[e] * sg * (1.0f + g[e] * (1.0f - sg));
        du[e] = d[e] * g[e] * sg;
        av[e] = silu_f(g[e]) * u[e];
      }
      count_f16_sat8(dg, sat);
      count_f16_sat8(du, sat);
      count_f16_sat8(av, sat);
    }
    bf16_t* op = dgu_rows + r * 2 * I + 16 * j;
    *reinterpret_cast<uint4*>(op) = pack8_h(dg);
    *reinterpret_cast<uint4*>(op + 8) = pack8_h(du);
    *reinterpret_cast<uint4*>(act_rows + r * I + 8 * j) = pack8_h(av);
  }
}

// exact-erf GELU (nn.GELU default; [site] fast_vlm/modeling_fast_vlm.py:47): h = gelu(pre) as split bf16, and its backward in place
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ pre, bf16_t* __restrict__ out, int ldo, int lo_off, long R, int C) {
  const int c8 = C >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < R * c8; i += (long)gridDim.x * 256) {
    const long r = i / c8;
    const int c = (int)(i % c8) * 8;
    float v[8], h[8], l[8];
    load8(pre + r * C + c, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752f));
    const uint4 hv = pack8(v);
    *reinterpret_cast<uint4*>(out + r * ldo + c) = hv;
    unpack8(hv, h);
#pragma unroll
    for (int e = 0; e < 8; ++e) l[e] = v[e] - h[e];
    *reinterpret_cast<uint4*>(out + r * ldo + lo_off + c) = pack8(l);
  }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(float* __restrict__ dh, const float* __restrict__ pre, long n8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    float d[8], x[8];
    load8(dh + i * 8, d);
    load8(pre + i * 8, x);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float cdf = 0.5f * (1.0f + erff(x[e] * 0.70710678118654752f));
      const float pdf = 0.3989422804014327f * __expf(-0.5f * x[e] * x[e]);
      d[e] *= cdf + x[e] * pdf;
    }
    store8(dh + i * 8, d);
  }
}

// RMSNorm backward.  y = w x r, r = rsqrt(mean(x^2) + eps):  dx = dres + r w dy - (r^3 / H) x sum_i(w_i dy_i x_i);  dw_i = sum_rows dy_i x_i r.
// One wave per row, `rpw` consecutive rows per wave (4: 2560 waves at 10 240 rows -- with 16 the launch was 160 blocks of serial, latency-exposed
// rows: 57 us for 147 MB); the row's x and dy stay in registers between the two passes; a lane keeps the dw partial sums of its columns in
// registers across its rows and every WAVE writes one partial row dw_part[wave][H] (summed in a fixed order by colsum_kernel).  H <= 512 NCH.
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ dy,
                                                           const float* __restrict__ dres, float* __restrict__ dx, float* __restrict__ dw_part,
                                                           long rows, int H, float eps, int rpw) {
  const int lane = threadIdx.x & 63;
  const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  float acc[NCH][8], wv8[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int i = lane * 8 + 512 * c;
#pragma unroll
    for (int e = 0; e < 8; ++e) { acc[c][e] = 0.f; wv8[c][e] = 0.f; }
    if (i < H) load8(w + i, wv8[c]);
  }
  for (int k = 0; k < rpw; ++k) {
    const long row = wv * rpw + k;
    if (row >= rows) break;
    const float* xr = x + row * H;
    const float* dyr = dy + row * H;
    float xv[NCH][8], dv[NCH][8];
    float ss = 0.f, sd = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int i = lane * 8 + 512 * c;
      if (i < H) {
        load8(xr + i, xv[c]); load8(dyr + i, dv[c]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { ss += xv[c][e] * xv[c][e]; sd += wv8[c][e] * dv[c][e] * xv[c][e]; }
      }
    }
    ss = wave_sum(ss);
    sd = wave_sum(sd);
    const float r = rsqrtf(ss / (float)H + eps);
    const float coef = r * r * r * sd / (float)H;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int i = lane * 8 + 512 * c;
      if (i < H) {
        float o[8];
        if (dres) load8(dres + row * H + i, o);
        else {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          o[e] += r * wv8[c][e] * dv[c][e] - coef * xv[c][e];
          acc[c][e] += dv[c][e] * xv[c][e] * r;
        }
        store8(dx + row * H + i, o);
      }
    }
  }
  if (dw_part) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int i = lane * 8 + 512 * c;
      if (i < H) store8(dw_part + wv * H + i, acc[c]);
    }
  }
}

// deterministic column sums of in [R][ld] (C columns): stage 1 = `chunks` row ranges -> part[chunk][C]; stage 2 (chunks == 1) -> out.
// (Two launches on purpose.  A one-launch form -- the block of a column range that arrives last folds the chunks -- was built and measured at
// 36-66 us against 5 + 6: on this chip the device-scope release / acquire it needs writes back and invalidates a whole XCD's L2, the eight L2s not
// being coherent with each other.)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, int ld, long R, int C, float* __restrict__ out, long rows_per_chunk) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const long r0 = (long)blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  long r = r0;
  for (; r + 3 < r1; r += 4) {
    s0 += in[r * ld + c]; s1 += in[(r + 1) * ld + c]; s2 += in[(r + 2) * ld + c]; s3 += in[(r + 3) * ld + c];
  }
  for (; r < r1; ++r) s0 += in[r * ld + c];
  out[(size_t)blockIdx.y * C + c] = (s0 + s1) + (s2 + s3);
}

// rows of the pooled position (last_token: Ni + max(len - 1, 0)) between the [B * Tt][H] stream and a compact [B][H] buffer
__global__ __launch_bounds__(256) void pool_rows_kernel(float* __restrict__ stream, float* __restrict__ compact, const int32_t* __restrict__ lens,
                                                         int Tt, int Ni, int H, int scatter) {
  const int b = blockIdx.x;
  int len = lens ? lens[b] : (Tt - Ni);
  len = min(max(len, 0), Tt - Ni);
  float* s = stream + ((size_t)b * Tt + Ni + max(len - 1, 0)) * H;
  float* c = compact + (size_t)b * H;
  for (int i = threadIdx.x * 4; i < H; i += 1024) {
    if (scatter) *reinterpret_cast<float4*>(s + i) = *reinterpret_cast<const float4*>(c + i);
    else *reinterpret_cast<float4*>(c + i) = *reinterpret_cast<const float4*>(s + i);
  }
}

// image-position rows of the stream <-> a compact [B * Ni][H] buffer (the projector's output / its gradient)
__global__ __launch_bounds__(256) void image_rows_kernel(const float* __restrict__ stream, float* __restrict__ compact, int Tt, int Ni, int H, long n4) {
  const int h4 = H >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long row = i / h4;
    const int c = (int)(i % h4) * 4;
    const long b = row / Ni, t = row % Ni;
    *reinterpret_cast<float4*>(compact + row * H + c) = *reinterpret_cast<const float4*>(stream + (b * Tt + t) * H + c);
  }
}

// d embed_tokens: dE[id][:] = sum over the valid text positions holding `id` of dx[row][:], in row order (no atomics).  One block
// per text position; the block of an id's FIRST valid occurrence sums all of them, the others return.  dE is zero-filled by the caller.
// The (clamped id | -1 for padding) of every position is staged in LDS first: read from global memory inside the two scans, each of the
// B T-long loops was a chain of dependent L2 round trips (0.57 ms per step at B T = 2048 for a few MB of work).
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int32_t* __restrict__ ids, const int32_t* __restrict__ lens, const float* __restrict__ dx,
                                                         float* __restrict__ dE, int B, int T, int Ni, int H, int vocab) {
  extern __shared__ int s_ids[];
  const int pos = blockIdx.x, n = B * T;
  for (int p = threadIdx.x; p < n; p += 256) {
    const int pb = p / T, pt = p - pb * T;
    const int pl = lens ? min(max(lens[pb], 0), T) : T;
    s_ids[p] = pt < pl ? min(max(ids[p], 0), vocab - 1) : -1;
  }
  __syncthreads();
  const int id = s_ids[pos];
  if (id < 0) return;
  int mine = 1;
  for (int p = threadIdx.x; p < pos; p += 256)
    if (s_ids[p] == id) mine = 0;
  if (!__syncthreads_and(mine)) return;
  const int Tt = Ni + T;
  for (int c = threadIdx.x * 4; c < H; c += 1024) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = pos; p < n; ++p) {
      if (s_ids[p] != id) continue;
      const int pb = p / T, pt = p - pb * T;
      const float4 v = *reinterpret_cast<const float4*>(dx + ((size_t)pb * Tt + Ni + pt) * H + c);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(dE + (size_t)id * H + c) = acc;
  }
}

// ------------------------------------------------------------------------------------------------------------------ attention backward
// Causal GQA attention with key masking (key j visible to query i iff j <= i and j < len_b), fp32 on v_mfma_f32_16x16x4_f32 like
// attention_f32_mfma_kernel (decoder_kernels.hip), which writes the row statistics lse = max + log(sum) this pass starts from:
//   P = exp(S * scale - lse),  dP = dO . V^T,  delta_i = sum_d dO_id O_id,  dS = P o (dP - delta) * scale,
//   dQ = dS . K,  dK = dS^T . Q,  dV = P^T . dO        (Q, K = the ROTATED projections; the gradient is rotated back on the way out)
// Two kernels, both recomputing S and dP, so that every gradient element is summed by ONE wave in a fixed order:
//   attn_bwd_dq_kernel   block = 64 queries of one (batch, q head); K / V chunks through LDS; also writes delta
//   attn_bwd_dkv_kernel  block = 64 keys of one (batch, kv head); loops over the group's q heads and their query chunks (Q, dO in LDS)
// Fragment conventions as in the forward kernel: lane = (fr = lane & 15, fg = lane >> 4); an MFMA takes A[i = fr][k = fg] and
// B[k = fg][j = fr] and returns D[i = 4 fg + r][j = fr] in register r.
template <int D>
__device__ __forceinline__ void rope_rows_to_lds(float* __restrict__ dst, const float* __restrict__ src_base, int ld, int col0, const float2* __restrict__ rope,
                                                 int row0, int nrows, int row_max, int LDR, int tid, bool rotate) {
  // rows row0 .. row0 + nrows - 1 (clamped to row_max) of width D at column col0 of src -> dst[row][D] (row stride LDR), rotated by the
  // row's position when `rotate` (rotate-half RoPE: d pairs with d + D/2)
  for (int i = tid; i < nrows * D / 8; i += 256) {
    const int rr = i / (D / 8), c4 = i % (D / 8);
    const int row = min(row0 + rr, row_max);
    const float* base = src_base + (size_t)row * ld + col0 + c4 * 4;
    const float4 a = *reinterpret_cast<const float4*>(base), bb = *reinterpret_cast<const float4*>(base + D / 2);
    if (rotate) {
      const float2* t = rope + (size_t)row * (D / 2) + c4 * 4;
      const float4 cs0 = *reinterpret_cast<const float4*>(t), cs1 = *reinterpret_cast<const float4*>(t + 2);
      *reinterpret_cast<float4*>(dst + rr * LDR + c4 * 4) =
          make_float4(a.x * cs0.x - bb.x * cs0.y, a.y * cs0.z - bb.y * cs0.w, a.z * cs1.x - bb.z * cs1.y, a.w * cs1.z - bb.w * cs1.w);
      *reinterpret_cast<float4*>(dst + rr * LDR + D / 2 + c4 * 4) =
          make_float4(bb.x * cs0.x + a.x * cs0.y, bb.y * cs0.z + a.y * cs0.w, bb.z * cs1.x + a.z * cs1.y, bb.w * cs1.z + a.w * cs1.w);
    } else {
      *reinterpret_cast<float4*>(dst + rr * LDR + c4 * 4) = a;
      *reinterpret_cast<float4*>(dst + rr * LDR + D / 2 + c4 * 4) = bb;
    }
  }
}

// fragments of one row for a wave's lane: f[c] = row[16 c + 4 fg .. + 3], rotated by `pos` when rope != null
template <int D>
__device__ __forceinline__ void load_row_frag(float4 (&f)[D / 16], const float* __restrict__ rowp, const float2* __restrict__ rope, int pos, int fg) {
  constexpr int DT = D / 16;
#pragma unroll
  for (int c = 0; c < DT; ++c) f[c] = *reinterpret_cast<const float4*>(rowp + 16 * c + 4 * fg);
  if (rope) {
    const float2* t = rope + (size_t)pos * (D / 2) + 4 * fg;
#pragma unroll
    for (int c = 0; c < DT / 2; ++c) {
      const float4 cs0 = *reinterpret_cast<const float4*>(t + 16 * c), cs1 = *reinterpret_cast<const float4*>(t + 16 * c + 2);
      const float4 a = f[c], b = f[c + DT / 2];
      f[c] = make_float4(a.x * cs0.x - b.x * cs0.y, a.y * cs0.z - b.y * cs0.w, a.z * cs1.x - b.z * cs1.y, a.w * cs1.z - b.w * cs1.w);
      f[c + DT / 2] = make_float4(b.x * cs0.x + a.x * cs0.y, b.y * cs0.z + a.y * cs0.w, b.z * cs1.x + a.z * cs1.y, b.w * cs1.z + a.w * cs1.w);
    }
  }
}

// gradient w.r.t. the rotated row (accumulator layout g[dt][r] = element 16 dt + 4 fg + r) -> gradient w.r.t. the un-rotated
// projection, stored at dst: rotated (q1, q2) = (a c - b s, b c + a s)  =>  da = g1 c + g2 s, db = -g1 s + g2 c
template <int D>
__device__ __forceinline__ void store_unrotated(float* __restrict__ dst, const f32x4 (&g)[D / 16], const float2* __restrict__ rope, int pos, int fg) {
  constexpr int DT = D / 16;
  if (!rope) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) *reinterpret_cast<float4*>(dst + 16 * dt + 4 * fg) = make_float4(g[dt][0], g[dt][1], g[dt][2], g[dt][3]);
    return;
  }
  const float2* t = rope + (size_t)pos * (D / 2) + 4 * fg;
#pragma unroll
  for (int dt = 0; dt < DT / 2; ++dt) {
    const float4 cs0 = *reinterpret_cast<const float4*>(t + 16 * dt), cs1 = *reinterpret_cast<const float4*>(t + 16 * dt + 2);
    const f32x4 g1 = g[dt], g2 = g[dt + DT / 2];
    *reinterpret_cast<float4*>(dst + 16 * dt + 4 * fg) =
        make_float4(g1[0] * cs0.x + g2[0] * cs0.y, g1[1] * cs0.z + g2[1] * cs0.w, g1[2] * cs1.x + g2[2] * cs1.y, g1[3] * cs1.z + g2[3] * cs1.w);
    *reinterpret_cast<float4*>(dst + D / 2 + 16 * dt + 4 * fg) =
        make_float4(g2[0] * cs0.x - g1[0] * cs0.y, g2[1] * cs0.z - g1[1] * cs0.w, g2[2] * cs1.x - g1[2] * cs1.y, g2[3] * cs1.z - g1[3] * cs1.w);
  }
}

#define MFMA4(acc, af, bf)                                                   \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((af).x, (bf).x, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((af).y, (bf).y, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((af).z, (bf).z, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32((af).w, (bf).w, acc, 0, 0, 0)

template <int D>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const float* __restrict__ qkv, int ld, const bf16_t* __restrict__ o_hi, const bf16_t* __restrict__ o_lo,
                                                              int ldo, const float* __restrict__ dO, int lddo, const float* __restrict__ lse,
                                                              float* __restrict__ delta, float* __restrict__ dqkv, const int32_t* __restrict__ lens,
                                                              int len_add, int T, int heads, int kv_heads, float scale, const float2* __restrict__ rope) {
  constexpr int DT = D / 16;
  constexpr int KCH = D == 64 ? 64 : 32;
  constexpr int LDR = D + 4;
  __shared__ __attribute__((aligned(16))) float sK[KCH * LDR];
  __shared__ __attribute__((aligned(16))) float sV[KCH * LDR];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int qblocks = (T + 63) >> 6;
  int bid = blockIdx.x;
  const int qb = bid % qblocks; bid /= qblocks;
  const int h = bid % heads;
  const int b = bid / heads;
  const int hk = h / (heads / kv_heads);
  int len = lens ? lens[b] + len_add : T;
  len = max(1, min(len, T));
  const int q0 = qb * 64 + wid * 16, qg = q0 + fr, qc = min(qg, T - 1);
  const int qd = heads * D, kd = kv_heads * D;
  const size_t rowq = (size_t)b * T + qc;

  float4 fq[DT], fdo[DT];
  load_row_frag<D>(fq, qkv + rowq * ld + h * D, rope, qc, fg);
  load_row_frag<D>(fdo, dO + rowq * lddo + h * D, nullptr, 0, fg);
  float dl = 0.f;
  {
    const bf16_t* ph = o_hi + rowq * ldo + h * D + 4 * fg;
    const bf16_t* pl = o_lo + rowq * ldo + h * D + 4 * fg;
#pragma unroll
    for (int c = 0; c < DT; ++c) {
      const uint2 hv = *reinterpret_cast<const uint2*>(ph + 16 * c), lv = *reinterpret_cast<const uint2*>(pl + 16 * c);
      dl += fdo[c].x * (bf_lo(hv.x) + bf_lo(lv.x)) + fdo[c].y * (bf_hi(hv.x) + bf_hi(lv.x)) + fdo[c].z * (bf_lo(hv.y) + bf_lo(lv.y)) +
            fdo[c].w * (bf_hi(hv.y) + bf_hi(lv.y));
    }
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
  }
  const float my_lse = lse[((size_t)b * heads + h) * T + qc];
  if (fg == 0 && qg < T) delta[((size_t)b * heads + h) * T + qg] = dl;

  f32x4 dq[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* kbase = qkv + (size_t)b * T * ld;
  const int kend = min(len, qb * 64 + 64);
  for (int k0 = 0; k0 < kend; k0 += KCH) {
    __syncthreads();
    rope_rows_to_lds<D>(sK, kbase, ld, qd + hk * D, rope, k0, KCH, T - 1, LDR, tid, rope != nullptr);
    rope_rows_to_lds<D>(sV, kbase, ld, qd + kd + hk * D, nullptr, k0, KCH, T - 1, LDR, tid, false);
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < KCH / 16; ++kt) {
      const int kb = k0 + kt * 16;
      if (kb > q0 + 15 || kb >= len) break;
      f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f}, dpacc = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* kr = sK + (kt * 16 + fr) * LDR + 4 * fg;
      const float* vrw = sV + (kt * 16 + fr) * LDR + 4 * fg;
#pragma unroll
      for (int c = 0; c < DT; ++c) {
        const float4 kf = *reinterpret_cast<const float4*>(kr + 16 * c);
        const float4 vf = *reinterpret_cast<const float4*>(vrw + 16 * c);
        MFMA4(sacc, kf, fq[c]);
        MFMA4(dpacc, vf, fdo[c]);
      }
      float ds[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kg = kb + 4 * fg + r;
        const float p = (kg <= qg && kg < len) ? __expf(sacc[r] * scale - my_lse) : 0.f;
        ds[r] = p * (dpacc[r] - dl) * scale;
      }
      const float* kc = sK + (kt * 16 + 4 * fg) * LDR + fr;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kc[r * LDR + 16 * dt], ds[r], dq[dt], 0, 0, 0);
      }
    }
  }
  if (qg >= T) return;
  store_unrotated<D>(dqkv + ((size_t)b * T + qg) * ld + h * D, dq, rope, qg, fg);
}

// PART: one block per (batch, kv head, key block, q head OF THE GROUP): the block writes that head's share of dK / dV to
// part[hh][row][2 kd] (k | v of every kv head side by side), summed over hh in a fixed order by attn_dkv_reduce_kernel -- 7x the blocks of
// the looped form at 14 q / 2 kv heads (320 -> 2240 at B = 32, T = 320: the looped form left the chip 1.25 rounds of unbalanced blocks)
template <int D, bool PART>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const float* __restrict__ qkv, int ld, const float* __restrict__ dO, int lddo,
                                                               const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dqkv,
                                                               const int32_t* __restrict__ lens, int len_add, int T, int heads, int kv_heads,
                                                               float scale, const float2* __restrict__ rope, float* __restrict__ part, long part_stride) {
  constexpr int DT = D / 16;
  constexpr int QCH = D == 64 ? 64 : 32;
  constexpr int LDR = D + 4;
  __shared__ __attribute__((aligned(16))) float sQ[QCH * LDR];
  __shared__ __attribute__((aligned(16))) float sD[QCH * LDR];
  __shared__ float sLse[QCH], sDel[QCH];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int kblocks = (T + 63) >> 6;
  const int grp = heads / kv_heads;
  int bid = blockIdx.x;
  int hh0 = 0, hh1 = grp;
  if constexpr (PART) { hh0 = bid % grp; hh1 = hh0 + 1; bid /= grp; }
  const int kbk = bid % kblocks; bid /= kblocks;
  const int hk = bid % kv_heads;
  const int b = bid / kv_heads;
  int len = lens ? lens[b] + len_add : T;
  len = max(1, min(len, T));
  const int k0w = kbk * 64 + wid * 16, kg = k0w + fr, kc = min(kg, T - 1);
  const int qd = heads * D, kd = kv_heads * D;
  const float* base = qkv + (size_t)b * T * ld;

  float4 fk[DT], fv[DT];
  load_row_frag<D>(fk, base + (size_t)kc * ld + qd + hk * D, rope, kc, fg);
  load_row_frag<D>(fv, base + (size_t)kc * ld + qd + kd + hk * D, nullptr, 0, fg);
  f32x4 dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int qstart = (kbk * 64) / QCH * QCH;       // causal: queries before the block's first key never see it
  if (kbk * 64 < len) {
    for (int hh = hh0; hh < hh1; ++hh) {
      const int h = hk * grp + hh;
      for (int qc0 = qstart; qc0 < len; qc0 += QCH) {
        __syncthreads();
        rope_rows_to_lds<D>(sQ, base, ld, h * D, rope, qc0, QCH, T - 1, LDR, tid, rope != nullptr);
        rope_rows_to_lds<D>(sD, dO + (size_t)b * T * lddo, lddo, h * D, nullptr, qc0, QCH, T - 1, LDR, tid, false);
        if (tid < QCH) {
          const int qi = min(qc0 + tid, T - 1);
          sLse[tid] = lse[((size_t)b * heads + h) * T + qi];
          sDel[tid] = delta[((size_t)b * heads + h) * T + qi];
        }
        __syncthreads();
#pragma unroll 1
        for (int qt = 0; qt < QCH / 16; ++qt) {
          const int qb = qc0 + qt * 16;
          if (qb >= len) break;
          if (qb + 15 < k0w || k0w >= len) continue;   // wave-uniform: every query of the tile precedes every key of the wave / masked keys
          f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f}, dpacc = f32x4{0.f, 0.f, 0.f, 0.f};
          const float* qr = sQ + (qt * 16 + fr) * LDR + 4 * fg;
          const float* dr = sD + (qt * 16 + fr) * LDR + 4 * fg;
#pragma unroll
          for (int c = 0; c < DT; ++c) {
            const float4 qf = *reinterpret_cast<const float4*>(qr + 16 * c);
            const float4 df = *reinterpret_cast<const float4*>(dr + 16 * c);
            MFMA4(sacc, qf, fk[c]);
            MFMA4(dpacc, df, fv[c]);
          }
          float p[4], ds[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = qb + 4 * fg + r;
            const bool vis = kg <= qi && kg < len && qi < len;
            p[r] = vis ? __expf(sacc[r] * scale - sLse[qt * 16 + 4 * fg + r]) : 0.f;
            ds[r] = p[r] * (dpacc[r] - sDel[qt * 16 + 4 * fg + r]) * scale;
          }
          const float* qcol = sQ + (qt * 16 + 4 * fg) * LDR + fr;
          const float* dcol = sD + (qt * 16 + 4 * fg) * LDR + fr;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              dv[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(dcol[r * LDR + 16 * dt], p[r], dv[dt], 0, 0, 0);
              dk[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qcol[r * LDR + 16 * dt], ds[r], dk[dt], 0, 0, 0);
            }
          }
        }
      }
    }
  }
  if (kg >= T) return;
  if constexpr (PART) {
    float* prow = part + (size_t)hh0 * part_stride + ((size_t)b * T + kg) * (2 * kd);
    store_unrotated<D>(prow + hk * D, dk, rope, kg, fg);
    store_unrotated<D>(prow + kd + hk * D, dv, nullptr, 0, fg);
  } else {
    float* drow = dqkv + ((size_t)b * T + kg) * ld;
    store_unrotated<D>(drow + qd + hk * D, dk, rope, kg, fg);
    store_unrotated<D>(drow + qd + kd + hk * D, dv, nullptr, 0, fg);
  }
}

// dqkv[row][qd ..] = sum over the group's q heads (fixed order) of part[hh][row][2 kd]
__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(const float* __restrict__ part, long part_stride, int grp, float* __restrict__ dqkv, int ld, int qd,
                                                               int w2, long n4) {
  const int c4 = w2 >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long row = i / c4;
    const int c = (int)(i % c4) * 4;
    float4 a = *reinterpret_cast<const float4*>(part + row * w2 + c);
    for (int g = 1; g < grp; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(part + (size_t)g * part_stride + row * w2 + c);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(dqkv + row * ld + qd + c) = a;
  }
}

inline unsigned grid_for(long work, int per_block = 256, unsigned cap = 4096) {
  long g = (work + per_block - 1) / per_block;
  return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// ================================================================================================================ launchers
int launch_split_rows(const float* in, int ldi, bf16_t* out, int ldo, int lo_off, long R, int C, hipStream_t s) {
  if (!in || !out) return fv_fail(FV_ERR_ARG, "split_rows: null pointer");
  if (R <= 0 || C <= 0 || C % 8 || ldi % 4 || ldi < C || ldo % 8 || lo_off % 8 || ldo < (lo_off ? lo_off + C : C) || (lo_off && lo_off < C))
    return fv_fail(FV_ERR_ARG, "split_rows: bad shape R=%ld C=%d ldi=%d ldo=%d lo_off=%d", R, C, ldi, ldo, lo_off);
  hipLaunchKernelGGL(split_rows_kernel, dim3(grid_for(R * (C / 8))), dim3(256), 0, s, in, ldi, out, ldo, lo_off, R, C);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_lo8_rows(const float* in, int ldi, bf16_t* out, int ldo, long R, int C, hipStream_t s) {
  if (!in || !out || R <= 0 || C <= 0 || C % 8 || ldi % 4 || ldi < C || ldo % 8 || ldo * 2 < 3 * C) return fv_fail(FV_ERR_ARG, "lo8_rows: bad shape");
  hipLaunchKernelGGL(lo8_rows_kernel, dim3(grid_for(R * (C / 8))), dim3(256), 0, s, in, ldi, out, ldo, R, C);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_bf16_to_f32(const bf16_t* in, float* out, size_t n, hipStream_t s) {
  if (!in || !out || n == 0 || n % 8 || (((uintptr_t)in | (uintptr_t)out) & 15)) return fv_fail(FV_ERR_ARG, "bf16_to_f32: n %% 8 == 0 and 16-byte aligned pointers");
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(grid_for((long)(n / 8))), dim3(256), 0, s, in, out, (long)(n / 8));
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

// in [R][C] fp32 (in_bf16 == 0) or bf16 -> out bf16 [C][ldo]: hi at [c][r], lo (fp32 input only) at [c][lo_off + r]; rows [R, Rp) zero
int launch_transpose_to_bf16(const void* in, int in_bf16, int ldi, bf16_t* out, int ldo, int lo_off, int R, int Rp, int C, hipStream_t s) {
  if (!in || !out) return fv_fail(FV_ERR_ARG, "transpose: null pointer");
  if (R <= 0 || C <= 0 || C % 8 || Rp < R || Rp % 8 || ldi < C || ldi % (in_bf16 ? 8 : 4) || ldo % 8 || lo_off % 8 || ldo < (lo_off ? lo_off + Rp : Rp) ||
      (lo_off && (lo_off < Rp || in_bf16)))
    return fv_fail(FV_ERR_ARG, "transpose: bad shape R=%d Rp=%d C=%d ldi=%d ldo=%d lo_off=%d", R, Rp, C, ldi, ldo, lo_off);
  const dim3 g((Rp + TP - 1) / TP, (C + TP - 1) / TP);
  if (in_bf16) hipLaunchKernelGGL(transpose_kernel<bf16_t>, g, dim3(256), 0, s, static_cast<const bf16_t*>(in), ldi, out, ldo, 0, R, Rp, C);
  else hipLaunchKernelGGL(transpose_kernel<float>, g, dim3(256), 0, s, static_cast<const float*>(in), ldi, out, ldo, lo_off, R, Rp, C);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_transpose_to_f16(const void* in, int in_kind, int ldi, int lo_in, bf16_t* out, int ldo, int R, int Rp, int C, unsigned* sat, hipStream_t s,
                            bf16_t* rows_out, int ldro) {
  if (!in || !out) return fv_fail(FV_ERR_ARG, "transpose_f16: null pointer");
  if (rows_out && (ldro < C || ldro % 8)) return fv_fail(FV_ERR_ARG, "transpose_f16: bad row output stride %d", ldro);
  if (R <= 0 || C <= 0 || C % 8 || Rp < R || Rp % 8 || ldi < C || ldi % (in_kind ? 8 : 4) || ldo % 8 || ldo < Rp || in_kind < 0 || in_kind > 3 || (in_kind == 2 && (lo_in % 8 || lo_in < C)))
    return fv_fail(FV_ERR_ARG, "transpose_f16: bad shape R=%d Rp=%d C=%d ldi=%d ldo=%d kind=%d", R, Rp, C, ldi, ldo, in_kind);
  const dim3 g((Rp + TP - 1) / TP, (C + TP - 1) / TP);
  if (in_kind == 0) hipLaunchKernelGGL(transpose_f16_kernel<0>, g, dim3(256), 0, s, in, ldi, lo_in, out, ldo, R, Rp, C, sat, rows_out, ldro);
  else if (in_kind == 1) hipLaunchKernelGGL(transpose_f16_kernel<1>, g, dim3(256), 0, s, in, ldi, lo_in, out, ldo, R, Rp, C, sat, rows_out, ldro);
  else if (in_kind == 3) hipLaunchKernelGGL(transpose_f16_kernel<3>, g, dim3(256), 0, s, in, ldi, lo_in, out, ldo, R, Rp, C, sat, rows_out, ldro);
  else hipLaunchKernelGGL(transpose_f16_kernel<2>, g, dim3(256), 0, s, in, ldi, lo_in, out, ldo, R, Rp, C, sat, rows_out, ldro);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_rows_to_f16(const void* in, int in_kind, int ldi, int lo_in, bf16_t* out, int ldo, long R, int C, unsigned* sat, hipStream_t s, long Rp) {
  if (!in || !out || R <= 0 || C <= 0 || C % 8 || ldi < C || ldi % (in_kind ? 8 : 4) || ldo % 8 || ldo < C || in_kind < 0 || in_kind > 2 || (in_kind == 2 && (lo_in % 8 || lo_in < C)) || (Rp && Rp < R))
    return fv_fail(FV_ERR_ARG, "rows_to_f16: bad arguments");
  if (!Rp) Rp = R;
  const dim3 g(grid_for(Rp * (C / 8)));
  if (in_kind == 0) hipLaunchKernelGGL(rows_f16_kernel<0>, g, dim3(256), 0, s, in, ldi, lo_in, out, ldo, R, Rp, C, sat);
  else if (in_kind == 1) hipLaunchKernelGGL(rows_f16_kernel<1>, g, dim3(256), 0, s, in, ldi, lo_in, out, ldo, R, Rp, C, sat);
  else hipLaunchKernelGGL(rows_f16_kernel<2>, g, dim3(256), 0, s, in, ldi, lo_in, out, ldo, R, Rp, C, sat);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_swiglu_bwd(float* gu, const float* dact, long rows, int I, hipStream_t s, bf16_t* out_split) {
  if (!gu || !dact || rows <= 0 || I <= 0 || I % 8) return fv_fail(FV_ERR_ARG, "swiglu_bwd: bad arguments");
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(rows * (I / 8))), dim3(256), 0, s, gu, dact, rows, I, out_split);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}
int launch_commit(const CommitDesc* desc_dev, int ndesc, int ntiles, const float* flat, int f16_transposes, unsigned* sat, hipStream_t s) {
  if (!desc_dev || !flat || ndesc <= 0 || ntiles <= 0 || (f16_transposes && !sat)) return fv_fail(FV_ERR_ARG, "commit: bad arguments");
  hipLaunchKernelGGL(commit_kernel, dim3((unsigned)ntiles), dim3(256), 0, s, desc_dev, ndesc, flat, f16_transposes, sat);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}
int launch_swiglu_bwd_f16(const void* gu, int gu_f16, const float* dact, int rows, int Rp, int I, bf16_t* out_rows, bf16_t* outT, unsigned* sat, hipStream_t s, bf16_t* actT) {
  if (!gu || !dact || !out_rows || !outT || !sat || rows <= 0 || Rp < rows || Rp % 8 || I <= 0 || I % 8) return fv_fail(FV_ERR_ARG, "swiglu_bwd_f16: bad arguments");
  const dim3 g((Rp + TP - 1) / TP, (2 * I + TP - 1) / TP);
  if (gu_f16) hipLaunchKernelGGL(swiglu_bwd_f16_kernel<true>, g, dim3(256), 0, s, gu, dact, rows, Rp, I, out_rows, outT, sat, actT);
  else hipLaunchKernelGGL(swiglu_bwd_f16_kernel<false>, g, dim3(256), 0, s, gu, dact, rows, Rp, I, out_rows, outT, sat, actT);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}
int launch_swiglu_bwd_rows(const void* gu, int gu_f16, const float* dact, long rows, long Rp, int I, bf16_t* dgu_rows, bf16_t* act_rows, unsigned* sat, hipStream_t s) {
  if (!gu || !dact || !dgu_rows || !act_rows || !sat || rows <= 0 || Rp < rows || I <= 0 || I % 8) return fv_fail(FV_ERR_ARG, "swiglu_bwd_rows: bad arguments");
  const dim3 g(grid_for(Rp * (I / 8)));
  if (gu_f16) hipLaunchKernelGGL(swiglu_bwd_rows_kernel<true>, g, dim3(256), 0, s, gu, dact, rows, Rp, I, dgu_rows, act_rows, sat);
  else hipLaunchKernelGGL(swiglu_bwd_rows_kernel<false>, g, dim3(256), 0, s, gu, dact, rows, Rp, I, dgu_rows, act_rows, sat);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}
int launch_gelu_fwd(const float* pre, bf16_t* out, int ldo, int lo_off, long R, int C, hipStream_t s) {
  if (!pre || !out || R <= 0 || C <= 0 || C % 8 || ldo % 8 || lo_off < C || lo_off % 8 || ldo < lo_off + C) return fv_fail(FV_ERR_ARG, "gelu_fwd: bad arguments");
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(R * (C / 8))), dim3(256), 0, s, pre, out, ldo, lo_off, R, C);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}
int launch_gelu_bwd(float* dh, const float* pre, size_t n, hipStream_t s) {
  if (!dh || !pre || n == 0 || n % 8) return fv_fail(FV_ERR_ARG, "gelu_bwd: bad arguments");
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for((long)(n / 8))), dim3(256), 0, s, dh, pre, (long)(n / 8));
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

size_t rmsnorm_bwd_scratch_floats(long rows, int H) {   // dw partial rows + one stage of column sums
  const long waves = (rows + RMS_BWD_RPW - 1) / RMS_BWD_RPW;
  return (size_t)((waves + 3) / 4 * 4 + COLSUM_CHUNKS) * H;
}
int launch_colsum(const float* in, int ld, long R, int C, float* out, float* scratch, hipStream_t s) {
  if (!in || !out || !scratch || R <= 0 || C <= 0 || ld < C) return fv_fail(FV_ERR_ARG, "colsum: bad arguments");
  int chunks = R >= 4 * COLSUM_CHUNKS ? COLSUM_CHUNKS : 1;
  const long rpc = (R + chunks - 1) / chunks;
  chunks = (int)((R + rpc - 1) / rpc);
  const dim3 g((C + 255) / 256, chunks);
  if (chunks == 1) {
    hipLaunchKernelGGL(colsum_kernel, g, dim3(256), 0, s, in, ld, R, C, out, rpc);
  } else {
    hipLaunchKernelGGL(colsum_kernel, g, dim3(256), 0, s, in, ld, R, C, scratch, rpc);
    hipLaunchKernelGGL(colsum_kernel, dim3((C + 255) / 256, 1), dim3(256), 0, s, scratch, C, (long)chunks, C, out, (long)chunks);
  }
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}
// dx = dres + rmsnorm'(dy) and dw (H floats); scratch >= rmsnorm_bwd_scratch_floats(rows, H).  dx may alias dres or dy.
int launch_rmsnorm_bwd(const float* x, const float* w, const float* dy, const float* dres, float* dx, float* dw, float* scratch, long rows, int H,
                       float eps, hipStream_t s) {
  if (!x || !w || !dy || !dx || !scratch) return fv_fail(FV_ERR_ARG, "rmsnorm_bwd: null pointer");
  if (rows <= 0 || H <= 0 || H % 8 || H > 4096) return fv_fail(FV_ERR_ARG, "rmsnorm_bwd: bad shape rows=%ld H=%d", rows, H);
  const long waves = (rows + RMS_BWD_RPW - 1) / RMS_BWD_RPW;
  const long wpad = (waves + 3) / 4 * 4;
  // (waves of the last block that own no row write zero partial rows themselves)
  const dim3 g((unsigned)(wpad / 4));
  float* part = dw ? scratch : nullptr;
  if (H <= 1024) hipLaunchKernelGGL(rmsnorm_bwd_kernel<2>, g, dim3(256), 0, s, x, w, dy, dres, dx, part, rows, H, eps, RMS_BWD_RPW);
  else if (H <= 2048) hipLaunchKernelGGL(rmsnorm_bwd_kernel<4>, g, dim3(256), 0, s, x, w, dy, dres, dx, part, rows, H, eps, RMS_BWD_RPW);
  else hipLaunchKernelGGL(rmsnorm_bwd_kernel<8>, g, dim3(256), 0, s, x, w, dy, dres, dx, part, rows, H, eps, RMS_BWD_RPW);
  FV_HIP_CHECK(hipGetLastError());
  if (dw) return launch_colsum(scratch, H, wpad, H, dw, scratch + wpad * H, s);
  return FV_OK;
}

int launch_pool_rows(float* stream, float* compact, const int32_t* lens, int B, int Tt, int Ni, int H, int scatter, hipStream_t s) {
  if (!stream || !compact || B <= 0 || Tt <= Ni || H % 4) return fv_fail(FV_ERR_ARG, "pool_rows: bad arguments");
  hipLaunchKernelGGL(pool_rows_kernel, dim3(B), dim3(256), 0, s, stream, compact, lens, Tt, Ni, H, scatter);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}
int launch_image_rows(const float* stream, float* compact, int B, int Tt, int Ni, int H, hipStream_t s) {
  if (!stream || !compact || B <= 0 || Ni <= 0 || Tt < Ni || H % 4) return fv_fail(FV_ERR_ARG, "image_rows: bad arguments");
  const long n4 = (long)B * Ni * (H / 4);
  hipLaunchKernelGGL(image_rows_kernel, dim3(grid_for(n4)), dim3(256), 0, s, stream, compact, Tt, Ni, H, n4);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}
int launch_embed_bwd(const int32_t* ids, const int32_t* lens, const float* dx, float* dE, int B, int T, int Ni, int H, int vocab, hipStream_t s) {
  if (!ids || !dx || !dE || B <= 0 || T <= 0 || H % 4 || vocab <= 0) return fv_fail(FV_ERR_ARG, "embed_bwd: bad arguments");
  if ((long)B * T > 16384) return fv_fail(FV_ERR_UNSUPPORTED, "embed_bwd: %ld text positions (the id table lives in 64 KB of LDS)", (long)B * T);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(B * T), dim3(256), (size_t)B * T * sizeof(int), s, ids, lens, dx, dE, B, T, Ni, H, vocab);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

// dqkv (fp32, the layout of qkv: [rows][ld] = q | k | v, gradients w.r.t. the UN-rotated projections) from dO (fp32 [rows][lddo]),
// the forward's inputs qkv, its output O = o_hi + o_lo (bf16, [rows][ldo]) and row statistics lse [B][heads][T]; delta: scratch
// [B][heads][T].  head_dim 64 / 128.
int launch_attention_bwd(const float* qkv, int ld, const bf16_t* o_hi, const bf16_t* o_lo, int ldo, const float* dO, int lddo, const float* lse,
                         float* delta, float* dqkv, int B, int T, int heads, int kv_heads, int D, const int32_t* lens, int len_add, float scale,
                         const float2* rope, hipStream_t s, float* part, void* split_scratch) {
  if (!qkv || !o_hi || !o_lo || !dO || !lse || !delta || !dqkv) return fv_fail(FV_ERR_ARG, "attention_bwd: null pointer");
  if (D != 64 && D != 128) return fv_fail(FV_ERR_UNSUPPORTED, "attention_bwd: head_dim must be 64 or 128 (got %d)", D);
  if (B <= 0 || T <= 0 || heads % kv_heads || ld % 4 || ld < (heads + 2 * kv_heads) * D || ldo % 8 || ldo < heads * D || lddo % 4 || lddo < heads * D)
    return fv_fail(FV_ERR_ARG, "attention_bwd: bad shape");
  const int blocks = (T + 63) / 64, grp = heads / kv_heads, kd = kv_heads * D;
  const long pstride = (long)B * T * 2 * kd;      // floats per q-head share: part >= grp * pstride floats
  const bool parts = part != nullptr && grp > 1;
  const dim3 gq(B * heads * blocks), gk(B * kv_heads * blocks * (parts ? grp : 1));
  static const bool no_split = fv_ab_env("FASTVLA_NO_ATTN_SPLIT") != nullptr;   // A/B: the fp32-MFMA kernels below
  if (!no_split && split_scratch) {
    FV_TRY_RC(launch_attention_split_bwd(qkv, ld, o_hi, o_lo, ldo, dO, lddo, lse, delta, dqkv, B, T, heads, kv_heads, D, lens, len_add, scale, rope, s, part, pstride, split_scratch));
  } else if (D == 64) {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, gq, dim3(256), 0, s, qkv, ld, o_hi, o_lo, ldo, dO, lddo, lse, delta, dqkv, lens, len_add, T, heads, kv_heads, scale, rope);
    if (parts) hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, true>), gk, dim3(256), 0, s, qkv, ld, dO, lddo, lse, delta, dqkv, lens, len_add, T, heads, kv_heads, scale, rope, part, pstride);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, false>), gk, dim3(256), 0, s, qkv, ld, dO, lddo, lse, delta, dqkv, lens, len_add, T, heads, kv_heads, scale, rope, part, pstride);
  } else {
    hipLaunchKernelGGL(attn_bwd_dq_kernel<128>, gq, dim3(256), 0, s, qkv, ld, o_hi, o_lo, ldo, dO, lddo, lse, delta, dqkv, lens, len_add, T, heads, kv_heads, scale, rope);
    if (parts) hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, true>), gk, dim3(256), 0, s, qkv, ld, dO, lddo, lse, delta, dqkv, lens, len_add, T, heads, kv_heads, scale, rope, part, pstride);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, false>), gk, dim3(256), 0, s, qkv, ld, dO, lddo, lse, delta, dqkv, lens, len_add, T, heads, kv_heads, scale, rope, part, pstride);
  }
  if (parts) {
    const long n4 = (long)B * T * (2 * kd / 4);
    hipLaunchKernelGGL(attn_dkv_reduce_kernel, dim3(grid_for(n4)), dim3(256), 0, s, part, pstride, grp, dqkv, ld, heads * D, 2 * kd, n4);
  }
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

}  // namespace fv
