// attention_split.hip -- the decoder's causal GQA attention, forward (with the row statistics the backward starts from) and backward, on the
// bf16 matrix core with SPLIT operands (the unfrozen training path; reference call site of the whole VLM:
// src/vla_fastvlm/model/fastvlm_adapter.py:533; the attention itself is [site] transformers/models/qwen2/modeling_qwen2.py:105-135, 167-190).
//
// Why.  attention_f32_mfma_kernel / attn_bwd_*_kernel run on v_mfma_f32_16x16x4_f32: exact fp32 products at 32 MACs per clock and SIMD.  At the
// training shape (B = 32, T = 320: 256 image + 64 text positions, 14 q heads) the three kernels of one layer are 20.5 GFLOP = 131 us AT that
// pipe's peak and took 391 us -- 9.4 ms of a 93 ms step.  Here every fp32 operand x is carried as hi + lo (two bf16, 16 significant bits) and
// every product as hi.hi + lo.hi + hi.lo on v_mfma_f32_16x16x32_bf16 (512 MACs per clock: 5.3x the fp32 pipe after the three passes; the
// dropped lo.lo term is 2^-18 relative), accumulated in fp32 -- the arithmetic of llm_precision = 1's projections, applied to both operands.
//
// Layout conventions (lane = (fr = lane & 15, fg = lane >> 4); an MFMA takes A[i = fr][k = 8 fg .. 8 fg + 7], B[k = 8 fg ..][j = fr] and
// returns D[i = 4 fg + r][j = fr] in register r):
//   * a product contracting over head_dim reads both operands in their natural [row][d] form: one 16-byte LDS / register fragment per k-step;
//   * a product contracting over keys (or queries) takes the score tile it follows AS ITS B OPERAND: two 16-row score tiles of a 32-row step
//     leave lane (fr, fg) holding rows {4 fg + r} and {16 + 4 fg + r} of column fr -- eight values, i.e. the k-slots 8 fg .. 8 fg + 7 of a
//     contraction whose slot order is that permutation.  The OTHER operand (V^T, K^T, Q^T, dO^T) is written to LDS transposed with its 64 columns
//     in the same permuted order (position 32 g + 8 q + 4 t + e  <->  row 32 g + 16 t + 4 q + e), so its fragment is one 16-byte read too.
// Blocks: 4 waves; a wave owns NT 16-row tiles (2 at head_dim 64: each LDS fragment then feeds two tiles -- with one the LDS port, not the
// matrix pipe, bounds the kernel; 1 at head_dim 128, where two do not fit the registers).  Every gradient element is summed by ONE wave in a
// fixed order (no atomics): launches are bit-repeatable.
#include "kernels.h"

#ifndef FV_TRY_RC
#define FV_TRY_RC(expr) do { const int rc_ = (expr); if (rc_ != FV_OK) return rc_; } while (0)
#endif

namespace fv {
namespace {

constexpr int ACH = 64;            // rows (keys / queries) per LDS chunk
constexpr int ALDT = ACH + 8;      // transposed rows: 64 permuted positions + 16 bytes (an odd multiple of 16 bytes: conflict-free b128 reads)

__device__ __forceinline__ f32x4 mm(const bf16x8& a, const bf16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
#define AS_MM3(ACC, AH, AL, BH, BL) \
  ACC = mm(AH, BH, ACC);            \
  ACC = mm(AL, BH, ACC);            \
  ACC = mm(AH, BL, ACC)

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  const uint4 h = pack8(v);
  float hf[8], l[8];
  unpack8(h, hf);
#pragma unroll
  for (int e = 0; e < 8; ++e) l[e] = v[e] - hf[e];
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, pack8(l));
}

// one row's fragments for lane group fg: v[s][e] = row[32 s + 8 fg + e], rotated by `pos` (rotate-half RoPE: d pairs with d + D/2 = the same
// lane's k-step s + KS/2) when rope != null
template <int D>
__device__ __forceinline__ void load_row(float (&v)[D / 32][8], const float* __restrict__ rowp, const float2* __restrict__ rope, int pos, int fg) {
  constexpr int KS = D / 32;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float4 a = *reinterpret_cast<const float4*>(rowp + 32 * s + 8 * fg), b = *reinterpret_cast<const float4*>(rowp + 32 * s + 8 * fg + 4);
    v[s][0] = a.x; v[s][1] = a.y; v[s][2] = a.z; v[s][3] = a.w; v[s][4] = b.x; v[s][5] = b.y; v[s][6] = b.z; v[s][7] = b.w;
  }
  if (rope) {
    const float2* t = rope + (size_t)pos * (D / 2) + 8 * fg;
#pragma unroll
    for (int s = 0; s < KS / 2; ++s) {
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const float4 cs = *reinterpret_cast<const float4*>(t + 32 * s + 2 * e2);   // (cos, sin) of d = 32 s + 8 fg + 2 e2, + 1
        const float a0 = v[s][2 * e2], b0 = v[s + KS / 2][2 * e2], a1 = v[s][2 * e2 + 1], b1 = v[s + KS / 2][2 * e2 + 1];
        v[s][2 * e2] = a0 * cs.x - b0 * cs.y; v[s + KS / 2][2 * e2] = b0 * cs.x + a0 * cs.y;
        v[s][2 * e2 + 1] = a1 * cs.z - b1 * cs.w; v[s + KS / 2][2 * e2 + 1] = b1 * cs.z + a1 * cs.w;
      }
    }
  }
}

// gradient w.r.t. the rotated row (g[dt][r] = element 16 dt + 4 fg + r) -> gradient w.r.t. the un-rotated projection, stored at dst:
// rotated (q1, q2) = (a c - b s, b c + a s)  =>  da = g1 c + g2 s, db = -g1 s + g2 c
template <int D>
__device__ __forceinline__ void store_unrot(float* __restrict__ dst, const f32x4 (&g)[D / 16], const float2* __restrict__ rope, int pos, int fg) {
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

// 64 rows (row0 .., clamped to row_max) of width D at column col0 of src (fp32, row stride ld), rotated by the row's position when rope != null,
// split into bf16 hi / lo and written NATurally ([64][D + 8]) and / or TRansposed ([D][ALDT], columns in the permuted k-slot order).
// A thread takes RPI (2 or 4) consecutive rows x (4 d and their 4 rotation partners d + D/2): 256 items at head_dim 64 with RPI = 2.
template <int D, bool NAT, bool TR, int RPI = 2>
__device__ __forceinline__ void stage_rows(bf16_t* __restrict__ nh, bf16_t* __restrict__ nl, bf16_t* __restrict__ th, bf16_t* __restrict__ tl,
                                           const float* __restrict__ src_base, int ld, int col0, const float2* __restrict__ rope, int row0, int row_max, int tid) {
  constexpr int LDN = D + 8, G = D / 8;
  static_assert(RPI == 2 || RPI == 4, "rows per item");
  for (int i = tid; i < (64 / RPI) * G; i += 256) {
    const int rq = i / G, c4 = i % G;
    uint32_t hb[RPI][2][2], lb[RPI][2][2];   // [row][half: d / d + D/2][pair] bf16 bits
#pragma unroll
    for (int r = 0; r < RPI; ++r) {
      const int row = min(row0 + RPI * rq + r, row_max);
      const float* p = src_base + (size_t)row * ld + col0 + 4 * c4;
      float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + D / 2);
      if (rope) {
        const float2* t = rope + (size_t)row * (D / 2) + 4 * c4;
        const float4 cs0 = *reinterpret_cast<const float4*>(t), cs1 = *reinterpret_cast<const float4*>(t + 2);
        const float4 ra = make_float4(a.x * cs0.x - b.x * cs0.y, a.y * cs0.z - b.y * cs0.w, a.z * cs1.x - b.z * cs1.y, a.w * cs1.z - b.w * cs1.w);
        const float4 rb = make_float4(b.x * cs0.x + a.x * cs0.y, b.y * cs0.z + a.y * cs0.w, b.z * cs1.x + a.z * cs1.y, b.w * cs1.z + a.w * cs1.w);
        a = ra; b = rb;
      }
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const float4 x = hf ? b : a;
        const uint32_t h0 = pack_bf2(x.x, x.y), h1 = pack_bf2(x.z, x.w);
        hb[r][hf][0] = h0; hb[r][hf][1] = h1;
        lb[r][hf][0] = pack_bf2(x.x - bf_lo(h0), x.y - bf_hi(h0));
        lb[r][hf][1] = pack_bf2(x.z - bf_lo(h1), x.w - bf_hi(h1));
        if constexpr (NAT) {
          const int o = (RPI * rq + r) * LDN + 4 * c4 + hf * (D / 2);
          *reinterpret_cast<uint2*>(nh + o) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(nl + o) = make_uint2(lb[r][hf][0], lb[r][hf][1]);
        }
      }
    }
    if constexpr (TR) {
      const int R0 = RPI * rq, lr = R0 & 31, pos = (R0 & 32) + 8 * ((lr & 15) >> 2) + 4 * (lr >> 4) + (lr & 3);
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int d = 4 * c4 + e + hf * (D / 2), w = e >> 1, sh = (e & 1) * 16;
          const uint32_t h01 = ((hb[0][hf][w] >> sh) & 0xffffu) | (((hb[1][hf][w] >> sh) & 0xffffu) << 16);
          const uint32_t l01 = ((lb[0][hf][w] >> sh) & 0xffffu) | (((lb[1][hf][w] >> sh) & 0xffffu) << 16);
          if constexpr (RPI == 4) {
            const uint32_t h23 = ((hb[2][hf][w] >> sh) & 0xffffu) | (((hb[3][hf][w] >> sh) & 0xffffu) << 16);
            const uint32_t l23 = ((lb[2][hf][w] >> sh) & 0xffffu) | (((lb[3][hf][w] >> sh) & 0xffffu) << 16);
            *reinterpret_cast<uint2*>(th + d * ALDT + pos) = make_uint2(h01, h23);
            *reinterpret_cast<uint2*>(tl + d * ALDT + pos) = make_uint2(l01, l23);
          } else {
            *reinterpret_cast<uint32_t*>(th + d * ALDT + pos) = h01;
            *reinterpret_cast<uint32_t*>(tl + d * ALDT + pos) = l01;
          }
        }
    }
  }
}

#ifdef FASTVLA_AB_SWITCHES
__device__ int g_as_abl = 0;   // tools only: 1 = stage the first chunk only, 2 = no products, 4 = no score arithmetic
#define AS_ABL(bit) (g_as_abl & (bit))
#else
#define AS_ABL(bit) 0
#endif
#define AS_FRAG(BASE, OFF) (*reinterpret_cast<const bf16x8*>((BASE) + (OFF)))

template <int D> struct ASGeo {
  static constexpr int NT = D == 64 ? 2 : 1;        // 16-row tiles per wave
  static constexpr int RB = 64 * NT;                // rows per block (4 waves)
  static constexpr int NATB = ACH * (D + 8) * 2;    // bytes of one natural array (hi or lo)
  static constexpr int TRB = D * ALDT * 2;          // bytes of one transposed array
  // the K / V record of one (batch, kv head, 64-key chunk) in global memory, written once per layer by attn_prep_kv_kernel: the LDS images
  // (row padding included) in the order [K nat hi | K nat lo | V^T hi | V^T lo | V nat hi | V nat lo | K^T hi | K^T lo] -- the forward copies the
  // first four flat, the dq kernel the first two and the last four
  static constexpr int REC_FWD = 2 * NATB + 2 * TRB, REC_VN = REC_FWD, REC = 4 * NATB + 4 * TRB;
};

// flat 16-byte copy global -> LDS by the whole block
__device__ __forceinline__ void copy_flat(char* __restrict__ dst, const char* __restrict__ src, int bytes, int tid, int nthreads = 256) {
  for (int i = tid * 16; i < bytes; i += nthreads * 16) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
}

// K (rotated) and V of every (batch, kv head) as split bf16 in both forms, one record per 64-key chunk: what every q head's and every query block's
// forward / dq block would otherwise convert for itself (14 q heads / 2 kv heads x 3 query blocks: ~15x at the training shape)
template <int D>
__global__ __launch_bounds__(256) void attn_prep_kv_kernel(const float* __restrict__ qkv, int ld, int T, int heads, int kv_heads, const float2* __restrict__ rope,
                                                           char* __restrict__ rec, int nchunks) {
  using GEO = ASGeo<D>;
  extern __shared__ __attribute__((aligned(16))) char as_smem[];
  bf16_t* a0 = reinterpret_cast<bf16_t*>(as_smem);
  bf16_t* a1 = reinterpret_cast<bf16_t*>(as_smem + GEO::NATB);
  bf16_t* a2 = reinterpret_cast<bf16_t*>(as_smem + 2 * GEO::NATB);
  bf16_t* a3 = reinterpret_cast<bf16_t*>(as_smem + 2 * GEO::NATB + GEO::TRB);
  bf16_t* a4 = reinterpret_cast<bf16_t*>(as_smem + 2 * GEO::NATB + 2 * GEO::TRB);
  bf16_t* a5 = reinterpret_cast<bf16_t*>(as_smem + 3 * GEO::NATB + 2 * GEO::TRB);
  bf16_t* a6 = reinterpret_cast<bf16_t*>(as_smem + 4 * GEO::NATB + 2 * GEO::TRB);
  bf16_t* a7 = reinterpret_cast<bf16_t*>(as_smem + 4 * GEO::NATB + 3 * GEO::TRB);
  int bid = blockIdx.x;
  const int ch = bid % nchunks; bid /= nchunks;
  const int hk = bid % kv_heads;
  const int b = bid / kv_heads;
  const int tid = threadIdx.x, qd = heads * D, kd = kv_heads * D;
  const float* base = qkv + (size_t)b * T * ld;
  // (pad bytes of the LDS images are never read by a fragment: left as they are)
  stage_rows<D, true, true>(a0, a1, a6, a7, base, ld, qd + hk * D, rope, ch * ACH, T - 1, tid);
  stage_rows<D, true, true>(a4, a5, a2, a3, base, ld, qd + kd + hk * D, nullptr, ch * ACH, T - 1, tid);
  __syncthreads();
  char* out = rec + ((size_t)(b * kv_heads + hk) * nchunks + ch) * GEO::REC;
  for (int i = tid * 16; i < GEO::REC; i += 256 * 16) *reinterpret_cast<uint4*>(out + i) = *reinterpret_cast<const uint4*>(as_smem + i);
}


// ------------------------------------------------------------------------------------------------------------------------------- forward
// out = softmax(Q K^T * scale + causal / key-length mask) V as split bf16 (hi | lo), lse = max + log(sum) per (batch, head, query)
// GQ: one block = the q heads of ONE GQA group (a wave per head, 2 <= group <= 8) on one 16 NT-query block -- the group's K / V records go through
// LDS once for all of its heads (2.6x fewer L2 -> LDS bytes at 14 q / 2 kv heads, T = 320: the staging, not the products, bounds this kernel)
template <int D, bool GQ>
__global__ __launch_bounds__(GQ ? 512 : 256, 2) void attn_fwd_split_kernel(const float* __restrict__ qkv, int ld, bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo, int ldo,
                                                                 const int32_t* __restrict__ lens, int len_add, int T, int heads, int kv_heads, float scale,
                                                                 const float2* __restrict__ rope, float* __restrict__ lse, const char* __restrict__ rec, int nchunks) {
  using GEO = ASGeo<D>;
  constexpr int KS = D / 32, DT = D / 16, NT = GEO::NT, RB = GEO::RB, LDN = D + 8;
  extern __shared__ __attribute__((aligned(16))) char as_smem[];
  bf16_t* sKh = reinterpret_cast<bf16_t*>(as_smem);
  bf16_t* sKl = reinterpret_cast<bf16_t*>(as_smem + GEO::NATB);
  bf16_t* sVTh = reinterpret_cast<bf16_t*>(as_smem + 2 * GEO::NATB);
  bf16_t* sVTl = reinterpret_cast<bf16_t*>(as_smem + 2 * GEO::NATB + GEO::TRB);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int QB = GQ ? 16 * NT : RB;   // queries per block (GQ: every wave takes the same queries of its own head)
  const int qblocks = (T + QB - 1) / QB;
  // the query blocks with the longest causal key range first: the launch is ~1.3 rounds of resident blocks, and its tail should be the short ones
  // (blocks of one GQA group deliberately NOT gathered on one XCD: with xcd_remap the forward takes 68 instead of 58 us and dq 134 instead of 116 --
  // the seven heads' copies of a K / V record are better spread over eight L2s than served by one; AS_ABL(8) in the tools build is that A/B)
  int bid = AS_ABL(8) ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  int h, hk, b, qb;
  if constexpr (GQ) {
    const int grp = heads / kv_heads, nb_ = (int)(gridDim.x / (kv_heads * qblocks));
    hk = bid % kv_heads; bid /= kv_heads;
    b = bid % nb_; qb = qblocks - 1 - bid / nb_;
    h = hk * grp + wid;
  } else {
    h = bid % heads; bid /= heads;
    b = bid % (int)(gridDim.x / (heads * qblocks));
    qb = qblocks - 1 - bid / (int)(gridDim.x / (heads * qblocks));
    hk = h / (heads / kv_heads);
  }
  int len = lens ? lens[b] + len_add : T;
  len = max(1, min(len, T));
  const int q0w = GQ ? qb * QB : qb * RB + wid * (16 * NT);
  const int nthr = (int)blockDim.x;
  const float* base = qkv + (size_t)b * T * ld;

  bf16x8 Qh[NT][KS], Ql[NT][KS];
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int qc = min(q0w + 16 * u + fr, T - 1);
    float v[KS][8];
    load_row<D>(v, base + (size_t)qc * ld + h * D, rope, qc, fg);
#pragma unroll
    for (int s = 0; s < KS; ++s) split8(v[s], Qh[u][s], Ql[u][s]);
  }
  f32x4 o[NT][DT];
  float m_run[NT], l_run[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    m_run[u] = -1e30f; l_run[u] = 0.f;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int kend = min(len, qb * QB + QB);   // causal: keys beyond the block's last query are never visible
  for (int k0 = 0; k0 < kend; k0 += ACH) {
    __syncthreads();
    copy_flat(as_smem, rec + ((size_t)(b * kv_heads + hk) * nchunks + k0 / ACH) * GEO::REC, GEO::REC_FWD, tid, nthr);
    __syncthreads();
#pragma unroll 1
    for (int kp = 0; kp < ACH / 32; ++kp) {
      const int kb = k0 + 32 * kp;
      if (kb > q0w + 16 * NT - 1 || kb >= len || q0w >= T || AS_ABL(2)) break;   // wave-uniform: the rest of the chunk is above the diagonal / past the prompt (or the wave owns no query)
      f32x4 sacc[NT][2];
#pragma unroll
      for (int u = 0; u < NT; ++u) { sacc[u][0] = f32x4{0.f, 0.f, 0.f, 0.f}; sacc[u][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int off = (32 * kp + 16 * t + fr) * LDN + 32 * s + 8 * fg;
          const bf16x8 kh = AS_FRAG(sKh, off), kl = AS_FRAG(sKl, off);
#pragma unroll
          for (int u = 0; u < NT; ++u) { AS_MM3(sacc[u][t], kh, kl, Qh[u][s], Ql[u][s]); }
        }
      bf16x8 Ph[NT], Pl[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int qg = q0w + 16 * u + fr;
        float sc[8], mloc = -1e30f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kg = kb + 16 * t + 4 * fg + r;
            sc[4 * t + r] = (kg <= qg && kg < len) ? sacc[u][t][r] * scale : -1e30f;
            mloc = fmaxf(mloc, sc[4 * t + r]);
          }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run[u], mloc);
        const float alpha = __expf(m_run[u] - m_new);
        m_run[u] = m_new;
        float pv[8], psum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          pv[e] = sc[e] > -1e29f ? __expf(sc[e] - m_new) : 0.f;
          psum += pv[e];
        }
        l_run[u] = l_run[u] * alpha + psum;   // per-lane partial; the four key groups are summed at the end
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[u][dt] *= alpha;
        split8(pv, Ph[u], Pl[u]);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int off = (16 * dt + fr) * ALDT + 32 * kp + 8 * fg;
        const bf16x8 vh = AS_FRAG(sVTh, off), vl = AS_FRAG(sVTl, off);
#pragma unroll
        for (int u = 0; u < NT; ++u) { AS_MM3(o[u][dt], vh, vl, Ph[u], Pl[u]); }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int qg = q0w + 16 * u + fr;
    float l = l_run[u];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    if (qg >= T) continue;
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (lse && fg == 0) lse[((size_t)b * heads + h) * T + qg] = m_run[u] + __logf(l);
    const size_t ob = ((size_t)b * T + qg) * ldo + h * D + 4 * fg;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const float x0 = o[u][dt][0] * inv, x1 = o[u][dt][1] * inv, x2 = o[u][dt][2] * inv, x3 = o[u][dt][3] * inv;
      const uint32_t h0 = pack_bf2(x0, x1), h1 = pack_bf2(x2, x3);
      *reinterpret_cast<uint2*>(out_hi + ob + 16 * dt) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(out_lo + ob + 16 * dt) = make_uint2(pack_bf2(x0 - bf_lo(h0), x1 - bf_hi(h0)), pack_bf2(x2 - bf_lo(h1), x3 - bf_hi(h1)));
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------ backward
//   P = exp(S * scale - lse),  dP = dO . V^T,  delta_i = sum_d dO_id O_id,  dS = P o (dP - delta) * scale,
//   dQ = dS . K,  dK = dS^T . Q,  dV = P^T . dO        (Q, K = the ROTATED projections; the gradient is rotated back on the way out)
// dq kernel: block = RB queries of one (batch, q head); K (natural and transposed) and V chunks through LDS; also writes delta.
template <int D, bool GQ>   // GQ: as attn_fwd_split_kernel -- one block per GQA group and 16 NT-query block, a wave per q head
__global__ __launch_bounds__(GQ ? 512 : 256, GQ ? 1 : 2) void attn_bwd_dq_split_kernel(const float* __restrict__ qkv, int ld, const bf16_t* __restrict__ o_hi, const bf16_t* __restrict__ o_lo,
                                                                    int ldo, const float* __restrict__ dO, int lddo, const float* __restrict__ lse,
                                                                    float* __restrict__ delta, float* __restrict__ dqkv, const int32_t* __restrict__ lens,
                                                                    int len_add, int T, int heads, int kv_heads, float scale, const float2* __restrict__ rope,
                                                                    const char* __restrict__ rec, int nchunks) {
  using GEO = ASGeo<D>;
  constexpr int KS = D / 32, DT = D / 16, NT = GEO::NT, RB = GEO::RB, LDN = D + 8;
  extern __shared__ __attribute__((aligned(16))) char as_smem[];
  bf16_t* sKh = reinterpret_cast<bf16_t*>(as_smem);
  bf16_t* sKl = reinterpret_cast<bf16_t*>(as_smem + GEO::NATB);
  bf16_t* sVh = reinterpret_cast<bf16_t*>(as_smem + 2 * GEO::NATB);
  bf16_t* sVl = reinterpret_cast<bf16_t*>(as_smem + 3 * GEO::NATB);
  bf16_t* sKTh = reinterpret_cast<bf16_t*>(as_smem + 4 * GEO::NATB);
  bf16_t* sKTl = reinterpret_cast<bf16_t*>(as_smem + 4 * GEO::NATB + GEO::TRB);
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  constexpr int QB = GQ ? 16 * NT : RB;
  const int qblocks = (T + QB - 1) / QB;
  int bid = AS_ABL(8) ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;   // (query blocks with the longest causal range first, as in the forward)
  int h, hk, b, qb;
  if constexpr (GQ) {
    const int grp = heads / kv_heads, nb_ = (int)(gridDim.x / (kv_heads * qblocks));
    hk = bid % kv_heads; bid /= kv_heads;
    b = bid % nb_; qb = qblocks - 1 - bid / nb_;
    h = hk * grp + wid;
  } else {
    h = bid % heads; bid /= heads;
    b = bid % (int)(gridDim.x / (heads * qblocks));
    qb = qblocks - 1 - bid / (int)(gridDim.x / (heads * qblocks));
    hk = h / (heads / kv_heads);
  }
  int len = lens ? lens[b] + len_add : T;
  len = max(1, min(len, T));
  const int q0w = GQ ? qb * QB : qb * RB + wid * (16 * NT);
  const int nthr = (int)blockDim.x;
  const float* base = qkv + (size_t)b * T * ld;

  bf16x8 Qh[NT][KS], Ql[NT][KS], Gh[NT][KS], Gl[NT][KS];   // Q (rotated) and dO rows as B operands
  float dl[NT], my_lse[NT];
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int qg = q0w + 16 * u + fr, qc = min(qg, T - 1);
    const size_t rowq = (size_t)b * T + qc;
    float v[KS][8];
    load_row<D>(v, base + (size_t)qc * ld + h * D, rope, qc, fg);
#pragma unroll
    for (int s = 0; s < KS; ++s) split8(v[s], Qh[u][s], Ql[u][s]);
    load_row<D>(v, dO + rowq * lddo + h * D, nullptr, 0, fg);
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      float oh[8], ol[8];
      unpack8(*reinterpret_cast<const uint4*>(o_hi + rowq * ldo + h * D + 32 * s + 8 * fg), oh);
      unpack8(*reinterpret_cast<const uint4*>(o_lo + rowq * ldo + h * D + 32 * s + 8 * fg), ol);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc += v[s][e] * (oh[e] + ol[e]);
      split8(v[s], Gh[u][s], Gl[u][s]);
    }
    acc += __shfl_xor(acc, 16, 64);
    acc += __shfl_xor(acc, 32, 64);
    dl[u] = acc;
    my_lse[u] = lse[((size_t)b * heads + h) * T + qc];
    if (fg == 0 && qg < T) delta[((size_t)b * heads + h) * T + qg] = acc;
  }
  f32x4 dq[NT][DT];
#pragma unroll
  for (int u = 0; u < NT; ++u)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) dq[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int kend = min(len, qb * QB + QB);
  for (int k0 = 0; k0 < kend; k0 += ACH) {
    __syncthreads();
    if (!AS_ABL(1) || k0 == 0) {
      const char* r0 = rec + ((size_t)(b * kv_heads + hk) * nchunks + k0 / ACH) * GEO::REC;
      copy_flat(as_smem, r0, 2 * GEO::NATB, tid, nthr);                                              // K natural
      copy_flat(as_smem + 2 * GEO::NATB, r0 + GEO::REC_VN, 2 * GEO::NATB + 2 * GEO::TRB, tid, nthr);  // V natural, K transposed
    }
    __syncthreads();
#pragma unroll 1
    for (int kp = 0; kp < ACH / 32; ++kp) {
      const int kb = k0 + 32 * kp;
      if (kb > q0w + 16 * NT - 1 || kb >= len || q0w >= T || AS_ABL(2)) break;
      f32x4 sacc[NT][2], dpacc[NT][2];
#pragma unroll
      for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t) { sacc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f}; dpacc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int off = (32 * kp + 16 * t + fr) * LDN + 32 * s + 8 * fg;
          const bf16x8 kh = AS_FRAG(sKh, off), kl = AS_FRAG(sKl, off), vh = AS_FRAG(sVh, off), vl = AS_FRAG(sVl, off);
#pragma unroll
          for (int u = 0; u < NT; ++u) {
            AS_MM3(sacc[u][t], kh, kl, Qh[u][s], Ql[u][s]);
            AS_MM3(dpacc[u][t], vh, vl, Gh[u][s], Gl[u][s]);
          }
        }
      bf16x8 Sh[NT], Sl[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int qg = q0w + 16 * u + fr;
        float ds[8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int kg = kb + 16 * t + 4 * fg + r;
            const float p = AS_ABL(4) ? sacc[u][t][r] : (kg <= qg && kg < len) ? __expf(sacc[u][t][r] * scale - my_lse[u]) : 0.f;
            ds[4 * t + r] = AS_ABL(4) ? p + dpacc[u][t][r] : p * (dpacc[u][t][r] - dl[u]) * scale;
          }
        split8(ds, Sh[u], Sl[u]);
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int off = (16 * dt + fr) * ALDT + 32 * kp + 8 * fg;
        const bf16x8 th = AS_FRAG(sKTh, off), tl = AS_FRAG(sKTl, off);
#pragma unroll
        for (int u = 0; u < NT; ++u) { AS_MM3(dq[u][dt], th, tl, Sh[u], Sl[u]); }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int qg = q0w + 16 * u + fr;
    if (qg >= T) continue;
    store_unrot<D>(dqkv + ((size_t)b * T + qg) * ld + h * D, dq[u], rope, qg, fg);
  }
}

// dkv kernel: block = RB keys of one (batch, kv head) [PART: and ONE q head of the group -- its share goes to part[hh][row][2 kd], summed over
// hh in a fixed order by the reduce kernel]; loops over the query chunks from the block's first key on (Q and dO, natural and transposed, in LDS)
template <int D, bool PART>
__global__ __launch_bounds__(256, D == 64 ? 2 : 1) void attn_bwd_dkv_split_kernel(const float* __restrict__ qkv, int ld, const float* __restrict__ dO, int lddo,
                                                                     const float* __restrict__ lse, const float* __restrict__ delta, float* __restrict__ dqkv,
                                                                     const int32_t* __restrict__ lens, int len_add, int T, int heads, int kv_heads,
                                                                     float scale, const float2* __restrict__ rope, float* __restrict__ part, long part_stride) {
  using GEO = ASGeo<D>;
  constexpr int KS = D / 32, DT = D / 16, NT = GEO::NT, RB = GEO::RB, LDN = D + 8;
  extern __shared__ __attribute__((aligned(16))) char as_smem[];
  bf16_t* sQh = reinterpret_cast<bf16_t*>(as_smem);
  bf16_t* sQl = reinterpret_cast<bf16_t*>(as_smem + GEO::NATB);
  bf16_t* sGh = reinterpret_cast<bf16_t*>(as_smem + 2 * GEO::NATB);
  bf16_t* sGl = reinterpret_cast<bf16_t*>(as_smem + 3 * GEO::NATB);
  bf16_t* sQTh = reinterpret_cast<bf16_t*>(as_smem + 4 * GEO::NATB);
  bf16_t* sQTl = reinterpret_cast<bf16_t*>(as_smem + 4 * GEO::NATB + GEO::TRB);
  bf16_t* sGTh = reinterpret_cast<bf16_t*>(as_smem + 4 * GEO::NATB + 2 * GEO::TRB);
  bf16_t* sGTl = reinterpret_cast<bf16_t*>(as_smem + 4 * GEO::NATB + 3 * GEO::TRB);
  float* sLse = reinterpret_cast<float*>(as_smem + 4 * GEO::NATB + 4 * GEO::TRB);
  float* sDel = sLse + ACH;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int kblocks = (T + RB - 1) / RB;
  const int grp = heads / kv_heads;
  // key blocks in ascending order: the first sees every query chunk (the longest loop), the last only its own
  int bid = AS_ABL(8) ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  int hh0 = 0, hh1 = grp;
  if constexpr (PART) { hh0 = bid % grp; hh1 = hh0 + 1; bid /= grp; }
  const int hk = bid % kv_heads; bid /= kv_heads;
  const int nb_ = (int)(gridDim.x / ((PART ? grp : 1) * kv_heads * kblocks));
  const int b = bid % nb_;
  const int kbk = bid / nb_;
  int len = lens ? lens[b] + len_add : T;
  len = max(1, min(len, T));
  const int k0w = kbk * RB + wid * (16 * NT);
  const int qd = heads * D, kd = kv_heads * D;
  const float* base = qkv + (size_t)b * T * ld;

  bf16x8 Kh[NT][KS], Kl[NT][KS], Vh[NT][KS], Vl[NT][KS];   // the wave's keys as B operands (lane fr = key)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int kc = min(k0w + 16 * t + fr, T - 1);
    float v[KS][8];
    load_row<D>(v, base + (size_t)kc * ld + qd + hk * D, rope, kc, fg);
#pragma unroll
    for (int s = 0; s < KS; ++s) split8(v[s], Kh[t][s], Kl[t][s]);
    load_row<D>(v, base + (size_t)kc * ld + qd + kd + hk * D, nullptr, 0, fg);
#pragma unroll
    for (int s = 0; s < KS; ++s) split8(v[s], Vh[t][s], Vl[t][s]);
  }
  f32x4 dk[NT][DT], dv[NT][DT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dk[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  const int qstart = (kbk * RB) / ACH * ACH;       // causal: queries before the block's first key never see it
  if (kbk * RB < len) {
    for (int hh = hh0; hh < hh1; ++hh) {
      const int h = hk * grp + hh;
      for (int qc0 = qstart; qc0 < len; qc0 += ACH) {
        __syncthreads();
        stage_rows<D, true, true>(sQh, sQl, sQTh, sQTl, base, ld, h * D, rope, qc0, T - 1, tid);
        stage_rows<D, true, true>(sGh, sGl, sGTh, sGTl, dO + (size_t)b * T * lddo, lddo, h * D, nullptr, qc0, T - 1, tid);
        if (tid < ACH) {
          const int qi = min(qc0 + tid, T - 1);
          sLse[tid] = lse[((size_t)b * heads + h) * T + qi];
          sDel[tid] = delta[((size_t)b * heads + h) * T + qi];
        }
        __syncthreads();
#pragma unroll 1
        for (int qp = 0; qp < ACH / 32; ++qp) {
          const int qb = qc0 + 32 * qp;
          if (qb >= len || AS_ABL(2)) break;
          if (qb + 31 < k0w || k0w >= len) continue;   // wave-uniform: every query of the step precedes every key of the wave / masked keys
          f32x4 sacc[NT][2], dpacc[NT][2];             // [key tile t][query tile v]: D[i = query 4 fg + r][j = key fr]
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int v = 0; v < 2; ++v) { sacc[t][v] = f32x4{0.f, 0.f, 0.f, 0.f}; dpacc[t][v] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
          for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
              const int off = (32 * qp + 16 * v + fr) * LDN + 32 * s + 8 * fg;
              const bf16x8 qh = AS_FRAG(sQh, off), ql = AS_FRAG(sQl, off), gh = AS_FRAG(sGh, off), gl = AS_FRAG(sGl, off);
#pragma unroll
              for (int t = 0; t < NT; ++t) {
                AS_MM3(sacc[t][v], qh, ql, Kh[t][s], Kl[t][s]);
                AS_MM3(dpacc[t][v], gh, gl, Vh[t][s], Vl[t][s]);
              }
            }
          bf16x8 Ph[NT], Pl[NT], Sh[NT], Sl[NT];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int kg = k0w + 16 * t + fr;
            float p[8], ds[8];
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int ql_ = 32 * qp + 16 * v + 4 * fg + r, qi = qc0 + ql_;
                const bool vis = kg <= qi && kg < len && qi < len;
                p[4 * v + r] = vis ? __expf(sacc[t][v][r] * scale - sLse[ql_]) : 0.f;
                ds[4 * v + r] = p[4 * v + r] * (dpacc[t][v][r] - sDel[ql_]) * scale;
              }
            split8(p, Ph[t], Pl[t]);
            split8(ds, Sh[t], Sl[t]);
          }
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const int off = (16 * dt + fr) * ALDT + 32 * qp + 8 * fg;
            const bf16x8 gth = AS_FRAG(sGTh, off), gtl = AS_FRAG(sGTl, off), qth = AS_FRAG(sQTh, off), qtl = AS_FRAG(sQTl, off);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              AS_MM3(dv[t][dt], gth, gtl, Ph[t], Pl[t]);
              AS_MM3(dk[t][dt], qth, qtl, Sh[t], Sl[t]);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int kg = k0w + 16 * t + fr;
    if (kg >= T) continue;
    if constexpr (PART) {
      float* prow = part + (size_t)hh0 * part_stride + ((size_t)b * T + kg) * (2 * kd);
      store_unrot<D>(prow + hk * D, dk[t], rope, kg, fg);
      store_unrot<D>(prow + kd + hk * D, dv[t], nullptr, 0, fg);
    } else {
      float* drow = dqkv + ((size_t)b * T + kg) * ld;
      store_unrot<D>(drow + qd + hk * D, dk[t], rope, kg, fg);
      store_unrot<D>(drow + qd + kd + hk * D, dv[t], nullptr, 0, fg);
    }
  }
}

template <typename K>
int set_lds(K kernel, int bytes) {
  FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return FV_OK;
}

}  // namespace

// LDS bytes of the three kernels (forward: K natural + V transposed; dq: K natural + transposed, V natural; dkv: Q and dO in both forms + row statistics)
template <int D> constexpr int as_fwd_lds() { return 2 * ASGeo<D>::NATB + 2 * ASGeo<D>::TRB; }
template <int D> constexpr int as_dq_lds() { return 4 * ASGeo<D>::NATB + 2 * ASGeo<D>::TRB; }
template <int D> constexpr int as_dkv_lds() { return 4 * ASGeo<D>::NATB + 4 * ASGeo<D>::TRB + 2 * ACH * 4; }

#ifdef FASTVLA_AB_SWITCHES
static void as_set_abl() {
  static bool done = false;
  if (done) return;
  done = true;
  const char* e = fv_ab_env("FASTVLA_ATTN_ABL");
  const int v = e ? atoi(e) : 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_as_abl), &v, sizeof(int));
}
#endif

size_t attention_split_scratch_bytes(int B, int T, int kv_heads, int D) {
  const size_t rec = D == 64 ? ASGeo<64>::REC : ASGeo<128>::REC;
  return (size_t)B * kv_heads * ((T + ACH - 1) / ACH) * rec;
}

static int launch_prep_kv(const float* qkv, int ld, int B, int T, int heads, int kv_heads, int D, const float2* rope, char* rec, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    FV_TRY_RC(set_lds(attn_prep_kv_kernel<64>, ASGeo<64>::REC));
    FV_TRY_RC(set_lds(attn_prep_kv_kernel<128>, ASGeo<128>::REC));
    attr = true;
  }
  const int nch = (T + ACH - 1) / ACH;
  if (D == 64) hipLaunchKernelGGL(attn_prep_kv_kernel<64>, dim3(B * kv_heads * nch), dim3(256), ASGeo<64>::REC, s, qkv, ld, T, heads, kv_heads, rope, rec, nch);
  else hipLaunchKernelGGL(attn_prep_kv_kernel<128>, dim3(B * kv_heads * nch), dim3(256), ASGeo<128>::REC, s, qkv, ld, T, heads, kv_heads, rope, rec, nch);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_attention_split_fwd(const float* qkv, int ld, bf16_t* out_hi, bf16_t* out_lo, int ldo, int B, int T, int heads, int kv_heads, int D,
                               const int32_t* lens, int len_add, float scale, const float2* rope, float* lse, void* scratch, hipStream_t s) {
  if (!qkv || !out_hi || !out_lo || !scratch) return fv_fail(FV_ERR_ARG, "attention_split_fwd: null pointer");
  if (D != 64 && D != 128) return fv_fail(FV_ERR_UNSUPPORTED, "attention_split_fwd: head_dim must be 64 or 128 (got %d)", D);
  if (B <= 0 || T <= 0 || heads <= 0 || kv_heads <= 0 || heads % kv_heads || ld % 4 || ld < (heads + 2 * kv_heads) * D || ldo % 8 || ldo < heads * D)
    return fv_fail(FV_ERR_ARG, "attention_split_fwd: bad shape");
  static bool attr = false;
  if (!attr) {
    FV_TRY_RC(set_lds(attn_fwd_split_kernel<64, false>, as_fwd_lds<64>()));
    FV_TRY_RC(set_lds(attn_fwd_split_kernel<128, false>, as_fwd_lds<128>()));
    FV_TRY_RC(set_lds(attn_fwd_split_kernel<64, true>, as_fwd_lds<64>()));
    FV_TRY_RC(set_lds(attn_fwd_split_kernel<128, true>, as_fwd_lds<128>()));
    attr = true;
  }
#ifdef FASTVLA_AB_SWITCHES
  as_set_abl();
#endif
  char* rec = static_cast<char*>(scratch);
  const int nch = (T + ACH - 1) / ACH, grp = heads / kv_heads;
  FV_TRY_RC(launch_prep_kv(qkv, ld, B, T, heads, kv_heads, D, rope, rec, s));
  static const bool no_gq = fv_ab_env("FASTVLA_ATTN_NO_GQ") != nullptr;   // A/B: one block per q head
  const bool gq = !no_gq && grp >= 2 && grp <= 8;
  if (D == 64) {
    if (gq) {
      const long nb = (long)B * kv_heads * ((T + 16 * ASGeo<64>::NT - 1) / (16 * ASGeo<64>::NT));
      hipLaunchKernelGGL((attn_fwd_split_kernel<64, true>), dim3((unsigned)nb), dim3(64 * grp), as_fwd_lds<64>(), s, qkv, ld, out_hi, out_lo, ldo, lens, len_add, T, heads, kv_heads, scale, rope, lse, rec, nch);
    } else {
      const long nb = (long)B * heads * ((T + ASGeo<64>::RB - 1) / ASGeo<64>::RB);
      hipLaunchKernelGGL((attn_fwd_split_kernel<64, false>), dim3((unsigned)nb), dim3(256), as_fwd_lds<64>(), s, qkv, ld, out_hi, out_lo, ldo, lens, len_add, T, heads, kv_heads, scale, rope, lse, rec, nch);
    }
  } else {
    if (gq) {
      const long nb = (long)B * kv_heads * ((T + 16 * ASGeo<128>::NT - 1) / (16 * ASGeo<128>::NT));
      hipLaunchKernelGGL((attn_fwd_split_kernel<128, true>), dim3((unsigned)nb), dim3(64 * grp), as_fwd_lds<128>(), s, qkv, ld, out_hi, out_lo, ldo, lens, len_add, T, heads, kv_heads, scale, rope, lse, rec, nch);
    } else {
      const long nb = (long)B * heads * ((T + ASGeo<128>::RB - 1) / ASGeo<128>::RB);
      hipLaunchKernelGGL((attn_fwd_split_kernel<128, false>), dim3((unsigned)nb), dim3(256), as_fwd_lds<128>(), s, qkv, ld, out_hi, out_lo, ldo, lens, len_add, T, heads, kv_heads, scale, rope, lse, rec, nch);
    }
  }
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

// dq, then dk / dv (PART when `part` is given and the group has more than one q head); the caller (launch_attention_bwd) validated the shapes and
// runs the fixed-order reduce of the parts

int launch_attention_split_bwd(const float* qkv, int ld, const bf16_t* o_hi, const bf16_t* o_lo, int ldo, const float* dO, int lddo, const float* lse,
                               float* delta, float* dqkv, int B, int T, int heads, int kv_heads, int D, const int32_t* lens, int len_add, float scale,
                               const float2* rope, hipStream_t s, float* part, long pstride, void* scratch) {
  if (D != 64 && D != 128) return fv_fail(FV_ERR_UNSUPPORTED, "attention_split_bwd: head_dim must be 64 or 128 (got %d)", D);
  if (!scratch) return fv_fail(FV_ERR_ARG, "attention_split_bwd: no scratch");
#ifdef FASTVLA_AB_SWITCHES
  as_set_abl();
#endif
  char* rec = static_cast<char*>(scratch);
  const int nch = (T + ACH - 1) / ACH;
  FV_TRY_RC(launch_prep_kv(qkv, ld, B, T, heads, kv_heads, D, rope, rec, s));
  static bool attr = false;
  if (!attr) {
    FV_TRY_RC(set_lds(attn_bwd_dq_split_kernel<64, false>, as_dq_lds<64>()));
    FV_TRY_RC(set_lds(attn_bwd_dq_split_kernel<128, false>, as_dq_lds<128>()));
    FV_TRY_RC(set_lds(attn_bwd_dq_split_kernel<64, true>, as_dq_lds<64>()));
    FV_TRY_RC(set_lds(attn_bwd_dq_split_kernel<128, true>, as_dq_lds<128>()));
    FV_TRY_RC(set_lds(attn_bwd_dkv_split_kernel<64, true>, as_dkv_lds<64>()));
    FV_TRY_RC(set_lds(attn_bwd_dkv_split_kernel<64, false>, as_dkv_lds<64>()));
    FV_TRY_RC(set_lds(attn_bwd_dkv_split_kernel<128, true>, as_dkv_lds<128>()));
    FV_TRY_RC(set_lds(attn_bwd_dkv_split_kernel<128, false>, as_dkv_lds<128>()));
    attr = true;
  }
  const int grp = heads / kv_heads;
  const bool parts = part != nullptr && grp > 1;
  static const bool no_gq = fv_ab_env("FASTVLA_ATTN_NO_GQ") != nullptr;   // A/B: one block per q head (dq: 118 -> 86 us with the group's heads in one block, forward 58 -> 49)
  const bool gqm = !no_gq && grp >= 2 && grp <= 8;
  if (D == 64) {
    const int blocks = (T + ASGeo<64>::RB - 1) / ASGeo<64>::RB;
    const dim3 gq(B * heads * blocks), gk(B * kv_heads * blocks * (parts ? grp : 1));
    if (gqm) hipLaunchKernelGGL((attn_bwd_dq_split_kernel<64, true>), dim3(B * kv_heads * ((T + 16 * ASGeo<64>::NT - 1) / (16 * ASGeo<64>::NT))), dim3(64 * grp), as_dq_lds<64>(), s, qkv, ld, o_hi, o_lo, ldo, dO, lddo, lse, delta, dqkv, lens, len_add, T, heads, kv_heads, scale, rope, rec, nch);
    else hipLaunchKernelGGL((attn_bwd_dq_split_kernel<64, false>), gq, dim3(256), as_dq_lds<64>(), s, qkv, ld, o_hi, o_lo, ldo, dO, lddo, lse, delta, dqkv, lens, len_add, T, heads, kv_heads, scale, rope, rec, nch);
    if (parts) hipLaunchKernelGGL((attn_bwd_dkv_split_kernel<64, true>), gk, dim3(256), as_dkv_lds<64>(), s, qkv, ld, dO, lddo, lse, delta, dqkv, lens, len_add, T, heads, kv_heads, scale, rope, part, pstride);
    else hipLaunchKernelGGL((attn_bwd_dkv_split_kernel<64, false>), gk, dim3(256), as_dkv_lds<64>(), s, qkv, ld, dO, lddo, lse, delta, dqkv, lens, len_add, T, heads, kv_heads, scale, rope, part, pstride);
  } else {
    const int blocks = (T + ASGeo<128>::RB - 1) / ASGeo<128>::RB;
    const dim3 gq(B * heads * blocks), gk(B * kv_heads * blocks * (parts ? grp : 1));
    if (gqm) hipLaunchKernelGGL((attn_bwd_dq_split_kernel<128, true>), dim3(B * kv_heads * ((T + 16 * ASGeo<128>::NT - 1) / (16 * ASGeo<128>::NT))), dim3(64 * grp), as_dq_lds<128>(), s, qkv, ld, o_hi, o_lo, ldo, dO, lddo, lse, delta, dqkv, lens, len_add, T, heads, kv_heads, scale, rope, rec, nch);
    else hipLaunchKernelGGL((attn_bwd_dq_split_kernel<128, false>), gq, dim3(256), as_dq_lds<128>(), s, qkv, ld, o_hi, o_lo, ldo, dO, lddo, lse, delta, dqkv, lens, len_add, T, heads, kv_heads, scale, rope, rec, nch);
    if (parts) hipLaunchKernelGGL((attn_bwd_dkv_split_kernel<128, true>), gk, dim3(256), as_dkv_lds<128>(), s, qkv, ld, dO, lddo, lse, delta, dqkv, lens, len_add, T, heads, kv_heads, scale, rope, part, pstride);
    else hipLaunchKernelGGL((attn_bwd_dkv_split_kernel<128, false>), gk, dim3(256), as_dkv_lds<128>(), s, qkv, ld, dO, lddo, lse, delta, dqkv, lens, len_add, T, heads, kv_heads, scale, rope, part, pstride);
  }
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

}  // namespace fv
