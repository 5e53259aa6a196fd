// convffn_fused.hip -- the pointwise half of FastViT's ConvFFN in ONE kernel:
//     out = res + ls * ( fc2( GELU( fc1(x) + b1 ) ) + b2 )          x = BN(dw7x7(.)) output, [M][C] bf16 NHWC rows
// ([UNVENDORED] mci.py ConvFFN.forward + RepMixerBlock/AttentionBlock layer-scale residual).  The 4C-wide hidden tensor
// never reaches HBM (unfused it is 0.8-3.2 GB written and re-read per layer at B = 64, which made the K = 96..384
// GEMMs HBM/epilogue-bound at 150-410 TFLOP/s).
//
// Formulation (per wave: MT tiles of 16 pixels; per block: 4 waves sharing the weight stream):
//   for each chunk of 32 hidden units:
//     H^T = W1[chunk] . x^T        A = W1 rows (hidden on the MFMA row, ds_read_b128), B = x fragments held in registers
//     GELU in registers; the C/D map (col = pixel, row = 4*(lane>>4)+r = hidden) is already the B operand of the second
//     product once W2's hidden columns are permuted the same way (done at pack time)
//     out^T += W2[:, chunk] . H^T  A = W2 rows (output channel on the MFMA row), accumulators live across all chunks
// so neither the hidden activations nor the output accumulators touch LDS; LDS only carries the double-buffered weight
// chunks (register-staged, one barrier per chunk, rows padded by 32 B = conflict-free b128 fragment reads).
// One wave per SIMD, ~400 registers: x fragments MT*C/8, output accumulators MT*C/4.  Bound: MFMA (4*C*4C flop per
// pixel) with the GELU (1 exp + 1 rcp per hidden element) hidden in the MFMA issue gaps for C >= 384.
#include "kernels.h"

namespace fv {
namespace {

struct FfnParams {
  const bf16_t* x; const bf16_t* w1; const float* b1; const bf16_t* w2p; const float* b2; const float* ls;
  const bf16_t* res; bf16_t* out; int M, nchunks;
};

// The 192 output accumulators are pinned to the accumulator half of the register file ("+a") and the hidden-tile
// accumulators to the architectural half ("+v"): left to itself hipcc (ROCm 7.2) shuffled ~320 v_accvgpr_read/write/mov
// per 96 MFMAs between the two halves at this register pressure.  The MFMA D -> MFMA C chain needs no wait states;
// the one VALU-written operand (the packed GELU output) is followed by an explicit s_nop before its first MFMA.
__device__ __forceinline__ void mfma_acc_a(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// Hidden-tile accumulators: HA = false keeps them in architectural registers (the GELU reads them directly); HA = true puts
// them in the accumulator half too.  The 8-row-tile variants need that: with 64 hidden accumulators among the VGPRs hipcc
// ran out, and its spill copies (v_accvgpr_write of an accumulator right behind one of these opaque asm MFMAs) read stale
// data -- the XDL-write -> VALU-read wait states are invisible to the compiler here.
template <bool HA>
__device__ __forceinline__ void mfma_acc_h(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  if constexpr (HA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// first k-step of a hidden tile: C = 0 as an inline constant, so no VALU-written zero feeds the MFMA
template <bool HA>
__device__ __forceinline__ void mfma_init_h(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  if constexpr (HA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
}
// after the last k-step of a hidden tile the GELU (VALU) reads the accumulator -- hipcc pads nothing after inline asm, so
// the XDL-write -> VALU-read wait states are spelled out, tied to the accumulator
template <bool HA>
__device__ __forceinline__ void settle_acc(f32x4& acc) {
  if constexpr (HA) asm volatile("s_nop 7\n\ts_nop 3" : "+a"(acc));
  else asm volatile("s_nop 7\n\ts_nop 3" : "+v"(acc));
}
// VALU-written B operand (packed GELU output) -> MFMA read: tie the wait states to the operand
__device__ __forceinline__ void settle_operand(bf16x8& v) { asm volatile("s_nop 3" : "+v"(v)); }

template <int C, int MT>
__global__ __launch_bounds__(256, 1) void convffn_kernel(FfnParams p) {
#ifndef FFN_HA
#define FFN_HA (MT >= 8)
#endif
  constexpr bool HA = FFN_HA;          // hidden accumulators in AGPRs (see mfma_acc_h)
  constexpr int KS = C / 32;            // k-steps of the first product
  constexpr int NT = C / 16;            // output-channel tiles of the second product
  constexpr int W1_STRIDE = C * 2 + 32; // bytes per hidden row in LDS
  constexpr int W2_STRIDE = 96;         // bytes per output-channel row in LDS (64 used)
  constexpr int W1_BYTES = 32 * W1_STRIDE, W2_BYTES = C * W2_STRIDE, BUF = W1_BYTES + W2_BYTES;
  constexpr int W1_CH = 32 * C / 8, W2_CH = C * 4;           // 16-byte chunks per weight chunk
  constexpr int NLD = (W1_CH + W2_CH + 255) / 256;            // staging loads per thread per chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][BUF]

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const long m0 = (long)blockIdx.x * (64 * MT) + wid * (16 * MT);

  // ---- x fragments: lane holds x[pixel m][32 ks + 8 fg .. +8] for its MT pixel tiles
  bf16x8 xf[MT][KS];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const long m = min(m0 + mt * 16 + fr, (long)p.M - 1);
    const bf16_t* xp = p.x + m * C + fg * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[mt][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xp + ks * 32));
  }
  f32x4 oacc[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) oacc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- weight-chunk staging (global -> registers -> LDS); every index below is compile-time after unrolling
  uint4 st[NLD];
#define FFN_STAGE_LOAD(HC)                                                                                   \
  {                                                                                                          \
    const bf16_t* g1 = p.w1 + (size_t)(HC) * 32 * C;  /* [32][C] */                                          \
    const bf16_t* g2 = p.w2p + (size_t)(HC) * C * 32; /* [C][32], hidden permuted */                         \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                        \
      const int c = tid + 256 * i;                                                                           \
      const bf16_t* src = c < W1_CH ? g1 + (size_t)c * 8 : g2 + (size_t)(c - W1_CH) * 8;                     \
      st[i] = (c < W1_CH + W2_CH) ? *reinterpret_cast<const uint4*>(src) : make_uint4(0, 0, 0, 0);           \
    }                                                                                                        \
  }
#define FFN_STAGE_STORE(BUFI)                                                                                \
  {                                                                                                          \
    char* base = smem + (BUFI) * BUF;                                                                        \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                        \
      const int c = tid + 256 * i;                                                                           \
      const int c2 = c - W1_CH;                                                                              \
      const int off = c < W1_CH ? (c / (C / 8)) * W1_STRIDE + (c % (C / 8)) * 16                             \
                                : W1_BYTES + (c2 >> 2) * W2_STRIDE + (c2 & 3) * 16;                          \
      if (c < W1_CH + W2_CH) *reinterpret_cast<uint4*>(base + off) = st[i];                                  \
    }                                                                                                        \
  }

  FFN_STAGE_LOAD(0)
  FFN_STAGE_STORE(0)
  __syncthreads();

  // Fragment stream of one chunk: reads 0..2KS-1 are W1 rows (hidden tile ht = i / KS, k-step i % KS), reads 2KS..NR-1
  // are W2 rows (output tile i - 2KS).  With one wave per SIMD nothing else hides LDS latency, so the reads run PD
  // steps ahead of the MFMAs that consume them (ring of PD fragments); the first-product bias rides one chunk ahead.
#ifndef FFN_PD
#define FFN_PD 6
#endif
  constexpr int NR = 2 * KS + NT, PD = FFN_PD;
#define FFN_FRAG(I, W1S, W2S)                                                                                       \
  ((I) < 2 * KS ? *reinterpret_cast<const uint4*>((W1S) + (((I) / KS) * 16 + fr) * W1_STRIDE + ((I) % KS) * 64 + fg * 16) \
                : *reinterpret_cast<const uint4*>((W2S) + (((I) - 2 * KS) * 16 + fr) * W2_STRIDE + fg * 16))
  float4 bA_n = *reinterpret_cast<const float4*>(p.b1 + fg * 4), bB_n = *reinterpret_cast<const float4*>(p.b1 + 16 + fg * 4);
  for (int hc = 0; hc < p.nchunks; ++hc) {
    const int cur = hc & 1;
    const float4 bA = bA_n, bB = bB_n;
    // the next chunk's weights: NLD global loads spread over the first product's steps, their NLD LDS stores over the
    // second product's (a burst of 12 loads costs ~800 issue cycles while the L1 path drains 64 B/clk, a burst of 12
    // ds_write_b128 ~700: in-order issue makes both dead MFMA time unless they are dealt out between the MFMAs).  The
    // last iteration re-stages the final chunk into the idle slot, which keeps the stream branch-free.
    const int hn = min(hc + 1, p.nchunks - 1);
    const bf16_t* g1n = p.w1 + (size_t)hn * 32 * C;
    const bf16_t* g2n = p.w2p + (size_t)hn * C * 32;
    char* nbase = smem + (cur ^ 1) * BUF;
    if (hc + 1 < p.nchunks) {
      bA_n = *reinterpret_cast<const float4*>(p.b1 + (hc + 1) * 32 + fg * 4);
      bB_n = *reinterpret_cast<const float4*>(p.b1 + (hc + 1) * 32 + 16 + fg * 4);
    }
    const char* w1s = smem + cur * BUF;
    const char* w2s = w1s + W1_BYTES;
    uint4 ring[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) ring[i] = FFN_FRAG(i, w1s, w2s);
    f32x4 hacc[2][MT];
    bf16x8 hf[MT];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const bf16x8 a = __builtin_bit_cast(bf16x8, ring[i % PD]);
      if (i + PD < NR) ring[i % PD] = FFN_FRAG(i + PD, w1s, w2s);
      if (i < 2 * KS) {  // H^T[ht] += W1[ht rows, k-step] . x^T
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (i % KS == 0) mfma_init_h<HA>(hacc[i / KS][mt], a, xf[mt][0]);
          else mfma_acc_h<HA>(hacc[i / KS][mt], a, xf[mt][i % KS]);
          if (i % KS == KS - 1) settle_acc<HA>(hacc[i / KS][mt]);
        }
        if (i == 2 * KS - 1) {  // bias + GELU in registers -> B operand of the second product
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            uint4 u;
            f32x2 g[4] = {{hacc[0][mt][0] + bA.x, hacc[0][mt][1] + bA.y}, {hacc[0][mt][2] + bA.z, hacc[0][mt][3] + bA.w},
                          {hacc[1][mt][0] + bB.x, hacc[1][mt][1] + bB.y}, {hacc[1][mt][2] + bB.z, hacc[1][mt][3] + bB.w}};
#ifndef FFN_ABLATE_GELU  // tools/ffn_micro.hip only: identity activation, to price the GELU
            if (MT >= 8) {  // the 8-row-tile variants have no registers left for four chains' temporaries
              f32x2 ga[2] = {g[0], g[1]}, gb[2] = {g[2], g[3]};
              gelu2_n<2>(ga);
              gelu2_n<2>(gb);
              g[0] = ga[0]; g[1] = ga[1]; g[2] = gb[0]; g[3] = gb[1];
            } else {
              gelu2_n<4>(g);
            }
#endif
            const f32x2 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
            u.x = pack_bf2(g0.x, g0.y); u.y = pack_bf2(g1.x, g1.y); u.z = pack_bf2(g2.x, g2.y); u.w = pack_bf2(g3.x, g3.y);
            hf[mt] = __builtin_bit_cast(bf16x8, u);
            settle_operand(hf[mt]);
            __builtin_amdgcn_sched_barrier(0);  // one row tile's GELU at a time: interleaving all MT of them spills
          }
        }
      } else {           // out^T[nt] += W2[nt rows, chunk] . H^T
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) mfma_acc_a(oacc[i - 2 * KS][mt], a, hf[mt]);
      }
      // NLD == KS == NT/2 for every C: load j is issued after step 2j, store j after step 2KS + 2j
      static_assert(NLD == KS && 2 * NLD == NT && NLD * 256 == W1_CH + W2_CH, "staging schedule: one whole load / store per two steps");
      if ((i & 1) == 0) {
        const int j = (i < 2 * KS ? i : i - 2 * KS) >> 1;
        const int c = tid + 256 * j, c2 = c - W1_CH;
        if (i < 2 * KS) {
          const bf16_t* src = c < W1_CH ? g1n + (size_t)c * 8 : g2n + (size_t)c2 * 8;
          st[j] = *reinterpret_cast<const uint4*>(src);
        } else {
          const int off = c < W1_CH ? (c / (C / 8)) * W1_STRIDE + (c % (C / 8)) * 16 : W1_BYTES + (c2 >> 2) * W2_STRIDE + (c2 & 3) * 16;
          *reinterpret_cast<uint4*>(nbase + off) = st[j];
        }
      }
    }
    __syncthreads();
  }
#undef FFN_FRAG

#undef FFN_STAGE_LOAD
#undef FFN_STAGE_STORE
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // last XDL writes of the output accumulators -> VALU reads below
  // ---- epilogue: lane owns pixel (m0 + mt*16 + fr), output channels nt*16 + 4*fg .. +4
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const long m = m0 + mt * 16 + fr;
    if (m >= p.M) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 16 + fg * 4;
      const float4 b2 = *reinterpret_cast<const float4*>(p.b2 + n);
      const float4 ls = *reinterpret_cast<const float4*>(p.ls + n);
      const uint2 r = *reinterpret_cast<const uint2*>(p.res + m * C + n);
      uint2 o;
      o.x = pack_bf2(bf_lo(r.x) + ls.x * (oacc[nt][mt][0] + b2.x), bf_hi(r.x) + ls.y * (oacc[nt][mt][1] + b2.y));
      o.y = pack_bf2(bf_lo(r.y) + ls.z * (oacc[nt][mt][2] + b2.z), bf_hi(r.y) + ls.w * (oacc[nt][mt][3] + b2.w));
      *reinterpret_cast<uint2*>(p.out + m * C + n) = o;
    }
  }
}

template <int C, int MT>
int launch_one(const FfnParams& p, hipStream_t s) {
  constexpr int BUF = 32 * (C * 2 + 32) + C * 96;
  static bool attr_set = false;
  if (!attr_set) {  // > 64 KB of dynamic LDS needs the opt-in once per kernel
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&convffn_kernel<C, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF));
    attr_set = true;
  }
  const long blocks = ((long)p.M + 64 * MT - 1) / (64 * MT);
  hipLaunchKernelGGL((convffn_kernel<C, MT>), dim3((unsigned)blocks), dim3(256), 2 * BUF, s, p);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

}  // namespace

bool convffn_supported(int C, int ratio) { return ratio == 4 && (C == 32 || C == 64 || C == 96 || C == 128 || C == 192 || C == 384); }

// w2 [C][hidden] row-major -> [hidden/32][C][32] with slot (g, j) of each 32-block holding hidden 16*(j>>2) + 4*g + (j&3)
void convffn_pack_w2(const float* w2, float* out, int C, int hidden) {
  for (int hc = 0; hc < hidden / 32; ++hc)
    for (int n = 0; n < C; ++n)
      for (int g = 0; g < 4; ++g)
        for (int j = 0; j < 8; ++j)
          out[((size_t)hc * C + n) * 32 + 8 * g + j] = w2[(size_t)n * hidden + hc * 32 + 16 * (j >> 2) + 4 * g + (j & 3)];
}

int launch_convffn(const bf16_t* x, const bf16_t* w1, const float* b1, const bf16_t* w2p, const float* b2, const float* ls,
                   const bf16_t* res, bf16_t* out, int M, int C, int hidden, hipStream_t s) {
  if (!x || !w1 || !b1 || !w2p || !b2 || !ls || !res || !out) return fv_fail(FV_ERR_ARG, "convffn: null pointer");
  if (M <= 0 || hidden != 4 * C || !convffn_supported(C, 4)) return fv_fail(FV_ERR_UNSUPPORTED, "convffn: unsupported C=%d hidden=%d", C, hidden);
  if (((uintptr_t)x | (uintptr_t)w1 | (uintptr_t)w2p | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)ls) & 15 || ((uintptr_t)res | (uintptr_t)out) & 7)
    return fv_fail(FV_ERR_ARG, "convffn: misaligned pointer");
  if (x == out) return fv_fail(FV_ERR_ARG, "convffn: x must not alias out");
  FfnParams p{x, w1, b1, w2p, b2, ls, res, out, M, hidden / 32};
  switch (C) {
    case 32: return launch_one<32, 8>(p, s);
    case 64: return launch_one<64, 8>(p, s);
#ifndef FFN_MT96
#define FFN_MT96 4
#endif
    case 96: return launch_one<96, FFN_MT96>(p, s);
    case 128: return launch_one<128, 4>(p, s);
    case 192: return launch_one<192, 4>(p, s);
    case 384: return launch_one<384, 2>(p, s);
  }
  return fv_fail(FV_ERR_UNSUPPORTED, "convffn: unsupported C=%d", C);
}

}  // namespace fv
