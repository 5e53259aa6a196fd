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
#ifdef FFN_STAMPS  // tools/ffn_micro.hip diagnostic build: per-block {shader cycles, 100 MHz ticks} around the chunk loop
  unsigned long long* stamps;
#endif
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

// largest ring depth <= want that divides the chunk's step count (the ring index must line up across chunks)
constexpr int ring_depth(int nr, int want) { return nr % want == 0 ? want : ring_depth(nr, want - 1); }

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
  FFN_STAGE_LOAD(min(1, p.nchunks - 1))   // rides in registers through chunk 0, stored to LDS during it
  __syncthreads();

  // Fragment stream of one chunk: reads 0..2KS-1 are W1 rows (hidden tile ht, k-step kk), reads 2KS..NR-1 are W2 rows
  // (output tile i - 2KS).  With one wave per SIMD nothing else hides LDS latency, so the reads run PD steps ahead of
  // the MFMAs that consume them (ring of PD fragments) -- across the chunk boundary too: the chunk's one barrier sits
  // PD steps before its end, where the last read of this chunk's slot has been issued and every store of the next
  // chunk's weights is long done, so the ring refills from the next slot behind it and no chunk starts on a cold ring
  // behind a barrier.  The first-product bias rides one chunk ahead.
#ifndef FFN_PD
#define FFN_PD 6
#endif
  constexpr int NR = 2 * KS + NT, PD = ring_depth(NR, FFN_PD < NR / 2 ? FFN_PD : NR / 2);
  static_assert(NLD == KS && 2 * NLD == NT && NLD * 256 == W1_CH + W2_CH && 2 * NLD <= NR - PD, "staging schedule");
  static_assert(NR % PD == 0, "fragment f lives in ring[f % PD] across the chunk boundary");
#ifndef FFN_ORDER_SEQ  // first-product step i -> hidden tile / k-step.  Interleaving the two hidden tiles doubles the distance
#define FFN_HT(I) ((I) & 1)   // between MFMAs on the same accumulator (MT = 2: from 2 to 4 MFMAs)
#define FFN_KK(I) ((I) >> 1)
#else
#define FFN_HT(I) ((I) / KS)
#define FFN_KK(I) ((I) % KS)
#endif
#define FFN_FRAG(I, W1S, W2S)                                                                                       \
  ((I) < 2 * KS ? *reinterpret_cast<const uint4*>((W1S) + (FFN_HT(I) * 16 + fr) * W1_STRIDE + FFN_KK(I) * 64 + fg * 16) \
                : *reinterpret_cast<const uint4*>((W2S) + (((I) - 2 * KS) * 16 + fr) * W2_STRIDE + fg * 16))
  float4 bA_n = *reinterpret_cast<const float4*>(p.b1 + fg * 4), bB_n = *reinterpret_cast<const float4*>(p.b1 + 16 + fg * 4);
  uint4 ring[PD];
#pragma unroll
  for (int i = 0; i < PD; ++i) ring[i] = FFN_FRAG(i, smem, smem + W1_BYTES);
#ifdef FFN_STAMPS
  const unsigned long long t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
#ifdef FFN_STAMPS_FINE  // per-phase cycle sums (perturbs the ring: every stamp drains lgkmcnt)
  unsigned long long tl = t0c, ta[6] = {0, 0, 0, 0, 0, 0};
#define FFN_STAMP(K) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ta[K] += t_ - tl; tl = t_; }
#else
#define FFN_STAMP(K)
#endif
#else
#define FFN_STAMP(K)
#endif
  for (int hc = 0; hc < p.nchunks; ++hc) {
    const int cur = hc & 1;
    const float4 bA = bA_n, bB = bB_n;
    // Weight staging, two chunks deep so no store ever waits on its load: during chunk hc the registers loaded during
    // chunk hc-1 (chunk hc+1's weights) go to the idle LDS slot and are refilled with chunk hc+2's.  One store or one
    // load per step of the first product: a burst of 12 loads costs ~800 issue cycles while the L1 path drains
    // 64 B/clk, a burst of 12 ds_write_b128 ~700, and in-order issue makes both dead MFMA time.  Past the last chunk
    // the final one is staged again into the idle slot, which keeps the stream branch-free.
    const int hn = min(hc + 2, p.nchunks - 1);
    const bf16_t* g1n = p.w1 + (size_t)hn * 32 * C;
    const bf16_t* g2n = p.w2p + (size_t)hn * C * 32;
    char* nbase = smem + (cur ^ 1) * BUF;
    if (hc + 1 < p.nchunks) {
      bA_n = *reinterpret_cast<const float4*>(p.b1 + (hc + 1) * 32 + fg * 4);
      bB_n = *reinterpret_cast<const float4*>(p.b1 + (hc + 1) * 32 + 16 + fg * 4);
    }
    const char* w1s = smem + cur * BUF;
    const char* w2s = w1s + W1_BYTES;
    f32x4 hacc[2][MT];
    bf16x8 hf[MT];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      if (i == NR - PD) { FFN_STAMP(3) __syncthreads(); FFN_STAMP(4) }
      const bf16x8 a = __builtin_bit_cast(bf16x8, ring[i % PD]);
      ring[i % PD] = i + PD < NR ? FFN_FRAG(i + PD, w1s, w2s) : FFN_FRAG(i + PD - NR, nbase, nbase + W1_BYTES);
      if (i < 2 * KS) {  // H^T[ht] += W1[ht rows, k-step] . x^T
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (FFN_KK(i) == 0) mfma_init_h<HA>(hacc[FFN_HT(i)][mt], a, xf[mt][0]);
          else mfma_acc_h<HA>(hacc[FFN_HT(i)][mt], a, xf[mt][FFN_KK(i)]);
          if (FFN_KK(i) == KS - 1) settle_acc<HA>(hacc[FFN_HT(i)][mt]);
        }
        {  // staging: even steps store register j to the idle slot, odd steps reload it for the chunk after
          const int j = i >> 1;
          const int c = tid + 256 * j, c2 = c - W1_CH;
          if ((i & 1) == 0) {
            const int off = c < W1_CH ? (c / (C / 8)) * W1_STRIDE + (c % (C / 8)) * 16 : W1_BYTES + (c2 >> 2) * W2_STRIDE + (c2 & 3) * 16;
            *reinterpret_cast<uint4*>(nbase + off) = st[j];
          } else {
            const bf16_t* src = c < W1_CH ? g1n + (size_t)c * 8 : g2n + (size_t)c2 * 8;
            st[j] = *reinterpret_cast<const uint4*>(src);
          }
        }
        if (i == 2 * KS - 1) {  // bias + GELU in registers -> B operand of the second product
          FFN_STAMP(0)
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
          FFN_STAMP(1)
        }
      } else {           // out^T[nt] += W2[nt rows, chunk] . H^T
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) mfma_acc_a(oacc[i - 2 * KS][mt], a, hf[mt]);
      }
    }
    FFN_STAMP(5)
  }
#ifdef FFN_STAMPS
  if (tid == 0 && p.stamps) {
    p.stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0c;
    p.stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
#ifdef FFN_STAMPS_FINE
    if (blockIdx.x == 0) for (int k = 0; k < 6; ++k) p.stamps[2 * gridDim.x + k] = ta[k];
#endif
  }
#endif
#undef FFN_FRAG

#undef FFN_STAGE_LOAD
#undef FFN_STAGE_STORE
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // last XDL writes of the output accumulators -> VALU reads below
  // ---- epilogue.  The accumulator layout (lane = pixel fr, 4 channels per tile) would leave as 8-byte accesses 32 B apart
  // -- 2 NT of them per row tile, store-issue-bound with nothing to overlap them at one wave per SIMD.  Instead each
  // wave turns its 16-row tiles through its own corner of the (now idle) weight slots as fp32: layer scale and bias are
  // applied on the way in (channels fixed per lane), residual add and bf16 rounding on the way out, where a lane owns
  // 8 consecutive channels and every global access is 16 B of a fully used line.
  constexpr int ORB = C * 4 + 16;                      // fp32 row + 16 B (conflict-free 16-B column writes)
  static_assert(4 * 16 * ORB <= 2 * BUF, "epilogue tile must fit in the weight slots");
  __syncthreads();                                     // every wave is done reading weights
  char* so = smem + wid * (16 * ORB);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    uint4 rr[KS];                                      // residual rows first: their latency hides behind the LDS turn
#pragma unroll
    for (int it = 0; it < KS; ++it) {
      const int ch = it * 64 + lane, row = ch / (C / 8), c8 = ch % (C / 8);
      const long m = min(m0 + mt * 16 + row, (long)p.M - 1);
      rr[it] = *reinterpret_cast<const uint4*>(p.res + m * C + c8 * 8);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nt * 16 + fg * 4;
      const float4 b2 = *reinterpret_cast<const float4*>(p.b2 + n);
      const float4 ls = *reinterpret_cast<const float4*>(p.ls + n);
      float4 v;
      v.x = ls.x * (oacc[nt][mt][0] + b2.x); v.y = ls.y * (oacc[nt][mt][1] + b2.y);
      v.z = ls.z * (oacc[nt][mt][2] + b2.z); v.w = ls.w * (oacc[nt][mt][3] + b2.w);
      *reinterpret_cast<float4*>(so + fr * ORB + n * 4) = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-local hand-over: LDS serves a wave's accesses in order
#pragma unroll
    for (int it = 0; it < KS; ++it) {
      const int ch = it * 64 + lane, row = ch / (C / 8), c8 = ch % (C / 8);
      const long m = m0 + mt * 16 + row;
      const float4 y0 = *reinterpret_cast<const float4*>(so + row * ORB + c8 * 32);
      const float4 y1 = *reinterpret_cast<const float4*>(so + row * ORB + c8 * 32 + 16);
      uint4 o;
      o.x = pack_bf2(bf_lo(rr[it].x) + y0.x, bf_hi(rr[it].x) + y0.y);
      o.y = pack_bf2(bf_lo(rr[it].y) + y0.z, bf_hi(rr[it].y) + y0.w);
      o.z = pack_bf2(bf_lo(rr[it].z) + y1.x, bf_hi(rr[it].z) + y1.y);
      o.w = pack_bf2(bf_lo(rr[it].w) + y1.z, bf_hi(rr[it].w) + y1.w);
      if (m < p.M) *reinterpret_cast<uint4*>(p.out + m * C + c8 * 8) = o;
    }
    asm volatile("" ::: "memory");                      // the next tile's writes stay behind these reads
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

#ifdef FFN_STAMPS
unsigned long long* g_ffn_stamps = nullptr;
#endif
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
#ifdef FFN_STAMPS
  p.stamps = g_ffn_stamps;
#endif
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
