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
#ifdef FFN_STAMPS  // tools/ffn_micro.hip diagnostic build: per block {prologue, first tile's chunk loop, its epilogue} cycles + clock
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
// BA: the B operand (an x fragment) lives in the accumulator half of the register file too.  At C >= 192 the x fragments
// are 96 registers; keeping half of them in AGPRs (MFMA reads A / B from either half) is what leaves room for the
// staging registers and the GELU stages without spilling -- the 192 output accumulators leave 64 AGPRs unused.
template <bool HA, bool BA = false>
__device__ __forceinline__ void mfma_acc_h(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  if constexpr (HA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else if constexpr (BA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
// first k-step of a hidden tile with the first-product bias as the C operand (D layout: row 4 fg + r = hidden unit, so a
// lane's four accumulator registers take its four biases): the GELU then needs no bias add -- VALU issue slots are what
// the chunk is short of.  The bias registers come from a global load, not from the VALU: no hazard into the MFMA.
__device__ __forceinline__ void mfma_init_hb(f32x4& acc, const bf16x8& a, const bf16x8& b, const f32x4& bias) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=v"(acc) : "v"(a), "v"(b), "v"(bias));
}
// first k-step of a hidden tile: C = 0 as an inline constant, so no VALU-written zero feeds the MFMA
template <bool HA>
__device__ __forceinline__ void mfma_init_h(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  if constexpr (HA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "v"(b));
}
// Hazards hipcc cannot see through the asm MFMAs, each paid ONCE per chunk (an s_nop wait state is 4 clocks, so a settle per
// tile cost 450 clocks a chunk at MT = 4).  settle_accs sits behind the first product's LAST MFMA and ties every hidden
// accumulator to it: the GELU's reads are then ordered after all those MFMAs, and the XDL-write -> VALU-read distance
// (7 wait states for this 4-pass MFMA) is owed only by the last-issued tile; 10 wait states cover it with a margin.
// settle_operands does the same for the VALU-written B operands (packed GELU output) ahead of the second product.
template <bool HA, int N>
__device__ __forceinline__ void settle_accs(f32x4 (&h)[2][N]) {
  static_assert(N == 2 || N == 4 || N == 8, "row tiles per wave");
  if constexpr (N == 2) {
    if constexpr (HA) asm volatile("s_nop 7\n\ts_nop 1" : "+a"(h[0][0]), "+a"(h[0][1]), "+a"(h[1][0]), "+a"(h[1][1]));
    else asm volatile("s_nop 7\n\ts_nop 1" : "+v"(h[0][0]), "+v"(h[0][1]), "+v"(h[1][0]), "+v"(h[1][1]));
  } else if constexpr (N == 4) {
    if constexpr (HA) asm volatile("s_nop 7\n\ts_nop 1" : "+a"(h[0][0]), "+a"(h[0][1]), "+a"(h[0][2]), "+a"(h[0][3]), "+a"(h[1][0]), "+a"(h[1][1]), "+a"(h[1][2]), "+a"(h[1][3]));
    else asm volatile("s_nop 7\n\ts_nop 1" : "+v"(h[0][0]), "+v"(h[0][1]), "+v"(h[0][2]), "+v"(h[0][3]), "+v"(h[1][0]), "+v"(h[1][1]), "+v"(h[1][2]), "+v"(h[1][3]));
  } else {
    if constexpr (HA) asm volatile("s_nop 7\n\ts_nop 1" : "+a"(h[0][0]), "+a"(h[0][1]), "+a"(h[0][2]), "+a"(h[0][3]), "+a"(h[0][4]), "+a"(h[0][5]), "+a"(h[0][6]), "+a"(h[0][7]),
                                               "+a"(h[1][0]), "+a"(h[1][1]), "+a"(h[1][2]), "+a"(h[1][3]), "+a"(h[1][4]), "+a"(h[1][5]), "+a"(h[1][6]), "+a"(h[1][7]));
    else asm volatile("s_nop 7\n\ts_nop 1" : "+v"(h[0][0]), "+v"(h[0][1]), "+v"(h[0][2]), "+v"(h[0][3]), "+v"(h[0][4]), "+v"(h[0][5]), "+v"(h[0][6]), "+v"(h[0][7]),
                                  "+v"(h[1][0]), "+v"(h[1][1]), "+v"(h[1][2]), "+v"(h[1][3]), "+v"(h[1][4]), "+v"(h[1][5]), "+v"(h[1][6]), "+v"(h[1][7]));
  }
}
template <int N>
__device__ __forceinline__ void settle_operands(bf16x8 (&v)[N]) {
  if constexpr (N == 2) asm volatile("s_nop 3" : "+v"(v[0]), "+v"(v[1]));
  else if constexpr (N == 4) asm volatile("s_nop 3" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
  else asm volatile("s_nop 3" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
}

// Staging registers are twelve named locals picked by select chains on the (unrolled) step counter: an array or a struct
// indexed by it was left in scratch memory whenever the full unroll came after the last SROA run.
#define FFN_ST_GET(J)                                                                                              \
  ((J) == 0 ? st0 : (J) == 1 ? st1 : (J) == 2 ? st2 : (J) == 3 ? st3 : (J) == 4 ? st4 : (J) == 5 ? st5 : (J) == 6 ? st6 \
   : (J) == 7 ? st7 : (J) == 8 ? st8 : (J) == 9 ? st9 : (J) == 10 ? st10 : st11)
#define FFN_ST_SET(J, V)                                                                                           \
  {                                                                                                                \
    const uint4 v_ = (V);                                                                                          \
    if ((J) == 0) st0 = v_; else if ((J) == 1) st1 = v_; else if ((J) == 2) st2 = v_; else if ((J) == 3) st3 = v_;  \
    else if ((J) == 4) st4 = v_; else if ((J) == 5) st5 = v_; else if ((J) == 6) st6 = v_; else if ((J) == 7) st7 = v_; \
    else if ((J) == 8) st8 = v_; else if ((J) == 9) st9 = v_; else if ((J) == 10) st10 = v_; else st11 = v_;        \
  }

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
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][BUF] weight slots, then b2[C], ls[C] fp32

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int nch = p.nchunks;                                   // C / 8: even, >= 4
  const uint32_t tid16 = (uint32_t)tid * 16u;                  // the lane's byte offset inside a 4 KB group of staging loads
  const int ntiles = (p.M + 64 * MT - 1) / (64 * MT);

  // ---- x fragments: lane holds x[pixel m][32 ks + 8 fg .. +8] for its MT pixel tiles (rows past M clamp to M - 1)
  bf16x8 xf[MT][KS];
#define FFN_LOAD_X(TILE)                                                                                     \
  {                                                                                                          \
    int lx_ = lane;                                                                                          \
    asm volatile("" : "+v"(lx_)); /* keeps this address math out of the chunk loop's live registers */       \
    const long mb_ = (long)(TILE) * (64 * MT) + wid * (16 * MT);                                             \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                      \
      const long m_ = min(mb_ + mt * 16 + (lx_ & 15), (long)p.M - 1);                                        \
      const bf16_t* xp_ = p.x + m_ * C + (lx_ >> 4) * 8;                                                     \
      _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                                      \
        xf[mt][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(xp_ + ks * 32));             \
    }                                                                                                        \
  }
#ifdef FFN_STAMPS
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
  FFN_LOAD_X(blockIdx.x)
  f32x4 oacc[NT][MT];

  // ---- weight-chunk staging (global -> registers -> LDS); every index below is compile-time after unrolling
  static_assert(NLD <= 12, "twelve staging registers");
  uint4 st0, st1, st2, st3, st4, st5, st6, st7, st8, st9, st10, st11;
  st0 = st1 = st2 = st3 = st4 = st5 = st6 = st7 = st8 = st9 = st10 = st11 = make_uint4(0, 0, 0, 0);
#define FFN_STAGE_LOAD(HC)                                                                                   \
  {                                                                                                          \
    const bf16_t* g1 = p.w1 + (size_t)(HC) * 32 * C;  /* [32][C] */                                          \
    const bf16_t* g2 = p.w2p + (size_t)(HC) * C * 32; /* [C][32], hidden permuted */                         \
    _Pragma("unroll") for (int i = 0; i < NLD; ++i) {                                                        \
      const int c = tid + 256 * i;                                                                           \
      const bf16_t* src = c < W1_CH ? g1 + (size_t)c * 8 : g2 + (size_t)(c - W1_CH) * 8;                     \
      FFN_ST_SET(i, *reinterpret_cast<const uint4*>(src))                                                    \
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
      *reinterpret_cast<uint4*>(base + off) = FFN_ST_GET(i);                                                 \
    }                                                                                                        \
  }

  FFN_STAGE_LOAD(0)
  FFN_STAGE_STORE(0)
  FFN_STAGE_LOAD(1)                       // rides in registers through chunk 0, stored to LDS during it
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
  constexpr int NB = NT;                   // second-product steps per chunk
  constexpr bool XA = !HA && C >= 192;     // upper half of the x fragments in AGPRs (see mfma_acc_h)
  constexpr int NR = 2 * KS + NB, PD = ring_depth(NR, FFN_PD < NR / 2 ? FFN_PD : NR / 2);
  static_assert(NLD == KS && 2 * NLD == NT && NLD * 256 == W1_CH + W2_CH && 2 * NLD <= NR - PD, "staging schedule");
  static_assert(NR % PD == 0, "fragment f lives in ring[f % PD] across the chunk boundary");
  // first-product step i -> hidden tile / k-step: interleaving the two hidden tiles doubles the distance between MFMAs on
  // the same accumulator (MT = 2: from 2 to 4 MFMAs)
#define FFN_HT(I) ((I) & 1)
#define FFN_KK(I) ((I) >> 1)
#define FFN_FRAG(I, W1S, W2S)                                                                                       \
  ((I) < 2 * KS ? *reinterpret_cast<const uint4*>((W1S) + (FFN_HT(I) * 16 + fr) * W1_STRIDE + FFN_KK(I) * 64 + fg * 16) \
                : *reinterpret_cast<const uint4*>((W2S) + (((I) - 2 * KS) * 16 + fr) * W2_STRIDE + fg * 16))
  float4 bA_n = *reinterpret_cast<const float4*>(p.b1 + fg * 4), bB_n = *reinterpret_cast<const float4*>(p.b1 + 16 + fg * 4);
  uint4 ring[PD];
#pragma unroll
  for (int i = 0; i < PD; ++i) ring[i] = FFN_FRAG(i, smem, smem + W1_BYTES);

  // epilogue geometry: one pass turns 16 rows x C/2 channels of fp32 through the wave's quarter of a weight slot
  constexpr int ORB = C * 2 + 16;                      // fp32 half row + 16 B (conflict-free 16-B column writes)
  constexpr int RP = (C + 63) / 64;                    // 8-channel chunks per lane per pass (16 rows x C/16 chunks = C)
  static_assert(16 * ORB <= BUF / 4 && (BUF / 4) % 16 == 0 && NT % 2 == 0, "epilogue pass must fit in a quarter of a weight slot");
  float* sb2 = reinterpret_cast<float*>(smem + 2 * BUF);   // second-product bias and layer scale, staged once per block
  float* sls = sb2 + C;
  for (int i = tid; i < C / 4; i += 256) {
    reinterpret_cast<float4*>(sb2)[i] = reinterpret_cast<const float4*>(p.b2)[i];
    reinterpret_cast<float4*>(sls)[i] = reinterpret_cast<const float4*>(p.ls)[i];
  }
  // residual rows, fetched RD - 1 passes ahead (pass q in rr[q % RD]): the epilogue is bound by how many bytes a wave keeps in
  // flight towards HBM (one pass = 1 KB per load), not by its instructions, so the small-C variants that have the
  // registers look further ahead
  constexpr int RD = RP <= 3 ? 5 : 3;
  uint4 rr[RD][RP];
#define FFN_LOAD_RES(Q, MB)                                                                                  \
  {                                                                                                          \
    int lr_ = lane;                                                                                          \
    asm volatile("" : "+v"(lr_));                                                                            \
    _Pragma("unroll") for (int it = 0; it < RP; ++it) {                                                      \
      const int ch_ = min(it * 64 + lr_, C - 1), row_ = ch_ / (C / 16), c8_ = ch_ % (C / 16);                \
      const long m_ = min((MB) + ((Q) >> 1) * 16 + row_, (long)p.M - 1);                                     \
      rr[(Q) % RD][it] = *reinterpret_cast<const uint4*>(p.res + m_ * C + ((Q) & 1) * (C / 2) + c8_ * 8);     \
    }                                                                                                        \
  }

  // One chunk of 32 hidden units.  (One instantiation on purpose: a peeled copy of the last chunk gets its own accumulator
  // registers, and the copies hipcc then inserts behind the opaque asm MFMAs read stale data.)
#define FFN_CHUNK                                                                                                   \
  {                                                                                                                 \
    const int cur = hc & 1;                                                                                         \
    const float4 bA = bA_n, bB = bB_n;                                                                              \
    const f32x4 bAv = f32x4{bA.x, bA.y, bA.z, bA.w}, bBv = f32x4{bB.x, bB.y, bB.z, bB.w};                           \
    /* Weight staging, two chunks deep so no store ever waits on its load: during chunk hc the registers loaded   */ \
    /* during chunk hc-1 (chunk hc+1's weights) go to the idle LDS slot and are refilled with chunk hc+2's; the   */ \
    /* chunk index wraps, so the stream runs on into the next tile.  One store or one load per step of the first  */ \
    /* product: a burst of 12 loads costs ~800 issue cycles while the L1 path drains 64 B/clk, a burst of 12      */ \
    /* ds_write_b128 ~700, and in-order issue makes both dead MFMA time.                                          */ \
    const int hn = hc + 2 - (hc + 2 >= nch ? nch : 0), hb = hc + 1 - (hc + 1 >= nch ? nch : 0);                     \
    const bf16_t* g1n = p.w1 + (size_t)hn * 32 * C;                                                                 \
    const bf16_t* g2n = p.w2p + (size_t)hn * C * 32;                                                                \
    char* nbase = smem + (cur ^ 1) * BUF;                                                                           \
    bA_n = *reinterpret_cast<const float4*>(p.b1 + hb * 32 + fg * 4);                                               \
    bB_n = *reinterpret_cast<const float4*>(p.b1 + hb * 32 + 16 + fg * 4);                                          \
    const char* w1s = smem + cur * BUF;                                                                             \
    const char* w2s = w1s + W1_BYTES;                                                                               \
    f32x4 hacc[2][MT];                                                                                              \
    bf16x8 hf[MT];                                                                                                  \
    _Pragma("unroll") for (int i = 0; i < NR; ++i) {                                                                \
      if (i == NR - PD) __syncthreads();                                                                            \
      const bf16x8 a = __builtin_bit_cast(bf16x8, ring[i % PD]);                                                    \
      ring[i % PD] = i + PD < NR ? FFN_FRAG(i + PD, w1s, w2s) : FFN_FRAG(i + PD - NR, nbase, nbase + W1_BYTES);     \
      if (i < 2 * KS) { /* H^T[ht] += W1[ht rows, k-step] . x^T */                                                  \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                         \
          if (FFN_KK(i) == 0 && !HA) mfma_init_hb(hacc[FFN_HT(i)][mt], a, xf[mt][0], FFN_HT(i) ? bBv : bAv);        \
          else if (FFN_KK(i) == 0) mfma_init_h<HA>(hacc[FFN_HT(i)][mt], a, xf[mt][0]);                              \
          else if (XA && FFN_KK(i) >= KS / 2) mfma_acc_h<HA, XA>(hacc[FFN_HT(i)][mt], a, xf[mt][FFN_KK(i) >= KS / 2 ? FFN_KK(i) : KS - 1]); \
          else mfma_acc_h<HA>(hacc[FFN_HT(i)][mt], a, xf[mt][FFN_KK(i)]);                                           \
        }                                                                                                           \
        { /* staging: even steps store register j to the idle slot, odd steps reload it for the chunk after */      \
          const int j = i >> 1;                                                                                     \
          const int c = tid + 256 * j, c2 = c - W1_CH;                                                              \
          if ((i & 1) == 0) {                                                                                       \
            const int off = c < W1_CH ? (c / (C / 8)) * W1_STRIDE + (c % (C / 8)) * 16                              \
                                      : W1_BYTES + (c2 >> 2) * W2_STRIDE + (c2 & 3) * 16;                           \
            *reinterpret_cast<uint4*>(nbase + off) = FFN_ST_GET(j);                                                 \
          } else { /* uniform base (W1 or W2 part is decided by j alone when 256 divides W1_CH) + the lane's 16 tid */ \
            if constexpr (W1_CH % 256 == 0) {                                                                       \
              const char* sb_ = j * 256 < W1_CH ? reinterpret_cast<const char*>(g1n) + j * 4096                     \
                                                : reinterpret_cast<const char*>(g2n) + (j * 256 - W1_CH) * 16;      \
              FFN_ST_SET(j, *reinterpret_cast<const uint4*>(sb_ + tid16))                                           \
            } else {                                                                                                \
              const bf16_t* src = c < W1_CH ? g1n + (size_t)c * 8 : g2n + (size_t)c2 * 8;                           \
              FFN_ST_SET(j, *reinterpret_cast<const uint4*>(src))                                                   \
            }                                                                                                       \
          }                                                                                                         \
        }                                                                                                           \
        if (i == 2 * KS - 1) { /* bias + GELU in registers -> B operand of the second product */                    \
          settle_accs<HA, MT>(hacc);                                                                                \
          _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                       \
            f32x2 g[4] = {{hacc[0][mt][0], hacc[0][mt][1]}, {hacc[0][mt][2], hacc[0][mt][3]},                       \
                          {hacc[1][mt][0], hacc[1][mt][1]}, {hacc[1][mt][2], hacc[1][mt][3]}};                      \
            if constexpr (HA) { /* AGPR accumulators start from the inline 0: the bias is added here */             \
              g[0] += f32x2{bA.x, bA.y}; g[1] += f32x2{bA.z, bA.w}; g[2] += f32x2{bB.x, bB.y}; g[3] += f32x2{bB.z, bB.w}; \
            }                                                                                                       \
            FFN_GELU(g)                                                                                             \
            uint4 u;                                                                                                \
            u.x = pack_bf2(g[0].x, g[0].y); u.y = pack_bf2(g[1].x, g[1].y);                                         \
            u.z = pack_bf2(g[2].x, g[2].y); u.w = pack_bf2(g[3].x, g[3].y);                                         \
            hf[mt] = __builtin_bit_cast(bf16x8, u);                                                                 \
            __builtin_amdgcn_sched_barrier(0); /* one row tile's GELU at a time: interleaving all MT spills */      \
          }                                                                                                         \
          settle_operands<MT>(hf);                                                                                  \
        }                                                                                                           \
      } else { /* out^T[nt] += W2[nt rows, chunk] . H^T */                                                          \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) mfma_acc_a(oacc[i - 2 * KS][mt], a, hf[mt]);              \
      }                                                                                                             \
    }                                                                                                               \
  }
#ifdef FFN_ABLATE_GELU  // tools/ffn_micro.hip only: identity activation, to price the GELU
#define FFN_GELU(G)
#else
#define FFN_GELU(G)                                                                                            \
  if (MT >= 8) { /* the 8-row-tile variants have no registers left for four chains' temporaries */             \
    f32x2 ga[2] = {G[0], G[1]}, gb[2] = {G[2], G[3]};                                                          \
    gelu2_n<2>(ga);                                                                                            \
    gelu2_n<2>(gb);                                                                                            \
    G[0] = ga[0]; G[1] = ga[1]; G[2] = gb[0]; G[3] = gb[1];                                                    \
  } else {                                                                                                     \
    gelu2_n<4>(G);                                                                                             \
  }
#endif

  // ---- persistent loop over this block's row tiles.  With 512 registers per lane there is one wave per SIMD and nothing
  // to overlap a tile's input loads, residual loads and output stores with -- except the neighbouring tile's MFMAs.
  for (int tile = blockIdx.x; tile < ntiles; tile += (int)gridDim.x) {
    const long mb = (long)tile * (64 * MT) + wid * (16 * MT);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) oacc[nt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef FFN_STAMPS
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime(), t_real = __builtin_amdgcn_s_memrealtime();
#endif
    for (int hc = 0; hc < nch; ++hc) FFN_CHUNK
#ifdef FFN_STAMPS
    const unsigned long long t_epi = __builtin_amdgcn_s_memtime(), t_real2 = __builtin_amdgcn_s_memrealtime();
#endif
    __builtin_amdgcn_sched_barrier(0);  // keep the epilogue's loads out of the chunk: hoisted, they spill its registers
#pragma unroll
    for (int q0 = 0; q0 < RD - 1 && q0 < 2 * MT; ++q0) FFN_LOAD_RES(q0, mb)
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");  // last XDL writes of the output accumulators -> VALU reads below
    __builtin_amdgcn_sched_barrier(0);  // ... and nothing that reads them moves above the wait states
    // ---- epilogue.  The accumulator layout (lane = pixel fr, 4 channels per tile) would leave as 8-byte accesses 32 B
    // apart -- 2 NT of them per row tile, store-issue-bound.  Instead each wave turns 16 rows x C/2 channels at a time
    // through its quarter of the slot the last chunk just released (the other slot already holds the next tile's first
    // chunk) as raw fp32; on the way out a lane owns 8 consecutive channels of a row: bias, layer scale (both staged in
    // LDS), residual add and the one bf16 rounding happen there, and every global access is 16 B of a fully used line.
    // Nobody reads that slot after the last chunk's barrier, so no block-wide sync is needed to start.
    char* so = smem + ((nch - 1) & 1) * BUF + wid * (BUF / 4);
    int le = lane;
    asm volatile("" : "+v"(le));                         // epilogue address math stays out of the chunk loop's registers
    const int fre = le & 15, fge = le >> 4;
#pragma unroll
    for (int q = 0; q < 2 * MT; ++q) {                   // pass q: row tile q / 2, channel half q % 2
      const int mt = q >> 1, half = q & 1;
      // the next tile's x fragments fly during the last pass only (rows past M clamp): earlier they would pin C / 4
      // registers through the whole epilogue, and every residual wait would wait for them too (vmcnt is in order)
      if (q == 2 * MT - 1) FFN_LOAD_X(tile + (int)gridDim.x)
#pragma unroll
      for (int nh = 0; nh < NT / 2; ++nh)
        *reinterpret_cast<f32x4*>(so + fre * ORB + (nh * 16 + fge * 4) * 4) = oacc[half * (NT / 2) + nh][mt];
      asm volatile("" ::: "memory");  // wave-local hand-over: LDS serves a wave's accesses in order, no wait needed
#pragma unroll
      for (int it = 0; it < RP; ++it) {
        const int ch = it * 64 + le, chc = min(ch, C - 1), row = chc / (C / 16), c8 = chc % (C / 16);
        const int cb = half * (C / 2) + c8 * 8;
        const float4 y0 = *reinterpret_cast<const float4*>(so + row * ORB + c8 * 32);
        const float4 y1 = *reinterpret_cast<const float4*>(so + row * ORB + c8 * 32 + 16);
        const float4 b0 = *reinterpret_cast<const float4*>(sb2 + cb), b1 = *reinterpret_cast<const float4*>(sb2 + cb + 4);
        const float4 l0 = *reinterpret_cast<const float4*>(sls + cb), l1 = *reinterpret_cast<const float4*>(sls + cb + 4);
        const uint4 r4 = rr[q % RD][it];
        uint4 o;
        o.x = pack_bf2(bf_lo(r4.x) + l0.x * (y0.x + b0.x), bf_hi(r4.x) + l0.y * (y0.y + b0.y));
        o.y = pack_bf2(bf_lo(r4.y) + l0.z * (y0.z + b0.z), bf_hi(r4.y) + l0.w * (y0.w + b0.w));
        o.z = pack_bf2(bf_lo(r4.z) + l1.x * (y1.x + b1.x), bf_hi(r4.z) + l1.y * (y1.y + b1.y));
        o.w = pack_bf2(bf_lo(r4.w) + l1.z * (y1.z + b1.z), bf_hi(r4.w) + l1.w * (y1.w + b1.w));
        const long m = mb + mt * 16 + row;
        if (ch < C && m < p.M) *reinterpret_cast<uint4*>(p.out + m * C + half * (C / 2) + c8 * 8) = o;
        if (RP > 3 && (it & 1)) __builtin_amdgcn_sched_barrier(0);   // C = 384: two chunks' operands in flight at a time, not all six
      }
      if (q + RD - 1 < 2 * MT) FFN_LOAD_RES(q + RD - 1, mb)   // residual rows RD - 1 passes ahead: one pass is far shorter than an HBM read
      asm volatile("" ::: "memory");                      // the next pass's writes stay behind these reads
    }
    __syncthreads();  // the slot is the next tile's staging target again
#ifdef FFN_STAMPS
    if (tid == 0 && p.stamps && tile == (int)blockIdx.x) {
      unsigned long long* d = p.stamps + 4 * blockIdx.x;
      d[0] = t_loop - t_entry; d[1] = t_epi - t_loop; d[2] = __builtin_amdgcn_s_memtime() - t_epi; d[3] = t_real2 - t_real;
    }
#endif
  }
#undef FFN_CHUNK
#undef FFN_GELU
#undef FFN_LOAD_RES
#undef FFN_LOAD_X
#undef FFN_FRAG
#undef FFN_HT
#undef FFN_KK
#undef FFN_STAGE_LOAD
#undef FFN_STAGE_STORE
}

int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) return 256;
    n = prop.multiProcessorCount;
  }
  return n;
}

template <int C, int MT>
int launch_one(const FfnParams& p, hipStream_t s) {
  constexpr int BUF = 32 * (C * 2 + 32) + C * 96;
  static bool attr_set = false;
  if (!attr_set) {  // > 64 KB of dynamic LDS needs the opt-in once per kernel
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&convffn_kernel<C, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF + 8 * C));
    attr_set = true;
  }
  const long tiles = ((long)p.M + 64 * MT - 1) / (64 * MT);
  const long blocks = tiles < num_cus() ? tiles : num_cus();   // one persistent block per CU, tiles dealt round-robin
  hipLaunchKernelGGL((convffn_kernel<C, MT>), dim3((unsigned)blocks), dim3(256), 2 * BUF + 8 * C, s, p);
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
