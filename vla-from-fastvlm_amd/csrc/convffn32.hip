// convffn32.hip -- the fused ConvFFN pointwise half (see convffn_fused.hip for the formulation) on the 32x32x16 MFMA:
//     out = res + ls * ( fc2( GELU( fc1(x) + b1 ) ) + b2 )          C in {96, 192, 384}, hidden = 4C
// ([UNVENDORED] mci.py ConvFFN.forward + RepMixerBlock layer-scale residual; call site of the whole VLM:
// reference src/vla_fastvlm/model/fastvlm_adapter.py:533).
//
// Why a second kernel.  convffn_kernel is bound by what ONE wave per SIMD has to issue beside its MFMAs:
// per 32-hidden chunk 96 x v_mfma_f32_16x16x32_bf16 hold the SIMD's vector issue for 8 of their 16 cycles each -- 768 of the
// chunk's 1536 MFMA cycles -- and the fragment reads, the weight staging and the GELU queue up behind them.  The same
// products on v_mfma_f32_32x32x16_bf16 are 48 instructions of 32 cycles that hold issue for 8: 384 cycles held, 1152 free,
// with the SAME operand traffic (a 1 KB fragment still feeds 16 K MACs), the same registers (C/2 output accumulators,
// C/4 x-fragment registers per 32 rows) and half the MFMA instructions to interleave with.
//
//   per wave: MT tiles of 32 pixels; per block: NW = 4 waves (one per SIMD, 512 registers each; C = 384, 96) or 8 (two per SIMD, 256
//   registers, no AGPRs; C = 192) sharing the weight stream (one persistent block per CU)
//   chunk of 32 hidden units:
//     H^T[32 hid x 32 px] = W1[chunk] . x^T      A = W1 rows (lane r = hidden row, half h = k 8h..8h+7 of the 16-deep step)
//                                                B = x fragments in registers
//     bias + GELU in registers.  D layout: column = pixel (lane & 31), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): registers
//     8s .. 8s+7 converted pairwise ARE the B fragment of k-step s of the second product once W2's hidden columns are
//     permuted to 16 s + 8 (j >> 2) + 4 h + (j & 3) at pack time (cdna_hip_programming.md section 3, "An accumulator tile as the
//     next MFMA's operand")
//     out^T[32 ch x 32 px] += W2[:, chunk] . H^T  two k-steps per output tile, accumulators live across all chunks
//
// Round 3: two pack-time folds, both exact, take VALU work out of the chunk.
//   * W1 and b1 are divided by 4 and W2 multiplied by 4 (powers of two: exact in bf16 / fp32).  The first product then yields
//     y = (fc1(x) + b1) / 4 and the GELU works in y: u = clamp(y^2, 0, 1) costs one clamped multiply instead of two v_med3 + a
//     multiply, and g = y * clamp(1/2 + y Q(u), 0, 1) = GELU(4 y) / 4 is rounded to bf16 exactly as GELU(x) would be
//     (tools/gelu_fit_scaled.py: degree-6 Q, max |error| 1.9e-4 in x units; gelu2s_n in common.h);
//   * b1 / 4 enters as the C operand of each hidden tile's first MFMA (its 16 registers ARE the four float4 the bias table holds
//     for this lane half, fetched one chunk ahead), not as 16 VALU adds per tile.
//   10 VALU issues per hidden pair instead of 13: -3.5 % / 0 / -4 % per launch at C = 384 / 192 / 96 (tools/ffn_bench.py, four rounds).
// Late round 3: the template also carries the 16x16x32 form of the same chunk (S16: a wave's 32 pixels as two 16-pixel column tiles,
// every fragment read feeds two MFMAs, W2 packed for ONE K = 32 step) behind FFN32_S16_MASK -- bit-correct, same time as this form at
// every width (the launch runs at the board's power cap: DESIGN.md section 5.0), so it is not enabled.
// What was ALSO built this round and is NOT in this file (tools/experiments/convffn32_dma_staged_epilogue.patch): a row-per-lane
// epilogue straight from the accumulators (W2 rows permuted so that a lane holds 8 consecutive channels) and one that moves
// residual, output and the next tile's x fragments through wave-private 6 KB LDS units by LDS-DMA with hand-counted vmcnt.  Both
// are bit-correct; neither beats the fp32 LDS turn below by more than 1 % (see the patch header for the stamps).
#include <vector>

#include "kernels.h"

#ifndef FFN32_BUF   /* bit 0: the x fragments through a buffer descriptor too (residual and output always are) */
#define FFN32_BUF 0
#endif
#define F32_BUFX ((FFN32_BUF & 1) != 0)
// cache-policy bits (buffer aux operand; gfx950: bit 1 = nt) of the three activation streams -- x rows, residual rows, output rows.  They pass through a launch once
// while the 4.8 GB weight stream is re-read from L2 by every block: marked non-temporal they should not evict it (VERDICT r4 #2-iv: 764 MB of HBM traffic per
// C = 384 launch against 604 MB algorithmic).  tools/ffn32_variants.sh A/B; defaults = the measured winner.
#ifndef FFN32_NT_X
#define FFN32_NT_X 0
#endif
#ifndef FFN32_NT_RES
#define FFN32_NT_RES 0
#endif
#ifndef FFN32_NT_OUT
#define FFN32_NT_OUT 0
#endif

#ifdef FFN32_STAMPS   /* tools/ffn32_variants.sh diagnostic build only: cycle sums per phase of block 0..255, wave 0 */
__device__ unsigned long long g_ffn32_stamps[256 * 8];
extern "C" int fv_dbg_ffn32_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ffn32_stamps), sizeof(g_ffn32_stamps));
}
#define F32_STAMP(VAR) unsigned long long VAR; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(VAR) :: "memory");
#else
#define F32_STAMP(VAR)
#endif

namespace fv {
namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct Ffn32Params {
  const bf16_t* x; const bf16_t* wq; const float* b1; const float* b2; const float* ls;
  const bf16_t* res; bf16_t* out; int M, nchunks;
  float* part = nullptr; int hsplit = 1;   // PART instances: nchunks = chunks PER RANGE, raw fp32 sums to part[range][M][C]
  // STASH instances (the tower's TRAINING forward, round 6): the pre-activation leaves the chip on the way -- sy[M][4C] fp16 = a / 4 (the kernel's own scaling) -- so
  // that the backward needs no recompute GEMM: the fc2 input gradient's epilogue (FV_EPI_MUL_GELUP) multiplies by gelu'(4 sy) and writes gelu(4 sy), the fc2 weight
  // gradient's operand, from the same read.  (Stashing gelu(a) here as well was measured: the kernel runs at the power cap, every byte it stores costs its energy in
  // time -- 315 -> 471 us per C = 384 launch with both tensors; the memory-bound epilogue over there writes the second tensor almost for free.)
  bf16_t* sy = nullptr;
};
__device__ __forceinline__ uint32_t pk_h2_rtz(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b)); }

// output accumulators pinned to the accumulator half of the register file, hidden-tile accumulators to the architectural
// half (the GELU reads them); see convffn_fused.hip for why these MFMAs are asm statements
template <bool ACC_A = true>   // ACC_A: the accumulator half of the file (one wave per SIMD); false: plain VGPRs (two waves per SIMD share 512)
__device__ __forceinline__ void mfma32_out(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  if constexpr (ACC_A) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <bool BA>  // BA: the x fragment lives in the accumulator half too
__device__ __forceinline__ void mfma32_hid(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  if constexpr (BA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma32_hid_init(f32x16& acc, const bf16x8& a, const bf16x8& b, const f32x16& bias) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(bias));
}
// the same three on v_mfma_f32_16x16x32_bf16 (S16 form of the kernel: a 32-pixel wave tile is two 16-pixel column tiles)
template <bool ACC_A = true>
__device__ __forceinline__ void mfma16_out(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  if constexpr (ACC_A) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <bool BA>
__device__ __forceinline__ void mfma16_hid(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  if constexpr (BA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_hid_init(f32x4& acc, const bf16x8& a, const bf16x8& b, const f32x4& bias) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "v"(b), "v"(bias));
}
// XDL write -> VALU read of the hidden accumulators (8-pass MFMA: 12+ wait states), paid once per chunk behind the first
// product's last MFMA; and VALU write -> MFMA B operand ahead of the second product
template <int N>
__device__ __forceinline__ void settle_hid(f32x16 (&h)[N]) {
  static_assert(N == 1 || N == 2 || N == 4, "row tiles per wave");
  if constexpr (N == 1) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(h[0]));
  else if constexpr (N == 2) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(h[0]), "+v"(h[1]));
  else asm volatile("s_nop 7\n\ts_nop 7" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]));
}
template <int N>
__device__ __forceinline__ void settle_ops(bf16x8 (&v)[N][2]) {
  if constexpr (N == 1) asm volatile("s_nop 3" : "+v"(v[0][0]), "+v"(v[0][1]));
  else if constexpr (N == 2) asm volatile("s_nop 3" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[1][0]), "+v"(v[1][1]));
  else asm volatile("s_nop 3" : "+v"(v[0][0]), "+v"(v[0][1]), "+v"(v[1][0]), "+v"(v[1][1]), "+v"(v[2][0]), "+v"(v[2][1]), "+v"(v[3][0]), "+v"(v[3][1]));
}

#define F32_ST_GET(J)                                                                                              \
  ((J) == 0 ? st0 : (J) == 1 ? st1 : (J) == 2 ? st2 : (J) == 3 ? st3 : (J) == 4 ? st4 : (J) == 5 ? st5 : (J) == 6 ? st6 \
   : (J) == 7 ? st7 : (J) == 8 ? st8 : (J) == 9 ? st9 : (J) == 10 ? st10 : st11)
#define F32_ST_SET(J, V)                                                                                           \
  {                                                                                                                \
    const uint4 v_ = (V);                                                                                          \
    if ((J) == 0) st0 = v_; else if ((J) == 1) st1 = v_; else if ((J) == 2) st2 = v_; else if ((J) == 3) st3 = v_;  \
    else if ((J) == 4) st4 = v_; else if ((J) == 5) st5 = v_; else if ((J) == 6) st6 = v_; else if ((J) == 7) st7 = v_; \
    else if ((J) == 8) st8 = v_; else if ((J) == 9) st9 = v_; else if ((J) == 10) st10 = v_; else st11 = v_;        \
  }

constexpr int ring_depth32(int nr, int want) { return nr % want == 0 ? want : ring_depth32(nr, want - 1); }

// LDS plan shared by the kernel, its launcher and the weight packer.
// A weight slot holds one chunk's W1 image (32 hidden rows x C, 2C bytes per row) followed by its W2 image (C output-channel
// rows x 32 hidden, 64 bytes per row), both UNPADDED: staging writes them in whole 256-byte runs (ds_write_addtid_b32: a
// wave-instruction stores 64 lanes x 4 bytes at M0 + offset + 4 * lane, no address register, 2 cycles -- stage_micro.hip:
// buffer load + 4 of these cost 12 clk of a wave's issue time per 1 KB piece beside 32x32x16 MFMAs, a global load + ds_write_b128
// 47), so row padding is not available and the bank spread comes from an XOR swizzle of the 16-byte chunk index instead:
//   W1 chunk c of row r lives at chunk c ^ ((r >> SWS1) & SWM1); W2 chunk c (= 2 s + h) of row n at chunk c ^ ((n >> 2) & 3).
// Both make the 16 lanes of every ds_read_b128 lane group (distinct rows mod 16, same logical chunk) hit 16 different 16-byte
// slots of the 256-byte bank row.
template <int C, int MT, int NW = 4>
struct Ffn32Lds {
  static constexpr int ROW1 = C * 2, ROW2 = 64;
  static constexpr int SWS1 = C == 384 ? 0 : C == 192 ? 1 : 2, SWM1 = C == 384 ? 15 : C == 192 ? 7 : 3;
  static constexpr int W1_BYTES = 32 * ROW1, W2_BYTES = C * ROW2, BUF = W1_BYTES + W2_BYTES;   // 128 C bytes, a multiple of 1 KB
  static constexpr int CQ = 96;                  // channels per epilogue pass
  static constexpr int ORB = CQ * 4 + 16;        // fp32 row of a pass + 16 B (conflict-free 16-B column writes)
  static constexpr int EPI_WAVE = 32 * ORB;      // one pass of one wave
  static constexpr int TABLES = 2 * C * 4 + 4 * C * 4;                      // ls * b2, ls, b1 (fp32)
  static constexpr int TOTAL = 2 * BUF + TABLES + NW * EPI_WAVE;
  static_assert(BUF % 1024 == 0 && TOTAL <= 160 * 1024, "LDS");
};

// NW = waves per block.  4 (C = 384, 96): one wave per SIMD with all 512 registers.  8 (C = 192): two waves per SIMD with 256 registers each,
// no AGPRs, 32 * MT rows per wave as before -- one wave's GELU and epilogue run beside its partner's MFMAs (FFN32_NW192 = 4 restores the old shape).
// PART (few row tiles: one to four observations give C = 384 32 .. 128 tiles for 256 CUs, and a tile costs its whole 2.4 MB weight pass whoever else is
// idle): block = (row tile, one of hsplit ranges of hidden chunks).  The kernel is the same chunk loop over a weight stream that starts at the range's
// first chunk; the epilogue leaves the raw fp32 output sums in p.part[range] and ffn32_reduce_kernel adds the ranges, bias, layer scale and residual.
template <int C, int MT, int NW = 4, bool S16 = false, bool PART = false, bool STASH = false>
__global__ __launch_bounds__(64 * NW, NW / 4) void convffn32_kernel(Ffn32Params p) {
  static_assert(!STASH || (!S16 && !PART), "the stash rides on the one-launch 32x32x16 form");
  using L = Ffn32Lds<C, MT, NW>;
  constexpr int NTH = 64 * NW, TROWS = 32 * NW * MT;   // threads per block, rows per row tile set
  constexpr int N1 = C / 16;                   // fragment reads of the first product (either MFMA shape)
  constexpr int KS = S16 ? C / 32 : C / 16;    // k-steps of the first product (S16: two fragments -- hidden tiles -- per step)
  constexpr int NT = S16 ? C / 16 : C / 32;    // output-channel tiles of the second product (32x32x16: two k-steps each; S16: one)
  constexpr int ROW1 = L::ROW1, W1_BYTES = L::W1_BYTES, BUF = L::BUF;
  constexpr int SW = (BUF / 1024) % NW == 0 ? NW : 4;         // waves that stage the weight stream (all of them when the chunk's KBs divide evenly)
  constexpr int NLD = BUF / 1024 / SW;                        // 1 KB staging pieces per staging wave per chunk
  static_assert(BUF % (1024 * SW) == 0 && NLD <= 12 && 2 * NLD <= N1, "staging schedule: one store or one load per first-product read");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][BUF] weight slots, ls*b2[C], ls[C], b1[4C], epilogue staging

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = S16 ? lane & 15 : lane & 31, fh = S16 ? lane >> 4 : lane >> 5;
  const int nch = p.nchunks;                                   // C / 8 (PART: per range): even, >= 4
  const int ntiles = (p.M + TROWS - 1) / TROWS;
  const int hrange = PART ? (int)blockIdx.x % p.hsplit : 0, tile0 = PART ? (int)blockIdx.x / p.hsplit : (int)blockIdx.x;
  // the packed weight stream through a buffer descriptor: a buffer load costs a wave ~6 clk of issue beside the MFMAs where a
  // global load costs ~23 (tools/stage_micro.hip); lane offset in a VGPR (piece of wave wid, 16 B per lane), chunk and piece in
  // the scalar offset
  const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(const_cast<bf16_t*>(p.wq)) + (size_t)hrange * nch * BUF, 0, nch * BUF, 0x00020000);
  const uint32_t wvoff = (uint32_t)tid * 16u;
  // activations through descriptors too (32-bit byte offsets: M * C * 2 < 4 GiB, checked by the launcher)
  const uint32_t act_bytes = (uint32_t)p.M * (uint32_t)(C * 2);
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, act_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.res), 0, act_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, act_bytes, 0x00020000);
  // STASH: [M][4C] fp16 rows (8 C bytes each: the launcher keeps M * 8 C below 2 GiB); a row past M carries an offset the descriptor drops
  const __amdgpu_buffer_rsrc_t syrsrc = __builtin_amdgcn_make_buffer_rsrc(STASH ? p.sy : nullptr, 0, STASH ? act_bytes * 4u : 0u, 0x00020000);
  uint32_t srow[STASH ? MT : 1];   // byte offset of the lane's pixel row (+ 16 fh) in the stash tensors, per tile
  // LDS address of this wave's first 1 KB piece in slot 0 (wave-uniform: M0 of the add-tid stores)
  const uint32_t wm0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)smem + (uint32_t)wid * 1024u);

  // ---- x fragments: lane holds x[pixel m = fr][16 ks + 8 fh .. +8] for its MT pixel tiles (rows past M clamp to M - 1)
  // S16: xg[mt][n][ks] = x[pixel 16 n + (lane & 15)][32 ks + 8 (lane >> 4) .. +8]
  bf16x8 xf[S16 ? 1 : MT][S16 ? 1 : KS];
  bf16x8 xg[S16 ? MT : 1][2][S16 ? KS : 1];
#define F32_LOAD_X(TILE)                                                                                     \
  {                                                                                                          \
    int lx_ = lane;                                                                                          \
    asm volatile("" : "+v"(lx_)); /* keeps this address math out of the chunk loop's live registers */       \
    const long mb_ = (long)(TILE) * TROWS + wid * (32 * MT);                                                 \
    if constexpr (S16) {                                                                                     \
      _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                      \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                      \
          const long m_ = min(mb_ + mt * 32 + n * 16 + (lx_ & 15), (long)p.M - 1);                           \
          const uint32_t xo_ = (uint32_t)m_ * (uint32_t)(C * 2) + (uint32_t)(lx_ >> 4) * 16u;                \
          _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                                  \
            xg[mt][n][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.x) + xo_ + ks * 64)); \
        }                                                                                                    \
    } else                                                                                                   \
    _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                      \
      const long m_ = min(mb_ + mt * 32 + (lx_ & 31), (long)p.M - 1);                                        \
      const uint32_t xo_ = (uint32_t)m_ * (uint32_t)(C * 2) + (uint32_t)(lx_ >> 5) * 16u;                    \
      _Pragma("unroll") for (int ks = 0; ks < KS; ++ks)                                                      \
        xf[mt][ks] = F32_BUFX ? __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xo_ + ks * 32, 0, FFN32_NT_X)) \
                              : __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.x) + xo_ + ks * 32)); \
    }                                                                                                        \
  }
  F32_LOAD_X(tile0)
  f32x16 oacc[S16 ? 1 : NT][S16 ? 1 : MT];
  f32x4 oacc16[S16 ? NT : 1][S16 ? MT : 1][2];

  uint4 st0, st1, st2, st3, st4, st5, st6, st7, st8, st9, st10, st11;
  st0 = st1 = st2 = st3 = st4 = st5 = st6 = st7 = st8 = st9 = st10 = st11 = make_uint4(0, 0, 0, 0);
  // piece j of this wave: bytes [(4 j + wid) KB, +1 KB) of the chunk; each lane's 16 bytes leave as four dword stores that land
  // 256 bytes apart ([dword][lane] inside the KB) -- the packer (convffn32_pack) lays the global image out so that this IS the
  // swizzled LDS image
#define F32_PIECE_LOAD(J, HC) F32_ST_SET(J, __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, (HC) * BUF + (J) * (1024 * SW), 0)))
#define F32_PIECE_STORE(J, SLOT)                                                                             \
  {                                                                                                          \
    const uint4 v_ = F32_ST_GET(J);                                                                          \
    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tds_write_addtid_b32 %0\n\tds_write_addtid_b32 %1 offset:256\n\t"   \
                 "ds_write_addtid_b32 %2 offset:512\n\tds_write_addtid_b32 %3 offset:768"                     \
                 :: "v"(v_.x), "v"(v_.y), "v"(v_.z), "v"(v_.w), "s"(wm0 + (uint32_t)((SLOT) * BUF + (J) * (1024 * SW))) : "memory"); \
  }
  const bool stager = SW == NW || wid < SW;   // wave-uniform
  if (stager) {
#pragma unroll
    for (int j = 0; j < NLD; ++j) F32_PIECE_LOAD(j, 0)
#pragma unroll
    for (int j = 0; j < NLD; ++j) F32_PIECE_STORE(j, 0)
#pragma unroll
    for (int j = 0; j < NLD; ++j) F32_PIECE_LOAD(j, 1)   // rides in registers through chunk 0, stored to LDS during it
  }
  // second-product bias, layer scale and first-product bias, staged once per block
  float* sb2 = reinterpret_cast<float*>(smem + 2 * BUF);
  float* sls = sb2 + C;
  float* sb1 = sls + C;
  for (int i = tid; i < C / 4; i += NTH) {   // sb2 holds ls * b2: out = res + (ls * acc + ls * b2), one fma per channel
    const float4 b = reinterpret_cast<const float4*>(p.b2)[i], l = reinterpret_cast<const float4*>(p.ls)[i];
    reinterpret_cast<float4*>(sb2)[i] = make_float4(b.x * l.x, b.y * l.y, b.z * l.z, b.w * l.w);
    reinterpret_cast<float4*>(sls)[i] = l;
  }
  for (int i = tid; i < (PART ? nch * 8 : C); i += NTH) {       // sb1 holds b1 / 4: the first product runs on W1 / 4 (exact), see the GELU
    const float4 b = reinterpret_cast<const float4*>(p.b1 + (size_t)hrange * nch * 32)[i];
    reinterpret_cast<float4*>(sb1)[i] = make_float4(0.25f * b.x, 0.25f * b.y, 0.25f * b.z, 0.25f * b.w);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the add-tid stores are invisible to hipcc's counters
  __syncthreads();

  // Fragment stream of one chunk: reads 0..KS-1 are W1 fragments (k-step i), reads KS..NR-1 are W2 fragments (output tile
  // (i - KS) >> 1, k-step (i - KS) & 1).  The reads run PD steps ahead of the MFMAs that consume them, across the chunk
  // boundary too (the chunk's one barrier sits PD steps before its end): see convffn_fused.hip.
#ifndef FFN32_PD
#define FFN32_PD 6
#endif
  constexpr bool XA = NW == 4;             // upper half of the x fragments in AGPRs (MFMA reads A / B from either half); NW = 8: no AGPRs at all
                                           // (with any "a" operand hipcc splits a 256-register wave 128 / 128 and spills the VGPR side)
  constexpr int NR = N1 + C / 16, PD = ring_depth32(NR, FFN32_PD < NR / 2 ? FFN32_PD : NR / 2);
  static_assert(2 * NLD <= NR - PD && NR % PD == 0, "ring");
  // swizzled fragment addresses: W1 step i reads logical chunk 2 i + fh of row fr, i.e. physical chunk (2 i + fh) ^ s1 -- the XOR
  // touches the low bits only, so the lane part repeats every NA1 steps and the rest is an immediate offset; W2 step (t, s)
  // reads logical chunk 2 s + fh of row 32 t + fr
  // S16: read 2 ks + m of the first product is hidden tile m (rows 16 m + fr), logical chunk 4 ks + fh; read t of the second is
  // W2 rows 16 t + fr, chunk fh ^ s2 with s2 = (-(row >> 2)) & 3 (the table that keeps the 16x16x32 lane groups conflict-free)
  constexpr int GB1 = L::SWM1 + 1, NA1 = S16 ? (GB1 >= 4 ? GB1 / 4 : 1) : GB1 / 2;
  uint32_t fa1[NA1], fa2[2];
  {
    const uint32_t s1 = (uint32_t)(fr >> L::SWS1) & L::SWM1, s2 = S16 ? (4u - ((uint32_t)fr >> 2)) & 3u : (uint32_t)(fr >> 2) & 3u;
#pragma unroll
    for (int m = 0; m < NA1; ++m) fa1[m] = (uint32_t)fr * ROW1 + 16u * ((((uint32_t)((S16 ? 4 : 2) * m + fh)) & L::SWM1) ^ s1);
#pragma unroll
    for (int q = 0; q < 2; ++q) fa2[q] = (uint32_t)W1_BYTES + (uint32_t)fr * 64u + 16u * ((uint32_t)(S16 ? fh : 2 * q + fh) ^ s2);
  }
#define F32_FRAG_32(I, SLOTP)                                                                                       \
  ((I) < N1 ? *reinterpret_cast<const uint4*>((SLOTP) + fa1[(I) % NA1] + ((I) / NA1) * (GB1 * 16))                  \
            : *reinterpret_cast<const uint4*>((SLOTP) + fa2[((I) - N1) & 1] + (((I) - N1) >> 1) * 2048))
#define F32_FRAG_16(I, SLOTP)                                                                                       \
  ((I) < N1 ? *reinterpret_cast<const uint4*>((SLOTP) + fa1[((I) >> 1) % NA1] + (((I) >> 1) / NA1) * (GB1 * 16) + ((I) & 1) * (16 * ROW1)) \
            : *reinterpret_cast<const uint4*>((SLOTP) + fa2[0] + ((I) - N1) * 1024))
#define F32_FRAG(I, SLOTP) (S16 ? F32_FRAG_16(I, SLOTP) : F32_FRAG_32(I, SLOTP))
  uint4 ring[PD];
#pragma unroll
  for (int i = 0; i < PD; ++i) ring[i] = F32_FRAG(i, smem);

  // epilogue geometry: one pass turns 32 rows x CQ channels of fp32 through the wave's staging area
  constexpr int CQ = L::CQ, ORB = L::ORB, NPASS = C / CQ;      // passes per row tile
  // read-out of a pass: a lane keeps ONE 8-channel chunk (ec = lane % 12; lanes 60..63 idle) and walks the rows five at a time
  // (er = lane / 12): its bias / layer-scale registers are loaded once per pass, the index arithmetic once per tile, and an
  // instruction still covers five whole 192-byte row segments
  constexpr int ECH = CQ / 8, ERW = 64 / ECH, RP = (32 + ERW - 1) / ERW;   // 12 chunks, 5 rows per instruction, 7 instructions
  static_assert(C % CQ == 0 && CQ % 32 == 0 && ECH == 12, "epilogue pass");
  // epilogue lane constants (tile-independent)
  int le = lane;
  asm volatile("" : "+v"(le));
  const int er = le / ECH, ec8 = (le - er * ECH) * 8;                       // row within a group of ERW, first of the lane's 8 channels
  // idle lanes / rows: an offset beyond any descriptor (the launcher keeps M * C * 2 < 2 GiB, so adding a row offset to it never
  // wraps).  The hardware's range check covers the VGPR offset only, NOT the scalar offset: every byte of the row address
  // therefore goes into the VGPR -- a ragged last tile's rows past M are then dropped by the same check.
  const uint32_t OOB = 0x80000000u;
  const uint32_t eoff = er < ERW ? (uint32_t)(er * C + ec8) * 2u : OOB;      // byte offset of the lane's chunk in row er
  const uint32_t eoff_last = (er < ERW && (RP - 1) * ERW + er < 32) ? eoff : OOB;   // last instruction: rows 30, 31 only
  char* const so = smem + 2 * BUF + L::TABLES + wid * L::EPI_WAVE;           // the wave's fp32 staging area
  char* const sow = so + (S16 ? le & 15 : le & 31) * ORB + (S16 ? le >> 4 : le >> 5) * 16;   // write side: pixel fr, channel quad 4 fh
  const char* const sor = so + min(er, ERW - 1) * ORB + ec8 * 4;            // read side: row er, the lane's 8 channels
  const char* const sor_last = so + min((RP - 1) * ERW + er, 31) * ORB + ec8 * 4;
#ifndef FFN32_XPASS   /* epilogue pass in which the next tile's x fragments are requested (default: the last) */
#define F32_XPASS (NPASS * MT - 1)
#else
#define F32_XPASS (FFN32_XPASS < NPASS * MT ? FFN32_XPASS : NPASS * MT - 1)
#endif
#ifndef FFN32_RD
#define FFN32_RD (NPASS * MT)
#endif
  // residual rows: ALL passes of the tile are requested at the top of the epilogue (rr reuses the registers the x fragments
  // just vacated): a pass is far shorter than an HBM read, so fetching one pass ahead exposed one memory latency per pass
  constexpr int RD = FFN32_RD;
  uint4 rr[RD][RP];
  // rows past M (ragged last tile) and the idle lanes / rows carry an offset outside the descriptor: loads return 0, stores vanish
#define F32_LOAD_RES(Q, MB)                                                                                  \
  {                                                                                                          \
    const uint32_t so_ = (uint32_t)((MB) + ((Q) / NPASS) * 32) * (uint32_t)(C * 2) + (uint32_t)(((Q) % NPASS) * CQ * 2); \
    _Pragma("unroll") for (int it = 0; it < RP; ++it)                                                        \
      rr[(Q) % RD][it] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(                    \
          rrsrc, F32_ABL_NORES ? OOB : (it == RP - 1 ? eoff_last : eoff) + (so_ + (uint32_t)(it * (ERW * C * 2))), 0, FFN32_NT_RES)); \
  }

#ifdef FFN32_ABL_NOOUT  /* tools/ffn32_variants.sh only: every output store falls outside the descriptor (dropped) */
#define F32_ABL_NOOUT 1
#else
#define F32_ABL_NOOUT 0
#endif
#ifdef FFN32_ABL_NORES  /* ... every residual load falls outside the descriptor (returns 0, no memory traffic) */
#define F32_ABL_NORES 1
#else
#define F32_ABL_NORES 0
#endif
#ifdef FFN32_ABL_GELU   /* tools/ffn32_ablate.sh only: identity activation, to price the GELU */
#define F32_GELU(G)
#else
#ifdef FFN32_GELU9
#define F32_GELU(G) gelu2_n<4>(G);
#else
#define F32_GELU(G) gelu2s_n<4>(G);   /* y = x / 4 in, GELU(x) / 4 out: the hidden is rounded to bf16 right behind it */
#endif
#endif
#ifdef FFN32_ABL_STAGE  /* ... no weight staging (global loads, LDS stores): wrong results, same MFMAs */
#define F32_ABL_STAGE 1
#else
#define F32_ABL_STAGE 0
#endif
#ifdef FFN32_ABL_FRAG   /* ... no fragment reads past the first ring fill */
#define F32_ABL_FRAG 1
#else
#define F32_ABL_FRAG 0
#endif
  // the chunk's bias b1 / 4 in the first product's D layout: registers 4 q .. 4 q + 3 = hidden 8 q + 4 fh + 0..3 of the chunk.
  // Loaded one chunk ahead (behind the previous chunk's GELU, long after the MFMAs that read the old value as C have retired).
  f32x16 bias;
  f32x4 bias16[2];   // S16: hidden 16 m + 4 fh + 0..3 of the chunk
#define F32_LOAD_BIAS(HC)                                                                                    \
  {                                                                                                          \
    if constexpr (S16) {                                                                                     \
      _Pragma("unroll") for (int m_ = 0; m_ < 2; ++m_) {                                                     \
        const float4 b_ = *reinterpret_cast<const float4*>(sb1 + (HC) * 32 + 16 * m_ + 4 * fh);              \
        bias16[m_] = f32x4{b_.x, b_.y, b_.z, b_.w};                                                          \
      }                                                                                                      \
    } else                                                                                                   \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                       \
      const float4 b_ = *reinterpret_cast<const float4*>(sb1 + (HC) * 32 + 8 * q_ + 4 * fh);                 \
      bias[4 * q_] = b_.x; bias[4 * q_ + 1] = b_.y; bias[4 * q_ + 2] = b_.z; bias[4 * q_ + 3] = b_.w;        \
    }                                                                                                        \
  }
  F32_LOAD_BIAS(0)
#define F32_CHUNK                                                                                                   \
  {                                                                                                                 \
    const int hb = hc + 1 == nch ? 0 : hc + 1;                                                                      \
    const int cur = hc & 1;                                                                                         \
    /* weight staging two chunks deep: during chunk hc the registers loaded during chunk hc-1 (chunk hc+1's weights) */ \
    /* go to the idle slot and are refilled with chunk hc+2's; the chunk index wraps into the next tile            */ \
    const int hn = hc + 2 - (hc + 2 >= nch ? nch : 0);                                                              \
    const char* cbase = smem + cur * BUF;                                                                           \
    const char* nbase = smem + (cur ^ 1) * BUF;                                                                     \
    f32x16 hacc[MT];                                                                                                \
    bf16x8 hf[MT][2];                                                                                               \
    f32x4 hacc16[MT][2][2]; /* S16: [pixel tile n][hidden tile m] */                                                \
    bf16x8 hf16[MT][2];                                                                                             \
    _Pragma("unroll") for (int i = 0; i < NR; ++i) {                                                                \
      if (i == NR - PD) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __syncthreads(); }                     \
      const bf16x8 a = __builtin_bit_cast(bf16x8, ring[i % PD]);                                                    \
      if (!F32_ABL_FRAG) ring[i % PD] = i + PD < NR ? F32_FRAG(i + PD, cbase) : F32_FRAG(i + PD - NR, nbase);       \
      if (i < N1) {                                                                                                 \
        if constexpr (S16) { /* H^T[hidden tile m] += W1[rows 16 m .., k-step ks] . x^T, both pixel tiles */         \
          const int ks = i >> 1, m = i & 1;                                                                         \
          _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                         \
            _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                         \
              if (ks == 0) mfma16_hid_init(hacc16[mt][n][m], a, xg[mt][n][0], bias16[m]);                            \
              else if (XA && ks >= KS / 2) mfma16_hid<XA>(hacc16[mt][n][m], a, xg[mt][n][ks >= KS / 2 ? ks : KS - 1]); \
              else mfma16_hid<false>(hacc16[mt][n][m], a, xg[mt][n][ks]);                                           \
            }                                                                                                       \
        } else { /* H^T += W1[chunk rows, k-step i] . x^T */                                                        \
          _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                       \
            if (i == 0) mfma32_hid_init(hacc[mt], a, xf[mt][0], bias);                                              \
            else if (XA && i >= KS / 2) mfma32_hid<XA>(hacc[mt], a, xf[mt][i >= KS / 2 ? i : KS - 1]);              \
            else mfma32_hid<false>(hacc[mt], a, xf[mt][i]);                                                         \
          }                                                                                                         \
        }                                                                                                           \
        if (!F32_ABL_STAGE) { /* staging: even steps store piece j to the idle slot, odd steps reload its register for the chunk after */ \
          const int j = i >> 1;                                                                                     \
          if (j < NLD && stager) {                                                                                  \
            if ((i & 1) == 0) F32_PIECE_STORE(j, cur ^ 1)                                                           \
            else F32_PIECE_LOAD(j, hn)                                                                              \
          }                                                                                                         \
        }                                                                                                           \
        if (i == N1 - 1) { /* bias + GELU in registers -> the B fragments of the second product */                  \
          if constexpr (S16) {                                                                                      \
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");                                                        \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                       \
              _Pragma("unroll") for (int n = 0; n < 2; ++n) asm volatile("" : "+v"(hacc16[mt][n][0]), "+v"(hacc16[mt][n][1])); \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                     \
              _Pragma("unroll") for (int n = 0; n < 2; ++n) { /* k slot 8 fh + j: hidden 4 fh + j (j < 4), 16 + 4 fh + j - 4 (j >= 4) */ \
                const f32x4 h0 = hacc16[mt][n][0], h1 = hacc16[mt][n][1];                                           \
                f32x2 g[4] = {{h0[0], h0[1]}, {h0[2], h0[3]}, {h1[0], h1[1]}, {h1[2], h1[3]}};                      \
                F32_GELU(g)                                                                                         \
                uint4 u;                                                                                            \
                u.x = pack_bf2(g[0].x, g[0].y); u.y = pack_bf2(g[1].x, g[1].y);                                     \
                u.z = pack_bf2(g[2].x, g[2].y); u.w = pack_bf2(g[3].x, g[3].y);                                     \
                hf16[mt][n] = __builtin_bit_cast(bf16x8, u);                                                        \
                __builtin_amdgcn_sched_barrier(0);                                                                  \
              }                                                                                                     \
            }                                                                                                       \
            F32_LOAD_BIAS(hb)                                                                                       \
            asm volatile("s_nop 3" ::: "memory");                                                                   \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(hf16[mt][0]), "+v"(hf16[mt][1])); \
          } else {                                                                                                  \
            settle_hid<MT>(hacc);                                                                                   \
            _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                                     \
              _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                       \
                f32x2 g[4] = {{hacc[mt][8 * s + 0], hacc[mt][8 * s + 1]}, {hacc[mt][8 * s + 2], hacc[mt][8 * s + 3]}, \
                              {hacc[mt][8 * s + 4], hacc[mt][8 * s + 5]}, {hacc[mt][8 * s + 6], hacc[mt][8 * s + 7]}}; \
                uint32_t yq[4];                                                                                     \
                if constexpr (STASH) { _Pragma("unroll") for (int e = 0; e < 4; ++e) yq[e] = pk_h2_rtz(g[e].x, g[e].y); } \
                F32_GELU(g)                                                                                         \
                uint4 u;                                                                                            \
                u.x = pack_bf2(g[0].x, g[0].y); u.y = pack_bf2(g[1].x, g[1].y);                                     \
                u.z = pack_bf2(g[2].x, g[2].y); u.w = pack_bf2(g[3].x, g[3].y);                                     \
                hf[mt][s] = __builtin_bit_cast(bf16x8, u);                                                          \
                if constexpr (STASH) {                                                                              \
                  /* the lane holds a / 4 of hidden 16 s + 4 fh + 0..3 (dwords 0, 1) and 16 s + 8 + 4 fh + 0..3 (dwords 2, 3) of its pixel; the lane 32 further   */ \
                  /* (fh ^ 1) holds the runs in between: one v_permlane32_swap per dword pair gives fh = 0 the hidden 16 s + 0..7 and fh = 1 the hidden            */ \
                  /* 16 s + 8..15 -- 16 contiguous bytes per lane, 32 per pixel row and instruction                                                                */ \
                  _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                   \
                    const auto sy_ = __builtin_amdgcn_permlane32_swap(yq[e], yq[e + 2], false, false);              \
                    yq[e] = sy_[0]; yq[e + 2] = sy_[1];                                                             \
                  }                                                                                                 \
                  typedef __attribute__((ext_vector_type(4))) unsigned int su4_;                                    \
                  /* the chunk's column offset rides in the VGPR offset, NOT in the scalar one: with a register in soffset hipcc assumes that a VALU write of the    */ \
                  /* store's data registers right behind a 16-byte buffer store is safe (GCNHazardRecognizer: "only if not using a register in soffset") -- on gfx950 */ \
                  /* it is not: the C = 192 instance stored its NEXT instruction's results in lanes 12-15 / 28-31 (tests/test_gpu_ops.py, stash)                      */ \
                  __builtin_amdgcn_raw_buffer_store_b128(su4_{yq[0], yq[1], yq[2], yq[3]}, syrsrc, srow[mt] + (uint32_t)(hc * 64 + s * 32), 0, 0); \
                  asm volatile("s_nop 1" ::: "memory"); /* ... and the statements that follow are inline asm, which hipcc pads for nothing: the wait states by hand */ \
                }                                                                                                   \
                __builtin_amdgcn_sched_barrier(0); /* one group of four chains at a time */                         \
              }                                                                                                     \
            }                                                                                                       \
            F32_LOAD_BIAS(hb)                                                                                       \
            settle_ops<MT>(hf);                                                                                     \
          }                                                                                                         \
        }                                                                                                           \
      } else if constexpr (S16) { /* out^T[tile] += W2[rows 16 t .., chunk] . H^T, both pixel tiles */               \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                                           \
          _Pragma("unroll") for (int n = 0; n < 2; ++n) mfma16_out<NW == 4>(oacc16[i - N1][mt][n], a, hf16[mt][n]); \
      } else { /* out^T[tile] += W2[tile rows, chunk k-step] . H^T */                                               \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) mfma32_out<NW == 4>(oacc[(i - N1) >> 1][mt], a, hf[mt][(i - N1) & 1]); \
      }                                                                                                             \
    }                                                                                                               \
  }

#ifdef FFN32_STAGGER   /* experiment: de-phase the blocks so that their epilogues (the launch's whole HBM traffic) do not coincide */
  {
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
    const unsigned long long wait_ = (unsigned long long)(((blockIdx.x >> 3) % FFN32_STAGGER_K) * (FFN32_STAGGER));
    while (__builtin_amdgcn_s_memtime() - t0_ < wait_) __builtin_amdgcn_s_sleep(32);
  }
#endif
  // PART: the partial sums of this range through a descriptor of their own (fp32: twice the byte offsets of the bf16 rows)
  const __amdgpu_buffer_rsrc_t prsrc = __builtin_amdgcn_make_buffer_rsrc(PART ? p.part + (size_t)hrange * p.M * C : nullptr, 0, PART ? act_bytes * 2u : 0u, 0x00020000);
  const uint32_t peoff = er < ERW ? (uint32_t)(er * C + ec8) * 4u : OOB;
  const uint32_t peoff_last = (er < ERW && (RP - 1) * ERW + er < 32) ? peoff : OOB;
  for (int tile = tile0; tile < ntiles; tile += PART ? ntiles : (int)gridDim.x) {
    const long mb = (long)tile * TROWS + wid * (32 * MT);
    if constexpr (STASH) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const long m_ = mb + mt * 32 + fr;
        srow[mt] = m_ < (long)p.M ? (uint32_t)m_ * (uint32_t)(C * 8) + (uint32_t)fh * 16u : 0x80000000u;
      }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if constexpr (S16) { oacc16[nt][mt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; oacc16[nt][mt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        else {
#pragma unroll
          for (int e = 0; e < 16; ++e) oacc[nt][mt][e] = 0.f;
        }
      }
    F32_STAMP(ts0)
    for (int hc = 0; hc < nch; ++hc) F32_CHUNK
    __builtin_amdgcn_sched_barrier(0);  // keep the epilogue's loads out of the chunk
    F32_STAMP(ts1)
#ifdef FFN32_ABL_EPI   /* tools/ffn32_variants.sh only: no epilogue (wrong results); the accumulators stay live through one store */
    {
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) sum += f32x4{oacc[nt][mt][0], oacc[nt][mt][5], oacc[nt][mt][10], oacc[nt][mt][15]};
      if (sum[0] == 1234.5f) *reinterpret_cast<f32x4*>(p.out) = sum;
#ifndef FFN32_ABL_XLOAD
      F32_LOAD_X(tile + (int)gridDim.x)
#endif
      __syncthreads();
      continue;
    }
#endif
    if constexpr (!PART) {
#pragma unroll
      for (int q0 = 0; q0 < (RD == NPASS * MT ? RD : RD - 1) && q0 < NPASS * MT; ++q0) F32_LOAD_RES(q0, mb)
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");  // last XDL writes of the output accumulators -> VALU reads below
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue: each wave turns 32 rows x CQ channels at a time through LDS as raw fp32 (accumulator registers 4q..4q+3 of
    // tile t are channels 32 t + 8 q + 4 fh + 0..3 of pixel fr: one 16-byte write each); on the way out a lane owns 8
    // consecutive channels of a row: bias, layer scale, residual add and the one bf16 rounding happen there, every global
    // access is 16 B of a fully used line.
    F32_STAMP(ts2)
#ifdef FFN32_STAMPS
    unsigned long long tw_ = 0;
#endif
#pragma unroll
    for (int q = 0; q < NPASS * MT; ++q) {
      const int mt = q / NPASS, pass = q % NPASS;
      if (!PART && q == F32_XPASS) F32_LOAD_X(tile + (int)gridDim.x)   // the next tile's x fragments fly from this pass on
      if constexpr (S16) {   // tile t, pixel tile n: channels 16 t + 4 fh + 0..3 of pixel 16 n + fr
#pragma unroll
        for (int tl = 0; tl < CQ / 16; ++tl)
#pragma unroll
          for (int n = 0; n < 2; ++n) *reinterpret_cast<f32x4*>(sow + n * (16 * ORB) + tl * 64) = oacc16[pass * (CQ / 16) + tl][mt][n];
      } else {
#pragma unroll
        for (int tl = 0; tl < CQ / 32; ++tl)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const f32x16& o = oacc[pass * (CQ / 32) + tl][mt];
            *reinterpret_cast<f32x4*>(sow + (tl * 32 + qd * 8) * 4) = f32x4{o[4 * qd], o[4 * qd + 1], o[4 * qd + 2], o[4 * qd + 3]};
          }
      }
      asm volatile("" ::: "memory");  // wave-local hand-over: LDS serves a wave's accesses in order
#ifdef FFN32_STAMPS
      F32_STAMP(tp1_)
      if (q == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      F32_STAMP(tp2_)
      tw_ += tp2_ - tp1_;
#endif
      const float4 b0 = *reinterpret_cast<const float4*>(sb2 + pass * CQ + ec8), b1 = *reinterpret_cast<const float4*>(sb2 + pass * CQ + ec8 + 4);
      const float4 l0 = *reinterpret_cast<const float4*>(sls + pass * CQ + ec8), l1 = *reinterpret_cast<const float4*>(sls + pass * CQ + ec8 + 4);
      const uint32_t so_ = (uint32_t)(mb + mt * 32) * (uint32_t)(C * 2) + (uint32_t)(pass * CQ * 2);   // wave-uniform part of the row address
#pragma unroll
      for (int it = 0; it < RP; ++it) {
        const char* sr = it == RP - 1 ? sor_last : sor + it * (ERW * ORB);
        const float4 y0 = *reinterpret_cast<const float4*>(sr), y1 = *reinterpret_cast<const float4*>(sr + 16);
        if constexpr (PART) {
          typedef __attribute__((ext_vector_type(4))) unsigned int pu4;
          const uint32_t po_ = (it == RP - 1 ? peoff_last : peoff) + 2u * (so_ + (uint32_t)(it * (ERW * C * 2)));
          __builtin_amdgcn_raw_buffer_store_b128(pu4{__float_as_uint(y0.x), __float_as_uint(y0.y), __float_as_uint(y0.z), __float_as_uint(y0.w)}, prsrc, po_, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(pu4{__float_as_uint(y1.x), __float_as_uint(y1.y), __float_as_uint(y1.z), __float_as_uint(y1.w)}, prsrc, po_, 16, 0);
          continue;
        }
        const uint4 r4 = rr[q % RD][it];
        uint4 o;
        o.x = pack_bf2(bf_lo(r4.x) + fmaf(l0.x, y0.x, b0.x), bf_hi(r4.x) + fmaf(l0.y, y0.y, b0.y));
        o.y = pack_bf2(bf_lo(r4.y) + fmaf(l0.z, y0.z, b0.z), bf_hi(r4.y) + fmaf(l0.w, y0.w, b0.w));
        o.z = pack_bf2(bf_lo(r4.z) + fmaf(l1.x, y1.x, b1.x), bf_hi(r4.z) + fmaf(l1.y, y1.y, b1.y));
        o.w = pack_bf2(bf_lo(r4.w) + fmaf(l1.z, y1.z, b1.z), bf_hi(r4.w) + fmaf(l1.w, y1.w, b1.w));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned int, o), orsrc,
                                               F32_ABL_NOOUT ? OOB : (it == RP - 1 ? eoff_last : eoff) + (so_ + (uint32_t)(it * (ERW * C * 2))), 0, FFN32_NT_OUT);
        if (it & 1) __builtin_amdgcn_sched_barrier(0);   // two items' operands in flight at a time
      }
      if (!PART && RD < NPASS * MT && q + RD - 1 < NPASS * MT) F32_LOAD_RES(q + RD - 1, mb)
      asm volatile("" ::: "memory");                      // the next pass's writes stay behind these reads
    }
#ifdef FFN32_STAMPS
    {
      F32_STAMP(ts3)
      if (tid == 0 && blockIdx.x < 256) {
        unsigned long long* d = g_ffn32_stamps + blockIdx.x * 8;
        if (tile == (int)blockIdx.x) { for (int z = 0; z < 8; ++z) d[z] = 0; }
        d[0] += ts1 - ts0;   // chunk loop
        d[1] += ts2 - ts1;   // residual requests + settle
        d[2] += ts3 - ts2;   // passes
        d[3] += tw_;         // of which: wait for the residual loads at the first pass
        d[4] += 1;           // tiles
      }
    }
#endif
  }
#undef F32_CHUNK
#undef F32_LOAD_BIAS
#undef F32_GELU
#undef F32_LOAD_RES
#undef F32_LOAD_X
#undef F32_FRAG
#undef F32_FRAG_16
#undef F32_FRAG_32
#undef F32_PIECE_LOAD
#undef F32_PIECE_STORE
}

int num_cus32() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) return 256;
    n = prop.multiProcessorCount;
  }
  return n;
}

// out = res + ls * (sum over ranges of part[r] + b2), 8 channels per thread, ranges in order; the same fma as the fused epilogue
__global__ __launch_bounds__(256) void ffn32_reduce_kernel(const float* __restrict__ part, int hsplit, long M, int C, const float* __restrict__ b2,
                                                            const float* __restrict__ ls, const bf16_t* res, bf16_t* out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int c8 = C >> 3;
  if (i >= M * c8) return;
  const long m = i / c8;
  const int c = (int)(i - m * c8) * 8;
  float y[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = 0; r < hsplit; ++r) {
    const float* pp = part + ((size_t)r * M + m) * C + c;
    const float4 a = *reinterpret_cast<const float4*>(pp), b = *reinterpret_cast<const float4*>(pp + 4);
    y[0] += a.x; y[1] += a.y; y[2] += a.z; y[3] += a.w; y[4] += b.x; y[5] += b.y; y[6] += b.z; y[7] += b.w;
  }
  float rv[8], o[8];
  unpack8(*reinterpret_cast<const uint4*>(res + m * C + c), rv);
#pragma unroll
  for (int e = 0; e < 8; ++e) { const float l = ls[c + e]; o[e] = rv[e] + fmaf(l, y[e], b2[c + e] * l); }
  *reinterpret_cast<uint4*>(out + m * C + c) = pack8(o);
}

template <int C, int MT, int NW>
int launch_part32(Ffn32Params p, float* part, size_t part_bytes, hipStream_t s) {
  constexpr int LDS = Ffn32Lds<C, MT, NW>::TOTAL;
  static bool attr_set = false;
  if (!attr_set) {
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&convffn32_kernel<C, MT, NW, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const long tiles = ((long)p.M + 32 * NW * MT - 1) / (32 * NW * MT);
  const int nch = p.nchunks;
  // the largest power of two that keeps >= 6 chunks (an even count) per range, ONE round of blocks (every range writes and the reduce pass re-reads
  // M x C floats: 16 x the bf16 output per range -- a second round costs more than the ranges save) and fits the scratch
  int hs = 1;
  static const int max_tiles = fv_ab_env("FASTVLA_FFN32_RANGE_TILES") ? atoi(fv_ab_env("FASTVLA_FFN32_RANGE_TILES")) : num_cus32() / 4;   // A/B (tools build)
  for (int c = 2; c <= 8; c *= 2)
    if (nch % c == 0 && (nch / c) % 2 == 0 && nch / c >= 6 && tiles * c <= num_cus32() && tiles <= max_tiles && (size_t)c * p.M * C * sizeof(float) <= part_bytes) hs = c;   // (128 tiles in 2 ranges: measured slower than one launch)
  if (hs == 1) return -1;   // not worth it: the caller takes the one-launch form
  p.part = part; p.hsplit = hs; p.nchunks = nch / hs;
  hipLaunchKernelGGL((convffn32_kernel<C, MT, NW, false, true>), dim3((unsigned)(tiles * hs)), dim3(64 * NW), LDS, s, p);
  const long n8 = (long)p.M * (C / 8);
  hipLaunchKernelGGL(ffn32_reduce_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, part, hs, (long)p.M, C, p.b2, p.ls, p.res, p.out);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

template <int C, int MT, int NW = 4, bool S16 = false, bool STASH = false>
int launch_one32(const Ffn32Params& p, hipStream_t s) {
  constexpr int LDS = Ffn32Lds<C, MT, NW>::TOTAL;
  static bool attr_set = false;
  if (!attr_set) {
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&convffn32_kernel<C, MT, NW, S16, false, STASH>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const long tiles = ((long)p.M + 32 * NW * MT - 1) / (32 * NW * MT);
  const long blocks = tiles < num_cus32() ? tiles : num_cus32();   // one persistent block per CU, tiles dealt round-robin
  hipLaunchKernelGGL((convffn32_kernel<C, MT, NW, S16, false, STASH>), dim3((unsigned)blocks), dim3(64 * NW), LDS, s, p);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

}  // namespace

bool convffn32_supported(int C, int ratio) { return ratio == 4 && (C == 96 || C == 192 || C == 384); }
#ifdef FFN32_NW192
#define FFN32_NW192_DEFAULT FFN32_NW192
#else
#define FFN32_NW192_DEFAULT 8
#endif
#if defined(FFN32_MT96) || defined(FFN32_NW96)
#define FFN32_MT96_DEFAULT 0   /* variant builds of tools/ffn32_variants.sh keep the one-launch form */
#else
#define FFN32_MT96_DEFAULT 4
#endif

#ifndef FFN32_S16_MASK   /* which widths run on v_mfma_f32_16x16x32_bf16: bit 0 C = 384, bit 1 C = 192, bit 2 C = 96 */
#define FFN32_S16_MASK 0
#endif
constexpr bool shape16(int C) { return ((FFN32_S16_MASK) >> (C == 384 ? 0 : C == 192 ? 1 : 2)) & 1; }
int convffn32_mfma_shape(int C) { return shape16(C) ? 16 : 32; }

// fc1 weight w1 [4C][C] and fc2 weight w2 [C][4C] (row-major fp32, values already bf16-representable or to be rounded by the
// caller) -> ONE stream out[4C/32][64 C]: per 32-hidden chunk the byte image the kernel's weight slot holds, in the order its
// staging reads it.  Values: W1 / 4 and 4 W2 (exact; the kernel's GELU works in y = x / 4, see the file header).  Slot image T (elements of 2 bytes): W1 part = 32 rows x C with 8-element chunk c of row r at chunk
// c ^ ((r >> SWS1) & SWM1); W2 part = C rows x 32 with the element j of lane half h in k-step s (hidden 16 s + 8 (j >> 2) + 4 h +
// (j & 3): the order in which a converted 32x32 accumulator tile is the next product's B operand) in chunk (2 s + h) ^ ((n >> 2) & 3).
// Staging moves each KB as lane l's 16 bytes -> four dwords stored 256 bytes apart, so the global image G of a KB is
// G[8 l + 2 d + b] = T[128 d + 2 l + b] (l < 64 lanes, d < 4 dwords, b < 2 elements).
void convffn32_pack(const float* w1, const float* w2, float* out, int C) {
  const int hidden = 4 * C, nch = hidden / 32, be = 64 * C;   // elements per chunk
  const int sh1 = C == 384 ? 0 : C == 192 ? 1 : 2, mask1 = C == 384 ? 15 : C == 192 ? 7 : 3;
  std::vector<float> T((size_t)be);
  for (int hc = 0; hc < nch; ++hc) {
    for (int r = 0; r < 32; ++r)
      for (int c = 0; c < C / 8; ++c)
        for (int e = 0; e < 8; ++e) T[(size_t)r * C + (size_t)(c ^ ((r >> sh1) & mask1)) * 8 + e] = 0.25f * w1[(size_t)(hc * 32 + r) * C + c * 8 + e];
    if (shape16(C)) {   // 16x16x32: one k-step; lane group g holds k slots 8 g + j = hidden 4 g + j (j < 4), 16 + 4 g + j - 4 (j >= 4)
      for (int n = 0; n < C; ++n)
        for (int g = 0; g < 4; ++g)
          for (int j = 0; j < 8; ++j)
            T[(size_t)32 * C + (size_t)n * 32 + (size_t)(g ^ ((4 - ((n >> 2) & 3)) & 3)) * 8 + j] =
                4.0f * w2[(size_t)n * hidden + hc * 32 + (j < 4 ? 4 * g + j : 16 + 4 * g + j - 4)];
    } else
    for (int n = 0; n < C; ++n)
      for (int s = 0; s < 2; ++s)
        for (int h = 0; h < 2; ++h)
          for (int j = 0; j < 8; ++j)
            T[(size_t)32 * C + (size_t)n * 32 + (size_t)((2 * s + h) ^ ((n >> 2) & 3)) * 8 + j] =
                4.0f * w2[(size_t)n * hidden + hc * 32 + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)];
    float* G = out + (size_t)hc * be;
    for (int kb = 0; kb < be / 512; ++kb)
      for (int l = 0; l < 64; ++l)
        for (int d = 0; d < 4; ++d)
          for (int b = 0; b < 2; ++b) G[kb * 512 + 8 * l + 2 * d + b] = T[(size_t)kb * 512 + 128 * d + 2 * l + b];
  }
}

bool convffn32_stash_supported(int M, int C) { return convffn32_supported(C, 4) && !shape16(C) && (size_t)M * C * 8 < ((size_t)1 << 31); }

int launch_convffn32(const bf16_t* x, const bf16_t* wq, const float* b1, const float* b2, const float* ls,
                     const bf16_t* res, bf16_t* out, int M, int C, int hidden, hipStream_t s, float* part, size_t part_bytes, bf16_t* stash_y) {
  if (!x || !wq || !b1 || !b2 || !ls || !res || !out) return fv_fail(FV_ERR_ARG, "convffn32: null pointer");
  if (M <= 0 || hidden != 4 * C || !convffn32_supported(C, 4)) return fv_fail(FV_ERR_UNSUPPORTED, "convffn32: unsupported C=%d hidden=%d", C, hidden);
  if (((uintptr_t)x | (uintptr_t)wq | (uintptr_t)b1 | (uintptr_t)b2 | (uintptr_t)ls | (uintptr_t)res | (uintptr_t)out) & 15)
    return fv_fail(FV_ERR_ARG, "convffn32: misaligned pointer");
  if (x == out) return fv_fail(FV_ERR_ARG, "convffn32: x must not alias out");
  if ((size_t)M * C * 2 >= ((size_t)1 << 31)) return fv_fail(FV_ERR_UNSUPPORTED, "convffn32: M * C * 2 must stay below 2 GiB (32-bit buffer offsets)");
  Ffn32Params p{x, wq, b1, b2, ls, res, out, M, hidden / 32};
  if (stash_y) {   // the training forward: the one-launch form, hidden activations written out on the way (never the hidden-range forms)
    if (((uintptr_t)stash_y & 15) || !convffn32_stash_supported(M, C))
      return fv_fail(FV_ERR_ARG, "convffn32: the stash must be 16-byte aligned and needs the 32x32x16 form and M * 8C below 2 GiB");
    p.sy = stash_y;
    if (C == 96) return launch_one32<96, 4, 4, false, true>(p, s);
    if (C == 192) return launch_one32<192, 1, 8, false, true>(p, s);
    return launch_one32<384, 1, 4, false, true>(p, s);
  }
  // few row tiles and scratch supplied: (tile, hidden range) blocks + a reduce pass (the default instances' tile heights)
  if (part && ((uintptr_t)part & 15) == 0 && (size_t)M * C * 4 < ((size_t)1 << 31)) {
    const int trows = C == 384 ? 128 : C == 192 ? 256 : 512;
    if ((M + trows - 1) / trows <= num_cus32() / 2) {   // (launch_part32 narrows this to a quarter of the CUs: 128 tiles in 2 ranges measured no faster)
      int rc = -1;
      if (C == 384) rc = launch_part32<384, 1, 4>(p, part, part_bytes, s);
      else if (C == 192 && FFN32_NW192_DEFAULT == 8) rc = launch_part32<192, 1, 8>(p, part, part_bytes, s);
      else if (C == 96 && FFN32_MT96_DEFAULT == 4) rc = launch_part32<96, 4, 4>(p, part, part_bytes, s);
      if (rc != -1) return rc;
    }
  }
  switch (C) {
#ifndef FFN32_MT96
#define FFN32_MT96 4
#endif
#ifndef FFN32_NW96
#define FFN32_NW96 4
#endif
    case 96:
      if constexpr (FFN32_NW96 == 8) return launch_one32<96, FFN32_MT96 <= 2 ? FFN32_MT96 : 1, 8>(p, s);
      else return launch_one32<96, FFN32_MT96, 4, shape16(96)>(p, s);
#ifndef FFN32_NW192
#define FFN32_NW192 8   /* C = 192: eight waves (two per SIMD, 32 rows each, no AGPRs): -3.5 % against four waves of 64 rows (tools/ffn_bench.py) */
#endif
    case 192:
      if constexpr (FFN32_NW192 == 8) return launch_one32<192, 1, 8, shape16(192)>(p, s);
      else return launch_one32<192, 2, 4, shape16(192)>(p, s);
    case 384: return launch_one32<384, 1, 4, shape16(384)>(p, s);
  }
  return fv_fail(FV_ERR_UNSUPPORTED, "convffn32: unsupported C=%d", C);
}

}  // namespace fv
