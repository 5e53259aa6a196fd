// gemm_bf16.hip -- out[M,N] = A[M,K] x W[N,K]^T, bf16 operands, fp32 accumulation on MFMA, fused epilogues.
//
// Used for every dense contraction of the path: all 1x1 convs of FastViT-HD (NHWC activations make a 1x1 conv a
// row-major GEMM with M = B*H*W), the MHSA qkv/proj, the mm_projector, and the Qwen2 QKV / O / gate-up / down
// projections ([site] modeling_qwen2.py:46-48,189-192).
//
// Geometry (gfx950): 128x128 block tile, BK = 64, 256 threads = 4 waves in 2(M) x 2(N), 64x64 per wave as 4x4 tiles
// of v_mfma_f32_16x16x32_bf16.  Both operands are K-contiguous, so every fragment is one ds_read_b128.  Tiles are
// register-staged (global_load_dwordx4 -> ds_write_b128) into a double-buffered, XOR-swizzled LDS image
// (chunk ^= row & 7: conflict-free for the b128 fragment reads and for the staging writes); the next tile's global
// loads are issued before the MFMAs of the current one and written after them, one barrier per K-tile.
// The epilogue goes through an fp32 LDS image of the 128x128 tile so that bias / GELU / layer-scale+residual /
// SwiGLU / fp32-residual all run on 8 contiguous columns per lane with 16-byte global accesses.
// Bounded by the MFMA roof for K >= ~512, by HBM for the K = 96..384 tower shapes (kernel table).
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace fv {
namespace {

constexpr int BN = 128, BK = 64;
constexpr int LDC = BN + 4;                       // fp32 epilogue image row stride (floats)

struct Params {
  const bf16_t* A; const bf16_t* W; const float* bias; const float* scale; const void* res; void* out;
  int M, N, K, lda, ldr, ldo, epi, tiles_n, nwg, ksplit;   // (fp16 operands are a template parameter of the kernels, not a field)
  int splits = 1, npad = 0; float* part = nullptr;   // split-K: units = tiles x splits, partials [split][M][npad]
  int lo_off = 0;                                    // SWIGLU_SPLIT: column of the lo half (N / 2 of the WHOLE problem when this launch is a column range of it)
  // gemm256 only: tiles_m > 0 = walk the tiles column-major (the row tiles of ONE weight column tile are neighbours: same XCD, same
  // time).  For few row tiles against many weight columns (the 7B decoder at M = 1024: 4 x 148) the row-major walk makes every XCD
  // stream 32 different 3.7 MB weight tiles per round and each weight tile is fetched by four XCDs: 2.2 GB per launch, 4.3 TB/s.
  int tiles_m = 0;
  // gemm256 only: group_m = 2 / 4 (dividing the row-tile count) = walk the tiles in groups of group_m tile ROWS, column-major inside a group: the 32 CUs
  // of an XCD work on ~32 consecutive tiles of the walk, which row-major are 1 row x 32 columns = 33 operand panels through that XCD's L2 and
  // grouped by 4 are 4 x 8 = 12 (a 4096^3 problem, 16 x 16 tiles, sat at 790 TF whatever the grid: L2 fill, not the K loop)
  int group_m = 0, tiles_mt = 0;
  int ldw = 0;   // gemm256 TN instance: W's row stride (elements); lda is A's
  const uint8_t* W8 = nullptr;   // ksplit == 2: fp8 copy of W (x 2^6), row stride 2K bytes
  unsigned* sat = nullptr;   // FV_EPI_SWIGLU_F16: device counter of 8-value groups clamped to the fp16 range (0 in a healthy model)
  void* stash = nullptr;     // FV_EPI_SWIGLU_SPLIT: raw gate/up accumulators, [M][N] fp32 or (stash_f16) fp16
  int stash_f16 = 0;
};

#ifdef FASTVLA_AB_SWITCHES
__device__ int g_g2_krot = 0;
#endif
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * BK + ((chunk ^ (row & 7)) << 3); }

typedef __attribute__((ext_vector_type(8))) int i32x8;
// the lo8 product step: 16 x 16 x 128 on fp8 e4m3 operands (32 bytes per lane = two 16-byte LDS pieces), the 2^-14 of the operand scales
// applied by the instruction (E8M0 scale on the first operand, 1.0 on the second)
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ i32x8 ld_f8op(const void* p0, const void* p1) {   // two ds_read_b128 into the halves of one 8-register operand
  return __builtin_shufflevector(*reinterpret_cast<const i32x4*>(p0), *reinterpret_cast<const i32x4*>(p1), 0, 1, 2, 3, 4, 5, 6, 7);
}
// Inline asm with the accumulator TIED (dst = src C): through the builtin hipcc picks the untied form, gives every product a fresh
// destination and keeps two copies of the 128 accumulators alive (420 bytes of scratch per lane in the 256-tile kernel).  The compiler
// cannot see that this is a matrix-pipe write: the kernels put their own wait states between the last lo8 product and the first
// read of an accumulator (lo8_settle).
__device__ __forceinline__ f32x4 mfma_lo8(const i32x8& a, const i32x8& b, f32x4 c) {
  const int sa = FV_LO8_MFMA_SCALE, sb = 127;
  asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
  return c;
}
// XDL write -> VALU / LDS read of the result: 18 wait states for a 16-pass MFMA (cdna4 ISA 4.5); s_nop n = n + 1 wait states
__device__ __forceinline__ void lo8_settle() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }

// one 16x16x32 product step; F16: the same 16-byte fragments hold IEEE binary16 (llm_precision = 2's gate/up and down projections)
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// BM = 128: 2x2 waves of 64x64.  BM = 64: 2x2 waves of 32x64, for grids that would otherwise leave CUs idle (the
// M = B*T = 4096 decoder GEMMs) -- twice the blocks, 3 co-resident per CU.
template <int BM, bool F16 = false, bool LO8 = false>   // LO8: the hi + lo8 instance (p.ksplit == 2); the others carry none of its code
__global__ __launch_bounds__(256, 2) void gemm_kernel(Params p) {
  constexpr int MI = BM / 32;                             // 16-row MFMA tiles per wave along M
  constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK;
  constexpr int STAGE_BYTES = 2 * (A_ELEMS + B_ELEMS) * 2, EPI_BYTES = BM * LDC * 4;
  constexpr int LDS_BYTES = STAGE_BYTES > EPI_BYTES ? STAGE_BYTES : EPI_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem);          // [2][BM*BK]
  bf16_t* sB = sA + 2 * A_ELEMS;                         // [2][BN*BK]
  float* sC = reinterpret_cast<float*>(smem);            // [BM][LDC], reused after the K loop

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid >> 1, wc = wid & 1;
  // p.splits > 1 (few-row problems: the control loop's M = 64 ... 128 decoder, the tower's last stages at B <= 4): blocks = tiles x K ranges, a block
  // leaves the raw fp32 sums of its range in p.part[range][M][npad] and splitk_reduce_* finishes the epilogue -- 7 blocks walking 152 K-tiles each
  // (the decoder's down projection at M = 64: 108 us of exposed load latency) become 266 blocks of 4
  const int unit = xcd_remap(blockIdx.x, p.nwg);
  const int tiles = p.nwg / p.splits, krange = unit / tiles, logical = unit - krange * tiles;
  const int tn = logical % p.tiles_n, tm = logical / p.tiles_n;
  const int bm = tm * BM, bn = tn * BN;

  // staging map: 4 chunks (16 B) per operand per thread; chunk id c = tid + 256 i -> row c>>3, k-chunk c&7
  const int srow = tid >> 3, skc = tid & 7;
  constexpr int NA = BM / 32;
  uint4 ra[NA], rb[4];
  // ksplit: A carries [hi | lo] halves of a split-bf16 operand side by side (2K columns); the second half of the K
  // loop re-reads the same weight columns, so out = (A_hi + A_lo) . W^T in one launch.
  // ksplit == 2 ("hi + lo8"): after the nk1 bf16 tiles of A's hi half come K / 128 fp8 tiles: 128 one-byte remainders per row from byte
  // offset 2K of A's rows against W8's rows (row stride 2K bytes) -- the same 128-byte LDS rows, swizzle and 16-byte pieces
  const int nk1 = (p.K + BK - 1) / BK;
  const int nk = LO8 ? nk1 + p.K / 128 : (p.ksplit ? 2 * nk1 : nk1);
  auto load_tile = [&](int kt) {
    const bool second = kt >= nk1;
    if (LO8 && second) {
      const size_t kb = (size_t)(kt - nk1) * 128 + skc * 16;    // byte column of the fp8 tile
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int gm = bm + srow + 32 * i;
        ra[i] = gm < p.M ? *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.A + (size_t)gm * p.lda) + 2 * (size_t)p.K + kb) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int gn = bn + srow + 32 * i;
        rb[i] = gn < p.N ? *reinterpret_cast<const uint4*>(p.W8 + (size_t)gn * 2 * p.K + kb) : make_uint4(0, 0, 0, 0);
      }
      return;
    }
    const int kw = (second ? kt - nk1 : kt) * BK + skc * 8;   // weight column
    const int ka = second ? p.K + kw : kw;                      // activation column
    const bool kin = kw < p.K;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int gm = bm + srow + 32 * i;
      ra[i] = (kin && gm < p.M) ? *reinterpret_cast<const uint4*>(p.A + (size_t)gm * p.lda + ka) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gn = bn + srow + 32 * i;
      rb[i] = (kin && gn < p.N) ? *reinterpret_cast<const uint4*>(p.W + (size_t)gn * p.K + kw) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<uint4*>(sA + buf * A_ELEMS + lds_off(srow + 32 * i, skc)) = ra[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(sB + buf * B_ELEMS + lds_off(srow + 32 * i, skc)) = rb[i];
  };

  f32x4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Staging pipeline: the registers loaded during iteration kt - 1 (tile kt + 1) are written to LDS right AFTER iteration
  // kt's barrier and immediately refilled with tile kt + 2, so a load has a whole K-tile of MFMAs to land before its store
  // needs it (stored at the end of the same iteration, as before, it had only the MFMAs of its own tile: 512 clk against
  // >1000 of memory latency).  Still one barrier per K-tile: it publishes tile kt and retires the reads of tile kt - 1,
  // whose buffer the store then overwrites.
  const int kt0 = p.splits > 1 ? krange * nk / p.splits : 0, kt1 = p.splits > 1 ? (krange + 1) * nk / p.splits : nk;
  load_tile(kt0);
  store_tile(0);
  if (kt0 + 1 < kt1) load_tile(kt0 + 1);

  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = kt0; kt < kt1; ++kt) {
    const int cur = (kt - kt0) & 1;
    __syncthreads();
    if (kt + 1 < kt1) store_tile(cur ^ 1);
    if (kt + 2 < kt1) load_tile(kt + 2);
    const bf16_t* a_base = sA + cur * A_ELEMS;
    const bf16_t* b_base = sB + cur * B_ELEMS;
    if (LO8 && kt >= nk1) {   // fp8 tile: both 16-byte pieces of a row (k-chunks fq and 4 + fq) form one 32-byte operand
      i32x8 fb8[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fb8[i] = ld_f8op(b_base + lds_off(wc * 64 + i * 16 + fr, fq), b_base + lds_off(wc * 64 + i * 16 + fr, 4 + fq));
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const i32x8 fa8 = ld_f8op(a_base + lds_off(wr * (BM / 2) + i * 16 + fr, fq), a_base + lds_off(wr * (BM / 2) + i * 16 + fr, 4 + fq));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma_lo8(fa8, fb8[j], acc[i][j]);
      }
      continue;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int chunk = ks * 4 + fq;
      bf16x8 fa[MI], fb[4];
#pragma unroll
      for (int i = 0; i < MI; ++i)
        fa[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a_base + lds_off(wr * (BM / 2) + i * 16 + fr, chunk)));
#pragma unroll
      for (int i = 0; i < 4; ++i)
        fb[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(b_base + lds_off(wc * 64 + i * 16 + fr, chunk)));
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<F16>(fa[i], fb[j], acc[i][j]);
    }
  }
  if constexpr (LO8) lo8_settle();
  __syncthreads();  // the last tile's fragment reads are done: the epilogue image reuses the staging buffers

  // ---- epilogue: accumulators -> fp32 LDS image (C/D map: col = lane&15, row = (lane>>4)*4 + reg) ----
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        sC[(wr * (BM / 2) + i * 16 + fq * 4 + r) * LDC + wc * 64 + j * 16 + fr] = acc[i][j][r];
  __syncthreads();

  if (p.splits > 1) {   // raw partial sums of this K range
#pragma unroll
    for (int i = 0; i < BM / 16; ++i) {
      const int c = tid + 256 * i;
      const int row = c >> 4, cc = c & 15;
      const int gm = bm + row, gn = bn + cc * 8;
      if (gm >= p.M || gn >= p.N) continue;
      float* pp = p.part + ((size_t)krange * p.M + gm) * p.npad + gn;
      *reinterpret_cast<float4*>(pp) = *reinterpret_cast<const float4*>(sC + row * LDC + cc * 8);
      *reinterpret_cast<float4*>(pp + 4) = *reinterpret_cast<const float4*>(sC + row * LDC + cc * 8 + 4);
    }
    return;
  }
  const int epi = p.epi;
  if (epi == FV_EPI_SWIGLU || epi == FV_EPI_SWIGLU_SPLIT || epi == FV_EPI_SWIGLU_F16) {
    // W rows are interleaved [8 gate | 8 up]: 16 accumulator columns -> 8 outputs.  SPLIT also writes the bf16
    // remainder at column offset N/2 (split-bf16 operand of the down projection: out is [M][hi N/2 | lo N/2]); F16 writes
    // silu(gate) * up / 16 as fp16 (the down projection's fp16 weights carry the 16).
    bf16_t* out = static_cast<bf16_t*>(p.out);
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) {
      const int c = tid + 256 * i;
      const int row = c >> 3, pr = c & 7;
      const int gm = bm + row, gn = bn + pr * 16;
      if (gm < p.M && gn < p.N) {
        const float* src = sC + row * LDC + pr * 16;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = silu_f(src[e]) * src[8 + e];
        if (p.stash) {
          if (p.stash_f16) {
            float gv[8], uv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { gv[e] = src[e]; uv[e] = src[8 + e]; }
            count_f16_sat8(gv, p.sat); count_f16_sat8(uv, p.sat);
            bf16_t* sp = static_cast<bf16_t*>(p.stash) + (size_t)gm * p.N + gn;
            *reinterpret_cast<uint4*>(sp) = pack8_h(gv);
            *reinterpret_cast<uint4*>(sp + 8) = pack8_h(uv);
          } else {
            float* sp = static_cast<float*>(p.stash) + (size_t)gm * p.N + gn;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(sp + 4 * q) = *reinterpret_cast<const float4*>(src + 4 * q);
          }
        }
        if (epi == FV_EPI_SWIGLU_F16) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] *= 0.0625f;
          count_f16_sat8(o, p.sat);
          *reinterpret_cast<uint4*>(out + (size_t)gm * p.ldo + (gn >> 1)) = pack8_h(o);
          continue;
        }
        const uint4 hv = pack8(o);
        *reinterpret_cast<uint4*>(out + (size_t)gm * p.ldo + (gn >> 1)) = hv;
        if (epi == FV_EPI_SWIGLU_SPLIT) {
          float h8[8], l8[8];
          unpack8(hv, h8);
#pragma unroll
          for (int e = 0; e < 8; ++e) l8[e] = o[e] - h8[e];
          if (LO8) *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(out + (size_t)gm * p.ldo + p.lo_off) + (gn >> 1)) = pack_lo8(l8);
          else *reinterpret_cast<uint4*>(out + (size_t)gm * p.ldo + p.lo_off + (gn >> 1)) = pack8(l8);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < BM / 16; ++i) {
    const int c = tid + 256 * i;
    const int row = c >> 4, cc = c & 15;
    const int gm = bm + row, gn = bn + cc * 8;
    if (gm >= p.M || gn >= p.N) continue;
    float v[8];
    {
      const float4 lo = *reinterpret_cast<const float4*>(sC + row * LDC + cc * 8);
      const float4 hi = *reinterpret_cast<const float4*>(sC + row * LDC + cc * 8 + 4);
      v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    }
    if (p.bias) {
      const float4 lo = *reinterpret_cast<const float4*>(p.bias + gn);
      const float4 hi = *reinterpret_cast<const float4*>(p.bias + gn + 4);
      v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w; v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
    }
    if (epi == FV_EPI_GELU_GRAD || epi == FV_EPI_MUL_AUX || epi == FV_EPI_MUL_GELUP || epi == FV_EPI_F16) {   // the tower backward's fp16 outputs
      if (epi == FV_EPI_GELU_GRAD) {
        float g8[8];
        gelu_and_grad8(v, g8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e]));   // the forward's hidden is a bf16 MFMA operand: the same value here
        *reinterpret_cast<uint4*>(static_cast<bf16_t*>(p.stash) + (size_t)gm * p.ldo + gn) = pack8_h(g8);
        if (!p.out) continue;
      } else if (epi == FV_EPI_MUL_AUX) {
        float a8[8];
        unpack8_h(*reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(p.res) + (size_t)gm * p.ldr + gn), a8);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= a8[e];
      } else if (epi == FV_EPI_MUL_GELUP) {   // aux = the pre-activation / 4 as the training forward's fused ConvFFN stashed it: times gelu'(a); gelu(a) leaves too
        float a8[8], h8[8];
        unpack8_h(*reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(p.res) + (size_t)gm * p.ldr + gn), a8);
        mul_gelu_grad8(v, a8, h8);
        if (p.stash) {
#pragma unroll
          for (int e = 0; e < 8; ++e) h8[e] = bf2f(f2bf(h8[e]));   // the forward's hidden is a bf16 MFMA operand: the same rounding here
          *reinterpret_cast<uint4*>(static_cast<bf16_t*>(p.stash) + (size_t)gm * p.ldo + gn) = pack8_h(h8);
        }
      }
      count_f16_sat8(v, p.sat);
      *reinterpret_cast<uint4*>(static_cast<bf16_t*>(p.out) + (size_t)gm * p.ldo + gn) = pack8_h(v);
      continue;
    }
    if (epi == FV_EPI_BIAS_GELU) {  // packed, transcendental-free form (common.h): half the VALU issues of gelu_f
      f32x2 g[4] = {{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}};
      gelu2_n<4>(g);
      v[0] = g[0].x; v[1] = g[0].y; v[2] = g[1].x; v[3] = g[1].y; v[4] = g[2].x; v[5] = g[2].y; v[6] = g[3].x; v[7] = g[3].y;
    } else if (epi == FV_EPI_LS_RES) {
      float r[8], sc[8];
      unpack8(*reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(p.res) + (size_t)gm * p.ldr + gn), r);
      const float4 lo = *reinterpret_cast<const float4*>(p.scale + gn);
      const float4 hi = *reinterpret_cast<const float4*>(p.scale + gn + 4);
      sc[0] = lo.x; sc[1] = lo.y; sc[2] = lo.z; sc[3] = lo.w; sc[4] = hi.x; sc[5] = hi.y; sc[6] = hi.z; sc[7] = hi.w;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = r[e] + sc[e] * v[e];
    }
    if (epi == FV_EPI_RES_F32 || epi == FV_EPI_F32) {
      float* out = static_cast<float*>(p.out) + (size_t)gm * p.ldo + gn;
      if (epi == FV_EPI_RES_F32) {
        const float* rp = static_cast<const float*>(p.res) + (size_t)gm * p.ldr + gn;
        const float4 lo = *reinterpret_cast<const float4*>(rp);
        const float4 hi = *reinterpret_cast<const float4*>(rp + 4);
        v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w; v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
      }
      *reinterpret_cast<float4*>(out) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(out + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      *reinterpret_cast<uint4*>(static_cast<bf16_t*>(p.out) + (size_t)gm * p.ldo + gn) = pack8(v);
    }
  }
}


// ------------------------------------------------------------------------------------------------ 256 x 256 tile variant
// For the tower's large, well-shaped GEMMs (M, N multiples of 256, K of 64; bias / bias+GELU / layer-scale+residual).  What
// the 128-tile kernel above pays per MFMA -- a 16-byte LDS store and half a fragment read -- this one does not:
//   * both operands go global -> LDS by `global_load_lds_dwordx4` (no staging registers, no ds_write); the LDS image stays
//     lane-linear as that instruction requires and the XOR swizzle (chunk ^= row & 7) is applied to the per-lane SOURCE
//     address, the same involution the fragment reads apply;
//   * 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave: 24 ds_read_b128 per 64 MFMAs instead of 16 per 32;
//   * the next K-tile's 64 KB is in flight during all 64 MFMAs of the current one; one barrier per K-tile.
// Operands are swapped (A = weight rows, B = activation rows) so a lane's four accumulators are four consecutive output
// columns: the epilogue turns 16 rows x 64 columns through LDS with 16-byte writes and leaves as full 128-byte lines.
#ifndef G2_DEAL
#define G2_DEAL 0   // 1: deal the LDS-DMA pieces between the MFMA steps (same K-tile time: the issue cost moves, it does not hide; tools/gemm_stamps.py)
#endif
#ifdef G2_STAMPS
__device__ unsigned long long g_g2_stamps[256 * 8];   // diagnostics (tools/gemm_stamps.py): per block, clocks of wave 0 summed over K-tiles
#define G2_T(I) { const unsigned long long n_ = __builtin_readcyclecounter(); st_[I] += n_ - t0_; t0_ = n_; }
#else
#define G2_T(I)
#endif
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((ext_vector_type(4))) short g2_s16x4;
// TN fragment of k-step ks (32 deep) for the 16-column tile c16 of a region laid out in 1 KB pieces [k / 8][cols / 64], a piece = [column-block
// pair][k-row][64 bytes]: the two transposed reads take k-rows 32 ks + 4 fq + (0..3) and 32 ks + 16 + 4 fq + (0..3) (piece rows 4 ks + (fq >> 1)
// and + 2).  (64-byte source runs per lane quad: the first layout, 32-byte blocks, was conflict-free in LDS and 24 % slower than the NT kernel --
// half-line requests on the L2 -> LDS path; here the two 16-lane groups of a half share banks, 2 extra LDS cycles per read.)
__device__ __forceinline__ bf16x8 g2_tr_frag(const char* region, int cols16, int c16, int ks, uint32_t tr_lane, int fq_hi) {
  const char* p0 = region + ((((4 * ks + fq_hi) * (cols16 >> 2) + (c16 >> 2)) << 10) + (((c16 >> 1) & 1) << 9) + ((c16 & 1) << 5) + tr_lane);
  const g2_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g2_s16x4 __attribute__((address_space(3)))*)(p0));
  const g2_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((g2_s16x4 __attribute__((address_space(3)))*)(p0 + ((2 * (cols16 >> 2)) << 10)));
  const uint2 lu = __builtin_bit_cast(uint2, lo), hu = __builtin_bit_cast(uint2, hi);
  return __builtin_bit_cast(bf16x8, make_uint4(lu.x, lu.y, hu.x, hu.y));
}

// MI = 16-row tiles per wave along M (the block has 2 waves along M), WN = waves along N (64 columns each):
// <8, 4> = 256 x 256 with 8 waves (one block per CU), <4, 2> = 128 x 128 with 4 waves (two per CU) for problems with too few
// 256-tiles to fill the chip (the decoder's N = 896 / 1152 projections at M = 4096).
// ASYM (<8, 4, true>): only the wr == 0 waves -- one per SIMD -- issue the LDS-DMA, their own 1 KB pieces and those of the wave four
// above them (same lanes, 32 rows further: a scalar offset for A; W rows are clamped per row, hence offw2).  The wr == 1 waves go
// straight to their fragment reads and MFMAs and run ahead.  Measured (tools/gemm_shapes.py, A/B in one session): +6 % on the decoder's
// M = 4096 gate/up and split-K down projections, +-2 % on the tower's shapes, -15 % at 8192^3 (the staging waves' 16 pieces become the
// critical path of a long K loop with many tiles per CU) -- launch_gemm uses it for M <= 8192 only.
// TN (<8, 4, false, F16, false, true>): both operands are row-major over the CONTRACTION index (A [K][M], W [K][N]: a weight gradient's dY and X as
// they are produced, no transposed copies).  A K-tile's 64 x 256 slab lands in LDS as [k/8][m/16] blocks of 8 k-rows x 16 columns (32 bytes
// per row: the per-lane source address of the lane-linear LDS-DMA picks the block's bytes), and a fragment is two ds_read_b64_tr_b16 -- the
// 16 lanes of group fq address 4 k-rows x 16 columns and each receives its column's 4 values -- giving k-slots (fq, e) <-> k = 4 fq + e (e < 4),
// 16 + 4 fq + e - 4 (e >= 4) of a 32-deep step on BOTH operands (attention32_kernel's V recipe).
template <int MI, int WN, bool ASYM = false, bool F16 = false, bool LO8 = false, bool TN = false>   // LO8: the hi + lo8 instance (p.ksplit == 2)
__global__ __launch_bounds__(128 * WN, WN == 4 ? 1 : 2) void gemm256_kernel(Params p) {   // <8,4> 256x256, <4,2> 128x128
  static_assert(!ASYM || WN == 4, "asymmetric staging pairs wave w with wave w + 4");
  static_assert(!TN || (!ASYM && !LO8 && MI == 8 && WN == 4), "the TN instance is the plain 256 x 256 one");
  constexpr int BMT = 32 * MI, BNT = 64 * WN, NTH = 128 * WN, BUFB = (BMT + BNT) * 128, AB = BMT * 128;
  constexpr int SA = BMT * 8 / NTH, SW = BNT * 8 / NTH;   // 16-byte staging slots per thread: activations / weights
  static_assert(SA * NTH == BMT * 8 && SW * NTH == BNT * 8 && SA <= 4 && SW <= 4, "whole slots per thread");
  extern __shared__ __attribute__((aligned(16))) char g2_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wr = wid / WN, wc = wid % WN;            // wave's M half, 64-column N slice
  const int fr = lane & 15, fq = lane >> 4;
  // ksplit 1: A = [hi | lo] bf16, the weight columns are walked twice; 2: nk1 bf16 tiles of A's hi half, then K / 128 fp8 tiles (A's
  // one-byte remainders at byte offset 2K of its rows, W8's rows at the bf16 copy's row stride: the per-lane offsets serve both)
  const int nk1 = p.K / BK, nk = LO8 ? nk1 + p.K / 128 : (p.ksplit ? 2 * nk1 : nk1);
#ifdef FASTVLA_AB_SWITCHES
  // tools only (FASTVLA_GEMM_KROT=1): every block starts its K walk at a different K-tile -- does a one-round launch lose time because all
  // 256 CUs read the same k columns at the same moment?  (plain operands, one K range per tile only)
  const int krot = (g_g2_krot && !p.ksplit && p.splits == 1) ? (int)((blockIdx.x * 13u) % (unsigned)nk) : 0;
  auto a_boff = [&](int kt0_) { const int kt = krot ? (kt0_ + krot) % nk : kt0_; if constexpr (TN) return kt * BK * p.lda * 2; else return LO8 && kt >= nk1 ? 2 * p.K + (kt - nk1) * 128 : ((kt >= nk1 ? kt - nk1 : kt) * BK + (kt >= nk1 ? p.K : 0)) * 2; };
  auto w_boff = [&](int kt0_) { const int kt = krot ? (kt0_ + krot) % nk : kt0_; if constexpr (TN) return kt * BK * p.ldw * 2; else return LO8 && kt >= nk1 ? (kt - nk1) * 128 : (kt >= nk1 ? kt - nk1 : kt) * BK * 2; };
#else
  auto a_boff = [&](int kt) { if constexpr (TN) return kt * BK * p.lda * 2; else return LO8 && kt >= nk1 ? 2 * p.K + (kt - nk1) * 128 : ((kt >= nk1 ? kt - nk1 : kt) * BK + (kt >= nk1 ? p.K : 0)) * 2; };
  auto w_boff = [&](int kt) { if constexpr (TN) return kt * BK * p.ldw * 2; else return LO8 && kt >= nk1 ? (kt - nk1) * 128 : (kt >= nk1 ? kt - nk1 : kt) * BK * 2; };
#endif
  const int wslot = __builtin_amdgcn_readfirstlane(wid) * 1024;

  // staging: slot s = j * 512 + tid is 16 B of row s >> 3 at LDS position s & 7, filled from k-chunk (s & 7) ^ (row & 7)
  uint32_t offa[4], offw[4], offa_n[4], offw_n[4];
  uint32_t offw2[4] = {0u, 0u, 0u, 0u}, offw2_n[4] = {0u, 0u, 0u, 0u};   // ASYM: the partner wave's W rows
  const int a32 = 32 * p.lda * 2;
  // a unit = (output tile, K split): split-K (p.splits > 1) gives problems with few output tiles and a long K loop -- the
  // decoder's down projection at M = 4096 -- one unit per CU; each unit leaves raw fp32 partial sums for splitk_reduce_kernel
  auto tile_offsets = [&](int unit, uint32_t (&oa)[4], uint32_t (&ow)[4], uint32_t (&ow2)[4], int& bm, int& bn) {
    const int logical = xcd_remap(unit, p.nwg) / p.splits;
    if (p.tiles_m) {
      bn = (logical / p.tiles_m) * BNT;
      bm = (logical % p.tiles_m) * BMT;
    } else if (p.group_m) {   // group_m = 2 or 4 and divides the row-tile count (launch side): one division, as the row-major walk has
      const int sh = p.group_m >> 1, gsz = p.tiles_n << sh, g = logical / gsz, r = logical - g * gsz;   // (>> 1 of 2 / 4 = its log2)
      bm = ((g << sh) + (r & (p.group_m - 1))) * BMT;
      bn = (r >> sh) * BNT;
    } else {
      bm = (logical / p.tiles_n) * BMT;
      bn = (logical % p.tiles_n) * BNT;
    }
    if constexpr (TN) {
      // slot sl = 16 bytes = 8 columns of ONE k-row.  A 1 KB piece (one wave-instruction of the LDS-DMA) = 8 k-rows x 64 columns; its quad t = (sl & 63) >> 2
      // (four consecutive lanes: 64 contiguous source bytes, one request) holds row t & 7, column-block pair t >> 3; columns past the edge repeat
      // the last 8 (the epilogue drops them)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sl = j * NTH + tid, piece = sl >> 6, t = (sl & 63) >> 2, w8 = sl & 3;
        const int rr = t & 7, cpair = t >> 3;
        const int ka = 8 * (piece / (BMT / 64)) + rr, ma = 64 * (piece % (BMT / 64)) + 32 * cpair + 8 * w8;
        const int kw = 8 * (piece / (BNT / 64)) + rr, nw = 64 * (piece % (BNT / 64)) + 32 * cpair + 8 * w8;
        oa[j] = j < SA ? (uint32_t)(((size_t)ka * p.lda + min(bm + ma, p.M - 8)) * 2) : 0u;
        ow[j] = j < SW ? (uint32_t)(((size_t)kw * p.ldw + min(bn + nw, p.N - 8)) * 2) : 0u;
      }
    } else
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int sl = j * NTH + tid, row = sl >> 3, chunk = (sl & 7) ^ (row & 7);
      oa[j] = j < SA ? (uint32_t)(((size_t)min(bm + row, p.M - 1) * p.lda + chunk * 8) * 2) : 0u;   // rows past M (ragged last row tile, fp32 epilogues) repeat the last
      ow[j] = j < SW ? (uint32_t)(((size_t)min(bn + row, p.N - 1) * p.K + chunk * 8) * 2) : 0u;   // rows past N (padded last tile) repeat the last
      if (ASYM) ow2[j] = j < SW ? (uint32_t)(((size_t)min(bn + row + 32, p.N - 1) * p.K + chunk * 8) * 2) : 0u;
    }
  };
  // LDS-DMA through buffer descriptors (buffer_load_dwordx4 ... lds): per-lane 32-bit byte offset in a VGPR, the K-tile's offset
  // in an SGPR.  Beside MFMAs a wave pays ~31 clk of issue for such a piece where the global_load_lds form (64-bit per-lane
  // address) pays ~52 (tools/stage_micro.hip); launch_gemm keeps operands of 4 GiB or more away from this kernel.
  const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.A), 0, 0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc_bf = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.W), 0, 0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrsrc_f8 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(LO8 ? p.W8 : reinterpret_cast<const uint8_t*>(p.W)), 0, 0xffffffffu, 0x00020000);
  auto stage = [&](const uint32_t (&oa)[4], const uint32_t (&ow)[4], const uint32_t (&ow2)[4], int kt, int buf) {
    const int aoff = a_boff(kt), woff = w_boff(kt);
    const __amdgpu_buffer_rsrc_t wrsrc = LO8 && kt >= nk1 ? wrsrc_f8 : wrsrc_bf;
    char* la = g2_smem + buf * BUFB + wslot;
    char* lw = la + AB;
    if constexpr (ASYM) {
      if (wr == 0) {   // wave-uniform
#pragma unroll
        for (int j = 0; j < SA; ++j) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_ptr_t)(la + j * (NTH * 16)), 16, oa[j], aoff, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_ptr_t)(la + j * (NTH * 16) + 4096), 16, oa[j], aoff + a32, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < SW; ++j) {
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(lw + j * (NTH * 16)), 16, ow[j], woff, 0, 0);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(lw + j * (NTH * 16) + 4096), 16, ow2[j], woff, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < SA; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_ptr_t)(la + j * (NTH * 16)), 16, oa[j], aoff, 0, 0);
#pragma unroll
      for (int j = 0; j < SW; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(lw + j * (NTH * 16)), 16, ow[j], woff, 0, 0);
    }
  };
  // TN: the lane's byte offset inside a block's half for ds_read_b64_tr_b16: k-row 4 (fq & 1) + (fr >> 2) of the block, columns 4 (fr & 3) ..
  const uint32_t tr_lane = (uint32_t)((4 * (fq & 1) + (fr >> 2)) * 64 + (fr & 3) * 8);
  // lane's fragment offsets: row fr of a 16-row tile, swizzled chunk per k-step
  const uint32_t fo0 = (uint32_t)(fr * 128 + (((0 * 4 + fq) ^ (fr & 7)) << 4));
  const uint32_t fo1 = (uint32_t)(fr * 128 + (((1 * 4 + fq) ^ (fr & 7)) << 4));

  // Persistent: one block per CU walks the tiles; the first K-tile of the NEXT output tile is staged under the last K-tile
  // of this one, so only the very first tile of a block waits for memory with nothing to do, and the epilogue's stores
  // drain under the next tile's MFMAs.  pb = LDS buffer holding the current tile's K-tile 0.
  int tile = blockIdx.x, pb = 0, bm, bn, bm_n = 0, bn_n = 0;
  if (tile >= p.nwg) return;
  tile_offsets(tile, offa, offw, offw2, bm, bn);
  auto k_lo = [&](int unit) { return (xcd_remap(unit, p.nwg) % p.splits) * nk / p.splits; };
  auto k_hi = [&](int unit) { return (xcd_remap(unit, p.nwg) % p.splits + 1) * nk / p.splits; };
  stage(offa, offw, offw2, k_lo(tile), 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef G2_STAMPS
  unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_readcyclecounter();
#endif
  for (; tile < p.nwg; tile += (int)gridDim.x) {
    const int next = tile + (int)gridDim.x;
    if (next < p.nwg) tile_offsets(next, offa_n, offw_n, offw2_n, bm_n, bn_n);
    const int kt0 = k_lo(tile), kt1 = k_hi(tile), kn0 = next < p.nwg ? k_lo(next) : 0;
    f32x4 acc[4][MI];   // [n tile][m tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the K-tile step as a generic lambda so that the bf16 tiles and (LO8) the fp8 tiles run in TWO loops: inside one loop the two
    // fragment sets shared a live range and the 256-tile instance spilled 400 bytes per lane
    auto ktile = [&](auto f8_tag, int kt) {
      constexpr bool F8T = decltype(f8_tag)::value;
        const int cur = (pb + kt - kt0) & 1;
        G2_T(4)
        // the next K-tile's SA + SW LDS-DMA pieces are dealt one per MFMA step below (G2_DEAL): issued in a burst here, before any
        // MFMA of this K-tile, they cost every wave ~470 clk with the matrix pipe idle (tools/gemm_stamps.py)
        const bool st_same = kt + 1 < kt1, st_any = st_same || next < p.nwg;
  #if G2_DEAL
        const int st_kt = st_same ? kt + 1 : kn0;
        const int st_aoff = a_boff(st_kt), st_woff = w_boff(st_kt);
        const __amdgpu_buffer_rsrc_t wrsrc = LO8 && st_kt >= nk1 ? wrsrc_f8 : wrsrc_bf;
        char* st_la = g2_smem + (cur ^ 1) * BUFB + wslot;
        char* st_lw = st_la + AB;
  #else
        if (st_same) stage(offa, offw, offw2, kt + 1, cur ^ 1);
        else if (st_any) stage(offa_n, offw_n, offw2_n, kn0, cur ^ 1);
  #endif
        G2_T(0)
        const char* la = g2_smem + cur * BUFB + (wr * (16 * MI)) * 128;
        const char* lw = g2_smem + cur * BUFB + AB + (wc * 64) * 128;
        // Fragment reads run two steps (8 MFMAs, ~130 clk) ahead of their use: left to itself hipcc reads each pair of
        // fragments right before the MFMAs that need them and waits out the LDS latency every 8 MFMAs.  A step = one
        // activation fragment (m tile i of k-step ks) against the k-step's four weight fragments.
        if constexpr (F8T) {
          // fp8 tile: a row's two 16-byte pieces (k-chunks fq, 4 + fq) are ONE 32-byte operand; MI steps of four 16x16x128 products, each as
          // long on the matrix pipe as the two bf16 steps it replaces, over twice the K
          i32x8 w8f[4], x8f[2];
  #define G2_RD8(BASE, T) ld_f8op((BASE) + (T) * 2048 + fo0, (BASE) + (T) * 2048 + fo1)
  #pragma unroll
          for (int j = 0; j < 4; ++j) w8f[j] = G2_RD8(lw, j);
          x8f[0] = G2_RD8(la, 0);
  #pragma unroll
          for (int t = 0; t < MI; ++t) {
            if (t + 1 < MI) x8f[(t + 1) & 1] = G2_RD8(la, t + 1);
  #if G2_DEAL
            if (t < SA + SW && st_any) {
              if (t < SA) __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_ptr_t)(st_la + (t % 4) * (NTH * 16)), 16, st_same ? offa[t % 4] : offa_n[t % 4], st_aoff, 0, 0);
              else __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(st_lw + ((t - SA) % 4) * (NTH * 16)), 16, st_same ? offw[(t - SA) % 4] : offw_n[(t - SA) % 4], st_woff, 0, 0);
            }
  #endif
  #pragma unroll
            for (int j = 0; j < 4; ++j) acc[j][t] = mfma_lo8(w8f[j], x8f[t & 1], acc[j][t]);
            __builtin_amdgcn_sched_barrier(0);
          }
  #undef G2_RD8
        } else {
          bf16x8 fwA[4], fwB[4], far[3];
  #define G2_RD_A(T) (TN ? g2_tr_frag(g2_smem + cur * BUFB, BMT / 16, wr * MI + (T) % MI, (T) / MI, tr_lane, fq >> 1) \
                         : __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(la + ((T) % MI) * 2048 + ((T) / MI ? fo1 : fo0))))
  #define G2_RD_W(KS, J) (TN ? g2_tr_frag(g2_smem + cur * BUFB + AB, BNT / 16, wc * 4 + (J), (KS), tr_lane, fq >> 1) \
                             : __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(lw + (J) * 2048 + ((KS) ? fo1 : fo0))))
  #pragma unroll
          for (int j = 0; j < 4; ++j) fwA[j] = G2_RD_W(0, j);
          far[0] = G2_RD_A(0);
          far[1] = G2_RD_A(1);
          // (s_setprio(1) around the product steps, the guide's T5: measured +-0.5 % on every shape of tools/gemm_shapes.py -- not kept)
  #pragma unroll
          for (int t = 0; t < 2 * MI; ++t) {
            if (t + 2 < 2 * MI) far[(t + 2) % 3] = G2_RD_A(t + 2);
  #if G2_DEAL
            if (t < SA + SW && st_any) {
              if (t < SA) __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_ptr_t)(st_la + (t % 4) * (NTH * 16)), 16, st_same ? offa[t % 4] : offa_n[t % 4], st_aoff, 0, 0);
              else __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_ptr_t)(st_lw + ((t - SA) % 4) * (NTH * 16)), 16, st_same ? offw[(t - SA) % 4] : offw_n[(t - SA) % 4], st_woff, 0, 0);
            }
  #endif
            if (t == MI - 3) {
  #pragma unroll
              for (int j = 0; j < 4; ++j) fwB[j] = G2_RD_W(1, j);
            }
  #pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[j][t % MI] = mfma16<F16>(t < MI ? fwA[j] : fwB[j], far[t % 3], acc[j][t % MI]);
            __builtin_amdgcn_sched_barrier(0);   // keep the reads where they are dealt (hipcc sinks them back to their use)
          }
  #undef G2_RD_A
  #undef G2_RD_W
        }
        G2_T(1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next K-tile has landed ...
        G2_T(2)
        __syncthreads();                                    // ... for everybody, and nobody still reads this one
        G2_T(3)
  #ifdef G2_STAMPS
        st_[6] += 1;
  #endif
    };
    {
      const int kmid = LO8 ? (kt1 < nk1 ? kt1 : nk1) : kt1;
      for (int kt = kt0; kt < kmid; ++kt) ktile(std::false_type{}, kt);
      if constexpr (LO8) for (int kt = kt0 > nk1 ? kt0 : nk1; kt < kt1; ++kt) ktile(std::true_type{}, kt);
    }

    // ---- epilogue: per 16-row tile, 64 columns of fp32 through the wave's 4.25 KB of the buffer the last K-tile just
    // vacated (the other one already holds the next output tile's first K-tile)
    constexpr int ORB = 64 * 4 + 16;
    if constexpr (LO8) lo8_settle();
    char* so = g2_smem + ((pb + kt1 - kt0 - 1) & 1) * BUFB + wid * (16 * ORB);
    const int c8 = lane & 7;                       // the lane's 8 output columns on the way out (same for every row tile)
    const int gn = bn + wc * 64 + c8 * 8;
    float bs[8], sc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { bs[e] = 0.f; sc[e] = 1.f; }
    const bool col_ok = gn < p.N;                  // ragged last column tile (N % 8 == 0: a lane's 8 columns are in or out together)
    if (p.bias && p.splits == 1 && col_ok) {
      const float4 lo = *reinterpret_cast<const float4*>(p.bias + gn), hi = *reinterpret_cast<const float4*>(p.bias + gn + 4);
      bs[0] = lo.x; bs[1] = lo.y; bs[2] = lo.z; bs[3] = lo.w; bs[4] = hi.x; bs[5] = hi.y; bs[6] = hi.z; bs[7] = hi.w;
    }
    if (p.epi == FV_EPI_LS_RES && col_ok) {
      const float4 lo = *reinterpret_cast<const float4*>(p.scale + gn), hi = *reinterpret_cast<const float4*>(p.scale + gn + 4);
      sc[0] = lo.x; sc[1] = lo.y; sc[2] = lo.z; sc[3] = lo.w; sc[4] = hi.x; sc[5] = hi.y; sc[6] = hi.z; sc[7] = hi.w;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(so + fr * ORB + (j * 16 + fq * 4) * 4) = acc[j][i];
      asm volatile("" ::: "memory");   // wave-local hand-over: LDS serves a wave's accesses in order
      if (p.splits > 1) {   // raw partial sums; bias, residual and the column bound are the reducer's business
        const int sp = xcd_remap(tile, p.nwg) % p.splits;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int row = it * 8 + (lane >> 3);
          const int gm = bm + wr * (16 * MI) + i * 16 + row;
          if (gm >= p.M) continue;                 // ragged last row tile
          float* pp = p.part + ((size_t)sp * p.M + gm) * p.npad + gn;
          *reinterpret_cast<float4*>(pp) = *reinterpret_cast<const float4*>(so + row * ORB + c8 * 32);
          *reinterpret_cast<float4*>(pp + 4) = *reinterpret_cast<const float4*>(so + row * ORB + c8 * 32 + 16);
        }
        asm volatile("" ::: "memory");
        continue;
      }
      if (p.epi == FV_EPI_SWIGLU_SPLIT || p.epi == FV_EPI_SWIGLU_F16) {
        // W rows are interleaved [8 gate | 8 up]: a lane takes 16 accumulator columns of one row -> 8 outputs, written as
        // the bf16 value and, N/2 columns further, its bf16 remainder (split-bf16 operand of the down projection) -- or, F16,
        // as ONE fp16 value scaled by 1/16 (fp16 operand of the down projection)
        const int row = lane >> 2, pr = lane & 3;
        const int gm = bm + wr * (16 * MI) + i * 16 + row, go = (bn + wc * 64 + pr * 16) >> 1;
        const float* src = reinterpret_cast<const float*>(so + row * ORB) + pr * 16;
        if (gm >= p.M) { asm volatile("" ::: "memory"); continue; }   // ragged last row tile (the unfrozen training path's row counts)
        float o8[8], h8[8], l8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = silu_f(src[e]) * src[8 + e];
        if (p.stash) {
          if (p.stash_f16) {
            float gv[8], uv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { gv[e] = src[e]; uv[e] = src[8 + e]; }
            count_f16_sat8(gv, p.sat); count_f16_sat8(uv, p.sat);
            bf16_t* sp = static_cast<bf16_t*>(p.stash) + (size_t)gm * p.N + (bn + wc * 64 + pr * 16);
            *reinterpret_cast<uint4*>(sp) = pack8_h(gv);
            *reinterpret_cast<uint4*>(sp + 8) = pack8_h(uv);
          } else {
            float* sp = static_cast<float*>(p.stash) + (size_t)gm * p.N + (bn + wc * 64 + pr * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(sp + 4 * q) = *reinterpret_cast<const float4*>(src + 4 * q);
          }
        }
        if (p.epi == FV_EPI_SWIGLU_F16) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] *= 0.0625f;
          count_f16_sat8(o8, p.sat);
          *reinterpret_cast<uint4*>(static_cast<bf16_t*>(p.out) + (size_t)gm * p.ldo + go) = pack8_h(o8);
          asm volatile("" ::: "memory");
          continue;
        }
        const uint4 hv = pack8(o8);
        unpack8(hv, h8);
#pragma unroll
        for (int e = 0; e < 8; ++e) l8[e] = o8[e] - h8[e];
        bf16_t* op = static_cast<bf16_t*>(p.out) + (size_t)gm * p.ldo + go;
        *reinterpret_cast<uint4*>(op) = hv;
        if (LO8) *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(static_cast<bf16_t*>(p.out) + (size_t)gm * p.ldo + p.lo_off) + go) = pack_lo8(l8);
        else *reinterpret_cast<uint4*>(op + p.lo_off) = pack8(l8);
        asm volatile("" ::: "memory");
        continue;
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int row = it * 8 + (lane >> 3);
        const int gm = bm + wr * (16 * MI) + i * 16 + row;
        if (gm >= p.M || !col_ok) continue;        // ragged edge tiles (fp32 epilogues only: launch_gemm keeps the others on whole tiles)
        const float4 y0 = *reinterpret_cast<const float4*>(so + row * ORB + c8 * 32);
        const float4 y1 = *reinterpret_cast<const float4*>(so + row * ORB + c8 * 32 + 16);
        float v[8] = {y0.x + bs[0], y0.y + bs[1], y0.z + bs[2], y0.w + bs[3], y1.x + bs[4], y1.y + bs[5], y1.z + bs[6], y1.w + bs[7]};
        if (p.epi == FV_EPI_GELU_GRAD || p.epi == FV_EPI_MUL_AUX || p.epi == FV_EPI_MUL_GELUP || p.epi == FV_EPI_F16) {   // the tower backward's fp16 outputs
          if (p.epi == FV_EPI_GELU_GRAD) {
            float g8[8];
            gelu_and_grad8(v, g8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e]));   // the forward's hidden is a bf16 MFMA operand: the same value here
            *reinterpret_cast<uint4*>(static_cast<bf16_t*>(p.stash) + (size_t)gm * p.ldo + gn) = pack8_h(g8);
            if (!p.out) continue;
          } else if (p.epi == FV_EPI_MUL_AUX) {
            float a8[8];
            unpack8_h(*reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(p.res) + (size_t)gm * p.ldr + gn), a8);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= a8[e];
          } else if (p.epi == FV_EPI_MUL_GELUP) {
            float a8[8], h8[8];
            unpack8_h(*reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(p.res) + (size_t)gm * p.ldr + gn), a8);
            mul_gelu_grad8(v, a8, h8);
            if (p.stash) {
#pragma unroll
              for (int e = 0; e < 8; ++e) h8[e] = bf2f(f2bf(h8[e]));
              *reinterpret_cast<uint4*>(static_cast<bf16_t*>(p.stash) + (size_t)gm * p.ldo + gn) = pack8_h(h8);
            }
          }
          count_f16_sat8(v, p.sat);
          *reinterpret_cast<uint4*>(static_cast<bf16_t*>(p.out) + (size_t)gm * p.ldo + gn) = pack8_h(v);
          continue;
        }
        if (p.epi == FV_EPI_BIAS_GELU) {
          f32x2 g[4] = {{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}};
          gelu2_n<4>(g);
          v[0] = g[0].x; v[1] = g[0].y; v[2] = g[1].x; v[3] = g[1].y; v[4] = g[2].x; v[5] = g[2].y; v[6] = g[3].x; v[7] = g[3].y;
        } else if (p.epi == FV_EPI_LS_RES) {
          float r[8];
          unpack8(*reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(p.res) + (size_t)gm * p.ldr + gn), r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = r[e] + sc[e] * v[e];
        }
        if (p.epi == FV_EPI_RES_F32 || p.epi == FV_EPI_F32) {   // fp32 residual stream / fp32 out (decoder)
          float* op = static_cast<float*>(p.out) + (size_t)gm * p.ldo + gn;
          if (p.epi == FV_EPI_RES_F32) {
            const float* rp = static_cast<const float*>(p.res) + (size_t)gm * p.ldr + gn;
            const float4 lo = *reinterpret_cast<const float4*>(rp), hi = *reinterpret_cast<const float4*>(rp + 4);
            v[0] += lo.x; v[1] += lo.y; v[2] += lo.z; v[3] += lo.w; v[4] += hi.x; v[5] += hi.y; v[6] += hi.z; v[7] += hi.w;
          }
          *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
        } else {
          *reinterpret_cast<uint4*>(static_cast<bf16_t*>(p.out) + (size_t)gm * p.ldo + gn) = pack8(v);
        }
      }
      asm volatile("" ::: "memory");   // the next tile's writes stay behind these reads
    }
    __syncthreads();   // the scratch buffer becomes a staging target again in the next tile's first iteration
    G2_T(5)
    pb = (pb + kt1 - kt0) & 1;
    bm = bm_n; bn = bn_n;
#pragma unroll
    for (int j = 0; j < 4; ++j) { offa[j] = offa_n[j]; offw[j] = offw_n[j]; offw2[j] = offw2_n[j]; }
  }
#ifdef G2_STAMPS
  if (tid == 0 && blockIdx.x < 256) for (int z = 0; z < 8; ++z) g_g2_stamps[blockIdx.x * 8 + z] = st_[z];
#endif
}

// out[m][n] = (res ? res[m][n] : 0) + (bias ? bias[n] : 0) + sum over splits of part[s][m][n], 4 columns per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int splits, int M, int N, int npad,
                                                             const float* __restrict__ bias, const float* res, int ldr, float* out,
                                                             int ldo) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int n4 = N >> 2;
  if (i >= (long)M * n4) return;
  const int m = (int)(i / n4), n = (int)(i % n4) * 4;
  float4 acc = res ? *reinterpret_cast<const float4*>(res + (size_t)m * ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + n); acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w; }
  for (int s = 0; s < splits; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)s * M + m) * npad + n);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  *reinterpret_cast<float4*>(out + (size_t)m * ldo + n) = acc;
}

// The same reduction with the NEXT layer's RMSNorm fused in ([site] modeling_qwen2.py:247-252): one wave per row sums the partial
// slabs (+ bias, + fp32 residual), writes the residual stream and, knowing the whole row, its normalised bf16 hi (+ lo) GEMM
// operand -- the decoder's down projection hands the next layer's input_layernorm output over without another launch.
__global__ __launch_bounds__(256) void splitk_reduce_norm_kernel(const float* __restrict__ part, int splits, int M, int N, int npad,
                                                                  const float* __restrict__ bias, const float* res, int ldr, float* out,
                                                                  int ldo, const float* __restrict__ nw, bf16_t* __restrict__ y,
                                                                  bf16_t* __restrict__ ylo, int ldy, float eps, int lo8) {
  const int lane = threadIdx.x & 63;
  const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float ss = 0.f;
  for (int n = lane * 4; n < N; n += 256) {
    float4 acc = res ? *reinterpret_cast<const float4*>(res + (size_t)m * ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + n); acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w; }
    for (int s = 0; s < splits; ++s) {
      const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)s * M + m) * npad + n);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)m * ldo + n) = acc;
    ss += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
  }
  const float r = rsqrtf(wave_sum(ss) / (float)N + eps);
  for (int n = lane * 4; n < N; n += 256) {   // the lane re-reads the four values it has just written
    const float4 a = *reinterpret_cast<const float4*>(out + (size_t)m * ldo + n), w = *reinterpret_cast<const float4*>(nw + n);
    const float o0 = w.x * (a.x * r), o1 = w.y * (a.y * r), o2 = w.z * (a.z * r), o3 = w.w * (a.w * r);
    uint2 hv;
    hv.x = pack_bf2(o0, o1); hv.y = pack_bf2(o2, o3);
    *reinterpret_cast<uint2*>(y + (size_t)m * ldy + n) = hv;
    if (ylo && lo8) {
      *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(ylo + (size_t)m * ldy) + n) =
          pack_f8x4((o0 - bf_lo(hv.x)) * FV_LO8_SCALE, (o1 - bf_hi(hv.x)) * FV_LO8_SCALE, (o2 - bf_lo(hv.y)) * FV_LO8_SCALE, (o3 - bf_hi(hv.y)) * FV_LO8_SCALE);
    } else if (ylo) {
      uint2 lv;
      lv.x = pack_bf2(o0 - bf_lo(hv.x), o1 - bf_hi(hv.x)); lv.y = pack_bf2(o2 - bf_lo(hv.y), o3 - bf_hi(hv.y));
      *reinterpret_cast<uint2*>(ylo + (size_t)m * ldy + n) = lv;
    }
  }
}

// The same for few rows and many K ranges (the control loop's M = 64: 38 ranges of the down projection): one BLOCK per row, a thread owns four columns
// (+ 1024 j), the ranges' loads are independent and issued eight at a time (the one-wave-per-row form above walks 38 dependent iterations: 47 us per
// launch at M = 64); sums in range order, sum of squares by a fixed tree: bit-repeatable
__global__ __launch_bounds__(256) void splitk_reduce_norm_row_kernel(const float* __restrict__ part, int splits, int M, int N, int npad,
                                                                      const float* __restrict__ bias, const float* res, int ldr, float* out,
                                                                      int ldo, const float* __restrict__ nw, bf16_t* __restrict__ y,
                                                                      bf16_t* __restrict__ ylo, int ldy, float eps) {
  __shared__ float wsum[4];
  const int m = blockIdx.x, tid = threadIdx.x;
  float4 keep[4];   // N <= 4096
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = tid * 4 + 1024 * j;
    if (n >= N) break;
    float4 acc = res ? *reinterpret_cast<const float4*>(res + (size_t)m * ldr + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) { const float4 b = *reinterpret_cast<const float4*>(bias + n); acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w; }
    const float* pp = part + (size_t)m * npad + n;
    const size_t stride = (size_t)M * npad;
    int s0 = 0;
    for (; s0 + 8 <= splits; s0 += 8) {
      float4 v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = *reinterpret_cast<const float4*>(pp + (size_t)(s0 + e) * stride);
#pragma unroll
      for (int e = 0; e < 8; ++e) { acc.x += v[e].x; acc.y += v[e].y; acc.z += v[e].z; acc.w += v[e].w; }
    }
    for (; s0 < splits; ++s0) {
      const float4 v = *reinterpret_cast<const float4*>(pp + (size_t)s0 * stride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)m * ldo + n) = acc;
    keep[j] = acc;
    ss += acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) wsum[tid >> 6] = ss;
  __syncthreads();
  const float r = rsqrtf(((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) / (float)N + eps);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = tid * 4 + 1024 * j;
    if (n >= N) break;
    const float4 a = keep[j], w = *reinterpret_cast<const float4*>(nw + n);
    const float o0 = w.x * (a.x * r), o1 = w.y * (a.y * r), o2 = w.z * (a.z * r), o3 = w.w * (a.w * r);
    uint2 hv;
    hv.x = pack_bf2(o0, o1); hv.y = pack_bf2(o2, o3);
    *reinterpret_cast<uint2*>(y + (size_t)m * ldy + n) = hv;
    if (ylo) {
      uint2 lv;
      lv.x = pack_bf2(o0 - bf_lo(hv.x), o1 - bf_hi(hv.x)); lv.y = pack_bf2(o2 - bf_lo(hv.y), o3 - bf_hi(hv.y));
      *reinterpret_cast<uint2*>(ylo + (size_t)m * ldy + n) = lv;
    }
  }
}

// bf16-output epilogues behind K ranges (the tower's last stages and the projector at B <= 4: 48 .. 192 tiles of 64 x 128 walking 48 .. 96 K-tiles each):
// out = epi(sum of ranges + bias), epi in {bias, bias + GELU, res + scale * ( . )} with the one-launch epilogue's own arithmetic; 8 columns per thread
__global__ __launch_bounds__(256) void splitk_reduce_bf16_kernel(const float* __restrict__ part, int splits, int M, int N, int npad, const float* __restrict__ bias,
                                                                  int epi, const float* __restrict__ scale, const bf16_t* res, int ldr, bf16_t* out, int ldo) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int n8 = N >> 3;
  if (i >= (long)M * n8) return;
  const int m = (int)(i / n8), n = (int)(i % n8) * 8;
  float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = 0; s < splits; ++s) {
    const float* pp = part + ((size_t)s * M + m) * npad + n;
    const float4 a = *reinterpret_cast<const float4*>(pp), b = *reinterpret_cast<const float4*>(pp + 4);
    v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
  }
  if (bias) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += bias[n + e];
  }
  if (epi == FV_EPI_BIAS_GELU) {
    f32x2 g[4] = {{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}};
    gelu2_n<4>(g);
    v[0] = g[0].x; v[1] = g[0].y; v[2] = g[1].x; v[3] = g[1].y; v[4] = g[2].x; v[5] = g[2].y; v[6] = g[3].x; v[7] = g[3].y;
  } else if (epi == FV_EPI_LS_RES) {
    float r[8];
    unpack8(*reinterpret_cast<const uint4*>(res + (size_t)m * ldr + n), r);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = r[e] + scale[n + e] * v[e];
  }
  *reinterpret_cast<uint4*>(out + (size_t)m * ldo + n) = pack8(v);
}

// The few-row split-K of gate/up: part[s][m][16 j .. 16 j + 15] = 8 gate | 8 up sums of range s -> silu(gate) * up as hi | lo bf16 halves ([M][N/2 | N/2]),
// exactly the FV_EPI_SWIGLU_SPLIT epilogue on the summed accumulators
__global__ __launch_bounds__(256) void splitk_reduce_swiglu_kernel(const float* __restrict__ part, int splits, int M, int N, int npad, bf16_t* __restrict__ out, int ldo, int lo_off) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int n16 = N >> 4;
  if (i >= (long)M * n16) return;
  const int m = (int)(i / n16), n = (int)(i % n16) * 16;
  float g[8] = {0, 0, 0, 0, 0, 0, 0, 0}, u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int s = 0; s < splits; ++s) {
    const float* pp = part + ((size_t)s * M + m) * npad + n;
    const float4 a = *reinterpret_cast<const float4*>(pp), b = *reinterpret_cast<const float4*>(pp + 4);
    const float4 c = *reinterpret_cast<const float4*>(pp + 8), d = *reinterpret_cast<const float4*>(pp + 12);
    g[0] += a.x; g[1] += a.y; g[2] += a.z; g[3] += a.w; g[4] += b.x; g[5] += b.y; g[6] += b.z; g[7] += b.w;
    u[0] += c.x; u[1] += c.y; u[2] += c.z; u[3] += c.w; u[4] += d.x; u[5] += d.y; u[6] += d.z; u[7] += d.w;
  }
  float o[8], h8[8], l8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = silu_f(g[e]) * u[e];
  const uint4 hv = pack8(o);
  unpack8(hv, h8);
#pragma unroll
  for (int e = 0; e < 8; ++e) l8[e] = o[e] - h8[e];
  *reinterpret_cast<uint4*>(out + (size_t)m * ldo + (n >> 1)) = hv;
  *reinterpret_cast<uint4*>(out + (size_t)m * ldo + lo_off + (n >> 1)) = pack8(l8);
}

// ---- pointwise conv with a small square weight (K = N = C in {96, 192}: the stem's third conv and the first PatchEmbed
// 1x1), 0.9 ms of HBM-bound work per step that the tiled kernels ran at 2.3-3.1 TB/s: with a K loop of two or three tiles a
// block is mostly prologue and epilogue.  Here the weight sits in LDS for the life of a persistent block, a wave streams
// its 32 rows straight from global memory into MFMA B operands (no LDS, no barrier; the next tile's rows are in flight
// during the products), and the bf16 results leave through a wave-private LDS stage as one contiguous 32 x 2C-byte run.
template <int C>
__global__ __launch_bounds__(256, C == 96 ? 3 : 1) void pwconv_kernel(Params p) {
  constexpr int KS = C / 32, NT = C / 16, WP = C * 2 + 16, OP = C * 2 + 16, CH = C / 8;
  extern __shared__ __attribute__((aligned(16))) char pw_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  char* sW = pw_smem;
  char* sO = pw_smem + C * WP + wid * (32 * OP);
  for (int c = tid; c < C * CH; c += 256) {
    const int n = c / CH, ch = c % CH;
    *reinterpret_cast<uint4*>(sW + n * WP + ch * 16) = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(p.W) + (size_t)n * C + ch * 8);
  }
  float4 bv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bv[j] = p.bias ? *reinterpret_cast<const float4*>(p.bias + j * 16 + fq * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  __syncthreads();

  const int ntiles = p.M >> 7;
  const bf16_t* A = static_cast<const bf16_t*>(p.A);
  bf16_t* Y = static_cast<bf16_t*>(p.out);
  uint4 xa[2][KS], xn[2][KS];
  auto fetch = [&](int tile, uint4 (&d)[2][KS]) {
    const bf16_t* ap = A + ((size_t)tile * 128 + wid * 32 + fr) * C + fq * 8;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) d[mt][ks] = *reinterpret_cast<const uint4*>(ap + (size_t)mt * 16 * C + ks * 32);
  };
  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile, xn);
  for (; tile < ntiles; tile += (int)gridDim.x) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xa[mt][ks] = xn[mt][ks];
    if (tile + (int)gridDim.x < ntiles) fetch(tile + (int)gridDim.x, xn);
#pragma unroll
    for (int jp = 0; jp < NT / 2; ++jp) {
      f32x4 acc[2][2];   // [n tile of the pair][m tile]
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const char* wr = sW + ((2 * jp + jj) * 16 + fr) * WP + fq * 16;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) acc[jj][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const bf16x8 wf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wr + ks * 64));
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            acc[jj][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, __builtin_bit_cast(bf16x8, xa[mt][ks]), acc[jj][mt], 0, 0, 0);
        }
      }
      // D: column = lane & 15 = row of x, rows 4 fq + r = output channels: four channels of one pixel per lane
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const float4 b0 = bv[2 * jp], b1 = bv[2 * jp + 1];
        f32x2 g[4] = {{acc[0][mt][0] + b0.x, acc[0][mt][1] + b0.y}, {acc[0][mt][2] + b0.z, acc[0][mt][3] + b0.w},
                      {acc[1][mt][0] + b1.x, acc[1][mt][1] + b1.y}, {acc[1][mt][2] + b1.z, acc[1][mt][3] + b1.w}};
        if (p.epi == FV_EPI_BIAS_GELU) gelu2_n<4>(g);
        uint2 o0, o1;
        o0.x = pack_bf2(g[0].x, g[0].y); o0.y = pack_bf2(g[1].x, g[1].y);
        o1.x = pack_bf2(g[2].x, g[2].y); o1.y = pack_bf2(g[3].x, g[3].y);
        char* dst = sO + (mt * 16 + fr) * OP + (2 * jp * 16 + fq * 4) * 2;
        *reinterpret_cast<uint2*>(dst) = o0;
        *reinterpret_cast<uint2*>(dst + 32) = o1;
      }
    }
    asm volatile("" ::: "memory");   // wave-local hand-over: LDS serves a wave's accesses in order
    bf16_t* yp = Y + ((size_t)tile * 128 + wid * 32) * C;
#pragma unroll
    for (int t = 0; t < 32 * CH / 64; ++t) {
      const int q = lane + 64 * t, row = q / CH, ch = q % CH;
      *reinterpret_cast<uint4*>(yp + (size_t)row * C + ch * 8) = *reinterpret_cast<const uint4*>(sO + row * OP + ch * 16);
    }
    asm volatile("" ::: "memory");   // the next tile's stage writes stay behind these reads
  }
}

// which glds kernel (0 = none, 256 or 128) takes the problem.  256-tiles when there are enough of them to keep one block per
// CU busy; otherwise 128-tiles (two blocks per CU) if the shape allows.
int gemm_glds_tile(const GemmArgs& a) {
  static const bool off = fv_ab_env("FASTVLA_NO_GEMM256") != nullptr;
  if (off || a.K % 64 || a.K < 128) return 0;
  // gemm256_kernel keeps per-lane source byte offsets in 32 bits ((row * lda) * 2 for A, (row * K) * 2 for W): operands of
  // 4 GiB or more go to the register-staged kernel, which addresses with size_t
  if ((size_t)a.M * a.lda * 2 >= ((size_t)1 << 32) || (size_t)a.N * a.K * 2 >= ((size_t)1 << 32)) return 0;
  const bool f32 = a.epi == FV_EPI_RES_F32 || a.epi == FV_EPI_F32;
  const bool f16_epi = a.epi == FV_EPI_GELU_GRAD || a.epi == FV_EPI_MUL_AUX || a.epi == FV_EPI_MUL_GELUP || a.epi == FV_EPI_F16;   // the tower backward's (2-byte fp16 outputs: the bf16 epilogues' store loop)
  if (a.epi != FV_EPI_BIAS && a.epi != FV_EPI_BIAS_GELU && a.epi != FV_EPI_LS_RES && a.epi != FV_EPI_SWIGLU_SPLIT && a.epi != FV_EPI_SWIGLU_F16 && !f32 && !f16_epi) return 0;
  if ((a.epi == FV_EPI_SWIGLU_SPLIT || a.epi == FV_EPI_SWIGLU_F16) && a.bias) return 0;
  // (a 128 x 256 variant for shapes whose last round of 256-tiles is mostly idle -- the decoder's gate/up, 608 tiles = 2.4
  // rounds -- was measured slower, 184 vs 150 us: 64 x 64 per wave reads a third more LDS per MFMA)
  // few rows (M <= 2048, the column-major walk): the alternative is the register-staged kernel on 64- / 128-row tiles, and a half-filled
  // round of 256-tiles beats it -- 7B gate/up at M = 512 (296 tiles) 10.5 -> 8.9 ms per step, 0.5B at M = 1024 (152 tiles) 0.96 -> 0.71
  // round 6 experiment (tools build only, FASTVLA_F16_TILE128=1): the tower backward's short-K fp16 GEMMs (K = C <= 512: six K-tiles, then an epilogue that is
  // ~45 % of a 256 x 256 tile's time with ONE block per CU to hide its LDS turn and dependent aux loads) on 128 x 128 tiles at TWO blocks per CU, so that one
  // block's epilogue runs under the other's K loop
  static const bool f16_tile128 = fv_ab_env("FASTVLA_F16_TILE128") != nullptr;
  if (f16_tile128 && a.f16 && f16_epi && !a.tn && a.K <= 512 && a.M % 128 == 0 && a.N % 128 == 0 && (long)(a.M / 128) * (a.N / 128) >= 512) return 128;
  constexpr int min_tiles_small_m = 128;
  if (a.M % 256 == 0 && a.N % 256 == 0 && (long)(a.M / 256) * (a.N / 256) >= (a.M <= 2048 ? min_tiles_small_m : 320)) return 256;
  // fp32 epilogues (the decoder's projections and every dgrad / wgrad of the unfrozen training path: N = 896, 1152, 4864, M = 896 ...)
  // take RAGGED edge tiles on the 256-tile kernel -- staging clamps rows past M / N, the epilogue drops them -- when the edge waste is
  // small: the register-staged kernel they fell to runs at ~0.45 PF against ~0.9 here
  static const bool no_ragged = fv_ab_env("FASTVLA_NO_GEMM_RAGGED") != nullptr;   // A/B
  // (late round 4: the bf16 epilogues too -- bias / bias + GELU / layer-scale + residual share the guarded store loop: the PatchEmbed 1x1 at
  // 262144 x 384 x 384 and the projector's first Linear, N = 896, were the last two launches on the register-staged kernel)
  const bool bf16_epi = a.epi == FV_EPI_BIAS || a.epi == FV_EPI_BIAS_GELU || a.epi == FV_EPI_LS_RES || f16_epi;
  if (!no_ragged && (((f32 || bf16_epi) && a.N % 8 == 0) || (a.epi == FV_EPI_SWIGLU_SPLIT && a.N % 256 == 0))) {   // (SwiGLU: ragged rows only)
    const long tm = (a.M + 255) / 256, tn = (a.N + 255) / 256;
    const double fill = (double)a.M * a.N / ((double)tm * tn * 65536.0);
    if (tm * tn >= 128 && fill >= 0.75) return 256;
  }
  // 128-tiles: one wave per SIMD and a K-tile of 32 MFMAs per wave cannot cover a memory latency per K-tile, so a long K
  // loop (the decoder's down projection, K = 2 x 4864) is slower here than on the 128-tile register-staged kernel at three
  // blocks per CU; short ones (qkv / o, K <= 1024) are on par
  if (a.f16) return 0;   // fp16 operands: the 256-tile kernel or the register-staged one (the 128-tile instance below is the short-K backward's only)
  if (a.M % 128 == 0 && a.N % 128 == 0 && (long)(a.M / 128) * (a.N / 128) >= 128 && a.K >= 512 && a.K <= 1024) return 128;
  return 0;
}

thread_local bool g_norm_fused = false;   // set by launch_gemm_core when the split-K reducer took the RMSNorm with it

}  // namespace

static int launch_gemm_core(const GemmArgs& a, hipStream_t s);

// the second pass of every K-range form: sums the ranges' fp32 partials in range order and applies the launch's epilogue (bias / fp32 residual, + the
// NEXT RMSNorm where one is attached, SwiGLU + hi | lo split, or a bf16 epilogue).  lo8: the normalised operand's remainder leaves as fp8 (policy 5).
static int launch_splitk_reduce(const GemmArgs& a, int splits, int npad, bool few_rows, int lo8, hipStream_t s) {
  const float* res32 = a.epi == FV_EPI_RES_F32 ? static_cast<const float*>(a.res) : nullptr;
  if (a.epi == FV_EPI_SWIGLU_SPLIT) {
    const long n = (long)a.M * (a.N / 16);
    hipLaunchKernelGGL(splitk_reduce_swiglu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.splitk_ws, splits, a.M, a.N, npad, static_cast<bf16_t*>(a.out), a.ldo, a.lo_off ? a.lo_off : a.N / 2);
  } else if (a.epi == FV_EPI_BIAS || a.epi == FV_EPI_BIAS_GELU || a.epi == FV_EPI_LS_RES) {
    const long n = (long)a.M * (a.N / 8);
    hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, a.splitk_ws, splits, a.M, a.N, npad, a.bias, a.epi, a.scale,
                       static_cast<const bf16_t*>(a.res), a.ldr, static_cast<bf16_t*>(a.out), a.ldo);
  } else if (a.norm_w && (few_rows || a.few_rows) && a.N <= 4096 && !lo8) {   // a block per row (the wave-per-row form below walks splits x N / 256 dependent
                                                                              // steps: 32 us per launch at 1024 x 3584, 4.6 % of the 7B step); inference only
    g_norm_fused = true;
    hipLaunchKernelGGL(splitk_reduce_norm_row_kernel, dim3((unsigned)a.M), dim3(256), 0, s, a.splitk_ws, splits, a.M, a.N, npad, a.bias, res32, a.ldr,
                       static_cast<float*>(a.out), a.ldo, a.norm_w, a.norm_y, a.norm_ylo, a.norm_ld, a.norm_eps);
  } else if (a.norm_w) {
    g_norm_fused = true;
    hipLaunchKernelGGL(splitk_reduce_norm_kernel, dim3((unsigned)((a.M + 3) / 4)), dim3(256), 0, s, a.splitk_ws, splits, a.M, a.N, npad, a.bias, res32, a.ldr,
                       static_cast<float*>(a.out), a.ldo, a.norm_w, a.norm_y, a.norm_ylo, a.norm_ld, a.norm_eps, lo8);
  } else {
    const long quads = (long)a.M * (a.N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, a.splitk_ws, splits, a.M, a.N, npad, a.bias, res32, a.ldr,
                       static_cast<float*>(a.out), a.ldo);
  }
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_gemm(const GemmArgs& a, hipStream_t s) {
  if (a.norm_w) {   // a following RMSNorm of the fp32 output rows: fused into the split-K reducer when that path is taken
    const bool f32 = a.epi == FV_EPI_RES_F32 || a.epi == FV_EPI_F32;
    if (!f32 || !a.norm_y || a.ldo != a.N || a.norm_ld < a.N || a.N % 8) return fv_fail(FV_ERR_ARG, "gemm: fused norm needs an fp32 epilogue, ldo == N and N %% 8 == 0");
  }
  const int rc = launch_gemm_core(a, s);
  if (rc != FV_OK) return rc;
  if (a.norm_w && rc == FV_OK && !g_norm_fused)
    return launch_rmsnorm(static_cast<const float*>(a.out), a.norm_w, a.norm_y, a.norm_ylo, a.norm_ld, a.M, a.N, a.norm_eps, s, 0, nullptr, a.ksplit == 2 ? 1 : 0);
  return FV_OK;
}

// the grouped tile walk's group height: 4 or 2 tile rows, whichever divides the row-tile count (the kernel then needs one division and no
// short last group: a general group height cost gemm256_kernel 20 bytes of scratch per lane); 0 = row-major
static int pick_group_m(int tiles_m_total, int want) {
  if (want >= 4 && tiles_m_total % 4 == 0) return 4;
  if (want >= 2 && tiles_m_total % 2 == 0) return 2;
  return 0;
}

// TN problems (GemmArgs::tn): always the 256-tile kernel, ragged edges allowed, K ranges per tile from the same cost model as the NT path
static int launch_gemm_tn(const GemmArgs& a, hipStream_t s) {
  const bool f32out = a.epi == FV_EPI_RES_F32 || a.epi == FV_EPI_F32;
  if (!f32out || a.ksplit || a.norm_w || a.stash) return fv_fail(FV_ERR_UNSUPPORTED, "gemm (TN): fp32 epilogues only, no ksplit / fused norm / stash");
  if (a.K % 64 || a.M % 8 || a.N % 8 || a.lda % 8 || a.ldw % 8 || a.lda < a.M || a.ldw < a.N)
    return fv_fail(FV_ERR_ARG, "gemm (TN): K %% 64, M %% 8, N %% 8, lda >= M, ldw >= N (M=%d N=%d K=%d lda=%d ldw=%d)", a.M, a.N, a.K, a.lda, a.ldw);
  if (a.ldo < a.N || a.ldo % 4 || (a.epi == FV_EPI_RES_F32 && (!a.res || a.ldr % 4 || a.ldr < a.N))) return fv_fail(FV_ERR_ARG, "gemm (TN): bad ldo / residual");
  if (((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.out | (uintptr_t)a.res | (uintptr_t)a.bias) & 15) return fv_fail(FV_ERR_ARG, "gemm (TN): pointers must be 16-byte aligned");
  if ((size_t)a.K * a.lda * 2 >= ((size_t)1 << 31) || (size_t)a.K * a.ldw * 2 >= ((size_t)1 << 31)) return fv_fail(FV_ERR_UNSUPPORTED, "gemm (TN): operands of 2 GiB or more");
  Params p;
  p.A = a.A; p.W = a.W; p.bias = a.bias; p.scale = nullptr; p.res = a.res; p.out = a.out;
  p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldw = a.ldw; p.ldr = a.ldr; p.ldo = a.ldo; p.epi = a.epi; p.ksplit = 0;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<8, 4, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128));
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<8, 4, false, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128));
  }
  const int tmr = (a.M + 255) / 256, tn = (a.N + 255) / 256, tiles = tmr * tn, nkt = a.K / 64;
  int splits = 1;
  if (a.splitk_ws) {
    const double tile_us = (double)nkt * 64.0 * 131072.0 / 1.0e6 / 4.0, part_us = 2.0 * (double)a.M * (tn * 256) * 4.0 / 4.0e6;
    double best = 1e30;
    for (int sN = 1; sN <= 256; ++sN) {   // (up to 256 ranges: the tower's weight gradients are 1 .. 12 tiles over a contraction of 10^5 .. 10^7 pixels; the stem's is ONE tile over 8.4 M rows)
      if (sN > 1 && (nkt / sN < 16 || (size_t)sN * a.M * (tn * 256) * sizeof(float) > a.splitk_bytes)) break;
      const long units = (long)tiles * sN, rounds = (units + cus - 1) / cus;
      const double t = (double)rounds * tile_us / sN + (sN > 1 ? sN * part_us : 0.0);
      if (t < best * 0.97) { best = t; splits = sN; }
    }
  }
  p.tiles_n = tn; p.tiles_m = tmr <= 8 ? tmr : 0; p.tiles_mt = tmr; p.group_m = (!p.tiles_m && tn > 8) ? pick_group_m(tmr, 4) : 0;
  p.splits = splits; p.npad = tn * 256; p.part = splits > 1 ? a.splitk_ws : nullptr;
  p.nwg = tiles * splits;
  const int slots = cus / 8 * 8;
  const dim3 g(p.nwg < slots ? p.nwg : slots);
  if (a.f16) hipLaunchKernelGGL((gemm256_kernel<8, 4, false, true, false, true>), g, dim3(512), 2 * 512 * 128, s, p);
  else hipLaunchKernelGGL((gemm256_kernel<8, 4, false, false, false, true>), g, dim3(512), 2 * 512 * 128, s, p);
  if (splits > 1) {
    const long quads = (long)a.M * (a.N / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, a.splitk_ws, splits, a.M, a.N, p.npad, a.bias,
                       a.epi == FV_EPI_RES_F32 ? static_cast<const float*>(a.res) : nullptr, a.ldr, static_cast<float*>(a.out), a.ldo);
  }
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

static int launch_gemm_core(const GemmArgs& a, hipStream_t s) {
  g_norm_fused = false;
#ifdef FASTVLA_AB_SWITCHES
  {
    static bool done = false;
    if (!done) { done = true; const int v = fv_ab_env("FASTVLA_GEMM_KROT") ? 1 : 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_g2_krot), &v, sizeof(int)); }
  }
#endif
  if (!a.A || !a.W || (!a.out && a.epi != FV_EPI_GELU_GRAD)) return fv_fail(FV_ERR_ARG, "gemm: null operand");
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return fv_fail(FV_ERR_ARG, "gemm: empty shape M=%d N=%d K=%d", a.M, a.N, a.K);
  if (a.tn) return launch_gemm_tn(a, s);
  if (a.K % 8 || a.lda % 8 || a.N % 8) return fv_fail(FV_ERR_ARG, "gemm: K, lda, N must be multiples of 8 (K=%d lda=%d N=%d)", a.K, a.lda, a.N);
  if (a.lda < a.K) return fv_fail(FV_ERR_ARG, "gemm: lda < K");
  if (a.epi < FV_EPI_BIAS || a.epi > FV_EPI_MUL_GELUP || a.epi == 6) return fv_fail(FV_ERR_ARG, "gemm: bad epilogue %d", a.epi);
  if ((a.epi == FV_EPI_GELU_GRAD || a.epi == FV_EPI_MUL_AUX || a.epi == FV_EPI_MUL_GELUP || a.epi == FV_EPI_F16) && a.ksplit) return fv_fail(FV_ERR_ARG, "gemm: the fp16-output epilogues take no ksplit");
  if (a.epi == FV_EPI_GELU_GRAD && (!a.stash || ((uintptr_t)a.stash & 15))) return fv_fail(FV_ERR_ARG, "gemm: GELU_GRAD needs a 16-byte aligned stash (gelu' output)");
  if ((a.epi == FV_EPI_MUL_AUX || a.epi == FV_EPI_MUL_GELUP) && (!a.res || a.ldr % 8 || a.ldr < a.N)) return fv_fail(FV_ERR_ARG, "gemm: MUL_AUX / MUL_GELUP need the fp16 factor in res");
  if (a.f16 && a.ksplit) return fv_fail(FV_ERR_ARG, "gemm: fp16 operands are a single pass (no ksplit)");
  if (a.f16 && (a.epi == FV_EPI_BIAS_GELU || a.epi == FV_EPI_LS_RES || a.epi == FV_EPI_SWIGLU))
    return fv_fail(FV_ERR_UNSUPPORTED, "gemm: fp16 operands go with the BIAS / F32 / RES_F32 / SWIGLU_SPLIT / SWIGLU_F16 epilogues");
  const bool f32out = a.epi == FV_EPI_RES_F32 || a.epi == FV_EPI_F32;
  const bool swiglu = a.epi == FV_EPI_SWIGLU || a.epi == FV_EPI_SWIGLU_SPLIT || a.epi == FV_EPI_SWIGLU_F16;
  const int ncols = (a.epi == FV_EPI_SWIGLU || a.epi == FV_EPI_SWIGLU_F16) ? a.N / 2 : a.N;  // SPLIT: hi and lo halves side by side -> N columns
  if (swiglu && a.N % 16) return fv_fail(FV_ERR_ARG, "gemm: SwiGLU needs N %% 16 == 0");
  if (a.ldo < ncols || a.ldo % (f32out ? 4 : 8)) return fv_fail(FV_ERR_ARG, "gemm: bad ldo %d", a.ldo);
  if (a.epi == FV_EPI_LS_RES && (!a.res || !a.scale || a.ldr % 8 || a.ldr < a.N)) return fv_fail(FV_ERR_ARG, "gemm: LS_RES needs res/scale");
  if (a.epi == FV_EPI_RES_F32 && (!a.res || a.ldr % 4 || a.ldr < a.N)) return fv_fail(FV_ERR_ARG, "gemm: RES_F32 needs res");
  if (((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.out | (uintptr_t)a.res | (uintptr_t)a.bias | (uintptr_t)a.scale) & 15)
    return fv_fail(FV_ERR_ARG, "gemm: pointers must be 16-byte aligned");
  Params p;
  p.A = a.A; p.W = a.W; p.bias = a.bias; p.scale = a.scale; p.res = a.res; p.out = a.out; p.lo_off = a.lo_off ? a.lo_off : a.N / 2;
  p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldr = a.ldr; p.ldo = a.ldo; p.epi = a.epi;
  p.ksplit = a.ksplit == 2 ? 2 : (a.ksplit ? 1 : 0);
  p.sat = a.sat;
  p.stash = a.stash; p.stash_f16 = a.stash_f16;
  if (a.stash && a.epi == FV_EPI_MUL_GELUP) { if ((uintptr_t)a.stash & 15) return fv_fail(FV_ERR_ARG, "gemm: MUL_GELUP's gelu(a) output must be 16-byte aligned"); }
  else if (a.stash && a.epi != FV_EPI_GELU_GRAD && ((a.epi != FV_EPI_SWIGLU_SPLIT && a.epi != FV_EPI_SWIGLU_F16) || ((uintptr_t)a.stash & 15) || (a.stash_f16 && !a.sat)))
    return fv_fail(FV_ERR_ARG, "gemm: stash goes with FV_EPI_SWIGLU_SPLIT / FV_EPI_SWIGLU_F16 (16-byte aligned; the fp16 form with a saturation counter)");
  p.W8 = static_cast<const uint8_t*>(a.W8);
  if (a.ksplit == 2) {
    if (!a.W8 || a.K % 128 || a.lda * 2 < 3 * a.K || ((uintptr_t)a.W8 & 15)) return fv_fail(FV_ERR_ARG, "gemm: the hi + lo8 form needs W8, K %% 128 == 0 and lda >= 1.5 K");
    if (a.f16) return fv_fail(FV_ERR_ARG, "gemm: hi + lo8 goes with bf16 hi operands");
  } else if (a.ksplit && (a.K % BK || a.lda < 2 * a.K)) return fv_fail(FV_ERR_ARG, "gemm: ksplit needs K %% 64 == 0 and lda >= 2K");
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<8, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128));
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128));
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<8, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128));
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<8, 4, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128));
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<8, 4, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128));
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<8, 4, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128));
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<4, 2, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128));
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<4, 2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 128));
  }
  static const bool colmajor_ok = fv_ab_env("FASTVLA_NO_GEMM_COLMAJOR") == nullptr;   // A/B
  constexpr int cm_max = 8;   // (16 measured at the headline shape in round 3: gate/up -3.5 %, split-K down +9 %, step unchanged)
  static const bool no_asym = fv_ab_env("FASTVLA_NO_GEMM_ASYM") != nullptr;   // A/B
  static const int min_kt_per_range = fv_ab_env("FASTVLA_GEMM_MIN_KT") ? atoi(fv_ab_env("FASTVLA_GEMM_MIN_KT")) : 16;   // A/B
  static const int group_m_default = fv_ab_env("FASTVLA_GEMM_GROUP_M") ? atoi(fv_ab_env("FASTVLA_GEMM_GROUP_M")) : 4;   // A/B (0 = row-major walk)
  const bool asym = !no_asym && a.M <= 8192 && a.M % 256 == 0;   // the asymmetric staging addresses row + 32 from a (clamped) base row: whole row tiles only
  // The tail.  A problem of r >= 1 full rounds of 256-tiles plus a short last round (7B gate/up at M = 1024: 592 tiles = 2.31 rounds on 256 CUs, i.e. three
  // rounds of time) is cut at a column: the full rounds' column panels run as they are, the remaining panels as a K-range problem of their own (one more
  // short round + a reduce pass over those columns).  Whole column panels only; inference only (the cut depends on the row count).
  static const bool no_tail_env = fv_ab_env("FASTVLA_NO_GEMM_TAIL") != nullptr;   // A/B
  if (!no_tail_env && !a.no_tail && a.few_rows && a.splitk_ws && !a.norm_w && !a.stash && !a.f16 && a.ksplit != 2 && a.M % 256 == 0 && a.N % 256 == 0 &&
      a.K % 64 == 0 && (f32out || (a.epi == FV_EPI_SWIGLU_SPLIT && a.N % 512 == 0)) && gemm_glds_tile(a) == 256) {
    const int tmr = a.M / 256, tn = a.N / 256, tiles = tmr * tn, rem = tiles % cus, nkt = (a.ksplit ? 2 : 1) * (a.K / 64);
    if (tiles > cus && rem > 0 && 2 * rem <= cus && rem % tmr == 0 && nkt >= 32) {
      const int n0 = (tn - rem / tmr) * 256;   // columns [0, n0): the full rounds; [n0, N): the tail
      const int lo = a.lo_off ? a.lo_off : a.N / 2;
      GemmArgs a1 = a, a2 = a;
      a1.N = n0; a1.no_tail = 1; a1.lo_off = lo;
      a2.N = a.N - n0; a2.no_tail = 1; a2.lo_off = lo;
      a2.W = a.W + (size_t)n0 * a.K;
      if (a.bias) a2.bias = a.bias + n0;
      if (f32out) {
        a2.out = static_cast<float*>(a.out) + n0;
        if (a.res) a2.res = static_cast<const float*>(a.res) + n0;
      } else {
        a2.out = static_cast<bf16_t*>(a.out) + n0 / 2;
      }
      const int rc = launch_gemm(a1, s);
      return rc != FV_OK ? rc : launch_gemm(a2, s);
    }
  }
  // few rows (the control loop: M = 64 B rows of the decoder at B <= 4): 64-row tiles of the register-staged kernel cut along K until the chip is covered
  // twice, >= 4 K-tiles per range; fp32 epilogues through the same reduce kernels as the 256-tile split-K below
  static const bool no_skinny_env = fv_ab_env("FASTVLA_NO_SKINNY_SPLITK") != nullptr;   // A/B
  const bool no_skinny = no_skinny_env || !a.few_rows;
  // (the SwiGLU reducer writes its lo half as bf16: the hi + lo8 form, whose consumer reads fp8 bytes there, and the fp16 form keep the one-launch
  // epilogue -- policy 5's gate/up at 256 x 17920 x 1536 would otherwise pick 2 ranges here)
  const bool swiglu_sk = a.epi == FV_EPI_SWIGLU_SPLIT && !a.stash && a.N % 16 == 0 && a.ksplit != 2 && !a.f16;
  // (M <= 256, or a row count the 256-tile split-K below does not take whole: the spliced control loop's 320 B rows)
  if (!no_skinny && a.splitk_ws && (f32out || swiglu_sk) && (a.M <= 256 || (a.M % 256 != 0 && a.M <= 1024)) && a.N % 8 == 0 && a.ksplit != 2 && !a.f16) {
    const int tn = (a.N + BN - 1) / BN, tiles = ((a.M + 63) / 64) * tn, nkt = (a.ksplit ? 2 : 1) * ((a.K + BK - 1) / BK);
    const int npad = tn * BN;
    int splits = tiles < cus ? (2 * cus + tiles - 1) / tiles : 1;   // (gate/up at M = 256 is 304 tiles: it keeps the one-launch form)
    if (splits > nkt / 4) splits = nkt / 4;
    while (splits > 1 && (size_t)splits * a.M * npad * sizeof(float) > a.splitk_bytes) --splits;
    if (splits > 1) {
      p.tiles_n = tn; p.splits = splits; p.npad = npad; p.part = a.splitk_ws; p.nwg = tiles * splits;
      hipLaunchKernelGGL(gemm_kernel<64>, dim3(p.nwg), dim3(256), 0, s, p);
      return launch_splitk_reduce(a, splits, npad, true, 0, s);
    }
  }
  // bf16 epilogues with few 64-row tiles and a long K (scratch supplied by the inference tower only): K ranges of >= 8 K-tiles until the chip is covered twice
  if (!no_skinny && a.splitk_ws && (a.epi == FV_EPI_BIAS || a.epi == FV_EPI_BIAS_GELU || a.epi == FV_EPI_LS_RES) && !a.ksplit && !a.f16 && a.N % 8 == 0 &&
      a.K % BK == 0 && a.K >= 24 * BK) {
    const int tn = (a.N + BN - 1) / BN, tiles = ((a.M + 63) / 64) * tn, nkt = a.K / BK, npad = tn * BN;
    // (<= 3/8 of the CUs: one or two observations.  B = 4 -- configs[0], the shape the oracle checks -- keeps the kernels of the large batches, so that
    // its rows stay within one rounding of theirs: tests/test_gpu_fullsize.py, batch-row properties)
    int splits = tiles * 8 <= cus * 3 ? (2 * cus + tiles - 1) / tiles : 1;
    if (splits > nkt / 8) splits = nkt / 8;
    while (splits > 1 && (size_t)splits * a.M * npad * sizeof(float) > a.splitk_bytes) --splits;
    if (splits > 1) {
      p.tiles_n = tn; p.splits = splits; p.npad = npad; p.part = a.splitk_ws; p.nwg = tiles * splits;
      hipLaunchKernelGGL(gemm_kernel<64>, dim3(p.nwg), dim3(256), 0, s, p);
      return launch_splitk_reduce(a, splits, npad, true, 0, s);
    }
  }
  // split-K: fp32 output, few 256-tiles, long K, scratch supplied -> one (tile, K-range) unit per CU, then a reduce pass
  static const bool no_splitk = fv_ab_env("FASTVLA_NO_SPLITK") != nullptr, no_g256 = fv_ab_env("FASTVLA_NO_GEMM256") != nullptr;
  static const bool no_ragged_sk = fv_ab_env("FASTVLA_NO_GEMM_RAGGED") != nullptr;   // A/B
  if (!no_splitk && a.splitk_ws && (f32out || (swiglu_sk && a.M % 256 == 0 && a.N % 256 == 0)) && !no_g256 && (a.M % 256 == 0 || (!no_ragged_sk && a.N % 8 == 0)) && a.K % 64 == 0 && a.N % 4 == 0 &&
      (size_t)a.M * a.lda * 2 < ((size_t)1 << 32) && (size_t)a.N * a.K * 2 < ((size_t)1 << 32)) {
    const int tmr = (a.M + 255) / 256;            // a ragged last row tile: staging clamps its rows, the partial-sum stores skip them
    const int tn = (a.N + 255) / 256, tiles = tmr * tn, nkt = a.ksplit == 2 ? a.K / 64 + a.K / 128 : (a.ksplit ? 2 : 1) * (a.K / 64);
    // K ranges per tile: the count that minimises (rounds of `cus` units, each 1 / s of a tile's K loop) + the partial sums' trip through memory
    // (s x M x npad floats written and read back: ~0.5 of a 2048-deep K loop per range at 10 240 x 1024) -- 160 tiles (the unfrozen path's
    // 10 240 x 896 projections) take 3 ranges = 480 units = 1.9 rounds instead of 0.6 of one
    int splits = 1;
    static const bool old_rule = fv_ab_env("FASTVLA_SPLITK_OLD") != nullptr;   // A/B: round 3's rule (cus / tiles ranges, only below one round)
    if (old_rule) {
      splits = tiles < cus ? cus / tiles : 1;
      if (splits > 8) splits = 8;
      while (splits > 1 && nkt / splits < 16) --splits;
      while (splits > 1 && (size_t)splits * a.M * (tn * 256) * sizeof(float) > a.splitk_bytes) --splits;
    } else {
      const double tile_us = (double)nkt * 64.0 * 131072.0 / 1.0e6 / 4.0;                 // ~a 256 x 256 x (64 nkt) tile on one CU at ~4 GF/us/CU
      const double part_us = 2.0 * (double)a.M * (tn * 256) * 4.0 / 4.0e6;               // one range's partials written + read at ~4 TB/s
      double best = 1e30;
      for (int sN = 1; sN <= 8; ++sN) {
        if (sN > 1 && (nkt / sN < min_kt_per_range || (size_t)sN * a.M * (tn * 256) * sizeof(float) > a.splitk_bytes)) break;
        const long units = (long)tiles * sN, rounds = (units + cus - 1) / cus;
        const double t = (double)rounds * tile_us / sN + (sN > 1 ? sN * part_us : 0.0);
        if (t < best * 0.97) { best = t; splits = sN; }
      }
    }
    if (splits > 1) {
      p.tiles_n = tn;
      p.tiles_m = colmajor_ok && tmr <= cm_max ? tmr : 0;
      p.tiles_mt = tmr; p.group_m = (!p.tiles_m && tn > 8) ? pick_group_m(tmr, group_m_default) : 0;
      p.splits = splits; p.npad = tn * 256; p.part = a.splitk_ws;
      p.nwg = tiles * splits;
      const dim3 g2(p.nwg < cus ? p.nwg : cus / 8 * 8);
      /* (no asymmetric-staging instance of the hi + lo8 kernel: it spills) */ if (a.ksplit == 2) hipLaunchKernelGGL((gemm256_kernel<8, 4, false, false, true>), g2, dim3(512), 2 * 512 * 128, s, p);
      else if (a.f16 && asym) hipLaunchKernelGGL((gemm256_kernel<8, 4, true, true>), g2, dim3(512), 2 * 512 * 128, s, p);
      else if (a.f16) hipLaunchKernelGGL((gemm256_kernel<8, 4, false, true>), g2, dim3(512), 2 * 512 * 128, s, p);
      else if (asym) hipLaunchKernelGGL((gemm256_kernel<8, 4, true>), g2, dim3(512), 2 * 512 * 128, s, p);
      else hipLaunchKernelGGL((gemm256_kernel<8, 4>), g2, dim3(512), 2 * 512 * 128, s, p);
      return launch_splitk_reduce(a, splits, p.npad, false, a.ksplit == 2 ? 1 : 0, s);
    }
  }
  static const bool no_pw = fv_ab_env("FASTVLA_NO_PWCONV") != nullptr;
  if (!no_pw && !a.ksplit && !a.f16 && a.N == a.K && (a.K == 96 || a.K == 192) && a.lda == a.K && a.ldo == a.N && a.M % 128 == 0 &&
      (a.epi == FV_EPI_BIAS || a.epi == FV_EPI_BIAS_GELU)) {
    const int tiles = a.M / 128;
    if (a.K == 96) {
      constexpr int LDS = 96 * (96 * 2 + 16) + 4 * 32 * (96 * 2 + 16);
      static bool set96 = false;
      if (!set96) { FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_kernel<96>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); set96 = true; }
      hipLaunchKernelGGL(pwconv_kernel<96>, dim3(tiles < 3 * cus ? tiles : 3 * cus), dim3(256), LDS, s, p);
    } else {
      constexpr int LDS = 192 * (192 * 2 + 16) + 4 * 32 * (192 * 2 + 16);
      static bool set192 = false;
      if (!set192) { FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pwconv_kernel<192>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); set192 = true; }
      hipLaunchKernelGGL(pwconv_kernel<192>, dim3(tiles < cus ? tiles : cus), dim3(256), LDS, s, p);
    }
    FV_HIP_CHECK(hipGetLastError());
    return FV_OK;
  }
  if (const int gt = gemm_glds_tile(a)) {
    const int tmr = (a.M + gt - 1) / gt;        // ragged edge tiles only come back from gemm_glds_tile for fp32 epilogues
    p.tiles_n = (a.N + gt - 1) / gt;
    p.nwg = tmr * p.tiles_n;
    p.tiles_m = colmajor_ok && tmr <= cm_max ? tmr : 0;
    p.tiles_mt = tmr; p.group_m = (!p.tiles_m && p.tiles_n > 8) ? pick_group_m(tmr, group_m_default) : 0;   // (the 128-tile instance shares the kernel: two blocks per CU, 64 tiles per XCD at a time)
    int slots = (gt == 128 ? 2 * cus : cus) / 8 * 8;   // persistent; a multiple of 8 keeps the XCD remap exact
    if (const char* e = fv_ab_env("FASTVLA_GEMM_GRID")) { const int v = atoi(e); if (v >= 8 && gt == 256) slots = v / 8 * 8; }   // tools only
    const int grid = p.nwg < slots ? p.nwg : slots;
    if (gt == 256 && a.ksplit == 2) hipLaunchKernelGGL((gemm256_kernel<8, 4, false, false, true>), dim3(grid), dim3(512), 2 * 512 * 128, s, p);
    else if (a.ksplit == 2) hipLaunchKernelGGL((gemm256_kernel<4, 2, false, false, true>), dim3(grid), dim3(256), 2 * 256 * 128, s, p);
    else if (gt == 256 && a.f16 && asym) hipLaunchKernelGGL((gemm256_kernel<8, 4, true, true>), dim3(grid), dim3(512), 2 * 512 * 128, s, p);
    else if (gt == 256 && a.f16) hipLaunchKernelGGL((gemm256_kernel<8, 4, false, true>), dim3(grid), dim3(512), 2 * 512 * 128, s, p);
    else if (gt == 256 && asym) hipLaunchKernelGGL((gemm256_kernel<8, 4, true>), dim3(grid), dim3(512), 2 * 512 * 128, s, p);
    else if (gt == 256) hipLaunchKernelGGL((gemm256_kernel<8, 4>), dim3(grid), dim3(512), 2 * 512 * 128, s, p);
    else if (a.f16) hipLaunchKernelGGL((gemm256_kernel<4, 2, false, true>), dim3(grid), dim3(256), 2 * 256 * 128, s, p);
    else hipLaunchKernelGGL((gemm256_kernel<4, 2>), dim3(grid), dim3(256), 2 * 256 * 128, s, p);
    FV_HIP_CHECK(hipGetLastError());
    return FV_OK;
  }
  p.tiles_n = (a.N + BN - 1) / BN;
  // pick the row-tile height: 64-row tiles when 128-row tiles would give fewer than two blocks per CU
  const long blocks128 = (long)((a.M + 127) / 128) * p.tiles_n;
  if (blocks128 >= 512) {
    p.nwg = (int)blocks128;
    if (a.ksplit == 2) hipLaunchKernelGGL((gemm_kernel<128, false, true>), dim3(p.nwg), dim3(256), 0, s, p);
    else if (a.f16) hipLaunchKernelGGL((gemm_kernel<128, true>), dim3(p.nwg), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(gemm_kernel<128>, dim3(p.nwg), dim3(256), 0, s, p);
  } else {
    p.nwg = ((a.M + 63) / 64) * p.tiles_n;
    if (a.ksplit == 2) hipLaunchKernelGGL((gemm_kernel<64, false, true>), dim3(p.nwg), dim3(256), 0, s, p);
    else if (a.f16) hipLaunchKernelGGL((gemm_kernel<64, true>), dim3(p.nwg), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(gemm_kernel<64>, dim3(p.nwg), dim3(256), 0, s, p);
  }
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

}  // namespace fv

#ifdef G2_STAMPS
extern "C" int fv_dbg_g2_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fv::g_g2_stamps), 256 * 8 * 8); }
#endif
