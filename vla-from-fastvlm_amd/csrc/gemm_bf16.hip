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
#include "kernels.h"

namespace fv {
namespace {

constexpr int BN = 128, BK = 64;
constexpr int LDC = BN + 4;                       // fp32 epilogue image row stride (floats)

struct Params {
  const bf16_t* A; const bf16_t* W; const float* bias; const float* scale; const void* res; void* out;
  int M, N, K, lda, ldr, ldo, epi, tiles_n, nwg, ksplit;
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * BK + ((chunk ^ (row & 7)) << 3); }

// BM = 128: 2x2 waves of 64x64.  BM = 64: 2x2 waves of 32x64, for grids that would otherwise leave CUs idle (the
// M = B*T = 4096 decoder GEMMs) -- twice the blocks, 3 co-resident per CU.
template <int BM>
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
  const int logical = xcd_remap(blockIdx.x, p.nwg);
  const int tn = logical % p.tiles_n, tm = logical / p.tiles_n;
  const int bm = tm * BM, bn = tn * BN;

  // staging map: 4 chunks (16 B) per operand per thread; chunk id c = tid + 256 i -> row c>>3, k-chunk c&7
  const int srow = tid >> 3, skc = tid & 7;
  constexpr int NA = BM / 32;
  uint4 ra[NA], rb[4];
  // ksplit: A carries [hi | lo] halves of a split-bf16 operand side by side (2K columns); the second half of the K
  // loop re-reads the same weight columns, so out = (A_hi + A_lo) . W^T in one launch.
  const int nk1 = (p.K + BK - 1) / BK;
  const int nk = p.ksplit ? 2 * nk1 : nk1;
  auto load_tile = [&](int kt) {
    const bool second = kt >= nk1;
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
  load_tile(0);
  store_tile(0);
  if (nk > 1) load_tile(1);

  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    __syncthreads();
    if (kt + 1 < nk) store_tile(cur ^ 1);
    if (kt + 2 < nk) load_tile(kt + 2);
    const bf16_t* a_base = sA + cur * A_ELEMS;
    const bf16_t* b_base = sB + cur * B_ELEMS;
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
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
  }
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

  const int epi = p.epi;
  if (epi == FV_EPI_SWIGLU || epi == FV_EPI_SWIGLU_SPLIT) {
    // W rows are interleaved [8 gate | 8 up]: 16 accumulator columns -> 8 outputs.  SPLIT also writes the bf16
    // remainder at column offset N/2 (split-bf16 operand of the down projection: out is [M][hi N/2 | lo N/2]).
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
        const uint4 hv = pack8(o);
        *reinterpret_cast<uint4*>(out + (size_t)gm * p.ldo + (gn >> 1)) = hv;
        if (epi == FV_EPI_SWIGLU_SPLIT) {
          float h8[8], l8[8];
          unpack8(hv, h8);
#pragma unroll
          for (int e = 0; e < 8; ++e) l8[e] = o[e] - h8[e];
          *reinterpret_cast<uint4*>(out + (size_t)gm * p.ldo + (p.N >> 1) + (gn >> 1)) = pack8(l8);
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

}  // namespace

int launch_gemm(const GemmArgs& a, hipStream_t s) {
  if (!a.A || !a.W || !a.out) return fv_fail(FV_ERR_ARG, "gemm: null operand");
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return fv_fail(FV_ERR_ARG, "gemm: empty shape M=%d N=%d K=%d", a.M, a.N, a.K);
  if (a.K % 8 || a.lda % 8 || a.N % 8) return fv_fail(FV_ERR_ARG, "gemm: K, lda, N must be multiples of 8 (K=%d lda=%d N=%d)", a.K, a.lda, a.N);
  if (a.lda < a.K) return fv_fail(FV_ERR_ARG, "gemm: lda < K");
  if (a.epi < FV_EPI_BIAS || (a.epi > FV_EPI_F32 && a.epi != FV_EPI_SWIGLU_SPLIT)) return fv_fail(FV_ERR_ARG, "gemm: bad epilogue %d", a.epi);
  const bool f32out = a.epi == FV_EPI_RES_F32 || a.epi == FV_EPI_F32;
  const bool swiglu = a.epi == FV_EPI_SWIGLU || a.epi == FV_EPI_SWIGLU_SPLIT;
  const int ncols = a.epi == FV_EPI_SWIGLU ? a.N / 2 : a.N;  // SPLIT: hi and lo halves side by side -> N columns
  if (swiglu && a.N % 16) return fv_fail(FV_ERR_ARG, "gemm: SwiGLU needs N %% 16 == 0");
  if (a.ldo < ncols || a.ldo % (f32out ? 4 : 8)) return fv_fail(FV_ERR_ARG, "gemm: bad ldo %d", a.ldo);
  if (a.epi == FV_EPI_LS_RES && (!a.res || !a.scale || a.ldr % 8 || a.ldr < a.N)) return fv_fail(FV_ERR_ARG, "gemm: LS_RES needs res/scale");
  if (a.epi == FV_EPI_RES_F32 && (!a.res || a.ldr % 4 || a.ldr < a.N)) return fv_fail(FV_ERR_ARG, "gemm: RES_F32 needs res");
  if (((uintptr_t)a.A | (uintptr_t)a.W | (uintptr_t)a.out | (uintptr_t)a.res | (uintptr_t)a.bias | (uintptr_t)a.scale) & 15)
    return fv_fail(FV_ERR_ARG, "gemm: pointers must be 16-byte aligned");
  Params p;
  p.A = a.A; p.W = a.W; p.bias = a.bias; p.scale = a.scale; p.res = a.res; p.out = a.out;
  p.M = a.M; p.N = a.N; p.K = a.K; p.lda = a.lda; p.ldr = a.ldr; p.ldo = a.ldo; p.epi = a.epi;
  p.ksplit = a.ksplit ? 1 : 0;
  if (a.ksplit && (a.K % BK || a.lda < 2 * a.K)) return fv_fail(FV_ERR_ARG, "gemm: ksplit needs K %% 64 == 0 and lda >= 2K");
  p.tiles_n = (a.N + BN - 1) / BN;
  // pick the row-tile height: 64-row tiles when 128-row tiles would give fewer than two blocks per CU
  const long blocks128 = (long)((a.M + 127) / 128) * p.tiles_n;
  if (blocks128 >= 512) {
    p.nwg = (int)blocks128;
    hipLaunchKernelGGL(gemm_kernel<128>, dim3(p.nwg), dim3(256), 0, s, p);
  } else {
    p.nwg = ((a.M + 63) / 64) * p.tiles_n;
    hipLaunchKernelGGL(gemm_kernel<64>, dim3(p.nwg), dim3(256), 0, s, p);
  }
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

}  // namespace fv
