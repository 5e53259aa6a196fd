// tools/dw7_prologue.hip -- VERDICT r5 #1: the "dw7x7 + BN as the fused ConvFFN's prologue" prototype, C = 384.
//
// SURVEY section 7's fix for the RepMixer block's 12 B / element: the depthwise pair writes x' only, and convffn32_kernel produces t = dw7x7(x') + b
// itself, into the registers its first product reads (xf[ks] = t[pixel fr][16 ks + 8 fh .. + 8], the B operand of v_mfma_f32_32x32x16_bf16).  This file
// is THE NEW PART of that kernel on its own, under the constraints the fused kernel would give it, so that its cost and its HBM traffic can be measured
// before a kernel that is tuned to the last register is rebuilt around it:
//   * one 256-thread block per CU (4 waves, as convffn32_kernel<384,1,4>), persistent over 8 x 16-pixel tiles (= the kernel's 128 rows); an image's 32
//     tiles run on ONE XCD at the same time, so the 3-pixel halo of a tile is some neighbour's centre in that XCD's L2;
//   * LDS: 43.9 KB of Toeplitz cells + 20.5 KB for the t tile = 64.4 KB -- what the fused kernel can lend between two tiles (its epilogue staging area,
//     51.2 KB, and the idle weight slot, 49.2 KB); the other 98 KB stay with the weight stream;
//   * per 64-channel slab (6 per tile): halo tile (14 rows x 24 columns) global -> registers -> v_perm transpose -> LDS cells [row][plane][quad][channel][4 px]
//     (dwpair_march_kernel's layout), 21 x 4 x 2 v_mfma_f32_4x4x4_16B_bf16 per wave (a wave = 16 channels x 8 rows x 16 columns; the arithmetic and its order
//     are dwpair_march_kernel's 7x7 phase: bit-identical t), then t -> LDS as [pixel][64 ch] and back as the four 16-byte B fragments of the slab;
//   * PF = 1 requests slab s + 1's halo while slab s is in the MFMAs (48 registers in flight -- the fused kernel would have to find them among the 192
//     accumulators that are idle between tiles); PF = 0 waits for every slab's loads where it needs them.
// STORE = 1 writes t (from the fragments, NHWC) for the check against the CPU; timing runs write 32 bytes per block.
//
//   hipcc -O3 --offload-arch=gfx950 tools/dw7_prologue.hip -o tools/bin/dw7_prologue ; tools/dw7_prologue.sh   (GPU box)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int C = 384, TW = 16, TH = 8, NQ = 6, NTQ = 4, SLAB = 64, NSLAB = C / SLAB;
constexpr int PLB = NQ * 256, RS = 2 * PLB + 64;   // a halo row: 2 planes of 32 channels x 6 column quads x (32 ch x 4 px x 2 B) + 64 B (== 64 mod 256)
constexpr int HR = TH + 6;                         // halo rows
constexpr int CELLS = HR * RS;                     // 43 904 B
constexpr int TS = 160;                            // t tile: bytes per pixel (64 ch x 2 B + 32: a column pair lands 16 banks away)
constexpr int TBYTES = TH * TW * TS;               // 20 480 B
constexpr int LDS_USED = CELLS + TBYTES;
constexpr int LDS_ALLOC = 100 * 1024;              // one block per CU, as the fused kernel
constexpr int NTASK = HR * NQ * 8, TPT = (NTASK + 255) / 256;

__device__ __forceinline__ uint32_t pack_bf2(float a, float b) {   // round to nearest even, as common.h
  uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fffu + ((ua >> 16) & 1u);
  ub += 0x7fffu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xffff0000u);
}

template <int PF, int STORE>
__global__ __launch_bounds__(256, 1) void dw7_prologue_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ ttab, const float* __restrict__ bias,
                                                              bf16_t* __restrict__ tout, uint32_t* __restrict__ chk, int B, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;            // Toeplitz cells of the slab's halo tile
  char* sT = smem + CELLS;    // t tile [pixel][64 ch]
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int bch = lane >> 2, jr = lane & 3;                      // depthwise phase: lane = (channel of the wave's 16, row within 4)
  const int fr = lane & 31, fh = lane >> 5;                      // fragment phase: lane = (pixel of the wave's 32, k half)
  const int tiles_x = W / TW, tpi = tiles_x * (H / TH);
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
  const uint32_t rowbytes = (uint32_t)W * C * 2u, tbytes = (uint32_t)B * H * rowbytes;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(tout, 0, STORE ? tbytes : 0u, 0x00020000);
  constexpr uint32_t OOB = 0x80000000u;

  // ---- task constants: task = (halo row, column quad, 16-byte channel group) of 4 pixels x 8 channels
  uint32_t xdst[TPT];
  int trow[TPT], tcol[TPT], tcg[TPT];
#pragma unroll
  for (int tt = 0; tt < TPT; ++tt) {
    const int task = tid + 256 * tt;
    const int cg = task & 7, quad = (task >> 3) % NQ, row = (task >> 3) / NQ;
    trow[tt] = row; tcol[tt] = quad * 4 - 3; tcg[tt] = cg;
    xdst[tt] = (uint32_t)(row * RS + (cg >> 2) * PLB + quad * 256 + (cg & 3) * 64) | (uint32_t)((((quad & 3) << 1) | ((cg & 3) >> 1)) << 3);
  }
  uint32_t sw[4];   // the lane's swizzled channel offsets (one per quad & 3), plane included
  const uint32_t pl = (uint32_t)(wid >> 1) * PLB;
#pragma unroll
  for (int v = 0; v < 4; ++v) sw[v] = pl + (uint32_t)((((wid & 1) * 16 + bch) ^ ((v << 1) | (wid & 1))) << 3);

  u32x4 px[TPT][4];
  uint32_t xo[TPT][4];
#define LOAD_SLAB(S)                                                                                              \
  {                                                                                                               \
    /* every wave issues all TPT x 4 loads (idle tasks and pixels outside the map carry an offset the descriptor drops): no branch around a load, */ \
    /* so hipcc can count the loads in flight and waits for exactly the ones an instruction needs                                               */ \
    _Pragma("unroll") for (int tt = 0; tt < TPT; ++tt)                                                            \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                               \
        px[tt][j] = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xo[tt][j] == OOB ? OOB : xo[tt][j] + (uint32_t)((S) * (SLAB * 2)), 0, 0); \
  }
#define TILE_OFFSETS(IMG, TY, TX)                                                                                 \
  {                                                                                                               \
    _Pragma("unroll") for (int tt = 0; tt < TPT; ++tt) {                                                          \
      const int iy = (TY) * TH - 3 + trow[tt];                                                                    \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                             \
        const int ix = (TX) * TW + tcol[tt] + j;                                                                  \
        xo[tt][j] = (tid + 256 * tt < NTASK && iy >= 0 && iy < H && ix >= 0 && ix < W)                            \
                        ? (uint32_t)(IMG) * (uint32_t)H * rowbytes + (uint32_t)iy * rowbytes + (uint32_t)ix * (C * 2) + (uint32_t)tcg[tt] * 16u \
                        : OOB;                                                                                    \
      }                                                                                                           \
    }                                                                                                             \
  }

  uint32_t acc_chk = 0;
  int idx = slot;
  int img = xcd + 8 * (idx / tpi);
  if (img >= B) return;
  int ti = idx % tpi, ty = ti / tiles_x, tx = ti % tiles_x;
  TILE_OFFSETS(img, ty, tx)
  LOAD_SLAB(0)
  while (true) {
    // the tile that follows (its first slab is requested during this tile's last one)
    const int idx_n = idx + per_xcd, img_n = xcd + 8 * (idx_n / tpi), ti_n = idx_n % tpi, ty_n = ti_n / tiles_x, tx_n = ti_n % tiles_x;
    const bool more = img_n < B;
#pragma unroll 1
    for (int s = 0; s < NSLAB; ++s) {
      // ---- Toeplitz fragments + bias of the wave's 16 channels of this slab (L2-resident table, 21 x 8 B per lane)
      s16x4 a7[7][3];
      {
        const char* s7 = reinterpret_cast<const char*>(ttab) + (size_t)(4 * s + wid) * (7 * 3 * 512) + lane * 8;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
          for (int m = 0; m < 3; ++m) a7[ky][m] = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(s7 + (ky * 3 + m) * 512));
      }
      const float bv7 = bias[s * SLAB + wid * 16 + bch];
      if (!PF && s) LOAD_SLAB(s)
      // ---- halo slab: registers -> transposed cells
#pragma unroll
      for (int tt = 0; tt < TPT; ++tt) {
        if (tid + 256 * tt < NTASK) {
          const uint32_t d[4][4] = {{px[tt][0].x, px[tt][0].y, px[tt][0].z, px[tt][0].w}, {px[tt][1].x, px[tt][1].y, px[tt][1].z, px[tt][1].w},
                                    {px[tt][2].x, px[tt][2].y, px[tt][2].z, px[tt][2].w}, {px[tt][3].x, px[tt][3].y, px[tt][3].z, px[tt][3].w}};
#pragma unroll
          for (int dd = 0; dd < 4; ++dd) {   // dword dd of a pixel holds channels 2dd (low half) and 2dd + 1 (high half)
            uint2 ev, od;
            ev.x = __builtin_amdgcn_perm(d[1][dd], d[0][dd], 0x05040100u);
            ev.y = __builtin_amdgcn_perm(d[3][dd], d[2][dd], 0x05040100u);
            od.x = __builtin_amdgcn_perm(d[1][dd], d[0][dd], 0x07060302u);
            od.y = __builtin_amdgcn_perm(d[3][dd], d[2][dd], 0x07060302u);
            *reinterpret_cast<uint2*>(sA + (xdst[tt] ^ (uint32_t)((2 * dd) * 8))) = ev;
            *reinterpret_cast<uint2*>(sA + (xdst[tt] ^ (uint32_t)((2 * dd + 1) * 8))) = od;
          }
        }
      }
      if (PF) {   // the next slab (or the next tile's first; after the last tile: 12 dropped loads) flies while this one is in the MFMAs
        int sn = s + 1;
        if (sn == NSLAB) {
          sn = 0;
          if (more) TILE_OFFSETS(img_n, ty_n, tx_n)
          else {
#pragma unroll
            for (int tt = 0; tt < TPT; ++tt)
#pragma unroll
              for (int j = 0; j < 4; ++j) xo[tt][j] = OOB;
          }
        }
        LOAD_SLAB(sn)
      }
      __syncthreads();   // B1: the cells are in place; every wave has read the previous slab's fragments (the t tile may be rewritten)
      // ---- t = dw7x7 over the cells: the wave's 16 channels x 2 row groups x 4 column quads; order = dwpair_march_kernel's 7x7 phase
      f32x4 acc[2][NTQ];
#pragma unroll
      for (int rgi = 0; rgi < 2; ++rgi)
#pragma unroll
        for (int q = 0; q < NTQ; ++q) acc[rgi][q] = f32x4{bv7, bv7, bv7, bv7};
      {
        s16x4 xr[3][NQ];
#define RD7(R, DST)                                                                                               \
        {                                                                                                         \
          const uint32_t ro_ = (uint32_t)(((R) / 7) * 4 + jr + (R) % 7) * RS;                                     \
          _Pragma("unroll") for (int t = 0; t < NQ; ++t)                                                          \
            DST[t] = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(sA + ro_ + sw[t & 3] + t * 256));  \
        }
        RD7(0, xr[0])
        RD7(1, xr[1])
#pragma unroll
        for (int r = 0; r < 14; ++r) {
          if (r + 2 < 14) RD7(r + 2, xr[(r + 2) % 3])
#pragma unroll
          for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int q = 0; q < NTQ; ++q) acc[r / 7][q] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a7[r % 7][m], xr[r % 3][q + m], acc[r / 7][q], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
#undef RD7
      }
      // ---- t tile -> LDS as [pixel][64 ch]: a lane pair (channels bch, bch ^ 1 = lanes l, l ^ 4) trades halves so that each writes whole channel-pair dwords
      {
        const bool even = (bch & 1) == 0;
        const uint32_t cb = 32u * (uint32_t)(wid ^ jr) + 2u * (uint32_t)(bch & ~1);     // 32-byte channel blocks swizzled by the row (bank spread of the 4 rows)
#pragma unroll
        for (int rgi = 0; rgi < 2; ++rgi)
#pragma unroll
          for (int q = 0; q < NTQ; ++q) {
            const uint32_t ux = pack_bf2(acc[rgi][q][0], acc[rgi][q][1]), uy = pack_bf2(acc[rgi][q][2], acc[rgi][q][3]);   // columns (0,1), (2,3) of the lane's channel
            const uint32_t recv = (uint32_t)__builtin_amdgcn_ds_swizzle((int)(even ? uy : ux), 0x101F);                     // lane ^ 4
            const uint32_t lo = even ? ux : recv, hi = even ? recv : uy;                                                    // even: columns 0,1; odd: columns 2,3; (lo, hi) = channels (bch & ~1, + 1)
            const uint32_t p0 = (uint32_t)((rgi * 4 + jr) * TW + 4 * q + (even ? 0 : 2));
            *reinterpret_cast<uint32_t*>(sT + p0 * TS + cb) = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
            *reinterpret_cast<uint32_t*>(sT + (p0 + 1) * TS + cb) = __builtin_amdgcn_perm(hi, lo, 0x07060302u);
          }
      }
      __syncthreads();   // B3: the t tile is complete; every wave is done with the cells (the next slab may overwrite them)
      // ---- the slab's four B fragments of this wave's 32 pixels (rows 2 wid, 2 wid + 1 of the tile)
      {
        const int prow = 2 * wid + (fr >> 4), pcol = fr & 15;
        const char* tp = sT + (prow * TW + pcol) * TS + 16 * fh;
        uint4 f[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = *reinterpret_cast<const uint4*>(tp + 32 * (j ^ (prow & 3)));
#pragma unroll
        for (int j = 0; j < 4; ++j) acc_chk ^= f[j].x ^ f[j].y ^ f[j].z ^ f[j].w;
        if (STORE) {
          const uint32_t o = (uint32_t)img * (uint32_t)H * rowbytes + (uint32_t)(ty * TH + prow) * rowbytes + (uint32_t)(tx * TW + pcol) * (C * 2) +
                             (uint32_t)(s * SLAB * 2) + 16u * fh;
#pragma unroll
          for (int j = 0; j < 4; ++j) __builtin_amdgcn_raw_buffer_store_b128(u32x4{f[j].x, f[j].y, f[j].z, f[j].w}, orsrc, o + 32u * j, 0, 0);
        }
      }
    }
    if (!more) break;
    idx = idx_n; img = img_n; ty = ty_n; tx = tx_n;
    if (!PF) { TILE_OFFSETS(img, ty, tx) LOAD_SLAB(0) }
  }
  if (lane == 0) chk[blockIdx.x * 4 + wid] = acc_chk;
#undef LOAD_SLAB
#undef TILE_OFFSETS
}

// ---- host side
static float bf2f(bf16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static bf16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (bf16_t)(u >> 16); }
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float rnd() {   // ~N(0, 1): sum of 4 uniforms
  float s = 0.f;
  for (int i = 0; i < 4; ++i) { rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull; s += (float)((rng_state >> 40) & 0xffffff) / 16777216.0f; }
  return (s - 2.0f) * 1.7320508f;
}

template <int PF, int STORE>
static float run(const bf16_t* x, const bf16_t* ttab, const float* bias, bf16_t* tout, uint32_t* chk, int B, int H, int W, int reps) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dw7_prologue_kernel<PF, STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ALLOC));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((dw7_prologue_kernel<PF, STORE>), dim3(256), dim3(256), LDS_ALLOC, 0, x, ttab, bias, tout, chk, B, H, W);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipGetLastError());
  return best * 1e3f;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64, H = 64, W = 64, reps = argc > 2 ? atoi(argv[2]) : 20;
  if (B % 8 || W % TW || H % TH) { fprintf(stderr, "B %% 8, W %% 16, H %% 8\n"); return 1; }
  const size_t n = (size_t)B * H * W * C;
  std::vector<bf16_t> hx(n);
  for (size_t i = 0; i < n; ++i) hx[i] = f2bf(rnd());
  std::vector<float> w((size_t)C * 49), hb(C);
  for (auto& v : w) v = bf2f(f2bf(rnd() / 7.0f));
  for (auto& v : hb) v = 0.1f * rnd();
  // Toeplitz table [C/16][7][3][16 ch][4 i][4 kk] = w[ky][4m + kk - i] (include/fastvla_hip_testops.h, fv_op_dwconv_mfma)
  std::vector<bf16_t> ht((size_t)(C / 16) * 7 * 3 * 256);
  for (int g = 0; g < C / 16; ++g)
    for (int ky = 0; ky < 7; ++ky)
      for (int m = 0; m < 3; ++m)
        for (int ch = 0; ch < 16; ++ch)
          for (int i = 0; i < 4; ++i)
            for (int kk = 0; kk < 4; ++kk) {
              const int kx = 4 * m + kk - i;
              ht[((((size_t)g * 7 + ky) * 3 + m) * 16 + ch) * 16 + i * 4 + kk] = (kx >= 0 && kx < 7) ? f2bf(w[(size_t)(g * 16 + ch) * 49 + ky * 7 + kx]) : 0;
            }
  bf16_t *dx, *dt, *dtab; float* db; uint32_t* dchk;
  CK(hipMalloc(&dx, n * 2)); CK(hipMalloc(&dt, n * 2)); CK(hipMalloc(&dtab, ht.size() * 2)); CK(hipMalloc(&db, C * 4)); CK(hipMalloc(&dchk, 256 * 4 * 4));
  CK(hipMemcpy(dx, hx.data(), n * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dtab, ht.data(), ht.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dt, 0xff, n * 2));
  // ---- correctness: t of two whole images (the first and the last: borders, tile seams, every slab) against fp64 on the host
  run<1, 1>(dx, dtab, db, dt, dchk, B, H, W, 1);
  std::vector<bf16_t> got(n);
  CK(hipMemcpy(got.data(), dt, n * 2, hipMemcpyDeviceToHost));
  double worst = 0.0; size_t bad = 0;
  for (int b : {0, B - 1})
    for (int y = 0; y < H; ++y)
      for (int xx = 0; xx < W; ++xx)
        for (int c = 0; c < C; ++c) {
          double a = hb[c];
          for (int ky = 0; ky < 7; ++ky)
            for (int kx = 0; kx < 7; ++kx) {
              const int iy = y + ky - 3, ix = xx + kx - 3;
              if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
              a += (double)w[(size_t)c * 49 + ky * 7 + kx] * bf2f(hx[(((size_t)b * H + iy) * W + ix) * C + c]);
            }
          const double g = bf2f(got[(((size_t)b * H + y) * W + xx) * C + c]);
          const double err = fabs(g - a), tol = 0.0045 * fabs(a) + 1e-3;   // one bf16 rounding (2^-8 relative) + fp32 summation order
          if (err > worst) worst = err;
          if (err > tol) ++bad;
        }
  run<0, 1>(dx, dtab, db, dt, dchk, B, H, W, 1);
  std::vector<bf16_t> got0(n);
  CK(hipMemcpy(got0.data(), dt, n * 2, hipMemcpyDeviceToHost));
  const bool same = memcmp(got.data(), got0.data(), n * 2) == 0;
  printf("check: 2 images x %d x %d x %d outputs against fp64: worst |err| %.4f, outside one bf16 rounding: %zu; PF=0 and PF=1 outputs identical: %s\n", H, W, C, worst, bad,
         same ? "yes" : "NO");
  if (bad || !same) return 2;
  // ---- timing (B images of 64 x 64 x 384 = one C = 384 ConvFFN launch of the headline step at B = 64)
  const float t_pf0 = run<0, 0>(dx, dtab, db, dt, dchk, B, H, W, reps), t_pf1 = run<1, 0>(dx, dtab, db, dt, dchk, B, H, W, reps);
  const float t_st = run<1, 1>(dx, dtab, db, dt, dchk, B, H, W, reps);
  const double mb = n * 2 / 1e6;
  printf("B=%d: x' %.0f MB; %d tiles of 8x16 px on 256 CUs (%d per CU); LDS used %d B (cells %d + t tile %d)\n", B, mb, B * (H / TH) * (W / TW), B * (H / TH) * (W / TW) / 256, LDS_USED,
         CELLS, TBYTES);
  printf("prologue alone, loads waited for per slab (PF=0): %.1f us per launch = %.2f us per tile\n", t_pf0, t_pf0 / (B * (H / TH) * (W / TW) / 256.0));
  printf("prologue alone, next slab requested a slab ahead (PF=1): %.1f us per launch = %.2f us per tile\n", t_pf1, t_pf1 / (B * (H / TH) * (W / TW) / 256.0));
  printf("... + t written to HBM (PF=1, STORE=1: the standalone 7x7 this replaces would do that): %.1f us per launch\n", t_st);
  return 0;
}
