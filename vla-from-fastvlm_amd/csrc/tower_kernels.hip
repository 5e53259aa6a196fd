// tower_kernels.hip -- the HBM/VALU-bound pieces of the FastViT-HD tower, NHWC bf16 activations.
//   letterbox        resize_with_pad semantics (reference model/fastvlm_adapter.py:36-55) on device
//   stem conv        dense 3x3 s2, 3(+1 pad)->C0, +bias +GELU           (mci.py convolutional_stem[0])
//   depthwise conv   k in {3,7}, stride {1,2}, channel multiplier {1,2}   (RepMixer, ConvFFN.conv(+BN folded), RepCPE,
//                    PatchEmbed grouped 7x7 s2, stem dw3x3 s2, conv_exp)
//   LayerNormChannel per-pixel LN over C                                  (AttentionBlock.norm)
//   SE + GELU        conv_exp tail
// All are bounded by HBM streaming or by the fp32 VALU (49 taps/element for the 7x7), never MFMA-shaped; loads and
// stores are 8 or 16 B per lane with the channel index fastest across lanes.
#include "kernels.h"
#include <cstdlib>

namespace fv {
namespace {

// ------------------------------------------------------------------------------------------------ letterbox
struct LbParams {
  const void* img; bf16_t* pix; int dtype, B, C, Hin, Win, S, rh, rw, pt, pl; float pad, sh, sw;
};

__device__ __forceinline__ float lb_fetch(const LbParams& p, size_t plane, int y, int x) {
  const size_t i = plane + (size_t)y * p.Win + x;
  return p.dtype == FV_U8 ? (float)static_cast<const uint8_t*>(p.img)[i] : static_cast<const float*>(p.img)[i];
}

// every product and sum is rounded on its own (fp contract off: HIP's __f*_rn are plain operators and would still fuse), so that the two
// kernels that evaluate these expressions -- letterbox_kernel and lb_pixel inside the stem -- agree bit for bit whatever hipcc would have
// contracted to an fma in either context
__device__ __forceinline__ float lb_lerp(float a, float b, float w) {
#pragma clang fp contract(off)
  return (1.0f - w) * a + w * b;
}
__device__ __forceinline__ float lb_src(int d, float scale) {
#pragma clang fp contract(off)
  return fmaxf(((float)d + 0.5f) * scale - 0.5f, 0.0f);
}
__device__ __forceinline__ float lb_frac(float s, int i) {
#pragma clang fp contract(off)
  return s - (float)i;
}

// ONE letterboxed pixel (RGB + zero pad, bf16): the arithmetic of letterbox_kernel below, expression for expression, for the stem
// kernel that samples the source image itself (stem_fused_kernel<true>: the 1024^2 frame then never exists in HBM)
__device__ __forceinline__ uint2 lb_pixel(const LbParams& p, int b, int y, int x) {
  float v[3] = {p.pad, p.pad, p.pad};
  const int dx = x - p.pl, dy = y - p.pt;
  if (dx >= 0 && dx < p.rw && dy >= 0 && dy < p.rh) {
    const float sx = lb_src(dx, p.sw);
    const int x0 = min((int)sx, p.Win - 1), x1 = min(x0 + 1, p.Win - 1);
    const float wx = lb_frac(sx, x0);
    const float sy = lb_src(dy, p.sh);
    const int y0 = min((int)sy, p.Hin - 1), y1 = min(y0 + 1, p.Hin - 1);
    const float wy = lb_frac(sy, y0);
    const int nc = p.C >= 3 ? 3 : 1;
    for (int c = 0; c < nc; ++c) {
      const size_t plane = ((size_t)b * p.C + c) * p.Hin * p.Win;
      const float t0 = lb_lerp(lb_fetch(p, plane, y0, x0), lb_fetch(p, plane, y0, x1), wx);
      const float t1 = lb_lerp(lb_fetch(p, plane, y1, x0), lb_fetch(p, plane, y1, x1), wx);
      v[c] = lb_lerp(t0, t1, wy);
    }
    if (nc == 1) v[1] = v[2] = v[0];
  }
  uint2 o;
  o.x = pack_bf2(v[0], v[1]);
  o.y = pack_bf2(v[2], 0.0f);
  return o;
}

// One thread = one output column x LB_R consecutive output rows: when upscaling (the path's case: 336 -> 1024, ~3 output rows per
// source row) consecutive rows share their two source rows, so the 4 taps x 3 channels are fetched again only when y0 moves -- a
// third of the loads and of the x arithmetic of the one-pixel-per-thread form, the same values bit for bit.
constexpr int LB_R = 4;
// MODE 1 / 2: FastVLMBackbone._maybe_normalize_imagenet (model/fastvlm_adapter.py:463-477) on top of the letterbox.  The reference decides from the maximum of the
// WHOLE letterboxed batch (pad pixels included) whether the values are 0..255 (`x.max() > 1.5` -> x / 255), then applies (x - mean) / std per channel, every step
// in fp32 on the host.  Here: MODE 1 evaluates the same letterboxed values without storing them and leaves their maximum in nrm.vmax (an ordered-integer image of
// the float, atomicMax: exact and order-independent); MODE 2 reads that maximum on the device (no host synchronisation), divides by 255 if the reference would
// (IEEE division, as ATen's), subtracts and divides by the per-channel constants (IEEE, one rounding each: TF.normalize's sub_ / div_) and rounds to bf16 ONCE.
struct LbNorm { float mean[3], std[3]; unsigned* vmax; int heuristic; };
__device__ __forceinline__ unsigned lb_ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float lb_unord(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
template <int MODE>
__global__ __launch_bounds__(256) void letterbox_kernel(LbParams p, LbNorm nrm) {
  bool by255 = false;
  if constexpr (MODE == 2) by255 = nrm.heuristic && lb_unord(*nrm.vmax) > 1.5f;
  float vmax = -INFINITY;
  const int x = blockIdx.x * 256 + threadIdx.x, yb = blockIdx.y * LB_R, b = blockIdx.z;
  if (MODE != 1 && x >= p.S) return;
  const int dx = x - p.pl;
  const bool xin = dx >= 0 && dx < p.rw;
  // ATen area_pixel_compute_source_index, align_corners=False: src = max((dst+0.5)*scale-0.5, 0)
  const float sx = lb_src(dx, p.sw);
  const int x0 = min((int)sx, p.Win - 1), x1 = min(x0 + 1, p.Win - 1);
  const float wx = lb_frac(sx, x0);
  const int nc = p.C >= 3 ? 3 : 1;
  float t0[3] = {0.f, 0.f, 0.f}, t1[3] = {0.f, 0.f, 0.f};   // the two source rows, already blended along x
  int have = -1;
#pragma unroll
  for (int r = 0; r < LB_R; ++r) {
    const int y = yb + r;
    if (y >= p.S || x >= p.S) break;
    float v[3] = {p.pad, p.pad, p.pad};
    const int dy = y - p.pt;
    if (xin && dy >= 0 && dy < p.rh) {
      const float sy = lb_src(dy, p.sh);
      const int y0 = min((int)sy, p.Hin - 1), y1 = min(y0 + 1, p.Hin - 1);
      const float wy = lb_frac(sy, y0);
      if (y0 != have) {
        have = y0;
        for (int c = 0; c < nc; ++c) {
          const size_t plane = ((size_t)b * p.C + c) * p.Hin * p.Win;
          const float p00 = lb_fetch(p, plane, y0, x0), p01 = lb_fetch(p, plane, y0, x1);
          const float p10 = lb_fetch(p, plane, y1, x0), p11 = lb_fetch(p, plane, y1, x1);
          t0[c] = lb_lerp(p00, p01, wx);
          t1[c] = lb_lerp(p10, p11, wx);
        }
      }
      for (int c = 0; c < nc; ++c) v[c] = lb_lerp(t0[c], t1[c], wy);
      if (nc == 1) v[1] = v[2] = v[0];  // gray -> repeat (fastvlm_adapter.py:445-446)
    }
    if constexpr (MODE == 1) { vmax = fmaxf(vmax, fmaxf(v[0], fmaxf(v[1], v[2]))); continue; }
    if constexpr (MODE == 2) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float t = v[c];
        if (by255) t = __fdiv_rn(t, 255.0f);
        v[c] = __fdiv_rn(__fsub_rn(t, nrm.mean[c]), nrm.std[c]);
      }
    }
    uint2 o;
    o.x = pack_bf2(v[0], v[1]);
    o.y = pack_bf2(v[2], 0.0f);
    *reinterpret_cast<uint2*>(p.pix + (((size_t)b * p.S + y) * p.S + x) * 4) = o;
  }
  if constexpr (MODE == 1) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
    if ((threadIdx.x & 63) == 0 && vmax > -INFINITY) atomicMax(nrm.vmax, lb_ord(vmax));
  }
}

// ------------------------------------------------------------------------------------------------ stem conv
// thread = 4 horizontally adjacent output pixels x 8 output channels; weights [27][Cout] fp32 staged in LDS.
__global__ __launch_bounds__(256) void stem_conv_kernel(const bf16_t* __restrict__ pix, const float* __restrict__ w,
                                                         const float* __restrict__ bias, bf16_t* __restrict__ y, int B,
                                                         int S, int Cout) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [27][Cout]
  for (int i = threadIdx.x; i < 27 * Cout; i += 256) sw[i] = w[i];
  __syncthreads();
  const int So = S >> 1, G = Cout >> 3, WQ = (So + 3) >> 2;
  const int per_block = 256 / G;
  const int g = threadIdx.x % G, ql = threadIdx.x / G;
  if (ql >= per_block) return;
  long q = (long)blockIdx.x * per_block + ql;
  const int xq = (int)(q % WQ); q /= WQ;
  const int oy = (int)(q % So);
  const int b = (int)(q / So);
  if (b >= B) return;
  const int ox0 = xq * 4, co0 = g * 8;
  float acc[4][8];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[r][e] = bias[co0 + e];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy * 2 - 1 + ky;
    if (iy < 0 || iy >= S) continue;
    float xin[9][3];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const int ix = ox0 * 2 - 1 + c;
      if (ix >= 0 && ix < S) {
        const uint2 u = *reinterpret_cast<const uint2*>(pix + (((size_t)b * S + iy) * S + ix) * 4);
        xin[c][0] = bf_lo(u.x); xin[c][1] = bf_hi(u.x); xin[c][2] = bf_lo(u.y);
      } else {
        xin[c][0] = xin[c][1] = xin[c][2] = 0.0f;
      }
    }
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        const float* wp = sw + ((ky * 3 + kx) * 3 + ci) * Cout + co0;
        const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
        const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[r][e] += xin[2 * r + kx][ci] * wv[e];
      }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (ox0 + r >= So) break;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[r][e] = gelu_f(acc[r][e]);
    *reinterpret_cast<uint4*>(y + (((size_t)b * So + oy) * So + ox0 + r) * Cout + co0) = pack8(acc[r]);
  }
}

// ------------------------------------------------------------------------------------------------ stem conv on MFMA
// The dense 3x3 stride-2 stem as an implicit GEMM on v_mfma_f32_16x16x32_bf16 with K padded 27 -> 64 so that every
// operand fragment is contiguous pixel memory: k-slot (kstep, g, e) = kernel row ky = 2*kstep + (g>>1) (ky < 3), pixel
// pair pp = g&1 (input columns 2ox-1+2pp, +1), e = 4*(pixel in pair) + channel -- i.e. 8 consecutive bf16 of the
// (B,S,S,4) pixel image.  Operands are swapped (A = weights [cout][k], B = pixels) so D has the pixel on the lane and 4
// consecutive output channels in its registers (8-byte NHWC stores).  Weights (6 n-tiles x 2 k-steps) stay in
// registers while the wave walks TPW tiles of 16 output pixels.  27/64 of the MACs are useful and it does not matter:
// the kernel is bound by its 3.2 GB of output and the 96 GELUs per pixel.
template <int TPW>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const bf16_t* __restrict__ pix, const bf16_t* __restrict__ wp,
                                                         const float* __restrict__ bias, bf16_t* __restrict__ y, int B,
                                                         int S, int Cout, long ntiles) {
  const int lane = threadIdx.x & 63, fr = lane & 15, fg = lane >> 4;
  const int So = S >> 1, tiles_per_row = (So + 15) >> 4;
  const int NTn = Cout >> 4;  // 16-channel tiles (<= 8)
  bf16x8 wf[8][2];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      wf[nt][ks] = nt < NTn ? __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wp + ((size_t)(nt * 16 + fr) * 64 + ks * 32 + fg * 8)))
                            : __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0));
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int t = 0; t < TPW; ++t) {
    const long tile = wave * TPW + t;
    if (tile >= ntiles) break;
    const int tx = (int)(tile % tiles_per_row);
    const int oy = (int)((tile / tiles_per_row) % So);
    const long b = tile / ((long)tiles_per_row * So);
    const int ox = tx * 16 + fr;
    // B fragments: 2 pixels x 4 channels per (k-step, g)
    bf16x8 xf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ky = 2 * ks + (fg >> 1), pp = fg & 1;
      const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + 2 * pp;
      uint2 lo = make_uint2(0, 0), hi = make_uint2(0, 0);
      if (ky < 3 && iy >= 0 && iy < S && ox < So) {
        const bf16_t* rp = pix + (((size_t)b * S + iy) * S) * 4;
        if (ix >= 0 && ix < S) lo = *reinterpret_cast<const uint2*>(rp + (size_t)ix * 4);
        if (ix + 1 >= 0 && ix + 1 < S) hi = *reinterpret_cast<const uint2*>(rp + (size_t)(ix + 1) * 4);
      }
      xf[ks] = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
    }
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      if (nt >= NTn) break;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][0], xf[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][1], xf[1], acc, 0, 0, 0);
      if (ox < So) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + nt * 16 + fg * 4);
        uint2 o;
        o.x = pack_bf2(gelu_f(acc[0] + bv.x), gelu_f(acc[1] + bv.y));
        o.y = pack_bf2(gelu_f(acc[2] + bv.z), gelu_f(acc[3] + bv.w));
        *reinterpret_cast<uint2*>(y + (((size_t)b * So + oy) * So + ox) * Cout + nt * 16 + fg * 4) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ depthwise conv
// block = 256 threads over (pixel quads) x (a slice of GS <= 16 channel groups of 8); the slice's weights [k*k][GS*8]
// fp32 are staged in LDS once per block.  thread = R=4 outputs along x  x  8 output channels.  Per kernel row the 7 (or
// 3) taps' weights sit in registers and the input columns stream through one at a time (load 16 B, convert once, FMA
// into every output it feeds), which keeps the kernel near 110 VGPRs (4 waves/SIMD) instead of holding a whole
// (R+K-1)-column window.
template <int K, int S, int MULT>
__global__ __launch_bounds__(256) void dwconv_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, bf16_t* __restrict__ y, int B,
                                                      int H, int W, int C, int Ho, int Wo, int gelu, int GS,
                                                      int nslices, long nquads) {
  constexpr int R = 4, NCOL = (R - 1) * S + K, CI = 8 / MULT, PAD = K / 2;
  __shared__ __attribute__((aligned(16))) float sw[K * K * 16 * 8];
  const int Cout = C * MULT, WQ = (Wo + R - 1) / R;
  const int slice = blockIdx.x % nslices;
  const long qblock = blockIdx.x / nslices;
  const int sc = GS * 8;  // channels in this slice
  for (int i = threadIdx.x; i < K * K * sc; i += 256) sw[i] = w[(size_t)(i / sc) * Cout + slice * sc + (i % sc)];
  __syncthreads();
  const int qpb = 256 / GS;
  const int g = threadIdx.x % GS, ql = threadIdx.x / GS;
  long q = qblock * qpb + ql;
  if (ql >= qpb || q >= nquads) return;
  const int xq = (int)(q % WQ); q /= WQ;
  const int oy = (int)(q % Ho);
  const long b = q / Ho;
  const int ox0 = xq * R, co0 = slice * sc + g * 8, ci0 = co0 / MULT;
  float acc[R][8];
  {
    const float4 b0 = *reinterpret_cast<const float4*>(bias + co0), b1 = *reinterpret_cast<const float4*>(bias + co0 + 4);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      acc[r][0] = b0.x; acc[r][1] = b0.y; acc[r][2] = b0.z; acc[r][3] = b0.w;
      acc[r][4] = b1.x; acc[r][5] = b1.y; acc[r][6] = b1.z; acc[r][7] = b1.w;
    }
  }
#pragma unroll 1
  for (int ky = 0; ky < K; ++ky) {
    const int iy = oy * S - PAD + ky;
    if (iy < 0 || iy >= H) continue;
    float wr[K][8];
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const float* wp = sw + (ky * K + kx) * sc + g * 8;
      const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
      wr[kx][0] = w0.x; wr[kx][1] = w0.y; wr[kx][2] = w0.z; wr[kx][3] = w0.w;
      wr[kx][4] = w1.x; wr[kx][5] = w1.y; wr[kx][6] = w1.z; wr[kx][7] = w1.w;
    }
    const bf16_t* rowp = x + ((size_t)b * H + iy) * W * C + ci0;
#pragma unroll
    for (int c = 0; c < NCOL; ++c) {
      const int ix = ox0 * S - PAD + c;
      const bool ok = ix >= 0 && ix < W;
      float xv[CI];
      if (MULT == 1) {
        const uint4 u = ok ? *reinterpret_cast<const uint4*>(rowp + (size_t)ix * C) : make_uint4(0, 0, 0, 0);
        unpack8(u, xv);
      } else {
        const uint2 u = ok ? *reinterpret_cast<const uint2*>(rowp + (size_t)ix * C) : make_uint2(0, 0);
        xv[0] = bf_lo(u.x); xv[1] = bf_hi(u.x); xv[2] = bf_lo(u.y); xv[3] = bf_hi(u.y);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int kx = c - r * S;  // compile-time after unrolling
        if (kx >= 0 && kx < K) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[r][e] += xv[e / MULT] * wr[kx][e];
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (ox0 + r >= Wo) break;
    if (gelu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[r][e] = gelu_f(acc[r][e]);
    }
    *reinterpret_cast<uint4*>(y + (((size_t)b * Ho + oy) * Wo + ox0 + r) * Cout + co0) = pack8(acc[r]);
  }
}

// ------------------------------------------------------------------------------------------------ depthwise conv, LDS-tiled
// stride-1 depthwise k x k for the large feature maps (W >= 32): block = 8 x 32 output pixels x 32 channels.  The
// (8+k-1) x (32+k-1) x 32ch halo tile is staged ONCE into LDS with 64-byte pixel segments (global side: each input
// element fetched ~2x from L2, ~1x from HBM), and the 17.5 re-reads per output of the 7x7 window come from LDS as
// ds_read_b128.  LDS image: pixel = 64 B, plus one 64-B pad every 4 pixels, so the four pixel quads of a 16-lane
// ds_read_b128 group land on four different 64-B bank windows (conflict-free).  thread = 4 outputs along x x 8 ch.
template <int K>
__global__ __launch_bounds__(256, 3) void dwconv_tile_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                           int H, int W, int C, int gelu, int tiles_x, int tiles_y,
                                                           int nslices) {
  constexpr int TH = 8, TW = 32, CS = 32, PAD = K / 2, IH = TH + K - 1, IW = TW + K - 1;
  constexpr int ROWB = IW * 64 + ((IW + 3) / 4) * 64;  // bytes per LDS tile row
  __shared__ __attribute__((aligned(16))) char smem[IH * ROWB + K * K * CS * 4];
  float* sw = reinterpret_cast<float*>(smem + IH * ROWB);
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int slice = bid % nslices; bid /= nslices;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int tyb = bid % tiles_y;
  const long b = bid / tiles_y;
  const int c0 = slice * CS;
  for (int i = tid; i < K * K * CS; i += 256) sw[i] = w[(size_t)(i / CS) * C + c0 + (i % CS)];
  for (int i = tid; i < IH * IW * 4; i += 256) {
    const int chunk = i & 3, pc = i >> 2;
    const int col = pc % IW, row = pc / IW;
    const int iy = tyb * TH - PAD + row, ix = tx * TW - PAD + col;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + iy) * W + ix) * C + c0 + chunk * 8);
    *reinterpret_cast<uint4*>(smem + row * ROWB + col * 64 + (col >> 2) * 64 + chunk * 16) = v;
  }
  __syncthreads();
  const int g = tid & 3, qx = (tid >> 2) & 7, r_ = tid >> 5;
  const int oy = tyb * TH + r_, ox0 = tx * TW + qx * 4;
  float acc[4][8];
  {
    const float4 b0 = *reinterpret_cast<const float4*>(bias + c0 + g * 8), b1 = *reinterpret_cast<const float4*>(bias + c0 + g * 8 + 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      acc[r][0] = b0.x; acc[r][1] = b0.y; acc[r][2] = b0.z; acc[r][3] = b0.w;
      acc[r][4] = b1.x; acc[r][5] = b1.y; acc[r][6] = b1.z; acc[r][7] = b1.w;
    }
  }
#pragma unroll 1
  for (int ky = 0; ky < K; ++ky) {
    float wr[K][8];
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const float* wp = sw + (ky * K + kx) * CS + g * 8;
      const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
      wr[kx][0] = w0.x; wr[kx][1] = w0.y; wr[kx][2] = w0.z; wr[kx][3] = w0.w;
      wr[kx][4] = w1.x; wr[kx][5] = w1.y; wr[kx][6] = w1.z; wr[kx][7] = w1.w;
    }
    const char* rowp = smem + (r_ + ky) * ROWB + g * 16;
#pragma unroll
    for (int c = 0; c < K + 3; ++c) {
      const int col = qx * 4 + c;
      float xv[8];
      unpack8(*reinterpret_cast<const uint4*>(rowp + col * 64 + (col >> 2) * 64), xv);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kx = c - r;
        if (kx >= 0 && kx < K) {
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[r][e] += xv[e] * wr[kx][e];
        }
      }
    }
  }
  if (oy >= H) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (ox0 + r >= W) break;
    if (gelu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[r][e] = gelu_f(acc[r][e]);
    }
    *reinterpret_cast<uint4*>(y + (((size_t)b * H + oy) * W + ox0 + r) * C + c0 + g * 8) = pack8(acc[r]);
  }
}

// ------------------------------------------------------------------------------------------------ fused stem
// conv 3x3 s2 (3 -> 96) + GELU and depthwise 3x3 s2 + GELU in one kernel ([UNVENDORED] mci.py convolutional_stem, first two
// MobileOne blocks after reparameterisation).  Unfused, the 96-channel half-resolution map is the largest tensor of the
// whole path: at B = 64 it is 3.2 GB written by the first conv and read straight back by the second, 2.5 ms of pure
// traffic.  Here a block owns 4 x 32 pixels of the quarter-resolution output: it computes the 9 x 65 half-resolution
// pixels they need on MFMA (implicit GEMM as in stem_mfma_kernel, K 27 -> 64), rounds them to bf16 into LDS exactly as the
// unfused pair rounds them to HBM, and runs the depthwise conv out of LDS.  The halo recompute costs 14 %; HBM sees the
// image once and the quarter-resolution map once.  Persistent blocks (one per CU, 8 waves), weights loaded once.
constexpr int SF_C = 96, SF_TR = 4, SF_TC = 32, SF_R1 = 2 * SF_TR + 1, SF_C1 = 2 * SF_TC + 1;
constexpr int SF_NSEG = (SF_R1 * SF_C1 + 15) / 16;           // 37 segments of 16 half-resolution pixels
constexpr int SF_PS = SF_C * 2 + 16;                      // bytes per half-resolution pixel in LDS (16-B aligned, 2-way banks)
constexpr int SF_S1 = SF_R1 * SF_C1 * SF_PS;              // 121,680 B
constexpr int SF_PR = 4 * SF_TR + 3, SF_PCH = (4 * SF_TC + 4) / 2;   // input patch: 19 rows x 66 16-byte chunks (132 pixels)
constexpr int SF_PATCH = SF_PR * SF_PCH * 16 + 16;        // 20,064 B + one zero chunk (the last row's unused fourth kernel column reads it)
constexpr int SF_LDS = SF_S1 + 10 * SF_C * 4 + SF_PATCH;  // + depthwise taps [9][96] and bias [96] fp32 + the pixel patch
constexpr int SF_PLD = (SF_PR * SF_PCH + 511) / 512;      // 16-byte patch chunks per thread
// LB (the fused on-device input pipeline): the tile's input patch is not read from a letterboxed frame but SAMPLED from
// the source image (lb_pixel: letterbox_kernel's arithmetic), so the (B,S,S,4) frame -- 537 MB at B = 64 -- is neither written nor read.
template <bool LB>
__global__ __launch_bounds__(512, 1) void stem_fused_kernel(const bf16_t* __restrict__ pix, const bf16_t* __restrict__ wp,
                                                             const float* __restrict__ b1, const float* __restrict__ w2,
                                                             const float* __restrict__ b2, bf16_t* __restrict__ y, int B, int S,
                                                             long ntiles, LbParams lb) {
  extern __shared__ __attribute__((aligned(16))) char sf_smem[];
  char* s1 = sf_smem;
  float* sw2 = reinterpret_cast<float*>(sf_smem + SF_S1);   // [9][96] taps, then [96] bias
  char* spx = sf_smem + SF_S1 + 10 * SF_C * 4;              // the tile's input pixels, [19][132] x 8 B, zero outside the image
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const int S1 = S >> 1, S2 = S >> 2;                        // half / quarter resolution
  const int tiles_x = (S2 + SF_TC - 1) / SF_TC, tiles_y = (S2 + SF_TR - 1) / SF_TR;
  for (int i = tid; i < 10 * SF_C; i += 512) sw2[i] = i < 9 * SF_C ? w2[i] : b2[i - 9 * SF_C];
  bf16x8 wf[6][2];
#pragma unroll
  for (int nt = 0; nt < 6; ++nt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      wf[nt][ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(wp + ((size_t)(nt * 16 + fr) * 64 + ks * 32 + fg * 8)));
  float4 bv1[6];
#pragma unroll
  for (int nt = 0; nt < 6; ++nt) bv1[nt] = *reinterpret_cast<const float4*>(b1 + nt * 16 + fg * 4);

  // The tile's input pixels go through LDS: rows 4 ty TR - 3 .. + 18, pixels 4 tx TC - 4 .. + 127 (two pixels per 16-byte
  // chunk, so a chunk is inside or outside the image as a whole).  The NEXT tile's patch is fetched into registers under
  // the depthwise pass and written once that pass is over -- fetched per segment from global memory instead, the first conv
  // stalled on one memory latency per segment (two waves per SIMD cannot hide it).
  uint4 pre[SF_PLD];
  // tile arithmetic in 32 bits (the launcher bounds ntiles): 64-bit division is emulated, hundreds of scalar instructions
  auto patch_fetch = [&](unsigned tile) {
    const unsigned r_ = tile / (unsigned)tiles_x;
    const int tx = (int)(tile - r_ * (unsigned)tiles_x), ty = (int)(r_ % (unsigned)tiles_y);
    const long b = r_ / (unsigned)tiles_y;
    const int iy0 = 4 * ty * SF_TR - 3, ix0 = 4 * tx * SF_TC - 4;
#pragma unroll
    for (int j = 0; j < SF_PLD; ++j) {
      const int c = tid + 512 * j, r = c / SF_PCH, cc = c % SF_PCH;
      const int iy = iy0 + r, ix = ix0 + 2 * cc;
      const bool in_ = c < SF_PR * SF_PCH && iy >= 0 && iy < S && ix >= 0 && ix < S;
      if constexpr (LB) {
        pre[j] = make_uint4(0, 0, 0, 0);
        if (in_) {
          const uint2 p0 = lb_pixel(lb, (int)b, iy, ix), p1 = lb_pixel(lb, (int)b, iy, ix + 1);
          pre[j] = make_uint4(p0.x, p0.y, p1.x, p1.y);
        }
      } else {
        pre[j] = in_ ? *reinterpret_cast<const uint4*>(pix + (((size_t)b * S + iy) * S + ix) * 4) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto patch_store = [&]() {
#pragma unroll
    for (int j = 0; j < SF_PLD; ++j) {
      const int c = tid + 512 * j;
      if (c < SF_PR * SF_PCH) *reinterpret_cast<uint4*>(spx + c * 16) = pre[j];
    }
  };
  if (tid == 0) *reinterpret_cast<uint4*>(spx + SF_PR * SF_PCH * 16) = make_uint4(0, 0, 0, 0);
  if ((long)blockIdx.x < ntiles) { patch_fetch(blockIdx.x); patch_store(); }

  for (unsigned tile = blockIdx.x; tile < (unsigned)ntiles; tile += gridDim.x) {
    const unsigned r_ = tile / (unsigned)tiles_x;
    const int tx = (int)(tile - r_ * (unsigned)tiles_x), ty = (int)(r_ % (unsigned)tiles_y);
    const long b = r_ / (unsigned)tiles_y;
    const int y1_0 = 2 * ty * SF_TR - 1, x1_0 = 2 * tx * SF_TC - 1;   // half-resolution origin of the halo region
    __syncthreads();   // the patch is written, the previous tile's depthwise pass is done with s1 (and the taps are staged)
    // ---- first conv: the region's 9 x 65 half-resolution pixels, rows flattened, in SF_NSEG segments of 16 dealt to the 8
    // waves (per row that would be 5 segments for 65 pixels: 45 instead of 37).  Fragment of half-res
    // pixel (r1, cs): k-slot group fg = (kernel row 2 ks + (fg >> 1), pixel pair fg & 1) -> patch row 2 r1 + ky, pixels
    // 2 cs + 2 pp + 1 and + 2 (kernel row 3 does not exist: zero operand against zero weights)
#define SF_FETCH(SEG, XF, LIVE)                                                                                  \
  {                                                                                                              \
    const int f_ = min((SEG) * 16 + fr, SF_R1 * SF_C1 - 1), r1_ = f_ / SF_C1, cs_ = f_ - r1_ * SF_C1;            \
    const int oy_ = y1_0 + r1_, ox_ = x1_0 + cs_;                                                                \
    LIVE = oy_ >= 0 && oy_ < S1 && ox_ >= 0 && ox_ < S1;                                                         \
    const char* pp_ = spx + ((2 * r1_ + (fg >> 1)) * SF_PCH * 2 + 2 * cs_ + 2 * (fg & 1) + 1) * 8;               \
    const uint2 l0_ = *reinterpret_cast<const uint2*>(pp_), h0_ = *reinterpret_cast<const uint2*>(pp_ + 8);      \
    uint2 l1_ = make_uint2(0, 0), h1_ = make_uint2(0, 0);                                                        \
    if (fg < 2) {                                                                                                \
      l1_ = *reinterpret_cast<const uint2*>(pp_ + 2 * SF_PCH * 16);                                              \
      h1_ = *reinterpret_cast<const uint2*>(pp_ + 2 * SF_PCH * 16 + 8);                                          \
    }                                                                                                            \
    XF[0] = __builtin_bit_cast(bf16x8, make_uint4(l0_.x, l0_.y, h0_.x, h0_.y));                                  \
    XF[1] = __builtin_bit_cast(bf16x8, make_uint4(l1_.x, l1_.y, h1_.x, h1_.y));                                  \
  }
    bf16x8 xf[2], xn[2];
    bool live, live_n;
    SF_FETCH(wid, xf, live)   // outside the map: the depthwise conv's zero padding
    for (int seg = wid; seg < SF_NSEG; seg += 8) {
      SF_FETCH(seg + 8, xn, live_n)
      const int f = seg * 16 + fr;                                     // pixel of the region, rows flattened
      char* dst = s1 + f * SF_PS + fg * 8;
      // all twelve MFMAs first, then the GELUs four pairs at a time: four Horner chains in lockstep instead of two
      f32x4 acc[6];
#pragma unroll
      for (int nt = 0; nt < 6; ++nt) {
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][0], xf[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][1], xf[1], acc[nt], 0, 0, 0);
      }
#pragma unroll
      for (int np = 0; np < 3; ++np) {
        const int n0 = 2 * np, n1 = 2 * np + 1;
        f32x2 g[4] = {{acc[n0][0] + bv1[n0].x, acc[n0][1] + bv1[n0].y}, {acc[n0][2] + bv1[n0].z, acc[n0][3] + bv1[n0].w},
                      {acc[n1][0] + bv1[n1].x, acc[n1][1] + bv1[n1].y}, {acc[n1][2] + bv1[n1].z, acc[n1][3] + bv1[n1].w}};
        gelu2_n<4>(g);
        uint2 o0, o1;
        o0.x = live ? pack_bf2(g[0].x, g[0].y) : 0u;
        o0.y = live ? pack_bf2(g[1].x, g[1].y) : 0u;
        o1.x = live ? pack_bf2(g[2].x, g[2].y) : 0u;
        o1.y = live ? pack_bf2(g[3].x, g[3].y) : 0u;
        if (f < SF_R1 * SF_C1) {
          *reinterpret_cast<uint2*>(dst + n0 * 32) = o0;
          *reinterpret_cast<uint2*>(dst + n1 * 32) = o1;
        }
      }
      xf[0] = xn[0]; xf[1] = xn[1]; live = live_n;
    }
#undef SF_FETCH
    __syncthreads();   // s1 is complete; nobody reads the patch any more
    const unsigned tile_n = tile + gridDim.x;
    if (tile_n < (unsigned)ntiles) patch_fetch(tile_n);   // in flight under the depthwise pass
    // ---- depthwise 3x3 stride 2 out of LDS: item = (output pixel, 8-channel group), 128 x 12 items over 512 threads
#pragma unroll 1
    for (int it = 0; it < SF_TR * SF_TC * (SF_C / 8) / 512; ++it) {
      const int item = tid + 512 * it;
      const int cg = item % (SF_C / 8), p = item / (SF_C / 8), pr = p / SF_TC, pc = p % SF_TC;
      float acc[8];
      {
        const float4 a0 = *reinterpret_cast<const float4*>(sw2 + 9 * SF_C + cg * 8), a1 = *reinterpret_cast<const float4*>(sw2 + 9 * SF_C + cg * 8 + 4);
        acc[0] = a0.x; acc[1] = a0.y; acc[2] = a0.z; acc[3] = a0.w; acc[4] = a1.x; acc[5] = a1.y; acc[6] = a1.z; acc[7] = a1.w;
      }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          float xv[8];
          unpack8(*reinterpret_cast<const uint4*>(s1 + ((2 * pr + ky) * SF_C1 + 2 * pc + kx) * SF_PS + cg * 16), xv);
          const float* wt = sw2 + (ky * 3 + kx) * SF_C + cg * 8;
          const float4 w0 = *reinterpret_cast<const float4*>(wt), w1 = *reinterpret_cast<const float4*>(wt + 4);
          acc[0] += xv[0] * w0.x; acc[1] += xv[1] * w0.y; acc[2] += xv[2] * w0.z; acc[3] += xv[3] * w0.w;
          acc[4] += xv[4] * w1.x; acc[5] += xv[5] * w1.y; acc[6] += xv[6] * w1.z; acc[7] += xv[7] * w1.w;
        }
      f32x2 g[4] = {{acc[0], acc[1]}, {acc[2], acc[3]}, {acc[4], acc[5]}, {acc[6], acc[7]}};
      gelu2_n<4>(g);
      const int oy = ty * SF_TR + pr, ox = tx * SF_TC + pc;
      if (oy < S2 && ox < S2) {
        uint4 o;
        o.x = pack_bf2(g[0].x, g[0].y); o.y = pack_bf2(g[1].x, g[1].y); o.z = pack_bf2(g[2].x, g[2].y); o.w = pack_bf2(g[3].x, g[3].y);
        *reinterpret_cast<uint4*>(y + (((size_t)b * S2 + oy) * S2 + ox) * SF_C + cg * 8) = o;
      }
    }
    if (tile_n < (unsigned)ntiles) patch_store();
  }
}

// ------------------------------------------------------------------------------------------------ depthwise conv on MFMA
// The 7x7 depthwise conv is 49 fp32 FMAs per output element on the VALU (157 TFLOP/s chip-wide against 2.5 PFLOP/s of
// matrix rate), and it was the second largest item of the step.  Per channel a 1-D convolution along x is a banded
// (Toeplitz) matrix product, and v_mfma_f32_4x4x4_16b_bf16 multiplies SIXTEEN independent 4x4 blocks per instruction,
// so one instruction serves 16 channels with no cross-channel mixing:
//     block = channel c;  A_c[i][k] = w_c[ky][4m + k - i]  (Toeplitz piece: 4 outputs i x 4 input columns k, k-block m)
//                         B_c[k][j] = x[row r0 + j + ky][col 4(q+m) + k][c]     (4 input columns k x 4 rows j)
//                         D_c[i][j] += A_c B_c                                   (4 output columns i x 4 rows j)
// 3 k-blocks cover the 10-column window of 4 outputs: 21 MFMAs per (16 ch x 4 rows x 4 cols), 58 % of their MACs
// useful, still ~2.3x the VALU's peak rate and freeing the VALU.  The price is layout: operands want 4 consecutive
// PIXELS of one channel per lane, NHWC gives 8 channels of one pixel per 16 B, so the halo tile is transposed through
// registers (v_perm_b32) on its way into LDS ([row][column quad][channel][4 columns]) and the 8x32x32 output tile goes
// back through LDS to leave as 16-byte NHWC stores.  Weights arrive as a precomputed bf16 Toeplitz table.
typedef __attribute__((ext_vector_type(4))) short s16x4;

template <int K, int TH>
__global__ __launch_bounds__(256, TH == 8 ? 3 : 2) void dwconv_mfma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ ttab,
                                                              const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                              int H, int W, int C, int gelu, int tiles_x, int tiles_y,
                                                              int nslices) {
  constexpr int TW = 32, NP = TH / 8, PAD = K / 2, IH = TH + K - 1, IW = TW + K - 1;
  constexpr int NQ = (IW + 3) / 4, NM = (K + 3 + 3) / 4;       // column quads in the halo tile; k-blocks per output quad
  constexpr int RS = NQ * 256 + 64;                            // LDS bytes per halo row (== 64 mod 256: 4 rows, 4 bank windows)
  constexpr int T_BYTES = 2 * K * NM * 512, X_BYTES = IH * RS;
  __shared__ __attribute__((aligned(16))) char smem[X_BYTES];
  // LDS image, input and output alike: [row][column quad][32 channel slots][4 columns] bf16, 256 B per (row, quad) cell,
  // channel ch in slot ch ^ (2 (quad & 3) | ch >> 4).  Unswizzled, the 16 lanes one ds_write_b64 services together
  // (4 quads x 4 channel groups, one channel each) hit 2 bank pairs, an 8-way conflict that made the transpose, not HBM,
  // the bound of this kernel; swizzled they cover all 32 banks.  The MFMA-side accesses only see a per-quad constant.
  char* sX = smem;  // the output tile reuses it once every wave is done reading the halo tile

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // The channel slices of one pixel tile share 128-B lines and neighbouring tiles share halo rows/columns: keep them on
  // one XCD (one L2) instead of dealing them round-robin over all eight.
#ifdef DW_NO_XCD
  int bid = blockIdx.x;
#else
  int bid = xcd_remap(blockIdx.x, gridDim.x);
#endif
  const int slice = bid % nslices; bid /= nslices;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int tyb = bid % tiles_y;
  const long b = bid / tiles_y;
  const int c0 = slice * 32;
  const int gg = wid & 1, rg = wid >> 1;          // wave = (16-channel group, 4 output rows)
  const int bch = lane >> 2, jr = lane & 3;       // lane = (channel within the group, row within the 4)

  // ---- Toeplitz fragments straight from the (L2-resident) table into registers: 8 B per lane per (ky, m)
  s16x4 afr[K][NM];
  {
    const char* tsrc = reinterpret_cast<const char*>(ttab) + (size_t)slice * T_BYTES + (size_t)gg * K * NM * 512 + lane * 8;
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int m = 0; m < NM; ++m) afr[ky][m] = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(tsrc + (ky * NM + m) * 512));
  }
  // ---- halo tile: 4 pixels x 8 channels per task, transposed in registers to 8 channels x 4 pixels.  All of a
  // thread's loads are issued before any is consumed (one exposed HBM/L2 latency per block instead of one per task).
  constexpr int NTASK = IH * NQ * 4, TPT = (NTASK + 255) / 256;
  uint4 px[TPT][4];
#pragma unroll
  for (int tt = 0; tt < TPT; ++tt) {
    const int task = tid + 256 * tt;
    const int cg = task & 3, quad = (task >> 2) % NQ, row = (task >> 2) / NQ;
    const int iy = tyb * TH - PAD + row, ix0 = tx * TW - PAD + quad * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ix = ix0 + j;
      px[tt][j] = (task < NTASK && iy >= 0 && iy < H && ix >= 0 && ix < W && quad * 4 + j < IW)
                      ? *reinterpret_cast<const uint4*>(x + (((size_t)b * H + iy) * W + ix) * C + c0 + cg * 8)
                      : make_uint4(0, 0, 0, 0);
    }
  }
#pragma unroll
  for (int tt = 0; tt < TPT; ++tt) {
    const int task = tid + 256 * tt;
    if (task < NTASK) {
      const int cg = task & 3, quad = (task >> 2) % NQ, row = (task >> 2) / NQ;
      const uint32_t d[4][4] = {{px[tt][0].x, px[tt][0].y, px[tt][0].z, px[tt][0].w}, {px[tt][1].x, px[tt][1].y, px[tt][1].z, px[tt][1].w},
                                {px[tt][2].x, px[tt][2].y, px[tt][2].z, px[tt][2].w}, {px[tt][3].x, px[tt][3].y, px[tt][3].z, px[tt][3].w}};
      const uint32_t dst = (uint32_t)(row * RS + quad * 256 + cg * 64) | (uint32_t)((((quad & 3) << 1) | (cg >> 1)) << 3);
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) {  // dword dd of a pixel holds channels 2dd (low half) and 2dd+1 (high half)
        uint2 ev, od;
        ev.x = __builtin_amdgcn_perm(d[1][dd], d[0][dd], 0x05040100u);
        ev.y = __builtin_amdgcn_perm(d[3][dd], d[2][dd], 0x05040100u);
        od.x = __builtin_amdgcn_perm(d[1][dd], d[0][dd], 0x07060302u);
        od.y = __builtin_amdgcn_perm(d[3][dd], d[2][dd], 0x07060302u);
        *reinterpret_cast<uint2*>(sX + (dst ^ (uint32_t)((2 * dd) * 8))) = ev;
        *reinterpret_cast<uint2*>(sX + (dst ^ (uint32_t)((2 * dd + 1) * 8))) = od;
      }
    }
  }
  __syncthreads();

  // lane's four swizzled channel offsets (one per quad & 3) on its first halo row; everything else is an immediate
  const char* lrow[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) lrow[v] = sX + (rg * 4 * NP + jr) * RS + (((gg * 16 + bch) ^ ((v << 1) | gg)) << 3);
  f32x4 acc[NP][8];
#pragma unroll
  for (int pp = 0; pp < NP; ++pp)
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[pp][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int pp = 0; pp < NP; ++pp)   // a wave's NP groups of 4 output rows, one after the other
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      s16x4 xq[NQ];
#pragma unroll
      for (int t = 0; t < NQ; ++t)
        xq[t] = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(lrow[t & 3] + (pp * 4 + ky) * RS + t * 256));
#pragma unroll
      for (int q = 0; q < 8; ++q)
#pragma unroll
        for (int m = 0; m < NM; ++m)
#ifdef DW_ABL_MFMA  // tools/dw_micro.hip only: one MFMA per (ky, q) instead of NM, to price the matrix work
          if (q + m < NQ && m == 0) acc[pp][q] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(afr[ky][m], xq[q + m], acc[pp][q], 0, 0, 0);
#else
          if (q + m < NQ) acc[pp][q] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(afr[ky][m], xq[q + m], acc[pp][q], 0, 0, 0);
#endif
    }
  __syncthreads();  // all waves are done with the halo tile; reuse it for the output tile

  // ---- bias (+GELU), then back through LDS in the same cell layout: one 8-byte store per quad (4 columns of the lane's
  // channel); the NHWC transpose is done by the reading side with v_perm, mirroring the load path
  {
    const float bv = bias[c0 + gg * 16 + bch];
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = acc[pp][q][i] + bv;
          if (gelu) v[i] = gelu_f(v[i]);
        }
        uint2 u;
        u.x = pack_bf2(v[0], v[1]);
        u.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(const_cast<char*>(lrow[q & 3]) + pp * 4 * RS + q * 256) = u;
      }
  }
  __syncthreads();
#pragma unroll
  for (int ot = 0; ot < NP; ++ot) {
    const int task = tid + 256 * ot;
    const int cg = task & 3, quad = (task >> 2) & 7, row = task >> 5;   // TH rows x 8 quads x 4 channel groups
    const uint32_t src = (uint32_t)(row * RS + quad * 256 + cg * 64) | (uint32_t)((((quad & 3) << 1) | (cg >> 1)) << 3);
    uint2 r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = *reinterpret_cast<const uint2*>(sX + (src ^ (uint32_t)(e * 8)));
    const int oy = tyb * TH + row, ox0 = tx * TW + quad * 4;
    if (oy < H) {
      bf16_t* yp = y + (((size_t)b * H + oy) * W + ox0) * C + c0 + cg * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (ox0 + j < W) {
          uint4 o;
          const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
#define FV_PX(A, B) __builtin_amdgcn_perm((j & 2) ? r[B].y : r[B].x, (j & 2) ? r[A].y : r[A].x, sel)
          o.x = FV_PX(0, 1); o.y = FV_PX(2, 3); o.z = FV_PX(4, 5); o.w = FV_PX(6, 7);
#undef FV_PX
          *reinterpret_cast<uint4*>(yp + (size_t)j * C) = o;
        }
      }
    }
  }
}

// ---- the PatchEmbed large-kernel conv (7x7, stride 2, two output channels per input channel) on the same MFMA scheme.
// For four adjacent outputs ox = 4Q + i of a row the input window is columns 8Q .. 8Q + 12 of the halo tile, i.e. column
// quads 2Q .. 2Q + 3, and the Toeplitz piece of k-block m is A[i][kk] = w[ky][4m + kk - 2i]; the B operand of output row
// j is halo row 2j + ky.  The two output channels of an input channel share B: two A tables (e = 0, 1), two accumulators.
// Block = 8 x 16 outputs x 32 input channels (64 output channels): 21 x 37 halo pixels, 224 MFMAs per wave.  On the VALU
// this layer was 49 FMAs per output at 0.9 ms for the first (256^2 -> 128^2) map; see dwconv_kernel for the fallback.
__global__ __launch_bounds__(256, 2) void dwconv_s2_mfma_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ ttab,
                                                                 const float* __restrict__ bias, bf16_t* __restrict__ y, int H,
                                                                 int W, int C, int Ho, int Wo, int gelu, int tiles_x, int tiles_y,
                                                                 int nslices, int nimg) {
  constexpr int K = 7, TH = 8, TW = 16, PAD = 3, IH = 2 * TH + K - 2, IW = 2 * TW + K - 2;   // 21 x 37
  constexpr int NQ = (IW + 3) / 4, NM = 4;                     // 10 column quads; 4 k-blocks per output quad
  constexpr int RS = NQ * 256 + 64;                            // LDS bytes per halo row (== 64 mod 256)
  constexpr int X_BYTES = IH * RS;                             // 55,104
  constexpr int OC = 512 + 16;                                 // output cell (row, quad): 64 channels x 4 columns bf16, padded
  static_assert(TH * 4 * OC <= X_BYTES, "the output tile reuses the halo tile's LDS");
  __shared__ __attribute__((aligned(16))) char smem[X_BYTES];
  char* sX = smem;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // persistent: a block keeps its channel slice (and that slice's Toeplitz fragments) and walks pixel tiles; the next
  // tile's halo pixels are in flight in registers while the current one is in the MFMAs and the epilogue -- one tile per
  // block at two blocks per CU left every block's load latency exposed (13 us per tile)
  // Block -> (XCD, chain, slice): blocks b and b + 8 share an XCD, and a 64-byte channel slice is half a cache line, so the
  // nslices blocks that walk the same tiles (and the chains working on neighbouring tiles, which share halos) are put on
  // one XCD: dealt round-robin, every L2 fetched every line for itself.  XCD x owns the tile range [x tpx, (x+1) tpx).
  const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
  const int slice = lb % nslices, chain = lb / nslices, chains = (int)(gridDim.x >> 3) / nslices;
  const int ntiles = nimg * tiles_x * tiles_y;   // 32-bit on purpose: 64-bit division is emulated
  const int tpx = (ntiles + 7) >> 3, t_end = min((xcd + 1) * tpx, ntiles), per_slice = chains;
  const int c0 = slice * 32;                      // first input channel of the slice
  const int gg = wid & 1, eo = wid >> 1;          // wave = (16-input-channel group, which of the two outputs per channel)
  const int bch = lane >> 2, jr = lane & 3;       // lane = (channel within the group, output row within a group of 4)

  // Toeplitz fragments [C/16][e][ky][m][16 ch][4 i][4 kk] straight from the (L2-resident) table
  s16x4 afr[K][NM];
  {
    const char* tsrc = reinterpret_cast<const char*>(ttab) + (size_t)(slice * 2 + gg) * (2 * K * NM * 512) + (size_t)eo * (K * NM * 512) + lane * 8;
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int m = 0; m < NM; ++m) afr[ky][m] = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(tsrc + (ky * NM + m) * 512));
  }
  // halo tile, transposed in registers to [row][column quad][channel slot][4 columns] (see dwconv_mfma_kernel)
  constexpr int NTASK = IH * NQ * 4, TPT = (NTASK + 255) / 256;
  uint4 px[TPT][4];
  // a task's place in the halo tile does not depend on the tile: decode it once (the per-tile address and bounds work was
  // most of this kernel's instruction stream when it was redone for every load)
  int trow[TPT], tcol[TPT], toff[TPT];
  uint32_t tdst[TPT];
#pragma unroll
  for (int tt = 0; tt < TPT; ++tt) {
    const int task = min(tid + 256 * tt, NTASK - 1);   // surplus threads of the last round repeat the last task
    const int cg = task & 3, quad = (task >> 2) % NQ, row = (task >> 2) / NQ;
    trow[tt] = row;
    tcol[tt] = quad * 4;
    toff[tt] = (row * W + quad * 4) * C + cg * 8;
    tdst[tt] = (uint32_t)(row * RS + quad * 256 + cg * 64) | (uint32_t)((((quad & 3) << 1) | (cg >> 1)) << 3);
  }
  auto fetch = [&](int tx, int tyb, int b) {
    const int iy0 = 2 * tyb * TH - PAD, ix0 = 2 * tx * TW - PAD;
    const bf16_t* org = x + ((long)b * H * W + (long)iy0 * W + ix0) * C + c0;   // may lie before the image: only in-range taps are read
#pragma unroll
    for (int tt = 0; tt < TPT; ++tt) {
      const bool rok = (unsigned)(iy0 + trow[tt]) < (unsigned)H;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        px[tt][j] = (rok && (unsigned)(ix0 + tcol[tt] + j) < (unsigned)W) ? *reinterpret_cast<const uint4*>(org + toff[tt] + j * C)
                                                                         : make_uint4(0, 0, 0, 0);
    }
  };
  auto decode = [&](int tile, int& tx, int& tyb, int& b) {
    const unsigned t = (unsigned)tile, r = t / (unsigned)tiles_x;
    tx = (int)(t - r * (unsigned)tiles_x);
    b = (int)(r / (unsigned)tiles_y);
    tyb = (int)(r - (unsigned)b * (unsigned)tiles_y);
  };
  int tile = xcd * tpx + chain, tx = 0, tyb = 0, b = 0, txn = 0, tybn = 0, bn = 0;
  if (tile < t_end) { decode(tile, tx, tyb, b); fetch(tx, tyb, b); }
  for (; tile < t_end; tile += per_slice) {
#pragma unroll
  for (int tt = 0; tt < TPT; ++tt) {
    if (tid + 256 * tt < NTASK) {
      const uint32_t d[4][4] = {{px[tt][0].x, px[tt][0].y, px[tt][0].z, px[tt][0].w}, {px[tt][1].x, px[tt][1].y, px[tt][1].z, px[tt][1].w},
                                {px[tt][2].x, px[tt][2].y, px[tt][2].z, px[tt][2].w}, {px[tt][3].x, px[tt][3].y, px[tt][3].z, px[tt][3].w}};
      const uint32_t dst = tdst[tt];
#pragma unroll
      for (int dd = 0; dd < 4; ++dd) {
        uint2 ev, od;
        ev.x = __builtin_amdgcn_perm(d[1][dd], d[0][dd], 0x05040100u);
        ev.y = __builtin_amdgcn_perm(d[3][dd], d[2][dd], 0x05040100u);
        od.x = __builtin_amdgcn_perm(d[1][dd], d[0][dd], 0x07060302u);
        od.y = __builtin_amdgcn_perm(d[3][dd], d[2][dd], 0x07060302u);
        *reinterpret_cast<uint2*>(sX + (dst ^ (uint32_t)((2 * dd) * 8))) = ev;
        *reinterpret_cast<uint2*>(sX + (dst ^ (uint32_t)((2 * dd + 1) * 8))) = od;
      }
    }
  }
  __syncthreads();
  if (tile + per_slice < t_end) { decode(tile + per_slice, txn, tybn, bn); fetch(txn, tybn, bn); }

  const char* lrow[4];   // lane's swizzled channel offset per quad & 3, on halo row 2 jr (row group 0)
#pragma unroll
  for (int v = 0; v < 4; ++v) lrow[v] = sX + (2 * jr) * RS + (((gg * 16 + bch) ^ ((v << 1) | gg)) << 3);
  f32x4 acc[2][4];       // [row group of 4 output rows][output quad]
#pragma unroll
  for (int rg = 0; rg < 2; ++rg)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[rg][q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ky = 0; ky < K; ++ky)
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
      s16x4 xq[NQ];
#pragma unroll
      for (int t = 0; t < NQ; ++t) xq[t] = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(lrow[t & 3] + (rg * 8 + ky) * RS + t * 256));
      // k-block outermost: the four MFMAs of one accumulator are then four instructions apart instead of back to back
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[rg][q] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(afr[ky][m], xq[2 * q + m], acc[rg][q], 0, 0, 0);
    }
  __syncthreads();  // all waves are done with the halo tile; reuse it for the output tile

  // bias (+GELU), then through LDS: cell (row, quad) = 64 output-channel slots x 4 columns, slot = co ^ (row >> 1 & 1): with
  // the 16-byte cell pad the 16 lanes of a ds_write_b64 (4 channels x 4 rows) cover all 32 banks
  {
    const int co = 2 * (gg * 16 + bch) + eo, sig = (jr >> 1) & 1;
    const float bv = bias[2 * c0 + co];
#pragma unroll
    for (int rg = 0; rg < 2; ++rg)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x2 g[2] = {{acc[rg][q][0] + bv, acc[rg][q][1] + bv}, {acc[rg][q][2] + bv, acc[rg][q][3] + bv}};
        if (gelu) gelu2_n<2>(g);
        uint2 u;
        u.x = pack_bf2(g[0].x, g[0].y);
        u.y = pack_bf2(g[1].x, g[1].y);
        *reinterpret_cast<uint2*>(sX + ((rg * 4 + jr) * 4 + q) * OC + ((co ^ sig) << 3)) = u;
      }
  }
  __syncthreads();
  {
    const int cg = tid & 7, quad = (tid >> 3) & 3, row = tid >> 5;   // 8 rows x 4 quads x 8 channel groups of 8
    const int sig = (row >> 1) & 1;
    uint2 r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = *reinterpret_cast<const uint2*>(sX + (row * 4 + quad) * OC + (((cg * 8 + e) ^ sig) << 3));
    const int oy = tyb * TH + row, ox0 = tx * TW + quad * 4;
    if (oy < Ho) {
      bf16_t* yp = y + (((size_t)b * Ho + oy) * Wo + ox0) * (2 * C) + 2 * c0 + cg * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (ox0 + j < Wo) {
          uint4 o;
          const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
#define FV_PX(A, B) __builtin_amdgcn_perm((j & 2) ? r[B].y : r[B].x, (j & 2) ? r[A].y : r[A].x, sel)
          o.x = FV_PX(0, 1); o.y = FV_PX(2, 3); o.z = FV_PX(4, 5); o.w = FV_PX(6, 7);
#undef FV_PX
          *reinterpret_cast<uint4*>(yp + (size_t)j * (2 * C)) = o;
        }
      }
    }
  }
  __syncthreads();   // the output tile is read out before the next halo tile is written over it
  tx = txn; tyb = tybn; b = bn;
  }
}

// ---- the same conv, marching: one block per (image, 32-column strip, 32-channel slice) walks DOWN the strip in groups of
// 8 output rows over a 16-row ring of halo rows in LDS.  A tile-per-block 7x7 reads 14 input rows per 8 output rows, and
// those re-reads (L2 hits or not) go through the same ~10 B/clk a CU can pull: it ran at 2.7 TB/s of algorithmic traffic
// with the 3x3 at 4.1.  Marching reads every row once (only the 6 halo columns twice), and the next 8 rows are in flight in
// registers while the current group is in the MFMAs.  Ring row of image row r = (r - PAD) & 15; "unit" u = rows
// 8u + PAD .. 8u + PAD + 7 occupies ring rows 8 (u & 1) .. + 7; group g needs units g - 1 and g.
#ifndef DW_MARCH_OCC
#define DW_MARCH_OCC 2
#endif
// GRAD (round 5, the tower's backward): the same march as the INPUT-GRADIENT of a stride-1 depthwise conv -- x = dL/dy as fp16, ttab = the Toeplitz table of the
// FLIPPED taps as fp16 (correlation = convolution with the mirrored kernel under "same" padding), products on v_mfma_f32_4x4x4_16b_f16, no bias / GELU, the
// result (+ the fp16 residual `res`: dL/dx' = G + dw7^T(dL/dt)) written as fp16.  Everything between the loads and the stores moves 16-bit cells and does not care.
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
template <int K, bool GRAD = false>
__global__ __launch_bounds__(256, DW_MARCH_OCC) void dwconv_march_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ ttab,
                                                               const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                               int H, int W, int C, int gelu, int tiles_x, int nslices, const bf16_t* __restrict__ res = nullptr) {
  constexpr int TW = 32, PAD = K / 2, IW = TW + K - 1;
  constexpr int NQ = (IW + 3) / 4, NM = (K + 3 + 3) / 4;
  constexpr int RS = NQ * 256 + 64, RSO = 8 * 256 + 64;        // ring / output-tile row strides (== 64 mod 256)
  constexpr int T_BYTES = 2 * K * NM * 512;
#if DW_MARCH_OCC >= 3   // the output tile borrows the ring rows unit g - 1 has just vacated (one more barrier per group)
  __shared__ __attribute__((aligned(16))) char smem[16 * RS];
#else
  __shared__ __attribute__((aligned(16))) char smem[16 * RS + 8 * RSO];
#endif
  char* sX = smem;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int bid = xcd_remap(blockIdx.x, gridDim.x);     // slices and neighbouring strips of one image on one XCD
  const int slice = bid % nslices; bid /= nslices;
  const int tx = bid % tiles_x;
  const long b = bid / tiles_x;
  const int c0 = slice * 32;
  const int gg = wid & 1, rg = wid >> 1;          // wave = (16-channel group, 4 output rows)
  const int bch = lane >> 2, jr = lane & 3;       // lane = (channel within the group, row within the 4)
  const int ng = (H + 7) / 8;

  s16x4 afr[K][NM];
  {
    const char* tsrc = reinterpret_cast<const char*>(ttab) + (size_t)slice * T_BYTES + (size_t)gg * K * NM * 512 + lane * 8;
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int m = 0; m < NM; ++m) afr[ky][m] = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(tsrc + (ky * NM + m) * 512));
  }
  const float bv = GRAD ? 0.f : bias[c0 + gg * 16 + bch];

  // one unit = 8 rows x NQ quads x 4 channel groups of (4 pixels x 8 channels) tasks
  constexpr int NTASK = 8 * NQ * 4, TPT = (NTASK + 255) / 256;
  uint4 px[TPT][4];
#define DW_LOAD_UNIT(U)                                                                                          \
  {                                                                                                              \
    _Pragma("unroll") for (int tt = 0; tt < TPT; ++tt) {                                                         \
      const int task = tid + 256 * tt;                                                                           \
      const int cg = task & 3, quad = (task >> 2) % NQ, row = (task >> 2) / NQ;                                  \
      const int iy = 8 * (U) + PAD + row, ix0 = tx * TW - PAD + quad * 4;                                        \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                            \
        const int ix = ix0 + j;                                                                                  \
        px[tt][j] = (task < NTASK && iy >= 0 && iy < H && ix >= 0 && ix < W && quad * 4 + j < IW)                \
                        ? *reinterpret_cast<const uint4*>(x + (((size_t)b * H + iy) * W + ix) * C + c0 + cg * 8)  \
                        : make_uint4(0, 0, 0, 0);                                                                \
      }                                                                                                          \
    }                                                                                                            \
  }
#define DW_WRITE_UNIT(U)                                                                                         \
  {                                                                                                              \
    _Pragma("unroll") for (int tt = 0; tt < TPT; ++tt) {                                                         \
      const int task = tid + 256 * tt;                                                                           \
      if (task < NTASK) {                                                                                        \
        const int cg = task & 3, quad = (task >> 2) % NQ, row = (task >> 2) / NQ;                                \
        const uint32_t d[4][4] = {{px[tt][0].x, px[tt][0].y, px[tt][0].z, px[tt][0].w},                          \
                                  {px[tt][1].x, px[tt][1].y, px[tt][1].z, px[tt][1].w},                          \
                                  {px[tt][2].x, px[tt][2].y, px[tt][2].z, px[tt][2].w},                          \
                                  {px[tt][3].x, px[tt][3].y, px[tt][3].z, px[tt][3].w}};                         \
        const uint32_t dst = (uint32_t)((8 * ((U) & 1) + row) * RS + quad * 256 + cg * 64) |                     \
                             (uint32_t)((((quad & 3) << 1) | (cg >> 1)) << 3);                                   \
        _Pragma("unroll") for (int dd = 0; dd < 4; ++dd) {                                                       \
          uint2 ev, od;                                                                                          \
          ev.x = __builtin_amdgcn_perm(d[1][dd], d[0][dd], 0x05040100u);                                         \
          ev.y = __builtin_amdgcn_perm(d[3][dd], d[2][dd], 0x05040100u);                                         \
          od.x = __builtin_amdgcn_perm(d[1][dd], d[0][dd], 0x07060302u);                                         \
          od.y = __builtin_amdgcn_perm(d[3][dd], d[2][dd], 0x07060302u);                                         \
          *reinterpret_cast<uint2*>(sX + (dst ^ (uint32_t)((2 * dd) * 8))) = ev;                                 \
          *reinterpret_cast<uint2*>(sX + (dst ^ (uint32_t)((2 * dd + 1) * 8))) = od;                             \
        }                                                                                                        \
      }                                                                                                          \
    }                                                                                                            \
  }
  // lane's four swizzled channel offsets (one per quad & 3)
  uint32_t sw[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) sw[v] = (uint32_t)(((gg * 16 + bch) ^ ((v << 1) | gg)) << 3);

  DW_LOAD_UNIT(-1)
  DW_WRITE_UNIT(-1)
  DW_LOAD_UNIT(0)
  DW_WRITE_UNIT(0)
  if (ng > 1) DW_LOAD_UNIT(1)

  for (int g = 0; g < ng; ++g) {
    __syncthreads();   // unit g is in the ring; the previous group's output tile has been read
    f32x4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int rb = 8 * g + rg * 4 + jr - 2 * PAD;   // ring row (mod 16) of the lane's ky = 0 input row
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const uint32_t ro = (uint32_t)((rb + ky) & 15) * RS;
      s16x4 xq[NQ];
#pragma unroll
      for (int t = 0; t < NQ; ++t) xq[t] = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(sX + ro + sw[t & 3] + t * 256));
#pragma unroll
      for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (q + m < NQ) {
            if constexpr (GRAD) acc[q] = __builtin_amdgcn_mfma_f32_4x4x4f16(__builtin_bit_cast(f16x4_t, afr[ky][m]), __builtin_bit_cast(f16x4_t, xq[q + m]), acc[q], 0, 0, 0);
            else acc[q] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(afr[ky][m], xq[q + m], acc[q], 0, 0, 0);
          }
    }
    __syncthreads();   // every wave is done with unit g - 1: its ring rows take unit g + 1
#if DW_MARCH_OCC >= 3
    char* sO = sX + 8 * ((g + 1) & 1) * RS;
#else
    char* sO = smem + 16 * RS;
    if (g + 1 < ng) DW_WRITE_UNIT(g + 1)
    if (g + 2 < ng) DW_LOAD_UNIT(g + 2)
#endif
    // ---- bias (+GELU), through the output tile in the same cell layout, NHWC transpose by v_perm on the reading side
    {
      char* orow = sO + (rg * 4 + jr) * RSO;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = acc[q][i] + bv;
          if (gelu) v[i] = gelu_f(v[i]);
        }
        uint2 u;
        if constexpr (GRAD) { u.x = pack_h2(v[0], v[1]); u.y = pack_h2(v[2], v[3]); }
        else { u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); }
        *reinterpret_cast<uint2*>(orow + sw[q & 3] + q * 256) = u;
      }
    }
    __syncthreads();
    {
      const int cg = tid & 3, quad = (tid >> 2) & 7, row = tid >> 5;   // 8 rows x 8 quads x 4 channel groups
      const uint32_t src = (uint32_t)(row * RSO + quad * 256 + cg * 64) | (uint32_t)((((quad & 3) << 1) | (cg >> 1)) << 3);
      uint2 r[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = *reinterpret_cast<const uint2*>(sO + (src ^ (uint32_t)(e * 8)));
      const int oy = g * 8 + row, ox0 = tx * TW + quad * 4;
      if (oy < H) {
        bf16_t* yp = y + (((size_t)b * H + oy) * W + ox0) * C + c0 + cg * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (ox0 + j < W) {
            uint4 o;
            const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
#define FV_PX(A, B) __builtin_amdgcn_perm((j & 2) ? r[B].y : r[B].x, (j & 2) ? r[A].y : r[A].x, sel)
            o.x = FV_PX(0, 1); o.y = FV_PX(2, 3); o.z = FV_PX(4, 5); o.w = FV_PX(6, 7);
#undef FV_PX
            if constexpr (GRAD) {
              if (res) {   // + the residual branch's gradient (a second fp16 rounding of the sum: 2^-12)
                float a8[8], r8[8];
                unpack8_h(o, a8);
                unpack8_h(*reinterpret_cast<const uint4*>(res + (((size_t)b * H + oy) * W + ox0 + j) * C + c0 + cg * 8), r8);
#pragma unroll
                for (int e = 0; e < 8; ++e) a8[e] += r8[e];
                o = pack8_h(a8);
              }
            }
            *reinterpret_cast<uint4*>(yp + (size_t)j * C) = o;
          }
        }
      }
    }
#if DW_MARCH_OCC >= 3
    __syncthreads();   // the borrowed rows have been read
    if (g + 1 < ng) DW_WRITE_UNIT(g + 1)
    if (g + 2 < ng) DW_LOAD_UNIT(g + 2)
#endif
  }
#undef DW_LOAD_UNIT
#undef DW_WRITE_UNIT
}

// ---- RepMixer pair, marching: x' = dw3x3(x) (the reparameterised token mixer) and t = dw7x7(x') (the ConvFFN's conv) in
// ONE walk down the strip.  As two kernels the pair moves 4.5 tensor-passes through the CUs' ~10 B/clk load/store path
// (3x3: 1.33 in + 1 out, 7x7: 1.19 in + 1 out) and that path, not HBM or the MFMAs, is what bounds them; fused it is 3.2:
// x is read once (plus the 8 halo columns), x' and t are written once, and x' reaches the 7x7 through LDS in the MFMA
// operand layout the 3x3 produced it in -- no transposition in between.  Two 16-row rings: x rows (ring row (r - 4) & 15,
// "x unit" u = rows 8u + 4 .. 8u + 11) and x' rows (ring row (r - 3) & 15, "x' unit" u = rows 8u + 3 .. 8u + 10).  Step g:
// x' unit g from x units g - 1, g; emit its rows; t rows 8g .. 8g + 7 from x' units g - 1, g; emit them.  x' outside the
// map is stored as zero (it is the 7x7's zero padding, not the 3x3's response to in-map neighbours).
// LDS is what sets the occupancy: the x ring is 12 rows (row (r - 4) mod 12; 10 are live), the x' ring 16, and the t tile
// borrows the x' rows unit g - 1 vacates once group g's 7x7 is done -- 73.5 KB, two blocks per CU (at one, with three
// barriers a step and nothing else resident, the fused kernel was slower than the two it replaces).
#ifndef DP_DEFAULT_GEO
#define DP_DEFAULT_GEO 1
#endif
#ifndef DP_ABL
#define DP_ABL 0   // diagnostics (tools/dw_variants.sh; results are wrong): 1 no x loads, 2 no x' stores, 4 no t stores, 8 no MFMAs, 16 no emit at all
#endif
// Geometry by channels per block.  CHB = 32: a 32-column x 32-channel strip (64-byte runs per pixel: any C % 32 == 0).  CHB = 64:
// a 16-column x 64-channel strip -- every global access is a full 128-byte line.  tools/slice_micro.hip: a pure stream with this
// kernel's traffic (one tensor in, two out) tops out at 3.6-4.0 TB/s with 64-byte runs and reaches 4.9-5.1 TB/s with 128-byte ones.
// A ring row holds CHB / 32 planes of (NQ quads x 256 B): inside a plane everything is the 32-channel layout.  The rings are as
// short as liveness allows at CHB = 64 (x: 8 new rows + the 2 the next 3x3 still needs; x': 8 new + the 6 the next 7x7 needs;
// the t tile sits in the slots the NEXT x' unit will take) so that two blocks still share a CU.
template <int CHB, int TW_>
struct DpGeo {
  static constexpr int TW = TW_, NTQ = TW / 4, NQ = NTQ + 2, NEQ = NTQ + 1;
  static constexpr int NPL = CHB / 32, PLB = NQ * 256, RS = NPL * PLB + 64;
  static constexpr int XR = (CHB == 32 && TW == 32) ? 12 : 10, PR = (CHB == 32 && TW == 32) ? 16 : 14;
  static constexpr int LDS = (XR + PR) * RS;
  static constexpr int NCG = CHB / 8;      // 16-byte channel groups per pixel
  static constexpr int NGG = CHB / 16;     // 16-channel MFMA groups = waves across channels
  static constexpr int NRG = NGG / 2;      // 4-row groups one wave computes per 8-row unit
  static constexpr int OCC = 3 * LDS <= 160 * 1024 ? 3 : 2;   // blocks per CU the kernel is compiled for (registers: 512 / (4 * OCC / 4))
};
// (DP_ABL & 32, tools/dw_variants.sh only, results wrong: the step's four barriers removed -- an upper bound on what FEWER barriers per row, i.e. a 16-row march step, could buy)
#if DP_ABL & 32
#define DP_BARRIER asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
#define DP_BARRIER __syncthreads();
#endif
#ifdef DP_STAMPS
__device__ unsigned long long g_dp_stamps[512 * 16];   // diagnostics: per block (first 512), clocks per phase summed over the steps (wave 0)
#define DP_T(I) { const unsigned long long n_ = __builtin_readcyclecounter(); if (tid == 0) st_[I] += n_ - t0_; t0_ = __builtin_readcyclecounter(); }
#else
#define DP_T(I)
#endif
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
template <int N> __device__ __forceinline__ int dp_wrap(int v) { return v >= N ? v - N : v; }   // v in [0, 2N)
template <int CHB, int TW_>
__global__ __launch_bounds__(256, (DpGeo<CHB, TW_>::OCC)) void dwpair_march_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ t3,
                                                               const float* __restrict__ b3, const bf16_t* __restrict__ t7,
                                                               const float* __restrict__ b7, bf16_t* __restrict__ y1,
                                                               bf16_t* __restrict__ y2, int H, int W, int C, int tiles_x,
                                                               int nslices, int nseg) {
  using G = DpGeo<CHB, TW_>;
  constexpr int TW = G::TW, NQ = G::NQ, NTQ = G::NTQ, NEQ = G::NEQ, RS = G::RS, XR = G::XR, PR = G::PR, PLB = G::PLB;
  constexpr int NCG = G::NCG, NGG = G::NGG, NRG = G::NRG;
  extern __shared__ __attribute__((aligned(16))) char dp_smem[];
  char* sX = dp_smem;                 // x ring: row R (absolute, R = -12 ...) at slot (R + 120) % XR
  char* sP = dp_smem + XR * RS;       // x' ring: row R' at slot (R' + 112) % PR; the t tile borrows the slots of the next unit

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  // nseg > 1 (few strips: B <= 4): the march is cut into nseg row segments, each its own block.  Segment [g0, g1) starts one unit early (x' unit
  // g0 - 1 feeds the first t rows; its own rows are the previous segment's to store) -- every output row is computed by the same arithmetic as in
  // the uncut march, by exactly one block
  const int seg = bid % nseg; bid /= nseg;
  const int slice = bid % nslices; bid /= nslices;
  const int tx = bid % tiles_x;
  const long b = bid / tiles_x;
  const int c0 = slice * CHB;
  const int gg = wid % NGG, rg0 = (wid / NGG) * NRG;   // wave = (16-channel group, first of its NRG 4-row groups)
  const int bch = lane >> 2, jr = lane & 3;            // lane = (channel within the group, row within the 4)
  const int ng = (H + 7) / 8;
  const int g0 = (int)((long)seg * ng / nseg), g1 = (int)((long)(seg + 1) * ng / nseg);
  const uint32_t pl = (uint32_t)(gg >> 1) * PLB;       // the wave's plane inside a ring row

  s16x4 a3[3][2], a7[7][3];
  {
    const size_t grp = (size_t)(c0 / 16 + gg);
    const char* s3 = reinterpret_cast<const char*>(t3) + grp * (3 * 2 * 512) + lane * 8;
    const char* s7 = reinterpret_cast<const char*>(t7) + grp * (7 * 3 * 512) + lane * 8;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int m = 0; m < 2; ++m) a3[ky][m] = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(s3 + (ky * 2 + m) * 512));
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
      for (int m = 0; m < 3; ++m) a7[ky][m] = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(s7 + (ky * 3 + m) * 512));
  }
  const float bv3 = b3[c0 + gg * 16 + bch], bv7 = b7[c0 + gg * 16 + bch];

  // one x unit = 8 rows x NQ quads x NCG channel groups of (4 pixels x 8 channels) tasks; x columns start at tx*TW - 4
  constexpr int NTASK = 8 * NQ * NCG, TPT = (NTASK + 255) / 256;
  u32x4 px[TPT][4];
  // Global accesses go through buffer descriptors with 32-bit per-lane offsets: a lane's STATIC part (image, its task's row and
  // column, channel slice) is computed once, a step adds one scalar, and anything outside the map becomes an out-of-range offset the
  // descriptor drops (loads return 0 = the zero padding).  As 64-bit pointers every access carried a u64 multiply chain and an
  // exec-mask branch: 20 of them per thread and step.  COLOOB + any step offset stays in [tensor bytes, 2^32) (launcher's guard).
  constexpr uint32_t COLOOB = 0x7FFFFFF0u, ROWOOB = 0x80000000u;
  const uint32_t rowbytes = (uint32_t)W * (uint32_t)C * 2u;
  const uint32_t tbytes = (uint32_t)((size_t)gridDim.x / ((size_t)tiles_x * nslices * nseg) * H * rowbytes);
  // x loads are inline asm: vmcnt counts loads and stores in one in-order queue, and hipcc, unable to count the stores of a step
  // across the loop's branches, waited vmcnt(0) for the x unit -- i.e. for the write acknowledgements of every store issued after
  // those loads.  The loads are a full step old when they are needed; DP_WAIT_X leaves the younger stores outstanding.
  const uint64_t xa_ = (uint64_t)reinterpret_cast<uintptr_t>(x);
  const i32x4 xrsrc = {(int)__builtin_amdgcn_readfirstlane((uint32_t)xa_), (int)__builtin_amdgcn_readfirstlane((uint32_t)(xa_ >> 32) & 0xffffu),
                       (int)__builtin_amdgcn_readfirstlane(tbytes), 0x00020000};
  const __amdgpu_buffer_rsrc_t y1rsrc = __builtin_amdgcn_make_buffer_rsrc(y1, 0, tbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t y2rsrc = __builtin_amdgcn_make_buffer_rsrc(y2, 0, tbytes, 0x00020000);
  const uint32_t img = (uint32_t)b * (uint32_t)H * rowbytes + (uint32_t)c0 * 2u;
  uint32_t xo[TPT][4];     // x unit 0: row 4 + task row, columns tx*TW - 4 + quad*4 + j
  uint32_t xdst[TPT];      // LDS position inside a ring row
  int xrow[TPT];
#pragma unroll
  for (int tt = 0; tt < TPT; ++tt) {
    const int task = tid + 256 * tt;
    const int cg = task % NCG, quad = (task / NCG) % NQ, row = (task / NCG) / NQ;
    xrow[tt] = row + 4;
    xdst[tt] = (uint32_t)((cg >> 2) * PLB + quad * 256 + (cg & 3) * 64) | (uint32_t)((((quad & 3) << 1) | ((cg & 3) >> 1)) << 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ix = tx * TW - 4 + quad * 4 + j;
      xo[tt][j] = (!(DP_ABL & 1) && task < NTASK && ix >= 0 && ix < W)
                      ? img + (uint32_t)(row + 4) * rowbytes + (uint32_t)ix * (uint32_t)C * 2u + (uint32_t)cg * 16u
                      : COLOOB;
    }
  }
#define DP_LOAD_XUNIT(U)                                                                                         \
  {                                                                                                              \
    const uint32_t so_ = (uint32_t)(8 * (U)) * rowbytes;   /* modular: U may be negative */                      \
    _Pragma("unroll") for (int tt = 0; tt < TPT; ++tt) {                                                         \
      if (tt == 0 || wid < (NTASK - 256 * tt + 63) / 64) {   /* wave-uniform: the last tasks live in the first waves */ \
        const bool ok_ = (unsigned)(xrow[tt] + 8 * (U)) < (unsigned)H;                                           \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                          \
          const uint32_t v_ = ok_ ? xo[tt][j] + so_ : ROWOOB;                                                    \
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(px[tt][j]) : "v"(v_), "s"(xrsrc) : "memory"); \
        }                                                                                                        \
      }                                                                                                          \
    }                                                                                                            \
  }
#define DP_WAIT_X(N)                                                                                             \
  if constexpr (TPT == 2)                                                                                        \
    asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(px[0][0]), "+v"(px[0][1]), "+v"(px[0][2]), "+v"(px[0][3]),    \
                 "+v"(px[TPT - 1][0]), "+v"(px[TPT - 1][1]), "+v"(px[TPT - 1][2]), "+v"(px[TPT - 1][3]) :: "memory"); \
  else                                                                                                           \
    asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(px[0][0]), "+v"(px[0][1]), "+v"(px[0][2]), "+v"(px[0][3]) :: "memory");
#define DP_WRITE_XUNIT(U)                                                                                        \
  {                                                                                                              \
    _Pragma("unroll") for (int tt = 0; tt < TPT; ++tt) {                                                         \
      const int task = tid + 256 * tt;                                                                           \
      if (task < NTASK) {                                                                                        \
        const uint32_t d[4][4] = {{px[tt][0].x, px[tt][0].y, px[tt][0].z, px[tt][0].w},                          \
                                  {px[tt][1].x, px[tt][1].y, px[tt][1].z, px[tt][1].w},                          \
                                  {px[tt][2].x, px[tt][2].y, px[tt][2].z, px[tt][2].w},                          \
                                  {px[tt][3].x, px[tt][3].y, px[tt][3].z, px[tt][3].w}};                         \
        const int xr_ = (8 * (U) + xrow[tt] + 120) % XR;                                                         \
        const uint32_t dst = (uint32_t)(xr_ * RS) + xdst[tt];   /* RS is a multiple of 64: the low bits stay xdst's */ \
        _Pragma("unroll") for (int dd = 0; dd < 4; ++dd) {                                                       \
          uint2 ev, od;                                                                                          \
          ev.x = __builtin_amdgcn_perm(d[1][dd], d[0][dd], 0x05040100u);                                         \
          ev.y = __builtin_amdgcn_perm(d[3][dd], d[2][dd], 0x05040100u);                                         \
          od.x = __builtin_amdgcn_perm(d[1][dd], d[0][dd], 0x07060302u);                                         \
          od.y = __builtin_amdgcn_perm(d[3][dd], d[2][dd], 0x07060302u);                                         \
          *reinterpret_cast<uint2*>(sX + (dst ^ (uint32_t)((2 * dd) * 8))) = ev;                                 \
          *reinterpret_cast<uint2*>(sX + (dst ^ (uint32_t)((2 * dd + 1) * 8))) = od;                             \
        }                                                                                                        \
      }                                                                                                          \
    }                                                                                                            \
  }
  constexpr int NET = 8 * NEQ * NCG;   // x' emit tasks: 8 rows x the NEQ quads that hold strip columns x NCG channel groups
  constexpr int NEP = (NET + 255) / 256, NTT = 8 * NTQ * NCG;   // x' emit task slots per thread; t emit tasks (<= 256)
  static_assert(NEP <= 2 && NTT <= 256, "emit task maps");
  uint32_t y1o[NEP][4], y2o[4];   // static parts of the x' / t store offsets (step g = 0)
  uint32_t e1src[NEP], e2src;     // their LDS positions inside a ring row
  int e1row[NEP], e2row;
#pragma unroll
  for (int et = 0; et < NEP; ++et) {
    const int task = tid + 256 * et;
    const int cg = task % NCG, quad = (task / NCG) % NEQ, row = (task / NCG) / NEQ;
    e1row[et] = row;
    e1src[et] = (uint32_t)((cg >> 2) * PLB + quad * 256 + (cg & 3) * 64) | (uint32_t)((((quad & 3) << 1) | ((cg & 3) >> 1)) << 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int lc = 4 * quad - 3 + j, ox = tx * TW + lc;
      y1o[et][j] = (!(DP_ABL & 2) && task < NET && lc >= 0 && lc < TW && ox < W)
                       ? img + (uint32_t)(row + 3) * rowbytes + (uint32_t)ox * (uint32_t)C * 2u + (uint32_t)cg * 16u
                       : COLOOB;
    }
  }
  {
    const int cg = tid % NCG, quad = (tid / NCG) % NTQ, row = tid / (NCG * NTQ);
    e2row = row;
    e2src = (uint32_t)((cg >> 2) * PLB + quad * 256 + (cg & 3) * 64) | (uint32_t)((((quad & 3) << 1) | ((cg & 3) >> 1)) << 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ox = tx * TW + quad * 4 + j;
      y2o[j] = (!(DP_ABL & 4) && tid < NTT && ox < W) ? img + (uint32_t)row * rowbytes + (uint32_t)ox * (uint32_t)C * 2u + (uint32_t)cg * 16u : COLOOB;
    }
  }
  uint32_t sw[4];   // lane's four swizzled channel offsets (one per quad & 3), plane included
#pragma unroll
  for (int v = 0; v < 4; ++v) sw[v] = pl + (uint32_t)((((gg & 1) * 16 + bch) ^ ((v << 1) | (gg & 1))) << 3);

  static_assert(TPT <= 2, "DP_WAIT_X names the first and the last task slot");
  // x unit -2 (rows -12 .. -5) is never written: step -1 reads its last two rows, but only into x' rows < 0, which are stored as
  // zero whatever the 3x3 saw.  (Writing it raced with unit -1: their ring slots overlap and no barrier separates the two writes.)
  // (a later segment's warm-up unit g0 - 1 would need the last two rows of x unit g0 - 2 for its first two x' rows: those rows, 8 g0 - 5 and - 4, are
  // neither stored nor read by this segment's 7x7, which starts at x' row 8 g0 - 3)
  DP_LOAD_XUNIT(g0 - 1)
  DP_WAIT_X(0)
  DP_WRITE_XUNIT(g0 - 1)
  DP_LOAD_XUNIT(g0)
#define DP_DUMMY_STORES(N) _Pragma("unroll") for (int z_ = 0; z_ < (N); ++z_) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" :: "v"(px[0][0]), "v"(ROWOOB), "s"(xrsrc) : "memory");
  DP_DUMMY_STORES(8)

#ifdef DP_STAMPS
  unsigned long long st_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0_ = __builtin_readcyclecounter();
#endif
  for (int g = g0 - 1; g < g1; ++g) {
    DP_T(11)
    DP_BARRIER   // x unit g is in its ring; the previous step's output tile and x' rows have been read
    DP_T(0)
    // ---- x' unit g = dw3x3 over x units g - 1, g: lane's row r' = 8g + 3 + rg*4 + jr, NQ quads of columns.
    // The operand rows (one per (row group, ky): NQ ds_read_b64) are requested TWO rows ahead of the MFMAs that take them and the
    // order is pinned: left to itself hipcc reads a row 3-10 MFMAs before its use and the MFMAs then run at LDS latency (19-25 clk
    // each in the stamps, for an instruction that issues every 8-11).
    {
      constexpr int NR3 = NRG * 3;
      s16x4 xr[3][NQ];
      f32x4 acc[NRG][NQ];
      int rb3[NRG];
#pragma unroll
      for (int rgi = 0; rgi < NRG; ++rgi) {
        rb3[rgi] = (8 * g + 2 + (rg0 + rgi) * 4 + jr + 120) % XR;   // x ring slot of ky = 0 (row r' - 1)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[rgi][q] = f32x4{bv3, bv3, bv3, bv3};   // the bias rides in as the accumulators' start
      }
#define DP_RD3(R, DST)                                                                                              \
      {                                                                                                             \
        const uint32_t ro_ = (uint32_t)dp_wrap<XR>(rb3[(R) / 3] + (R) % 3) * RS;                                    \
        _Pragma("unroll") for (int t = 0; t < NQ; ++t)                                                              \
          DST[t] = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(sX + ro_ + sw[t & 3] + t * 256));      \
      }
      DP_RD3(0, xr[0])
      if (NR3 > 1) DP_RD3(1, xr[1])
#pragma unroll
      for (int r = 0; r < NR3; ++r) {
        if (r + 2 < NR3) DP_RD3(r + 2, xr[(r + 2) % 3])
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            if (q + m < NQ && !(DP_ABL & 8)) acc[r / 3][q] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a3[r % 3][m], xr[r % 3][q + m], acc[r / 3][q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#undef DP_RD3
#pragma unroll
      for (int rgi = 0; rgi < NRG; ++rgi) {
        const int r1 = 8 * g + 3 + (rg0 + rgi) * 4 + jr;
        const bool rowok = r1 >= 0 && r1 < H;
        char* prow = sP + (uint32_t)((r1 + 112) % PR) * RS;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          float v[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int col = tx * TW + 4 * q - 3 + i;
            v[i] = (rowok && col >= 0 && col < W) ? acc[rgi][q][i] : 0.0f;
          }
          uint2 u;
          u.x = pack_bf2(v[0], v[1]);
          u.y = pack_bf2(v[2], v[3]);
          *reinterpret_cast<uint2*>(prow + sw[q & 3] + q * 256) = u;
        }
      }
    }
    DP_T(1)
    DP_BARRIER   // x' unit g is in its ring; every wave is done with x unit g - 1
    DP_T(2)
    if (g + 1 < g1) {
      // x unit g + 1 was requested a step ago; younger than it are the 4 x' and 4 t stores of step g - 1 (the prologue and step -1
      // issue dropped out-of-range stores in their place so that ONE wait count fits every step)
      DP_WAIT_X(8)
      DP_WRITE_XUNIT(g + 1)
    }
    DP_T(3)
    if (g + 2 < g1) DP_LOAD_XUNIT(g + 2)
    DP_T(4)
    // ---- emit x' rows 8g + 3 .. 8g + 10 (columns of this strip only), v_perm transpose back to channel-contiguous pixels
    {
      const uint32_t so1 = (uint32_t)(8 * g) * rowbytes;
#pragma unroll
      for (int et = 0; et < NEP; ++et) {
        // every wave that owns a task of this slot issues all four stores (the vmcnt of DP_WAIT_X counts them); lanes past the
        // last task read a ring row that exists and store to an out-of-range offset
        if ((et == 0 || wid < (NET - 256 + 63) / 64) && !(DP_ABL & 16)) {
          const uint32_t src = (uint32_t)((8 * g + 3 + e1row[et] + 112) % PR) * RS + e1src[et];
          uint2 r[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) r[e] = *reinterpret_cast<const uint2*>(sP + (src ^ (uint32_t)(e * 8)));
          const bool ok = (unsigned)(8 * g + 3 + e1row[et]) < (unsigned)H && (g >= g0 || g0 == 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            u32x4 o;
            const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
#define FV_PX(A, B) __builtin_amdgcn_perm((j & 2) ? r[B].y : r[B].x, (j & 2) ? r[A].y : r[A].x, sel)
            o.x = FV_PX(0, 1); o.y = FV_PX(2, 3); o.z = FV_PX(4, 5); o.w = FV_PX(6, 7);
#undef FV_PX
            __builtin_amdgcn_raw_buffer_store_b128(o, y1rsrc, ok ? y1o[et][j] + so1 : ROWOOB, 0, 0);
          }
        }
      }
    }
    DP_T(5)
#ifdef DP_NO_T7   /* tools/dw_variants.sh only (VERDICT r5 #1's table): the pair WITHOUT its 7x7 half -- x -> x' alone, what the march costs once t is made by the ConvFFN */
    { DP_DUMMY_STORES(4) continue; }
#endif
    if (g < g0) { DP_DUMMY_STORES(4) continue; }
    // ---- t rows 8g .. 8g + 7 = dw7x7 over x' units g - 1, g
    {
      f32x4 acc[NRG][NTQ];
      constexpr int NR7 = NRG * 7;
      s16x4 xr[3][NQ];
      int rb7[NRG];
#pragma unroll
      for (int rgi = 0; rgi < NRG; ++rgi) {
#pragma unroll
        for (int q = 0; q < NTQ; ++q) acc[rgi][q] = f32x4{bv7, bv7, bv7, bv7};
        rb7[rgi] = (8 * g + (rg0 + rgi) * 4 + jr - 3 + 112) % PR;     // x' ring slot of ky = 0 (row o - 3)
      }
#define DP_RD7(R, DST)                                                                                              \
      {                                                                                                             \
        const uint32_t ro_ = (uint32_t)dp_wrap<PR>(rb7[(R) / 7] + (R) % 7) * RS;                                    \
        _Pragma("unroll") for (int t = 0; t < NQ; ++t)                                                              \
          DST[t] = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2*>(sP + ro_ + sw[t & 3] + t * 256));      \
      }
      DP_RD7(0, xr[0])
      DP_RD7(1, xr[1])
#pragma unroll
      for (int r = 0; r < NR7; ++r) {   // same two-rows-ahead operand schedule as the 3x3
        if (r + 2 < NR7) DP_RD7(r + 2, xr[(r + 2) % 3])
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int q = 0; q < NTQ; ++q)
            if (!(DP_ABL & 8)) acc[r / 7][q] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a7[r % 7][m], xr[r % 3][q + m], acc[r / 7][q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#undef DP_RD7
      DP_T(6)
      DP_BARRIER   // every wave is done with x' unit g - 1: the slots the next unit will take carry the t tile out
      DP_T(7)
#pragma unroll
      for (int rgi = 0; rgi < NRG; ++rgi) {
        char* orow = sP + (uint32_t)((8 * g + 11 + (rg0 + rgi) * 4 + jr + 112) % PR) * RS;
#pragma unroll
        for (int q = 0; q < NTQ; ++q) {
          uint2 u;
          u.x = pack_bf2(acc[rgi][q][0], acc[rgi][q][1]);
          u.y = pack_bf2(acc[rgi][q][2], acc[rgi][q][3]);
          *reinterpret_cast<uint2*>(orow + sw[q & 3] + q * 256) = u;
        }
      }
    }
    DP_T(8)
    DP_BARRIER
    DP_T(9)
    {
      const uint32_t src = (uint32_t)((8 * g + 11 + e2row + 112) % PR) * RS + e2src;
      uint2 r[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = *reinterpret_cast<const uint2*>(sP + (src ^ (uint32_t)(e * 8)));
      if (!(DP_ABL & 16)) {
        const bool ok = g * 8 + e2row < H;
        const uint32_t so2 = (uint32_t)(8 * g) * rowbytes;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          u32x4 o;
          const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
#define FV_PX(A, B) __builtin_amdgcn_perm((j & 2) ? r[B].y : r[B].x, (j & 2) ? r[A].y : r[A].x, sel)
          o.x = FV_PX(0, 1); o.y = FV_PX(2, 3); o.z = FV_PX(4, 5); o.w = FV_PX(6, 7);
#undef FV_PX
          __builtin_amdgcn_raw_buffer_store_b128(o, y2rsrc, ok ? y2o[j] + so2 : ROWOOB, 0, 0);
        }
      }
    }
    DP_T(10)
  }
#ifdef DP_STAMPS
  if (tid == 0 && blockIdx.x < 512) { for (int z = 0; z < 12; ++z) g_dp_stamps[blockIdx.x * 16 + z] = st_[z]; g_dp_stamps[blockIdx.x * 16 + 12] = (unsigned long long)ng + 1; }
#endif
#undef DP_LOAD_XUNIT
#undef DP_WRITE_XUNIT
#undef DP_WAIT_X
#undef DP_DUMMY_STORES
}

// ------------------------------------------------------------------------------------------------ LayerNormChannel
// one wave per row (pixel); C <= 2048, C % 8 == 0; two-pass in registers.
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bb, bf16_t* __restrict__ y,
                                                              int rows, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = C >> 3;
  float v[4][8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      unpack8(*reinterpret_cast<const uint4*>(x + row * C + ch * 8), v[i]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[i][e];
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (lane + 64 * i < nch) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; q += d * d; }
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (v[i][e] - mean) * rstd * w[ch * 8 + e] + bb[ch * 8 + e];
      *reinterpret_cast<uint4*>(y + row * C + ch * 8) = pack8(o);
    }
  }
}

// ------------------------------------------------------------------------------------------------ SE + GELU
__global__ __launch_bounds__(256) void se_pool_kernel(const bf16_t* __restrict__ x, float* __restrict__ pooled, int P,
                                                       int C) {
  // grid (C/8/16 ceil, B); thread = (8 channels, one of 16 interleaved pixel classes): 16 partial sums per channel, added in class order
  // (one observation used to walk its 256 pixels on 2 blocks: 70 us of a 5 ms control-loop step)
  __shared__ float part[16][16][8];
  const int cgl = threadIdx.x & 15, pr = threadIdx.x >> 4;
  const int ch = blockIdx.x * 16 + cgl, b = blockIdx.y;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (ch < (C >> 3)) {
    for (int p = pr; p < P; p += 16) {
      float v[8];
      unpack8(*reinterpret_cast<const uint4*>(x + ((size_t)b * P + p) * C + ch * 8), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[pr][cgl][e] = a[e];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x >> 3, e = threadIdx.x & 7, chn = blockIdx.x * 16 + c;
    if (chn < (C >> 3)) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += part[q][c][e];
      pooled[(size_t)b * C + chn * 8 + e] = t / (float)P;
    }
  }
}

// y[b][n] = act(sum_k x[b][k] W[n][k] + bias[n]); one wave per (b, n); act 0 = relu, 1 = sigmoid
__global__ __launch_bounds__(256) void se_fc_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ y, int N,
                                                     int K, int act) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
  if (n >= N) return;
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += x[(size_t)b * K + k] * W[(size_t)n * K + k];
  s = wave_sum(s) + bias[n];
  if (lane == 0) y[(size_t)b * N + n] = act == 0 ? fmaxf(s, 0.f) : sigmoid_f(s);
}

__global__ __launch_bounds__(256) void se_apply_gelu_kernel(const bf16_t* __restrict__ x, const float* __restrict__ sc,
                                                             bf16_t* __restrict__ y, int P, int C, long total_chunks) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total_chunks) return;
  const int nch = C >> 3;
  const int ch = (int)(i % nch);
  const long b = i / ((long)nch * P);
  float v[8];
  unpack8(*reinterpret_cast<const uint4*>(x + i * 8), v);
  const float* s = sc + b * C + ch * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e] * s[e]);
  *reinterpret_cast<uint4*>(y + i * 8) = pack8(v);
}

}  // namespace

#ifndef FV_TRY_RC
#define FV_TRY_RC(expr) do { const int rc_ = (expr); if (rc_ != FV_OK) return rc_; } while (0)
#endif
namespace {
// geometry of reference resize_with_pad (model/fastvlm_adapter.py:36-55) for one call; shared by the letterbox kernel and the stem
// that samples the source image itself
int lb_params(const void* img, int dtype, int B, int C, int Hin, int Win, int S, float pad_value, int letterbox, bf16_t* pix, LbParams& p) {
  if (B <= 0 || Hin <= 0 || Win <= 0 || S <= 0) return fv_fail(FV_ERR_ARG, "letterbox: empty shape");
  if (C != 1 && C != 3 && C != 4) return fv_fail(FV_ERR_ARG, "letterbox: C must be 1, 3 or 4 (got %d)", C);
  if (dtype != FV_F32 && dtype != FV_U8) return fv_fail(FV_ERR_ARG, "letterbox: dtype must be f32 or u8");
  p.img = img; p.pix = pix; p.dtype = dtype; p.B = B; p.C = C; p.Hin = Hin; p.Win = Win; p.S = S; p.pad = pad_value;
  if (letterbox) {
    // reference: ratio = max(W/S, H/S); resized = int(dim / ratio) in Python double arithmetic
    const double ratio = ((double)Win / S > (double)Hin / S) ? (double)Win / S : (double)Hin / S;
    p.rh = (int)((double)Hin / ratio);
    p.rw = (int)((double)Win / ratio);
    if (p.rh > S || p.rw > S || p.rh < 1 || p.rw < 1) return fv_fail(FV_ERR_ARG, "letterbox: degenerate resize %dx%d", p.rh, p.rw);
  } else {
    p.rh = S; p.rw = S;
  }
  p.pt = S - p.rh; p.pl = S - p.rw;
  p.sh = (float)Hin / (float)p.rh;
  p.sw = (float)Win / (float)p.rw;
  return FV_OK;
}
}  // namespace

int launch_letterbox(const void* img, int dtype, int B, int C, int Hin, int Win, int S, float pad_value, int letterbox,
                     bf16_t* pix, hipStream_t s) {
  if (!img || !pix) return fv_fail(FV_ERR_ARG, "letterbox: null pointer");
  LbParams p;
  FV_TRY_RC(lb_params(img, dtype, B, C, Hin, Win, S, pad_value, letterbox, pix, p));
  hipLaunchKernelGGL(letterbox_kernel<0>, dim3((S + 255) / 256, (S + LB_R - 1) / LB_R, B), dim3(256), 0, s, p, LbNorm{});
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

// letterbox + _maybe_normalize_imagenet (fastvlm_adapter.py:463-477).  range_heuristic != 0: the torchvision branch's `x.max() > 1.5 -> x / 255` (two launches:
// the maximum of the letterboxed batch, then the store; vmax_scratch = 4 device bytes); 0: the branch without it ((x - mean) / std alone)
int launch_letterbox_norm(const void* img, int dtype, int B, int C, int Hin, int Win, int S, float pad_value, int letterbox, const float* mean3,
                          const float* std3, int range_heuristic, unsigned* vmax_scratch, bf16_t* pix, hipStream_t s) {
  if (!img || !pix || !mean3 || !std3 || (range_heuristic && !vmax_scratch)) return fv_fail(FV_ERR_ARG, "letterbox_norm: null pointer");
  LbParams p;
  FV_TRY_RC(lb_params(img, dtype, B, C, Hin, Win, S, pad_value, letterbox, pix, p));
  LbNorm n{};
  for (int c = 0; c < 3; ++c) {
    if (!(std3[c] != 0.0f)) return fv_fail(FV_ERR_ARG, "letterbox_norm: std[%d] must be non-zero", c);
    n.mean[c] = mean3[c]; n.std[c] = std3[c];
  }
  n.vmax = vmax_scratch; n.heuristic = range_heuristic ? 1 : 0;
  const dim3 grid((S + 255) / 256, (S + LB_R - 1) / LB_R, B);
  if (range_heuristic) {
    FV_HIP_CHECK(hipMemsetAsync(vmax_scratch, 0, 4, s));   // ordered image 0 = below every float
    hipLaunchKernelGGL(letterbox_kernel<1>, grid, dim3(256), 0, s, p, n);
  }
  hipLaunchKernelGGL(letterbox_kernel<2>, grid, dim3(256), 0, s, p, n);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_stem_conv(const bf16_t* pix, const float* w, const float* bias, bf16_t* y, int B, int S, int Cout,
                     hipStream_t s) {
  if (!pix || !w || !bias || !y) return fv_fail(FV_ERR_ARG, "stem_conv: null pointer");
  if (B <= 0 || S < 2 || (S & 1) || Cout % 8 || Cout < 8 || Cout > 2048) return fv_fail(FV_ERR_ARG, "stem_conv: bad shape S=%d Cout=%d", S, Cout);
  const int So = S / 2, G = Cout / 8, WQ = (So + 3) / 4, per_block = 256 / G;
  const long quads = (long)B * So * WQ;
  const long blocks = (quads + per_block - 1) / per_block;
  hipLaunchKernelGGL(stem_conv_kernel, dim3((unsigned)blocks), dim3(256), 27 * Cout * sizeof(float), s, pix, w, bias, y, B, S, Cout);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

// [Cout][64] bf16 image of the stem weights for stem_mfma_kernel from the tap-major [27][Cout] fp32 layout
void stem_mfma_pack(const float* w27, float* out, int Cout) {
  for (int co = 0; co < Cout; ++co)
    for (int ks = 0; ks < 2; ++ks)
      for (int g = 0; g < 4; ++g)
        for (int e = 0; e < 8; ++e) {
          const int ky = 2 * ks + (g >> 1), kx = 2 * (g & 1) + (e >> 2), ch = e & 3;
          out[(size_t)co * 64 + ks * 32 + g * 8 + e] = (ky < 3 && kx < 3 && ch < 3) ? w27[(size_t)((ky * 3 + kx) * 3 + ch) * Cout + co] : 0.0f;
        }
}

int launch_stem_mfma(const bf16_t* pix, const bf16_t* wp, const float* bias, bf16_t* y, int B, int S, int Cout, hipStream_t s) {
  if (!pix || !wp || !bias || !y) return fv_fail(FV_ERR_ARG, "stem_mfma: null pointer");
  if (B <= 0 || S < 2 || (S & 1) || Cout % 16 || Cout > 128) return fv_fail(FV_ERR_UNSUPPORTED, "stem_mfma: bad shape S=%d Cout=%d", S, Cout);
  const int So = S / 2;
  const long ntiles = (long)B * So * ((So + 15) / 16);
  constexpr int TPW = 8;
  const long blocks = (ntiles + 4 * TPW - 1) / (4 * TPW);
  hipLaunchKernelGGL(stem_mfma_kernel<TPW>, dim3((unsigned)blocks), dim3(256), 0, s, pix, wp, bias, y, B, S, Cout, ntiles);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_dwconv(const bf16_t* x, const float* w, const float* bias, bf16_t* y, int B, int H, int W, int C, int k,
                  int stride, int mult, int gelu, hipStream_t s) {
  if (!x || !w || !bias || !y) return fv_fail(FV_ERR_ARG, "dwconv: null pointer");
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C * mult) % 8) return fv_fail(FV_ERR_ARG, "dwconv: bad shape C=%d mult=%d", C, mult);
  const int pad = k / 2;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  if (stride == 1 && mult == 1 && (k == 3 || k == 7) && W >= 32 && C % 32 == 0) {  // large maps: LDS-tiled variant
    const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8, nsl = C / 32;
    const long nblk = (long)B * tiles_x * tiles_y * nsl;
    if (nblk > 0x7fffffffL) return fv_fail(FV_ERR_ARG, "dwconv: grid too large");
    if (k == 7) hipLaunchKernelGGL(dwconv_tile_kernel<7>, dim3((unsigned)nblk), dim3(256), 0, s, x, w, bias, y, H, W, C, gelu, tiles_x, tiles_y, nsl);
    else hipLaunchKernelGGL(dwconv_tile_kernel<3>, dim3((unsigned)nblk), dim3(256), 0, s, x, w, bias, y, H, W, C, gelu, tiles_x, tiles_y, nsl);
    FV_HIP_CHECK(hipGetLastError());
    return FV_OK;
  }
  const int G = C * mult / 8, WQ = (Wo + 3) / 4;
  int GS = 1;
  for (int d = 1; d <= 16; ++d)
    if (G % d == 0) GS = d;  // largest divisor of G that is <= 16: the block's channel slice
  const int nslices = G / GS, qpb = 256 / GS;
  const long nquads = (long)B * Ho * WQ;
  const long blocks = (nquads + qpb - 1) / qpb * nslices;
  if (blocks > 0x7fffffffL) return fv_fail(FV_ERR_ARG, "dwconv: grid too large");
  const dim3 grid((unsigned)blocks);
#define FV_DW(K_, S_, M_)                                                                                        \
  if (k == K_ && stride == S_ && mult == M_) {                                                                   \
    hipLaunchKernelGGL((dwconv_kernel<K_, S_, M_>), grid, dim3(256), 0, s, x, w, bias, y, B, H, W, C, Ho, Wo, gelu, \
                       GS, nslices, nquads);                                                                     \
    FV_HIP_CHECK(hipGetLastError());                                                                             \
    return FV_OK;                                                                                                \
  }
  FV_DW(3, 1, 1) FV_DW(7, 1, 1) FV_DW(3, 2, 1) FV_DW(7, 2, 2) FV_DW(3, 1, 2)
#undef FV_DW
  return fv_fail(FV_ERR_UNSUPPORTED, "dwconv: unsupported k=%d stride=%d mult=%d", k, stride, mult);
}

// bf16 Toeplitz table for dwconv_mfma_kernel: [C/16][K][NM][16 ch][4 out cols i][4 in cols k] = w[ky][4m + k - i]
size_t dwconv_toeplitz_elems(int C, int k) { return (size_t)(C / 16) * k * ((k + 6) / 4) * 256; }
void dwconv_toeplitz_pack(const float* w_tapmajor, float* out, int C, int k) {
  const int NM = (k + 6) / 4;
  for (int g = 0; g < C / 16; ++g)
    for (int ky = 0; ky < k; ++ky)
      for (int m = 0; m < NM; ++m)
        for (int bch = 0; bch < 16; ++bch)
          for (int i = 0; i < 4; ++i)
            for (int kk = 0; kk < 4; ++kk) {
              const int kx = 4 * m + kk - i;
              out[((((size_t)(g * k + ky) * NM + m) * 16 + bch) * 4 + i) * 4 + kk] =
                  (kx >= 0 && kx < k) ? w_tapmajor[(size_t)(ky * k + kx) * C + g * 16 + bch] : 0.0f;
            }
}
// bf16 Toeplitz table for dwconv_s2_mfma_kernel: [C/16][e][7][4][16 ch][4 out cols i][4 in cols kk] = w[ky][4m + kk - 2i] of
// output channel 2 (16 g + ch) + e; w_tapmajor is [49][2C]
size_t dwconv_s2_toeplitz_elems(int C) { return (size_t)(C / 16) * 2 * 7 * 4 * 256; }
void dwconv_s2_toeplitz_pack(const float* w_tapmajor, float* out, int C) {
  for (int g = 0; g < C / 16; ++g)
    for (int e = 0; e < 2; ++e)
      for (int ky = 0; ky < 7; ++ky)
        for (int m = 0; m < 4; ++m)
          for (int bch = 0; bch < 16; ++bch)
            for (int i = 0; i < 4; ++i)
              for (int kk = 0; kk < 4; ++kk) {
                const int kx = 4 * m + kk - 2 * i;
                out[(((((size_t)(g * 2 + e) * 7 + ky) * 4 + m) * 16 + bch) * 4 + i) * 4 + kk] =
                    (kx >= 0 && kx < 7) ? w_tapmajor[(size_t)(ky * 7 + kx) * (2 * C) + 2 * (g * 16 + bch) + e] : 0.0f;
              }
}
bool dwconv_s2_mfma_supported(int H, int W, int C, int k, int stride, int mult) {
  return k == 7 && stride == 2 && mult == 2 && C % 32 == 0 && H % 2 == 0 && W % 2 == 0 && W >= 16;
}
int launch_dwconv_s2_mfma(const bf16_t* x, const bf16_t* ttab, const float* bias, bf16_t* y, int B, int H, int W, int C, int gelu,
                          hipStream_t s) {
  if (!x || !ttab || !bias || !y) return fv_fail(FV_ERR_ARG, "dwconv_s2_mfma: null pointer");
  if (B <= 0 || H <= 0 || !dwconv_s2_mfma_supported(H, W, C, 7, 2, 2)) return fv_fail(FV_ERR_UNSUPPORTED, "dwconv_s2_mfma: unsupported shape H=%d W=%d C=%d", H, W, C);
  const int Ho = H / 2, Wo = W / 2;
  const int tiles_x = (Wo + 15) / 16, tiles_y = (Ho + 7) / 8, nsl = C / 32;
  const long nblk = (long)B * tiles_x * tiles_y * nsl;
  if (nblk > 0x7fffffffL) return fv_fail(FV_ERR_ARG, "dwconv_s2_mfma: grid too large");
  // persistent: two blocks per CU, a whole number of blocks per channel slice
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  // grid = 8 XCDs x chains x slices, about two blocks per CU; a chain walks every chains-th tile of its XCD's tile range
  const long tiles = nblk / nsl, tpx = (tiles + 7) / 8;
  long chains = (2L * cus / 8) / nsl;
  if (chains < 1) chains = 1;
  if (chains > tpx) chains = tpx;
  hipLaunchKernelGGL(dwconv_s2_mfma_kernel, dim3((unsigned)(8 * chains * nsl)), dim3(256), 0, s, x, ttab, bias, y, H, W, C, Ho, Wo, gelu, tiles_x, tiles_y, nsl, B);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

bool dwconv_mfma_supported(int W, int C, int k, int stride, int mult) {
  return stride == 1 && mult == 1 && (k == 3 || k == 7) && W >= 16 && C % 32 == 0;   // (W = 16, the last stage's maps: half of a 32-column strip is masked, and it still beats the VALU kernel)
}

int launch_dwconv_mfma(const bf16_t* x, const bf16_t* ttab, const float* bias, bf16_t* y, int B, int H, int W, int C, int k,
                       int gelu, hipStream_t s) {
  if (!x || !ttab || !bias || !y) return fv_fail(FV_ERR_ARG, "dwconv_mfma: null pointer");
  if (B <= 0 || H <= 0 || !dwconv_mfma_supported(W, C, k, 1, 1)) return fv_fail(FV_ERR_UNSUPPORTED, "dwconv_mfma: unsupported shape W=%d C=%d k=%d", W, C, k);
  // 7x7: 16-row tiles (halo rows 22/16 instead of 14/8 of the tile, 22 % fewer loads and transposes) when the map is
  // tall enough to still give every CU several blocks; 3x3 keeps 8 rows (its halo is small, occupancy matters more)
#ifndef DW_TH7
#define DW_TH7 8
#endif
  const int th = (k == 7 && H >= 32) ? DW_TH7 : 8;
  const int tiles_x = (W + 31) / 32, tiles_y = (H + th - 1) / th, nsl = C / 32;
  const long nblk = (long)B * tiles_x * tiles_y * nsl;
  if (nblk > 0x7fffffffL) return fv_fail(FV_ERR_ARG, "dwconv_mfma: grid too large");
#ifndef DW_NO_MARCH
  if (k == 7 && H >= 16) {   // marching strips (see dwconv_march_kernel); short maps keep the tile kernel
    const long nstrips = (long)B * tiles_x * nsl;
    hipLaunchKernelGGL((dwconv_march_kernel<7>), dim3((unsigned)nstrips), dim3(256), 0, s, x, ttab, bias, y, H, W, C, gelu, tiles_x, nsl);
    FV_HIP_CHECK(hipGetLastError());
    return FV_OK;
  }
#ifdef DW_MARCH3
  if (k == 3 && H >= 16) {
    const long nstrips = (long)B * tiles_x * nsl;
    hipLaunchKernelGGL((dwconv_march_kernel<3>), dim3((unsigned)nstrips), dim3(256), 0, s, x, ttab, bias, y, H, W, C, gelu, tiles_x, nsl);
    FV_HIP_CHECK(hipGetLastError());
    return FV_OK;
  }
#endif
#endif
  if (k == 7 && th == 16) hipLaunchKernelGGL((dwconv_mfma_kernel<7, 16>), dim3((unsigned)nblk), dim3(256), 0, s, x, ttab, bias, y, H, W, C, gelu, tiles_x, tiles_y, nsl);
  else if (k == 7) hipLaunchKernelGGL((dwconv_mfma_kernel<7, 8>), dim3((unsigned)nblk), dim3(256), 0, s, x, ttab, bias, y, H, W, C, gelu, tiles_x, tiles_y, nsl);
  else hipLaunchKernelGGL((dwconv_mfma_kernel<3, 8>), dim3((unsigned)nblk), dim3(256), 0, s, x, ttab, bias, y, H, W, C, gelu, tiles_x, tiles_y, nsl);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

// input gradient of a stride-1 depthwise k x k conv on the marching MFMA kernel: dy fp16 (B,H,W,C), ttab16 = fp16 Toeplitz table of the FLIPPED taps
// (dwconv_toeplitz_pack of w[k-1-ky][k-1-kx]), res = optional fp16 residual added to the result; dx fp16
bool dw_dgrad_mfma_supported(int H, int W, int C, int k) { return dwconv_mfma_supported(W, C, k, 1, 1) && H >= 16; }
int launch_dw_dgrad_mfma(const bf16_t* dy, const bf16_t* ttab16, const bf16_t* res, bf16_t* dx, int B, int H, int W, int C, int k, hipStream_t s) {
  if (!dy || !ttab16 || !dx) return fv_fail(FV_ERR_ARG, "dw_dgrad_mfma: null pointer");
  if (B <= 0 || !dw_dgrad_mfma_supported(H, W, C, k)) return fv_fail(FV_ERR_UNSUPPORTED, "dw_dgrad_mfma: unsupported shape H=%d W=%d C=%d k=%d", H, W, C, k);
  const int tiles_x = (W + 31) / 32, nsl = C / 32;
  const long nstrips = (long)B * tiles_x * nsl;
  if (nstrips > 0x7fffffffL) return fv_fail(FV_ERR_ARG, "dw_dgrad_mfma: grid too large");
  if (k == 7) hipLaunchKernelGGL((dwconv_march_kernel<7, true>), dim3((unsigned)nstrips), dim3(256), 0, s, dy, ttab16, static_cast<const float*>(nullptr), dx, H, W, C, 0, tiles_x, nsl, res);
  else hipLaunchKernelGGL((dwconv_march_kernel<3, true>), dim3((unsigned)nstrips), dim3(256), 0, s, dy, ttab16, static_cast<const float*>(nullptr), dx, H, W, C, 0, tiles_x, nsl, res);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

// pix (B,S,S,4) bf16 -> y (B,S/4,S/4,96) bf16: the first two stem convolutions with their GELUs (see stem_fused_kernel);
// wp = stem_mfma_pack image of the first, w2 / b2 = tap-major [9][96] fp32 weights and bias of the depthwise second
bool stem_fused_supported(int S, int C0) { return C0 == SF_C && S >= 8 && S % 4 == 0; }
int launch_stem_fused(const bf16_t* pix, const bf16_t* wp, const float* b1, const float* w2, const float* b2, bf16_t* y, int B,
                      int S, int C0, hipStream_t s) {
  if (!pix || !wp || !b1 || !w2 || !b2 || !y) return fv_fail(FV_ERR_ARG, "stem_fused: null pointer");
  if (B <= 0 || !stem_fused_supported(S, C0)) return fv_fail(FV_ERR_UNSUPPORTED, "stem_fused: bad shape S=%d C0=%d", S, C0);
  static bool attr_set = false;
  if (!attr_set) {
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SF_LDS));
    attr_set = true;
  }
  const int S2 = S / 4;
  const long ntiles = (long)B * ((S2 + SF_TC - 1) / SF_TC) * ((S2 + SF_TR - 1) / SF_TR);
  if (ntiles > 0x3fffffffL) return fv_fail(FV_ERR_ARG, "stem_fused: too many tiles");
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  const long blocks = ntiles < cus ? ntiles : cus;
  hipLaunchKernelGGL(stem_fused_kernel<false>, dim3((unsigned)blocks), dim3(512), SF_LDS, s, pix, wp, b1, w2, b2, y, B, S, ntiles, LbParams{});
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

// the same stem reading the SOURCE images (B,C,Hin,Win) f32 | u8 through the letterbox arithmetic (fv_preprocess fused in)
int launch_stem_fused_lb(const void* img, int dtype, int B, int C, int Hin, int Win, float pad_value, int letterbox, const bf16_t* wp,
                         const float* b1, const float* w2, const float* b2, bf16_t* y, int S, int C0, hipStream_t s) {
  if (!img || !wp || !b1 || !w2 || !b2 || !y) return fv_fail(FV_ERR_ARG, "stem_fused_lb: null pointer");
  if (B <= 0 || !stem_fused_supported(S, C0)) return fv_fail(FV_ERR_UNSUPPORTED, "stem_fused_lb: bad shape S=%d C0=%d", S, C0);
  LbParams p;
  FV_TRY_RC(lb_params(img, dtype, B, C, Hin, Win, S, pad_value, letterbox, nullptr, p));
  static bool attr_set = false;
  if (!attr_set) {
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SF_LDS));
    attr_set = true;
  }
  const int S2 = S / 4;
  const long ntiles = (long)B * ((S2 + SF_TC - 1) / SF_TC) * ((S2 + SF_TR - 1) / SF_TR);
  if (ntiles > 0x3fffffffL) return fv_fail(FV_ERR_ARG, "stem_fused_lb: too many tiles");
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  const long blocks = ntiles < cus ? ntiles : cus;
  hipLaunchKernelGGL(stem_fused_kernel<true>, dim3((unsigned)blocks), dim3(512), SF_LDS, s, nullptr, wp, b1, w2, b2, y, B, S, ntiles, p);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

// x (B,H,W,C) -> y1 = dw3x3(x) + b3 and y2 = dw7x7(y1) + b7 in one marching kernel (dwpair_march_kernel); t3 / t7 are the
// Toeplitz tables of dwconv_toeplitz_pack for k = 3 / 7.  x must not alias y1 or y2.
// (the kernel addresses its tensors through buffer descriptors with 32-bit offsets: B*H*W*C*2 bytes plus two units of rows must
// stay below its out-of-range markers; larger maps take the two separate kernels)
bool dwconv_pair_supported(int B, int H, int W, int C) {
  return H >= 16 && W >= 32 && C % 32 == 0 && ((long)B * H + 16) * W * C * 2 <= 0x7FFFFFF0L;
}
int launch_dwconv_pair(const bf16_t* x, const bf16_t* t3, const float* b3, const bf16_t* t7, const float* b7, bf16_t* y1, bf16_t* y2,
                       int B, int H, int W, int C, hipStream_t s) {
  if (!x || !t3 || !b3 || !t7 || !b7 || !y1 || !y2) return fv_fail(FV_ERR_ARG, "dwconv_pair: null pointer");
  if (B <= 0 || !dwconv_pair_supported(B, H, W, C)) return fv_fail(FV_ERR_UNSUPPORTED, "dwconv_pair: unsupported shape H=%d W=%d C=%d", H, W, C);
  if (x == y1 || x == y2 || y1 == y2) return fv_fail(FV_ERR_ARG, "dwconv_pair: buffers must be distinct");
  static bool attr_set = false;
  if (!attr_set) {
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwpair_march_kernel<32, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, DpGeo<32, 32>::LDS));
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwpair_march_kernel<64, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, DpGeo<64, 16>::LDS));
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwpair_march_kernel<32, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, DpGeo<32, 16>::LDS));
#if DP_DEFAULT_GEO == 3
    FV_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dwpair_march_kernel<64, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, DpGeo<64, 8>::LDS));
#endif
    attr_set = true;
  }
  // geometry: 0 = 32 columns x 32 channels (two blocks per CU), 1 = 16 x 64 (full cache lines), 2 = 16 x 32 (three blocks per CU)
  // (3, tools/dw_variants.sh only: 8 columns x 64 channels -- full cache lines AND three blocks per CU at 50.7 KB, for twice the halo columns per output column)
  int geo = DP_DEFAULT_GEO;
  if ((geo == 1 || geo == 3) && C % 64) geo = 0;
  const int chb = (geo == 1 || geo == 3) ? 64 : 32, tw = geo == 0 ? 32 : geo == 3 ? 8 : 16;
  const int tiles_x = (W + tw - 1) / tw, nsl = C / chb;
  long nstrips = (long)B * tiles_x * nsl;
  if (nstrips > 0x7fffffffL) return fv_fail(FV_ERR_ARG, "dwconv_pair: grid too large");
  // few strips (one observation: 24 per launch): row segments of >= 2 units until ~2 blocks per CU (each segment re-walks one unit)
  int nseg = 1;
  if (nstrips < 256) {
    const int ng = (H + 7) / 8;
    nseg = (int)((512 + nstrips - 1) / nstrips);
    if (nseg > ng / 2) nseg = ng / 2;
    if (nseg < 1) nseg = 1;
  }
  nstrips *= nseg;
#if DP_DEFAULT_GEO == 3
  if (geo == 3)
    hipLaunchKernelGGL((dwpair_march_kernel<64, 8>), dim3((unsigned)nstrips), dim3(256), (DpGeo<64, 8>::LDS), s, x, t3, b3, t7, b7, y1, y2, H, W, C, tiles_x, nsl, nseg);
  else
#endif
  if (geo == 1)
    hipLaunchKernelGGL((dwpair_march_kernel<64, 16>), dim3((unsigned)nstrips), dim3(256), (DpGeo<64, 16>::LDS), s, x, t3, b3, t7, b7, y1, y2, H, W, C, tiles_x, nsl, nseg);
  else if (geo == 2)
    hipLaunchKernelGGL((dwpair_march_kernel<32, 16>), dim3((unsigned)nstrips), dim3(256), (DpGeo<32, 16>::LDS), s, x, t3, b3, t7, b7, y1, y2, H, W, C, tiles_x, nsl, nseg);
  else
    hipLaunchKernelGGL((dwpair_march_kernel<32, 32>), dim3((unsigned)nstrips), dim3(256), (DpGeo<32, 32>::LDS), s, x, t3, b3, t7, b7, y1, y2, H, W, C, tiles_x, nsl, nseg);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_layernorm_rows(const bf16_t* x, const float* w, const float* b, bf16_t* y, int rows, int C, float eps,
                          hipStream_t s) {
  if (!x || !w || !b || !y) return fv_fail(FV_ERR_ARG, "layernorm_rows: null pointer");
  if (rows <= 0 || C % 8 || C > 2048 || C <= 0) return fv_fail(FV_ERR_ARG, "layernorm_rows: bad shape rows=%d C=%d", rows, C);
  hipLaunchKernelGGL(layernorm_rows_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, b, y, rows, C, eps);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_se_gelu(const bf16_t* x, const float* w1, const float* b1, const float* w2, const float* b2, bf16_t* y,
                   float* scratch, int B, int P, int C, int R, hipStream_t s) {
  if (!x || !w1 || !b1 || !w2 || !b2 || !y || !scratch) return fv_fail(FV_ERR_ARG, "se_gelu: null pointer");
  if (B <= 0 || P <= 0 || C % 8 || R <= 0) return fv_fail(FV_ERR_ARG, "se_gelu: bad shape");
  float* pooled = scratch;             // [B][C]
  float* hid = pooled + (size_t)B * C; // [B][R]
  float* sc = hid + (size_t)B * R;     // [B][C]
  hipLaunchKernelGGL(se_pool_kernel, dim3((C / 8 + 15) / 16, B), dim3(256), 0, s, x, pooled, P, C);
  hipLaunchKernelGGL(se_fc_kernel, dim3((R + 3) / 4, B), dim3(256), 0, s, pooled, w1, b1, hid, R, C, 0);
  hipLaunchKernelGGL(se_fc_kernel, dim3((C + 3) / 4, B), dim3(256), 0, s, hid, w2, b2, sc, C, R, 1);
  const long chunks = (long)B * P * (C / 8);
  hipLaunchKernelGGL(se_apply_gelu_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, x, sc, y, P, C, chunks);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

}  // namespace fv

#ifdef DP_STAMPS
extern "C" int fv_dbg_dp_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fv::g_dp_stamps), 512 * 16 * 8); }
#endif
