// tools/slice_micro.hip -- does the 64-byte channel slice cap the depthwise kernels' HBM rate?  A pure streaming kernel with the
// dwpair_march_kernel's traffic shape (one NHWC tensor read once, two written once, a block marching down a strip of columns) in
// two geometries of equal bytes per block: 32 columns x 32 channels (64-byte runs per pixel, the current slices) and 16 columns x 64
// channels (128-byte runs = full cache lines).  hipcc --offload-arch=gfx950 -O3 tools/slice_micro.hip -o tools/bin/slice_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int CH, int TW, int DEPTH>   // CH channels (CH*2 bytes per pixel run) x TW columns per block; DEPTH row-groups in flight
__global__ __launch_bounds__(256, 2) void stream_kernel(const char* __restrict__ x, char* __restrict__ y1, char* __restrict__ y2, int H, int W,
                                                         int C, int tiles_x, int nslices) {
  constexpr int LPP = CH * 2 / 16;              // lanes per pixel run
  constexpr int PPI = 256 / LPP;                // pixels per block-wide instruction
  constexpr int RPI = PPI / TW;                 // rows per instruction
  static_assert(RPI >= 1, "geometry");
  constexpr int NI = 8 / RPI;                   // instructions per 8-row unit
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  // same XCD-local order as the real kernel: consecutive logical blocks on one XCD
  { const int n = gridDim.x, per = n / 8; if (n % 8 == 0) bid = (bid % 8) * per + bid / 8; }
  const int slice = bid % nslices; bid /= nslices;
  const int tx = bid % tiles_x;
  const long b = bid / tiles_x;
  const int lp = tid % LPP, pix = tid / LPP, col = pix % TW, row = pix / TW;
  const size_t rowbytes = (size_t)W * C * 2;
  size_t off = ((size_t)b * H + row) * rowbytes + ((size_t)tx * TW + col) * C * 2 + (size_t)slice * CH * 2 + lp * 16;
  u32x4 v[DEPTH][NI];
  const int ng = H / 8;
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
#pragma unroll
    for (int i = 0; i < NI; ++i) v[d][i] = *reinterpret_cast<const u32x4*>(x + off + (size_t)(d * 8 + i * RPI) * rowbytes);
#pragma unroll 1
  for (int g0 = 0; g0 < ng; g0 += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int g = g0 + d;
      if (g < ng) {
        const int gl = g + DEPTH - 1;
        if (gl < ng) {
#pragma unroll
          for (int i = 0; i < NI; ++i) v[(d + DEPTH - 1) % DEPTH][i] = *reinterpret_cast<const u32x4*>(x + off + (size_t)(gl * 8 + i * RPI) * rowbytes);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          u32x4 a = v[d][i];
          *reinterpret_cast<u32x4*>(y1 + off + (size_t)(g * 8 + i * RPI) * rowbytes) = a;
          a.x ^= 0x10001u;
          *reinterpret_cast<u32x4*>(y2 + off + (size_t)(g * 8 + i * RPI) * rowbytes) = a;
        }
      }
    }
  }
}

template <int CH, int TW, int DEPTH>
static void run(const char* name, const char* x, char* y1, char* y2, int B, int H, int C) {
  const int tiles_x = H / TW, nsl = C / CH;
  const long nblk = (long)B * tiles_x * nsl;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int r = 0; r < 6; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<CH, TW, DEPTH>), dim3((unsigned)nblk), dim3(256), 0, 0, x, y1, y2, H, H, C, tiles_x, nsl);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r && ms < best) best = ms;
  }
  const double bytes = 3.0 * B * H * H * C * 2;
  printf("  %-44s C=%3d H=%3d: %7.1f us = %.2f TB/s\n", name, C, H, best * 1e3, bytes / best / 1e9);
}

int main() {
  const int B = 64;
  const size_t n = (size_t)B * 256 * 256 * 96 * 2;
  char *x, *y1, *y2;
  hipMalloc(&x, n); hipMalloc(&y1, n); hipMalloc(&y2, n);
  hipMemset(x, 1, n);
  const int shapes[3][2] = {{384, 64}, {192, 128}, {96, 256}};
  for (auto& s : shapes) {
    const int C = s[0], H = s[1];
    run<32, 32, 2>("64-B runs (32 ch x 32 col), 2 units in flight", x, y1, y2, B, H, C);
    run<32, 32, 3>("64-B runs (32 ch x 32 col), 3 units in flight", x, y1, y2, B, H, C);
    if (C % 64 == 0) {
      run<64, 16, 2>("128-B runs (64 ch x 16 col), 2 units in flight", x, y1, y2, B, H, C);
      run<64, 16, 3>("128-B runs (64 ch x 16 col), 3 units in flight", x, y1, y2, B, H, C);
      run<64, 32, 2>("128-B runs (64 ch x 32 col), 2 units", x, y1, y2, B, H, C);
    } else {
      (void)0;
    }
  }
  return 0;
}
