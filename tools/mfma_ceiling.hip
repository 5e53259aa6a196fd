// tools/mfma_ceiling.hip -- what the matrix pipes of THIS chip sustain on random bf16 operands, with nothing else to do: 8 x 4
// register-resident operand sets (a GEMM wave's fragments) rotating over 32 accumulators, one or two waves per SIMD, ~30 ms per
// configuration so that the power management has settled.  Prints TFLOP/s, the clock (s_memtime / wall) and clk per MFMA.
// The kernels' `roofline.frac` is quoted against the 2.5 PF datasheet peak; this is the ceiling a perfect schedule would see.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>   // 0: 16x16x32 (32 accumulators of 4), 1: 32x32x16 (8 accumulators of 16)
__global__ __launch_bounds__(512, 1) void k(const uint4* in, float* out, unsigned long long* clk, int iters, int zero) {
  bf16x8 a[8], b[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(bf16x8, zero ? make_uint4(0, 0, 0, 0) : in[threadIdx.x + 512 * i]);
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(bf16x8, zero ? make_uint4(0, 0, 0, 0) : in[threadIdx.x + 512 * (8 + i)]);
  float s = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  if constexpr (SHAPE == 0) {
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i + 4 * r]), "v"(b[j + 2 * r]));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int SHAPE>
static void run(const char* name, int threads, const uint4* in, float* out, unsigned long long* clk, int zero) {
  const double fl_per_iter_wave = SHAPE == 0 ? 32.0 * 16 * 16 * 32 * 2 : 16.0 * 32 * 32 * 16 * 2;
  const int mfma_per_iter = SHAPE == 0 ? 32 : 16;
  const int iters = 60000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int r = 0; r < 2; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<SHAPE>, dim3(256), dim3(threads), 0, 0, in, out, clk, iters, zero);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), clk, 256 * 8, hipMemcpyDeviceToHost);
  double c = 0;
  for (auto v : h) c += (double)v;
  c /= 256;
  const double waves = 256.0 * threads / 64;
  printf("  %-10s %d wave(s)/SIMD, %s operands: %6.1f ms  %7.0f TFLOP/s  clock %.2f GHz  %.1f clk per MFMA per wave\n", name, threads / 256,
         zero ? "zero  " : "random", ms, fl_per_iter_wave * iters * waves / ms / 1e9, c / ms / 1e6, c / ((double)iters * mfma_per_iter));
}

int main() {
  uint4* in; float* out; unsigned long long* clk;
  std::vector<unsigned short> h(512 * 12 * 8);
  srand(1);
  for (auto& v : h) { float f = (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  hipMalloc(&in, h.size() * 2); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&clk, 256 * 8);
  hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  for (int zero = 0; zero < 2; ++zero) {
    run<0>("16x16x32", 256, in, out, clk, zero);
    run<0>("16x16x32", 512, in, out, clk, zero);
    run<1>("32x32x16", 256, in, out, clk, zero);
    run<1>("32x32x16", 512, in, out, clk, zero);
  }
  return 0;
}
