// tools/mfma444_rate.hip -- issue rate of v_mfma_f32_4x4x4_16b_bf16 (the depthwise kernels' instruction): clocks per instruction with
// 1, 2 and 3 waves per SIMD, 8 independent accumulators per wave, and with plain VALU dealt between them.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
template <int NV>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* clk, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  s16x4 a = {(short)threadIdx.x, 1, 2, 3}, b = {4, 5, (short)threadIdx.x, 7};
  float v[4] = {1.f, 2.f, 3.f, 4.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int z = 0; z < NV; ++z) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[z & 3]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = v[0] + v[1] + v[2] + v[3];
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int NV>
static void run(int wps, float* out, unsigned long long* clk) {
  const int iters = 2000, blocks = 256 * wps;   // 256-thread blocks = one wave per SIMD each; wps blocks per CU
  hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
  hipDeviceSynchronize();
  unsigned long long h[4096];
  hipMemcpy(h, clk, blocks * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < blocks; ++i) s += (double)h[i];
  printf("  %d wave(s)/SIMD, %d VALU per MFMA: %.1f clk per MFMA per wave -> %.1f clk of SIMD time per MFMA\n", wps, NV, s / blocks / (iters * 8.0),
         s / blocks / (iters * 8.0) / wps);
}
int main() {
  float* out; unsigned long long* clk;
  hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 4096 * 8);
  for (int w = 1; w <= 3; ++w) { run<0>(w, out, clk); run<1>(w, out, clk); run<2>(w, out, clk); run<4>(w, out, clk); }
  return 0;
}
