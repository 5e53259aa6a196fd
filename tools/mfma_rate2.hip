// Calibration 2: MFMA stream whose A/B operands rotate over NA x NB distinct register sets (as in a real GEMM inner
// loop) instead of one fixed pair: does operand delivery (VGPR read ports / banking) limit the issue rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NA, int NB>
__global__ __launch_bounds__(256, 1) void k(const uint4* in, float* out, int iters) {
  bf16x8 a[NA], b[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) a[i] = __builtin_bit_cast(bf16x8, in[threadIdx.x + 256 * i]);
#pragma unroll
  for (int i = 0; i < NB; ++i) b[i] = __builtin_bit_cast(bf16x8, in[threadIdx.x + 256 * (NA + i)]);
  f32x4 acc[NA][NB];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 64 / (NA * NB); ++r)
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(a[i]), "v"(b[j]));
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j) s += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NA, int NB>
void run(const char* name, const uint4* in, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NA, NB>), dim3(256), dim3(256), 0, 0, in, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NA, NB>), dim3(256), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = 64.0 * iters;
  printf("%-20s %.3f ms  %.1f ns/MFMA/SIMD  %.0f TFLOP/s\n", name, ms, ms * 1e6 / n, 256.0 * 4 * n * 16384 / (ms * 1e-3) / 1e12);
}

int main() {
  uint4* in; float* out;
  hipMalloc(&in, 256 * 16 * 16); hipMalloc(&out, 256 * 256 * 4);
  unsigned short h[256 * 16 * 8];
  for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  run<1, 1>("1x1 operands", in, out);
  run<2, 2>("2x2 operands", in, out);
  run<4, 4>("4x4 operands", in, out);
  run<2, 8>("2x8 operands", in, out);
  run<1, 8>("1x8 operands", in, out);
  return 0;
}
