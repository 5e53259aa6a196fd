// Calibration: how many cycles one wave per SIMD needs per v_mfma_f32_16x16x32_bf16 in different dependency / register-file
// arrangements (wall time over a long loop; 256 blocks x 256 threads = one wave per SIMD on every CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC, bool AGPR, bool ASM>
__global__ __launch_bounds__(256, 1) void k(const uint4* in, float* out, int iters) {
  bf16x8 a = __builtin_bit_cast(bf16x8, in[threadIdx.x]), b = __builtin_bit_cast(bf16x8, in[threadIdx.x + 256]);
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 64 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (ASM) {
          if (AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
          else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        } else {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
      }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool AGPR, bool ASM>
void run(const char* name, const uint4* in, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, AGPR, ASM>), dim3(256), dim3(256), 0, 0, in, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NACC, AGPR, ASM>), dim3(256), dim3(256), 0, 0, in, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_wave = 64.0 * iters;
  const double tflops = 256.0 * 4 * mfma_per_wave * 16384 / (ms * 1e-3) / 1e12;
  printf("%-28s %.3f ms  %.1f ns/MFMA/SIMD (= %.1f cycles @2.1GHz)  %.0f TFLOP/s\n", name, ms, ms * 1e6 / mfma_per_wave,
         ms * 1e6 / mfma_per_wave * 2.1, tflops);
}

int main() {
  uint4* in; float* out;
  hipMalloc(&in, 512 * 16); hipMalloc(&out, 256 * 256 * 4);
  hipMemset(in, 0x3c, 512 * 16);
  const bool random_data = getenv("MFMA_RANDOM") != nullptr;
  if (random_data) {  // ~N(0,1)-ish bf16 with random sign/mantissa: the data-dependent power draw lowers the clock
    unsigned short h[512 * 8];
    for (auto& v : h) v = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  }
  printf("operands: %s\n", random_data ? "random" : "constant");
  run<16, false, false>("builtin 16 acc", in, out);
  run<4, false, false>("builtin 4 acc", in, out);
  run<2, false, false>("builtin 2 acc", in, out);
  run<1, false, false>("builtin 1 acc", in, out);
  run<16, false, true>("asm vgpr 16 acc", in, out);
  run<2, false, true>("asm vgpr 2 acc", in, out);
  run<16, true, true>("asm agpr 16 acc", in, out);
  run<2, true, true>("asm agpr 2 acc", in, out);
  return 0;
}
