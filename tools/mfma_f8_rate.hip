// Issue rate of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands) against v_mfma_f32_16x16x32_bf16: cycles per instruction for a stream of
// independent accumulators, one wave per SIMD and two.  hipcc --offload-arch=gfx950 tools/mfma_f8_rate.hip -o tools/bin/mfma_f8_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>
__global__ void rate(float* out, long long* cyc, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x * 0x01010101 * (i + 1); b[i] = 0x3c3c3c3c ^ (threadIdx.x * 7 + i); }
  typedef __attribute__((ext_vector_type(4))) int i32x4;
  const i32x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
  bf16x8 ab = __builtin_bit_cast(bf16x8, a4);
  bf16x8 bb = __builtin_bit_cast(bf16x8, b4);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[i], 0, 0, 0);
      else {
        const int sa = 113, sb = 127;
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]" : "+v"(acc[i]) : "v"(a), "v"(b), "v"(sa), "v"(sb));
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int waves = 1; waves <= 2; ++waves)
    for (int mode = 0; mode < 2; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      dim3 g(256), b(256 * waves);
      if (mode == 0) rate<0><<<g, b>>>(out, cyc, iters); else rate<1><<<g, b>>>(out, cyc, iters);
      hipEventRecord(e0);
      if (mode == 0) rate<0><<<g, b>>>(out, cyc, iters); else rate<1><<<g, b>>>(out, cyc, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      const double n = (double)iters * 8;
      const double flop = (mode == 0 ? 2.0 * 16 * 16 * 32 : 2.0 * 16 * 16 * 128) * n * 256 * 4 * waves;
      printf("%s  %d wave(s)/SIMD: %.1f cycles (s_memtime units) per MFMA per wave, %.3f ms, %.0f TFLOP/s\n", mode ? "fp8 16x16x128 scaled" : "bf16 16x16x32       ", waves,
             (double)c / n, ms, flop / ms / 1e9);
    }
  return 0;
}
