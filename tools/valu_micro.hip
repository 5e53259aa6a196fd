// tools/valu_micro.hip -- how many VALU instructions ride for free in the shadow of a v_mfma_f32_32x32x16_bf16 (one wave per SIMD)?
//   hipcc -O3 --offload-arch=gfx950 tools/valu_micro.hip -o tools/bin/valu_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) float f32x2;

template <int KIND, int N>   // KIND 0: none, 1: v_fma_f32, 2: v_pk_fma_f32, 3: v_pk_mul_f32, 4: v_cvt_pk_bf16_f32, 5: v_fmed3
__global__ __launch_bounds__(256, 1) void k(const uint4* __restrict__ g, unsigned long long* out, float* sink, int n) {
  const int tid = threadIdx.x;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const uint4 a = g[tid], b = g[tid + 256];
  float v[8]; f32x2 w[8];
  for (int i = 0; i < 8; ++i) { v[i] = __builtin_bit_cast(float, g[tid + 512 + i].x) * 1e-3f; w[i] = f32x2{v[i], v[i] + 1.f}; }
  const float c0 = 1.0001f, c1 = 1e-4f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[s & 3]) : "v"(__builtin_bit_cast(bf16x8, a)), "v"(__builtin_bit_cast(bf16x8, b)));
#pragma unroll
      for (int q = 0; q < N; ++q) {
        if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(c0), "v"(c1));
        if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[q & 7]) : "v"(f32x2{c0, c0}), "v"(f32x2{c1, c1}));
        if (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(w[q & 7]) : "v"(f32x2{c0, c0}));
        if (KIND == 4) { unsigned r; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(v[q & 7]), "v"(v[(q + 1) & 7])); asm volatile("" :: "v"(r)); }
        if (KIND == 5) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(-c0), "v"(c0));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) sum += acc[i][e];
  for (int i = 0; i < 8; ++i) sum += v[i] + w[i].x + w[i].y;
  if (sum == 1234.5f) sink[0] = sum;
  if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int KIND, int N>
void run(const char* name, const uint4* g, unsigned long long* out, float* sink, int n) {
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<KIND, N>), dim3(256), dim3(256), 0, 0, g, out, sink, n);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  (void)hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-18s x%2d per MFMA: %6.1f clk per MFMA\n", name, N, s / 256 / (8.0 * n));
}

int main() {
  uint4* g; unsigned long long* out; float* sink;
  (void)hipMalloc(&g, 1 << 20); (void)hipMalloc(&out, 256 * 8); (void)hipMalloc(&sink, 64);
  std::vector<unsigned> h((1 << 20) / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3f803f80u ^ (unsigned)(i * 2654435761u & 0x00ff00ffu);
  (void)hipMemcpy(g, h.data(), 1 << 20, hipMemcpyHostToDevice);
  const int n = 4000;
  run<0, 0>("none", g, out, sink, n);
  run<1, 2>("v_fma_f32", g, out, sink, n); run<1, 4>("v_fma_f32", g, out, sink, n); run<1, 6>("v_fma_f32", g, out, sink, n); run<1, 8>("v_fma_f32", g, out, sink, n); run<1, 12>("v_fma_f32", g, out, sink, n);
  run<2, 2>("v_pk_fma_f32", g, out, sink, n); run<2, 4>("v_pk_fma_f32", g, out, sink, n); run<2, 6>("v_pk_fma_f32", g, out, sink, n); run<2, 8>("v_pk_fma_f32", g, out, sink, n);
  run<3, 4>("v_pk_mul_f32", g, out, sink, n);
  run<4, 4>("v_cvt_pk_bf16_f32", g, out, sink, n);
  run<5, 4>("v_med3_f32", g, out, sink, n); run<5, 8>("v_med3_f32", g, out, sink, n);
  return 0;
}
