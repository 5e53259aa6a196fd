// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands (gfx950): lane l holds row/col l % 16 and the 32 K-elements of
// block l / 16 (bytes in order); D[i][j] = 2^(sa-127) 2^(sb-127) sum_k A[i][k] B[k][j], D layout as the other 16x16 MFMAs.
// Also probes v_cvt_pk_fp8_f32 (OCP e4m3fn on gfx950, saturating?).  Build: hipcc --offload-arch=gfx950 tools/mfma_f8_probe.hip -o tools/bin/mfma_f8_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe(const uint8_t* A, const uint8_t* B, float* D, int sa, int sb) {
  const int lane = threadIdx.x, r = lane & 15, kb = lane >> 4;
  i32x8 a, b;
  const int* ap = reinterpret_cast<const int*>(A + r * 128 + kb * 32);
  const int* bp = reinterpret_cast<const int*>(B + r * 128 + kb * 32);   // B stored [col][k]
  for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  for (int q = 0; q < 4; ++q) D[(4 * kb + q) * 16 + r] = c[q];   // D[i = 4 kb + q][j = r]
}
__global__ void cvt(const float* x, uint8_t* y, int n) {
  const int i = threadIdx.x;
  if (2 * i + 1 < n) {
    const int v = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
    y[2 * i] = v & 0xff; y[2 * i + 1] = (v >> 8) & 0xff;
  }
}
static float e4m3(uint8_t v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float f = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  if (e == 15 && m == 7) f = NAN;
  return s ? -f : f;
}
int main() {
  std::vector<uint8_t> A(16 * 128), B(16 * 128);
  srand(1);
  for (auto& v : A) { do v = rand() & 0xff; while ((v & 0x7f) == 0x7f); }
  for (auto& v : B) { do v = rand() & 0xff; while ((v & 0x7f) == 0x7f); }
  uint8_t *dA, *dB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  for (int t = 0; t < 2; ++t) {
    const int sa = t ? 113 : 127, sb = 127;
    probe<<<1, 64>>>(dA, dB, dD, sa, sb);
    std::vector<float> D(256);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    double worst = 0, mag = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = 0;
        for (int k = 0; k < 128; ++k) s += (double)e4m3(A[i * 128 + k]) * e4m3(B[j * 128 + k]);
        s *= ldexp(1.0, sa - 127);
        worst = fmax(worst, fabs(s - D[i * 16 + j])); mag = fmax(mag, fabs(s));
      }
    printf("scale_a=%d: max |err| %.3e of max |ref| %.3e\n", sa, worst, mag);
  }
  float xs[16] = {0.f, 1.f, -1.5f, 0.0625f, 448.f, 500.f, -1000.f, 0.001f, 0.0019f, 0.3f, 17.f, 1e-5f, 240.f, 3.2f, -0.007f, 1e9f};
  float* dx; uint8_t* dy; hipMalloc(&dx, 64); hipMalloc(&dy, 16);
  hipMemcpy(dx, xs, 64, hipMemcpyHostToDevice);
  cvt<<<1, 64>>>(dx, dy, 16);
  uint8_t ys[16]; hipMemcpy(ys, dy, 16, hipMemcpyDeviceToHost);
  for (int i = 0; i < 16; ++i) printf("cvt %g -> 0x%02x = %g\n", xs[i], ys[i], e4m3(ys[i]));
  return 0;
}
