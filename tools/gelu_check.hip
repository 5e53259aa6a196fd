// tools/gelu_check.hip -- evaluates common.h's gelu_f / gelu2_f on the device over a dense grid and prints the max |error|
// against the exact erf form (checks, among other things, that the VOP3P clamp bit saturates both packed halves).
#include <cmath>
#include <cstdio>
#include <vector>
#include "../vla-from-fastvlm_amd/csrc/common.h"
int fv_hip_fail(hipError_t, const char*) { return -1; }
__global__ void k(const float* x, float* y1, float* y2, int n) {
  const int i = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (i + 1 >= n) return;
  y1[i] = gelu_f(x[i]); y1[i + 1] = gelu_f(x[i + 1]);
  const f32x2 g = gelu2_f((f32x2){x[i], x[i + 1]});
  y2[i] = g.x; y2[i + 1] = g.y;
}
int main() {
  const int n = 1 << 22;
  std::vector<float> x(n), y1(n), y2(n);
  for (int i = 0; i < n; ++i) x[i] = -60.0f + 120.0f * i / (n - 1);
  for (int i = 0; i < 64; ++i) x[i] = (i & 1 ? -1.f : 1.f) * std::pow(10.f, (float)(i / 2));  // far tails, up to 1e31
  float *dx, *d1, *d2;
  hipMalloc(&dx, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  k<<<n / 512, 256>>>(dx, d1, d2, n);
  hipMemcpy(y1.data(), d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(y2.data(), d2, n * 4, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0, r2 = 0;
  for (int i = 0; i < n; ++i) {
    const double ref = 0.5 * x[i] * (1.0 + std::erf(x[i] / std::sqrt(2.0)));
    const double a1 = std::fabs(y1[i] - ref), a2 = std::fabs(y2[i] - ref);
    if (std::fabs(x[i]) < 100) { e1 = std::fmax(e1, a1); e2 = std::fmax(e2, a2); }
    else r2 = std::fmax(r2, a2 / std::fmax(std::fabs(ref), 1e-30));  // tails: relative (x or -0)
    if (!(a2 == a2)) { printf("NaN at x=%g\n", x[i]); return 1; }
  }
  printf("max|err| gelu_f %.3g  gelu2_f %.3g  (|x|<100);  gelu2_f far-tail max rel err %.3g\n", e1, e2, r2);
  return e2 < 1e-4 && r2 < 1e-6 ? 0 : 1;
}
