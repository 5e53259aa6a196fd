// Micro-benchmark for the MFMA depthwise conv: times launch_dwconv_mfma at the tower's three RepMixer stage shapes
// (B = 64) for k = 7 and k = 3 and reports the algorithmic HBM rate (one read + one write of the tensor).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../vla-from-fastvlm_amd/csrc/kernels.h"

static thread_local char g_err[512];
int fv_fail(int code, const char* fmt, ...) { snprintf(g_err, sizeof g_err, "%s", fmt); return code; }
int fv_hip_fail(hipError_t e, const char* what) { snprintf(g_err, sizeof g_err, "hip %d at %s", (int)e, what); return -3; }

#include "../vla-from-fastvlm_amd/csrc/tower_kernels.hip"

int main() {
  struct Shape { int B, S, C; } shapes[] = {{64, 256, 96}, {64, 128, 192}, {64, 64, 384}};
  for (int k : {7, 3})
    for (auto sh : shapes) {
      const size_t n = (size_t)sh.B * sh.S * sh.S * sh.C;
      bf16_t *x, *y, *tt;
      float* bias;
      (void)hipMalloc(&x, n * 2); (void)hipMalloc(&y, n * 2); (void)hipMalloc(&bias, sh.C * 4);
      const size_t te = fv::dwconv_toeplitz_elems(sh.C, k);
      (void)hipMalloc(&tt, te * 2);
      std::vector<uint16_t> hx(n);
      for (size_t i = 0; i < n; ++i) hx[i] = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
      (void)hipMemcpy(x, hx.data(), n * 2, hipMemcpyHostToDevice);
      std::vector<uint16_t> ht(te);
      for (auto& v : ht) v = 0x3c00 - (5 << 7) + (rand() & 0xff);
      (void)hipMemcpy(tt, ht.data(), te * 2, hipMemcpyHostToDevice);
      (void)hipMemset(bias, 0, sh.C * 4);
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      for (int i = 0; i < 3; ++i) fv::launch_dwconv_mfma(x, tt, bias, y, sh.B, sh.S, sh.S, sh.C, k, 0, 0);
      (void)hipDeviceSynchronize();
      const int it = 20;
      (void)hipEventRecord(e0, 0);
      for (int i = 0; i < it; ++i) fv::launch_dwconv_mfma(x, tt, bias, y, sh.B, sh.S, sh.S, sh.C, k, 0, 0);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      ms /= it;
      printf("k=%d B=%d %dx%d C=%d: %.3f ms  %.0f GB/s  (%s)\n", k, sh.B, sh.S, sh.S, sh.C, ms, 2.0 * n * 2 / ms / 1e6, g_err);
      (void)hipFree(x); (void)hipFree(y); (void)hipFree(tt); (void)hipFree(bias);
    }
  return 0;
}
