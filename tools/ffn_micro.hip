// Micro-benchmark for the fused ConvFFN kernel: times launch_convffn at the three tower shapes, optionally with pieces
// ablated (-DFFN_ABLATE_GELU / -DFFN_ABLATE_STAGE) to see which stage bounds it.  Build: tools/build_micro.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../vla-from-fastvlm_amd/csrc/kernels.h"

static thread_local char g_err[512];
int fv_fail(int code, const char* fmt, ...) { snprintf(g_err, sizeof g_err, "%s", fmt); return code; }
int fv_hip_fail(hipError_t e, const char* what) { snprintf(g_err, sizeof g_err, "hip %d at %s", (int)e, what); return -3; }

#include "../vla-from-fastvlm_amd/csrc/convffn_fused.hip"

int main() {
  struct Shape { int M, C; } shapes[] = {{262144, 384}, {1048576, 192}, {4194304, 96}};
  for (auto sh : shapes) {
    const int M = sh.M, C = sh.C, H = 4 * C;
    bf16_t *x, *res, *out, *w1, *w2;
    float *b1, *b2, *ls;
    hipMalloc(&x, (size_t)M * C * 2); hipMalloc(&res, (size_t)M * C * 2); hipMalloc(&out, (size_t)M * C * 2);
    hipMalloc(&w1, (size_t)H * C * 2); hipMalloc(&w2, (size_t)H * C * 2);
    hipMalloc(&b1, H * 4); hipMalloc(&b2, C * 4); hipMalloc(&ls, C * 4);
    std::vector<uint16_t> hx((size_t)M * C);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);  // ~N(0,1)-ish bf16
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(res, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    std::vector<uint16_t> hw((size_t)H * C);
    for (auto& v : hw) v = 0x3c00 - (5 << 7) + (rand() & 0xff) + ((rand() & 1) << 15);  // small weights
    hipMemcpy(w1, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(w2, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemset(b1, 0, H * 4); hipMemset(b2, 0, C * 4); hipMemset(ls, 0, C * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) fv::launch_convffn(x, w1, b1, w2, b2, ls, res, out, M, C, H, 0);
    hipDeviceSynchronize();
    const int it = 10;
    hipEventRecord(e0, 0);
    for (int i = 0; i < it; ++i) fv::launch_convffn(x, w1, b1, w2, b2, ls, res, out, M, C, H, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= it;
#ifdef FFN_STAMPS
    {  // per-block cycle stamps of the first tile (median over blocks) from one more launch after the warm loop
      const int nb = 1024;
      unsigned long long* ds;
      hipMalloc(&ds, nb * 32);
      hipMemset(ds, 0, nb * 32);
      fv::g_ffn_stamps = ds;
      fv::launch_convffn(x, w1, b1, w2, b2, ls, res, out, M, C, H, 0);
      hipDeviceSynchronize();
      std::vector<unsigned long long> hs(nb * 4);
      hipMemcpy(hs.data(), ds, hs.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> v[4];
      for (int b = 0; b < nb; ++b) if (hs[4 * b + 3]) for (int k = 0; k < 4; ++k) v[k].push_back((double)hs[4 * b + k]);
      for (auto& a : v) std::sort(a.begin(), a.end());
      const size_t n = v[0].size();
      if (n) printf("  %zu blocks: prologue %.0f  chunk loop %.0f (= %.0f per chunk)  epilogue %.0f cycles;  in-loop clock %.0f MHz\n", n, v[0][n / 2],
                    v[1][n / 2], v[1][n / 2] / (H / 32), v[2][n / 2], v[1][n / 2] / v[3][n / 2] * 100.0);
      fv::g_ffn_stamps = nullptr;
      hipFree(ds);
    }
#endif
    printf("M=%d C=%d: %.3f ms  %.1f TFLOP/s  (%s)\n", M, C, ms, 4.0 * M * C * H / ms / 1e9, g_err);
    hipFree(x); hipFree(res); hipFree(out); hipFree(w1); hipFree(w2); hipFree(b1); hipFree(b2); hipFree(ls);
  }
  return 0;
}
