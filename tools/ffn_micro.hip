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
    {  // in-kernel clock and wave cycles per chunk: median over workgroups of one more launch after the warm loop
      const int nblk = (M + 64 * 8 - 1) / 64;  // upper bound on the grid
      unsigned long long* ds;
      hipMalloc(&ds, (size_t)nblk * 16 + 64);
      hipMemset(ds, 0, (size_t)nblk * 16 + 64);
      fv::g_ffn_stamps = ds;
      fv::launch_convffn(x, w1, b1, w2, b2, ls, res, out, M, C, H, 0);
      hipDeviceSynchronize();
      std::vector<unsigned long long> hs((size_t)nblk * 2);
      hipMemcpy(hs.data(), ds, hs.size() * 8, hipMemcpyDeviceToHost);
      std::vector<double> cyc, clk;
      for (int b = 0; b < nblk; ++b) if (hs[2 * b + 1]) { cyc.push_back((double)hs[2 * b]); clk.push_back((double)hs[2 * b] / hs[2 * b + 1] * 100.0); }
      std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
      printf("  blocks %zu: median loop cycles %.0f = %.0f per 32-hidden chunk; in-kernel clock %.0f MHz\n", cyc.size(), cyc[cyc.size() / 2],
             cyc[cyc.size() / 2] / (H / 32), clk[clk.size() / 2]);
#ifdef FFN_STAMPS_FINE
      {
        unsigned long long fine[6];
        hipMemcpy(fine, ds + 2 * (cyc.size() - 3), sizeof fine, hipMemcpyDeviceToHost);  // the 6 fine sums sit behind the grid's pairs
        const double nc = H / 32;
        printf("  per chunk (block 0 wave 0): first product %.0f  gelu %.0f  second product to barrier %.0f  barrier %.0f  rest of second product %.0f\n",
               fine[0] / nc, fine[1] / nc, fine[3] / nc, fine[4] / nc, fine[5] / nc);
      }
#endif
      fv::g_ffn_stamps = nullptr;
      hipFree(ds);
    }
#endif
    printf("M=%d C=%d: %.3f ms  %.1f TFLOP/s  (%s)\n", M, C, ms, 4.0 * M * C * H / ms / 1e9, g_err);
    hipFree(x); hipFree(res); hipFree(out); hipFree(w1); hipFree(w2); hipFree(b1); hipFree(b2); hipFree(ls);
  }
  return 0;
}
