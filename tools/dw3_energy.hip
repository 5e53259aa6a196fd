// tools/dw3_energy.hip -- VERDICT r3 #5 priced before building: energy per multiply-accumulate of the three instruction streams a depthwise
// 3x3 could run on, register-resident, random operands, two waves per SIMD, each looped for ~2.5 s while a host thread samples the board power
// (amdgpu hwmon power1_input of the visible GPU):
//   mfma444   v_mfma_f32_4x4x4_16b_bf16   1024 executed MACs per instruction (the Toeplitz form dwpair_march_kernel uses: 37 % of them useful on a 3x3,
//                                          58 % on a 7x7)
//   pk_fma    v_pk_fma_f32                 128 MACs per instruction, fp32 operands (bf16 inputs unpacked first: one shift / and per pair, amortised)
//   dot2      v_dot2_f32_bf16              128 MACs per instruction on packed bf16 operands (two TAPS of one channel per register: needs a permute per pair)
// Prints time, mean power above the idle floor measured first, joules and pJ per executed MAC.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/dw3_energy tools/dw3_energy.hip -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <cctype>
#include <glob.h>
#include <string>
#include <thread>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// a random bf16 pair with exponents near 1 (|x| in [0.5, 2)): full-range mantissas and both signs, no overflow in a long accumulation of 1e-? products
__device__ __forceinline__ unsigned rnd_bf2(unsigned seed) {
  const unsigned r = mix(seed);
  const unsigned lo = (r & 0x80ffu) | 0x3f00u, hi = ((r >> 16) & 0x80ffu) | 0x3f00u;
  return lo | (hi << 16);
}

template <int KIND>
__global__ __launch_bounds__(256) void loop_kernel(float* out, int iters) {
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  float res = 0.f;
  if constexpr (KIND == 0) {
    f32x4 acc[8];
    s16x4 a[2], b[2];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) {
      const unsigned a0 = rnd_bf2(tid * 8 + i * 4), a1 = rnd_bf2(tid * 8 + i * 4 + 1), b0 = rnd_bf2(tid * 8 + i * 4 + 2), b1 = rnd_bf2(tid * 8 + i * 4 + 3);
      a[i] = s16x4{(short)a0, (short)(a0 >> 16), (short)a1, (short)(a1 >> 16)};
      b[i] = s16x4{(short)b0, (short)(b0 >> 16), (short)b1, (short)(b1 >> 16)};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_4x4x4_16b_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 1]), "v"(b[(i >> 1) & 1]));
      if ((it & 63) == 63) {   // keep the accumulators bounded (random signs: they random-walk) without touching the stream's rate
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] *= 0.5f;
      }
    }
    for (int i = 0; i < 8; ++i) res += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else if constexpr (KIND == 1) {
    f32x2 acc[8], x[2], w[2];
    for (int i = 0; i < 8; ++i) acc[i] = f32x2{0.f, 0.f};
    for (int i = 0; i < 2; ++i) {
      const unsigned a0 = rnd_bf2(tid * 4 + i * 2), b0 = rnd_bf2(tid * 4 + i * 2 + 1);
      x[i] = f32x2{__uint_as_float(a0 << 16), __uint_as_float(a0 & 0xffff0000u)};
      w[i] = f32x2{__uint_as_float(b0 << 16), __uint_as_float(b0 & 0xffff0000u)};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[i & 1]), "v"(w[(i >> 1) & 1]));
      if ((it & 63) == 63) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] *= 0.5f;
      }
    }
    for (int i = 0; i < 8; ++i) res += acc[i][0] + acc[i][1];
  } else {
    float acc[8];
    unsigned x[2], w[2];
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int i = 0; i < 2; ++i) { x[i] = rnd_bf2(tid * 4 + i * 2); w[i] = rnd_bf2(tid * 4 + i * 2 + 1); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x[i & 1]), "v"(w[(i >> 1) & 1]));
      if ((it & 63) == 63) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] *= 0.5f;
      }
    }
    for (int i = 0; i < 8; ++i) res += acc[i];
  }
  out[tid] = res;
}

static std::string power_path() {   // the hwmon sensor of the GPU this process runs on (matched by PCI slot)
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, sizeof(bus), 0) != hipSuccess) return "";
  std::string slot(bus);
  for (auto& c : slot) c = (char)tolower(c);
  glob_t g;
  std::string p;
  if (glob("/sys/class/drm/card*/device/uevent", 0, nullptr, &g) == 0) {
    for (size_t i = 0; i < g.gl_pathc && p.empty(); ++i) {
      std::ifstream f(g.gl_pathv[i]);
      std::string line, all;
      while (std::getline(f, line)) all += line + "\n";
      for (auto& c : all) c = (char)tolower(c);
      if (all.find("pci_slot_name=" + slot) == std::string::npos) continue;
      std::string dir(g.gl_pathv[i]);
      dir = dir.substr(0, dir.rfind('/'));
      glob_t h;
      if (glob((dir + "/hwmon/hwmon*/power1_input").c_str(), 0, nullptr, &h) == 0 && h.gl_pathc > 0) p = h.gl_pathv[0];
      globfree(&h);
    }
  }
  globfree(&g);
  return p;
}
static double read_w(const std::string& p) {
  std::ifstream f(p);
  double v = 0;
  f >> v;
  return v / 1e6;
}

struct Sampler {
  std::string path;
  std::atomic<bool> stop{false};
  std::vector<double> w;
  std::thread th;
  void start() {
    stop = false; w.clear();
    th = std::thread([this] { while (!stop) { w.push_back(read_w(path)); std::this_thread::sleep_for(std::chrono::milliseconds(20)); } });
  }
  double finish(double skip_s) {   // mean of the samples after the first skip_s seconds (the sensor averages over ~1 s)
    stop = true; th.join();
    const size_t s0 = (size_t)(skip_s / 0.02);
    double s = 0; size_t n = 0;
    for (size_t i = s0; i < w.size(); ++i) { s += w[i]; ++n; }
    return n ? s / n : 0.0;
  }
};

template <int KIND>
static void run(const char* name, double macs_per_instr, float* out, Sampler& smp, double idle_w) {
  const int blocks = 256 * 2;   // two 4-wave blocks per CU: two waves per SIMD
  int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(loop_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);   // calibrate
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(loop_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const int reps = (int)(2500.0 / ms) + 1;
  smp.start();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(loop_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  const double watts = smp.finish(1.0);
  const double instr = (double)reps * iters * 8.0 * blocks * 4.0;   // wave-instructions
  const double macs = instr * macs_per_instr;
  const double joule = watts * ms / 1e3, joule_dyn = (watts - idle_w) * ms / 1e3;
  printf("%-8s %8.1f ms  %7.1f W (idle %5.1f)  %6.2f T MAC/s  %6.2f pJ / executed MAC (board)  %6.2f pJ above idle  [%.3g wave-instructions]\n",
         name, ms, watts, idle_w, macs / ms / 1e9, joule / macs * 1e12, joule_dyn / macs * 1e12, instr);
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 256 * 4);
  Sampler smp;
  smp.path = power_path();
  if (smp.path.empty()) { printf("no hwmon power sensor found\n"); return 1; }
  std::this_thread::sleep_for(std::chrono::seconds(2));
  smp.start();
  std::this_thread::sleep_for(std::chrono::seconds(2));
  const double idle = smp.finish(0.5);
  { char bus[64] = {0}; (void)hipDeviceGetPCIBusId(bus, sizeof(bus), 0); printf("device %s, sensor %s, idle %.1f W\n", bus, smp.path.c_str(), idle); }
  run<0>("mfma444", 1024.0, out, smp, idle);
  run<1>("pk_fma", 128.0, out, smp, idle);
  run<2>("dot2", 128.0, out, smp, idle);
  return 0;
}
