// tools/stage_micro.hip -- what does it cost ONE wave per SIMD to move a 1 KB weight piece into LDS beside its MFMAs?
// 4 waves per block (one per SIMD), one block per CU, every wave: N steps of { v_mfma_f32_32x32x16_bf16 ; ds_read_b128 ;
// [every 2nd step: one staging action] }.  Prints cycles per step (s_memtime) for each staging variant.
//   hipcc -O3 --offload-arch=gfx950 tools/stage_micro.hip -o tools/bin/stage_micro && tools/bin/stage_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int V>
__global__ __launch_bounds__(256, 1) void k(const uint4* __restrict__ g, unsigned long long* out, float* sink, int n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const uint4 b = g[tid + 256];
  for (int i = tid; i < 6144; i += 256) reinterpret_cast<uint4*>(smem)[i] = g[i & 1023];
  __syncthreads();
  const char* rd = smem + (lane & 31) * 784 + (lane >> 5) * 16;
  char* wr = smem + 49152 + tid * 16;            // 12 x 4 KB staging target
  const uint4* src = g + tid;
  const unsigned m0base = __builtin_amdgcn_readfirstlane(49152u + (threadIdx.x >> 6) * 1024u);
  const unsigned toff = tid * 16u;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(g), 0, 1 << 22, 0x00020000);
  constexpr int PD = 6, NS = 24;
  uint4 ring[PD], st[12];
#pragma unroll
  for (int i = 0; i < PD; ++i) ring[i] = *reinterpret_cast<const uint4*>(rd + i * 32);
#pragma unroll
  for (int j = 0; j < 12; ++j) st[j] = src[j * 256];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < n; ++it) {
    const uint4* sn = src + ((it & 7) + 1) * 3072;   // next "chunk" of weights (L2-resident 384 KB window)
    const uint4* gu = g + ((it & 7) + 1) * 3072;     // the same, wave-uniform (SGPR base + 32-bit lane offset)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const bf16x8 a = __builtin_bit_cast(bf16x8, ring[s % PD]);
      if (V >= 1 && V != 14 && V != 15) ring[s % PD] = *reinterpret_cast<const uint4*>(rd + ((s + PD) % NS) * 32);
      if (V == 14 || V == 15) asm volatile("ds_read_b128 %0, %1" : "=v"(ring[s % PD]) : "v"((unsigned)(size_t)(rd + ((s + PD) % NS) * 32)) : "memory");
      if ((V == 14 || V == 15) && s % PD == PD - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[s & 3]) : "v"(a), "v"(__builtin_bit_cast(bf16x8, b)));
      const int j = s >> 1;
      if ((s & 1) == 0) {
        const uint4 v = st[j];
        if (V == 2) *reinterpret_cast<uint4*>(wr + j * 4096) = v;
        if (V == 3) { asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:8" :: "v"((unsigned)(size_t)(wr + j * 4096)), "v"(make_uint2(v.x, v.y)), "v"(make_uint2(v.z, v.w)) : "memory"); }
        if (V == 4) __builtin_amdgcn_global_load_lds(sn + j * 256, (__attribute__((address_space(3))) void*)(smem + m0base + j * 4096), 16, 0, 0);
        if (V == 5) { asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %2 offset:4\n\tds_write_b32 %0, %3 offset:8\n\tds_write_b32 %0, %4 offset:12" :: "v"((unsigned)(size_t)(wr + j * 4096)), "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w) : "memory"); }
        if (V == 6) { asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tds_write_addtid_b32 %0\n\tds_write_addtid_b32 %1 offset:256\n\tds_write_addtid_b32 %2 offset:512\n\tds_write_addtid_b32 %3 offset:768" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "s"(m0base + j * 4096) : "memory"); }
        if (V == 9 && wid == 0) *reinterpret_cast<uint4*>(wr + j * 4096) = v;       // as 2, wave 0 only
        if (V == 8 || V == 10 || V == 11 || V == 12) asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
        if (V == 13) { *reinterpret_cast<uint4*>(wr + j * 4096) = v; st[j] = sn[j * 256]; }
        if (V == 14) __builtin_amdgcn_global_load_lds(sn + j * 256, (__attribute__((address_space(3))) void*)(smem + m0base + j * 4096), 16, 0, 0);
        if (V == 15) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + m0base + j * 4096), 16, toff, ((it & 7) + 1) * 49152 + j * 4096, 0, 0);
        if (V == 16) *reinterpret_cast<uint4*>(wr + j * 4096) = v;
        if (V == 17) { asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tds_write_addtid_b32 %0\n\tds_write_addtid_b32 %1 offset:256\n\tds_write_addtid_b32 %2 offset:512\n\tds_write_addtid_b32 %3 offset:768" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "s"(m0base + j * 4096) : "memory"); }
        if (V == 12 && (j & 1) == 0) { st[j] = sn[j * 256]; st[j + 1] = sn[(j + 1) * 256]; }
      } else {
        if (V == 2 || V == 3 || V == 5 || V == 6 || V == 8 || V == 9) st[j] = sn[j * 256];   // the staging load for the next iteration
        if (V == 10) st[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(gu) + toff + j * 4096);
        if (V == 11 || V == 16 || V == 17) st[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, toff + j * 4096, 0, 0));
      }
    }
    if (V == 4 || V == 14 || V == 15) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) sum += acc[i][e];
  for (int j = 0; j < 12; ++j) sum += __builtin_bit_cast(float, st[j].x);
  for (int j = 0; j < PD; ++j) sum += __builtin_bit_cast(float, ring[j].x);
  sum += smem[49152 + tid];
  if (sum == 1234.5f) sink[0] = sum;
  if (tid == 0) out[blockIdx.x] = t1 - t0;
}

template <int V>
void run(const char* name, const uint4* g, unsigned long long* out, float* sink, int n) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<V>, dim3(256), dim3(256), 98304, 0, g, out, sink, n);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-44s %7.1f clk per MFMA step\n", name, s / 256 / (24.0 * n));
}

int main() {
  uint4* g; unsigned long long* out; float* sink;
  hipMalloc(&g, 1 << 22); hipMalloc(&out, 256 * 8); hipMalloc(&sink, 64);
  std::vector<unsigned> h((1 << 22) / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3f803f80u ^ (unsigned)(i * 2654435761u & 0x00ff00ffu);
  hipMemcpy(g, h.data(), 1 << 22, hipMemcpyHostToDevice);
  const int n = 2000;
  run<0>("MFMA only", g, out, sink, n);
  run<1>("+ ds_read_b128 per step", g, out, sink, n);
  run<8>("+ staging load every 2nd step (no store)", g, out, sink, n);
  run<10>("+ staging load, SGPR base + 32-bit offset", g, out, sink, n);
  run<11>("+ staging load, raw_buffer_load_b128", g, out, sink, n);
  run<12>("+ staging loads in pairs (2 per 4 steps)", g, out, sink, n);
  run<13>("+ load AND ds_write_b128 in the same step", g, out, sink, n);
  run<14>("+ global_load_lds with asm ds_reads", g, out, sink, n);
  run<15>("+ raw_buffer_load_lds with asm ds_reads", g, out, sink, n);
  run<16>("+ buffer load / ds_write_b128 alternating", g, out, sink, n);
  run<17>("+ buffer load / 4 x ds_write_addtid_b32", g, out, sink, n);
  run<2>("+ load / ds_write_b128 alternating", g, out, sink, n);
  run<3>("+ load / 2 x ds_write_b64", g, out, sink, n);
  run<5>("+ load / 4 x ds_write_b32", g, out, sink, n);
  run<6>("+ load / 4 x ds_write_addtid_b32", g, out, sink, n);
  run<4>("+ global_load_lds_dwordx4 every 2nd step", g, out, sink, n);
  return 0;
}
