// tools/l2_handover.hip -- prices VERDICT r4 #3: can the RepMixer block's intermediate tensors (t, x') be handed from the depthwise pair to the
// fused ConvFFN through an XCD's 4 MB L2 instead of through HBM?  Two questions, each answered by rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE per launch
// (tools/l2_handover.sh runs the passes) and by wall time:
//   W. do stores to a buffer that is OVERWRITTEN while still L2-resident stay in L2 (a write-back cache would send only the last version to memory)?
//      `overwrite<MODE>`: every block rewrites its own `slab` bytes `passes` times (256 blocks x slab <= 16 MB: half of the chip's 32 MB of L2).
//   R. does a block on the SAME XCD read those bytes from L2 (no HBM fetch), and at what rate, while the other streams of the two kernels pass through
//      the same L2?  `handover`: per XCD, P producer blocks write ring slabs (t and x' of one strip: 2 x slab bytes) after reading x from a stream of
//      unique addresses; the other 24 blocks of that XCD wait for the strip (agent-scope counter, sc1 polls), each reads ITS TILE of it with sc1 loads (L1
//      bypass), re-reads the whole `wbytes` weight image (the ConvFFN's 2.4 MB per 128-row tile, L2-resident today) and writes its tile of an output stream.
//      A strip is therefore 24 tiles: at the kernel's real tile (128 rows x 384 channels x 2 B = 96 KB) t and x' of ONE strip are 4.6 MB -- more than the
//      4 MB L2 they are to be handed through -- so the tile size is swept downwards to find where the ring starts to fit beside the weights.  Payload words carry (step, index)
//      so the consumer counts stale / torn reads.  Every spin is bounded (a timeout word ends the kernel), every wave reaches the exit.
// Build: hipcc -O3 --offload-arch=gfx950 tools/l2_handover.hip -o tools/bin/l2_handover
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } \
  } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_sc1(uint4* p, uint4 v) {
  const u32x4 r = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ void store_nt(uint4* p, uint4 v) {
  const u32x4 r = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ uint4 load_sc1(const uint4* p) {   // (the caller waits: s_waitcnt vmcnt(0) before the value is used)
  u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
  return make_uint4(r.x, r.y, r.z, r.w);
}
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u; }   // HW_REG_XCC_ID[3:0]

// ---- W: MODE 0 plain, 1 nt, 2 sc1 stores; READ: also read (sc1 loads) the slab of block b + 8 (same XCD under round-robin placement)
template <int MODE, bool READ>
__global__ __launch_bounds__(256) void overwrite(uint4* buf, size_t slab_u4, int passes, uint4* sink) {
  uint4* mine = buf + (size_t)blockIdx.x * slab_u4;
  const uint4* other = buf + (size_t)((blockIdx.x + 8) % gridDim.x) * slab_u4;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int p = 0; p < passes; ++p) {
    for (size_t i = threadIdx.x; i < slab_u4; i += 256) {
      const uint4 v = make_uint4((unsigned)p, (unsigned)i, blockIdx.x, 7u);
      if (MODE == 0) mine[i] = v;
      else if (MODE == 1) store_nt(mine + i, v);
      else store_sc1(mine + i, v);
    }
    if (READ) {
      for (size_t i = threadIdx.x; i < slab_u4; i += 256) {
        const uint4 v = load_sc1(other + i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc.x ^= v.x; acc.y += v.y;
      }
    }
    __syncthreads();
  }
  if (acc.x == 0x12345678u) sink[blockIdx.x] = acc;
}

// ---- R
struct HoParams {
  const uint4* x;      // [steps][8][slab_u4]   unique per step: HBM reads
  uint4* ring;         // [8][R][2][slab_u4]    t and x' of a strip
  const uint4* w;      // [wbytes / 16]         the weight image every consumer tile re-reads
  uint4* out;          // [steps][8][slab_u4]   output stream
  unsigned* ctl;       // [8][64]: 0 arrive, 16 produced (chunks), 32 consumed (chunks); [8*64]: total arrive; [8*64+1]: timeout; [8*64+2]: stale words; [8*64+3]: checked words
  size_t slab_u4, w_u4;
  int steps, R, P, direct;   // direct: no ring -- t, x' go to a [steps] stream like today's two launches (the baseline within the same launch shape)
  uint4* tstream;      // [steps][8][2][slab_u4] (direct)
};
constexpr long SPIN_CAP = 1L << 24;
__device__ __forceinline__ bool wait_ge(const unsigned* p, unsigned want, unsigned* timeout) {
  for (long it = 0; it < SPIN_CAP; ++it) {
    if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
    if ((it & 1023) == 1023 && __hip_atomic_load(timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    __builtin_amdgcn_s_sleep(2);
  }
  __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}
__global__ __launch_bounds__(256) void handover(HoParams p) {
  __shared__ unsigned s_idx, s_n, s_ok;
  const unsigned xcd = xcc_id() & 7u;
  unsigned* ctl = p.ctl + xcd * 64;
  unsigned* total = p.ctl + 8 * 64;
  unsigned* timeout = total + 1;
  if (threadIdx.x == 0) {
    s_idx = atomicAdd(ctl + 0, 1u);
    atomicAdd(total, 1u);
    s_ok = wait_ge(total, gridDim.x, timeout) ? 1u : 0u;   // every block is resident (grid = one block per CU): the XCD's head count is final
    s_n = __hip_atomic_load(ctl + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_ok) return;
  const int idx = (int)s_idx, nblk = (int)s_n, P = p.P, Cn = nblk - P;
  if (nblk != 32) {   // the host sized a strip for 32 - P consumer tiles (not seen: round-robin placement gives every XCD 32 of 256 blocks)
    if (threadIdx.x == 0) __hip_atomic_store(timeout, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const size_t S = p.slab_u4;   // one tensor of a strip = Cn consumer tiles
  unsigned stale = 0, checked = 0;
  uint4 acc = make_uint4(0, 0, 0, 0);
  if (idx < P) {
    // producer: chunk idx of every strip
    const size_t c0 = S * idx / P, c1 = S * (idx + 1) / P;
    for (int k = 0; k < p.steps; ++k) {
      uint4* dst;
      if (p.direct) dst = p.tstream + ((size_t)k * 8 + xcd) * 2 * S;
      else {
        if (k >= p.R) {   // the ring entry is free once every consumer chunk of strip k - R has been read
          if (threadIdx.x == 0) s_ok = wait_ge(ctl + 32, (unsigned)(k - p.R + 1) * (unsigned)Cn, timeout) ? 1u : 0u;
          __syncthreads();
          if (!s_ok) return;
        }
        dst = p.ring + ((size_t)xcd * p.R + (k % p.R)) * 2 * S;
      }
      const uint4* src = p.x + ((size_t)k * 8 + xcd) * S;
      for (size_t i = c0 + threadIdx.x; i < c1; i += 256) {
        const uint4 v = src[i];
        dst[i] = make_uint4((unsigned)k, (unsigned)i, v.x, v.y);          // t
        dst[S + i] = make_uint4((unsigned)k, (unsigned)i + 1u, v.z, v.w);  // x'
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) atomicAdd(ctl + 16, 1u);
    }
  } else {
    const int ci = idx - P;
    const size_t c0 = S * ci / Cn, c1 = S * (ci + 1) / Cn;
    for (int k = 0; k < p.steps; ++k) {
      if (threadIdx.x == 0) s_ok = wait_ge(ctl + 16, (unsigned)(k + 1) * (unsigned)P, timeout) ? 1u : 0u;
      __syncthreads();
      if (!s_ok) return;
      const uint4* src = p.direct ? p.tstream + ((size_t)k * 8 + xcd) * 2 * S : p.ring + ((size_t)xcd * p.R + (k % p.R)) * 2 * S;
      uint4* o = p.out + ((size_t)k * 8 + xcd) * S;
      // the weight image of this tile (L2-resident in today's kernel: every block re-reads it per 128-row tile)
      for (size_t i = threadIdx.x; i < p.w_u4; i += 256) { const uint4 v = p.w[i]; acc.x ^= v.x; acc.y += v.w; }
      for (size_t i0 = c0; i0 < c1; i0 += 1024) {   // 4 + 4 sc1 loads in flight per lane (tile sizes are multiples of 16 KB)
        u32x4 t[4], r[4];
        const size_t lim = c1 - 1 - threadIdx.x;   // tiles smaller than 16 KB: lanes past the tile re-load its last line and are not checked
        const size_t o0 = i0 < lim ? i0 : lim, o1 = i0 + 256 < lim ? i0 + 256 : lim, o2 = i0 + 512 < lim ? i0 + 512 : lim, o3 = i0 + 768 < lim ? i0 + 768 : lim;
        const uint4 *a0 = src + threadIdx.x, *b0 = src + S + threadIdx.x;
        asm volatile(
            "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\t"
            "global_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
            "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\t"
            "global_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
            : "v"(a0 + o0), "v"(a0 + o1), "v"(a0 + o2), "v"(a0 + o3), "v"(b0 + o0), "v"(b0 + o1), "v"(b0 + o2), "v"(b0 + o3)
            : "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const size_t i = i0 + threadIdx.x + 256 * j;
          if (i >= c1) continue;
          stale += (t[j].x != (unsigned)k || t[j].y != (unsigned)i) + (r[j].x != (unsigned)k || r[j].y != (unsigned)i + 1u);
          checked += 2;
          o[i] = make_uint4(t[j].z ^ r[j].z, t[j].w + r[j].w, acc.x, acc.y);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) atomicAdd(ctl + 32, 1u);
    }
  }
  if (stale) atomicAdd(total + 2, stale);
  if (checked) atomicAdd(total + 3, checked);
}

template <int MODE, bool READ>
static void run_overwrite(const char* name, uint4* buf, size_t slab_bytes, int passes, uint4* sink) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((overwrite<MODE, READ>), dim3(256), dim3(256), 0, 0, buf, slab_bytes / 16, 2, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((overwrite<MODE, READ>), dim3(256), dim3(256), 0, 0, buf, slab_bytes / 16, passes, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double bytes = 256.0 * slab_bytes * passes;
  printf("overwrite %-22s slab %4zu KB x 256 blocks x %d passes: %7.1f us, %6.2f TB/s stored%s  (algorithmic %0.1f MB stored, footprint %0.1f MB)\n", name, slab_bytes >> 10, passes,
         ms * 1e3, bytes / ms / 1e9, READ ? " (+ as much read)" : "", bytes / 1e6, 256.0 * slab_bytes / 1e6);
}

int main(int argc, char** argv) {
  const int passes = argc > 1 ? atoi(argv[1]) : 32;
  const int steps = argc > 2 ? atoi(argv[2]) : 128;
  uint4 *buf, *sink;
  CK(hipMalloc(&buf, 256ull * (1 << 20)));
  CK(hipMalloc(&sink, 4096 * 16));
  for (size_t slab : {(size_t)32 << 10, (size_t)64 << 10, (size_t)512 << 10}) {
    run_overwrite<0, false>("plain", buf, slab, passes, sink);
    run_overwrite<1, false>("nt", buf, slab, passes, sink);
    run_overwrite<2, false>("sc1", buf, slab, passes, sink);
    run_overwrite<0, true>("plain + neighbour read", buf, slab, passes, sink);
  }
  // ---- hand-over
  const size_t wbytes = 2359296;   // the C = 384 ConvFFN's packed weight image
  for (size_t tile : {(size_t)96 << 10, (size_t)32 << 10, (size_t)16 << 10, (size_t)8 << 10}) {   // bytes of t per consumer tile : 128 / 43 / 21 / 11 rows x 384 channels (bf16); the ring holds t and x'
    const size_t slab = tile * 24;
    for (int cfg = 0; cfg < 3; ++cfg) {
      const int direct = cfg == 0, R = cfg == 2 ? 4 : 2, P = 8;
      HoParams p{};
      p.slab_u4 = slab / 16; p.w_u4 = wbytes / 16; p.steps = steps; p.R = R; p.P = P; p.direct = direct;
      uint4 *x, *ring, *w, *out, *ts = nullptr;
      unsigned* ctl;
      CK(hipMalloc(&x, (size_t)steps * 8 * slab)); CK(hipMemset(x, 1, (size_t)steps * 8 * slab));
      CK(hipMalloc(&out, (size_t)steps * 8 * slab));
      CK(hipMalloc(&ring, (size_t)8 * R * 2 * slab)); CK(hipMemset(ring, 0xff, (size_t)8 * R * 2 * slab));
      CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 3, wbytes));
      if (direct) { CK(hipMalloc(&ts, (size_t)steps * 8 * 2 * slab)); CK(hipMemset(ts, 0xff, (size_t)steps * 8 * 2 * slab)); }
      CK(hipMalloc(&ctl, (8 * 64 + 16) * 4));
      p.x = x; p.ring = ring; p.w = w; p.out = out; p.ctl = ctl; p.tstream = ts;
      float best = 1e30f;
      unsigned host[8 * 64 + 16];
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctl, 0, (8 * 64 + 16) * 4));
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(handover, dim3(256), dim3(256), 0, 0, p);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
        CK(hipMemcpy(host, ctl, sizeof(host), hipMemcpyDeviceToHost));
      }
      const double strip = 8.0 * slab * steps;   // bytes of ONE tensor over the launch
      printf("handover %-26s tile %3zu KB (ring %5.2f MB/XCD + %0.2f MB weights), %d steps: %8.1f us; blocks on XCD 0: %u; timeout %u, stale %u of %u words; algorithmic: x %0.0f + out %0.0f MB%s\n",
             direct ? "through memory (as today)" : (R == 2 ? "ring of 2 strips" : "ring of 4 strips"), tile >> 10, direct ? 0.0 : R * 2.0 * slab / 1e6, wbytes / 1e6, steps, best * 1e3, host[0],
             host[8 * 64 + 1], host[8 * 64 + 2], host[8 * 64 + 3], strip / 1e6, strip / 1e6, direct ? " + t, x' written and read back: 4x as much" : "");
      CK(hipFree(x)); CK(hipFree(out)); CK(hipFree(ring)); CK(hipFree(w)); CK(hipFree(ctl));
      if (ts) CK(hipFree(ts));
    }
  }
  return 0;
}
