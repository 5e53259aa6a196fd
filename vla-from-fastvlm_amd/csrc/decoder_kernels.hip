// decoder_kernels.hip -- the HBM-bound glue of the Qwen2 decoder ([site] transformers/models/qwen2/modeling_qwen2.py):
//   embed gather (+ optional image-token splice)   embed_tokens, LLaVA prepare_inputs_labels_for_multimodal
//   RMSNorm fp32 -> bf16 GEMM operand              :247-252
//   rotate-half RoPE, in place on packed qkv       :105-135
//   last-token / mean pooling + final RMSNorm      reference model/fastvlm_adapter.py:337-359,551-559
// The residual stream is fp32 [tokens][H]; one wave per row, 16-byte accesses.
#include "kernels.h"

namespace fv {
namespace {

__global__ __launch_bounds__(256) void embed_gather_kernel(const int32_t* __restrict__ ids,
                                                            const bf16_t* __restrict__ table,
                                                            const float* __restrict__ img, float* __restrict__ x, int T,
                                                            int Ni, int H, int vocab, long rows) {
  // one wave per output row; rows = B * (Ni + T)
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int Tt = Ni + T;
  const long b = row / Tt;
  const int t = (int)(row % Tt);
  float* dst = x + row * H;
  if (t < Ni) {
    const float* src = img + (b * Ni + t) * H;
    for (int i = lane * 4; i < H; i += 256) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
  } else {
    int id = ids[b * T + (t - Ni)];
    id = min(max(id, 0), vocab - 1);
    const bf16_t* src = table + (size_t)id * H;
    for (int i = lane * 8; i < H; i += 512) {
      float v[8];
      unpack8(*reinterpret_cast<const uint4*>(src + i), v);
      *reinterpret_cast<float4*>(dst + i) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(dst + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}

__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       bf16_t* __restrict__ y, int rows, int H, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + row * H;
  float ss = 0.f;
  for (int i = lane * 8; i < H; i += 512) {
    const float4 a = *reinterpret_cast<const float4*>(xr + i), c = *reinterpret_cast<const float4*>(xr + i + 4);
    ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + c.x * c.x + c.y * c.y + c.z * c.z + c.w * c.w;
  }
  const float r = rsqrtf(wave_sum(ss) / (float)H + eps);
  for (int i = lane * 8; i < H; i += 512) {
    const float4 a = *reinterpret_cast<const float4*>(xr + i), c = *reinterpret_cast<const float4*>(xr + i + 4);
    const float4 wa = *reinterpret_cast<const float4*>(w + i), wc = *reinterpret_cast<const float4*>(w + i + 4);
    const float o[8] = {wa.x * (a.x * r), wa.y * (a.y * r), wa.z * (a.z * r), wa.w * (a.w * r),
                        wc.x * (c.x * r), wc.y * (c.y * r), wc.z * (c.z * r), wc.w * (c.w * r)};
    *reinterpret_cast<uint4*>(y + row * H + i) = pack8(o);
  }
}

// thread = (row, head, 8 consecutive d in the first half); cos/sin come from a host-built table [T][D/2] (float2)
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ qkv, const float2* __restrict__ cs, int ld,
                                                    long rows, int T, int nheads_total, int D) {
  const int per_head = D / 16;  // 8-wide chunks in half a head
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = rows * nheads_total * per_head;
  if (i >= total) return;
  const int c = (int)(i % per_head);
  const int hh = (int)((i / per_head) % nheads_total);
  const long row = i / ((long)per_head * nheads_total);
  const int pos = (int)(row % T);
  bf16_t* p1 = qkv + row * ld + hh * D + c * 8;
  bf16_t* p2 = p1 + D / 2;
  float a[8], b[8], oa[8], ob[8];
  unpack8(*reinterpret_cast<const uint4*>(p1), a);
  unpack8(*reinterpret_cast<const uint4*>(p2), b);
  const float2* t = cs + (size_t)pos * (D / 2) + c * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float co = t[e].x, si = t[e].y;
    oa[e] = a[e] * co - b[e] * si;   // q*cos + rotate_half(q)*sin, first half: -x2
    ob[e] = b[e] * co + a[e] * si;   // second half: +x1
  }
  *reinterpret_cast<uint4*>(p1) = pack8(oa);
  *reinterpret_cast<uint4*>(p2) = pack8(ob);
}

// one block per batch row: final RMSNorm on the pooled row(s).  mode 0 = last_token, 1 = mean over valid rows.
__global__ __launch_bounds__(256) void pool_norm_kernel(const float* __restrict__ x, const int32_t* __restrict__ lens,
                                                         const float* __restrict__ w, float* __restrict__ pooled,
                                                         int Ttot, int Ni, int H, float eps, int mode) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  int len = lens ? lens[b] : (Ttot - Ni);
  len = min(max(len, 0), Ttot - Ni);
  int r0, r1;
  if (mode == 0) { r0 = Ni + max(len - 1, 0); r1 = r0 + 1; } else { r0 = 0; r1 = Ni + len; }
  const float denom = mode == 0 ? 1.0f : fmaxf((float)(r1 - r0), 1e-6f);
  for (int i = tid; i < H; i += 256) pooled[(size_t)b * H + i] = 0.f;
  for (int r = r0; r < r1; ++r) {
    const float* xr = x + ((size_t)b * Ttot + r) * H;
    float ss = 0.f;
    for (int i = tid; i < H; i += 256) ss += xr[i] * xr[i];
    ss = wave_sum(ss);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float rs = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
    for (int i = tid; i < H; i += 256) pooled[(size_t)b * H + i] += w[i] * (xr[i] * rs) / denom;
  }
}

}  // namespace

int launch_embed_gather(const int32_t* ids, const bf16_t* table, const float* img_tokens, float* x, int B, int T,
                        int Ni, int H, int vocab, hipStream_t s) {
  if (!ids || !table || !x) return fv_fail(FV_ERR_ARG, "embed_gather: null pointer");
  if (Ni > 0 && !img_tokens) return fv_fail(FV_ERR_ARG, "embed_gather: Ni > 0 without image tokens");
  if (B <= 0 || T <= 0 || Ni < 0 || H % 8 || vocab <= 0) return fv_fail(FV_ERR_ARG, "embed_gather: bad shape");
  const long rows = (long)B * (Ni + T);
  hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, ids, table, img_tokens, x, T, Ni, H, vocab, rows);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_rmsnorm(const float* x, const float* w, bf16_t* y, int rows, int H, float eps, hipStream_t s) {
  if (!x || !w || !y) return fv_fail(FV_ERR_ARG, "rmsnorm: null pointer");
  if (rows <= 0 || H <= 0 || H % 8) return fv_fail(FV_ERR_ARG, "rmsnorm: bad shape rows=%d H=%d", rows, H);
  hipLaunchKernelGGL(rmsnorm_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, y, rows, H, eps);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

// inv_freq = theta^(-2i/D) and angle = pos * inv_freq in fp32, as Qwen2RotaryEmbedding does ([site] :85-104)
void rope_table_host(float* cs, int T, int D, float theta) {
  for (int pos = 0; pos < T; ++pos)
    for (int i = 0; i < D / 2; ++i) {
      const float inv = 1.0f / powf(theta, (float)(2 * i) / (float)D);
      const float ang = (float)pos * inv;
      cs[((size_t)pos * (D / 2) + i) * 2 + 0] = cosf(ang);
      cs[((size_t)pos * (D / 2) + i) * 2 + 1] = sinf(ang);
    }
}

int launch_rope(bf16_t* qkv, const float2* table, int ld, int rows, int T, int heads, int kv_heads, int D,
                hipStream_t s) {
  if (!qkv || !table) return fv_fail(FV_ERR_ARG, "rope: null pointer");
  if (rows <= 0 || T <= 0 || D % 16 || ld % 8 || ld < (heads + kv_heads) * D) return fv_fail(FV_ERR_ARG, "rope: bad shape");
  const long total = (long)rows * (heads + kv_heads) * (D / 16);
  hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, qkv, table, ld, (long)rows, T, heads + kv_heads, D);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_pool_norm(const float* x, const int32_t* lens, const float* w, float* pooled, int B, int Ttot, int Ni,
                     int H, float eps, int mode, hipStream_t s) {
  if (!x || !w || !pooled) return fv_fail(FV_ERR_ARG, "pool_norm: null pointer");
  if (B <= 0 || Ttot <= Ni || H <= 0 || (mode != 0 && mode != 1)) return fv_fail(FV_ERR_ARG, "pool_norm: bad shape/mode");
  hipLaunchKernelGGL(pool_norm_kernel, dim3(B), dim3(256), 0, s, x, lens, w, pooled, Ttot, Ni, H, eps, mode);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

}  // namespace fv
