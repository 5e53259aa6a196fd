// decoder_kernels.hip -- the HBM-bound glue of the Qwen2 decoder ([site] transformers/models/qwen2/modeling_qwen2.py):
//   embed gather (+ optional image-token splice)   embed_tokens, LLaVA prepare_inputs_labels_for_multimodal
//   RMSNorm fp32 -> bf16 GEMM operand              :247-252
//   rotate-half RoPE, in place on packed qkv       :105-135
//   last-token / mean pooling + final RMSNorm      reference model/fastvlm_adapter.py:337-359,551-559
// The residual stream is fp32 [tokens][H]; one wave per row, 16-byte accesses.
#include "kernels.h"

namespace fv {
#define FV_TRY_RC(expr) do { const int rc_ = (expr); if (rc_ != FV_OK) return rc_; } while (0)
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

// y (+ y_lo when given: x ~= y + y_lo to 16 significant bits, the split-bf16 operand of the parity-mode GEMMs)
template <bool F16>   // F16: y receives fp16 bits (11 significant bits: the single-pass operand of llm_precision = 2's gate/up GEMM)
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       bf16_t* __restrict__ y, bf16_t* __restrict__ y_lo, int ldy, int rows,
                                                       int H, float eps, unsigned* __restrict__ sat, int lo8) {
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
    if constexpr (F16) {
      count_f16_sat8(o, sat);
      *reinterpret_cast<uint4*>(y + row * ldy + i) = pack8_h(o);
      continue;
    }
    const uint4 hi = pack8(o);
    *reinterpret_cast<uint4*>(y + row * ldy + i) = hi;
    if (y_lo) {
      float h8[8], l8[8];
      unpack8(hi, h8);
#pragma unroll
      for (int e = 0; e < 8; ++e) l8[e] = o[e] - h8[e];
      if (lo8) *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(y_lo + row * ldy) + i) = pack_lo8(l8);   // fp8 remainders: one byte each
      else *reinterpret_cast<uint4*>(y_lo + row * ldy + i) = pack8(l8);
    }
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

// fp32 variant of the RoPE for the parity-mode decoder (qkv kept in fp32): thread = (row, head, 4 d of the first half)
__global__ __launch_bounds__(256) void rope_f32_kernel(float* __restrict__ qkv, const float2* __restrict__ cs, int ld,
                                                        long rows, int T, int nheads_total, int D) {
  const int per_head = D / 8;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * nheads_total * per_head) return;
  const int c = (int)(i % per_head);
  const int hh = (int)((i / per_head) % nheads_total);
  const long row = i / ((long)per_head * nheads_total);
  const int pos = (int)(row % T);
  float* p1 = qkv + row * ld + hh * D + c * 4;
  float* p2 = p1 + D / 2;
  const float4 a = *reinterpret_cast<const float4*>(p1), b = *reinterpret_cast<const float4*>(p2);
  const float2* t = cs + (size_t)pos * (D / 2) + c * 4;
  const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
  float oa[4], ob[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    oa[e] = av[e] * t[e].x - bv[e] * t[e].y;
    ob[e] = bv[e] * t[e].x + av[e] * t[e].y;
  }
  *reinterpret_cast<float4*>(p1) = make_float4(oa[0], oa[1], oa[2], oa[3]);
  *reinterpret_cast<float4*>(p2) = make_float4(ob[0], ob[1], ob[2], ob[3]);
}

// fp32 causal GQA attention for the parity-mode decoder (T <= a few hundred; 22 GFLOP per step at T = 64): one thread
// (or NT = D/DPT adjacent lanes) per (query position, q head); K/V chunks of 64 keys staged in LDS as fp32 and read as
// wave-wide broadcasts; online softmax over groups of 8 keys; output written as bf16 hi + lo for the o-projection.
template <int DPT, int NT>
__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ qkv, bf16_t* __restrict__ out_hi,
                                                             bf16_t* __restrict__ out_lo, const int32_t* __restrict__ lens,
                                                             int len_add, int ld, int ldo, int T, int heads, int kv_heads,
                                                             int QT, float scale, int lo8) {
  constexpr int D = DPT * NT;
  extern __shared__ __attribute__((aligned(16))) float skv[];  // [2][64][D]
  float* sK = skv;
  float* sV = skv + 64 * D;
  const int G = heads / kv_heads;
  const int qtiles = (T + QT - 1) / QT;
  int bid = blockIdx.x;
  const int qt = bid % qtiles; bid /= qtiles;
  const int hk = bid % kv_heads;
  const int b = bid / kv_heads;
  const int tid = threadIdx.x;
  const int part = tid % NT, rowid = tid / NT;       // rowid in [0, QT*G)
  const int pos_l = rowid % QT, hl = rowid / QT;
  const bool active = hl < G;
  const int pos = qt * QT + pos_l;
  const int h = hk * G + (active ? hl : 0);
  int len = lens ? lens[b] + len_add : T;
  len = max(1, min(len, T));
  const int qd = heads * D, kd = kv_heads * D;
  float q[DPT], acc[DPT];
  {
    const float* qp = qkv + ((size_t)b * T + min(pos, T - 1)) * ld + h * D + part * DPT;
#pragma unroll
    for (int d = 0; d < DPT; d += 4) {
      const float4 v = *reinterpret_cast<const float4*>(qp + d);
      q[d] = v.x * scale; q[d + 1] = v.y * scale; q[d + 2] = v.z * scale; q[d + 3] = v.w * scale;
    }
#pragma unroll
    for (int d = 0; d < DPT; ++d) acc[d] = 0.f;
  }
  float m_run = -1e30f, l_run = 0.f;
  const int kend = min(len, qt * QT + QT);   // causal: keys beyond the tile's last query are never visible
  for (int k0 = 0; k0 < kend; k0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * D / 4; i += 256) {
      const int key = i / (D / 4), c4 = i % (D / 4);
      const int krow = min(k0 + key, T - 1);
      const float* base = qkv + ((size_t)b * T + krow) * ld + qd + hk * D + c4 * 4;
      *reinterpret_cast<float4*>(sK + key * D + c4 * 4) = *reinterpret_cast<const float4*>(base);
      *reinterpret_cast<float4*>(sV + key * D + c4 * 4) = *reinterpret_cast<const float4*>(base + kd);
    }
    __syncthreads();
    const int kmax = min(64, kend - k0);
    for (int j0 = 0; j0 < kmax; j0 += 8) {
      float sc[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const float* kr = sK + (j0 + jj) * D + part * DPT;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < DPT; d += 4) {
          const float4 kv = *reinterpret_cast<const float4*>(kr + d);
          s += q[d] * kv.x + q[d + 1] * kv.y + q[d + 2] * kv.z + q[d + 3] * kv.w;
        }
        if (NT == 2) s += __shfl_xor(s, 1, 64);
        const int kg = k0 + j0 + jj;
        sc[jj] = (kg <= pos && kg < len) ? s : -1e30f;
      }
      float gm = sc[0];
#pragma unroll
      for (int jj = 1; jj < 8; ++jj) gm = fmaxf(gm, sc[jj]);
      const float m_new = fmaxf(m_run, gm);
      const float alpha = __expf(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DPT; ++d) acc[d] *= alpha;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const float pv = sc[jj] > -1e29f ? __expf(sc[jj] - m_new) : 0.f;
        l_run += pv;
        const float* vr = sV + (j0 + jj) * D + part * DPT;
#pragma unroll
        for (int d = 0; d < DPT; d += 4) {
          const float4 vv = *reinterpret_cast<const float4*>(vr + d);
          acc[d] += pv * vv.x; acc[d + 1] += pv * vv.y; acc[d + 2] += pv * vv.z; acc[d + 3] += pv * vv.w;
        }
      }
    }
  }
  if (!active || pos >= T) return;
  const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
  const size_t o = ((size_t)b * T + pos) * ldo + h * D + part * DPT;
#pragma unroll
  for (int d = 0; d < DPT; d += 8) {
    float v8[8], h8[8], l8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v8[e] = acc[d + e] * inv;
    const uint4 hv = pack8(v8);
    unpack8(hv, h8);
#pragma unroll
    for (int e = 0; e < 8; ++e) l8[e] = v8[e] - h8[e];
    *reinterpret_cast<uint4*>(out_hi + o + d) = hv;
    if (lo8) *reinterpret_cast<uint2*>(reinterpret_cast<uint8_t*>(out_lo + ((size_t)b * T + pos) * ldo) + h * D + part * DPT + d) = pack_lo8(l8);
    else *reinterpret_cast<uint4*>(out_lo + o + d) = pack8(l8);
  }
}

// The same attention on the fp32 matrix core (v_mfma_f32_16x16x4_f32: fp32 operands, exact products, fp32 accumulation --
// the VALU kernel above with another summation order), for head_dim 64 / 128.  The VALU form hands every lane whole K and
// V rows as LDS broadcasts (1 KB of register fill per 64 FMAs) and is bound by that; here a 16-key x 16-query score tile
// costs D/16 ds_read_b128 + D/4 MFMAs, and its P^T accumulators feed the second product directly:
//   S^T = K . Q^T   A = K[key fr][d = 16c + 4fg + e] (one b128 read = four MFMAs' A operands), B = the same d of Q[query fr]
//   O^T += V^T . P^T   B of MFMA r = p[r] (D layout: row 4fg + r = key  <->  k-slot fg), A = V[key 4fg + r][d = 16dt + fr]
// Block = 4 waves = 64 queries of one (batch, q head); K/V chunks of KCH keys in LDS, rows padded by 4 floats so that both
// read patterns are conflict-free.  Softmax bookkeeping is ~40 VALU issues per tile against 2 * D/4 MFMAs of 32 cycles.
template <int D>
__global__ __launch_bounds__(256, 2) void attention_f32_mfma_kernel(const float* __restrict__ qkv, bf16_t* __restrict__ out_hi,
                                                                     bf16_t* __restrict__ out_lo, const int32_t* __restrict__ lens,
                                                                     int len_add, int ld, int ldo, int T, int heads, int kv_heads,
                                                                     float scale, const float2* __restrict__ rope,
                                                                     const float* __restrict__ pre, int ldp, int Np, float* __restrict__ lse, int lo8) {
  // lse != null (training, Np == 0): the row statistics max + log(sum) of every (batch, head, query) for attn_bwd_* (train_kernels.hip)
  // pre != null (image-prefix reuse): the sequence is Np cached prefix positions + Ts = T - Np new ones.  qkv and
  // the outputs hold ONLY the new rows (row b * Ts + t - Np); keys / values of positions < Np come from `pre`, rows
  // (b * Np + pos) of [k (kv_heads * D) | v (kv_heads * D)] fp32, UN-rotated like qkv; queries exist for positions >= Np only.
  // rope != null: qkv holds the UN-rotated projections and the rotate-half RoPE ([site] modeling_qwen2.py:105-135) is applied
  // here, to the query fragments in registers and to the K rows on their way into LDS (position = index in the sequence;
  // table [pos][D/2] of (cos, sin)) -- one launch and one read-modify-write pass over q and k less per layer
  constexpr int DT = D / 16;               // 16-row tiles of O^T, and 16-float column groups of a K row
  constexpr int KCH = D == 64 ? 64 : 32;   // keys per LDS chunk
  constexpr int LDR = D + 4;               // padded row (floats)
  __shared__ __attribute__((aligned(16))) float sK[KCH * LDR];
  __shared__ __attribute__((aligned(16))) float sV[KCH * LDR];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int Ts = T - Np;                   // rows per batch entry in qkv / out (Np = 0 without a prefix)
  const int qblocks = (Ts + 63) >> 6;
  int bid = blockIdx.x;
  const int qb = bid % qblocks; bid /= qblocks;
  const int h = bid % heads;
  const int b = bid / heads;
  const int hk = h / (heads / kv_heads);
  int len = lens ? lens[b] + len_add : T;
  len = max(1, min(len, T));
  const int q0 = Np + qb * 64 + wid * 16, qg = q0 + fr;
  const int qd = heads * D, kd = kv_heads * D;

  float4 fq[DT];
  {
    const float* qp = qkv + ((size_t)b * Ts + (min(qg, T - 1) - Np)) * ld + h * D + 4 * fg;
#pragma unroll
    for (int c = 0; c < DT; ++c) fq[c] = *reinterpret_cast<const float4*>(qp + 16 * c);
    if (rope) {   // d = 16 c + 4 fg + e pairs with d + D/2 = 16 (c + DT/2) + 4 fg + e: the same lane
      const float2* t = rope + (size_t)min(qg, T - 1) * (D / 2) + 4 * fg;
#pragma unroll
      for (int c = 0; c < DT / 2; ++c) {
        const float4 cs0 = *reinterpret_cast<const float4*>(t + 16 * c), cs1 = *reinterpret_cast<const float4*>(t + 16 * c + 2);
        const float4 a = fq[c], b = fq[c + DT / 2];
        fq[c] = make_float4(a.x * cs0.x - b.x * cs0.y, a.y * cs0.z - b.y * cs0.w, a.z * cs1.x - b.z * cs1.y, a.w * cs1.z - b.w * cs1.w);
        fq[c + DT / 2] = make_float4(b.x * cs0.x + a.x * cs0.y, b.y * cs0.z + a.y * cs0.w, b.z * cs1.x + a.z * cs1.y, b.w * cs1.z + a.w * cs1.w);
      }
    }
  }
  f32x4 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;

  const int kend = min(len, Np + qb * 64 + 64);   // causal: keys beyond the block's last query are never visible
  for (int k0 = 0; k0 < kend; k0 += KCH) {
    __syncthreads();
    if (rope) {   // a thread takes the four d of the first half AND their partners d + D/2 of a key row
      for (int i = tid; i < KCH * D / 8; i += 256) {
        const int key = i / (D / 8), c4 = i % (D / 8);
        const int krow = min(k0 + key, T - 1);
        const int kvo = krow < Np ? kv_heads * D : kd;   // distance from a row's k to its v: [k | v] in the prefix cache, q | k | v in qkv
        const float* base = krow < Np ? pre + ((size_t)b * Np + krow) * ldp + hk * D + c4 * 4
                                      : qkv + ((size_t)b * Ts + (krow - Np)) * ld + qd + hk * D + c4 * 4;
        const float4 a = *reinterpret_cast<const float4*>(base), bb = *reinterpret_cast<const float4*>(base + D / 2);
        const float2* t = rope + (size_t)krow * (D / 2) + c4 * 4;
        const float4 cs0 = *reinterpret_cast<const float4*>(t), cs1 = *reinterpret_cast<const float4*>(t + 2);
        *reinterpret_cast<float4*>(sK + key * LDR + c4 * 4) =
            make_float4(a.x * cs0.x - bb.x * cs0.y, a.y * cs0.z - bb.y * cs0.w, a.z * cs1.x - bb.z * cs1.y, a.w * cs1.z - bb.w * cs1.w);
        *reinterpret_cast<float4*>(sK + key * LDR + D / 2 + c4 * 4) =
            make_float4(bb.x * cs0.x + a.x * cs0.y, bb.y * cs0.z + a.y * cs0.w, bb.z * cs1.x + a.z * cs1.y, bb.w * cs1.z + a.w * cs1.w);
        *reinterpret_cast<float4*>(sV + key * LDR + c4 * 4) = *reinterpret_cast<const float4*>(base + kvo);
        *reinterpret_cast<float4*>(sV + key * LDR + D / 2 + c4 * 4) = *reinterpret_cast<const float4*>(base + kvo + D / 2);
      }
    } else {
      for (int i = tid; i < KCH * D / 4; i += 256) {
        const int key = i / (D / 4), c4 = i % (D / 4);
        const int krow = min(k0 + key, T - 1);
        const int kvo = krow < Np ? kv_heads * D : kd;
        const float* base = krow < Np ? pre + ((size_t)b * Np + krow) * ldp + hk * D + c4 * 4
                                      : qkv + ((size_t)b * Ts + (krow - Np)) * ld + qd + hk * D + c4 * 4;
        *reinterpret_cast<float4*>(sK + key * LDR + c4 * 4) = *reinterpret_cast<const float4*>(base);
        *reinterpret_cast<float4*>(sV + key * LDR + c4 * 4) = *reinterpret_cast<const float4*>(base + kvo);
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < KCH / 16; ++kt) {
      const int kb = k0 + kt * 16;
      if (kb > q0 + 15 || kb >= len) break;   // wave-uniform: the rest of the chunk is above the diagonal / past the prompt
      f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* kr = sK + (kt * 16 + fr) * LDR + 4 * fg;
#pragma unroll
      for (int c = 0; c < DT; ++c) {
        const float4 kf = *reinterpret_cast<const float4*>(kr + 16 * c);
        sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, fq[c].x, sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, fq[c].y, sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, fq[c].z, sacc, 0, 0, 0);
        sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, fq[c].w, sacc, 0, 0, 0);
      }
      float sc[4], mloc = -1e30f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int kg = kb + 4 * fg + r;
        sc[r] = (kg <= qg && kg < len) ? sacc[r] * scale : -1e30f;
        mloc = fmaxf(mloc, sc[r]);
      }
      mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
      mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
      const float m_new = fmaxf(m_run, mloc);
      const float alpha = __expf(m_run - m_new);
      m_run = m_new;
      float pv[4], psum = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv[r] = sc[r] > -1e29f ? __expf(sc[r] - m_new) : 0.f;
        psum += pv[r];
      }
      l_run = l_run * alpha + psum;            // per-lane partial; the four key groups are summed at the end
      const float* vr = sV + (kt * 16 + 4 * fg) * LDR + fr;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vr[r * LDR + 16 * dt], pv[r], o[dt], 0, 0, 0);
      }
    }
  }
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (qg >= T) return;
  const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
  if (lse && fg == 0) lse[((size_t)b * heads + h) * T + qg] = m_run + __logf(fmaxf(l_run, 1e-37f));
  const size_t ob = ((size_t)b * Ts + (qg - Np)) * ldo + h * D + 4 * fg;   // lane: query qg, d = 16 dt + 4 fg + r
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    float v4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v4[r] = o[dt][r] * inv;
    uint2 hv, lv;
    hv.x = pack_bf2(v4[0], v4[1]); hv.y = pack_bf2(v4[2], v4[3]);
    *reinterpret_cast<uint2*>(out_hi + ob + 16 * dt) = hv;
    if (lo8) {   // fp8 remainders (x 2^8), one byte each, in the lo half's place
      *reinterpret_cast<uint32_t*>(reinterpret_cast<uint8_t*>(out_lo + ((size_t)b * Ts + (qg - Np)) * ldo) + h * D + 4 * fg + 16 * dt) =
          pack_f8x4((v4[0] - bf_lo(hv.x)) * FV_LO8_SCALE, (v4[1] - bf_hi(hv.x)) * FV_LO8_SCALE, (v4[2] - bf_lo(hv.y)) * FV_LO8_SCALE, (v4[3] - bf_hi(hv.y)) * FV_LO8_SCALE);
      continue;
    }
    lv.x = pack_bf2(v4[0] - bf_lo(hv.x), v4[1] - bf_hi(hv.x));
    lv.y = pack_bf2(v4[2] - bf_lo(hv.y), v4[3] - bf_hi(hv.y));
    *reinterpret_cast<uint2*>(out_lo + ob + 16 * dt) = lv;
  }
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

int launch_rmsnorm(const float* x, const float* w, bf16_t* y, bf16_t* y_lo, int ldy, int rows, int H, float eps, hipStream_t s, int f16, unsigned* sat, int lo8) {
  if (!x || !w || !y) return fv_fail(FV_ERR_ARG, "rmsnorm: null pointer");
  if (rows <= 0 || H <= 0 || H % 8 || ldy < H || ldy % 8) return fv_fail(FV_ERR_ARG, "rmsnorm: bad shape rows=%d H=%d ldy=%d", rows, H, ldy);
  if (f16 && y_lo) return fv_fail(FV_ERR_ARG, "rmsnorm: the fp16 form has no remainder output");
  if (f16) hipLaunchKernelGGL(rmsnorm_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, y, y_lo, ldy, rows, H, eps, sat, 0);
  else hipLaunchKernelGGL(rmsnorm_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, s, x, w, y, y_lo, ldy, rows, H, eps, nullptr, lo8);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

namespace {
__global__ __launch_bounds__(256) void bf16_to_f16_kernel(bf16_t* __restrict__ p, size_t n8, float scale, unsigned* __restrict__ maxbits) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float f[8];
    unpack8(reinterpret_cast<const uint4*>(p)[i], f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      f[e] *= scale;
      const float af = fabsf(f[e]);
      if (!(af <= m)) m = af == af ? af : __builtin_inff();   // a NaN weight reads as "out of range" too
    }
    reinterpret_cast<uint4*>(p)[i] = pack8_h(f);
  }
  if (maxbits && m > 0.f) atomicMax(maxbits, __float_as_uint(m));   // non-negative floats order like their bit patterns
}
}  // namespace

namespace {
// W [N][K] bf16 -> W8 [N][2K bytes]: fp8 e4m3 of W x 2^6 in the first K bytes of each row (the second K bytes are never read: the row
// stride equals the bf16 copy's so that gemm256_kernel's per-lane row offsets serve both copies)
__global__ __launch_bounds__(256) void bf16_to_w8_kernel(const bf16_t* __restrict__ w, uint8_t* __restrict__ w8, long n8, int K, unsigned* __restrict__ maxbits) {
  float m = 0.f;
  const int k8 = K >> 3;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
    const long row = i / k8;
    const int c = (int)(i % k8) * 8;
    float f[8];
    unpack8(*reinterpret_cast<const uint4*>(w + row * K + c), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      f[e] *= FV_W8_SCALE;
      const float af = fabsf(f[e]);
      if (!(af <= m)) m = af == af ? af : __builtin_inff();
    }
    uint2 u;
    u.x = pack_f8x4(f[0], f[1], f[2], f[3]);
    u.y = pack_f8x4(f[4], f[5], f[6], f[7]);
    *reinterpret_cast<uint2*>(w8 + row * 2 * K + c) = u;
  }
  if (maxbits && m > 0.f) atomicMax(maxbits, __float_as_uint(m));
}
}  // namespace

int launch_bf16_to_w8(const bf16_t* w, void* w8, size_t rows, int K, hipStream_t s, unsigned* maxbits) {
  if (!w || !w8 || rows == 0 || K <= 0 || K % 8 || (((uintptr_t)w | (uintptr_t)w8) & 15)) return fv_fail(FV_ERR_ARG, "bf16_to_w8: bad arguments");
  const long n8 = (long)rows * (K / 8);
  const unsigned blocks = (unsigned)((n8 + 255) / 256 < 65536 ? (n8 + 255) / 256 : 65536);
  hipLaunchKernelGGL(bf16_to_w8_kernel, dim3(blocks), dim3(256), 0, s, w, static_cast<uint8_t*>(w8), n8, K, maxbits);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_bf16_to_f16(bf16_t* p, size_t n, float scale, hipStream_t s, unsigned* maxbits) {
  if (!p || n == 0 || n % 8 || ((uintptr_t)p & 15)) return fv_fail(FV_ERR_ARG, "bf16_to_f16: n must be a positive multiple of 8 and p 16-byte aligned");
  const size_t n8 = n / 8;
  const unsigned blocks = (unsigned)((n8 + 255) / 256 < 65536 ? (n8 + 255) / 256 : 65536);
  hipLaunchKernelGGL(bf16_to_f16_kernel, dim3(blocks), dim3(256), 0, s, p, n8, scale, maxbits);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_rope_f32(float* qkv, const float2* table, int ld, int rows, int T, int heads, int kv_heads, int D, hipStream_t s) {
  if (!qkv || !table) return fv_fail(FV_ERR_ARG, "rope_f32: null pointer");
  if (rows <= 0 || T <= 0 || D % 8 || ld % 4 || ld < (heads + kv_heads) * D) return fv_fail(FV_ERR_ARG, "rope_f32: bad shape");
  const long total = (long)rows * (heads + kv_heads) * (D / 8);
  hipLaunchKernelGGL(rope_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, qkv, table, ld, (long)rows, T, heads + kv_heads, D);
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

int launch_attention_f32(float* qkv, int ld, bf16_t* out_hi, bf16_t* out_lo, int ldo, int B, int T, int heads,
                         int kv_heads, int D, const int32_t* lens, int len_add, float scale, hipStream_t s, const float2* rope,
                         const float* pre, int ldp, int Np, float* lse, int lo8, void* split_scratch) {
  if (!qkv || !out_hi || !out_lo) return fv_fail(FV_ERR_ARG, "attention_f32: null pointer");
  if (lse && (pre || D < 64)) return fv_fail(FV_ERR_UNSUPPORTED, "attention_f32: row statistics (training) need head_dim 64 / 128 and no cached prefix");
  if (pre && (Np <= 0 || Np >= T || ldp < 2 * kv_heads * D || ldp % 4 || D < 64))
    return fv_fail(FV_ERR_UNSUPPORTED, "attention_f32: a cached prefix needs 0 < Np < T, ldp >= 2 * kv_heads * D and head_dim 64 / 128");
  if (!pre) Np = 0;
  if (ldo < heads * D || ldo % 8) return fv_fail(FV_ERR_ARG, "attention_f32: bad ldo");
  if (B <= 0 || T <= 0 || heads <= 0 || kv_heads <= 0 || heads % kv_heads || ld % 4 || ld < (heads + 2 * kv_heads) * D)
    return fv_fail(FV_ERR_ARG, "attention_f32: bad shape");
  const int G = heads / kv_heads;
  const int NT = D == 128 ? 2 : 1;
  if (D != 32 && D != 64 && D != 128) return fv_fail(FV_ERR_UNSUPPORTED, "attention_f32: head_dim %d not in {32,64,128}", D);
  static const bool no_mfma = fv_ab_env("FASTVLA_NO_ATTN_F32_MFMA") != nullptr;
  static const bool no_split = fv_ab_env("FASTVLA_NO_ATTN_SPLIT") != nullptr;   // A/B: the fp32-MFMA kernels of round 4's first version
  // training (lse wanted): the split-bf16 kernel -- 5x the fp32 pipe's rate at 16 significant bits per operand (attention_split.hip)
  if (D >= 64 && (lse || T >= FV_ATTN_SPLIT_MIN_T) && !pre && !lo8 && !no_split && split_scratch)
    return launch_attention_split_fwd(qkv, ld, out_hi, out_lo, ldo, B, T, heads, kv_heads, D, lens, len_add, scale, rope, lse, split_scratch, s);
  if (D >= 64 && (!no_mfma || pre || lse)) {
    const long nb = (long)B * heads * ((T - Np + 63) / 64);
    // rope given: q and k are rotated inside the kernel (no separate pass over the packed projections)
    if (D == 64) hipLaunchKernelGGL((attention_f32_mfma_kernel<64>), dim3((unsigned)nb), dim3(256), 0, s, qkv, out_hi, out_lo, lens, len_add, ld, ldo, T, heads, kv_heads, scale, rope, pre, ldp, Np, lse, lo8);
    else hipLaunchKernelGGL((attention_f32_mfma_kernel<128>), dim3((unsigned)nb), dim3(256), 0, s, qkv, out_hi, out_lo, lens, len_add, ld, ldo, T, heads, kv_heads, scale, rope, pre, ldp, Np, lse, lo8);
    FV_HIP_CHECK(hipGetLastError());
    return FV_OK;
  }
  if (G * NT > 256) return fv_fail(FV_ERR_UNSUPPORTED, "attention_f32: too many q heads per kv head (%d)", G);
  if (rope) FV_TRY_RC(launch_rope_f32(qkv, rope, ld, B * T, T, heads, kv_heads, D, s));   // the VALU kernel reads rotated q / k
  // queries per block: as many as 256 threads hold (any count, not a power of two: G = 7 leaves 44 % of the lanes idle at
  // 16), then levelled over the tiles so the causal key ranges of the tiles are balanced
  int QT = 256 / (G * NT);
  if (QT > T) QT = T;
  const int qtiles = (T + QT - 1) / QT;
  QT = (T + qtiles - 1) / qtiles;
  const long blocks = (long)B * kv_heads * ((T + QT - 1) / QT);
  const size_t lds = (size_t)2 * 64 * D * sizeof(float);
  const dim3 grid((unsigned)blocks), blk(256);
  if (D == 32) hipLaunchKernelGGL((attention_f32_kernel<32, 1>), grid, blk, lds, s, qkv, out_hi, out_lo, lens, len_add, ld, ldo, T, heads, kv_heads, QT, scale, lo8);
  else if (D == 64) hipLaunchKernelGGL((attention_f32_kernel<64, 1>), grid, blk, lds, s, qkv, out_hi, out_lo, lens, len_add, ld, ldo, T, heads, kv_heads, QT, scale, lo8);
  else hipLaunchKernelGGL((attention_f32_kernel<64, 2>), grid, blk, lds, s, qkv, out_hi, out_lo, lens, len_add, ld, ldo, T, heads, kv_heads, QT, scale, lo8);
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
