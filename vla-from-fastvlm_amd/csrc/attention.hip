// attention.hip -- flash-style multi-head attention on MFMA for both users of the path:
//   * FastViT-HD MHSA (mci.py MHSA.forward): non-causal, head_dim 32, N = 1024 / 256 tokens
//   * Qwen2 self-attention ([site] modeling_qwen2.py:150-172,195-234): causal, GQA, head_dim 64 / 128, softmax in
//     fp32, keys >= len masked (right padding)
//
// Formulation (per wave: 16 queries; per block: 4 waves = 64 queries of one (batch, head); K/V tiles of 64 keys in LDS):
//   S^T = K . Q^T        A = K tile rows (ds_read_b128, key on the MFMA row), B = Q fragment held in registers
//   P^T stays in registers: the C/D map (col = lane&15 = query, row = 4*(lane>>4)+r = key) is already the B operand
//                        of the second product if the k-slot <-> key assignment is permuted the same way on V
//   O^T = V^T . P^T      A = V^T read from a transposed LDS image Vt[d][key] with two ds_read_b64 per fragment
// so the softmax needs only two cross-lane shuffles (xor 16, 32) per row statistic and P never touches LDS.
// Bounded by LDS/VALU (head_dim 32 gives only 8 MFMAs per 64-key tile); it is <3 % of the tower FLOPs.
#include "kernels.h"

namespace fv {
namespace {

struct AttnParams {
  const bf16_t *q, *k, *v; bf16_t* out; const int32_t* lens;
  int ldq, ldk, ldv, ldo, B, T, heads, kv_heads, causal, len_add; float scale;
};

// MASKED = false: no causal mask, no key-length mask and T % 64 == 0 (the FastViT-HD MHSA): the softmax then costs one
// FMA (scale * log2e folded in) and one v_exp_f32 per score -- with head_dim 32 this kernel is VALU-bound, not MFMA-bound.
template <int D, bool MASKED>
__global__ __launch_bounds__(256) void attention_kernel(AttnParams p) {
  constexpr int KS = D / 32;        // k-steps of the S^T product
  constexpr int DT = D / 16;        // 16-row tiles of O^T
  constexpr int KLD = D + 8;        // padded K row (elements)
  constexpr int VLD = 64 + 8;       // padded Vt row (elements)
  __shared__ __attribute__((aligned(16))) bf16_t smem[64 * KLD + D * VLD];
  bf16_t* sK = smem;
  bf16_t* sVt = smem + 64 * KLD;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int qblocks = (p.T + 63) >> 6;
  int bid = blockIdx.x;
  const int qb = bid % qblocks; bid /= qblocks;
  const int h = bid % p.heads;
  const int b = bid / p.heads;
  const int hk = h / (p.heads / p.kv_heads);
  int len = p.lens ? p.lens[b] + p.len_add : p.T;
  len = max(1, min(len, p.T));

  const int q0 = qb * 64 + wid * 16;
  const int qg = q0 + fr;                       // this lane's query (column of S^T / O^T)
  const int qrow = min(qg, p.T - 1);
  const bf16_t* qp = p.q + ((size_t)b * p.T + qrow) * p.ldq + h * D;
  bf16x8 fq[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) fq[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qp + ks * 32 + fg * 8));

  f32x4 o[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f, l_run = 0.f;

  int kend = len;                                // keys [0, kend) can be visible to this block
  if (p.causal) kend = min(kend, qb * 64 + 64);
  const int nkb = (kend + 63) >> 6;
  const bf16_t* kbase = p.k + (size_t)b * p.T * p.ldk + hk * D;
  const bf16_t* vbase = p.v + (size_t)b * p.T * p.ldv + hk * D;

  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();  // previous tile fully consumed
    // cooperative stage: K row-major, V transposed; 64 x D/8 chunks each
#pragma unroll
    for (int i = 0; i < (64 * D / 8 + 255) / 256; ++i) {
      const int c = tid + 256 * i;
      if (c < 64 * D / 8) {
        const int key = c / (D / 8), ch = c % (D / 8);
        const int krow = min(kb * 64 + key, p.T - 1);
        const uint4 ku = *reinterpret_cast<const uint4*>(kbase + (size_t)krow * p.ldk + ch * 8);
        *reinterpret_cast<uint4*>(sK + key * KLD + ch * 8) = ku;
        const uint4 vu = *reinterpret_cast<const uint4*>(vbase + (size_t)krow * p.ldv + ch * 8);
        const uint32_t vw[4] = {vu.x, vu.y, vu.z, vu.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) sVt[(ch * 8 + e) * VLD + key] = (bf16_t)(e & 1 ? vw[e >> 1] >> 16 : vw[e >> 1] & 0xffffu);
      }
    }
    __syncthreads();

    // S^T tiles: 4 x (16 keys x 16 queries)
    f32x4 sacc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      sacc[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 fk = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sK + (kt * 16 + fr) * KLD + ks * 32 + fg * 8));
        sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq[ks], sacc[kt], 0, 0, 0);
      }
    }
    // online softmax (per lane: query qg, keys kb*64 + kt*16 + 4*fg + r); scores are kept in the log2 domain
    const float c2 = p.scale * 1.4426950408889634f;
    float mloc = -1e30f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sv = sacc[kt][r] * c2;
        if (MASKED) {
          const int kg = kb * 64 + kt * 16 + fg * 4 + r;
          const bool ok = kg < len && (!p.causal || kg <= qg);
          sv = ok ? sv : -1e30f;
        }
        sacc[kt][r] = sv;
        mloc = fmaxf(mloc, sv);
      }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float m_new = fmaxf(m_run, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float pv = __builtin_amdgcn_exp2f(sacc[kt][r] - m_new);
        if (MASKED) pv = sacc[kt][r] > -1e29f ? pv : 0.f;
        sacc[kt][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;   // per-lane partial; reduced over the 4 key groups at the end
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[dt][r] *= alpha;

    // O^T += V^T . P^T : two k-steps of 32 keys; slot (fg, j) <-> key tile (2*ks2 + (j>>2))*16 + 4*fg + (j&3)
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      uint4 pu;
      pu.x = pack_bf2(sacc[2 * ks2][0], sacc[2 * ks2][1]);
      pu.y = pack_bf2(sacc[2 * ks2][2], sacc[2 * ks2][3]);
      pu.z = pack_bf2(sacc[2 * ks2 + 1][0], sacc[2 * ks2 + 1][1]);
      pu.w = pack_bf2(sacc[2 * ks2 + 1][2], sacc[2 * ks2 + 1][3]);
      const bf16x8 fp = __builtin_bit_cast(bf16x8, pu);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bf16_t* vr = sVt + (dt * 16 + fr) * VLD + fg * 4;
        const uint2 lo = *reinterpret_cast<const uint2*>(vr + (2 * ks2) * 16);
        const uint2 hi = *reinterpret_cast<const uint2*>(vr + (2 * ks2 + 1) * 16);
        const bf16x8 fv = __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fp, o[dt], 0, 0, 0);
      }
    }
  }

  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
  if (qg < p.T) {
    bf16_t* op = p.out + ((size_t)b * p.T + qg) * p.ldo + h * D;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      uint2 u;
      u.x = pack_bf2(o[dt][0] * inv, o[dt][1] * inv);
      u.y = pack_bf2(o[dt][2] * inv, o[dt][3] * inv);
      *reinterpret_cast<uint2*>(op + dt * 16 + fg * 4) = u;
    }
  }
}


// ---- the FastViT-HD MHSA specialisation: head_dim 32, no masks, T % 64 == 0 ------------------------------------------
// With 8 MFMAs per 64-key tile this shape is bound by the VALU work per score, so that is what is trimmed here:
//   * the running max is taken on the raw scores (v_max3, two per instruction); scale * log2e and the subtraction fold
//     into one FMA feeding v_exp_f32
//   * the row sum comes from the matrix core: a third A operand of ones beside the two V^T tiles sums the same
//     bf16-rounded P that the numerator uses (every row of that tile is the sum, so no cross-lane reduction either)
//   * the O rescale is skipped (wave-uniform branch) on tiles where no query's max moved: alpha is exactly 1 there
//   * V is staged row-major like K (one 16-byte write per thread) and read through ds_read_b64_tr_b16: the group of 16
//     lanes fg gets rows (keys) 4 fg .. 4 fg + 3 x 16 columns (d) delivered column-major, which is precisely the V^T
//     fragment of the permuted key order the P accumulators already have.  96-byte V rows keep those reads conflict-free.
//   * K/V tiles double-buffered in LDS with the next tile's global loads in flight during the products: one barrier/tile
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ float xor_max_32_16(float m) {
  // max over the four 16-lane rows (lane ^ 16, lane ^ 32) without LDS: swap halves / odd-even rows of two copies
  const auto a = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
  m = fmaxf(__builtin_bit_cast(float, a[0]), __builtin_bit_cast(float, a[1]));
  const auto b = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, m), __builtin_bit_cast(unsigned, m), false, false);
  return fmaxf(__builtin_bit_cast(float, b[0]), __builtin_bit_cast(float, b[1]));
}

__global__ __launch_bounds__(256, 2) void attention32_kernel(AttnParams p) {   // (, 2): <= 256 VGPRs, so the MFMAs take VGPR accumulators (no v_accvgpr copies)
  constexpr int D = 32, KLD = 40, VLD = 48, BUFE = 64 * KLD + 64 * VLD;
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * BUFE];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int qblocks = p.T >> 6;
  // the query blocks of one (batch, head) read the same K/V rows: a contiguous run of logical ids per XCD keeps them on one
  // L2 (dealt round-robin, each of the 8 L2s fetched every K/V tile for itself: 1.5 GB of HBM reads per launch by PMC)
  int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int qb = bid % qblocks; bid /= qblocks;
  const int h = bid % p.heads;
  const int b = bid / p.heads;
  const int hk = h / (p.heads / p.kv_heads);

  const int qg = qb * 64 + wid * 16 + fr;       // this lane's query (column of S^T / O^T)
  const bf16x8 fq = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p.q + ((size_t)b * p.T + qg) * p.ldq + h * D + fg * 8));

  f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, osum = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run = -1e30f;                          // raw-score domain
  const float c2 = p.scale * 1.4426950408889634f;
  const uint4 ones_u = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_u);

  const int nkb = p.T >> 6;
  const int skey = tid >> 2, sch = tid & 3;      // staging: this thread's key row and 16-byte chunk of it
  const bf16_t* kg = p.k + ((size_t)b * p.T + skey) * p.ldk + hk * D + sch * 8;
  const bf16_t* vg = p.v + ((size_t)b * p.T + skey) * p.ldv + hk * D + sch * 8;
  const size_t kstep = (size_t)64 * p.ldk, vstep = (size_t)64 * p.ldv;
  uint4 ku = *reinterpret_cast<const uint4*>(kg), vu = *reinterpret_cast<const uint4*>(vg);
  *reinterpret_cast<uint4*>(smem + skey * KLD + sch * 8) = ku;
  *reinterpret_cast<uint4*>(smem + 64 * KLD + skey * VLD + sch * 8) = vu;
  __syncthreads();

  for (int kb = 0; kb < nkb; ++kb) {
    const bf16_t* sK = smem + (kb & 1) * BUFE;
    const bf16_t* sV = sK + 64 * KLD;
    // the next tile's rows, in flight during the products (past the end: the last tile again, written to the idle buffer
    // and never read -- keeping the loop branch-free lets the wait for these loads sit at the LDS writes below)
    const int nxt = kb + 1 < nkb ? kb + 1 : kb;
    ku = *reinterpret_cast<const uint4*>(kg + nxt * kstep);
    vu = *reinterpret_cast<const uint4*>(vg + nxt * vstep);
    f32x4 sacc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const bf16x8 fk = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(sK + (kt * 16 + fr) * KLD + fg * 8));
      sacc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk, fq, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    float mloc = sacc[0][0];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, sacc[kt][r]);
    mloc = xor_max_32_16(mloc);
    const float m_new = fmaxf(m_run, mloc);
    if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c2);
#pragma unroll
      for (int r = 0; r < 4; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
      osum[0] *= alpha;
      m_run = m_new;
    }
    const float mc = -m_run * c2;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sacc[kt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt][r], c2, mc));

    // O^T += V^T . P^T : two k-steps of 32 keys; slot (fg, j) <-> key tile (2*ks2 + (j>>2))*16 + 4*fg + (j&3)
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) {
      uint4 pu;
      pu.x = pack_bf2(sacc[2 * ks2][0], sacc[2 * ks2][1]);
      pu.y = pack_bf2(sacc[2 * ks2][2], sacc[2 * ks2][3]);
      pu.z = pack_bf2(sacc[2 * ks2 + 1][0], sacc[2 * ks2 + 1][1]);
      pu.w = pack_bf2(sacc[2 * ks2 + 1][2], sacc[2 * ks2 + 1][3]);
      const bf16x8 fp = __builtin_bit_cast(bf16x8, pu);
      osum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fp, osum, 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        // lane 4q+p of the 16-lane group addresses row q (key 4 fg + q of the key tile), columns 4p .. 4p+3 of the d tile
        const bf16_t* vr = sV + ((2 * ks2) * 16 + 4 * fg + (fr >> 2)) * VLD + dt * 16 + 4 * (fr & 3);
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vr));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(vr + 16 * VLD));
        const uint2 lu = __builtin_bit_cast(uint2, lo), hu = __builtin_bit_cast(uint2, hi);
        const bf16x8 fv = __builtin_bit_cast(bf16x8, make_uint4(lu.x, lu.y, hu.x, hu.y));
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv, fp, o[dt], 0, 0, 0);
      }
    }
    bf16_t* nb = smem + ((kb + 1) & 1) * BUFE;
    *reinterpret_cast<uint4*>(nb + skey * KLD + sch * 8) = ku;
    *reinterpret_cast<uint4*>(nb + 64 * KLD + skey * VLD + sch * 8) = vu;
    __syncthreads();
  }

  const float inv = osum[0] > 0.f ? 1.0f / osum[0] : 0.f;   // every row of the ones tile holds this query's sum
  bf16_t* op = p.out + ((size_t)b * p.T + qg) * p.ldo + h * D;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    uint2 u;
    u.x = pack_bf2(o[dt][0] * inv, o[dt][1] * inv);
    u.y = pack_bf2(o[dt][2] * inv, o[dt][3] * inv);
    *reinterpret_cast<uint2*>(op + dt * 16 + fg * 4) = u;
  }
}

}  // namespace

int launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* v, int ldq, int ldk, int ldv, bf16_t* out,
                     int ldo, int B, int T, int heads, int kv_heads, int D, int causal, const int32_t* lens,
                     int len_add, float scale, hipStream_t s) {
  if (!q || !k || !v || !out) return fv_fail(FV_ERR_ARG, "attention: null pointer");
  if (B <= 0 || T <= 0 || heads <= 0 || kv_heads <= 0 || heads % kv_heads) return fv_fail(FV_ERR_ARG, "attention: bad shape B=%d T=%d heads=%d kv=%d", B, T, heads, kv_heads);
  if ((ldq | ldk | ldv) % 8 || ldo % 4) return fv_fail(FV_ERR_ARG, "attention: row strides must be multiples of 8 (out: 4)");
  if (ldq < heads * D || ldk < kv_heads * D || ldv < kv_heads * D || ldo < heads * D) return fv_fail(FV_ERR_ARG, "attention: row stride smaller than heads*D");
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) & 15 || ((uintptr_t)out & 7)) return fv_fail(FV_ERR_ARG, "attention: misaligned pointer");
  AttnParams p{q, k, v, out, lens, ldq, ldk, ldv, ldo, B, T, heads, kv_heads, causal, len_add, scale};
  const long blocks = (long)B * heads * ((T + 63) / 64);
  if (blocks > 0x7fffffffL) return fv_fail(FV_ERR_ARG, "attention: grid too large");
  const dim3 grid((unsigned)blocks), blk(256);
  const bool masked = causal || lens || (T & 63);
  if (D != 32 && D != 64 && D != 128) return fv_fail(FV_ERR_UNSUPPORTED, "attention: head_dim %d not in {32,64,128}", D);
  static const bool no32 = fv_ab_env("FASTVLA_NO_ATTN32") != nullptr;
  if (D == 32 && !masked && !no32) {
    hipLaunchKernelGGL(attention32_kernel, grid, blk, 0, s, p);
    FV_HIP_CHECK(hipGetLastError());
    return FV_OK;
  }
#define FV_ATT(D_)                                                                        \
  if (D == D_) {                                                                          \
    if (masked) hipLaunchKernelGGL((attention_kernel<D_, true>), grid, blk, 0, s, p);     \
    else hipLaunchKernelGGL((attention_kernel<D_, false>), grid, blk, 0, s, p);           \
  }
  FV_ATT(32) FV_ATT(64) FV_ATT(128)
#undef FV_ATT
  FV_HIP_CHECK(hipGetLastError());
  return FV_OK;
}

}  // namespace fv
