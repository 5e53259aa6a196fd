// kernels.h -- internal launch API shared by the engine (engine.hip) and the op-level C entry points (ops_api.hip).
// Every launcher validates the shapes its kernel and grid assume and returns an fv_status; nothing here allocates or
// synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fastvla_hip.h"
#include "../../include/fastvla_hip_testops.h"   // enum fv_gemm_epilogue (the fv_op_* declarations in it are test-only: ops_api.hip)
#include "common.h"

namespace fv {

struct GemmArgs {
  const bf16_t* A; int lda;     // [M,K] bf16
  const bf16_t* W;              // [N,K] bf16 (row stride K)
  int M, N, K;
  const float* bias;            // [N] or null
  const float* scale;           // [N] or null (FV_EPI_LS_RES)
  const void* res; int ldr;     // residual (bf16 for LS_RES, f32 for RES_F32) or null
  void* out; int ldo;           // bf16 or f32 by epilogue
  int epi;
  int ksplit = 0;               // 1: A holds [hi | lo] (2K columns, lda >= 2K); out = (hi + lo) . W^T in one launch
                                // 2: "hi + lo8": A rows = K bf16 then (at byte offset 2K) K fp8 remainders x 2^8 (lda >= 1.5 K); the lo product runs
                                //    against W8 on the scaled fp8 MFMA (K % 128 == 0); SWIGLU_SPLIT / fused-norm outputs leave in the same form
  int f16 = 0;                  // 1: A and W hold fp16 bits (v_mfma_f32_16x16x32_f16); not together with ksplit
  float* splitk_ws = nullptr;   // optional scratch for split-K partial sums (fp32 epilogues, few output tiles, long K)
  size_t splitk_bytes = 0;
  int lo_off = 0;               // FV_EPI_SWIGLU_SPLIT: column of the lo half in an output row (0 = N / 2).  launch_gemm's tail sub-launch (a column range of the
                                // problem) sets it to the whole problem's N / 2
  int no_tail = 0;              // 1: do not cut the last partial round of tiles off as a K-range sub-launch (set on the two sub-launches themselves)
  int few_rows = 1;             // 0: never the few-row K-range forms (64-row tiles x K ranges; the control loop's shapes).  The training path clears it: those
                                // forms add in another fp32 order, and a row's gradient must not depend on how many rows share its step
  // optional RMSNorm of the fp32 output rows (the decoder's next-layer input_layernorm): y (+ y_lo) = bf16 hi (+ lo) of
  // w * out * rsqrt(mean(out^2) + eps), row stride norm_ld; fused into the split-K reducer, a separate launch otherwise
  const float* norm_w = nullptr; bf16_t* norm_y = nullptr; bf16_t* norm_ylo = nullptr; int norm_ld = 0; float norm_eps = 0.f;
  const void* W8 = nullptr;     // ksplit == 2: fp8 e4m3 copy of W x 2^6, row stride 2K BYTES (the first K of each row valid: the bf16 copy's per-lane offsets serve both)
  unsigned* sat = nullptr;      // optional device counter: += 1 per 8-value group FV_EPI_SWIGLU_F16 had to clamp to the fp16 range
  void* stash = nullptr;        // FV_EPI_SWIGLU_SPLIT only: also keep the raw gate/up accumulators [M][N] (row stride N, the packed column order):
                                // what the SwiGLU backward of the unfrozen training path differentiates; fp32, or
  int stash_f16 = 0;            // 1: fp16 (saturating, clamps counted in *sat): half the bytes of a launch whose epilogue is write-bound
  // tn = 1 ("TN": the contraction runs over the ROWS of both operands, the wgrad's shape): A is [K][M] with row stride lda >= M, W is [K][N]
  // with row stride ldw >= N, out[m][n] = sum_k A[k][m] W[k][n].  K % 64 == 0 (rows past the data must be zero), M % 8 == 0, N % 8 == 0, fp32
  // epilogues, no ksplit; the 256-tile kernel only (LDS image in [k/8][n/16][8][16] blocks, fragments by ds_read_b64_tr_b16)
  int tn = 0, ldw = 0;
};
int launch_gemm(const GemmArgs& a, hipStream_t s);

// fused ConvFFN pointwise half: out = res + ls * (fc2(gelu(fc1(x) + b1)) + b2); w1 [4C][C] bf16, w2p = convffn_pack_w2 layout
bool convffn_supported(int C, int ratio);
void convffn_pack_w2(const float* w2, float* out, int C, int hidden);
int launch_convffn(const bf16_t* x, const bf16_t* w1, const float* b1, const bf16_t* w2p, const float* b2, const float* ls,
                   const bf16_t* res, bf16_t* out, int M, int C, int hidden, hipStream_t s);

// the same fused ConvFFN on v_mfma_f32_32x32x16_bf16 (convffn32.hip): C in {96, 192, 384}; wq = convffn32_pack stream (fc1 and
// fc2 weights, [4C/32 chunks][64 C elements])
bool convffn32_supported(int C, int ratio);
void convffn32_pack(const float* w1, const float* w2, float* out, int C);
int launch_convffn32(const bf16_t* x, const bf16_t* wq, const float* b1, const float* b2, const float* ls,
                     const bf16_t* res, bf16_t* out, int M, int C, int hidden, hipStream_t s, float* part = nullptr, size_t part_bytes = 0,
                     bf16_t* stash_y = nullptr);   // stash_y: [M][4C] fp16, the training forward's pre-activation / 4 (convffn32.hip STASH)
bool convffn32_stash_supported(int M, int C);

int launch_letterbox(const void* img, int dtype, int B, int C, int Hin, int Win, int S, float pad_value, int letterbox,
                     bf16_t* pix, hipStream_t s);
int launch_letterbox_norm(const void* img, int dtype, int B, int C, int Hin, int Win, int S, float pad_value, int letterbox, const float* mean3,
                          const float* std3, int range_heuristic, unsigned* vmax_scratch, bf16_t* pix, hipStream_t s);
int launch_stem_conv(const bf16_t* pix, const float* w, const float* bias, bf16_t* y, int B, int S, int Cout,
                     hipStream_t s);
// implicit-GEMM stem on MFMA; wp = stem_mfma_pack image ([Cout][64] bf16), Cout % 16 == 0
void stem_mfma_pack(const float* w27, float* out, int Cout);
int launch_stem_mfma(const bf16_t* pix, const bf16_t* wp, const float* bias, bf16_t* y, int B, int S, int Cout, hipStream_t s);
// fused stem conv 3x3 s2 + GELU + depthwise 3x3 s2 + GELU: pix (B,S,S,4) -> y (B,S/4,S/4,96); C0 must be 96
bool stem_fused_supported(int S, int C0);
int launch_stem_fused(const bf16_t* pix, const bf16_t* wp, const float* b1, const float* w2, const float* b2, bf16_t* y, int B,
                      int S, int C0, hipStream_t s);
// the same stem sampling the SOURCE images (B,C,Hin,Win) f32 | u8 through the letterbox arithmetic: no (B,S,S,4) frame in HBM
int launch_stem_fused_lb(const void* img, int dtype, int B, int C, int Hin, int Win, float pad_value, int letterbox, const bf16_t* wp,
                         const float* b1, const float* w2, const float* b2, bf16_t* y, int S, int C0, hipStream_t s);
int launch_dwconv(const bf16_t* x, const float* w, const float* bias, bf16_t* y, int B, int H, int W, int C, int k,
                  int stride, int mult, int gelu, hipStream_t s);
// MFMA (4x4x4, 16 channel blocks) depthwise conv for stride-1 k in {3,7} on maps with W >= 32; ttab from dwconv_toeplitz_pack
bool dwconv_mfma_supported(int W, int C, int k, int stride, int mult);
size_t dwconv_toeplitz_elems(int C, int k);
void dwconv_toeplitz_pack(const float* w_tapmajor, float* out, int C, int k);
size_t dwconv_s2_toeplitz_elems(int C);
void dwconv_s2_toeplitz_pack(const float* w_tapmajor, float* out, int C);   // 7x7 stride 2, two outputs per input channel
bool dwconv_s2_mfma_supported(int H, int W, int C, int k, int stride, int mult);
int launch_dwconv_s2_mfma(const bf16_t* x, const bf16_t* ttab, const float* bias, bf16_t* y, int B, int H, int W, int C, int gelu,
                          hipStream_t s);
// x' = dw3x3(x), t = dw7x7(x') in one marching kernel (RepMixer token mixer + ConvFFN conv); tables as for launch_dwconv_mfma
bool dwconv_pair_supported(int B, int H, int W, int C);
int launch_dwconv_pair(const bf16_t* x, const bf16_t* t3, const float* b3, const bf16_t* t7, const float* b7, bf16_t* y1, bf16_t* y2,
                       int B, int H, int W, int C, hipStream_t s);
int launch_dwconv_mfma(const bf16_t* x, const bf16_t* ttab, const float* bias, bf16_t* y, int B, int H, int W, int C, int k,
                       int gelu, hipStream_t s);
int launch_layernorm_rows(const bf16_t* x, const float* w, const float* b, bf16_t* y, int rows, int C, float eps,
                          hipStream_t s);
int launch_se_gelu(const bf16_t* x, const float* w1, const float* b1, const float* w2, const float* b2, bf16_t* y,
                   float* scratch, int B, int P, int C, int R, hipStream_t s);
// lens (B) int32 or null; the visible key count of batch row b is clamp(lens[b] + len_add, 1, T)
int launch_attention(const bf16_t* q, const bf16_t* k, const bf16_t* v, int ldq, int ldk, int ldv, bf16_t* out,
                     int ldo, int B, int T, int heads, int kv_heads, int D, int causal, const int32_t* lens,
                     int len_add, float scale, hipStream_t s);

// decoder elementwise
int launch_embed_gather(const int32_t* ids, const bf16_t* table, const float* img_tokens, float* x, int B, int T,
                        int Ni, int H, int vocab, hipStream_t s);
// y_lo != null: also writes the bf16 remainder (x ~= y + y_lo), the split operand of the parity-mode decoder GEMMs
int launch_rmsnorm(const float* x, const float* w, bf16_t* y, bf16_t* y_lo, int ldy, int rows, int H, float eps, hipStream_t s,
                   int f16 = 0, unsigned* sat = nullptr,    // f16 != 0: y receives fp16 bits (y_lo must be null), clamps counted in *sat
                   int lo8 = 0);                            // lo8 != 0: the remainder leaves as fp8 (x 2^8), one byte per element, at y_lo's row starts
// in place: n bf16 values -> the fp16 values scale * x (weights of the fp16-operand projections, once at load time); maxbits (device,
// optional) receives max(|scale * x|) as float bits by atomicMax, so the loader can refuse weights outside the fp16 range
int launch_bf16_to_f16(bf16_t* p, size_t n, float scale, hipStream_t s, unsigned* maxbits = nullptr);
// W [rows][K] bf16 -> fp8 e4m3 copy of W x 2^6 at row stride 2K bytes (GemmArgs::W8); maxbits as launch_bf16_to_f16
int launch_bf16_to_w8(const bf16_t* w, void* w8, size_t rows, int K, hipStream_t s, unsigned* maxbits = nullptr);
int launch_rope_f32(float* qkv, const float2* table, int ld, int rows, int T, int heads, int kv_heads, int D, hipStream_t s);
// rope != null: qkv holds un-rotated projections; the rotate-half RoPE is fused into the MFMA kernel (head_dim 64 / 128) or applied
// in place by a rope_f32 launch ahead of the VALU kernel (head_dim 32)
int launch_attention_f32(float* qkv, int ld, bf16_t* out_hi, bf16_t* out_lo, int ldo, int B, int T, int heads,
                         int kv_heads, int D, const int32_t* lens, int len_add, float scale, hipStream_t s, const float2* rope = nullptr,
                         const float* pre = nullptr, int ldp = 0, int Np = 0,    // pre: cached [k | v] rows of positions < Np (8f-1)
                         float* lse = nullptr,    // optional [B][heads][T] row statistics max + log(sum) for the backward pass (head_dim 64 / 128, Np = 0)
                         int lo8 = 0,             // remainders as fp8 bytes (the hi + lo8 operand form of llm_precision = 5)
                         void* split_scratch = nullptr);   // attention_split_scratch_bytes() bytes -> the split-bf16 kernel (attention_split.hip) when lse is wanted (training)
                                                           // or the sequence is at least FV_ATTN_SPLIT_MIN_T long (the spliced prefill); never with a cached prefix or lo8
constexpr int FV_ATTN_SPLIT_MIN_T = 128;
// table: [>=T][D/2] (cos, sin) pairs built by rope_table_host(); position = row % T
int launch_rope(bf16_t* qkv, const float2* table, int ld, int rows, int T, int heads, int kv_heads, int D,
                hipStream_t s);
void rope_table_host(float* cos_sin_pairs, int T, int D, float theta);
int launch_pool_norm(const float* x, const int32_t* lens, const float* w, float* pooled, int B, int Ttot, int Ni,
                     int H, float eps, int mode, hipStream_t s);

// ---- unfrozen-backbone training glue (train_kernels.hip) ------------------------------------------------------
constexpr int RMS_BWD_RPW = 4;      // rows per wave of rmsnorm_bwd_kernel (one dw partial row per wave)
constexpr int COLSUM_CHUNKS = 64;   // row ranges of the two-stage deterministic column sum
int launch_split_rows(const float* in, int ldi, bf16_t* out, int ldo, int lo_off, long R, int C, hipStream_t s);
int launch_lo8_rows(const float* in, int ldi, bf16_t* out, int ldo, long R, int C, hipStream_t s);   // the fp8 remainder bytes of the hi + lo8 operand form
int launch_bf16_to_f32(const bf16_t* in, float* out, size_t n, hipStream_t s);
int launch_transpose_to_bf16(const void* in, int in_bf16, int ldi, bf16_t* out, int ldo, int lo_off, int R, int Rp, int C, hipStream_t s);
// the same transpose to fp16 (saturating; clamps counted in *sat): in_kind 0 = fp32, 1 = bf16, 2 = split bf16 (value = in[c] + in[lo_in + c]: both halves summed before rounding), 3 = fp16
int launch_transpose_to_f16(const void* in, int in_kind, int ldi, int lo_in, bf16_t* out, int ldo, int R, int Rp, int C, unsigned* sat, hipStream_t s,
                            bf16_t* rows_out = nullptr, int ldro = 0);   // rows_out: the same values also as fp16 rows [R][ldro]
int launch_rows_to_f16(const void* in, int in_kind, int ldi, int lo_in, bf16_t* out, int ldo, long R, int C, unsigned* sat, hipStream_t s,
                       long Rp = 0);   // Rp > R: rows [R, Rp) of the output written as zeros (the TN wgrad's K-tile padding)   // row-major fp16 rows (in_kind as above)
int launch_swiglu_bwd(float* gu, const float* dact, long rows, int I, hipStream_t s, bf16_t* out_split = nullptr);   // out_split: [rows][4I] = [hi | lo] bf16 instead of fp32 in place
// dgu straight into the two fp16 operands of its consumers: rows [rows][2I] and columns [2I][Rp] (zero for rows in [rows, Rp)); clamps counted in *sat
// d gate/up and act as fp16 ROWS only, rows [rows, Rp) zero: the operands of the TN wgrads (and d gate/up's rows of the dgrad)
int launch_swiglu_bwd_rows(const void* gu, int gu_f16, const float* dact, long rows, long Rp, int I, bf16_t* dgu_rows, bf16_t* act_rows, unsigned* sat, hipStream_t s);
int launch_swiglu_bwd_f16(const void* gu, int gu_f16, const float* dact, int rows, int Rp, int I, bf16_t* out_rows, bf16_t* outT, unsigned* sat, hipStream_t s,
                          bf16_t* actT = nullptr);   // gu_f16: the kept accumulators are fp16 (GemmArgs::stash_f16)   // actT: also silu(gate) * up as fp16 columns [I][Rp] (the down projection's wgrad operand)
int launch_gelu_fwd(const float* pre, bf16_t* out, int ldo, int lo_off, long R, int C, hipStream_t s);
int launch_gelu_bwd(float* dh, const float* pre, size_t n, hipStream_t s);
size_t rmsnorm_bwd_scratch_floats(long rows, int H);
int launch_colsum(const float* in, int ld, long R, int C, float* out, float* scratch, hipStream_t s);   // scratch >= COLSUM_CHUNKS * C floats
int launch_rmsnorm_bwd(const float* x, const float* w, const float* dy, const float* dres, float* dx, float* dw, float* scratch, long rows, int H,
                       float eps, hipStream_t s);
int launch_pool_rows(float* stream, float* compact, const int32_t* lens, int B, int Tt, int Ni, int H, int scatter, hipStream_t s);
int launch_image_rows(const float* stream, float* compact, int B, int Tt, int Ni, int H, hipStream_t s);
int launch_embed_bwd(const int32_t* ids, const int32_t* lens, const float* dx, float* dE, int B, int T, int Ni, int H, int vocab, hipStream_t s);
int launch_attention_bwd(const float* qkv, int ld, const bf16_t* o_hi, const bf16_t* o_lo, int ldo, const float* dO, int lddo, const float* lse,
                         float* delta, float* dqkv, int B, int T, int heads, int kv_heads, int D, const int32_t* lens, int len_add, float scale,
                         const float2* rope, hipStream_t s, float* part = nullptr,    // part: optional scratch of (heads / kv_heads) * B * T * 2 * kv_heads * D floats
                         void* split_scratch = nullptr);   // attention_split_scratch_bytes() bytes -> the split-bf16 kernels (attention_split.hip)
                                                                                       // -> dK / dV per q head in parallel, then summed in a fixed order

// fv_train_commit in ONE launch: every trainable tensor of the flat fp32 master -> the library's operand copies.  One descriptor per tensor
// (device array, sorted by tile0); a block takes one 64 x 64 tile of a matrix (bf16 rows + the transposed dgrad copy, fp16 or bf16) or 4096
// floats of a vector
struct CommitDesc {
  long long src_off;     // offset into the flat master (floats)
  void* dst;             // bf16 [rows][cols] (matrix) or fp32 [numel] (vector)
  void* dstT16;          // fp16 [cols][rows] or null
  void* dstTb;           // bf16 [cols][rows] or null
  int rows, cols;        // vector: rows = 1, cols = numel
  int is_mat;
  int tile0;             // first tile of this tensor in the launch's tile numbering
  void* dst16;           // fp16 [rows][cols] copy x scale16 or null (the one-pass fp16 training forward's weight operand)
  float scale16;
};
int launch_commit(const CommitDesc* desc_dev, int ndesc, int ntiles, const float* flat, int f16_transposes, unsigned* sat, hipStream_t s);

// ---- tower backward (tower_bwd_kernels.hip; SURVEY.md 8f-4, second slice) ---------------------------------------------------------
// gradients NHWC fp16 (loss-scaled), activations NHWC bf16, weight gradients fp32 by fixed-order partial sums (no atomics)
size_t dw_bwd_scratch_floats(int B, int Ho, int Wo, int Co, int k);
int launch_dw_dgrad(const bf16_t* dy, const float* w, const bf16_t* res, bf16_t* dx, int B, int Hi, int Wi, int Ci, int k, int stride, int mult, unsigned* sat,
                    hipStream_t s, int round_w = 0);   // round_w: taps rounded to bf16 first (the forward ran on a bf16 Toeplitz table)
// the same input gradient on the marching MFMA depthwise kernel (stride 1, W >= 16, H >= 16, C % 32 == 0): ttab16 = fp16 Toeplitz table of the flipped taps
bool dw_dgrad_mfma_supported(int H, int W, int C, int k);
int launch_dw_dgrad_mfma(const bf16_t* dy, const bf16_t* ttab16, const bf16_t* res, bf16_t* dx, int B, int H, int W, int C, int k, hipStream_t s);
int launch_dw_wgrad(const bf16_t* x, const bf16_t* dy, float* dw, float* db, float* scratch, int B, int Hi, int Wi, int Ci, int k, int stride, int mult,
                    hipStream_t s);
constexpr int TOWER_COLSUM_CHUNKS = 512;   // row ranges of the tower backward's column sums (10^5 .. 10^6 rows of 96 .. 1536 columns: the grid must fill the chip)
int launch_colsum16(const bf16_t* in, int ld, long R, int C, float* out, float* scratch, hipStream_t s);   // scratch >= TOWER_COLSUM_CHUNKS * C floats
int launch_mul16(const bf16_t* a, const bf16_t* b, bf16_t* out, size_t n, unsigned* sat, hipStream_t s);
int launch_gelu_grad_mul(const bf16_t* dh, const bf16_t* pre, bf16_t* out, size_t n, unsigned* sat, hipStream_t s);   // out f16 = dh f16 * gelu'(pre bf16)
int launch_f16_to_f32(const bf16_t* in, float* out, size_t n, float scale, hipStream_t s);
int launch_scale_to_f16(const float* in, bf16_t* out, size_t n, float scale, unsigned* sat, hipStream_t s);   // out f16 = in * scale (saturating)
int launch_rescale_to_f16(const float* in, bf16_t* out, size_t n, float target, unsigned* bits, float* sc, unsigned* sat, hipStream_t s);   // the tower's own gradient scale (device-chosen power of two)
int launch_unscale_dev(float* p, size_t n, const float* sc, hipStream_t s);
int launch_ones_col(bf16_t* out, int ld, int C, long R, hipStream_t s);   // fp16 rows [R][ld]: columns [C, C + 8) <- {1, 0, ..., 0}
int launch_ls_grads(const float* dWraw, const float* dbraw, const bf16_t* W, const float* bias, const float* ls, float* dW, float* db, float* dls, int C, int Kd,
                    hipStream_t s);
size_t ln_bwd_scratch_floats(long rows, int C);
int launch_ln_bwd(const bf16_t* x, const bf16_t* dy, const float* w, const bf16_t* res, bf16_t* dx, float* dw, float* db, float* scratch, long rows, int C,
                  float eps, unsigned* sat, hipStream_t s);
int launch_tower_attn_bwd(const bf16_t* qkv, int ld, const bf16_t* dO, int lddo, bf16_t* dqkv, int ldd, float* stats, int B, int T, int heads, float scale,
                          hipStream_t s);   // stats: 2 * B * heads * T floats
int launch_se_bwd(const bf16_t* e, const bf16_t* dout, const float* se, const float* w1, const float* w2, bf16_t* de, float* dW1, float* db1, float* dW2,
                  float* db2, float* tmp, int B, int P, int C, int R, unsigned* sat, hipStream_t s);   // tmp >= B * (2 C + 2 R) floats
size_t stem0_wgrad_scratch_floats(int B, int S, int C0);
int launch_stem0_wgrad(const bf16_t* pix, const float* w, const float* bias, const bf16_t* dh0, float* dw, float* db, float* scratch, int B, int S, int C0,
                       hipStream_t s, int round_w = 0);
// fp16 patch rows of the stem's dense 3x3 s2 conv: P [B (S/2)^2][32], column (ky*3+kx)*3+ci, column 27 = 1 (the bias tap), 28 .. 31 = 0
int launch_stem_im2col16(const bf16_t* pix, bf16_t* P, int B, int S, hipStream_t s);
// fv_train_commit's tower part: one launch over a table of operations (device array, sorted by blk0; a block = 1024 destination elements)
//   kind 0: dst f32[i] = src[i]      1: dst bf16[i] = src[i]      2: dst bf16[i] = idx[i] >= 0 ? coef[i] * src[idx[i]] : 0   (the packed operand images)
//   kind 3: dst f16 [cols][rows] = src[rows][cols]^T              4: the same with row r scaled by flat[src2_off + r]         (dgrad operands)
//   kind 5: as 2 with the value rounded to bf16 first and stored as fp16 (the flipped-tap Toeplitz tables of the depthwise input gradients)
struct TowerCommitOp {
  long long src_off, src2_off, n;
  void* dst;
  const int* idx; const float* coef;
  int kind, rows, cols, blk0;
};
int launch_tower_commit(const TowerCommitOp* ops_dev, int nops, int nblocks, const float* flat, unsigned* sat, hipStream_t s);

// attention_split.hip: the same forward (training: with lse) and backward on the bf16 matrix core with split (hi + lo) operands, three passes per product
// scratch: attention_split_scratch_bytes(B, T, kv_heads, D) bytes -- K (rotated) and V of every kv head as split bf16 in both operand forms, written once
// per call by a small pre-pass and copied flat into LDS by every q head's / query block's thread block
size_t attention_split_scratch_bytes(int B, int T, int kv_heads, int D);
int launch_attention_split_fwd(const float* qkv, int ld, bf16_t* out_hi, bf16_t* out_lo, int ldo, int B, int T, int heads, int kv_heads, int D,
                               const int32_t* lens, int len_add, float scale, const float2* rope, float* lse, void* scratch, hipStream_t s);
int launch_attention_split_bwd(const float* qkv, int ld, const bf16_t* o_hi, const bf16_t* o_lo, int ldo, const float* dO, int lddo, const float* lse,
                               float* delta, float* dqkv, int B, int T, int heads, int kv_heads, int D, const int32_t* lens, int len_add, float scale,
                               const float2* rope, hipStream_t s, float* part, long pstride, void* scratch);

// action expert (all fp32)
struct HeadDims { int feat, ds, da, hid, fus; };
struct HeadOffsets { int64_t o[13]; };
HeadOffsets head_offsets(const HeadDims& d);
size_t head_saved_bytes(const HeadDims& d, int B);
// dataset normalisation folded into the head (device vectors; LeRobot MEAN_STD): states <- (states - state_mean) * state_istd
// ahead of the first LayerNorm; actions <- actions * action_std + action_mean behind the last Linear when not training
struct HeadIoNorm { const float *state_mean, *state_istd, *action_mean, *action_std; };
int launch_head_forward(const HeadDims& d, const float* P, const float* pooled, const float* states, int B,
                        int training, float drop_p, uint64_t seed, uint64_t offset, float* actions, float* saved,
                        hipStream_t s, const HeadIoNorm* io = nullptr);
// grad_actions != null: generic backward from dL/dactions (loss untouched); else fused MSE(actions, targets) + backward
// d_pooled (optional, [B][feat]): dL/d(pooled feature) -- the gradient an UNFROZEN backbone continues from
int launch_head_backward(const HeadDims& d, const float* P, const float* grad_actions, const float* actions,
                         const float* targets, int B, float drop_p, const float* saved, float* loss, float* G,
                         float* scratch, hipStream_t s, float* d_pooled = nullptr, float loss_scale = 1.0f);   // loss_scale: the fused MSE's dL/dactions times a power of two
size_t head_bwd_scratch_bytes(const HeadDims& d, int B);
size_t adamw_scratch_bytes();  // norm_scratch of launch_adamw_clip: [0] = sum of squares, [1..] per-block partials
int launch_axpy(float* y, const float* x, int64_t n, const float* scale_dev, hipStream_t s);  // y += x, or y *= *scale_dev when x is null
int launch_adamw_clip(float* p, const float* g, float* m, float* v, int64_t n, const fv_adamw_hparams& hp,
                      int64_t step, float* norm_scratch, float* grad_norm_out, hipStream_t s);

}  // namespace fv
