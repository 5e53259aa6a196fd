/* fastvla_hip_testops.h -- TEST-ONLY op-level entry points (one kernel each) over the internals of libfastvla_hip.so.
 *
 * Built by `make -C vla-from-fastvlm_amd/csrc` into tests/_native/libfastvla_hip_testops.so (csrc/ops_api.hip); its undefined `fv::launch_*`
 * references resolve against an already loaded libfastvla_hip.so (fastvla_hip._lib.load_testops() loads the product library RTLD_GLOBAL first).
 * Nothing of the product (fastvla_hip/, vla_fastvlm/, bench.py's timed region, __graft_entry__) binds these symbols: they exist so that
 * tests/test_gpu_ops.py and tools/ can check and time each kernel on its own against the oracle's op.  The product ABI a maintainer binds is
 * include/fastvla_hip.h alone.  (The epilogue enum lives here because only these entry points take it as an argument; the library's own
 * sources include this header for it.) */
#ifndef FASTVLA_HIP_TESTOPS_H
#define FASTVLA_HIP_TESTOPS_H

#include "fastvla_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- op-level entry points (used by the parity tests to check each kernel on its own) ------------------------- */
enum fv_gemm_epilogue {
  FV_EPI_BIAS = 0,        /* out bf16 = acc + bias                                  */
  FV_EPI_BIAS_GELU = 1,   /* out bf16 = gelu(acc + bias)                            */
  FV_EPI_LS_RES = 2,      /* out bf16 = res_bf16 + scale[n] * (acc + bias[n])       */
  FV_EPI_RES_F32 = 3,     /* out f32  = res_f32 + acc (+ bias)                      */
  FV_EPI_SWIGLU = 4,      /* out bf16[M,N/2] = silu(gate) * up, W rows 8-interleaved */
  FV_EPI_F32 = 5,         /* out f32  = acc + bias                                  */
  FV_EPI_SWIGLU_SPLIT = 7,/* out bf16[M,N] = [hi | lo] of silu(gate)*up (split-bf16 operand), W rows 8-interleaved */
  FV_EPI_SWIGLU_F16 = 8,  /* out f16[M,N/2] = silu(gate)*up / 16 (the fp16 operand of the down projection, whose fp16 weights carry the
                           * 16: the power of two keeps SwiGLU outputs up to 1e6 inside fp16's range), W rows 8-interleaved */
  /* the tower backward's three (all 2-byte outputs are fp16, saturating, whatever the operand type) */
  FV_EPI_GELU_GRAD = 9,   /* a = acc + bias: out f16 = gelu(a) (the 5e-5 minimax Phi of the forward kernels, rounded to a bf16 value first; out may be NULL), stash f16 = gelu'(a), both [M][ldo] */
  FV_EPI_MUL_AUX = 10,    /* out f16 = acc * aux_f16[m][n] (aux = res, row stride ldr; out may alias aux)                              */
  FV_EPI_F16 = 11,        /* out f16 = acc + bias                                                                                      */
  FV_EPI_MUL_GELUP = 12   /* out f16 = acc * gelu'(4 * aux_f16[m][n]); stash f16 (optional) = gelu(4 * aux) rounded to bf16: aux = the pre-activation / 4 the training forward's fused ConvFFN stashed (round 6) */
};
/* out[M,N] = A[M,K] (bf16, row stride lda) x W[N,K]^T (bf16) with fp32 accumulation on MFMA */
int fv_op_gemm(const void* A, int lda, const void* W, int M, int N, int K, const float* bias, const float* scale,
               const void* res, int ldr, void* out, int ldo, int epilogue, fv_stream s);
/* the split-bf16 form the parity-mode decoder uses: A (M, 2K) carries [hi | lo] halves side by side (lda >= 2K), and
 * out = (A_hi + A_lo) . W^T in one launch with a doubled K loop; epilogues as fv_op_gemm */
/* fp16 operands (A and W hold IEEE binary16 bits), fp32 accumulation on v_mfma_f32_16x16x32_f16; epilogues as fv_op_gemm plus
 * FV_EPI_SWIGLU_F16 (the path of llm_precision = 2's gate/up and down projections) */
int fv_op_gemm_f16(const void* A, int lda, const void* W, int M, int N, int K, const float* bias, const void* res, int ldr, void* out,
                   int ldo, int epilogue, void* ws, size_t ws_bytes, fv_stream s);
/* the "hi + lo8" form of llm_precision = 5: A rows = K bf16 values, then at byte offset 2K their K remainders as fp8 e4m3 (x 2^8); W8 = fp8 copy
 * of W x 2^6 at row stride 2K bytes.  out = (A_hi + A_lo8 2^-8) . W^T with the lo product on v_mfma_scale_f32_16x16x128_f8f6f4 against W8: 1.5
 * passes.  K % 128 == 0, lda >= 1.5 K.  fv_op_lo8_pack builds both operand forms from fp32 rows / a bf16 weight (either may be NULL). */
int fv_op_gemm_lo8(const void* A, int lda, const void* W, const void* W8, int M, int N, int K, const float* bias, const void* res, int ldr, void* out,
                   int ldo, int epilogue, void* ws, size_t ws_bytes, fv_stream s);
int fv_op_lo8_pack(const float* x, void* a_out, int lda, const void* W, void* w8_out, int M, int K, int N, fv_stream s);
int fv_op_gemm_ksplit(const void* A, int lda, const void* W, int M, int N, int K, const float* bias, const void* res, int ldr,
                      void* out, int ldo, int epilogue, fv_stream s);
/* fv_op_gemm_ksplit (ksplit != 0) or fv_op_gemm with a caller-owned scratch buffer for split-K partial sums: fp32 epilogues
 * (FV_EPI_RES_F32 / FV_EPI_F32) of problems with few output tiles and a long K are cut along K, one unit per CU, and
 * summed by a second kernel.  ws may be NULL (then exactly fv_op_gemm / fv_op_gemm_ksplit). */
/* out f32[M,N] = sum_k A[k][m] W[k][n] (+ bias[n]): both operands row-major over the CONTRACTION index (A [K][M] row stride lda, W [K][N] row stride
 * ldw; bf16, or fp16 bits with f16 != 0), K % 64 == 0, M % 8 == 0, N % 8 == 0 -- the weight-gradient shape (fv_train_set_options wgrad_f16 = 2) */
int fv_op_gemm_tn(const void* A, int lda, const void* W, int ldw, int M, int N, int K, int f16, const float* bias, void* out, int ldo, void* ws,
                  size_t ws_bytes, fv_stream s);
int fv_op_gemm_splitk(const void* A, int lda, const void* W, int M, int N, int K, const float* bias, const void* res, int ldr,
                      void* out, int ldo, int epilogue, int ksplit, void* ws, size_t ws_bytes, fv_stream s);
/* depthwise / channel-multiplier grouped conv, NHWC bf16: x (B,H,W,C) -> y (B,Ho,Wo,C*mult); w f32 [k*k][C*mult] */
int fv_op_dwconv(const void* x, const float* w, const float* bias, void* y, int B, int H, int W, int C, int k,
                 int stride, int mult, int gelu, fv_stream s);
/* dense 3x3 stride-2 stem conv on (B,S,S,4) bf16 -> (B,S/2,S/2,Cout) bf16, + bias + GELU; w f32 [27][Cout] */
int fv_op_stem_conv(const void* pix, const float* w, const float* bias, void* y, int B, int S, int Cout, fv_stream s);
/* the same stem as an implicit GEMM on MFMA: wp (Cout,64) bf16 with slot 32*ks + 8*g + e = weight of kernel row
 * ky = 2*ks + (g>>1), column kx = 2*(g&1) + (e>>2), channel e&3 (0 where ky, kx or channel is 3); Cout % 16 == 0 */
int fv_op_stem_mfma(const void* pix, const void* wp, const float* bias, void* y, int B, int S, int Cout, fv_stream s);
/* the first two stem convolutions fused (conv 3x3 s2 3->96 + GELU, depthwise 3x3 s2 + GELU): pix (B,S,S,4) bf16 ->
 * y (B,S/4,S/4,96) bf16; wp as for fv_op_stem_mfma, w2 f32 [9][96] tap-major, the half-resolution map stays in LDS
 * (rounded to bf16 there exactly as the unfused pair rounds it to memory).  Cout must be 96, S % 4 == 0. */
int fv_op_stem_fused(const void* pix, const void* wp, const float* b1, const float* w2, const float* b2, void* y, int B,
                     int S, int Cout, fv_stream s);
/* fv_op_stem_fused reading the SOURCE images (B,C,Hin,Win) f32 | u8 through the letterbox arithmetic of fv_preprocess (the kernel behind
 * fv_vision_forward_images): y must equal fv_op_stem_fused on fv_preprocess's output bit for bit */
int fv_op_stem_fused_images(const void* img, int dtype, int B, int C, int Hin, int Win, float pad_value, int resize_with_padding, const void* wp,
                            const float* b1, const float* w2, const float* b2, void* y, int S, int Cout, fv_stream s);
/* per-pixel LayerNorm over channels (LayerNormChannel), x,y (rows,C) bf16 */
int fv_op_layernorm_rows(const void* x, const float* w, const float* b, void* y, int rows, int C, float eps,
                         fv_stream s);
/* multi-head attention on packed projections: q/k/v element pointers with row strides, heads of width D.
 * causal != 0: key j visible to query i iff j <= i; lens (B) int32 or NULL masks keys >= len. */
int fv_op_attention(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, void* out, int ldo, int B,
                    int T, int heads, int kv_heads, int D, int causal, const int32_t* lens, float scale, fv_stream s);
int fv_op_rmsnorm(const float* x, const float* w, void* y_bf16, int rows, int H, float eps, fv_stream s);
/* in-place rotate-half RoPE on the q and k parts of a packed qkv (rows = B*T, position = row % T) */
int fv_op_rope(void* qkv, int ld, int rows, int T, int heads, int kv_heads, int D, float theta, fv_stream s);
/* backward of fv_op_attention's causal GQA form in fp32 (the kernels behind fv_train_forward_backward): qkv fp32 [B*T][ld] = UN-rotated
 * q | k | v (the rotate-half RoPE of `theta` is applied inside, as in the parity-mode forward), dO fp32 [B*T][heads*D] -> dqkv fp32 [B*T][ld]
 * (gradients w.r.t. the un-rotated projections).  Runs the forward first (its output and row statistics are scratch).  D in {64, 128}. */
int fv_op_attention_bwd(const float* qkv, int ld, const float* dO, float* dqkv, void* out_bf16_scratch, float* stat_scratch, int B, int T,
                        int heads, int kv_heads, int D, const int32_t* lens, float theta, fv_stream s);
/* backward of fv_op_rmsnorm: y = w x rsqrt(mean(x^2) + eps); dx (rows,H) f32 = dres (or 0) + dL/dx, dw (H) f32; scratch floats:
 * ((rows + 15) / 16 + 3) / 4 * 4 * H + 64 * H */
int fv_op_rmsnorm_bwd(const float* x, const float* w, const float* dy, const float* dres, float* dx, float* dw, float* scratch, int rows,
                      int H, float eps, fv_stream s);

/* SE + GELU tail of conv_exp: x (B,P,C) bf16 -> y = gelu(x * sigmoid(W2 relu(W1 mean_p(x) + b1) + b2)) */
int fv_op_se_gelu(const void* x, const float* w1, const float* b1, const float* w2, const float* b2, void* y,
                  float* scratch, int B, int P, int C, int R, fv_stream s);

/* stride-1 depthwise k x k (k in {3,7}) on the matrix cores (v_mfma_f32_4x4x4_16b_bf16, one 4x4 block per channel);
 * W >= 32, C % 32 == 0.  ttab = bf16 Toeplitz table [C/16][k][NM=(k+6)/4][16 ch][4 i][4 kk] = w[ky][4m + kk - i] (0 outside
 * the kernel), i.e. the depthwise weights rounded to bf16. */
int fv_op_dwconv_mfma(const void* x, const void* ttab, const float* bias, void* y, int B, int H, int W, int C, int k, int gelu,
                      fv_stream s);
/* the PatchEmbed large-kernel conv on the same scheme: 7x7, stride 2, two output channels per input channel (groups = C):
 * x (B,H,W,C) -> y (B,H/2,W/2,2C) bf16 NHWC; H, W even, W >= 16, C % 32 == 0.  ttab = bf16 Toeplitz table
 * [C/16][e=2][7][4][16 ch][4 i][4 kk] = w[ky][4m + kk - 2i] of output channel 2 (16 g + ch) + e (0 outside the 7 taps).
 * Replaces fv_op_dwconv(k=7, stride=2, mult=2) for these shapes (mci.py PatchEmbed, lkb_reparam). */
int fv_op_dwconv_s2_mfma(const void* x, const void* ttab, const float* bias, void* y, int B, int H, int W, int C, int gelu,
                         fv_stream s);
/* RepMixer pair in one marching kernel: y1 = dw3x3(x) + b3 (the reparameterised token mixer), y2 = dw7x7(y1) + b7 (the
 * ConvFFN's conv); x, y1, y2 (B,H,W,C) bf16 NHWC, distinct; t3 / t7 Toeplitz tables as for fv_op_dwconv_mfma (k = 3 / 7);
 * H >= 16, W >= 32, C % 32 == 0. */
int fv_op_dwconv_pair(const void* x, const void* t3, const float* b3, const void* t7, const float* b7, void* y1, void* y2, int B,
                      int H, int W, int C, fv_stream s);
/* fused ConvFFN pointwise half: out (M,C) bf16 = res + ls * (fc2(gelu(fc1(x) + b1)) + b2), hidden = 4C never leaves the
 * chip.  w1 (4C,C) bf16; w2p = fc2 weight (C,4C) re-laid as [4C/32][C][32] with slot 8g+j of each 32-block holding
 * hidden 16*(j>>2) + 4*g + (j&3).  C in {32,64,96,128,192,384}.  out may alias res, not x. */
int fv_op_convffn(const void* x, const void* w1, const float* b1, const void* w2p, const float* b2, const float* ls,
                  const void* res, void* out, int M, int C, fv_stream s);

/* the same fused ConvFFN on v_mfma_f32_32x32x16_bf16 (the kernel the engine uses for C in {96,192,384}).  wq = fc1 (4C,C) / 4 and
 * 4 * fc2 (C,4C) (powers of two: exact; the kernel's GELU runs in y = x / 4 and takes b1 / 4 itself from the plain b1 it is given) as
 * ONE bf16 stream [4C/32 chunks][64 C]: per 32-hidden chunk the kernel's LDS slot image in staging order.
 * Slot image T: W1 part = 32 rows x C, 8-element chunk c of row r at chunk c ^ ((r >> SH) & MASK) with (SH, MASK) = (0,15) / (1,7) /
 * (2,3) for C = 384 / 192 / 96; W2 part = C rows x 32, chunk (2s + h) ^ ((n >> 2) & 3) of row n holding hidden
 * 16s + 8(j>>2) + 4h + (j&3), j < 8 (the k order in which a 32x32 accumulator tile, converted pairwise to bf16, is the B operand
 * of the next product).  Every KB of T is stored transposed for the add-tid LDS stores: G[8l + 2d + b] = T[128d + 2l + b],
 * l < 64, d < 4, b < 2. */
int fv_op_convffn32(const void* x, const void* wq, const float* b1, const float* b2, const float* ls, const void* res, void* out,
                    int M, int C, fv_stream s);
/* The same operator when a launch has few row tiles (one to four observations: <= 128 tiles for 256 CUs): blocks = (row tile, one of 2 / 4 / 8 ranges
 * of the hidden units), fp32 partial sums in `part` (>= ranges x M x C floats, 16-byte aligned), then one pass adds the ranges in order and applies
 * b2, ls and the residual exactly as the one-launch epilogue does.  Falls back to fv_op_convffn32 when M is large or `part` too small.  The engine
 * takes this form by itself below 128 row tiles (fv_vision_forward at B <= 4). */
/* the TRAINING forward's form of fv_op_convffn32 (round 6): the same output bits, and the pre-activation on the way out -- stash_y (M,4C) fp16 = (fc1(x) + b1) / 4
 * (round towards zero), in natural column order.  M * 8C < 2 GiB. */
int fv_op_convffn32_stash(const void* x, const void* wq, const float* b1, const float* b2, const float* ls, const void* res, void* out,
                          int M, int C, void* stash_y, fv_stream s);
/* the fc2 input gradient of the tower's backward as it runs there: fp16 operands, out f16 = (A . W^T) * gelu'(4 aux), h_out f16 (may be NULL) = gelu(4 aux) rounded to
 * a bf16 value (the fc2 weight gradient's operand), aux = fv_op_convffn32_stash's stash_y */
int fv_op_gemm_f16_gelup(const void* A, int lda, const void* W, int M, int N, int K, const void* aux, int ldaux, void* out, int ldo, void* h_out, fv_stream s);
/* tap + bias gradients of a depthwise / channel-multiplier conv as the tower's backward computes them: x bf16 (B,H,W,C), dy fp16 bits (B,Ho,Wo,C*mult) ->
 * dw f32 tap-major [k*k][C*mult], db f32 [C*mult]; scratch_floats >= (B * ceil(W / 32) or 512) * (k*k + 1) * C*mult (the call checks).  Stride-1 maps with C % 32 == 0,
 * W >= 32, H >= 16 and k = 7 run on the matrix cores (dw_wgrad_mfma_kernel, round 6), everything else on the VALU forms. */
int fv_op_dw_wgrad(const void* x, const void* dy, float* dw, float* db, float* scratch, size_t scratch_floats, int B, int H, int W, int C, int k, int stride, int mult, fv_stream s);
int fv_op_convffn32_split(const void* x, const void* wq, const float* b1, const float* b2, const float* ls, const void* res, void* out,
                          int M, int C, float* part, size_t part_bytes, fv_stream s);

#ifdef __cplusplus
}
#endif
#endif /* FASTVLA_HIP_TESTOPS_H */
