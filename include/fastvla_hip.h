/*
 * fastvla_hip.h -- C ABI of libfastvla_hip.so: the MI355X-native (gfx950) FastVLA policy-step hot path.
 *
 * The reference (syun88/VLA-from-FastVLM) has NO FFI: its boundary is Python-level.  Each entry point below cites the
 * reference function whose arithmetic it replaces (paths relative to the reference root).  The only caller is the
 * host-side mirror in vla-from-fastvlm_amd/ (ctypes); tensors cross the boundary as raw device pointers + sizes.
 *
 * Conventions
 *   - return 0 on success, negative fv_status on error; message via fv_last_error(); never throws, never exits.
 *   - a handle is bound to ONE device; one handle per rank/process; calls are asynchronous on the supplied stream;
 *     no hidden hipDeviceSynchronize, no allocation inside forward/backward calls (graph-capturable).
 *   - caller owns inputs, outputs, trainable parameters, optimizer state and the workspace; the library owns only
 *     the packed frozen weights.
 *   - activations: NHWC bf16 in the tower, [tokens][hidden] f32 residual stream in the decoder.
 */
#ifndef FASTVLA_HIP_H
#define FASTVLA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fv_handle fv_handle;
typedef void* fv_stream; /* hipStream_t */

enum fv_status {
  FV_OK = 0,
  FV_ERR_ARG = -1,       /* bad argument / shape (reference: ValueError, fastvlm_adapter.py:41-42,150-154) */
  FV_ERR_STATE = -2,     /* weights/workspace missing (reference: RuntimeError, fastvlm_adapter.py:366-367,556) */
  FV_ERR_HIP = -3,       /* HIP runtime error */
  FV_ERR_MISSING = -4,   /* a required weight tensor was not supplied */
  FV_ERR_UNSUPPORTED = -5
};

enum fv_dtype { FV_F32 = 0, FV_BF16 = 1, FV_U8 = 2, FV_I32 = 3 };

#define FV_MAX_STAGES 8

typedef struct fv_model_desc {
  /* Qwen2 decoder ([site] transformers/models/qwen2/modeling_qwen2.py) */
  int32_t llm_hidden, llm_layers, llm_heads, llm_kv_heads, llm_head_dim, llm_inter, llm_vocab;
  float rope_theta, rms_eps;
  /* FastViT-HD tower ([UNVENDORED] mci.py `fastvithd`) */
  int32_t tower_stages;
  int32_t tower_layers[FV_MAX_STAGES];
  int32_t tower_dims[FV_MAX_STAGES];
  int32_t tower_is_attn[FV_MAX_STAGES];
  int32_t tower_mlp_ratio, tower_head_dim, tower_se_rd; /* se_rd = int(out_dim * 0.0625) */
  int32_t tower_out_dim;                                 /* 2 * dims[last] */
  float ln_eps, bn_eps;
  int32_t image_size;                                    /* S: square side the tower runs at (expected_size) */
  /* action expert (fastvla/fastvlm_with_expert.py:23-38) */
  int32_t state_dim, action_dim, hidden_dim, fusion_dim;
  /* capacity */
  int32_t max_batch, max_text_tokens;
  int32_t tower_microbatch; /* images per tower pass (0 = whole batch) */
  /* decoder arithmetic: 0 = bf16 MFMA operands (fastest; actions ~8e-3 from the fp32 reference at 0.5B),
   * 1 = split-bf16 activations (hi + lo, two MFMA passes, exact bf16 weights) + fp32 attention: 16 significant bits on
   * every GEMM operand, actions within 1e-3 of the fp32 reference (the parity bar of north_star).  Tower unaffected.
   * 2 = the per-GEMM budget of tests/precision_budget.py: qkv and o keep split-bf16 operands (12 % of the decoder's MACs), gate/up
   * and down run ONE pass on fp16 operands (11 significant bits; fp16 copies of those weights are exact for |w| >= 6e-5 and the
   * down projection carries a 2^4 scale against a 2^-4 on its operand), fp32 attention: actions ~5e-4 from the fp32 reference at
   * 0.56x the MFMA work of mode 1.  3 / 4 = the two halves of mode 2 on their own (3: the fp16 pass on gate/up only, its SwiGLU output
   * leaves as hi + lo bf16 for a split-bf16 down projection; 4: on down only): the per-family rows of the precision budget measured on
   * the product (tools/prec_sweep.py), not defaults of any preset.
   * 5 = "hi + lo8" (round 4): every split operand keeps its bf16 hi half and carries the remainder as ONE fp8 e4m3 byte (x 2^8); the lo product
   * runs on v_mfma_scale_f32_16x16x128_f8f6f4 against fp8 copies of the weights (x 2^6) at twice the bf16 MFMA rate: 1.5 passes instead of 2,
   * 13 significant bits per operand instead of 16 (5.8e-5 per GEMM; split-bf16 2.5e-6, one fp16 pass 2.1e-4); fp32 attention.  Needs hidden,
   * inter and heads * head_dim % 128 == 0; weights with |w| x 64 > 448 are refused at load time. */
  int32_t llm_precision;
} fv_model_desc;

typedef struct fv_tensor_desc {
  const char* name;   /* canonical checkpoint key, e.g. "model.layers.0.self_attn.q_proj.weight" */
  const void* data;   /* contiguous; HOST pointer unless device != 0 */
  int32_t dtype;      /* FV_F32 or FV_BF16 */
  int32_t ndim;
  int64_t shape[4];
  int32_t device;     /* != 0: data is a pointer into the handle's device memory space (copied device-to-device) */
  int32_t reserved;
} fv_tensor_desc;

/* fv_load_weights_cb: called once per tensor the packer needs, in packing order; fills *out (name may stay NULL) and
 * returns 0, or returns non-zero when the checkpoint has no such key.  out->data must stay valid until the next call of
 * the provider (or the end of fv_load_weights_cb): the host side can materialise one tensor at a time (a 7B checkpoint
 * never exists as one 30 GB fp32 dict). */
typedef int (*fv_tensor_provider)(void* user, const char* name, fv_tensor_desc* out);

/* opaque RCCL unique id (ncclUniqueId): made by rank 0 (fv_comm_unique_id), carried to the other ranks by the host side */
typedef struct fv_rccl_id { char internal[128]; } fv_rccl_id;

typedef struct fv_adamw_hparams {
  float lr, beta1, beta2, eps, weight_decay;
  float max_grad_norm; /* <= 0 : no clipping */
  float grad_scale;    /* multiplied into grads before use (1/world_size after a sum all-reduce) */
} fv_adamw_hparams;

/* ---- lifecycle -------------------------------------------------------------------------------------------- */
/* replaces FastVLMBackbone.__init__ / _load_model (model/fastvlm_adapter.py:90-201): create the engine ...
 * The handle owns its weights, the RoPE table, a few KB of counters and ~100 MB of scratch for the few-row launch forms of the tower (fused-ConvFFN
 * hidden ranges, K ranges of the last stages' GEMMs: one to four observations); everything batch-sized is the caller's (fv_workspace_bytes). */
int fv_create(const fv_model_desc* desc, int device, fv_handle** out);
/* ... and pack/fold the frozen weights (BN folding, gate/up interleave, qkv concat, bf16) into library memory. */
int fv_load_weights(fv_handle* h, const fv_tensor_desc* tensors, int n);
/* the same packing with the tensors pulled one at a time through `provider` (streaming load; bf16 sources, on the host
 * or on the device, reach the packed layouts without an fp32 round trip). */
int fv_load_weights_cb(fv_handle* h, fv_tensor_provider provider, void* user);
void fv_destroy(fv_handle* h);
/* last error of a call made on h (per handle); h == NULL: last error raised on the calling thread by fv_create or by a
 * handle-less fv_op_* entry point */
const char* fv_last_error(fv_handle* h);
const char* fv_version(void);

/* bytes of caller-owned scratch needed for batch B, T text tokens (+ image tokens when splice != 0) */
int fv_workspace_bytes(fv_handle* h, int B, int T, int splice, size_t* out_bytes);
int fv_bind_workspace(fv_handle* h, void* ws, size_t bytes);

/* ---- frozen backbone ---------------------------------------------------------------------------------------- */
/* replaces FastVLMBackbone._prepare_images_tensor -> _resize_image -> resize_with_pad
 * (model/fastvlm_adapter.py:36-55,444-461,479-488): img (B,C,Hin,Win) NCHW f32|u8 (C in {1,3,4}) -> pix (B,S,S,4) bf16
 * NHWC (4th channel 0).  resize_with_padding=0 stretches instead (fastvlm_adapter.py:459-460). */
int fv_preprocess(fv_handle* h, const void* img, int dtype, int B, int C, int Hin, int Win, float pad_value,
                  int resize_with_padding, void* pix_out, fv_stream s);
/* fv_preprocess followed by FastVLMBackbone._maybe_normalize_imagenet (model/fastvlm_adapter.py:463-477; `normalize_imagenet=True`, off by default):
 * (x - mean[c]) / std[c] per channel on the letterboxed fp32 values (pad pixels included, as the reference normalises after padding), one bf16 rounding
 * at the end.  range_heuristic != 0 = the torchvision branch (:471-477): if the maximum of the whole letterboxed batch exceeds 1.5 the values are divided
 * by 255 first -- decided on the device from a maximum pass over the same arithmetic (no host synchronisation); 0 = the branch without torchvision
 * (:466-470).  mean / std: 3 floats each in HOST memory. */
int fv_preprocess_normalized(fv_handle* h, const void* img, int dtype, int B, int C, int Hin, int Win, float pad_value, int resize_with_padding,
                             const float* mean3, const float* std3, int range_heuristic, void* pix_out, fv_stream s);
/* replaces the vision tower + mm_projector inside LlavaQwen2ForCausalLM.forward (call site fastvlm_adapter.py:533):
 * pix (B,S,S,4) bf16 -> img_tokens (B, (S/64)^2, llm_hidden) f32.  tower_out (B,(S/64)^2,tower_out_dim) bf16 may be
 * NULL.
 * Kernel forms follow a launch's own tile count: at B <= 2 the fused ConvFFN runs in hidden ranges and the last stages' GEMMs in K ranges (another fp32
 * summation order: <= 1 bf16 step per output), at B <= 4 the depthwise march in row segments (bit-identical to the uncut march); B >= 4 gives an image the
 * same tokens whatever its neighbours.  The same holds for the decoder (fv_llm_forward_pooled: K ranges for <= 256 rows and for ragged row counts <= 1024);
 * the fv_train_* entry points never take these forms. */
int fv_vision_forward(fv_handle* h, const void* pix, int B, void* img_tokens, void* tower_out, fv_stream s);
/* on != 0: the tower keeps the large-batch kernel forms at every batch size (no hidden ranges of the fused ConvFFN, no K ranges in its last stages): an
 * image evaluated alone then gets the tokens it gets inside a batch, bit for bit (measured), at ~0.9 ms per step of a one-observation control loop (4.75 instead of 3.83 ms).
 * Off by default (the host side switches it with FASTVLA_BATCH_INVARIANT=1).  The row-segmented depthwise march is bit-identical and stays on. */
int fv_set_batch_invariant(fv_handle* h, int on);
/* fv_preprocess + fv_vision_forward in one (the on-device input pipeline): the stem kernel samples the SOURCE images
 * (B,C,Hin,Win) f32 | u8 through resize_with_pad's arithmetic itself (model/fastvlm_adapter.py:36-55,479-488 then :533), so the letterboxed
 * (B,S,S,4) frame is never written or read.  Same tokens as the two-call form, bit for bit.  Needs the fused stem (first stage width 96). */
int fv_vision_forward_images(fv_handle* h, const void* img, int dtype, int B, int C, int Hin, int Win, float pad_value, int resize_with_padding,
                             void* img_tokens, void* tower_out, fv_stream s);
/* fv_vision_forward that also copies intermediate activation maps out (parity tests name the failing stage with them):
 * taps[0] <- stem output (B,S/4,S/4,dims[0]), taps[1+i] <- output of stage i (B, S/(4<<i), S/(4<<i), dims[i]), bf16 NHWC
 * device buffers; NULL entries are skipped; n_taps <= tower_stages + 1.  The mirror of the oracle's `taps=` argument. */
int fv_vision_forward_taps(fv_handle* h, const void* pix, int B, void* img_tokens, void* tower_out, void* const* taps,
                           int n_taps, fv_stream s);
/* Tower UNITS in execution order: unit 0 = the three stem convolutions, then per stage [RepCPE] block ... block [PatchEmbed]
 * (mci.py `network` ModuleList order; 51 units for fastvithd).  fv_vision_unit_info describes unit u's OUTPUT map
 * (kind 0 stem / 1 RepCPE / 2 block / 3 PatchEmbed; side x side x channels, bf16 NHWC) and fails with FV_ERR_ARG past the last
 * unit.  fv_vision_forward_unit_taps is fv_vision_forward that also copies every unit's output to taps[u] (NULL entries
 * skipped): unit u's input is taps[u-1], so a parity test can feed the ORACLE unit the ENGINE's own input and compare one unit
 * at a time (teacher forcing: no error amplification through the 44 blocks).  Same call site as fv_vision_forward
 * (model/fastvlm_adapter.py:533). */
int fv_vision_unit_info(fv_handle* h, int unit, int32_t* kind, int32_t* stage, int32_t* side, int32_t* channels);
int fv_vision_forward_unit_taps(fv_handle* h, const void* pix, int B, void* img_tokens, void* tower_out, void* const* taps,
                                int n_taps, fv_stream s);
/* replaces embed_tokens + L x Qwen2DecoderLayer + final RMSNorm + FastVLMBackbone._pool_hidden
 * (fastvlm_adapter.py:533,551-559,337-359): ids (B,T) int32 right-padded, lens (B) int32,
 * img_tokens NULL (literal reference: text-only sequence) or (B,Ni,H) f32 spliced in front of the text.
 * pool_mode 0 = last_token, 1 = mean_pool.  pooled (B,H) f32. */
int fv_llm_forward_pooled(fv_handle* h, const int32_t* ids, const int32_t* lens, const void* img_tokens, int Ni, int B,
                          int T, int pool_mode, void* pooled, fv_stream s);

/* llm_precision >= 2 runs gate/up and down on fp16 operands.  Weights outside the binary16 range are REFUSED at load time
 * (fv_load_weights* returns FV_ERR_UNSUPPORTED: load with llm_precision = 1); activations are converted with SATURATING casts
 * (an outlier channel of a real checkpoint clamps to +-65504 instead of becoming inf / NaN actions) and every clamped 8-value
 * group is counted.  This reads the counter (0 in a healthy model; non-zero = the fp16 budget does not hold for this
 * checkpoint / input, switch to llm_precision = 1).  A status call: synchronises the device.  Call site whose fp32 arithmetic
 * this guards: model/fastvlm_adapter.py:183-191 (the reference loads fp32), :533. */
int fv_llm_fp16_saturations(fv_handle* h, uint64_t* count_out, int reset);

/* Image-prefix reuse (the call site the reference has is ONE full prefill per env step:
 * lerobot_fastvla/modeling_fastvla.py:119-125 -> model/fastvlm_adapter.py:519-536).  With the image tokens spliced in FRONT of the
 * text under a causal mask, the keys / values of the Ni image positions depend on the image alone:
 *   fv_llm_prefix            runs those positions through the decoder once and writes every layer's un-rotated [k | v] rows to
 *                            kv_out ([llm_layers][B * Ni][2 * kv_heads * head_dim] f32, caller-owned, fv_llm_prefix_bytes);
 *   fv_llm_forward_pooled_prefixed  runs ONLY the T text positions against that cache: the same pooled rows as
 *                            fv_llm_forward_pooled(img_tokens, Ni) up to the summation order of other tile shapes.
 * A caller that keeps kv per image (or per camera frame) skips tower, projector and prefix for a repeated frame or for further
 * prompts on the same image.  llm_precision 1 / 2 and head_dim 64 / 128 only; pool_mode must be 0 (last_token). */
int fv_llm_prefix_bytes(fv_handle* h, int B, int Ni, size_t* out_bytes);
int fv_llm_prefix(fv_handle* h, const void* img_tokens, int Ni, int B, void* kv_out, fv_stream s);
int fv_llm_forward_pooled_prefixed(fv_handle* h, const int32_t* ids, const int32_t* lens, const void* kv, int Ni, int B, int T,
                                   int pool_mode, void* pooled, fv_stream s);

/* ---- action expert (trainable; parameters live in ONE caller-owned flat f32 buffer) --------------------------- */
/* element offsets of the 12 head tensors inside the flat buffer, in state-dict order
 * (state_projection.0.{weight,bias}, state_projection.1.*, fusion.0.*, fusion.1.*, fusion.4.*, action_head.*);
 * offsets[12] = total element count. */
int fv_head_layout(fv_handle* h, int64_t offsets[13]);
int fv_head_saved_bytes(fv_handle* h, int B, size_t* out_bytes);
/* replaces FastVLMWithExpert.forward after the backbone (fastvla/fastvlm_with_expert.py:50-54).
 * training: 0 = inference; 1 = training, Dropout(p) applied with a Philox mask derived from (seed, offset); 2 = training without
 * dropout (p = 0).  Any non-zero value keeps the actions in normalised space when fv_head_set_io_norm folded the dataset
 * statistics in (the loss is computed against normalised targets).  saved = activations for backward. */
int fv_head_forward(fv_handle* h, const float* flat_params, const float* pooled, const float* states, int B,
                    int training, float dropout_p, uint64_t seed, uint64_t offset, float* actions, void* saved,
                    fv_stream s);
/* Dataset statistics folded into the head: replaces the STATE / ACTION parts of LeRobot's
 * NormalizerProcessorStep / UnnormalizerProcessorStep around the policy (lerobot_fastvla/processor_fastvla.py:34-48, MEAN_STD per
 * lerobot_fastvla/configuration_fastvla.py:21-27): fv_head_forward then computes LayerNorm((states - state_mean) / (state_std + eps))
 * ... and, when training == 0, returns actions * action_std + action_mean.  HOST vectors of state_dim / action_dim floats; all four
 * NULL switches the folding off.  Training keeps its loss in normalised action space (targets arrive normalised).
 * A configuration call: it synchronises the device before overwriting the handle's ONE statistics vector.  A hipGraph captured
 * over fv_head_forward keeps the on/off state of its capture: re-capture after switching the folding on or off (new VALUES
 * written while it stays on are picked up by replays, the vector's address does not change). */
int fv_head_set_io_norm(fv_handle* h, const float* state_mean, const float* state_std, const float* action_mean,
                        const float* action_std, float eps);
/* replaces F.mse_loss + autograd backward of the head (fastvla/modeling_fastvla.py:56,
 * lerobot_fastvla/modeling_fastvla.py:132; trainer.py:175): writes loss (1 f32, device) and ALL 12 grads into
 * flat_grads (overwritten, not accumulated). */
int fv_head_mse_backward(fv_handle* h, const float* flat_params, const float* actions, const float* targets, int B,
                         float dropout_p, const void* saved, float* loss, float* flat_grads, fv_stream s);
/* generic backward of the head from dL/dactions (B,A) f32 (what autograd hands a custom Function when the loss is
 * computed outside, e.g. lerobot_fastvla/modeling_fastvla.py:132 + loss.backward()); overwrites flat_grads. */
int fv_head_backward(fv_handle* h, const float* flat_params, const float* grad_actions, int B, float dropout_p,
                     const void* saved, float* flat_grads, fv_stream s);
/* replaces clip_grad_norm_ + AdamW.step (training/trainer.py:60-66,178-180;
 * lerobot_fastvla/configuration_fastvla.py:51-55): fused global-norm clip + decoupled-decay Adam on the flat buffers.
 * step is 1-based.  grad_norm_out (1 f32, device, may be NULL) receives the pre-clip norm. */
int fv_adamw_clip_step(fv_handle* h, float* flat_params, const float* flat_grads, float* m, float* v, int64_t n,
                       const fv_adamw_hparams* hp, int64_t step, float* grad_norm_out, fv_stream s);

/* gradient accumulation (training/trainer.py:96,171: accelerate sums micro-batch gradients before the optimiser step):
 * acc += grads over n floats; both 16-byte aligned flat head buffers. */
int fv_grad_accumulate(fv_handle* h, float* acc, const float* grads, int64_t n, fv_stream s);
/* grads *= *scale_dev (a device scalar, e.g. the upstream dL/dloss autograd hands compute_loss's backward) */
int fv_grad_scale(fv_handle* h, float* grads, int64_t n, const float* scale_dev, fv_stream s);

/* ---- data-parallel exchange (replaces accelerate/DDP's gradient all-reduce, training/trainer.py:68-78,175) ---------- */
/* One RCCL communicator per rank over xGMI.  rank 0 calls fv_comm_unique_id, the host side carries the 128 bytes to every
 * rank (torch.distributed store / broadcast), every rank calls fv_comm_init.  comm is an ncclComm_t. */
int fv_comm_unique_id(fv_handle* h, fv_rccl_id* id_out);
int fv_comm_init(fv_handle* h, const fv_rccl_id* id, int rank, int world, void** comm_out);
int fv_comm_destroy(fv_handle* h, void* comm);
/* ONE all-reduce (sum, in place) of the flat head gradient on stream s; the 1/world average is fv_adamw_hparams.grad_scale */
int fv_allreduce_grads(fv_handle* h, void* comm, float* flat_grads, int64_t n, fv_stream s);

/* ---- unfrozen-backbone training ------------------------------------------------------------------------
 * The reference's knob is fastvla/configuration_fastvla.py:23 `freeze_backbone` (applied at model/fastvlm_adapter.py:170-173) and is made
 * moot there by the unconditional no_grad at model/fastvlm_adapter.py:501; the step body is training/trainer.py:60-66,171-182 over ALL
 * parameters.  This slice: Qwen2 decoder (embedding, every layer, final norm) + mm_projector + action expert trainable, FastViT-HD tower
 * frozen, image tokens SPLICED in front of the text (a text-only sequence gives the projector no gradient), last_token pooling.
 *
 * All trainable parameters live in ONE caller-owned flat fp32 buffer (the master copy; same for gradients and Adam's m / v):
 *   [ action expert (fv_head_layout) | projector | embedding | layer 0 .. L-1 | final norm ]
 * with every matrix in the layout the library packs it in: q | k | v rows concatenated (packing 1), gate / up rows interleaved in blocks of
 * 8 ([8 gate | 8 up], packing 2).  fv_train_layout lists every tensor (name, element offset, rows x cols, gradient bucket, packing). */
typedef struct fv_train_tensor {
  char name[104];   /* canonical checkpoint key; packed tensors: "...self_attn.qkv_proj.{weight,bias}", "...mlp.gate_up_proj.weight" */
  int64_t offset, numel;
  int32_t rows, cols;
  int32_t bucket;   /* 0 = action expert, 1 = projector, 2 = embedding, 3 + l = decoder layer l, 3 + L = final norm */
  int32_t packing;  /* 0 = plain [rows][cols], 1 = q|k|v concatenated along rows, 2 = gate/up rows interleaved by 8 */
} fv_train_tensor;
/* out may be NULL (sizes only).  n_buckets = 3 + llm_layers + 1. */
int fv_train_layout(fv_handle* h, fv_train_tensor* out, int max_tensors, int* n_tensors, int64_t* total_numel, int* n_buckets);
/* one-time: allocates the library-owned transposed bf16 weight copies the dgrad GEMMs read.  Needs llm_precision = 1, head_dim 64 / 128. */
int fv_train_begin(fv_handle* h);
/* backbone part of the flat master buffer <- the library's current weights (the head part is the caller's) */
int fv_train_export_params(fv_handle* h, float* flat_params, fv_stream s);
/* the library's MFMA operand copies <- the master, after an optimiser step: bf16 weights (RNE), their transposes, fp32 norms / biases.
 * The frozen-path entry points (fv_llm_forward_pooled, ...) see the updated weights from then on. */
int fv_train_commit(fv_handle* h, const float* flat_params, fv_stream s);
/* Arithmetic of the backward's contractions (defaults 2, 1, 12: every dgrad and wgrad ONE fp16 pass -- worst per-tensor gradient 8.4e-4 from fp32 autograd
 * through the whole 24-layer 0.5B decoder, 6.7e-4 at the 7B width; in this mode the gate/up accumulators are kept as fp16 for the backward):
 *   grad_split 2: ONE fp16 pass -- the loss-scaled gradient rounded once to 11 significant bits against an fp16 copy of the transposed weight (exact:
 *                 bf16 widens into fp16);
 *              1: the gradient operand of every dgrad GEMM is split bf16 (hi + lo, 16 significant bits) against the exact-bf16 transposed weights
 *                 (two passes; the most exact form: 3.3e-4 through all 24 layers).
 *   wgrad_f16  1: every weight gradient in ONE fp16 pass -- the gradient and the activation each rounded once to 11 significant bits (operands as
 *                 transposed fp16 copies, made where the data is produced);
 *              0: the split-bf16 gradient against the activation's bf16 hi half (two passes; the activation's 8 bits bound the result at ~1.8e-3).
 *   (Round 5 removed grad_split 0 -- plain-bf16 gradient operands, 3.7e-3: outside the 2e-3 bar -- and wgrad_f16 2 -- the decoder's wgrads on the TN GEMM
 *   instance: same gradients, 1 ms per step slower.  The TN instance lives on in the tower's weight gradients, whose contraction runs over 10^5 .. 10^6 pixel rows.)
 *   loss_scale_log2: dL/dactions is multiplied by 2^k, so EVERY gradient fv_train_forward_backward writes carries that factor (it keeps the wgrad's
 *              fp16 gradient operand inside binary16's range; saturating casts, clamps counted by fv_llm_fp16_saturations); pass
 *              grad_scale = 2^-k / world to fv_adamw_clip_step (fv_train_loss_scale returns 2^k).  The loss itself is not scaled. */
int fv_train_set_options(fv_handle* h, int grad_split, int wgrad_f16, int loss_scale_log2);
int fv_train_loss_scale(fv_handle* h, float* scale_out);
/* on != 0: the TRAINING forward's four projections per layer in ONE fp16 pass (normed rows / attention output / SwiGLU output rounded once to 11 significant bits
 * against exact fp16 copies of the bf16 weights; down's copy carries 2^4) instead of the split-bf16 form's two -- half the forward's MFMA work; everything the
 * backward differentiates is kept as before.  The inference entry points do not change.  Needs the default fp16 backward.
 * OPT-IN, outside the 1e-3 bar: through all 24 layers of the 0.5B decoder actions sit 1.0e-3 and gradients up to 2.1e-3 from fp32 autograd (default forward:
 * 1.2e-5 / 8.4e-4; tests/test_gpu_train_unfrozen.py). */
int fv_train_set_forward_f16(fv_handle* h, int on);
int fv_train_workspace_bytes(fv_handle* h, int B, int T, size_t* out_bytes);
/* called from inside fv_train_forward_backward, on the calling thread, right after the LAST kernel that writes bucket `bucket`'s gradient
 * has been enqueued on the stream: flat_grads[offset, offset + numel) is final once the stream reaches this point (record an event here and
 * start that bucket's all-reduce on a side stream: it runs under the rest of the backward pass).  Order: 0, 3 + L, 3 + L - 1, ..., 3, 2, 1. */
typedef void (*fv_bucket_cb)(void* user, int bucket, int64_t offset, int64_t numel);
/* ONE training step's forward + MSE + backward over every trainable tensor (replaces model.compute_loss + accelerator.backward,
 * training/trainer.py:173-175, for an unfrozen backbone):
 *   tower_out (B, Ni, tower_out_dim) bf16 = the frozen tower's embeddings (fv_vision_forward's tower_out); ids (B, T) int32 right-padded,
 *   T % 8 == 0; lens (B); states (B, state_dim), targets (B, action_dim) f32; training / dropout as fv_head_forward.
 *   -> actions (B, action_dim) (normalised space), loss (1 f32, device), flat_grads (overwritten: every tensor of fv_train_layout, TIMES the
 *      loss scale of fv_train_set_options -- 2^12 by default).
 * ws: caller-owned scratch of fv_train_workspace_bytes(B, T) bytes (every activation the backward needs is kept there: no recompute).
 * Asynchronous on s; allocates nothing; gradients are bit-reproducible (no float atomics). */
int fv_train_forward_backward(fv_handle* h, const float* flat_params, const void* tower_out, const int32_t* ids, const int32_t* lens,
                              const float* states, const float* targets, int B, int T, int training, float dropout_p, uint64_t seed,
                              uint64_t offset, void* ws, size_t ws_bytes, float* actions, float* loss, float* flat_grads, fv_bucket_cb cb,
                              void* user, fv_stream s);

/* ---- the TOWER half of the slice (FastViT-HD trainable too; csrc/tower_train.inc).  After fv_train_begin:
 * fv_train_tower_begin appends the tower's tensors (inference form: folded RepMixer / ConvFFN convolutions, fc1 / fc2, layer scales, attention, PatchEmbed,
 * stem, conv_exp + SE) to the flat master -- fv_train_layout then lists them after "model.norm.weight" (packing 3 = depthwise weights tap-major [k*k][C],
 * 4 = the stem's dense 3x3 as [27][C0] with row (ky*3+kx)*3+ci; new gradient buckets 3 + L + 1 ...: stem, then per stage [blocks (+RepCPE)] [PatchEmbed],
 * last conv_exp + SE) -- and fv_train_export_params / fv_train_commit cover them (commit refreshes every packed operand image the kernels read).
 * One step: fv_train_tower_forward (pixels -> tower_out, every unit's tensors kept in tws) -> fv_train_forward_backward with fv_train_set_tower_grad's buffer
 * bound (it then also leaves dL/d(tower_out) there, fp16 [B][tokens][tower_out_dim], loss-scaled) -> fv_train_tower_backward (tower gradients into the same
 * flat_grads, cb per tower bucket in backward order).  pix must stay alive between the two tower calls. */
int fv_train_tower_begin(fv_handle* h);
int fv_train_tower_workspace_bytes(fv_handle* h, int B, size_t* out_bytes);
int fv_train_tower_forward(fv_handle* h, const void* pix, int B, void* tws, size_t tws_bytes, void* tower_out, fv_stream s);
/* Binds (NULL: unbinds) the buffer fv_train_forward_backward leaves dL/d(tower_out) in.  That gradient carries a power-of-two scale of its own, chosen on the device every
 * step and taken out again by fv_train_tower_backward.  A CHANGE of binding resets the scale to 1 (a configuration call: it synchronises the device); binding the buffer
 * that is already bound is a no-op, so a training loop may call this every step.  A caller that fills a gradient buffer ITSELF and hands it to fv_train_tower_backward
 * (no fv_train_forward_backward in between) must have changed the binding since the last fv_train_forward_backward -- e.g. bind NULL first, as the unit tests do. */
int fv_train_set_tower_grad(fv_handle* h, void* d_tower_out_f16);
int fv_train_tower_backward(fv_handle* h, const void* pix, const void* d_tower_out_f16, int B, void* tws, size_t tws_bytes, float* flat_grads, fv_bucket_cb cb,
                            void* user, fv_stream s);
/* ONE tower unit, teacher-forced (parity tests): forward of unit `unit` (fv_vision_unit_info's index; the unit count itself = conv_exp + SE) on x_in (bf16 NHWC;
 * the stem: fv_preprocess pixels), then its backward from g_out (fp32 NHWC, dL/d(output)) x gscale.  y_out (bf16, optional) = the unit's output, g_in (fp32,
 * optional; ignored for the stem) = dL/d(input), the unit's weight gradients x gscale at their fv_train_layout offsets in flat_grads. */
/* unit `unit`'s output (bf16 NHWC) as the last fv_train_tower_forward left it in tws */
int fv_train_tower_read_unit(fv_handle* h, int unit, int B, const void* tws, size_t tws_bytes, void* out, fv_stream s);
int fv_train_tower_unit(fv_handle* h, int unit, const void* x_in, const float* g_out, float gscale, int B, void* tws, size_t tws_bytes, void* y_out, float* g_in,
                        float* flat_grads, fv_stream s);

/* ---- optional per-kernel-family HIP-event timing (bench.py roofline numbers) ------------------------------------- */
enum fv_family { FV_FAM_GEMM = 0, FV_FAM_DWCONV, FV_FAM_STEM, FV_FAM_ATTN, FV_FAM_NORM, FV_FAM_ELT, FV_FAM_HEAD, FV_FAM_COUNT };
typedef struct fv_profile_entry { double ms, flops, bytes; int64_t launches; } fv_profile_entry;
typedef struct fv_gemm_profile { int32_t m, n, k, epi; double ms; int64_t launches; } fv_gemm_profile;
/* enable != 0: every launch the engine issues is bracketed by hipEvents on the caller's stream (algorithmic flops and
 * bytes recorded beside it).  fv_profile_read synchronises those events, fills fam_out[FV_FAM_COUNT] and up to max_gemm
 * distinct GEMM shapes, and clears the record list. */
int fv_profile(fv_handle* h, int enable);
int fv_profile_read(fv_handle* h, fv_profile_entry* fam_out, fv_gemm_profile* gemm_out, int max_gemm, int* n_gemm);

/* The op-level entry points of the parity tests (fv_op_*: one kernel each) are NOT part of this library: they are declared in
 * include/fastvla_hip_testops.h and built into tests/_native/libfastvla_hip_testops.so by `make` (csrc/ops_api.hip). */

#ifdef __cplusplus
}
#endif
#endif /* FASTVLA_HIP_H */
