// engine.hip -- the native runtime behind the C ABI (include/fastvla_hip.h): owns the packed frozen weights, plans the
// caller-owned workspace and issues the kernel schedule of the policy step on the caller's stream.
//
// Replaces, on the reference side:
//   FastVLMBackbone.__init__/_load_model            model/fastvlm_adapter.py:90-201     -> fv_create + fv_load_weights
//   FastVLMBackbone._prepare_images_tensor          model/fastvlm_adapter.py:479-488    -> fv_preprocess
//   self.model(**inputs) [LlavaQwen2ForCausalLM]     model/fastvlm_adapter.py:533        -> fv_vision_forward + fv_llm_forward_pooled
//   FastVLMBackbone._pool_hidden                     model/fastvlm_adapter.py:337-359    -> fused into fv_llm_forward_pooled
//   FastVLMWithExpert head, MSE, backward, optimiser fastvla/fastvlm_with_expert.py:50-54, trainer.py:171-182 -> fv_head_*
// Work the reference does and this path deliberately does not: lm_head logits, retained per-layer hidden states,
// KV-cache write-back, the D2H -> CPU resize -> H2D round trip.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"

// ------------------------------------------------------------------------------------------------ error plumbing
// fv_fail() formats into a thread-local staging buffer (what fv_last_error(NULL) returns: errors of fv_create and of the
// handle-less op-level entry points) and, when the failing call was made on a handle, into that handle's own buffer:
// fv_last_error(h) is per handle, as the header promises.
static thread_local char g_err[768] = "";
static thread_local char* g_err_handle = nullptr;  // err[] of the handle whose entry point is executing on this thread
int fv_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  if (g_err_handle) memcpy(g_err_handle, g_err, sizeof(g_err));
  return code;
}
int fv_hip_fail(hipError_t e, const char* what) { return fv_fail(FV_ERR_HIP, "HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what); }

namespace {


const char* VT = "model.vision_tower.vision_tower.model.";
const char* PJ = "model.mm_projector.";
const char* LM = "model.";

inline bf16_t host_f2bf(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
inline float host_bf2f(bf16_t b) {
  uint32_t u = ((uint32_t)b) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

struct FFN { float *dw_w = nullptr, *dw_b = nullptr; bf16_t* fc1_w = nullptr; float* fc1_b = nullptr; bf16_t* fc2_w = nullptr; float *fc2_b = nullptr, *ls = nullptr; bf16_t* w2p = nullptr; bf16_t* w2q = nullptr; bf16_t* dw_t = nullptr; };
struct Block {
  float *mix_w = nullptr, *mix_b = nullptr; bf16_t* mix_t = nullptr;  // RepMixer (+ Toeplitz table for the MFMA path)
  float *ln_w = nullptr, *ln_b = nullptr; bf16_t *qkv_w = nullptr, *proj_w = nullptr; float *proj_b = nullptr, *ls1 = nullptr;  // attention
  FFN ffn;
};
struct Down { float *lk_w = nullptr, *lk_b = nullptr; bf16_t* lk_t = nullptr; bf16_t* pw_w = nullptr; float* pw_b = nullptr; };
struct Cpe { float *w = nullptr, *b = nullptr; bf16_t* t = nullptr; };
struct Tower {
  bf16_t* stem0_wp = nullptr;  // [C0][64] image for the MFMA stem
  float *stem0_w = nullptr, *stem0_b = nullptr, *stem1_w = nullptr, *stem1_b = nullptr; bf16_t* stem2_w = nullptr; float* stem2_b = nullptr;
  std::vector<std::vector<Block>> stages; std::vector<Down> downs; std::vector<Cpe> cpes;
  float *exp_w = nullptr, *exp_b = nullptr, *se_w1 = nullptr, *se_b1 = nullptr, *se_w2 = nullptr, *se_b2 = nullptr;
  bf16_t *pj0_w = nullptr, *pj2_w = nullptr; float *pj0_b = nullptr, *pj2_b = nullptr;
};
struct TrainLayerT {   // transposed weight copies (dgrad operands): bf16, and fp16 for the one-pass fp16 dgrad
  bf16_t *qkvT = nullptr, *oT = nullptr, *guT = nullptr, *downT = nullptr;
  bf16_t *qkvT16 = nullptr, *oT16 = nullptr, *guT16 = nullptr, *downT16 = nullptr;
  bf16_t *qkv16 = nullptr, *o16 = nullptr, *gu16 = nullptr, *down16 = nullptr;   // fp16 ROW copies (down x 2^4): the weight operands of the one-pass fp16 training forward (fv_train_set_forward_f16)
};
struct GTab { int* idx = nullptr; float* coef = nullptr; size_t n = 0; };   // gather table of one packed operand image (tower_train.inc make_gtab), device arrays
struct TowerTrainUnit {   // the backward's own operand images of one tower unit: transposed fp16 dgrad operands of the dense layers ...
  bf16_t *fc1T16 = nullptr, *fc2sT16 = nullptr, *qkvT16 = nullptr, *projsT16 = nullptr, *pwT16 = nullptr;
  bf16_t* stem0_w32 = nullptr;                  // (stem) the first conv as fp16 [C0][32]: taps (bf16-rounded, as the forward's MFMA image holds them) in columns 0 .. 26, zeros behind
  bf16_t *mixF16 = nullptr, *dwF16 = nullptr;   // ... and fp16 Toeplitz tables of the FLIPPED depthwise taps (3x3 mixer / RepCPE 7x7 in mixF16, the ConvFFN's 7x7 in dwF16): dgrad on the marching MFMA kernel
};
struct TrainState {
  bool ready = false; std::vector<TrainLayerT> layers; bf16_t *pj2T = nullptr, *pj2T16 = nullptr;
  bool fwd_f16 = false;   // fv_train_set_forward_f16: the TRAINING forward's projections in ONE fp16 pass (half the MFMA work of the split-bf16 form)
  // the tower half (tower_train.inc): its tensors join the flat master when `tower` is set
  bool tower = false; std::vector<TowerTrainUnit> tunits; std::map<std::string, GTab> gtabs;
  fv::TowerCommitOp* tower_ops = nullptr; int tower_nops = 0, tower_blocks = 0;
  bf16_t *pj0T = nullptr, *pj0T16 = nullptr;   // the projector's first Linear, transposed (its input gradient feeds the tower)
  float* tscale = nullptr;                     // device: [0] = the tower gradient stream's own power-of-two scale (chosen per step from dL/d(tower_out)), [1] = its inverse; [2] = amax scratch bits
  void* d_tower_out = nullptr;                 // fv_train_set_tower_grad: where fv_train_forward_backward leaves dL/d(tower_out) as fp16 rows
  fv::CommitDesc* commit_desc = nullptr; int commit_n = 0, commit_tiles = 0;   // fv_train_commit's descriptor table (device)
  int grad_split = 2;   // dgrad's gradient operand: 2 ONE fp16 pass against fp16 transposed weights, 1 split bf16 (hi + lo, two passes) (fv_train_set_options)
  int wgrad_f16 = 1;    // 1: the weight gradients in ONE fp16 pass (both operands 11 significant bits, the gradient carrying the loss scale); 0: split-bf16 gradient x bf16 activation
  int loss_scale_log2 = 12;   // every gradient the backward produces is multiplied by 2^this (the optimiser's grad_scale takes it out again): fp16's range for the wgrad operands
};
struct DecLayer {
  float* ln1 = nullptr; bf16_t* qkv_w = nullptr; float* qkv_b = nullptr; bf16_t* o_w = nullptr; float* ln2 = nullptr; bf16_t *gu_w = nullptr, *down_w = nullptr;
  void *qkv_w8 = nullptr, *o_w8 = nullptr, *gu_w8 = nullptr, *down_w8 = nullptr;   // llm_precision = 5: fp8 copies (x 2^6, row stride 2K bytes) for the lo8 products
};
struct Decoder { bf16_t* embed = nullptr; std::vector<DecLayer> layers; float* norm = nullptr; };

struct WsPlan {
  size_t bufA, bufB, bufH, se, tower_out, x, xn, qkv, att, act, xn_lo, att_lo, act_lo, qkvf, guf, splitk, splitk_bytes, head_scr, attn_scr, attn_scr_bytes, total;
};

// optional per-launch HIP-event timing on the caller's stream (bench.py's roofline numbers); off by default
struct ProfRec { int fam; double flops, bytes; int m, n, k, epi; hipEvent_t e0, e1; };
struct Profiler {
  bool on = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  hipEvent_t get() {
    if (used == pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; pool.push_back(e); }
    return pool[used++];
  }
};

}  // namespace

struct fv_handle;
namespace {
struct HandleScope {  // first statement of every entry point that takes a handle
  char* prev;
  explicit HandleScope(fv_handle* h);
  ~HandleScope() { g_err_handle = prev; }
};
}  // namespace

struct fv_handle {
  fv_model_desc d;
  int device = 0;
  bool loaded = false;
  std::vector<void*> allocs;
  Tower tw;
  Decoder dec;
  float2* rope = nullptr;
  int rope_rows = 0;
  float* norm_scratch = nullptr;
  void* ws = nullptr;
  size_t ws_bytes = 0;
  fv::HeadDims hd;
  Profiler prof;
  char err[768] = "";         // last error of a call made on THIS handle (fv_last_error(h)); fv_fail's thread-local
                              // buffer is only the staging area
  void* const* taps = nullptr;  // fv_vision_forward_taps: per-stage copies of the activation map (parity tests)
  int n_taps = 0;
  struct LbSrc { const void* img; int dtype, C, Hin, Win; float pad; int letterbox; };
  unsigned* lb_vmax = nullptr;   // fv_preprocess_normalized: the letterboxed batch's maximum (ordered-integer image), 4 device bytes
  const LbSrc* lbsrc = nullptr;  // fv_vision_forward_images: the stem samples the source images itself (no letterboxed frame)
  void* const* utaps = nullptr; // fv_vision_forward_unit_taps: one copy per tower unit (stem, RepCPE, block, PatchEmbed)
  int n_utaps = 0;
  void* rccl = nullptr;       // dlopen handle of librccl (fv_comm_* / fv_allreduce_grads), resolved on first use
  fv::HeadIoNorm io{nullptr, nullptr, nullptr, nullptr};   // fv_head_set_io_norm: dataset statistics folded into the head
  bool has_io = false;
  float* io_buf = nullptr;    // ONE device vector of 2 ds + 2 da floats per handle, overwritten by every fv_head_set_io_norm
  bool no_fused_ffn = false;  // FASTVLA_NO_FUSED_FFN=1: A/B switch back to the two-GEMM ConvFFN
  bool no_mfma_dw = false;    // FASTVLA_NO_MFMA_DW=1: A/B switch back to the VALU depthwise kernels
  bool no_ffn32 = false;      // FASTVLA_NO_FFN32=1: A/B switch back to the 16x16x32 fused ConvFFN
  TrainState train;               // unfrozen-backbone training (train_path.inc): library-owned transposed bf16 weight copies
  int batch_invariant = 0;        // fv_set_batch_invariant: the inference tower never takes the range forms either
  int train_depth = 0;            // > 0 while a training entry point runs (TrainScope): gemm_p then keeps the few-row GEMM forms off
  float* ffn_part = nullptr;      // device: the fused ConvFFN's partial sums when a launch has few row tiles (B <= 4: launch_convffn32's hidden ranges), FFN_PART_BYTES
  unsigned* f16_flags = nullptr;  // device: [0] = activation groups clamped to the fp16 range (fv_llm_fp16_saturations),
                                  // [1] = max |scaled weight| bits seen by the loader's in-place fp16 conversion
};

namespace {

HandleScope::HandleScope(fv_handle* h) : prev(g_err_handle) { g_err_handle = h ? h->err : nullptr; }

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// up to 8 ranges x 128 row tiles x 128 rows x 384 channels of fp32 (C = 384 at B = 4; the narrower stages need as much at most)
constexpr size_t FFN_PART_BYTES = (size_t)8 * 128 * 128 * 384 * 4 / 2 + (1 << 20);
struct TrainScope {   // RAII: every return path of a training entry point leaves the handle as it found it
  fv_handle* h;
  explicit TrainScope(fv_handle* hh) : h(hh) { if (h) ++h->train_depth; }   // null-tolerant as HandleScope is: the entry point's own check reports FV_ERR_ARG
  ~TrainScope() { if (h) --h->train_depth; }
  TrainScope(const TrainScope&) = delete;
  TrainScope& operator=(const TrainScope&) = delete;
};
int dev_alloc(fv_handle* h, size_t bytes, void** out) {
  void* p = nullptr;
  FV_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16));
  h->allocs.push_back(p);
  *out = p;
  return FV_OK;
}

struct Loader {
  fv_handle* h;
  std::map<std::string, const fv_tensor_desc*> idx;   // fv_load_weights: every tensor up front
  fv_tensor_provider prov = nullptr;                   // fv_load_weights_cb: one tensor at a time, on demand
  void* user = nullptr;
  fv_tensor_desc cur{};
  std::vector<char> stage;                             // host staging of a device-resident tensor that needs host packing
  int rc = FV_OK;

  static size_t count(const fv_tensor_desc* t) {
    size_t n = 1;
    for (int i = 0; i < t->ndim; ++i) n *= (size_t)t->shape[i];
    return n;
  }
  // the descriptor of `name` (valid until the next need()), or null with rc set
  const fv_tensor_desc* need(const std::string& name) {
    const fv_tensor_desc* t = nullptr;
    if (prov) {
      memset(&cur, 0, sizeof(cur));
      if (prov(user, name.c_str(), &cur) == 0 && cur.data) t = &cur;
    } else {
      auto it = idx.find(name);
      if (it != idx.end()) t = it->second;
    }
    if (!t) {
      if (rc == FV_OK) rc = fv_fail(FV_ERR_MISSING, "missing weight tensor '%s'", name.c_str());
      return nullptr;
    }
    if ((t->dtype != FV_F32 && t->dtype != FV_BF16) || t->ndim < 1 || t->ndim > 4) {
      if (rc == FV_OK) rc = fv_fail(FV_ERR_ARG, "tensor '%s': dtype must be f32 or bf16 and rank 1..4", name.c_str());
      return nullptr;
    }
    return t;
  }
  const fv_tensor_desc* need_n(const std::string& name, size_t n) {
    const fv_tensor_desc* t = need(name);
    if (t && count(t) != n) {
      if (rc == FV_OK) rc = fv_fail(FV_ERR_ARG, "tensor '%s' has %zu elements, expected %zu", name.c_str(), count(t), n);
      return nullptr;
    }
    return t;
  }
  bool to_f32(const fv_tensor_desc* t, std::vector<float>& out) {
    const size_t n = count(t), esz = t->dtype == FV_F32 ? 4 : 2;
    const void* src = t->data;
    if (t->device) {
      stage.resize(n * esz);
      if (hipMemcpy(stage.data(), t->data, n * esz, hipMemcpyDeviceToHost) != hipSuccess) {
        if (rc == FV_OK) rc = fv_fail(FV_ERR_HIP, "hipMemcpy (device tensor -> host) failed");
        return false;
      }
      src = stage.data();
    }
    out.resize(n);
    if (t->dtype == FV_F32) memcpy(out.data(), src, n * 4);
    else { const bf16_t* b = static_cast<const bf16_t*>(src); for (size_t i = 0; i < n; ++i) out[i] = host_bf2f(b[i]); }
    return true;
  }
  bool get(const std::string& name, std::vector<float>& out, std::vector<int64_t>* shape = nullptr) {
    const fv_tensor_desc* t = need(name);
    if (!t || !to_f32(t, out)) return false;
    if (shape) shape->assign(t->shape, t->shape + t->ndim);
    return true;
  }
  bool expect(const std::string& name, std::vector<float>& v, size_t n) {
    const fv_tensor_desc* t = need_n(name, n);
    return t && to_f32(t, v);
  }
  void* alloc(size_t bytes) {
    void* p = nullptr;
    if (dev_alloc(h, bytes, &p) != FV_OK) { if (rc == FV_OK) rc = FV_ERR_HIP; return nullptr; }
    return p;
  }
  float* up_f32(const std::vector<float>& v) {
    void* p = alloc(v.size() * 4);
    if (p && hipMemcpy(p, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { if (rc == FV_OK) rc = fv_fail(FV_ERR_HIP, "hipMemcpy failed"); }
    return static_cast<float*>(p);
  }
  bf16_t* up_bf16(const std::vector<float>& v) {
    std::vector<bf16_t> b(v.size());
    for (size_t i = 0; i < v.size(); ++i) b[i] = host_f2bf(v[i]);
    void* p = alloc(b.size() * 2);
    if (p && hipMemcpy(p, b.data(), b.size() * 2, hipMemcpyHostToDevice) != hipSuccess) { if (rc == FV_OK) rc = fv_fail(FV_ERR_HIP, "hipMemcpy failed"); }
    return static_cast<bf16_t*>(p);
  }
  // rows x row_elems bf16 of tensor t into dst with a destination pitch (elements): the building block of the packed
  // layouts (qkv concatenation, gate/up interleave).  bf16 sources (host or device) are copied as they are -- a 7B
  // checkpoint streams through without an fp32 round trip; f32 sources are rounded on the host first.
  bool put_rows(const fv_tensor_desc* t, bf16_t* dst, size_t dst_pitch, size_t row_elems, size_t rows, size_t src_pitch) {
    std::vector<bf16_t> tmp;
    const void* src = t->data;
    hipMemcpyKind kind = t->device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (t->dtype == FV_F32) {
      std::vector<float> v;
      if (!to_f32(t, v)) return false;
      tmp.resize(v.size());
      for (size_t i = 0; i < v.size(); ++i) tmp[i] = host_f2bf(v[i]);
      src = tmp.data();
      kind = hipMemcpyHostToDevice;
    }
    if (hipMemcpy2D(dst, dst_pitch * 2, src, src_pitch * 2, row_elems * 2, rows, kind) != hipSuccess) {
      if (rc == FV_OK) rc = fv_fail(FV_ERR_HIP, "hipMemcpy2D (weight packing) failed");
      return false;
    }
    return true;
  }
  // f32 vector straight up
  float* vec(const std::string& name, size_t n) { std::vector<float> v; return expect(name, v, n) ? up_f32(v) : nullptr; }
  // [N][K] (or [N,K,1,1]) -> bf16
  bf16_t* mat(const std::string& name, size_t n, size_t k) {
    const fv_tensor_desc* t = need_n(name, n * k);
    if (!t) return nullptr;
    bf16_t* p = static_cast<bf16_t*>(alloc(n * k * 2));
    return p && put_rows(t, p, n * k, n * k, 1, n * k) ? p : nullptr;
  }
  // depthwise [Cout,1,k,k] -> tap-major [k*k][Cout] f32, optionally scaled per out-channel (BN fold)
  // map_w > 0: also builds the bf16 Toeplitz table of the MFMA depthwise kernel when that kernel serves this layer
  // s2_cin > 0: a stride-2 layer with cout / s2_cin outputs per input channel on an s2_map-wide map; builds the table of
  // dwconv_s2_mfma_kernel when that kernel serves it
  float* dw(const std::string& name, int cout, int k, const std::vector<float>* scale = nullptr, int map_w = 0, bf16_t** ttab = nullptr,
            int s2_cin = 0, int s2_map = 0, bf16_t** s2_ttab = nullptr) {
    std::vector<float> v;
    if (!expect(name, v, (size_t)cout * k * k)) return nullptr;
    std::vector<float> o((size_t)cout * k * k);
    for (int c = 0; c < cout; ++c)
      for (int t = 0; t < k * k; ++t) o[(size_t)t * cout + c] = v[(size_t)c * k * k + t] * (scale ? (*scale)[c] : 1.0f);
    if (ttab && fv::dwconv_mfma_supported(map_w, cout, k, 1, 1)) {
      std::vector<float> tt(fv::dwconv_toeplitz_elems(cout, k));
      fv::dwconv_toeplitz_pack(o.data(), tt.data(), cout, k);
      *ttab = up_bf16(tt);
    }
    if (s2_ttab && s2_cin > 0 && cout % s2_cin == 0 && fv::dwconv_s2_mfma_supported(s2_map, s2_map, s2_cin, k, 2, cout / s2_cin)) {
      std::vector<float> tt(fv::dwconv_s2_toeplitz_elems(s2_cin));
      fv::dwconv_s2_toeplitz_pack(o.data(), tt.data(), s2_cin);
      *s2_ttab = up_bf16(tt);
    }
    return up_f32(o);
  }
};

int load_ffn(Loader& L, const std::string& pre, int C, int hidden, float bn_eps, FFN& f, const std::string& ls_name, int map_w) {
  std::vector<float> g, b, m, v;
  if (!L.expect(pre + "conv.bn.weight", g, C) || !L.expect(pre + "conv.bn.bias", b, C) ||
      !L.expect(pre + "conv.bn.running_mean", m, C) || !L.expect(pre + "conv.bn.running_var", v, C))
    return L.rc;
  std::vector<float> sc(C), bias(C);
  for (int c = 0; c < C; ++c) {
    sc[c] = g[c] / std::sqrt(v[c] + bn_eps);
    bias[c] = b[c] - m[c] * sc[c];
  }
  f.dw_w = L.dw(pre + "conv.conv.weight", C, 7, &sc, map_w, &f.dw_t);
  f.dw_b = L.up_f32(bias);
  f.fc1_w = L.mat(pre + "fc1.weight", hidden, C);
  f.fc1_b = L.vec(pre + "fc1.bias", hidden);
  f.fc2_w = L.mat(pre + "fc2.weight", C, hidden);
  if (hidden == 4 * C && fv::convffn_supported(C, 4)) {  // hidden-permuted, chunk-major copy for the fused kernel
    std::vector<float> w2, w2p;
    if (L.expect(pre + "fc2.weight", w2, (size_t)C * hidden)) {
      w2p.resize(w2.size());
      fv::convffn_pack_w2(w2.data(), w2p.data(), C, hidden);
      f.w2p = L.up_bf16(w2p);
      std::vector<float> w1;
      if (fv::convffn32_supported(C, 4) && L.expect(pre + "fc1.weight", w1, (size_t)hidden * C)) {  // the 32x32x16 kernel's stream
        std::vector<float> wq((size_t)2 * hidden * C);
        fv::convffn32_pack(w1.data(), w2.data(), wq.data(), C);
        f.w2q = L.up_bf16(wq);
      }
    }
  }
  f.fc2_b = L.vec(pre + "fc2.bias", C);
  f.ls = L.vec(ls_name, C);
  return L.rc;
}

WsPlan plan_ws(const fv_handle* h, int B, int T, int splice) {
  const fv_model_desc& d = h->d;
  WsPlan p{};
  const int mb = (d.tower_microbatch > 0 && d.tower_microbatch < B) ? d.tower_microbatch : B;
  const size_t S = d.image_size;
  size_t act = (S / 2) * (S / 2) * (size_t)d.tower_dims[0];  // stem0 output, per image (elements)
  size_t hid = 0;
  size_t hw = (S / 4) * (S / 4);
  for (int i = 0; i < d.tower_stages; ++i) {
    const size_t C = d.tower_dims[i];
    act = std::max(act, hw * C);
    hid = std::max(hid, hw * C * d.tower_mlp_ratio);
    if (d.tower_is_attn[i]) hid = std::max(hid, hw * C * 3);
    if (i + 1 < d.tower_stages) {
      hw /= 4;
      act = std::max(act, hw * (size_t)d.tower_dims[i + 1]);
    }
  }
  const size_t P = hw;
  act = std::max(act, P * (size_t)d.tower_out_dim);
  hid = std::max(hid, P * (size_t)d.llm_hidden);
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += align_up(bytes); return r; };
  p.bufA = take(act * mb * 2);
  p.bufB = take(act * mb * 2);
  p.bufH = take(hid * mb * 2);
  p.se = take(((size_t)mb * (2 * d.tower_out_dim + d.tower_se_rd)) * 4);
  p.tower_out = take((size_t)B * P * d.tower_out_dim * 2);
  const size_t rows = (size_t)B * (T + (splice ? P : 0));
  const size_t qkvw = (size_t)(d.llm_heads + 2 * d.llm_kv_heads) * d.llm_head_dim;
  p.x = take(rows * d.llm_hidden * 4);
  p.xn = take(rows * d.llm_hidden * 2);
  p.qkv = take(rows * qkvw * 2);
  p.att = take(rows * (size_t)d.llm_heads * d.llm_head_dim * 2);
  p.act = take(rows * d.llm_inter * 2);
  if (d.llm_precision >= 1) {
    p.xn_lo = take(rows * d.llm_hidden * 2 * 2);                               // [hi | lo] side by side
    p.att_lo = take(rows * (size_t)d.llm_heads * d.llm_head_dim * 2 * 2);
    p.act_lo = (d.llm_precision == 1 || d.llm_precision == 3 || d.llm_precision == 5) ? take(rows * d.llm_inter * 2 * 2) : 0;    // modes 2, 4: the SwiGLU output is ONE fp16 row (p.act)
    p.qkvf = take(rows * qkvw * 4);
    p.guf = 0;  // gate/up accumulators no longer round-trip through memory (SwiGLU + split fused into the GEMM epilogue)
  }
  // split-K partial sums of the down projection (fp32, up to 4 splits of [rows][hidden padded to 256]); launch_gemm only
  // uses it when the problem has too few output tiles for the chip
  // (round 3: qkv and o use it too when their 256-tiles are fewer than the CUs -- the 7B widths at M = 1024 --, so it is sized for the wider of the two outputs)
  // up to 8 splits for few rows (C5's rank shape, 7B at M = 512: 28 tiles of the down projection), 4 otherwise
  // few rows (rows <= 256: one to four observations in the control loop): up to 40 K ranges of 64-row tiles (launch_gemm's few-row rule)
  p.splitk_bytes = rows <= 256 || (rows % 256 != 0 && rows <= 1024) ? (size_t)(rows <= 256 ? 40 : 24) * rows * ((std::max((size_t)d.llm_hidden, qkvw) + 255) / 256 * 256) * 4
                               : rows % 256 == 0 ? (size_t)(rows <= 2048 ? 8 : 4) * rows * ((std::max((size_t)d.llm_hidden, qkvw) + 255) / 256 * 256) * 4 : 0;
  p.splitk = take(p.splitk_bytes);
  p.head_scr = take(fv::head_bwd_scratch_bytes(h->hd, B));
  // long sequences (the spliced prefill: 256 image + T text positions) run the split-bf16 attention kernel: its per-layer K / V records
  const int Tseq = T + (splice ? (int)P : 0);
  p.attn_scr_bytes = (d.llm_precision != 0 && Tseq >= fv::FV_ATTN_SPLIT_MIN_T && (d.llm_head_dim == 64 || d.llm_head_dim == 128))
                         ? fv::attention_split_scratch_bytes(B, Tseq, d.llm_kv_heads, d.llm_head_dim) : 0;
  p.attn_scr = take(p.attn_scr_bytes);
  p.total = o;
  return p;
}

int check_ready(fv_handle* h, bool need_ws) {
  if (!h) return fv_fail(FV_ERR_ARG, "null handle");
  if (!h->loaded) return fv_fail(FV_ERR_STATE, "weights not loaded: call fv_load_weights first");
  if (need_ws && !h->ws) return fv_fail(FV_ERR_STATE, "workspace not bound: call fv_bind_workspace first");
  return FV_OK;
}

#define FV_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != FV_OK) return _rc; \
  } while (0)

void prof_begin(fv_handle* h, int fam, double flops, double bytes, hipStream_t s, int m = 0, int n = 0, int k = 0, int epi = 0) {
  if (!h->prof.on) return;
  ProfRec r{fam, flops, bytes, m, n, k, epi, h->prof.get(), h->prof.get()};
  if (!r.e0 || !r.e1) return;
  (void)hipEventRecord(r.e0, s);
  h->prof.recs.push_back(r);
}
void prof_end(fv_handle* h, hipStream_t s) {
  if (!h->prof.on || h->prof.recs.empty()) return;
  (void)hipEventRecord(h->prof.recs.back().e1, s);
}
#define FV_P(fam, flops, bytes, call)            \
  do {                                           \
    prof_begin(h, fam, flops, bytes, s);         \
    int _rc = (call);                            \
    prof_end(h, s);                              \
    if (_rc != FV_OK) return _rc;                \
  } while (0)

int gemm_p(fv_handle* h, const fv::GemmArgs& g_in, hipStream_t s) {
  fv::GemmArgs g = g_in;
  if (h->train_depth > 0) g.few_rows = 0;   // inside fv_train_forward_backward / fv_train_tower_*: the large-batch kernels at every row count
  const double M = g.M, N = g.N, K = g.K;
  const bool f32o = g.epi == FV_EPI_RES_F32 || g.epi == FV_EPI_F32;
  double bytes = (M * K + N * K) * 2 + M * ((g.epi == FV_EPI_SWIGLU || g.epi == FV_EPI_SWIGLU_F16) ? N / 2 : N) * (f32o ? 4 : 2);  // SPLIT: N bf16 columns
  if (g.epi == FV_EPI_LS_RES) bytes += M * N * 2;
  if (g.epi == FV_EPI_RES_F32) bytes += M * N * 4;
  // algorithmic flops (2MNK) even when ksplit executes the K loop twice for the split-bf16 operand
  prof_begin(h, FV_FAM_GEMM, 2.0 * M * N * K, bytes + (g.ksplit ? M * K * 2 : 0), s, g.M, g.N, g.ksplit ? -g.K : g.K, g.epi);
  const int rc = fv::launch_gemm(g, s);
  prof_end(h, s);
  return rc;
}
double dw_flops(int B, int Ho, int Wo, int Cout, int k) { return 2.0 * B * Ho * Wo * Cout * k * k; }

// stride-1 depthwise conv: MFMA Toeplitz kernel when a table was packed for this layer, VALU kernels otherwise
int dw_s1(fv_handle* h, const bf16_t* x, const float* w, const bf16_t* ttab, const float* bias, bf16_t* y, int mb, int H, int C,
          int k, hipStream_t s) {
  if (ttab && !h->no_mfma_dw) return fv::launch_dwconv_mfma(x, ttab, bias, y, mb, H, H, C, k, 0, s);
  return fv::launch_dwconv(x, w, bias, y, mb, H, H, C, k, 1, 1, 0, s);
}

// fused ConvFFN pointwise half: the 32x32x16 kernel where it exists (C = 96 / 192 / 384), else the 16x16x32 one
// ranges: the inference tower lets launch_convffn32 cut the hidden units into ranges when the launch has few row tiles (B <= 2).  That changes the fp32
// summation order (<= 1 bf16 step per output), so the TRAINING forward keeps the one-launch form at every batch size: a row's gradient must not depend
// on how many rows share its step (tests/test_gpu_train_tower.py, B = 32 against B = 2).
// stash_y (the tower's training forward): the pre-activation / 4 written out by the one-launch 32x32x16 kernel (ffn_stash_ok says whether it can)
bool ffn_stash_ok(const fv_handle* h, const FFN& f, int M, int C) { return f.w2q && !h->no_ffn32 && fv::convffn32_stash_supported(M, C); }
int fused_ffn(fv_handle* h, const FFN& f, const bf16_t* t, const bf16_t* res, bf16_t* out, int M, int C, int hidden, hipStream_t s, bool ranges = true,
              bf16_t* stash_y = nullptr) {
  if (stash_y) return fv::launch_convffn32(t, f.w2q, f.fc1_b, f.fc2_b, f.ls, res, out, M, C, hidden, s, nullptr, 0, stash_y);
  if (f.w2q && !h->no_ffn32 && (size_t)M * C * 2 < ((size_t)1 << 31))
    return fv::launch_convffn32(t, f.w2q, f.fc1_b, f.fc2_b, f.ls, res, out, M, C, hidden, s, ranges && !h->batch_invariant ? h->ffn_part : nullptr,
                                ranges && !h->batch_invariant && h->ffn_part ? FFN_PART_BYTES : 0);
  return fv::launch_convffn(t, f.fc1_w, f.fc1_b, f.w2p, f.fc2_b, f.ls, res, out, M, C, hidden, s);
}

int run_ffn(fv_handle* h, const FFN& f, bf16_t* x_dw_in, bf16_t* dw_out, bf16_t* hid, bf16_t* res_out, int mb, int H, int W, int C,
            int ratio, hipStream_t s) {
  // dw_out = dw7x7(x_dw_in) (+BN folded); hid = gelu(fc1(dw_out)); res_out += ls * fc2(hid)
  const int M = mb * H * W;
  FV_P(FV_FAM_DWCONV, dw_flops(mb, H, W, C, 7), 4.0 * M * C, dw_s1(h, x_dw_in, f.dw_w, f.dw_t, f.dw_b, dw_out, mb, H, C, 7, s));
  if (f.w2p && !h->no_fused_ffn) {
    prof_begin(h, FV_FAM_GEMM, 4.0 * M * C * (double)(C * ratio), 6.0 * M * C, s, M, C, C * ratio, 6);
    const int rc = fused_ffn(h, f, dw_out, res_out, res_out, M, C, C * ratio, s);
    prof_end(h, s);
    return rc;
  }
  fv::GemmArgs g1{dw_out, C, f.fc1_w, M, C * ratio, C, f.fc1_b, nullptr, nullptr, 0, hid, C * ratio, FV_EPI_BIAS_GELU};
  FV_TRY(gemm_p(h, g1, s));
  fv::GemmArgs g2{hid, C * ratio, f.fc2_w, M, C, C * ratio, f.fc2_b, f.ls, res_out, C, res_out, C, FV_EPI_LS_RES};
  if (!h->batch_invariant) { g2.splitk_ws = h->ffn_part; g2.splitk_bytes = h->ffn_part ? FFN_PART_BYTES : 0; }   // few rows (B <= 2): launch_gemm cuts K = 4C into ranges (inference only)
  FV_TRY(gemm_p(h, g2, s));
  return FV_OK;
}

// parity-test hook: copy the activation map (NHWC bf16) of tap `t` for images [b0, b0 + mb) out of the ping-pong buffer
int tap_copy(fv_handle* h, int t, const bf16_t* src, int b0, int mb, size_t per_image, hipStream_t s) {
  if (!h->taps || t >= h->n_taps || !h->taps[t]) return FV_OK;
  bf16_t* dst = static_cast<bf16_t*>(h->taps[t]) + (size_t)b0 * per_image;
  FV_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)mb * per_image * 2, hipMemcpyDeviceToDevice, s));
  return FV_OK;
}

// the same per tower UNIT (fv_vision_unit_info order): unit u's output for images [b0, b0 + mb)
int unit_tap(fv_handle* h, int u, const bf16_t* src, int b0, int mb, size_t per_image, hipStream_t s) {
  if (!h->utaps || u >= h->n_utaps || !h->utaps[u]) return FV_OK;
  bf16_t* dst = static_cast<bf16_t*>(h->utaps[u]) + (size_t)b0 * per_image;
  FV_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)mb * per_image * 2, hipMemcpyDeviceToDevice, s));
  return FV_OK;
}

// tower units in execution order: 0 = stem (three convolutions), then per stage [RepCPE] block ... block [PatchEmbed]
struct UnitInfo { int kind, stage, side, channels; };   // kind: 0 stem, 1 RepCPE, 2 block, 3 PatchEmbed (side / channels of its OUTPUT)
std::vector<UnitInfo> tower_units(const fv_model_desc& d) {
  std::vector<UnitInfo> u;
  int H = d.image_size / 4;
  u.push_back({0, 0, H, d.tower_dims[0]});
  for (int i = 0; i < d.tower_stages; ++i) {
    if (d.tower_is_attn[i]) u.push_back({1, i, H, d.tower_dims[i]});
    for (int j = 0; j < d.tower_layers[i]; ++j) u.push_back({2, i, H, d.tower_dims[i]});
    if (i + 1 < d.tower_stages) { H /= 2; u.push_back({3, i, H, d.tower_dims[i + 1]}); }
  }
  return u;
}

int tower_pass(fv_handle* h, const bf16_t* pix, int b0, int mb, bf16_t* tower_out, float* img_tokens, const WsPlan& wp,
               hipStream_t s) {
  const fv_model_desc& d = h->d;
  const Tower& tw = h->tw;
  char* ws = static_cast<char*>(h->ws);
  bf16_t* cur = reinterpret_cast<bf16_t*>(ws + wp.bufA);
  bf16_t* oth = reinterpret_cast<bf16_t*>(ws + wp.bufB);
  bf16_t* hid = reinterpret_cast<bf16_t*>(ws + wp.bufH);
  float* se = reinterpret_cast<float*>(ws + wp.se);
  const int S = d.image_size, C0 = d.tower_dims[0];
  static const bool fuse_stem = !fv_ab_env("FASTVLA_NO_FUSED_STEM");
  if (h->lbsrc) {
    // letterbox + both stem convolutions in one kernel: neither the 1024^2 frame nor the half-resolution map exists in HBM
    const fv_handle::LbSrc& L = *h->lbsrc;
    if (!(tw.stem0_wp && fv::stem_fused_supported(S, C0))) return fv_fail(FV_ERR_UNSUPPORTED, "fv_vision_forward_images needs the fused stem (first stage width 96)");
    const size_t per = (size_t)L.C * L.Hin * L.Win * (L.dtype == FV_U8 ? 1 : 4);
    FV_P(FV_FAM_STEM, 2.0 * mb * (S / 2) * (S / 2) * 27 * C0 + dw_flops(mb, S / 4, S / 4, C0, 3) + 30.0 * mb * S * S,
         (double)mb * per + (double)mb * (S / 4) * (S / 4) * C0 * 2,
         fv::launch_stem_fused_lb(static_cast<const char*>(L.img) + (size_t)b0 * per, L.dtype, mb, L.C, L.Hin, L.Win, L.pad, L.letterbox, tw.stem0_wp,
                                  tw.stem0_b, tw.stem1_w, tw.stem1_b, oth, S, C0, s));
  } else if (fuse_stem && tw.stem0_wp && fv::stem_fused_supported(S, C0)) {
    // both convolutions in one kernel: the half-resolution 96-channel map (the path's largest tensor) never reaches HBM
    FV_P(FV_FAM_STEM, 2.0 * mb * (S / 2) * (S / 2) * 27 * C0 + dw_flops(mb, S / 4, S / 4, C0, 3),
         (double)mb * S * S * 8 + (double)mb * (S / 4) * (S / 4) * C0 * 2,
         fv::launch_stem_fused(pix, tw.stem0_wp, tw.stem0_b, tw.stem1_w, tw.stem1_b, oth, mb, S, C0, s));
  } else {
    FV_P(FV_FAM_STEM, 2.0 * mb * (S / 2) * (S / 2) * 27 * C0, (double)mb * S * S * 8 + (double)mb * (S / 2) * (S / 2) * C0 * 2,
         tw.stem0_wp ? fv::launch_stem_mfma(pix, tw.stem0_wp, tw.stem0_b, cur, mb, S, C0, s)
                     : fv::launch_stem_conv(pix, tw.stem0_w, tw.stem0_b, cur, mb, S, C0, s));
    FV_P(FV_FAM_DWCONV, dw_flops(mb, S / 4, S / 4, C0, 3), (double)mb * (S / 2) * (S / 2) * C0 * 2 * 1.25,
         fv::launch_dwconv(cur, tw.stem1_w, tw.stem1_b, oth, mb, S / 2, S / 2, C0, 3, 2, 1, 1, s));
  }
  int H = S / 4;
  {
    fv::GemmArgs g{oth, C0, tw.stem2_w, mb * H * H, C0, C0, tw.stem2_b, nullptr, nullptr, 0, cur, C0, FV_EPI_BIAS_GELU};
    FV_TRY(gemm_p(h, g, s));
  }
  FV_TRY(tap_copy(h, 0, cur, b0, mb, (size_t)H * H * C0, s));
  int unit = 0;
  FV_TRY(unit_tap(h, unit++, cur, b0, mb, (size_t)H * H * C0, s));
  for (int i = 0; i < d.tower_stages; ++i) {
    const int C = d.tower_dims[i];
    const int M = mb * H * H;
    if (d.tower_is_attn[i]) {
      FV_P(FV_FAM_DWCONV, dw_flops(mb, H, H, C, 7), 4.0 * M * C, dw_s1(h, cur, tw.cpes[i].w, tw.cpes[i].t, tw.cpes[i].b, oth, mb, H, C, 7, s));
      std::swap(cur, oth);
      FV_TRY(unit_tap(h, unit++, cur, b0, mb, (size_t)H * H * C, s));
    }
    for (const Block& b : tw.stages[i]) {
      if (!d.tower_is_attn[i]) {
        static const bool pair_ok = !fv_ab_env("FASTVLA_NO_DW_PAIR");
        if (pair_ok && !h->no_mfma_dw && !h->no_fused_ffn && b.mix_t && b.ffn.dw_t && b.ffn.w2p && fv::dwconv_pair_supported(mb, H, H, C)) {
          // token mixer and the ConvFFN's 7x7 in one marching kernel: cur -> oth (x') and hid (t); then oth += ls * ffn(t)
          FV_P(FV_FAM_DWCONV, dw_flops(mb, H, H, C, 3) + dw_flops(mb, H, H, C, 7), 6.0 * M * C,
               fv::launch_dwconv_pair(cur, b.mix_t, b.mix_b, b.ffn.dw_t, b.ffn.dw_b, oth, hid, mb, H, H, C, s));
          prof_begin(h, FV_FAM_GEMM, 4.0 * M * C * (double)(C * d.tower_mlp_ratio), 6.0 * M * C, s, M, C, C * d.tower_mlp_ratio, 6);
          const int rc = fused_ffn(h, b.ffn, hid, oth, oth, M, C, C * d.tower_mlp_ratio, s);
          prof_end(h, s);
          if (rc != FV_OK) return rc;
        } else {
          FV_P(FV_FAM_DWCONV, dw_flops(mb, H, H, C, 3), 4.0 * M * C, dw_s1(h, cur, b.mix_w, b.mix_t, b.mix_b, oth, mb, H, C, 3, s));  // x = RepMixer(x) -> oth
          FV_TRY(run_ffn(h, b.ffn, oth, cur, hid, oth, mb, H, H, C, d.tower_mlp_ratio, s));        // oth += ls * ffn(oth)
        }
        std::swap(cur, oth);
      } else {
        FV_P(FV_FAM_NORM, 8.0 * M * C, 4.0 * M * C, fv::launch_layernorm_rows(cur, b.ln_w, b.ln_b, oth, M, C, d.ln_eps, s));
        fv::GemmArgs gq{oth, C, b.qkv_w, M, 3 * C, C, nullptr, nullptr, nullptr, 0, hid, 3 * C, FV_EPI_BIAS};
        FV_TRY(gemm_p(h, gq, s));
        const int nh = C / d.tower_head_dim;
        FV_P(FV_FAM_ATTN, 4.0 * mb * (double)(H * H) * (H * H) * C, 8.0 * M * C,
             fv::launch_attention(hid, hid + C, hid + 2 * C, 3 * C, 3 * C, 3 * C, oth, C, mb, H * H, nh, nh,
                                  d.tower_head_dim, 0, nullptr, 0, 1.0f / std::sqrt((float)d.tower_head_dim), s));
        fv::GemmArgs gp{oth, C, b.proj_w, M, C, C, b.proj_b, b.ls1, cur, C, cur, C, FV_EPI_LS_RES};
        FV_TRY(gemm_p(h, gp, s));
        FV_TRY(run_ffn(h, b.ffn, cur, oth, hid, cur, mb, H, H, C, d.tower_mlp_ratio, s));
      }
      FV_TRY(unit_tap(h, unit++, cur, b0, mb, (size_t)H * H * C, s));
    }
    FV_TRY(tap_copy(h, 1 + i, cur, b0, mb, (size_t)H * H * C, s));
    if (i + 1 < d.tower_stages) {
      const int C2 = d.tower_dims[i + 1];
      FV_P(FV_FAM_DWCONV, dw_flops(mb, H / 2, H / 2, C2, 7), 2.0 * M * C + 2.0 * (M / 4) * C2,
           (tw.downs[i].lk_t && !h->no_mfma_dw)
               ? fv::launch_dwconv_s2_mfma(cur, tw.downs[i].lk_t, tw.downs[i].lk_b, oth, mb, H, H, C, 1, s)
               : fv::launch_dwconv(cur, tw.downs[i].lk_w, tw.downs[i].lk_b, oth, mb, H, H, C, 7, 2, C2 / C, 1, s));
      H /= 2;
      fv::GemmArgs g{oth, C2, tw.downs[i].pw_w, mb * H * H, C2, C2, tw.downs[i].pw_b, nullptr, nullptr, 0, cur, C2, FV_EPI_BIAS_GELU};
      FV_TRY(gemm_p(h, g, s));
      FV_TRY(unit_tap(h, unit++, cur, b0, mb, (size_t)H * H * C2, s));
    }
  }
  const int CL = d.tower_dims[d.tower_stages - 1], CO = d.tower_out_dim, P = H * H;
  FV_P(FV_FAM_DWCONV, dw_flops(mb, H, H, CO, 3), 2.0 * mb * P * (CL + CO), fv::launch_dwconv(cur, tw.exp_w, tw.exp_b, oth, mb, H, H, CL, 3, 1, CO / CL, 0, s));
  FV_P(FV_FAM_ELT, 20.0 * mb * P * CO, 6.0 * mb * P * CO, fv::launch_se_gelu(oth, tw.se_w1, tw.se_b1, tw.se_w2, tw.se_b2, tower_out, se, mb, P, CO, d.tower_se_rd, s));
  // mm_projector: Linear + GELU + Linear -> fp32 tokens ([site] fast_vlm/modeling_fast_vlm.py:51-55)
  fv::GemmArgs p0{tower_out, CO, tw.pj0_w, mb * P, d.llm_hidden, CO, tw.pj0_b, nullptr, nullptr, 0, hid, d.llm_hidden, FV_EPI_BIAS_GELU};
  if (!h->batch_invariant) { p0.splitk_ws = h->ffn_part; p0.splitk_bytes = h->ffn_part ? FFN_PART_BYTES : 0; }
  FV_TRY(gemm_p(h, p0, s));
  fv::GemmArgs p2{hid, d.llm_hidden, tw.pj2_w, mb * P, d.llm_hidden, d.llm_hidden, tw.pj2_b, nullptr, nullptr, 0, img_tokens, d.llm_hidden, FV_EPI_F32};
  FV_TRY(gemm_p(h, p2, s));
  return FV_OK;
}


// The parity-mode decoder stack (llm_precision 1 / 2) over rows = B * Tq residual rows already in ws.x.
//   DEC_FULL   : the whole sequence in one pass (Tq = every position; len_add = Ni image positions in front of the text)
//   DEC_PREFIX : the image-token prefix alone (Tq = Np positions): every layer's UN-rotated [k | v] rows go to kv
//                ([layer][B * Np][2 * kv_heads * D] fp32) and the stack stops after the last layer's projections
//   DEC_SUFFIX : Tq = the text positions only, attending to kv's cached prefix (Np positions) + themselves
// (the 256 image tokens come first in the causal sequence, so their K / V depend on the image alone.)
enum DecMode { DEC_FULL = 0, DEC_PREFIX = 1, DEC_SUFFIX = 2 };
int decoder_layers_split(fv_handle* h, const WsPlan& wp, int B, int Tq, const int32_t* lens, int len_add, float* kv, int Np, int mode,
                         hipStream_t s) {
  const fv_model_desc& d = h->d;
  char* ws = static_cast<char*>(h->ws);
  float* x = reinterpret_cast<float*>(ws + wp.x);
  bf16_t* xn = reinterpret_cast<bf16_t*>(ws + wp.xn);
  bf16_t* act = reinterpret_cast<bf16_t*>(ws + wp.act);
  const int rows = B * Tq, Hd = d.llm_hidden, D = d.llm_head_dim;
  const int qd = d.llm_heads * D, kd = d.llm_kv_heads * D, qkvw = qd + 2 * kd;
  const int Tt = mode == DEC_SUFFIX ? Np + Tq : Tq;      // positions the attention sees
  const float att_scale = 1.0f / std::sqrt((float)D);
  {
    // split-bf16 activations: every GEMM operand x is carried as hi + lo (16 significant bits) and multiplied in two
    // MFMA passes against the exact bf16 weights; qkv / gate-up accumulators and attention stay fp32.
    // hi and lo halves sit side by side ([rows][2K]) so each projection is ONE launch with a doubled K loop (ksplit)
    bf16_t* xs = reinterpret_cast<bf16_t*>(ws + wp.xn_lo);    // [rows][2*Hd]
    bf16_t* as = reinterpret_cast<bf16_t*>(ws + wp.att_lo);   // [rows][2*qd]
    bf16_t* cs = reinterpret_cast<bf16_t*>(ws + wp.act_lo);   // [rows][2*I]
    float* qkvf = reinterpret_cast<float*>(ws + wp.qkvf);
    const int I = d.llm_inter, I2 = 2 * I;
    const size_t kv_layer = (size_t)B * Np * 2 * kd;          // floats per layer of the prefix cache
    for (size_t li = 0; li < h->dec.layers.size(); ++li) {
      const DecLayer& L = h->dec.layers[li];
      // layer 0 norms its own input; every later layer's input_layernorm output arrives from the previous layer's down
      // projection (fused into its split-K reducer)
      // llm_precision = 5: every split operand is "hi + lo8" (bf16 + one fp8 byte of remainder per element, in the lo half's place) and
      // every projection takes ksplit = 2: the lo product on the scaled fp8 MFMA against the weights' fp8 copies -- 1.5 passes
      const int lo8 = d.llm_precision == 5, KS = lo8 ? 2 : 1;
      if (li == 0) FV_P(FV_FAM_NORM, 4.0 * rows * Hd, 8.0 * rows * Hd, fv::launch_rmsnorm(x, L.ln1, xs, xs + Hd, 2 * Hd, rows, Hd, d.rms_eps, s, 0, nullptr, lo8));
      fv::GemmArgs q1{xs, 2 * Hd, L.qkv_w, rows, qkvw, Hd, L.qkv_b, nullptr, nullptr, 0, qkvf, qkvw, FV_EPI_F32, KS};
      q1.W8 = L.qkv_w8;
      // few output tiles and a long K (7B: 72 tiles of 256 x 256 at M = 1024): the 256-tile kernel cut along K, as the down projection
      q1.splitk_ws = wp.splitk_bytes ? reinterpret_cast<float*>(ws + wp.splitk) : nullptr;
      q1.splitk_bytes = wp.splitk_bytes;
      FV_TRY(gemm_p(h, q1, s));
      if (mode == DEC_PREFIX) {   // this layer's [k | v] rows of the prefix, un-rotated (RoPE rides inside the attention kernel)
        FV_HIP_CHECK(hipMemcpy2DAsync(kv + li * kv_layer, (size_t)2 * kd * 4, qkvf + qd, (size_t)qkvw * 4, (size_t)2 * kd * 4, (size_t)rows,
                                      hipMemcpyDeviceToDevice, s));
        if (li + 1 == h->dec.layers.size()) break;   // nothing downstream of the last layer's K / V is ever read
      }
      // RoPE rides inside the attention kernel (q fragments in registers, K rows on their way into LDS)
      FV_P(FV_FAM_ATTN, 2.0 * B * (double)Tq * Tt * qd + 3.0 * rows * (qd + kd), 4.0 * rows * (qkvw + qd),
           fv::launch_attention_f32(qkvf, qkvw, as, as + qd, 2 * qd, B, Tt, d.llm_heads, d.llm_kv_heads, D, lens, len_add, att_scale, s, h->rope,
                                    mode == DEC_SUFFIX ? kv + li * kv_layer : nullptr, 2 * kd, mode == DEC_SUFFIX ? Np : 0, nullptr, lo8,
                                    wp.attn_scr_bytes && mode != DEC_SUFFIX ? ws + wp.attn_scr : nullptr));
      fv::GemmArgs o1{as, 2 * qd, L.o_w, rows, Hd, qd, nullptr, nullptr, x, Hd, x, Hd, FV_EPI_RES_F32, KS};
      o1.W8 = L.o_w8;
      o1.splitk_ws = q1.splitk_ws;
      o1.splitk_bytes = wp.splitk_bytes;
      // policies 1 / 5: post_attention_layernorm rides on the output projection (its split-K reducer where there is one, launch_gemm's own norm launch
      // otherwise), as input_layernorm rides on the down projection
      const bool ln2_fused = d.llm_precision == 1 || d.llm_precision == 5;
      if (ln2_fused) { o1.norm_w = L.ln2; o1.norm_y = xs; o1.norm_ylo = xs + Hd; o1.norm_ld = 2 * Hd; o1.norm_eps = d.rms_eps; }
      FV_TRY(gemm_p(h, o1, s));
      fv::GemmArgs d1{cs, I2, L.down_w, rows, Hd, I, nullptr, nullptr, x, Hd, x, Hd, FV_EPI_RES_F32, KS};
      d1.W8 = L.down_w8;
      if (d.llm_precision == 2) {
        // the MLP in ONE pass on fp16 operands (tests/precision_budget.py): post-norm rows as fp16, SwiGLU output / 16 as fp16
        FV_P(FV_FAM_NORM, 4.0 * rows * Hd, 6.0 * rows * Hd, fv::launch_rmsnorm(x, L.ln2, xn, nullptr, Hd, rows, Hd, d.rms_eps, s, 1, h->f16_flags));
        fv::GemmArgs g1{xn, Hd, L.gu_w, rows, I2, Hd, nullptr, nullptr, nullptr, 0, act, I, FV_EPI_SWIGLU_F16, 0};
        g1.f16 = 1; g1.sat = h->f16_flags;
        FV_TRY(gemm_p(h, g1, s));
        d1 = fv::GemmArgs{act, I, L.down_w, rows, Hd, I, nullptr, nullptr, x, Hd, x, Hd, FV_EPI_RES_F32, 0};
        d1.f16 = 1;
      } else if (d.llm_precision == 3) {
        // gate/up alone in one fp16 pass; its SwiGLU output leaves as hi + lo bf16 for a split-bf16 down projection
        FV_P(FV_FAM_NORM, 4.0 * rows * Hd, 6.0 * rows * Hd, fv::launch_rmsnorm(x, L.ln2, xn, nullptr, Hd, rows, Hd, d.rms_eps, s, 1, h->f16_flags));
        fv::GemmArgs g1{xn, Hd, L.gu_w, rows, I2, Hd, nullptr, nullptr, nullptr, 0, cs, I2, FV_EPI_SWIGLU_SPLIT, 0};
        g1.f16 = 1;
        FV_TRY(gemm_p(h, g1, s));
      } else if (d.llm_precision == 4) {
        // down alone in one fp16 pass: split-bf16 gate/up whose SwiGLU output / 16 leaves as fp16
        FV_P(FV_FAM_NORM, 4.0 * rows * Hd, 8.0 * rows * Hd, fv::launch_rmsnorm(x, L.ln2, xs, xs + Hd, 2 * Hd, rows, Hd, d.rms_eps, s));
        fv::GemmArgs g1{xs, 2 * Hd, L.gu_w, rows, I2, Hd, nullptr, nullptr, nullptr, 0, act, I, FV_EPI_SWIGLU_F16, 1};
        g1.sat = h->f16_flags;
        FV_TRY(gemm_p(h, g1, s));
        d1 = fv::GemmArgs{act, I, L.down_w, rows, Hd, I, nullptr, nullptr, x, Hd, x, Hd, FV_EPI_RES_F32, 0};
        d1.f16 = 1;
      } else {
        fv::GemmArgs g1{xs, 2 * Hd, L.gu_w, rows, I2, Hd, nullptr, nullptr, nullptr, 0, cs, I2, FV_EPI_SWIGLU_SPLIT, KS};
        g1.W8 = L.gu_w8;
        g1.splitk_ws = q1.splitk_ws; g1.splitk_bytes = wp.splitk_bytes;   // few rows: K ranges + a SwiGLU reduce (launch_gemm's few-row rule)
        FV_TRY(gemm_p(h, g1, s));  // silu(gate)*up and its hi / lo (or lo8) split happen in the epilogue: no fp32 round trip
      }
      d1.splitk_ws = wp.splitk_bytes ? reinterpret_cast<float*>(ws + wp.splitk) : nullptr;
      d1.splitk_bytes = wp.splitk_bytes;
      if (li + 1 < h->dec.layers.size()) {
        d1.norm_w = h->dec.layers[li + 1].ln1; d1.norm_y = xs; d1.norm_ylo = xs + Hd; d1.norm_ld = 2 * Hd; d1.norm_eps = d.rms_eps;
      }
      FV_TRY(gemm_p(h, d1, s));
    }
  }
  return FV_OK;
}

}  // namespace

// =============================================================================================== C ABI
extern "C" {

const char* fv_version(void) { return "fastvla_hip 0.1 (gfx950)"; }
const char* fv_last_error(fv_handle* h) { return h ? h->err : g_err; }

int fv_create(const fv_model_desc* desc, int device, fv_handle** out) {
  if (!desc || !out) return fv_fail(FV_ERR_ARG, "fv_create: null argument");
  const fv_model_desc& d = *desc;
  if (d.tower_stages < 1 || d.tower_stages > FV_MAX_STAGES) return fv_fail(FV_ERR_ARG, "tower_stages out of range");
  if (d.image_size <= 0 || d.image_size % (4 << (d.tower_stages - 1)))
    return fv_fail(FV_ERR_ARG, "image_size %d must be a positive multiple of %d", d.image_size, 4 << (d.tower_stages - 1));
  for (int i = 0; i < d.tower_stages; ++i) {
    if (d.tower_dims[i] % 8 || d.tower_dims[i] <= 0 || d.tower_layers[i] < 0) return fv_fail(FV_ERR_ARG, "tower dims must be positive multiples of 8");
    if (d.tower_is_attn[i] && d.tower_dims[i] % d.tower_head_dim) return fv_fail(FV_ERR_ARG, "attention stage dim %% head_dim != 0");
    if (i > 0 && d.tower_dims[i] != 2 * d.tower_dims[i - 1] && d.tower_dims[i] != d.tower_dims[i - 1]) return fv_fail(FV_ERR_UNSUPPORTED, "stage dims must double or stay (grouped 7x7 multiplier 1 or 2)");
  }
  if (d.tower_head_dim != 32 && d.tower_head_dim != 64 && d.tower_head_dim != 128) return fv_fail(FV_ERR_UNSUPPORTED, "tower head_dim must be 32/64/128");
  if (d.llm_head_dim != 32 && d.llm_head_dim != 64 && d.llm_head_dim != 128) return fv_fail(FV_ERR_UNSUPPORTED, "llm head_dim must be 32/64/128");
  if (d.llm_hidden % 8 || d.llm_inter % 8 || d.llm_heads % d.llm_kv_heads) return fv_fail(FV_ERR_ARG, "llm dims must be multiples of 8 and heads %% kv_heads == 0");
  if (d.tower_out_dim != 2 * d.tower_dims[d.tower_stages - 1] && d.tower_out_dim != d.tower_dims[d.tower_stages - 1]) return fv_fail(FV_ERR_UNSUPPORTED, "tower_out_dim must be 1x or 2x the last stage dim");
  if (d.llm_precision < 0 || d.llm_precision > 5)
    return fv_fail(FV_ERR_ARG, "llm_precision must be 0 (bf16), 1 (split-bf16), 2 (split-bf16 qkv/o + fp16 gate/up/down) or 5 (bf16 hi + fp8 lo)");
#ifndef FASTVLA_AB_SWITCHES
  // 3 (fp16 gate/up only) and 4 (fp16 down only) exist for tools/prec_sweep.py's per-family error budget: the tools build (make AB=1) only
  if (d.llm_precision == 3 || d.llm_precision == 4)
    return fv_fail(FV_ERR_UNSUPPORTED, "llm_precision = %d is a measurement mode of the tools build (make AB=1); the product ships 0, 1, 2 and 5", d.llm_precision);
#endif
  if (d.llm_precision == 5 && (d.llm_hidden % 128 || d.llm_inter % 128 || (d.llm_heads * d.llm_head_dim) % 128))
    return fv_fail(FV_ERR_UNSUPPORTED, "llm_precision = 5 (hi + lo8 operands) needs hidden, inter and heads * head_dim to be multiples of 128");
  if (d.state_dim <= 0 || d.action_dim <= 0 || d.hidden_dim <= 0 || d.fusion_dim <= 0) return fv_fail(FV_ERR_ARG, "head dims must be positive");
  FV_HIP_CHECK(hipSetDevice(device));
  fv_handle* h = new fv_handle();
  h->d = d;
  h->device = device;
  h->hd = fv::HeadDims{d.llm_hidden, d.state_dim, d.action_dim, d.hidden_dim, d.fusion_dim};
  if (const char* e = fv_ab_env("FASTVLA_NO_FUSED_FFN")) h->no_fused_ffn = e[0] == '1';
  if (const char* e = fv_ab_env("FASTVLA_NO_MFMA_DW")) h->no_mfma_dw = e[0] == '1';
  if (const char* e = fv_ab_env("FASTVLA_NO_FFN32")) h->no_ffn32 = e[0] == '1';
  // RoPE table for every position the path can see (text + spliced image tokens)
  const int P = (d.image_size >> (d.tower_stages + 1)) * (d.image_size >> (d.tower_stages + 1));
  h->rope_rows = d.max_text_tokens + P + 8;
  std::vector<float> cs((size_t)h->rope_rows * d.llm_head_dim);
  fv::rope_table_host(cs.data(), h->rope_rows, d.llm_head_dim, d.rope_theta);
  void* p = nullptr;
  int rc = dev_alloc(h, cs.size() * 4, &p);
  if (rc == FV_OK && hipMemcpy(p, cs.data(), cs.size() * 4, hipMemcpyHostToDevice) != hipSuccess) rc = fv_fail(FV_ERR_HIP, "rope table upload failed");
  h->rope = static_cast<float2*>(p);
  if (rc == FV_OK) { rc = dev_alloc(h, fv::adamw_scratch_bytes(), &p); h->norm_scratch = static_cast<float*>(p); }
  if (rc == FV_OK) {
    rc = dev_alloc(h, 256, &p);
    h->f16_flags = static_cast<unsigned*>(p);
    if (rc == FV_OK && hipMemset(p, 0, 256) != hipSuccess) rc = fv_fail(FV_ERR_HIP, "fp16 flag words: memset failed");
    h->lb_vmax = h->f16_flags + 32;   // a word of the same block (nothing is allocated inside fv_preprocess_normalized: the call stays capturable)
  }
  if (rc == FV_OK) { rc = dev_alloc(h, FFN_PART_BYTES, &p); h->ffn_part = static_cast<float*>(p); }
  if (rc != FV_OK) { fv_destroy(h); return rc; }
  *out = h;
  return FV_OK;
}

void fv_destroy(fv_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  for (void* p : h->allocs) (void)hipFree(p);
  for (hipEvent_t e : h->prof.pool) (void)hipEventDestroy(e);
  delete h;
}

static int load_impl(fv_handle* h, Loader& L);

int fv_load_weights(fv_handle* h, const fv_tensor_desc* tensors, int n) {
  HandleScope _hs(h);
  if (!h || !tensors || n <= 0) return fv_fail(FV_ERR_ARG, "fv_load_weights: null argument");
  if (h->loaded) return fv_fail(FV_ERR_STATE, "weights already loaded");
  Loader L{h};
  for (int i = 0; i < n; ++i) {
    if (!tensors[i].name || !tensors[i].data || tensors[i].ndim < 1 || tensors[i].ndim > 4) return fv_fail(FV_ERR_ARG, "tensor %d: bad descriptor", i);
    L.idx[tensors[i].name] = &tensors[i];
  }
  return load_impl(h, L);
}

int fv_load_weights_cb(fv_handle* h, fv_tensor_provider provider, void* user) {
  HandleScope _hs(h);
  if (!h || !provider) return fv_fail(FV_ERR_ARG, "fv_load_weights_cb: null argument");
  if (h->loaded) return fv_fail(FV_ERR_STATE, "weights already loaded");
  Loader L{h};
  L.prov = provider;
  L.user = user;
  return load_impl(h, L);
}

static int load_impl(fv_handle* h, Loader& L) {
  FV_HIP_CHECK(hipSetDevice(h->device));
  const fv_model_desc& d = h->d;
  Tower& tw = h->tw;
  const std::string vt = VT;
  const int C0 = d.tower_dims[0];
  {  // stem: [C0,3,3,3] -> [27][C0] with k = (ky*3+kx)*3+ci
    std::vector<float> v;
    if (L.expect(vt + "patch_embed.0.reparam_conv.weight", v, (size_t)C0 * 27)) {
      std::vector<float> o((size_t)27 * C0);
      for (int co = 0; co < C0; ++co)
        for (int ci = 0; ci < 3; ++ci)
          for (int t = 0; t < 9; ++t) o[(size_t)(t * 3 + ci) * C0 + co] = v[((size_t)co * 3 + ci) * 9 + t];
      tw.stem0_w = L.up_f32(o);
      if (C0 % 16 == 0 && C0 <= 128) {
        std::vector<float> wp((size_t)C0 * 64);
        fv::stem_mfma_pack(o.data(), wp.data(), C0);
        tw.stem0_wp = L.up_bf16(wp);
      }
    }
    tw.stem0_b = L.vec(vt + "patch_embed.0.reparam_conv.bias", C0);
    tw.stem1_w = L.dw(vt + "patch_embed.1.reparam_conv.weight", C0, 3);
    tw.stem1_b = L.vec(vt + "patch_embed.1.reparam_conv.bias", C0);
    tw.stem2_w = L.mat(vt + "patch_embed.2.reparam_conv.weight", C0, C0);
    tw.stem2_b = L.vec(vt + "patch_embed.2.reparam_conv.bias", C0);
  }
  tw.stages.resize(d.tower_stages);
  tw.downs.resize(d.tower_stages);
  tw.cpes.resize(d.tower_stages);
  int idx = 0;  // index into mci.py's `network` ModuleList: [RepCPE?] stage [PatchEmbed]
  for (int i = 0; i < d.tower_stages && L.rc == FV_OK; ++i) {
    const int C = d.tower_dims[i];
    const int map_w = d.image_size >> (2 + i);  // feature-map side at this stage
    if (d.tower_is_attn[i]) {
      const std::string pre = vt + "network." + std::to_string(idx++) + ".";
      tw.cpes[i].w = L.dw(pre + "reparam_conv.weight", C, 7, nullptr, map_w, &tw.cpes[i].t);
      tw.cpes[i].b = L.vec(pre + "reparam_conv.bias", C);
    }
    const int sidx = idx++;
    tw.stages[i].resize(d.tower_layers[i]);
    for (int j = 0; j < d.tower_layers[i] && L.rc == FV_OK; ++j) {
      const std::string pre = vt + "network." + std::to_string(sidx) + "." + std::to_string(j) + ".";
      Block& b = tw.stages[i][j];
      if (d.tower_is_attn[i]) {
        b.ln_w = L.vec(pre + "norm.weight", C);
        b.ln_b = L.vec(pre + "norm.bias", C);
        b.qkv_w = L.mat(pre + "token_mixer.qkv.weight", 3 * (size_t)C, C);
        b.proj_w = L.mat(pre + "token_mixer.proj.weight", C, C);
        b.proj_b = L.vec(pre + "token_mixer.proj.bias", C);
        b.ls1 = L.vec(pre + "layer_scale_1", C);
        load_ffn(L, pre + "convffn.", C, C * d.tower_mlp_ratio, d.bn_eps, b.ffn, pre + "layer_scale_2", map_w);
      } else {
        b.mix_w = L.dw(pre + "token_mixer.reparam_conv.weight", C, 3, nullptr, map_w, &b.mix_t);
        b.mix_b = L.vec(pre + "token_mixer.reparam_conv.bias", C);
        load_ffn(L, pre + "convffn.", C, C * d.tower_mlp_ratio, d.bn_eps, b.ffn, pre + "layer_scale", map_w);
      }
    }
    if (i + 1 < d.tower_stages) {
      const int C2 = d.tower_dims[i + 1];
      const std::string pre = vt + "network." + std::to_string(idx++) + ".proj.";
      tw.downs[i].lk_w = L.dw(pre + "0.lkb_reparam.weight", C2, 7, nullptr, 0, nullptr, C, map_w, &tw.downs[i].lk_t);
      tw.downs[i].lk_b = L.vec(pre + "0.lkb_reparam.bias", C2);
      tw.downs[i].pw_w = L.mat(pre + "1.reparam_conv.weight", C2, C2);
      tw.downs[i].pw_b = L.vec(pre + "1.reparam_conv.bias", C2);
    }
  }
  const int CO = d.tower_out_dim, RD = d.tower_se_rd;
  tw.exp_w = L.dw(vt + "conv_exp.reparam_conv.weight", CO, 3);
  tw.exp_b = L.vec(vt + "conv_exp.reparam_conv.bias", CO);
  tw.se_w1 = L.vec(vt + "conv_exp.se.reduce.weight", (size_t)RD * CO);
  tw.se_b1 = L.vec(vt + "conv_exp.se.reduce.bias", RD);
  tw.se_w2 = L.vec(vt + "conv_exp.se.expand.weight", (size_t)CO * RD);
  tw.se_b2 = L.vec(vt + "conv_exp.se.expand.bias", CO);
  const std::string pj = PJ;
  tw.pj0_w = L.mat(pj + "0.weight", d.llm_hidden, CO);
  tw.pj0_b = L.vec(pj + "0.bias", d.llm_hidden);
  tw.pj2_w = L.mat(pj + "2.weight", d.llm_hidden, d.llm_hidden);
  tw.pj2_b = L.vec(pj + "2.bias", d.llm_hidden);

  // ---- decoder
  const std::string lm = LM;
  const size_t Hd = d.llm_hidden, I = d.llm_inter;
  const size_t qd = (size_t)d.llm_heads * d.llm_head_dim, kd = (size_t)d.llm_kv_heads * d.llm_head_dim;
  h->dec.embed = L.mat(lm + "embed_tokens.weight", d.llm_vocab, Hd);
  h->dec.norm = L.vec(lm + "norm.weight", Hd);
  h->dec.layers.resize(d.llm_layers);
  for (int l = 0; l < d.llm_layers && L.rc == FV_OK; ++l) {
    const std::string pre = lm + "layers." + std::to_string(l) + ".";
    DecLayer& y = h->dec.layers[l];
    y.ln1 = L.vec(pre + "input_layernorm.weight", Hd);
    y.ln2 = L.vec(pre + "post_attention_layernorm.weight", Hd);
    {  // q | k | v rows concatenated into one [qd + 2 kd][Hd] matrix, biases likewise
      y.qkv_w = static_cast<bf16_t*>(L.alloc((qd + 2 * kd) * Hd * 2));
      const char* nm[3] = {"q_proj", "k_proj", "v_proj"};
      const size_t rows[3] = {qd, kd, kd};
      size_t r0 = 0;
      std::vector<float> qb, part;
      for (int j = 0; j < 3 && L.rc == FV_OK; ++j) {
        const fv_tensor_desc* t = L.need_n(pre + "self_attn." + nm[j] + ".weight", rows[j] * Hd);
        if (t && y.qkv_w) L.put_rows(t, y.qkv_w + r0 * Hd, rows[j] * Hd, rows[j] * Hd, 1, rows[j] * Hd);
        r0 += rows[j];
        if (L.expect(pre + "self_attn." + nm[j] + ".bias", part, rows[j])) qb.insert(qb.end(), part.begin(), part.end());
      }
      if (L.rc == FV_OK) y.qkv_b = L.up_f32(qb);
    }
    y.o_w = L.mat(pre + "self_attn.o_proj.weight", Hd, qd);
    {  // gate / up rows interleaved [8 gate | 8 up] so SwiGLU is a GEMM epilogue
      y.gu_w = static_cast<bf16_t*>(L.alloc(2 * I * Hd * 2));
      const fv_tensor_desc* g = L.need_n(pre + "mlp.gate_proj.weight", I * Hd);
      if (g && y.gu_w) L.put_rows(g, y.gu_w, 16 * Hd, 8 * Hd, I / 8, 8 * Hd);
      const fv_tensor_desc* u = L.need_n(pre + "mlp.up_proj.weight", I * Hd);
      if (u && y.gu_w) L.put_rows(u, y.gu_w + 8 * Hd, 16 * Hd, 8 * Hd, I / 8, 8 * Hd);
    }
    y.down_w = L.mat(pre + "mlp.down_proj.weight", Hd, I);
    if (d.llm_precision == 5 && L.rc == FV_OK) {
      // fp8 e4m3 copies (x 2^6) of the four projection matrices for the lo8 products; |w| x 64 must stay inside e4m3's 448
      struct { bf16_t* w; void** w8; size_t rows; int K; } m4[4] = {{y.qkv_w, &y.qkv_w8, qd + 2 * kd, (int)Hd}, {y.o_w, &y.o_w8, Hd, (int)qd},
                                                                  {y.gu_w, &y.gu_w8, 2 * I, (int)Hd}, {y.down_w, &y.down_w8, Hd, (int)I}};
      for (auto& t : m4) {
        *t.w8 = L.alloc(t.rows * 2 * t.K);
        if (!*t.w8 || fv::launch_bf16_to_w8(t.w, *t.w8, t.rows, t.K, nullptr, h->f16_flags + 2) != FV_OK) { L.rc = FV_ERR_HIP; break; }
      }
    } else if (d.llm_precision >= 2 && L.rc == FV_OK) {
      // fp16 copies IN PLACE of the two projections that run on fp16 operands: exact for |w| >= 6.1e-5 (smaller weights become
      // fp16 subnormals, absolute error <= 3e-8); down carries the 2^4 that its operand (FV_EPI_SWIGLU_F16) gives up
      if ((d.llm_precision != 4 && fv::launch_bf16_to_f16(y.gu_w, 2 * I * Hd, 1.0f, nullptr, h->f16_flags + 1) != FV_OK) ||
          (d.llm_precision != 3 && fv::launch_bf16_to_f16(y.down_w, Hd * I, 16.0f, nullptr, h->f16_flags + 1) != FV_OK))
        L.rc = FV_ERR_HIP;
    }
  }
  if (L.rc != FV_OK) return L.rc;
  FV_HIP_CHECK(hipDeviceSynchronize());
  if (d.llm_precision == 5) {
    unsigned bits = 0;
    FV_HIP_CHECK(hipMemcpy(&bits, h->f16_flags + 2, 4, hipMemcpyDeviceToHost));
    float mx;
    memcpy(&mx, &bits, 4);
    if (!(mx <= 448.0f))
      return fv_fail(FV_ERR_UNSUPPORTED, "llm_precision=5: a projection weight leaves the fp8 range of its lo-product copy (max |w| x 64 = %g > 448); "
                     "load this checkpoint with llm_precision=1", (double)mx);
  } else if (d.llm_precision >= 2) {
    // LOUD refusal instead of a silently clamped weight: the fp16 single-pass projections need every (scaled) weight inside the
    // binary16 range.  The host side falls back to llm_precision = 1 (split-bf16, no range limit) on this error.
    unsigned bits = 0;
    FV_HIP_CHECK(hipMemcpy(&bits, h->f16_flags + 1, 4, hipMemcpyDeviceToHost));
    float mx;
    memcpy(&mx, &bits, 4);
    if (!(mx <= 65504.0f))
      return fv_fail(FV_ERR_UNSUPPORTED, "llm_precision=%d: a gate/up/down weight leaves the fp16 range (max |w| x scale = %g, down carries x16); "
                     "load this checkpoint with llm_precision=1", d.llm_precision, (double)mx);
  }
  h->loaded = true;
  return FV_OK;
}

int fv_workspace_bytes(fv_handle* h, int B, int T, int splice, size_t* out_bytes) {
  HandleScope _hs(h);
  if (!h || !out_bytes) return fv_fail(FV_ERR_ARG, "fv_workspace_bytes: null argument");
  if (B <= 0 || T <= 0) return fv_fail(FV_ERR_ARG, "fv_workspace_bytes: B and T must be positive");
  *out_bytes = plan_ws(h, B, T, splice).total;
  return FV_OK;
}

int fv_bind_workspace(fv_handle* h, void* ws, size_t bytes) {
  HandleScope _hs(h);
  if (!h || !ws) return fv_fail(FV_ERR_ARG, "fv_bind_workspace: null argument");
  if ((uintptr_t)ws & 255) return fv_fail(FV_ERR_ARG, "workspace must be 256-byte aligned");
  h->ws = ws;
  h->ws_bytes = bytes;
  return FV_OK;
}

int fv_preprocess(fv_handle* h, const void* img, int dtype, int B, int C, int Hin, int Win, float pad_value,
                  int resize_with_padding, void* pix_out, fv_stream s) {
  HandleScope _hs(h);
  if (!h) return fv_fail(FV_ERR_ARG, "null handle");
  hipStream_t st = static_cast<hipStream_t>(s);
  const double S = h->d.image_size;
  prof_begin(h, FV_FAM_ELT, 30.0 * B * S * S, (double)B * C * Hin * Win * (dtype == FV_U8 ? 1 : 4) + B * S * S * 8.0, st);
  const int rc = fv::launch_letterbox(img, dtype, B, C, Hin, Win, h->d.image_size, pad_value, resize_with_padding,
                                      static_cast<bf16_t*>(pix_out), st);
  prof_end(h, st);
  return rc;
}

int fv_preprocess_normalized(fv_handle* h, const void* img, int dtype, int B, int C, int Hin, int Win, float pad_value, int resize_with_padding,
                             const float* mean3, const float* std3, int range_heuristic, void* pix_out, fv_stream s) {
  HandleScope _hs(h);
  if (!h) return fv_fail(FV_ERR_ARG, "null handle");
  if (!mean3 || !std3) return fv_fail(FV_ERR_ARG, "fv_preprocess_normalized: mean / std must be given (3 floats each, host memory)");
  hipStream_t st = static_cast<hipStream_t>(s);
  const double S = h->d.image_size;
  const double in_bytes = (double)B * C * Hin * Win * (dtype == FV_U8 ? 1 : 4);
  prof_begin(h, FV_FAM_ELT, (range_heuristic ? 66.0 : 36.0) * B * S * S, (range_heuristic ? 2.0 : 1.0) * in_bytes + B * S * S * 8.0, st);
  const int rc = fv::launch_letterbox_norm(img, dtype, B, C, Hin, Win, h->d.image_size, pad_value, resize_with_padding, mean3, std3, range_heuristic,
                                           h->lb_vmax, static_cast<bf16_t*>(pix_out), st);
  prof_end(h, st);
  return rc;
}

int fv_vision_forward(fv_handle* h, const void* pix, int B, void* img_tokens, void* tower_out, fv_stream s) {
  HandleScope _hs(h);
  FV_TRY(check_ready(h, true));
  if ((!pix && !h->lbsrc) || !img_tokens || B <= 0) return fv_fail(FV_ERR_ARG, "fv_vision_forward: bad argument");
  if (B > h->d.max_batch) return fv_fail(FV_ERR_ARG, "fv_vision_forward: B=%d exceeds max_batch=%d", B, h->d.max_batch);
  const fv_model_desc& d = h->d;
  // the tower part of the plan does not depend on T / splice
  const WsPlan wp = plan_ws(h, B, 1, 0);
  if (wp.x > h->ws_bytes) return fv_fail(FV_ERR_STATE, "workspace too small for B=%d (%zu > %zu)", B, wp.x, h->ws_bytes);
  const int mb = (d.tower_microbatch > 0 && d.tower_microbatch < B) ? d.tower_microbatch : B;
  const size_t S = d.image_size;
  const int Pside = d.image_size >> (d.tower_stages + 1);
  const size_t P = (size_t)Pside * Pside;
  bf16_t* tout = tower_out ? static_cast<bf16_t*>(tower_out) : reinterpret_cast<bf16_t*>(static_cast<char*>(h->ws) + wp.tower_out);
  for (int b0 = 0; b0 < B; b0 += mb) {
    const int nb = std::min(mb, B - b0);
    FV_TRY(tower_pass(h, pix ? static_cast<const bf16_t*>(pix) + (size_t)b0 * S * S * 4 : nullptr, b0, nb, tout + (size_t)b0 * P * d.tower_out_dim,
                      static_cast<float*>(img_tokens) + (size_t)b0 * P * d.llm_hidden, wp, static_cast<hipStream_t>(s)));
  }
  return FV_OK;
}

int fv_vision_forward_taps(fv_handle* h, const void* pix, int B, void* img_tokens, void* tower_out, void* const* taps,
                           int n_taps, fv_stream s) {
  HandleScope _hs(h);
  if (!h) return fv_fail(FV_ERR_ARG, "null handle");
  if (n_taps < 0 || n_taps > FV_MAX_STAGES + 1 || (n_taps && !taps)) return fv_fail(FV_ERR_ARG, "fv_vision_forward_taps: bad taps");
  h->taps = taps;
  h->n_taps = n_taps;
  const int rc = fv_vision_forward(h, pix, B, img_tokens, tower_out, s);
  h->taps = nullptr;
  h->n_taps = 0;
  return rc;
}

int fv_vision_forward_images(fv_handle* h, const void* img, int dtype, int B, int C, int Hin, int Win, float pad_value, int resize_with_padding,
                             void* img_tokens, void* tower_out, fv_stream s) {
  HandleScope _hs(h);
  if (!h || !img) return fv_fail(FV_ERR_ARG, "fv_vision_forward_images: null argument");
  const fv_handle::LbSrc src{img, dtype, C, Hin, Win, pad_value, resize_with_padding};
  h->lbsrc = &src;
  const int rc = fv_vision_forward(h, nullptr, B, img_tokens, tower_out, s);
  h->lbsrc = nullptr;
  return rc;
}

int fv_vision_unit_info(fv_handle* h, int unit, int32_t* kind, int32_t* stage, int32_t* side, int32_t* channels) {
  HandleScope _hs(h);
  if (!h) return fv_fail(FV_ERR_ARG, "null handle");
  const std::vector<UnitInfo> u = tower_units(h->d);
  if (unit < 0 || unit >= (int)u.size()) return fv_fail(FV_ERR_ARG, "fv_vision_unit_info: unit %d out of range [0, %d)", unit, (int)u.size());
  if (kind) *kind = u[unit].kind;
  if (stage) *stage = u[unit].stage;
  if (side) *side = u[unit].side;
  if (channels) *channels = u[unit].channels;
  return FV_OK;
}

int fv_vision_forward_unit_taps(fv_handle* h, const void* pix, int B, void* img_tokens, void* tower_out, void* const* taps,
                                int n_taps, fv_stream s) {
  HandleScope _hs(h);
  if (!h) return fv_fail(FV_ERR_ARG, "null handle");
  if (n_taps < 0 || (n_taps && !taps) || n_taps > (int)tower_units(h->d).size()) return fv_fail(FV_ERR_ARG, "fv_vision_forward_unit_taps: bad taps");
  h->utaps = taps;
  h->n_utaps = n_taps;
  const int rc = fv_vision_forward(h, pix, B, img_tokens, tower_out, s);
  h->utaps = nullptr;
  h->n_utaps = 0;
  return rc;
}

int fv_llm_forward_pooled(fv_handle* h, const int32_t* ids, const int32_t* lens, const void* img_tokens, int Ni, int B,
                          int T, int pool_mode, void* pooled, fv_stream st) {
  HandleScope _hs(h);
  FV_TRY(check_ready(h, true));
  if (!ids || !lens || !pooled || B <= 0 || T <= 0 || Ni < 0) return fv_fail(FV_ERR_ARG, "fv_llm_forward_pooled: bad argument");
  if (Ni > 0 && !img_tokens) return fv_fail(FV_ERR_ARG, "fv_llm_forward_pooled: Ni > 0 without img_tokens");
  if (!img_tokens) Ni = 0;
  const fv_model_desc& d = h->d;
  const int Tt = T + Ni;
  if (Tt > h->rope_rows) return fv_fail(FV_ERR_ARG, "sequence of %d tokens exceeds the RoPE table (%d)", Tt, h->rope_rows);
  if (B > d.max_batch) return fv_fail(FV_ERR_ARG, "B=%d exceeds max_batch=%d", B, d.max_batch);
  const int Pside = d.image_size >> (d.tower_stages + 1);
  const WsPlan wp = plan_ws(h, B, T, Ni > 0);
  if (Ni > Pside * Pside || wp.total > h->ws_bytes) return fv_fail(FV_ERR_STATE, "workspace too small (%zu > %zu) or Ni too large", wp.total, h->ws_bytes);
  hipStream_t s = static_cast<hipStream_t>(st);
  char* ws = static_cast<char*>(h->ws);
  float* x = reinterpret_cast<float*>(ws + wp.x);
  bf16_t* xn = reinterpret_cast<bf16_t*>(ws + wp.xn);
  bf16_t* qkv = reinterpret_cast<bf16_t*>(ws + wp.qkv);
  bf16_t* att = reinterpret_cast<bf16_t*>(ws + wp.att);
  bf16_t* act = reinterpret_cast<bf16_t*>(ws + wp.act);
  const int rows = B * Tt, Hd = d.llm_hidden, D = d.llm_head_dim;
  const int qd = d.llm_heads * D, kd = d.llm_kv_heads * D, qkvw = qd + 2 * kd;
  FV_P(FV_FAM_ELT, 0.0, 6.0 * rows * Hd, fv::launch_embed_gather(ids, h->dec.embed, static_cast<const float*>(img_tokens), x, B, T, Ni, Hd, d.llm_vocab, s));
  const float att_scale = 1.0f / std::sqrt((float)D);
  if (d.llm_precision == 0) {
    for (const DecLayer& L : h->dec.layers) {
      FV_P(FV_FAM_NORM, 4.0 * rows * Hd, 6.0 * rows * Hd, fv::launch_rmsnorm(x, L.ln1, xn, nullptr, Hd, rows, Hd, d.rms_eps, s));
      fv::GemmArgs gq{xn, Hd, L.qkv_w, rows, qkvw, Hd, L.qkv_b, nullptr, nullptr, 0, qkv, qkvw, FV_EPI_BIAS};
      FV_TRY(gemm_p(h, gq, s));
      FV_P(FV_FAM_ELT, 3.0 * rows * (qd + kd), 4.0 * rows * (qd + kd), fv::launch_rope(qkv, h->rope, qkvw, rows, Tt, d.llm_heads, d.llm_kv_heads, D, s));
      FV_P(FV_FAM_ATTN, 2.0 * B * (double)Tt * Tt * qd, 2.0 * rows * (qkvw + qd),
           fv::launch_attention(qkv, qkv + qd, qkv + qd + kd, qkvw, qkvw, qkvw, att, qd, B, Tt, d.llm_heads, d.llm_kv_heads,
                                D, 1, lens, Ni, att_scale, s));
      fv::GemmArgs go{att, qd, L.o_w, rows, Hd, qd, nullptr, nullptr, x, Hd, x, Hd, FV_EPI_RES_F32};
      FV_TRY(gemm_p(h, go, s));
      FV_P(FV_FAM_NORM, 4.0 * rows * Hd, 6.0 * rows * Hd, fv::launch_rmsnorm(x, L.ln2, xn, nullptr, Hd, rows, Hd, d.rms_eps, s));
      fv::GemmArgs gg{xn, Hd, L.gu_w, rows, 2 * d.llm_inter, Hd, nullptr, nullptr, nullptr, 0, act, d.llm_inter, FV_EPI_SWIGLU};
      FV_TRY(gemm_p(h, gg, s));
      fv::GemmArgs gd{act, d.llm_inter, L.down_w, rows, Hd, d.llm_inter, nullptr, nullptr, x, Hd, x, Hd, FV_EPI_RES_F32};
      FV_TRY(gemm_p(h, gd, s));
    }
  } else {
    FV_TRY(decoder_layers_split(h, wp, B, Tt, lens, Ni, nullptr, 0, DEC_FULL, s));
  }
  FV_P(FV_FAM_ELT, 4.0 * B * Hd, 8.0 * B * Hd, fv::launch_pool_norm(x, lens, h->dec.norm, static_cast<float*>(pooled), B, Tt, Ni, Hd, d.rms_eps, pool_mode, s));
  return FV_OK;
}

// ---- image-prefix reuse.  The spliced sequence is [Ni image tokens | T text tokens] under a causal mask: the
// keys / values of the image positions depend on the image alone.  fv_llm_prefix runs those positions once and keeps every layer's
// [k | v] rows; fv_llm_forward_pooled_prefixed then runs only the text positions against them -- same arithmetic as
// fv_llm_forward_pooled(img_tokens != NULL), 1/5 of its rows at T = 64, and no tower / projector / prefix pass at all for an image
// (or a batch of images) whose prefix is already held.
int fv_set_batch_invariant(fv_handle* h, int on) {
  HandleScope _hs(h);
  if (!h) return fv_fail(FV_ERR_ARG, "fv_set_batch_invariant: null handle");
  h->batch_invariant = on != 0;
  return FV_OK;
}

int fv_llm_fp16_saturations(fv_handle* h, uint64_t* count_out, int reset) {
  HandleScope _hs(h);
  if (!h || !count_out) return fv_fail(FV_ERR_ARG, "fv_llm_fp16_saturations: null argument");
  FV_HIP_CHECK(hipSetDevice(h->device));
  FV_HIP_CHECK(hipDeviceSynchronize());
  unsigned n = 0;
  FV_HIP_CHECK(hipMemcpy(&n, h->f16_flags, 4, hipMemcpyDeviceToHost));
  if (reset) FV_HIP_CHECK(hipMemset(h->f16_flags, 0, 4));
  *count_out = n;
  return FV_OK;
}

int fv_llm_prefix_bytes(fv_handle* h, int B, int Ni, size_t* out_bytes) {
  HandleScope _hs(h);
  if (!h || !out_bytes || B <= 0 || Ni <= 0) return fv_fail(FV_ERR_ARG, "fv_llm_prefix_bytes: bad argument");
  *out_bytes = (size_t)h->d.llm_layers * B * Ni * 2 * h->d.llm_kv_heads * h->d.llm_head_dim * sizeof(float);
  return FV_OK;
}

int fv_llm_prefix(fv_handle* h, const void* img_tokens, int Ni, int B, void* kv_out, fv_stream st) {
  HandleScope _hs(h);
  FV_TRY(check_ready(h, true));
  if (!img_tokens || !kv_out || B <= 0 || Ni <= 0) return fv_fail(FV_ERR_ARG, "fv_llm_prefix: bad argument");
  const fv_model_desc& d = h->d;
  if (d.llm_precision == 0 || d.llm_head_dim < 64) return fv_fail(FV_ERR_UNSUPPORTED, "fv_llm_prefix: needs llm_precision 1 / 2 and head_dim 64 / 128");
  const int Pside = d.image_size >> (d.tower_stages + 1);
  if (B > d.max_batch || Ni > Pside * Pside) return fv_fail(FV_ERR_ARG, "fv_llm_prefix: B=%d / Ni=%d exceed the handle's capacity", B, Ni);
  const WsPlan wp = plan_ws(h, B, 1, 1);
  if (wp.total > h->ws_bytes) return fv_fail(FV_ERR_STATE, "workspace too small (%zu > %zu)", wp.total, h->ws_bytes);
  hipStream_t s = static_cast<hipStream_t>(st);
  float* x = reinterpret_cast<float*>(static_cast<char*>(h->ws) + wp.x);
  FV_HIP_CHECK(hipMemcpyAsync(x, img_tokens, (size_t)B * Ni * d.llm_hidden * 4, hipMemcpyDeviceToDevice, s));   // the prefix IS the image tokens
  return decoder_layers_split(h, wp, B, Ni, nullptr, 0, static_cast<float*>(kv_out), Ni, DEC_PREFIX, s);
}

int fv_llm_forward_pooled_prefixed(fv_handle* h, const int32_t* ids, const int32_t* lens, const void* kv, int Ni, int B, int T,
                                   int pool_mode, void* pooled, fv_stream st) {
  HandleScope _hs(h);
  FV_TRY(check_ready(h, true));
  if (!ids || !lens || !kv || !pooled || B <= 0 || T <= 0 || Ni <= 0) return fv_fail(FV_ERR_ARG, "fv_llm_forward_pooled_prefixed: bad argument");
  const fv_model_desc& d = h->d;
  if (d.llm_precision == 0 || d.llm_head_dim < 64) return fv_fail(FV_ERR_UNSUPPORTED, "fv_llm_forward_pooled_prefixed: needs llm_precision 1 / 2 and head_dim 64 / 128");
  if (pool_mode != 0) return fv_fail(FV_ERR_UNSUPPORTED, "fv_llm_forward_pooled_prefixed: mean_pool averages the image positions too, which this pass does not recompute");
  if (T + Ni > h->rope_rows) return fv_fail(FV_ERR_ARG, "sequence of %d tokens exceeds the RoPE table (%d)", T + Ni, h->rope_rows);
  if (B > d.max_batch) return fv_fail(FV_ERR_ARG, "B=%d exceeds max_batch=%d", B, d.max_batch);
  const WsPlan wp = plan_ws(h, B, T, 1);
  if (wp.total > h->ws_bytes) return fv_fail(FV_ERR_STATE, "workspace too small (%zu > %zu)", wp.total, h->ws_bytes);
  hipStream_t s = static_cast<hipStream_t>(st);
  float* x = reinterpret_cast<float*>(static_cast<char*>(h->ws) + wp.x);
  const int Hd = d.llm_hidden;
  FV_P(FV_FAM_ELT, 0.0, 6.0 * B * T * Hd, fv::launch_embed_gather(ids, h->dec.embed, nullptr, x, B, T, 0, Hd, d.llm_vocab, s));
  FV_TRY(decoder_layers_split(h, wp, B, T, lens, Ni, const_cast<float*>(static_cast<const float*>(kv)), Ni, DEC_SUFFIX, s));
  FV_P(FV_FAM_ELT, 4.0 * B * Hd, 8.0 * B * Hd, fv::launch_pool_norm(x, lens, h->dec.norm, static_cast<float*>(pooled), B, T, 0, Hd, d.rms_eps, 0, s));
  return FV_OK;
}

int fv_head_layout(fv_handle* h, int64_t offsets[13]) {
  HandleScope _hs(h);
  if (!h || !offsets) return fv_fail(FV_ERR_ARG, "fv_head_layout: null argument");
  const fv::HeadOffsets ho = fv::head_offsets(h->hd);
  for (int i = 0; i < 13; ++i) offsets[i] = ho.o[i];
  return FV_OK;
}

int fv_head_saved_bytes(fv_handle* h, int B, size_t* out_bytes) {
  HandleScope _hs(h);
  if (!h || !out_bytes || B <= 0) return fv_fail(FV_ERR_ARG, "fv_head_saved_bytes: bad argument");
  *out_bytes = fv::head_saved_bytes(h->hd, B);
  return FV_OK;
}

int fv_head_forward(fv_handle* h, const float* flat_params, const float* pooled, const float* states, int B,
                    int training, float dropout_p, uint64_t seed, uint64_t offset, float* actions, void* saved,
                    fv_stream s) {
  HandleScope _hs(h);
  if (!h) return fv_fail(FV_ERR_ARG, "null handle");
  hipStream_t st = static_cast<hipStream_t>(s);
  prof_begin(h, FV_FAM_HEAD, 2.0 * B * fv::head_offsets(h->hd).o[12], 4.0 * fv::head_offsets(h->hd).o[12], st);
  const int rc = fv::launch_head_forward(h->hd, flat_params, pooled, states, B, training, dropout_p, seed, offset, actions,
                                         static_cast<float*>(saved), st, h->has_io ? &h->io : nullptr);
  prof_end(h, st);
  return rc;
}

int fv_head_set_io_norm(fv_handle* h, const float* state_mean, const float* state_std, const float* action_mean,
                        const float* action_std, float eps) {
  HandleScope _hs(h);
  if (!h) return fv_fail(FV_ERR_ARG, "null handle");
  if (!state_mean && !state_std && !action_mean && !action_std) { h->has_io = false; return FV_OK; }   // all NULL: folding off
  if (!state_mean || !state_std || !action_mean || !action_std) return fv_fail(FV_ERR_ARG, "fv_head_set_io_norm: give all four vectors or none");
  FV_HIP_CHECK(hipSetDevice(h->device));
  const int ds = h->hd.ds, da = h->hd.da;
  std::vector<float> v((size_t)2 * ds + 2 * da);
  for (int i = 0; i < ds; ++i) { v[i] = state_mean[i]; v[ds + i] = 1.0f / (state_std[i] + eps); }
  for (int i = 0; i < da; ++i) { v[2 * ds + i] = action_mean[i]; v[2 * ds + da + i] = action_std[i]; }
  if (!h->io_buf) {   // allocated once: toggling the folding on / off / on does not grow the handle
    void* p = nullptr;
    FV_TRY(dev_alloc(h, v.size() * 4, &p));
    h->io_buf = static_cast<float*>(p);
  }
  // a head kernel still in flight on any stream may be reading the previous statistics: a configuration call, so it simply waits
  FV_HIP_CHECK(hipDeviceSynchronize());
  FV_HIP_CHECK(hipMemcpy(h->io_buf, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  float* f = h->io_buf;
  h->io = fv::HeadIoNorm{f, f + ds, f + 2 * ds, f + 2 * ds + da};
  h->has_io = true;
  return FV_OK;
}

int fv_head_mse_backward(fv_handle* h, const float* flat_params, const float* actions, const float* targets, int B,
                         float dropout_p, const void* saved, float* loss, float* flat_grads, fv_stream s) {
  HandleScope _hs(h);
  if (!h) return fv_fail(FV_ERR_ARG, "null handle");
  if (!h->ws) return fv_fail(FV_ERR_STATE, "workspace not bound: call fv_bind_workspace first");
  if (B > h->d.max_batch) return fv_fail(FV_ERR_ARG, "B=%d exceeds max_batch=%d", B, h->d.max_batch);
  const WsPlan wp = plan_ws(h, B, 1, 0);
  if (wp.total > h->ws_bytes) return fv_fail(FV_ERR_STATE, "workspace too small for head backward");
  float* scr = reinterpret_cast<float*>(static_cast<char*>(h->ws) + wp.head_scr);
  hipStream_t st = static_cast<hipStream_t>(s);
  prof_begin(h, FV_FAM_HEAD, 4.0 * B * fv::head_offsets(h->hd).o[12], 8.0 * fv::head_offsets(h->hd).o[12], st);
  const int rc = fv::launch_head_backward(h->hd, flat_params, nullptr, actions, targets, B, dropout_p,
                                          static_cast<const float*>(saved), loss, flat_grads, scr, st);
  prof_end(h, st);
  return rc;
}

int fv_head_backward(fv_handle* h, const float* flat_params, const float* grad_actions, int B, float dropout_p,
                     const void* saved, float* flat_grads, fv_stream s) {
  HandleScope _hs(h);
  if (!h || !grad_actions) return fv_fail(FV_ERR_ARG, "fv_head_backward: null argument");
  if (!h->ws) return fv_fail(FV_ERR_STATE, "workspace not bound: call fv_bind_workspace first");
  if (B > h->d.max_batch) return fv_fail(FV_ERR_ARG, "B=%d exceeds max_batch=%d", B, h->d.max_batch);
  const WsPlan wp = plan_ws(h, B, 1, 0);
  if (wp.total > h->ws_bytes) return fv_fail(FV_ERR_STATE, "workspace too small for head backward");
  float* scr = reinterpret_cast<float*>(static_cast<char*>(h->ws) + wp.head_scr);
  return fv::launch_head_backward(h->hd, flat_params, grad_actions, nullptr, nullptr, B, dropout_p,
                                  static_cast<const float*>(saved), nullptr, flat_grads, scr, static_cast<hipStream_t>(s));
}

int fv_profile(fv_handle* h, int enable) {
  HandleScope _hs(h);
  if (!h) return fv_fail(FV_ERR_ARG, "null handle");
  h->prof.on = enable != 0;
  h->prof.recs.clear();
  h->prof.used = 0;
  return FV_OK;
}

int fv_profile_read(fv_handle* h, fv_profile_entry* fam_out, fv_gemm_profile* gemm_out, int max_gemm, int* n_gemm) {
  HandleScope _hs(h);
  if (!h || !fam_out) return fv_fail(FV_ERR_ARG, "fv_profile_read: null argument");
  for (int i = 0; i < FV_FAM_COUNT; ++i) fam_out[i] = fv_profile_entry{0.0, 0.0, 0.0, 0};
  std::map<std::vector<int>, fv_gemm_profile> shapes;
  for (ProfRec& r : h->prof.recs) {
    FV_HIP_CHECK(hipEventSynchronize(r.e1));
    float ms = 0.f;
    FV_HIP_CHECK(hipEventElapsedTime(&ms, r.e0, r.e1));
    fv_profile_entry& e = fam_out[r.fam];
    e.ms += ms; e.flops += r.flops; e.bytes += r.bytes; e.launches += 1;
    if (r.fam == FV_FAM_GEMM) {
      fv_gemm_profile& g = shapes[{r.m, r.n, r.k, r.epi}];
      g.m = r.m; g.n = r.n; g.k = r.k; g.epi = r.epi; g.ms += ms; g.launches += 1;
    }
  }
  int n = 0;
  if (gemm_out)
    for (auto& kv : shapes) { if (n >= max_gemm) break; gemm_out[n++] = kv.second; }
  if (n_gemm) *n_gemm = n;
  h->prof.recs.clear();
  h->prof.used = 0;
  return FV_OK;
}

int fv_adamw_clip_step(fv_handle* h, float* flat_params, const float* flat_grads, float* m, float* v, int64_t n,
                       const fv_adamw_hparams* hp, int64_t step, float* grad_norm_out, fv_stream s) {
  HandleScope _hs(h);
  if (!h || !hp) return fv_fail(FV_ERR_ARG, "fv_adamw_clip_step: null argument");
  return fv::launch_adamw_clip(flat_params, flat_grads, m, v, n, *hp, step, h->norm_scratch, grad_norm_out,
                               static_cast<hipStream_t>(s));
}


// ---- gradient accumulation helpers (training/trainer.py:96,171: accelerate's accumulate() sums micro-batch gradients)
int fv_grad_accumulate(fv_handle* h, float* acc, const float* grads, int64_t n, fv_stream s) {
  HandleScope _hs(h);
  if (!h || !acc || !grads || n <= 0) return fv_fail(FV_ERR_ARG, "fv_grad_accumulate: bad argument");
  return fv::launch_axpy(acc, grads, n, nullptr, static_cast<hipStream_t>(s));
}

int fv_grad_scale(fv_handle* h, float* grads, int64_t n, const float* scale_dev, fv_stream s) {
  HandleScope _hs(h);
  if (!h || !grads || !scale_dev || n <= 0) return fv_fail(FV_ERR_ARG, "fv_grad_scale: bad argument");
  return fv::launch_axpy(grads, nullptr, n, scale_dev, static_cast<hipStream_t>(s));
}

// ---- data-parallel exchange on RCCL, without torch in between.  librccl is resolved at first use with
// dlopen -- the copy the process already holds (torch ships one) when there is one -- so the library itself carries no link
// dependency on it and single-GPU users never load it.
namespace {
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, fv_rccl_id, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
int rccl_load() {
  if (g_rccl.lib) return FV_OK;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  void* lib = nullptr;
  for (const char* n : names) if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);   // already in the process (torch's)
  for (const char* n : names) if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return fv_fail(FV_ERR_UNSUPPORTED, "librccl.so cannot be loaded: %s", dlerror());
  Rccl r;
  r.lib = lib;
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
  r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(lib, "ncclAllReduce"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy) return fv_fail(FV_ERR_UNSUPPORTED, "librccl.so lacks the nccl* entry points");
  g_rccl = r;
  return FV_OK;
}
int rccl_fail(int e, const char* what) {
  return fv_fail(FV_ERR_HIP, "RCCL error %d (%s) in %s", e, g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "?", what);
}
}  // namespace

int fv_comm_unique_id(fv_handle* h, fv_rccl_id* id_out) {
  HandleScope _hs(h);
  if (!h || !id_out) return fv_fail(FV_ERR_ARG, "fv_comm_unique_id: null argument");
  FV_TRY(rccl_load());
  const int e = g_rccl.GetUniqueId(id_out);
  return e ? rccl_fail(e, "ncclGetUniqueId") : FV_OK;
}

int fv_comm_init(fv_handle* h, const fv_rccl_id* id, int rank, int world, void** comm_out) {
  HandleScope _hs(h);
  if (!h || !id || !comm_out || world < 1 || rank < 0 || rank >= world) return fv_fail(FV_ERR_ARG, "fv_comm_init: bad argument");
  FV_TRY(rccl_load());
  FV_HIP_CHECK(hipSetDevice(h->device));
  const int e = g_rccl.CommInitRank(comm_out, world, *id, rank);
  return e ? rccl_fail(e, "ncclCommInitRank") : FV_OK;
}

int fv_comm_destroy(fv_handle* h, void* comm) {
  HandleScope _hs(h);
  if (!h || !comm) return fv_fail(FV_ERR_ARG, "fv_comm_destroy: null argument");
  FV_TRY(rccl_load());
  const int e = g_rccl.CommDestroy(comm);
  return e ? rccl_fail(e, "ncclCommDestroy") : FV_OK;
}

int fv_allreduce_grads(fv_handle* h, void* comm, float* flat_grads, int64_t n, fv_stream s) {
  HandleScope _hs(h);
  if (!h || !comm || !flat_grads || n <= 0) return fv_fail(FV_ERR_ARG, "fv_allreduce_grads: bad argument");
  FV_TRY(rccl_load());
  const int e = g_rccl.AllReduce(flat_grads, flat_grads, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, static_cast<hipStream_t>(s));
  return e ? rccl_fail(e, "ncclAllReduce") : FV_OK;
}

}  // extern "C"

#include "train_path.inc"
#include "tower_train.inc"
