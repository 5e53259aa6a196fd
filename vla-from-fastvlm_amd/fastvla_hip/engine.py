"""Host-side wrapper around one fv_handle: torch owns device memory and streams, the library does the arithmetic.

One engine per process / GPU (one process per GPU under torch.distributed).  Everything here is plumbing: allocate
buffers with torch's caching allocator, hand raw pointers and the current HIP stream to libfastvla_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from .arch import ModelConfig

HEAD_KEYS = (
    "state_projection.0.weight", "state_projection.0.bias", "state_projection.1.weight", "state_projection.1.bias",
    "fusion.0.weight", "fusion.0.bias", "fusion.1.weight", "fusion.1.bias", "fusion.4.weight", "fusion.4.bias",
    "action_head.weight", "action_head.bias",
)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class FastVLAEngine:
    def __init__(self, model: ModelConfig, *, state_dim: int = 14, action_dim: int = 14, hidden_dim: int = 1024,
                 fusion_dim: int = 1024, device: Optional[torch.device] = None, max_batch: int = 64,
                 max_text_tokens: int = 64, tower_microbatch: int = 0, llm_precision: int = 1):
        self.lib = _lib.load()  # raises if the HIP library is missing: no fallback
        if not torch.cuda.is_available():
            raise _lib.FastVLAHipError("no HIP device visible: the FastVLA HIP path needs an MI355X (no CPU fallback)")
        self.device = torch.device(device if device is not None else "cuda:0")
        if self.device.type != "cuda":
            raise _lib.FastVLAHipError(f"FastVLAEngine needs a cuda (ROCm) device, got {self.device}")
        self.model = model
        self.head_dims = dict(feat=model.llm.hidden, ds=state_dim, da=action_dim, hid=hidden_dim, fus=fusion_dim)
        d = _lib.ModelDesc()
        l, t = model.llm, model.tower
        d.llm_hidden, d.llm_layers, d.llm_heads, d.llm_kv_heads = l.hidden, l.layers, l.heads, l.kv_heads
        d.llm_head_dim, d.llm_inter, d.llm_vocab = l.head_dim, l.inter, l.vocab
        d.rope_theta, d.rms_eps = l.rope_theta, l.rms_eps
        d.tower_stages = len(t.layers)
        for i in range(len(t.layers)):
            d.tower_layers[i], d.tower_dims[i], d.tower_is_attn[i] = t.layers[i], t.dims[i], int(i in t.attn_stages)
        d.tower_mlp_ratio, d.tower_head_dim, d.tower_se_rd, d.tower_out_dim = t.mlp_ratio, t.head_dim, t.se_rd, t.out_dim
        d.ln_eps, d.bn_eps, d.image_size = t.ln_eps, t.bn_eps, t.image_size
        d.state_dim, d.action_dim, d.hidden_dim, d.fusion_dim = state_dim, action_dim, hidden_dim, fusion_dim
        d.max_batch, d.max_text_tokens, d.tower_microbatch = max_batch, max_text_tokens, tower_microbatch
        d.llm_precision = int(llm_precision)
        self.llm_precision = int(llm_precision)
        self.desc = d
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fv_create(C.byref(d), self.device.index or 0, C.byref(h)), "fv_create")
        self.h = h
        self._ws: Optional[torch.Tensor] = None
        self._side: Optional[torch.cuda.Stream] = None
        # FASTVLA_FUSED_LETTERBOX=1: backbone() hands raw images to the stem (fv_vision_forward_images) instead of letterboxing first
        self.fused_letterbox = os.environ.get("FASTVLA_FUSED_LETTERBOX", "0") == "1" and model.tower.dims[0] == 96
        self.overlap_streams = os.environ.get("FASTVLA_OVERLAP", "1") == "1"
        # An image's tower tokens must not depend on the batch it is evaluated in once they are CONSUMED (splice mode: the small-batch kernel forms put 1.7e-2
        # between an observation alone and the same observation inside a batch).  FASTVLA_BATCH_INVARIANT=1 / 0 fixes the choice; unset, the numerics follow the
        # mode: tokens_consumed(True) -- the host side calls it whenever the tower output feeds the decoder -- switches the large-batch forms on, literal mode
        # (tokens computed and dropped) keeps the fast small-batch forms.
        env = os.environ.get("FASTVLA_BATCH_INVARIANT")
        self._invariant_policy: Optional[bool] = None if env in (None, "") else env == "1"
        self._invariant_on = False
        if self._invariant_policy:
            self.set_batch_invariant(True)
        offs = (C.c_int64 * 13)()
        _lib.check(self.lib.fv_head_layout(self.h, C.byref(offs)), "fv_head_layout")
        self.head_offsets = list(offs)
        self.loaded = False

    def close(self):
        if getattr(self, "h", None):
            self.lib.fv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- frozen weights
    def load_weights(self, state: Dict[str, torch.Tensor]) -> None:
        keep, descs = [], (_lib.TensorDesc * len(state))()
        for i, (k, v) in enumerate(state.items()):
            t = v.detach().to("cpu")
            if t.dtype not in (torch.float32, torch.bfloat16):
                t = t.float()
            t = t.contiguous()
            keep.append(t)
            nm = k.encode()
            keep.append(nm)
            descs[i].name = nm
            descs[i].data = t.data_ptr()
            descs[i].dtype = _lib.FV_F32 if t.dtype == torch.float32 else _lib.FV_BF16
            descs[i].ndim = max(t.ndim, 1)
            if t.ndim > 4:
                raise ValueError(f"{k}: rank > 4")
            for j, s in enumerate(t.shape if t.ndim else (1,)):
                descs[i].shape[j] = s
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fv_load_weights(self.h, descs, len(state)), "fv_load_weights", self.h)
        self.loaded = True

    def load_weights_streaming(self, provider) -> None:
        """Pack the frozen weights pulling ONE tensor at a time: provider(name) -> torch.Tensor (f32 or bf16, on the CPU or on
        this engine's device) or None when the checkpoint has no such key.  Nothing but the tensor being packed is alive on
        the host side, and bf16 tensors are copied as they are (fv_load_weights_cb)."""
        keep = {}

        def cb(_user, name, out):
            try:
                t = provider(name.decode())
            except Exception as exc:  # an exception must not unwind through the C frames
                keep["exc"] = exc
                return 1
            if t is None:
                return 1
            t = t.detach()
            if t.dtype not in (torch.float32, torch.bfloat16):
                t = t.float()
            if t.device.type == "cuda" and t.device != self.device:
                t = t.to(self.device)
            t = t.contiguous()
            if t.ndim > 4:
                keep["exc"] = ValueError(f"{name.decode()}: rank > 4")
                return 1
            if t.device.type == "cuda":
                # the library copies with blocking calls on the null stream: whatever produced t on torch's current stream
                # (a generator, a dtype cast, a .to(device)) must have finished before the pointer is handed over
                torch.cuda.current_stream(t.device).synchronize()
            keep["t"] = t  # alive until the next call
            d = out.contents
            d.data = t.data_ptr()
            d.dtype = _lib.FV_F32 if t.dtype == torch.float32 else _lib.FV_BF16
            d.ndim = max(t.ndim, 1)
            for j, n in enumerate(t.shape if t.ndim else (1,)):
                d.shape[j] = n
            d.device = int(t.device.type == "cuda")
            return 0

        fn = _lib.TENSOR_PROVIDER(cb)
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            rc = self.lib.fv_load_weights_cb(self.h, fn, None)
        if "exc" in keep:
            raise keep["exc"]
        _lib.check(rc, "fv_load_weights_cb", self.h)
        self.loaded = True

    # ---------------------------------------------------------------- workspace
    def workspace_bytes(self, B: int, T: int, splice: bool) -> int:
        n = C.c_size_t()
        _lib.check(self.lib.fv_workspace_bytes(self.h, B, T, int(splice), C.byref(n)), "fv_workspace_bytes")
        return n.value

    def ensure_workspace(self, B: int, T: int, splice: bool) -> None:
        need = self.workspace_bytes(B, T, splice)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
            base = self._ws.data_ptr()
            off = (-base) % 256
            _lib.check(self.lib.fv_bind_workspace(self.h, base + off, self._ws.numel() - off), "fv_bind_workspace")

    # ---------------------------------------------------------------- frozen backbone
    IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)   # reference model/fastvlm_adapter.py:468-469,475-476

    def preprocess(self, images: torch.Tensor, pad_value: float = 0.0, resize_with_padding: bool = True, normalize_imagenet: bool = False,
                   range_heuristic: bool = True) -> torch.Tensor:
        """letterbox (+ the reference's `_maybe_normalize_imagenet`, :463-477, when asked; range_heuristic = its torchvision branch, which is what an installed
        reference runs: torchvision is one of its dependencies)."""
        if images.ndim != 4:
            raise ValueError(f"(B,C,H,W) expected, but got shape {tuple(images.shape)}")
        if images.dtype == torch.uint8:
            dt = _lib.FV_U8
        else:
            images = images.to(torch.float32)
            dt = _lib.FV_F32
        images = images.to(self.device).contiguous()
        B, Cc, H, W = images.shape
        S = self.model.tower.image_size
        pix = torch.empty(B, S, S, 4, dtype=torch.bfloat16, device=self.device)
        if normalize_imagenet:
            import ctypes
            mean, std = (ctypes.c_float * 3)(*self.IMAGENET_MEAN), (ctypes.c_float * 3)(*self.IMAGENET_STD)
            _lib.check(self.lib.fv_preprocess_normalized(self.h, images.data_ptr(), dt, B, Cc, H, W, float(pad_value), int(resize_with_padding), mean, std,
                                                         int(range_heuristic), pix.data_ptr(), _stream()), "fv_preprocess_normalized")
            return pix
        _lib.check(self.lib.fv_preprocess(self.h, images.data_ptr(), dt, B, Cc, H, W, float(pad_value),
                                          int(resize_with_padding), pix.data_ptr(), _stream()), "fv_preprocess")
        return pix

    def vision_forward(self, pix: torch.Tensor, return_tower_out: bool = False):
        B = pix.shape[0]
        t, l = self.model.tower, self.model.llm
        self.ensure_workspace(B, 1, False)
        tok = torch.empty(B, t.num_tokens, l.hidden, dtype=torch.float32, device=self.device)
        tout = torch.empty(B, t.num_tokens, t.out_dim, dtype=torch.bfloat16, device=self.device) if return_tower_out else None
        _lib.check(self.lib.fv_vision_forward(self.h, pix.data_ptr(), B, tok.data_ptr(), _ptr(tout), _stream()),
                   "fv_vision_forward")
        return (tok, tout) if return_tower_out else tok

    def vision_forward_images(self, images: torch.Tensor, pad_value: float = 0.0, resize_with_padding: bool = True,
                              return_tower_out: bool = False):
        """preprocess() + vision_forward() as ONE call (fv_vision_forward_images): the stem samples the source images itself, the
        letterboxed 1024^2 frame never exists in HBM.  Same tokens, bit for bit."""
        if images.ndim != 4:
            raise ValueError(f"(B,C,H,W) expected, but got shape {tuple(images.shape)}")
        if images.dtype == torch.uint8:
            dt = _lib.FV_U8
        else:
            images = images.to(torch.float32)
            dt = _lib.FV_F32
        images = images.to(self.device).contiguous()
        B, Cc, H, W = images.shape
        t, l = self.model.tower, self.model.llm
        self.ensure_workspace(B, 1, False)
        tok = torch.empty(B, t.num_tokens, l.hidden, dtype=torch.float32, device=self.device)
        tout = torch.empty(B, t.num_tokens, t.out_dim, dtype=torch.bfloat16, device=self.device) if return_tower_out else None
        _lib.check(self.lib.fv_vision_forward_images(self.h, images.data_ptr(), dt, B, Cc, H, W, float(pad_value), int(resize_with_padding),
                                                     tok.data_ptr(), _ptr(tout), _stream()), "fv_vision_forward_images", self.h)
        return (tok, tout) if return_tower_out else tok

    def vision_forward_taps(self, pix: torch.Tensor):
        """-> (tokens f32, tower_out bf16, [stem, stage0, ...] NHWC bf16): per-stage activation maps for the parity tests."""
        B = pix.shape[0]
        t, l = self.model.tower, self.model.llm
        self.ensure_workspace(B, 1, False)
        tok = torch.empty(B, t.num_tokens, l.hidden, dtype=torch.float32, device=self.device)
        tout = torch.empty(B, t.num_tokens, t.out_dim, dtype=torch.bfloat16, device=self.device)
        side = t.image_size // 4
        taps = [torch.empty(B, side, side, t.dims[0], dtype=torch.bfloat16, device=self.device)]
        for i, c in enumerate(t.dims):
            taps.append(torch.empty(B, side >> i, side >> i, c, dtype=torch.bfloat16, device=self.device))
        arr = (C.c_void_p * len(taps))(*[x.data_ptr() for x in taps])
        _lib.check(self.lib.fv_vision_forward_taps(self.h, pix.data_ptr(), B, tok.data_ptr(), tout.data_ptr(), arr, len(taps),
                                                   _stream()), "fv_vision_forward_taps", self.h)
        return tok, tout, taps

    UNIT_KINDS = ("stem", "cpe", "block", "down")

    def tower_units(self):
        """-> [(kind, stage, side, channels)] of every tower unit in execution order (fv_vision_unit_info)."""
        out, u = [], 0
        k, st, sd, ch = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        while self.lib.fv_vision_unit_info(self.h, u, C.byref(k), C.byref(st), C.byref(sd), C.byref(ch)) == 0:
            out.append((self.UNIT_KINDS[k.value], st.value, sd.value, ch.value))
            u += 1
        return out

    def vision_forward_unit_taps(self, pix: torch.Tensor):
        """-> (tokens f32, tower_out bf16, [output of unit 0, 1, ...] NHWC bf16): unit u's input is taps[u - 1], so the parity
        test feeds each oracle unit the ENGINE's own input (teacher forcing)."""
        B = pix.shape[0]
        t, l = self.model.tower, self.model.llm
        self.ensure_workspace(B, 1, False)
        tok = torch.empty(B, t.num_tokens, l.hidden, dtype=torch.float32, device=self.device)
        tout = torch.empty(B, t.num_tokens, t.out_dim, dtype=torch.bfloat16, device=self.device)
        taps = [torch.empty(B, sd, sd, ch, dtype=torch.bfloat16, device=self.device) for _, _, sd, ch in self.tower_units()]
        arr = (C.c_void_p * len(taps))(*[x.data_ptr() for x in taps])
        _lib.check(self.lib.fv_vision_forward_unit_taps(self.h, pix.data_ptr(), B, tok.data_ptr(), tout.data_ptr(), arr, len(taps),
                                                        _stream()), "fv_vision_forward_unit_taps", self.h)
        return tok, tout, taps

    def llm_pooled(self, ids: torch.Tensor, lens: torch.Tensor, img_tokens: Optional[torch.Tensor] = None,
                   pool_mode: int = 0) -> torch.Tensor:
        B, T = ids.shape
        ids = ids.to(device=self.device, dtype=torch.int32).contiguous()
        lens = lens.to(device=self.device, dtype=torch.int32).contiguous()
        ni = 0 if img_tokens is None else img_tokens.shape[1]
        self.ensure_workspace(B, T, img_tokens is not None)
        pooled = torch.empty(B, self.model.llm.hidden, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.fv_llm_forward_pooled(self.h, ids.data_ptr(), lens.data_ptr(), _ptr(img_tokens), ni, B, T,
                                                  pool_mode, pooled.data_ptr(), _stream()), "fv_llm_forward_pooled")
        return pooled

    def set_batch_invariant(self, on: bool = True) -> None:
        """fv_set_batch_invariant: the tower keeps the large-batch kernel forms at every batch size, so an image evaluated alone gets the tokens it
        gets inside a batch (bit for bit); costs ~0.9 ms per one-observation step.  FASTVLA_BATCH_INVARIANT=1 switches it on at creation."""
        _lib.check(self.lib.fv_set_batch_invariant(self.h, int(bool(on))), "fv_set_batch_invariant", self.h)
        self._invariant_on = bool(on)

    def tokens_consumed(self, consumed: bool) -> None:
        """The caller says whether the tower tokens of the calls that follow feed the decoder (splice mode) or are dropped (the literal reference): unless
        FASTVLA_BATCH_INVARIANT pins the choice, batch invariance of the tower follows that."""
        if self._invariant_policy is None and bool(consumed) != self._invariant_on:
            self.set_batch_invariant(bool(consumed))

    def fp16_saturations(self, reset: bool = False) -> int:
        """How many 8-value activation groups the fp16 single-pass projections (llm_precision >= 2) had to clamp to +-65504 since the
        last reset (fv_llm_fp16_saturations): 0 in a healthy model; non-zero means the fp16 budget does not hold for this checkpoint /
        input and the engine should be rebuilt with llm_precision=1.  Synchronises the device."""
        n = C.c_uint64()
        _lib.check(self.lib.fv_llm_fp16_saturations(self.h, C.byref(n), int(reset)), "fv_llm_fp16_saturations", self.h)
        return int(n.value)

    # ---------------------------------------------------------------- image-prefix reuse
    def llm_prefix(self, img_tokens: torch.Tensor) -> torch.Tensor:
        """(B, Ni, H) f32 image tokens -> the decoder's prefix cache: every layer's un-rotated [k | v] rows of the Ni image positions,
        (layers, B, Ni, 2 * kv_heads * head_dim) f32.  Rows of different images are independent: slices along B can be kept and
        re-assembled per frame."""
        B, Ni, _ = img_tokens.shape
        l = self.model.llm
        self.ensure_workspace(B, 1, True)
        kv = torch.empty(l.layers, B, Ni, 2 * l.kv_heads * l.head_dim, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.fv_llm_prefix(self.h, img_tokens.contiguous().data_ptr(), Ni, B, kv.data_ptr(), _stream()), "fv_llm_prefix", self.h)
        return kv

    def llm_pooled_prefixed(self, ids: torch.Tensor, lens: torch.Tensor, kv: torch.Tensor) -> torch.Tensor:
        """The text positions alone against a prefix cache from llm_prefix(): the pooled rows of llm_pooled(ids, lens, img_tokens)."""
        B, T = ids.shape
        if kv.ndim != 4 or kv.shape[1] != B:
            raise ValueError(f"prefix cache must be (layers, B={B}, Ni, 2*kv_dim), got {tuple(kv.shape)}")
        ids = ids.to(device=self.device, dtype=torch.int32).contiguous()
        lens = lens.to(device=self.device, dtype=torch.int32).contiguous()
        self.ensure_workspace(B, T, True)
        pooled = torch.empty(B, self.model.llm.hidden, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.fv_llm_forward_pooled_prefixed(self.h, ids.data_ptr(), lens.data_ptr(), kv.contiguous().data_ptr(), kv.shape[2], B, T, 0,
                                                           pooled.data_ptr(), _stream()), "fv_llm_forward_pooled_prefixed", self.h)
        return pooled

    def backbone(self, images: Optional[torch.Tensor], ids: torch.Tensor, lens: torch.Tensor, *, splice: bool = False,
                 run_tower: bool = True, pad_value: float = 0.0, resize_with_padding: bool = True,
                 pool_mode: int = 0, pix: Optional[torch.Tensor] = None) -> torch.Tensor:
        """letterbox -> tower -> projector -> decoder -> pooled (B,H).  splice=False is the literal reference
        behaviour: the image tokens are computed and not consumed.  `pix`: pixels already letterboxed by
        preprocess() ((B,S,S,4) bf16) -- then `images` is not read."""
        B, T = ids.shape
        self.ensure_workspace(B, T, splice)
        if splice or not run_tower or not self.overlap_streams:
            tok = None
            if run_tower or splice:
                if pix is None and self.fused_letterbox:
                    tok = self.vision_forward_images(images, pad_value, resize_with_padding)
                else:
                    if pix is None:
                        pix = self.preprocess(images, pad_value, resize_with_padding)
                    tok = self.vision_forward(pix)
            return self.llm_pooled(ids, lens, tok if splice else None, pool_mode)
        # literal mode: the decoder does not consume the tower's output, so the two run on separate HIP streams (their
        # workspace regions are disjoint); the small-grid decoder kernels fill the tower kernels' tails.
        cur = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            pooled = self.llm_pooled(ids, lens, None, pool_mode)
        if pix is None and self.fused_letterbox:
            self._tok_keepalive = self.vision_forward_images(images, pad_value, resize_with_padding)
        else:
            if pix is None:
                pix = self.preprocess(images, pad_value, resize_with_padding)
            self._tok_keepalive = self.vision_forward(pix)
        cur.wait_stream(self._side)
        pooled.record_stream(cur)
        return pooled

    # ---------------------------------------------------------------- hipGraph capture of the inference step
    def capture_policy_step(self, images: torch.Tensor, ids: torch.Tensor, lens: torch.Tensor, flat_params: torch.Tensor,
                            states: torch.Tensor, *, splice: bool = False, pool_mode: int = 0):
        """Capture letterbox -> tower -> projector -> decoder -> pool -> head as ONE hipGraph (every entry point of the library is
        asynchronous on the caller's stream, allocates nothing and never synchronises: include/fastvla_hip.h).  The given
        tensors become the graph's static inputs: copy new data INTO them (same shapes) and call replay().
        -> (replay, actions): `replay()` launches the graph on the current stream, `actions` (B, A) is its static output.
        At B = 1 the eager step is launch-bound (~350 launches); the graph replays it in one submission."""
        B, T = ids.shape
        ids = ids.to(device=self.device, dtype=torch.int32).contiguous()
        lens = lens.to(device=self.device, dtype=torch.int32).contiguous()
        images, states = images.to(self.device).contiguous(), states.to(self.device, torch.float32).contiguous()
        self.ensure_workspace(B, T, splice)
        saved = self.head_saved(B)

        def step():
            pooled = self.backbone(images, ids, lens, splice=splice, pool_mode=pool_mode)
            act, _ = self.head_forward(flat_params, pooled, states, saved=saved)
            return act

        side = torch.cuda.Stream(device=self.device)   # warm-up off the default stream: one-time attribute calls, workspace, side stream
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            actions = step()
        return graph.replay, actions

    # ---------------------------------------------------------------- action expert
    def head_numel(self) -> int:
        return self.head_offsets[12]

    def head_views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """name -> view into the flat buffer (state-dict order, each tensor 16-byte aligned)."""
        hd = self.head_dims
        shapes = [(hd["ds"],), (hd["ds"],), (hd["hid"], hd["ds"]), (hd["hid"],), (hd["fus"], hd["feat"] + hd["hid"]),
                  (hd["fus"],), (hd["fus"],), (hd["fus"],), (hd["fus"], hd["fus"]), (hd["fus"],), (hd["da"], hd["fus"]),
                  (hd["da"],)]
        out = {}
        for k, shp, off in zip(HEAD_KEYS, shapes, self.head_offsets):
            n = 1
            for s in shp:
                n *= s
            out[k] = flat[off:off + n].view(*shp)
        return out

    def head_saved(self, B: int) -> torch.Tensor:
        n = C.c_size_t()
        _lib.check(self.lib.fv_head_saved_bytes(self.h, B, C.byref(n)), "fv_head_saved_bytes")
        return torch.empty(n.value // 4, dtype=torch.float32, device=self.device)

    def head_forward(self, flat_params: torch.Tensor, pooled: torch.Tensor, states: torch.Tensor, *, training: bool = False,
                     dropout_p: float = 0.0, seed: int = 0, offset: int = 0, saved: Optional[torch.Tensor] = None,
                     normalized_actions: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """normalized_actions: keep the output in normalised action space even without dropout (loss / training paths when
        set_io_norm folded the dataset statistics in)."""
        B = pooled.shape[0]
        states = states.to(device=self.device, dtype=torch.float32).contiguous()
        pooled = pooled.contiguous()
        if states.shape != (B, self.head_dims["ds"]):
            raise ValueError(f"states must be (B,{self.head_dims['ds']}), got {tuple(states.shape)}")
        saved = saved if saved is not None else self.head_saved(B)
        actions = torch.empty(B, self.head_dims["da"], dtype=torch.float32, device=self.device)
        # 1 = dropout active, 2 = training-mode arithmetic without dropout (folded action statistics are NOT applied), 0 = inference
        mode = 0 if not (training or normalized_actions) else (1 if training and dropout_p > 0.0 else 2)
        _lib.check(self.lib.fv_head_forward(self.h, flat_params.data_ptr(), pooled.data_ptr(), states.data_ptr(), B,
                                            mode, float(dropout_p), seed, offset, actions.data_ptr(),
                                            saved.data_ptr(), _stream()), "fv_head_forward")
        return actions, saved

    def set_io_norm(self, state_mean=None, state_std=None, action_mean=None, action_std=None, eps: float = 1e-8) -> None:
        """Fold the dataset's MEAN_STD statistics into the head kernels (fv_head_set_io_norm); all None switches it off."""
        vecs = [state_mean, state_std, action_mean, action_std]
        if all(v is None for v in vecs):
            _lib.check(self.lib.fv_head_set_io_norm(self.h, None, None, None, None, float(eps)), "fv_head_set_io_norm", self.h)
            return
        dims = [self.head_dims["ds"], self.head_dims["ds"], self.head_dims["da"], self.head_dims["da"]]
        keep = []
        for v, n in zip(vecs, dims):
            t = torch.as_tensor(v, dtype=torch.float32).detach().cpu().contiguous().reshape(-1)
            if t.numel() != n:
                raise ValueError(f"normalisation vector has {t.numel()} elements, expected {n}")
            keep.append(t)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fv_head_set_io_norm(self.h, *[t.data_ptr() for t in keep], float(eps)), "fv_head_set_io_norm", self.h)

    def head_backward(self, flat_params: torch.Tensor, actions: torch.Tensor, targets: torch.Tensor, saved: torch.Tensor,
                      dropout_p: float = 0.0, flat_grads: Optional[torch.Tensor] = None):
        B = actions.shape[0]
        self.ensure_workspace(B, 1, False)
        targets = targets.to(device=self.device, dtype=torch.float32).contiguous()
        if flat_grads is None:
            flat_grads = torch.zeros(self.head_numel(), dtype=torch.float32, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.fv_head_mse_backward(self.h, flat_params.data_ptr(), actions.data_ptr(), targets.data_ptr(), B,
                                                 float(dropout_p), saved.data_ptr(), loss.data_ptr(),
                                                 flat_grads.data_ptr(), _stream()), "fv_head_mse_backward")
        return loss, flat_grads

    def head_backward_from_grad(self, flat_params: torch.Tensor, grad_actions: torch.Tensor, saved: torch.Tensor,
                                dropout_p: float = 0.0, flat_grads: Optional[torch.Tensor] = None) -> torch.Tensor:
        """generic backward: dL/dactions (B,A) -> all 12 head grads in one flat buffer (overwritten)."""
        B = grad_actions.shape[0]
        self.ensure_workspace(B, 1, False)
        grad_actions = grad_actions.to(device=self.device, dtype=torch.float32).contiguous()
        if flat_grads is None:
            flat_grads = torch.zeros(self.head_numel(), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.fv_head_backward(self.h, flat_params.data_ptr(), grad_actions.data_ptr(), B, float(dropout_p),
                                             saved.data_ptr(), flat_grads.data_ptr(), _stream()), "fv_head_backward")
        return flat_grads

    def grad_accumulate(self, acc: torch.Tensor, grads: torch.Tensor) -> None:
        """acc += grads (flat head buffers) on the current stream."""
        _lib.check(self.lib.fv_grad_accumulate(self.h, acc.data_ptr(), grads.data_ptr(), acc.numel(), _stream()),
                   "fv_grad_accumulate", self.h)

    def grad_scale(self, grads: torch.Tensor, scale_dev: torch.Tensor) -> None:
        """grads *= scale_dev[0] (a device scalar) on the current stream."""
        _lib.check(self.lib.fv_grad_scale(self.h, grads.data_ptr(), grads.numel(), scale_dev.data_ptr(), _stream()),
                   "fv_grad_scale", self.h)

    # ---------------------------------------------------------------- unfrozen-backbone training (fv_train_*)
    def train_begin(self) -> None:
        """One-time set-up of the unfrozen decoder + projector training slice (library-owned transposed weight copies)."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fv_train_begin(self.h), "fv_train_begin", self.h)

    _TRAIN_OPTION_DEFAULTS = {"grad_split": 2, "wgrad_f16": True, "loss_scale_log2": 12}

    def train_set_options(self, grad_split: Optional[int] = None, wgrad_f16: Optional[bool] = None, loss_scale_log2: Optional[int] = None,
                          keep: bool = False) -> None:
        """Arithmetic of the backward's contractions (fv_train_set_options).  grad_split: dgrad's gradient operand -- 2 (default) ONE fp16 pass against
        fp16 transposed weights, 1 split bf16 (two passes, the most exact);
        wgrad_f16=False: weight gradients as split-bf16 gradient x bf16 activation (two passes) instead of ONE fp16 pass;
        loss_scale_log2: every gradient of train_forward_backward carries 2^k (train_loss_scale()): divide it out in the optimiser's grad_scale.
        An argument left None takes its default -- or, with keep=True, the value of the previous call (to change one knob only)."""
        base = getattr(self, "_train_options", None) if keep else None
        opts = dict(base or self._TRAIN_OPTION_DEFAULTS)
        for k, v in (("grad_split", grad_split), ("wgrad_f16", wgrad_f16), ("loss_scale_log2", loss_scale_log2)):
            if v is not None:
                opts[k] = v
        _lib.check(self.lib.fv_train_set_options(self.h, int(opts["grad_split"]), int(opts["wgrad_f16"]), int(opts["loss_scale_log2"])),   # wgrad_f16: False / True
                   "fv_train_set_options", self.h)
        self._train_options = opts

    def train_set_forward_f16(self, on: bool = True) -> None:
        """The TRAINING forward's projections in ONE fp16 pass instead of the split-bf16 form's two (fv_train_set_forward_f16); inference is untouched."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fv_train_set_forward_f16(self.h, int(on)), "fv_train_set_forward_f16", self.h)

    def train_loss_scale(self) -> float:
        v = C.c_float()
        _lib.check(self.lib.fv_train_loss_scale(self.h, C.byref(v)), "fv_train_loss_scale", self.h)
        return float(v.value)

    def train_layout(self):
        """-> (tensors, total_numel, n_buckets): every trainable tensor of the ONE flat fp32 buffer, in order: dicts with name, offset,
        numel, rows, cols, bucket (0 head, 1 projector, 2 embedding, 3 + l layer l, 3 + L final norm) and packing (0 plain, 1 q|k|v rows
        concatenated, 2 gate/up rows interleaved by 8)."""
        n, total, nb = C.c_int(), C.c_int64(), C.c_int()
        _lib.check(self.lib.fv_train_layout(self.h, None, 0, C.byref(n), C.byref(total), C.byref(nb)), "fv_train_layout", self.h)
        arr = (_lib.TrainTensor * n.value)()
        _lib.check(self.lib.fv_train_layout(self.h, arr, n.value, C.byref(n), C.byref(total), C.byref(nb)), "fv_train_layout", self.h)
        tensors = [dict(name=t.name.decode(), offset=t.offset, numel=t.numel, rows=t.rows, cols=t.cols, bucket=t.bucket, packing=t.packing) for t in arr]
        return tensors, int(total.value), int(nb.value)

    def train_export_params(self, flat: torch.Tensor) -> None:
        """backbone part of the flat master buffer <- the library's current weights (fv_train_export_params)"""
        _lib.check(self.lib.fv_train_export_params(self.h, flat.data_ptr(), _stream()), "fv_train_export_params", self.h)

    def train_commit(self, flat: torch.Tensor) -> None:
        """the library's bf16 operand copies (and their transposes) <- the master, after an optimiser step (fv_train_commit)"""
        _lib.check(self.lib.fv_train_commit(self.h, flat.data_ptr(), _stream()), "fv_train_commit", self.h)

    def train_workspace(self, B: int, T: int) -> torch.Tensor:
        n = C.c_size_t()
        _lib.check(self.lib.fv_train_workspace_bytes(self.h, B, T, C.byref(n)), "fv_train_workspace_bytes", self.h)
        ws = torch.empty(n.value + 256, dtype=torch.uint8, device=self.device)
        return ws

    def train_named_tensors(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """canonical checkpoint key -> a COPY of that tensor taken out of a flat buffer in the library's packed layout (parameters,
        gradients or Adam moments alike): q / k / v split back out of qkv_proj, gate / up de-interleaved, head tensors under "head.<key>"."""
        tensors, _, _ = self.train_layout()
        l = self.model.llm
        qd, kd = l.heads * l.head_dim, l.kv_heads * l.head_dim
        out = {}
        for t in tensors:
            v = flat[t["offset"]: t["offset"] + t["rows"] * t["cols"]]
            v = v.view(t["rows"], t["cols"]) if t["rows"] > 1 else v
            name = t["name"]
            if t["bucket"] == 0:
                out["head." + name] = v.clone()
            elif t["packing"] == 1:
                base, kind = name.rsplit(".qkv_proj.", 1)
                for nm, r0, r1 in (("q", 0, qd), ("k", qd, qd + kd), ("v", qd + kd, qd + 2 * kd)):
                    out[f"{base}.{nm}_proj.{kind}"] = v[r0:r1].clone()
            elif t["packing"] == 2:
                base = name.rsplit(".gate_up_proj.weight", 1)[0]
                g = v.view(l.inter // 8, 2, 8, l.hidden)
                out[base + ".gate_proj.weight"] = g[:, 0].reshape(l.inter, l.hidden).clone()
                out[base + ".up_proj.weight"] = g[:, 1].reshape(l.inter, l.hidden).clone()
            elif t["packing"] == 3:      # depthwise weight, tap-major [k*k][C] -> [C, 1, k, k]
                k = int(round(t["rows"] ** 0.5))
                out[name] = v.t().reshape(t["cols"], 1, k, k).clone()
            elif t["packing"] == 4:      # the stem's dense 3x3, [27][C0] with row (ky*3+kx)*3+ci -> [C0, 3, 3, 3]
                out[name] = v.view(3, 3, 3, t["cols"]).permute(3, 2, 0, 1).contiguous()
            else:
                out[name] = v.clone()
        return out

    def train_forward_backward(self, flat_params: torch.Tensor, tower_out: torch.Tensor, ids: torch.Tensor, lens: torch.Tensor, states: torch.Tensor,
                               targets: torch.Tensor, ws: torch.Tensor, *, training: bool = True, dropout_p: float = 0.0, seed: int = 0, offset: int = 0,
                               flat_grads: Optional[torch.Tensor] = None, bucket_cb=None):
        """One step's forward + MSE + backward over every trainable tensor (fv_train_forward_backward).  tower_out: (B, Ni, tower_out_dim)
        bf16 from vision_forward(..., return_tower_out=True).  bucket_cb(bucket, offset, numel) is called when a bucket's gradient has
        been enqueued completely.  -> (actions (B, A) in normalised space, loss (1,), flat_grads TIMES train_loss_scale())."""
        B, T = ids.shape
        ids = ids.to(device=self.device, dtype=torch.int32).contiguous()
        lens = lens.to(device=self.device, dtype=torch.int32).contiguous()
        states = states.to(device=self.device, dtype=torch.float32).contiguous()
        targets = targets.to(device=self.device, dtype=torch.float32).contiguous()
        tower_out = tower_out.contiguous()
        if tower_out.dtype != torch.bfloat16 or tower_out.shape != (B, self.model.tower.num_tokens, self.model.tower.out_dim):
            raise ValueError(f"tower_out must be (B, {self.model.tower.num_tokens}, {self.model.tower.out_dim}) bf16, got {tuple(tower_out.shape)} {tower_out.dtype}")
        if flat_grads is None:
            flat_grads = torch.empty_like(flat_params)
        actions = torch.empty(B, self.head_dims["da"], dtype=torch.float32, device=self.device)
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        err = {}

        def _cb(_user, bucket, off, numel):
            if bucket_cb is not None:
                try:
                    bucket_cb(int(bucket), int(off), int(numel))
                except Exception as exc:  # must not unwind through the C frames
                    err.setdefault("exc", exc)

        fn = _lib.BUCKET_CB(_cb)
        base = ws.data_ptr()
        pad = (-base) % 256
        rc = self.lib.fv_train_forward_backward(self.h, flat_params.data_ptr(), tower_out.data_ptr(), ids.data_ptr(), lens.data_ptr(), states.data_ptr(),
                                                targets.data_ptr(), B, T, int(training), float(dropout_p), seed, offset, base + pad, ws.numel() - pad,
                                                actions.data_ptr(), loss.data_ptr(), flat_grads.data_ptr(), fn, None, _stream())
        if "exc" in err:
            raise err["exc"]
        _lib.check(rc, "fv_train_forward_backward", self.h)
        return actions, loss, flat_grads

    # ---------------------------------------------------------------- the tower half of the slice (fv_train_tower_*; csrc/tower_train.inc)
    def train_tower_begin(self) -> None:
        """After train_begin(): the FastViT-HD tower's tensors (inference form) join the flat master -- train_layout() then lists them behind
        "model.norm.weight" (packing 3 = depthwise weights tap-major [k*k][C], 4 = the stem's [27][C0]) with their own gradient buckets."""
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fv_train_tower_begin(self.h), "fv_train_tower_begin", self.h)
        self._tower_training = True

    def train_tower_workspace(self, B: int) -> torch.Tensor:
        n = C.c_size_t()
        _lib.check(self.lib.fv_train_tower_workspace_bytes(self.h, B, C.byref(n)), "fv_train_tower_workspace_bytes", self.h)
        return torch.empty(n.value + 256, dtype=torch.uint8, device=self.device)

    @staticmethod
    def _aligned(ws: torch.Tensor):
        base = ws.data_ptr()
        pad = (-base) % 256
        return base + pad, ws.numel() - pad

    def train_tower_forward(self, pix: torch.Tensor, tws: torch.Tensor) -> torch.Tensor:
        """pixels (B, S, S, 4) bf16 (preprocess()) -> tower_out (B, tokens, tower_out_dim) bf16; every unit's tensors stay in tws for train_tower_backward."""
        B = pix.shape[0]
        t = self.model.tower
        tower_out = torch.empty(B, t.num_tokens, t.out_dim, dtype=torch.bfloat16, device=self.device)
        p, n = self._aligned(tws)
        _lib.check(self.lib.fv_train_tower_forward(self.h, pix.data_ptr(), B, p, n, tower_out.data_ptr(), _stream()), "fv_train_tower_forward", self.h)
        return tower_out

    def train_tower_unit_outputs(self, B: int, tws: torch.Tensor):
        """-> [output of unit 0, 1, ...] (bf16 NHWC) as the last train_tower_forward left them in tws"""
        p, n = self._aligned(tws)
        outs = []
        for u, (_, _, sd, ch) in enumerate(self.tower_units()):
            t = torch.empty(B, sd, sd, ch, dtype=torch.bfloat16, device=self.device)
            _lib.check(self.lib.fv_train_tower_read_unit(self.h, u, B, p, n, t.data_ptr(), _stream()), "fv_train_tower_read_unit", self.h)
            outs.append(t)
        return outs

    def train_set_tower_grad(self, buf: Optional[torch.Tensor]) -> None:
        """bind (or, with None, unbind) the fp16 (B, tokens, tower_out_dim) buffer train_forward_backward leaves dL/d(tower_out) in"""
        _lib.check(self.lib.fv_train_set_tower_grad(self.h, _ptr(buf)), "fv_train_set_tower_grad", self.h)

    def train_tower_backward(self, pix: torch.Tensor, d_tower_out: torch.Tensor, tws: torch.Tensor, flat_grads: torch.Tensor, bucket_cb=None) -> None:
        """the tower's backward from dL/d(tower_out) (fp16, loss-scaled): its gradients into flat_grads at their train_layout() offsets"""
        B = pix.shape[0]
        err = {}

        def _cb(_user, bucket, off, numel):
            if bucket_cb is not None:
                try:
                    bucket_cb(int(bucket), int(off), int(numel))
                except Exception as exc:  # must not unwind through the C frames
                    err.setdefault("exc", exc)

        fn = _lib.BUCKET_CB(_cb)
        p, n = self._aligned(tws)
        rc = self.lib.fv_train_tower_backward(self.h, pix.data_ptr(), d_tower_out.data_ptr(), B, p, n, flat_grads.data_ptr(), fn, None, _stream())
        if "exc" in err:
            raise err["exc"]
        _lib.check(rc, "fv_train_tower_backward", self.h)

    def train_tower_unit(self, unit: int, x_in: torch.Tensor, g_out: torch.Tensor, tws: torch.Tensor, flat_grads: torch.Tensor, gscale: float = 1.0):
        """ONE tower unit, teacher-forced (parity tests): x_in bf16 NHWC (stem: pixels), g_out fp32 NHWC = dL/d(output).
        -> (y bf16 NHWC, g_in fp32 NHWC or None for the stem); the unit's weight gradients x gscale land in flat_grads."""
        B = x_in.shape[0]
        units = self.tower_units()
        t = self.model.tower
        if unit == len(units):
            side, ch = t.tokens_side, t.out_dim
        else:
            side, ch = units[unit][2], units[unit][3]
        y = torch.empty(B, side, side, ch, dtype=torch.bfloat16, device=self.device)
        g_in = None if unit == 0 else torch.empty(tuple(x_in.shape), dtype=torch.float32, device=self.device)
        g_out = g_out.to(device=self.device, dtype=torch.float32).contiguous()
        assert g_out.numel() == y.numel(), (tuple(g_out.shape), tuple(y.shape))
        p, n = self._aligned(tws)
        _lib.check(self.lib.fv_train_tower_unit(self.h, unit, x_in.contiguous().data_ptr(), g_out.data_ptr(), float(gscale), B, p, n, y.data_ptr(), _ptr(g_in),
                                                flat_grads.data_ptr(), _stream()), "fv_train_tower_unit", self.h)
        return y, g_in

    # ---------------------------------------------------------------- RCCL without torch in between (fv_comm_*)
    def comm_unique_id(self) -> bytes:
        rid = _lib.RcclId()
        _lib.check(self.lib.fv_comm_unique_id(self.h, C.byref(rid)), "fv_comm_unique_id", self.h)
        return C.string_at(C.addressof(rid), 128)

    def comm_init(self, uid: bytes, rank: int, world: int) -> int:
        rid = _lib.RcclId()
        C.memmove(C.addressof(rid), uid, 128)
        comm = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fv_comm_init(self.h, C.byref(rid), rank, world, C.byref(comm)), "fv_comm_init", self.h)
        return comm.value

    def comm_destroy(self, comm: int) -> None:
        _lib.check(self.lib.fv_comm_destroy(self.h, comm), "fv_comm_destroy", self.h)

    def allreduce_grads(self, comm: int, flat_grads: torch.Tensor) -> None:
        """one in-place sum all-reduce of the flat head gradient on the CURRENT torch stream (ncclAllReduce over xGMI)."""
        _lib.check(self.lib.fv_allreduce_grads(self.h, comm, flat_grads.data_ptr(), flat_grads.numel(), _stream()),
                   "fv_allreduce_grads", self.h)

    # ---------------------------------------------------------------- profiling (bench.py)
    def profile(self, enable: bool) -> None:
        _lib.check(self.lib.fv_profile(self.h, int(enable)), "fv_profile")

    def profile_read(self):
        """-> (families: name -> dict(ms, flops, bytes, launches), gemm shapes: list of dicts).  Synchronises."""
        fam = (_lib.ProfileEntry * len(_lib.FAMILIES))()
        gem = (_lib.GemmProfile * 128)()
        n = C.c_int()
        _lib.check(self.lib.fv_profile_read(self.h, fam, gem, 128, C.byref(n)), "fv_profile_read")
        fams = {name: dict(ms=fam[i].ms, flops=fam[i].flops, bytes=fam[i].bytes, launches=fam[i].launches)
                for i, name in enumerate(_lib.FAMILIES)}
        shapes = [dict(m=gem[i].m, n=gem[i].n, k=gem[i].k, epi=gem[i].epi, ms=gem[i].ms, launches=gem[i].launches)
                  for i in range(n.value)]
        return fams, shapes

    def adamw_step(self, flat_params, flat_grads, m, v, step: int, *, lr: float, betas=(0.9, 0.95), eps: float = 1e-8,
                   weight_decay: float = 1e-4, max_grad_norm: float = 1.0, grad_scale: float = 1.0,
                   grad_norm_out: Optional[torch.Tensor] = None) -> None:
        hp = _lib.AdamWHParams(lr, betas[0], betas[1], eps, weight_decay, max_grad_norm if max_grad_norm else 0.0,
                               grad_scale)
        _lib.check(self.lib.fv_adamw_clip_step(self.h, flat_params.data_ptr(), flat_grads.data_ptr(), m.data_ptr(),
                                               v.data_ptr(), flat_params.numel(), C.byref(hp), step,
                                               _ptr(grad_norm_out), _stream()), "fv_adamw_clip_step")
