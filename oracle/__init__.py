"""CPU oracle for the FastVLA policy-step hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain fp32 PyTorch restatement of the arithmetic the reference
(syun88/VLA-from-FastVLM) executes for ``img + prompt (+state) -> action``:

  letterbox (preprocess.py)   <- src/vla_fastvlm/model/fastvlm_adapter.py:36-55, :444-488
  FastViT-HD tower            <- [UNVENDORED] apple ml-fastvlm mobileclip/mci.py `fastvithd`
  mlp2x_gelu projector        <- [UNVENDORED] llava mm_projector; [site] transformers fast_vlm/modeling_fast_vlm.py:39-56
  Qwen2 decoder               <- [site] transformers/models/qwen2/modeling_qwen2.py:35-48,105-135,150-172,195-234,247-252
  pooling                     <- src/vla_fastvlm/model/fastvlm_adapter.py:337-359
  action expert + MSE (+bwd)  <- src/vla_fastvlm/fastvla/fastvlm_with_expert.py:23-54, fastvla/modeling_fastvla.py:46-57
  AdamW / clip / schedules    <- src/vla_fastvlm/training/trainer.py:60-66,171-182,233-244,
                                 src/vla_fastvlm/lerobot_fastvla/configuration_fastvla.py:51-59

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and
only as the checker.  The product path (``vla-from-fastvlm_amd/``) never imports it and has no CPU fallback.

Pinning status
  * preprocess / pooling / head / loss / grads / schedulers / task normalisation / tower-size inference:
    PINNED against the imported reference (tests/golden/*.npz made by tests/golden/make_golden.py).
  * Qwen2 decoder: pinned against the installed third-party ``transformers`` Qwen2 (live, random weights);
    the reference holds no test or fixture for it.
  * FastViT-HD tower, projector and the LLaVA multimodal splice: PARITY UNPINNED.  The arithmetic lives in HF
    remote code (`apple/FastVLM-0.5B:llava_qwen.py`, no pinned revision) and `third_party/ml-fastvlm` holds only a
    LICENSE; neither `timm` nor the remote code is importable here.  The restatement follows the published
    FastViT / FastViT-HD inference-mode (re-parameterised) graph.
"""
from . import preprocess, fastvit_hd, qwen2, head, policy  # noqa: F401
