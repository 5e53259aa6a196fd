"""fastvla_hip: Python binding of libfastvla_hip.so (hand-written HIP kernels for gfx950) -- the compute half of the
MI355X-native FastVLA policy path.  Importing this package never touches the GPU; creating a FastVLAEngine does, and
raises FastVLAHipError when the library or the device is missing (there is no CPU fallback)."""
from ._lib import FastVLAHipError, library_path, load  # noqa: F401
from .arch import LLMConfig, ModelConfig, TowerConfig, PRESETS, preset  # noqa: F401
from .engine import HEAD_KEYS, FastVLAEngine  # noqa: F401
from . import weights  # noqa: F401
