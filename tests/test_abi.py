"""CPU: the C-ABI library loads and exports every symbol include/fastvla_hip.h declares; the ctypes table matches the
header's argument counts.  No compute call is made (there is no GPU here)."""
import re
from pathlib import Path

import pytest

import fastvla_hip
from fastvla_hip import _lib

HEADER = Path(__file__).resolve().parent.parent / "include" / "fastvla_hip.h"


def _declared():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|void|const char\*)\s+(fv_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return out


def test_every_declared_symbol_is_exported_and_bound():
    decl = _declared()
    assert len(decl) >= 25
    lib = fastvla_hip.load()
    for name, nargs in decl.items():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes prototype"
        assert len(_lib.SIGNATURES[name][1]) == nargs, f"{name}: header has {nargs} args, ctypes table {len(_lib.SIGNATURES[name][1])}"
    assert set(_lib.SIGNATURES) == set(decl)


def test_version_and_error_string_without_gpu():
    lib = fastvla_hip.load()
    assert b"gfx950" in lib.fv_version()
    assert lib.fv_op_gemm(None, 8, None, 8, 8, 8, None, None, None, 0, None, 8, 0, None) == -1  # argument check only
    assert b"null" in lib.fv_last_error(None)


def test_structs_match_header_layout():
    import ctypes as C
    assert C.sizeof(_lib.TensorDesc) == 8 + 8 + 4 + 4 + 32 + 4 + 4 and C.sizeof(_lib.RcclId) == 128
    assert C.sizeof(_lib.AdamWHParams) == 28
    assert C.sizeof(_lib.ProfileEntry) == 32 and C.sizeof(_lib.GemmProfile) == 32
    assert C.sizeof(_lib.ModelDesc) == 4 * (7 + 2 + 1 + 3 * 8 + 4 + 2 + 1 + 4 + 4)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setenv("FASTVLA_HIP_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_LIB", None)
    with pytest.raises(fastvla_hip.FastVLAHipError):
        _lib.load()
    monkeypatch.delenv("FASTVLA_HIP_LIB")
    monkeypatch.setattr(_lib, "_LIB", None)
    _lib.load()
