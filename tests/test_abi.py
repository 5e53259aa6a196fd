"""CPU: the C-ABI library loads and exports every symbol include/fastvla_hip.h declares; the ctypes table matches the
header's argument counts.  No compute call is made (there is no GPU here)."""
import re
from pathlib import Path

import pytest

import fastvla_hip
from fastvla_hip import _lib

INCLUDE = Path(__file__).resolve().parent.parent / "include"
HEADER, OPS_HEADER = INCLUDE / "fastvla_hip.h", INCLUDE / "fastvla_hip_testops.h"


def _declared(header=HEADER):
    text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|void|const char\*)\s+(fv_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return out


def _exports(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", str(path)], check=True, capture_output=True, text=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("fv_")}


def test_every_declared_symbol_is_exported_and_bound():
    decl = _declared()
    assert len(decl) >= 25
    lib = fastvla_hip.load()
    for name, nargs in decl.items():
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes prototype"
        assert len(_lib.SIGNATURES[name][1]) == nargs, f"{name}: header has {nargs} args, ctypes table {len(_lib.SIGNATURES[name][1])}"
    assert set(_lib.SIGNATURES) == set(decl)
    # the product ABI is exactly the header: nothing else named fv_* leaves the library -- in particular none of the tests' op-level entry points
    exp = _exports(fastvla_hip.library_path())
    assert exp == set(decl), (sorted(exp - set(decl)), sorted(set(decl) - exp))
    assert not any(n.startswith("fv_op_") for n in exp | set(decl) | set(_lib.SIGNATURES))


def test_test_only_op_library_matches_its_header():
    """the fv_op_* entry points the parity tests drive (25 at the time, 28 now) live in a library of their own (tests/_native/, csrc/ops_api.hip), declared in
    include/fastvla_hip_testops.h; its fv::launch_* references bind to the loaded product library."""
    decl = _declared(OPS_HEADER)
    assert len(decl) >= 20 and all(n.startswith("fv_op_") for n in decl)
    both = _lib.load_testops()
    for name, nargs in decl.items():
        assert hasattr(both, name), f"{name} declared in the test-ops header but not exported"
        assert len(_lib.OPS_SIGNATURES[name][1]) == nargs, f"{name}: header has {nargs} args, ctypes table {len(_lib.OPS_SIGNATURES[name][1])}"
    assert set(_lib.OPS_SIGNATURES) == set(decl) == _exports(_lib.testops_path())
    assert not hasattr(fastvla_hip.load(), "fv_op_gemm")


def test_version_and_error_string_without_gpu():
    lib = fastvla_hip.load()
    assert b"gfx950" in lib.fv_version()
    assert _lib.load_testops().fv_op_gemm(None, 8, None, 8, 8, 8, None, None, None, 0, None, 8, 0, None) == -1  # argument check only
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
    monkeypatch.setattr(_lib, "_OPS", None)
    _lib.load()


def test_convffn32_chunk_loop_has_no_register_file_copies(tmp_path):
    """Build-time guard for csrc/convffn32.hip.  Its MFMAs are inline asm, so hipcc pads no hazard around them: if the register
    allocator ever decides to keep an MFMA operand in the other half of the register file and copy it over per use
    (v_accvgpr_write / _read in the chunk loop), the MFMA behind the copy reads a stale operand now and then -- results that
    differ from run to run in the last bit (seen once, with the x fragments loaded through a buffer descriptor at C = 192).
    The chunk loop (the innermost loop holding the chunk's MFMAs: 48 per wave with four waves per block, 24 in the eight-wave
    C = 192 kernel) must contain no such copy and no scratch access."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc")
    if hipcc is None:
        pytest.skip("hipcc not on PATH")
    src = Path(__file__).resolve().parent.parent / "vla-from-fastvlm_amd" / "csrc" / "convffn32.hip"
    out = tmp_path / "convffn32.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-mllvm", "-pragma-unroll-threshold=4000000", "-S",
                    "--cuda-device-only", str(src), "-o", str(out)], check=True, capture_output=True)
    text = out.read_text()
    kernels = re.findall(r"^(_ZN2fv[^\n]*convffn32_kernel[^\n:]*):[^\n]*\n(.*?)s_endpgm", text, flags=re.S | re.M)
    assert len(kernels) == 9   # three widths, each also as the hidden-range (PART) instance of the control-loop path and as the STASH instance of the training forward
    for name, body in kernels:
        lines = body.split("\n")
        starts = [i for i, l in enumerate(lines) if "Inner Loop Header: Depth=2" in l]
        assert starts, name
        s0 = starts[0]
        end = next(i for i in range(s0 + 100, len(lines)) if re.search(r"s_cbranch_scc[01] \.LBB", lines[i]))
        loop = lines[s0:end]
        assert sum("v_mfma_f32_32x32x16_bf16" in l for l in loop) == (24 if "ILi192ELi1ELi8E" in name else 48), name
        assert not [l for l in loop if "v_accvgpr" in l], name
        assert not [l for l in loop if "scratch_" in l], name
    assert ".vgpr_spill_count: 0" in text or "vgpr_spill_count" not in text
