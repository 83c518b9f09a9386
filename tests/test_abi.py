"""The C-ABI library loads and exports every symbol include/ptrace.h declares; struct layouts
match the header.  No compute calls: this runs without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from pathtracing_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ptrace.h")


def declared_functions(path, prefix):
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions(HEADER, "pt_")
    assert len(names) >= 18
    lib = C.CDLL(_ffi.PTRACE_PATH)
    for n in names:
        assert hasattr(lib, n), f"libptrace.so does not export {n}"
        assert n in _ffi.PTRACE_SYMBOLS, f"{n} is declared in ptrace.h but not bound in _ffi.py"
    assert set(_ffi.PTRACE_SYMBOLS) == set(names)


def test_host_library_exports_its_header():
    names = declared_functions(os.path.join(ROOT, "pathtracing_amd", "host", "ptmaster.h"), "ptm_")
    lib = _ffi.ptmaster()
    for n in names:
        assert hasattr(lib, n), f"libptmaster.so does not export {n}"
    assert set(_ffi.PTMASTER_SYMBOLS) == set(names)


def test_struct_layouts_match_the_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ptrace.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(PtSphere),sizeof(PtMeshObject),sizeof(PtFrameParams),sizeof(PtStats),'
                   'offsetof(PtFrameParams,seed),offsetof(PtFrameParams,tile_x0),offsetof(PtStats,kernel_ms));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(_ffi.PtSphere), C.sizeof(_ffi.PtMeshObject), C.sizeof(_ffi.PtFrameParams), C.sizeof(_ffi.PtStats),
            _ffi.PtFrameParams.seed.offset, _ffi.PtFrameParams.tile_x0.offset, _ffi.PtStats.kernel_ms.offset]
    assert got == want
    assert got[0] == 56 and got[1] == 72          # strides of RayTracingMaster.cs:366 and :410


def test_header_compiles_as_c_and_cxx(tmp_path):
    for comp, ext in (("gcc", "c"), ("g++", "cpp")):
        f = tmp_path / f"t.{ext}"
        f.write_text('#include "ptrace.h"\n#include "ptrace_fmath.h"\nint main(void){return pt_sin(0.0f) != 0.0f;}\n')
        subprocess.run([comp, "-ffp-contract=off", "-mfma", "-I", os.path.join(ROOT, "include"), str(f), "-o", str(tmp_path / "t"), "-lm"], check=True)


def test_version_and_null_handling():
    lib = _ffi.ptrace()
    assert b"libptrace" in lib.pt_version()
    assert lib.pt_render(None, None) != 0
    assert lib.pt_get_stats(None, None) != 0
    lib.pt_destroy(None)     # no-op
    assert lib.pt_last_error(None) is not None


def test_create_without_gpu_fails_cleanly():
    """No silent CPU fallback: without a device pt_create reports an error (skipped where a GPU exists)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import pathtracing_amd as pa\n"
            "try:\n    pa.PathTracer(0)\nexcept pa.PtraceError as e:\n    print('ERR', e)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "ERR" in out.stdout and "pt_create" in out.stdout, out.stdout + out.stderr


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_ffi, "_ptrace", None)
    monkeypatch.setattr(_ffi, "PTRACE_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _ffi.ptrace()
