"""The C-ABI library: loads on a machine without a GPU, exports every symbol include/s360.h
declares, and its host-only entry points behave (no compute calls here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

from splatter360_amd import _lib

ROOT = Path(__file__).resolve().parent.parent


def _declared():
    txt = (ROOT / "include" / "s360.h").read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(s360_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = _declared()
    assert len(names) >= 11 and set(_lib.EXPORTS) == set(names)
    for n in names:
        assert hasattr(lib, n), n
    assert lib.s360_abi_version() == _lib.ABI_VERSION
    assert lib.s360_error_string(0) == b"ok" and lib.s360_error_string(-2) == b"workspace too small"


def test_layout_is_monotone_and_aligned():
    prm = _lib.S360Params(P=1000, V=6, H=64, W=64, sh_degree=4, M=25, flags=1, max_instances=5000)
    lay = _lib.layout(prm)
    offs = [getattr(lay, n) for n, _ in _lib.S360Layout._fields_ if n not in ("total_bytes", "backward_bytes")]
    assert offs == sorted(offs) and all(o % 16 == 0 for o in offs)
    assert (lay.rec_b, lay.rec_c) == (lay.rec_a + 16, lay.rec_a + 32)
    assert lay.total_bytes > offs[-1] and lay.part_c >= lay.surv_count + 6 * 16 * 4 and lay.backward_bytes >= 5000 * 4 * 48
    assert C.sizeof(_lib.S360Params) == 48


def test_bad_arguments_are_rejected_before_any_gpu_work():
    lib = _lib.lib()
    bad = _lib.S360Params(P=10, V=9, H=64, W=64, sh_degree=4, M=25, flags=0, max_instances=100)
    out = _lib.S360Layout()
    assert lib.s360_layout(C.byref(bad), C.byref(out)) == -1
    ok = _lib.S360Params(P=10, V=1, H=64, W=64, sh_degree=4, M=25, flags=0, max_instances=100)
    # null views / images / workspace
    assert lib.s360_forward(C.byref(ok), None, None, None, None, None, None, None, None, None, 0, None) == -1
    assert lib.s360_cube2erp_forward(None, None, None, 3, 64, 128, 256, None, None, None) == -1
    with pytest.raises(RuntimeError):
        _lib.check(-4, "x")


def test_rasterizer_refuses_cpu_tensors():
    import torch
    from splatter360_amd import rasterizer
    views = torch.zeros(1, rasterizer.VIEW_FLOATS)
    with pytest.raises(RuntimeError):
        rasterizer.rasterize_views(torch.zeros(4, 3), torch.zeros(4, 6), torch.zeros(4), colors_precomp=torch.zeros(4, 3),
                                   views=views, image_height=16, image_width=16)
    with pytest.raises(Exception):
        rasterizer.rasterize_views(torch.zeros(4, 3), torch.zeros(4, 6), torch.zeros(4), views=views, image_height=16, image_width=16)


def test_header_is_plain_c_and_a_c_program_can_call_the_library(tmp_path):
    """examples/c_abi_layout.c is compiled as C99 with -pedantic against include/s360.h, linked to libs360.so and
    run: the boundary is a C ABI (no C++ / torch types), usable without Python, and the host-only entry points
    need no GPU."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    _lib.lib()
    root = Path(__file__).resolve().parent.parent
    exe = tmp_path / "c_abi_layout"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", f"-I{root / 'include'}", str(root / "examples" / "c_abi_layout.c"),
           f"-L{root / 'splatter360_amd'}", "-ls360", f"-Wl,-rpath,{root / 'splatter360_amd'}", "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"abi {_lib.ABI_VERSION} " in r.stdout and "bad V -> -1" in r.stdout


def test_ctypes_argtypes_match_the_header_prototypes():
    """Every entry point's ctypes argtypes (splatter360_amd/_lib.py) has as many parameters as its prototype in
    include/s360.h, pointers where the header has pointers and sizes / integers / floats where it has those: a
    signature changed on one side only (the ABI moved twice in round 2) fails here, on CPU."""
    import ctypes as C
    text = re.sub(r"/\*.*?\*/", " ", (ROOT / "include" / "s360.h").read_text(), flags=re.S)
    lib = _lib.lib()
    checked = 0
    for m in re.finditer(r"\b(?:int|const char\s*\*)\s+(s360_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        name, params = m.group(1), m.group(2).strip()
        fn = getattr(lib, name)
        if fn.argtypes is None:
            continue
        plist = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
        assert len(plist) == len(fn.argtypes), f"{name}: header has {len(plist)} parameters, ctypes {len(fn.argtypes)}"
        for p, a in zip(plist, fn.argtypes):
            is_ptr_h = "*" in p
            is_ptr_c = a in (C.c_void_p, C.c_char_p) or hasattr(a, "_type_") and isinstance(a._type_, type) and issubclass(a, C._Pointer)
            assert is_ptr_h == bool(is_ptr_c), f"{name}: parameter '{p}' vs ctypes {a}"
            if not is_ptr_h:
                if "float" in p:
                    assert a is C.c_float, f"{name}: '{p}' vs {a}"
                elif "size_t" in p:
                    assert a in (C.c_size_t, C.c_uint64), f"{name}: '{p}' vs {a}"
                else:
                    assert a in (C.c_int, C.c_int32, C.c_uint32), f"{name}: '{p}' vs {a}"
        checked += 1
    assert checked >= 12


def test_loading_the_library_brings_torch_in_first():
    """libs360.so links the system libamdhip64; torch's wheel bundles its own.  Loaded before torch, the library would start
    a second HIP runtime in the process and every later launch on torch's streams fails (build() + smoke() in one process
    did exactly that).  lib() therefore imports torch before dlopen — checked in a fresh interpreter."""
    import subprocess
    import sys
    code = ("import sys; from splatter360_amd import _lib; assert 'torch' not in sys.modules; "
            "_lib.lib(); assert 'torch' in sys.modules; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], cwd=str(ROOT), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
