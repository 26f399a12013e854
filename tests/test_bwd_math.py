"""Bit-equivalence of the two formulations of the backward composite's per-(pixel, entry) arithmetic
(splatter360_amd/csrc/s360_bwd_math.h): the scalar reference formulation and the 2-vector one the kernel is built
with (v_pk_mul_f32 / v_pk_add_f32 on gfx950).  The header is plain C++ — the same source is compiled here for the
host, with the library's -ffp-contract=off, and compared bit for bit over 2 million random inputs (including exact
zeros, 1e-30-scale and 1e6-scale values).  IEEE single-precision multiply / add are correctly rounded on both the
host and the GPU, so equivalence of the operation sequences carries over."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _host_clang():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("amdclang++"), shutil.which("clang++")):
        if c and Path(c).exists():
            return c
    return None


def test_packed_and_scalar_backward_entry_math_are_bit_identical(tmp_path):
    cxx = _host_clang()
    if cxx is None:
        pytest.skip("no clang++ (ext_vector_type) on this machine")
    exe = tmp_path / "bwd_math_equiv"
    cmd = [cxx, "-O2", "-std=c++17", "-ffp-contract=off", f"-I{ROOT / 'splatter360_amd' / 'csrc'}",
           str(ROOT / "tests" / "native" / "bwd_math_equiv.cpp"), "-o", str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), "2000000"], capture_output=True, text=True)
    assert r.returncode == 0 and "0 mismatching values" in r.stdout, r.stdout


def test_kernel_is_built_with_the_packed_formulation():
    src = (ROOT / "splatter360_amd" / "csrc" / "s360_backward.hip").read_text()
    assert "#define S360_BWD_ENTRY bwd_entry_packed" in src and "S360_BWD_ENTRY(st, kc," in src
