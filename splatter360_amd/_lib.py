"""Loader / builder of the C-ABI shared library libs360.so (include/s360.h).

The product path has NO fallback: if the HIP library is missing or cannot be loaded, every
operator raises.  `build()` cross-compiles for gfx950 with hipcc (works without a GPU).
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
from pathlib import Path

_PKG = Path(__file__).resolve().parent
_CSRC = _PKG / "csrc"
LIB_PATH = _PKG / "libs360.so"
SOURCES = ("s360_forward.hip", "s360_backward.hip", "s360_backward_em.hip", "s360_stitch.hip", "s360_views.hip", "s360_adapter.hip")
# per-source extra flags (s360_backward_em.hip: see the launcher comment in csrc/s360_bwd_em.h)
SOURCE_FLAGS = {"s360_backward_em.hip": ("-fno-slp-vectorize",)}
HIPCC_FLAGS = ("--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-result")

S360_MAX_VIEWS = 8
FLAG_SHARED_CAMPOS = 1
FLAG_COV9 = 2
FLAG_SH_CHANNEL_MAJOR = 4
FLAG_FORWARD_ONLY = 8
FLAG_SH_DEG4_IGNORED = 16
FLAG_SPHERICAL = 32
FLAG_LEAN_LISTS = 64
FLAG_DEFER_LOSS = 128
FLAG_ATOMIC_GRADS = 256
FLAG_SPLIT_LISTS = 512
FLAG_RAW_INPUTS = 1024
FLAG_COOP_WALK = 2048
ABI_VERSION = 22


class S360Params(C.Structure):
    _fields_ = [("P", C.c_int32), ("V", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("sh_degree", C.c_int32), ("M", C.c_int32), ("flags", C.c_uint32),
                ("max_instances", C.c_uint32), ("max_segments", C.c_uint32), ("_reserved", C.c_uint32),
                ("header_mirror", C.c_void_p)]


class S360Layout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in (
        "total_bytes", "header", "tiles_touched", "vis_mask", "slot_base", "rec_a", "rec_b", "rec_c",
        "clamped", "depths", "tile_count", "slot_ticket", "merge_done", "seg_flag", "seg_arrive", "seg_arrive2", "tile_start", "tile_cursor", "chunk_start", "tile_order", "keys", "keys_alt", "list", "final_T", "n_contrib",
        "tile_max_contrib", "strip_last", "slot_pair", "long_pairs", "rgbc", "sh_jac", "surv", "surv_count", "part_c", "part_t", "part_e", "part_l", "part_n", "seg_c", "seg_t", "seg_cnt", "seg_info", "geo7", "backward_bytes")]


class S360RawInputs(C.Structure):
    """include/s360.h S360RawInputs: the encoder's raw per-pixel outputs (device pointers)."""
    _fields_ = [("extrinsics", C.c_void_p), ("depths", C.c_void_p), ("raw_gaussians", C.c_void_p), ("sh_rotation", C.c_void_p),
                ("n_views", C.c_int32), ("per_view", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("per_ray", C.c_int32),
                ("erp_convention", C.c_int32), ("scale_min", C.c_float), ("scale_max", C.c_float), ("eps", C.c_float)]


EXPORTS = ("s360_forward_raw", "s360_backward_raw", "s360_backward_raw_tail", "s360_abi_version", "s360_error_string", "s360_layout", "s360_forward", "s360_forward_depth", "s360_forward_mse", "s360_backward", "s360_backward_split", "s360_backward_composite", "s360_backward_gaussians", "s360_unpack_gradients", "s360_reduce_unpack_gradients", "s360_sh_backward", "s360_pack_views", "s360_adapter_forward", "s360_adapter_backward", "s360_sh_rotation_blocks",
           "s360_cube2erp_forward", "s360_cube2erp_backward", "s360_count_contributions", "s360_count_backward_slots", "s360_profile_slots", "s360_profile_slot_name",
           "s360_profile_enable", "s360_profile_collect")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found; cannot build libs360.so")


def needs_build() -> bool:
    if not LIB_PATH.exists():
        return True
    t = LIB_PATH.stat().st_mtime
    return any(d.stat().st_mtime > t for d in _dep_files())


def _dep_files():
    """Every file the library is compiled from: csrc/*.hip, csrc/*.h, include/*.h."""
    return sorted(list(_CSRC.glob("*.hip")) + list(_CSRC.glob("*.h")) + list((_PKG.parent / "include").glob("*.h")))


def source_hash() -> str:
    """sha256 (first 16 hex digits) over the kernel sources: stamps profiles/pmc_latest.json so that bench.py only
    quotes counter traffic collected from THIS code."""
    import hashlib
    h = hashlib.sha256()
    for f in _dep_files():
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 ... -> splatter360_amd/libs360.so (in-tree)."""
    if not force and not needs_build():
        return LIB_PATH
    extra = os.environ.get("S360_HIPCC_EXTRA", "").split()
    from concurrent.futures import ThreadPoolExecutor
    objdir = _PKG / "build"
    objdir.mkdir(exist_ok=True)

    def compile_one(src: str):
        obj = objdir / (src + ".o")
        cmd = [_hipcc(), *HIPCC_FLAGS, *SOURCE_FLAGS.get(src, ()), *extra, "-c", str(_CSRC / src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return obj, cmd, r

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        results = list(pool.map(compile_one, SOURCES))
    for obj, cmd, r in results:
        if verbose or r.returncode:
            print(" ".join(cmd))
            print(r.stdout, r.stderr)
        if r.returncode:
            raise RuntimeError("hipcc failed building libs360.so:\n" + r.stderr[-4000:])
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *[str(o) for o, _, _ in results], "-o", str(LIB_PATH)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode:
        print(" ".join(cmd))
        print(r.stdout, r.stderr)
    if r.returncode:
        raise RuntimeError("hipcc failed linking libs360.so:\n" + r.stderr[-4000:])
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """The loaded library.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch FIRST: its wheel bundles its own libamdhip64; if this library were loaded before torch, the system HIP runtime
    # it links against would come up as a second runtime in the process and every launch on torch's streams / memory would
    # fail (seen as "HIP launch / runtime error" when build() and smoke() ran in one process)
    import torch  # noqa: F401
    if not LIB_PATH.exists():
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP rasteriser was not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). There is no CPU fallback.")
    l = C.CDLL(str(LIB_PATH))
    vp, i32, sz = C.c_void_p, C.c_int32, C.c_size_t
    l.s360_abi_version.restype = C.c_int
    l.s360_error_string.restype = C.c_char_p
    l.s360_error_string.argtypes = [C.c_int]
    l.s360_layout.restype = C.c_int
    l.s360_layout.argtypes = [C.POINTER(S360Params), C.POINTER(S360Layout)]
    l.s360_forward.restype = C.c_int
    l.s360_forward.argtypes = [C.POINTER(S360Params)] + [vp] * 9 + [sz, vp]
    l.s360_forward_depth.restype = C.c_int
    l.s360_forward_depth.argtypes = [C.POINTER(S360Params)] + [vp] * 8 + [i32, vp, vp, sz, vp]
    l.s360_forward_mse.restype = C.c_int
    l.s360_forward_mse.argtypes = [C.POINTER(S360Params)] + [vp] * 8 + [i32, vp, vp, C.c_float, vp, vp, vp, vp, sz, vp]
    l.s360_backward.restype = C.c_int
    l.s360_backward.argtypes = [C.POINTER(S360Params)] + [vp] * 7 + [sz] + [vp] * 3 + [i32] + [vp] * 7 + [sz, vp]
    l.s360_backward_split.restype = C.c_int
    l.s360_backward_split.argtypes = [C.POINTER(S360Params)] + [vp] * 6 + [sz] + [vp] * 3 + [i32] + [vp] * 6 + [sz, vp]
    l.s360_backward_composite.restype = C.c_int
    l.s360_backward_composite.argtypes = [C.POINTER(S360Params), vp, vp, sz, vp, vp, vp, i32, vp, sz, vp]
    l.s360_backward_gaussians.restype = C.c_int
    l.s360_backward_gaussians.argtypes = [C.POINTER(S360Params), vp, vp, vp, vp, vp, sz, i32, i32, i32, i32, i32, vp, vp, vp, vp, sz, vp]
    l.s360_reduce_unpack_gradients.restype = C.c_int
    l.s360_reduce_unpack_gradients.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp]
    l.s360_unpack_gradients.restype = C.c_int
    l.s360_unpack_gradients.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    l.s360_sh_backward.restype = C.c_int
    l.s360_sh_backward.argtypes = [C.POINTER(S360Params), i32] + [vp] * 5
    l.s360_forward_raw.restype = C.c_int
    l.s360_forward_raw.argtypes = [C.POINTER(S360Params), vp, C.POINTER(S360RawInputs)] + [vp] * 5 + [i32, vp, vp, C.c_float] + [vp] * 4 + [sz, vp]
    l.s360_backward_raw_tail.restype = C.c_int
    l.s360_backward_raw_tail.argtypes = [C.POINTER(S360Params), vp, i32, C.POINTER(S360RawInputs), vp, vp, sz] + [vp] * 6
    l.s360_backward_raw.restype = C.c_int
    l.s360_backward_raw.argtypes = [C.POINTER(S360Params), vp, C.POINTER(S360RawInputs)] + [vp] * 4 + [sz] + [vp] * 3 + [i32, i32] + [vp] * 7 + [sz, vp]
    l.s360_pack_views.restype = C.c_int
    l.s360_pack_views.argtypes = [vp] * 5 + [i32, i32, i32, vp, vp]
    f32 = C.c_float
    l.s360_adapter_forward.restype = C.c_int
    l.s360_adapter_forward.argtypes = [vp] * 4 + [i32] * 6 + [f32] * 3 + [vp, vp, i32, vp, vp, vp, i32, vp]
    l.s360_adapter_backward.restype = C.c_int
    l.s360_adapter_backward.argtypes = [vp] * 4 + [i32] * 6 + [f32] * 3 + [vp, vp, i32, vp, vp, vp, i32, vp]
    l.s360_sh_rotation_blocks.restype = C.c_int
    l.s360_sh_rotation_blocks.argtypes = [vp, i32, i32, i32, vp, vp]
    l.s360_cube2erp_forward.restype = C.c_int
    l.s360_cube2erp_forward.argtypes = [vp, vp, vp, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(C.c_int64), vp]
    l.s360_cube2erp_backward.restype = C.c_int
    l.s360_cube2erp_backward.argtypes = [vp, vp, vp, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(C.c_int64), vp]
    l.s360_count_backward_slots.restype = C.c_int
    l.s360_count_backward_slots.argtypes = [C.POINTER(S360Params), vp, sz, vp, vp]
    l.s360_count_contributions.restype = C.c_int
    l.s360_count_contributions.argtypes = [C.POINTER(S360Params), vp, sz, vp, vp]
    l.s360_profile_slot_name.restype = C.c_char_p
    l.s360_profile_slot_name.argtypes = [C.c_int]
    l.s360_profile_enable.argtypes = [C.c_int]
    l.s360_profile_collect.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    if l.s360_abi_version() != ABI_VERSION:
        raise RuntimeError("libs360.so ABI version mismatch")
    _lib = l
    return l


def check(code: int, what: str) -> None:
    if code != 0:
        raise RuntimeError(f"{what} failed: {lib().s360_error_string(code).decode()} ({code})")


def layout(prm: S360Params) -> S360Layout:
    out = S360Layout()
    check(lib().s360_layout(C.byref(prm), C.byref(out)), "s360_layout")
    return out


def profile_enable(on: bool) -> None:
    check(lib().s360_profile_enable(int(bool(on))), "s360_profile_enable")


def profile_collect() -> dict:
    """{slot_name: (total_ms, launches)} since the previous collect (host-synchronising)."""
    n = lib().s360_profile_slots()
    ms = (C.c_float * n)()
    calls = (C.c_int32 * n)()
    check(lib().s360_profile_collect(ms, calls), "s360_profile_collect")
    return {lib().s360_profile_slot_name(i).decode(): (float(ms[i]), int(calls[i])) for i in range(n)}
