"""ctypes front-end of the CPU oracle (oracle/s360_oracle.c).

TEST INFRASTRUCTURE — only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  PARITY UNPINNED for the rasteriser internals (see s360_oracle.c header):
the reference's rasteriser is an un-vendored pip dependency
(/root/reference/requirements.txt:17, call site src/model/decoder/cuda_splatting.py:99-124).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIBS: dict = {}


def build(force: bool = False) -> None:
    """Compile liboracle_f32.so / liboracle_f64.so with gcc (oracle/Makefile)."""
    need = force or not all((_HERE / n).exists() for n in ("liboracle_f32.so", "liboracle_f64.so"))
    src_m = (_HERE / "s360_oracle.c").stat().st_mtime
    for n in ("liboracle_f32.so", "liboracle_f64.so"):
        f = _HERE / n
        if f.exists() and f.stat().st_mtime < src_m:
            need = True
    if need:
        subprocess.run(["make", "-C", str(_HERE), "-B"], check=True, capture_output=True)


def _lib(dtype):
    dtype = np.dtype(dtype)
    key = dtype.itemsize
    if key in _LIBS:
        return _LIBS[key]
    build()
    name = "liboracle_f32.so" if key == 4 else "liboracle_f64.so"
    lib = C.CDLL(str(_HERE / name))
    assert lib.orc_real_bytes() == key
    lib.orc_create.restype = C.c_void_p
    lib.orc_create.argtypes = [C.c_void_p] * 6
    lib.orc_destroy.argtypes = [C.c_void_p]
    lib.orc_forward.restype = C.c_uint64
    lib.orc_forward.argtypes = [C.c_void_p]
    lib.orc_num_rendered.restype = C.c_uint64
    lib.orc_num_rendered.argtypes = [C.c_void_p]
    lib.orc_backward.argtypes = [C.c_void_p] * 8
    lib.orc_set_parallel_backward.argtypes = [C.c_int]
    for fn in ("radii", "tiles_touched", "offsets", "rect", "xy", "depth", "conic_opacity", "rgb",
               "clamped", "keys", "values", "ranges", "image", "final_T", "n_contrib",
               "grad_xy_pix", "grad_conic", "grad_opacity_raster", "grad_rgb"):
        f = getattr(lib, "orc_" + fn)
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p]
    _LIBS[key] = lib
    return lib


def _params_struct(real):
    class OrcParams(C.Structure):
        _fields_ = [("P", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("sh_degree", C.c_int32),
                    ("M", C.c_int32), ("use_sh", C.c_int32), ("tanfovx", real), ("tanfovy", real),
                    ("bg", real * 3), ("viewmatrix", real * 16), ("projmatrix", real * 16),
                    ("campos", real * 3), ("spherical", C.c_int32)]
    return OrcParams


def _view(ptr, shape, dtype):
    n = int(np.prod(shape))
    if n == 0:
        return np.zeros(shape, dtype)
    buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape).copy()


class OracleRasterizer:
    """One rasteriser invocation: forward (all intermediates exposed) and backward."""

    def __init__(self, *, image_height, image_width, tanfovx, tanfovy, bg, viewmatrix, projmatrix,
                 sh_degree, campos, means3D, cov3D_precomp, opacities, shs=None,
                 colors_precomp=None, dtype=np.float32, spherical=False):
        self.dt = np.dtype(dtype)
        real = C.c_float if self.dt.itemsize == 4 else C.c_double
        self.lib = _lib(self.dt)
        a = lambda x, shape=None: np.ascontiguousarray(np.asarray(x, dtype=self.dt).reshape(shape) if shape else np.asarray(x, dtype=self.dt))
        self.means = a(means3D)
        self.P = int(self.means.shape[0]) if self.means.ndim == 2 else 0
        self.means = self.means.reshape(self.P, 3)
        self.cov6 = a(cov3D_precomp).reshape(self.P, 6)
        self.opac = a(opacities).reshape(self.P)
        self.use_sh = shs is not None
        if self.use_sh:
            self.shs = a(shs).reshape(self.P, -1, 3)
            self.M = int(self.shs.shape[1])
            self.colors = None
        else:
            self.colors = a(colors_precomp).reshape(self.P, 3)
            self.shs = None
            self.M = 0
        self.H, self.W = int(image_height), int(image_width)
        PS = _params_struct(real)
        prm = PS()
        prm.P, prm.H, prm.W = self.P, self.H, self.W
        prm.sh_degree, prm.M, prm.use_sh = int(sh_degree), self.M, int(self.use_sh)
        prm.tanfovx, prm.tanfovy = float(tanfovx), float(tanfovy)
        prm.bg[:] = [float(v) for v in np.asarray(bg).reshape(3)]
        prm.viewmatrix[:] = [float(v) for v in np.asarray(viewmatrix, dtype=self.dt).reshape(16)]
        prm.projmatrix[:] = [float(v) for v in np.asarray(projmatrix, dtype=self.dt).reshape(16)]
        prm.campos[:] = [float(v) for v in np.asarray(campos, dtype=self.dt).reshape(3)]
        prm.spherical = int(bool(spherical))
        self.NP = 2 * self.P if spherical else self.P      # rasterised pairs (spherical: Gaussian g + its seam ghost P + g)
        self._prm = prm
        p = lambda arr: arr.ctypes.data if arr is not None else None
        self.h = self.lib.orc_create(C.addressof(prm), p(self.means), p(self.cov6), p(self.opac),
                                     p(self.shs), p(self.colors))
        self.gx = (self.W + 15) // 16
        self.gy = (self.H + 15) // 16

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def forward(self) -> dict:
        L = int(self.lib.orc_forward(self.h))
        P, H, W, dt, lib, h = self.NP, self.H, self.W, self.dt, self.lib, self.h
        out = {
            "num_rendered": L,
            "radii": _view(lib.orc_radii(h), (P,), np.int32),
            "tiles_touched": _view(lib.orc_tiles_touched(h), (P,), np.uint32),
            "offsets": _view(lib.orc_offsets(h), (P,), np.uint32),
            "rect": _view(lib.orc_rect(h), (P, 4), np.int32),
            "xy": _view(lib.orc_xy(h), (P, 2), dt),
            "depth": _view(lib.orc_depth(h), (P,), dt),
            "conic_opacity": _view(lib.orc_conic_opacity(h), (P, 4), dt),
            "rgb": _view(lib.orc_rgb(h), (P, 3), dt),
            "clamped": _view(lib.orc_clamped(h), (P, 3), np.uint8),
            "keys": _view(lib.orc_keys(h), (L,), np.uint64),
            "values": _view(lib.orc_values(h), (L,), np.uint32),
            "ranges": _view(lib.orc_ranges(h), (self.gx * self.gy, 2), np.uint32),
            "image": _view(lib.orc_image(h), (3, H, W), dt),
            "final_T": _view(lib.orc_final_T(h), (H, W), dt),
            "n_contrib": _view(lib.orc_n_contrib(h), (H, W), np.uint32),
        }
        return out

    def backward(self, dL_dimage) -> dict:
        P, dt = self.P, self.dt
        g = np.ascontiguousarray(np.asarray(dL_dimage, dtype=dt).reshape(3, self.H, self.W))
        n = max(P, 1)
        d_means3D = np.zeros((n, 3), dt)
        d_means2D = np.zeros((n, 3), dt)
        d_cov6 = np.zeros((n, 6), dt)
        d_op = np.zeros((n,), dt)
        d_sh = np.zeros((n, max(self.M, 1), 3), dt) if self.use_sh else None
        d_col = None if self.use_sh else np.zeros((n, 3), dt)
        p = lambda arr: arr.ctypes.data if arr is not None else None
        self.lib.orc_backward(self.h, p(g), p(d_means3D), p(d_means2D), p(d_cov6), p(d_sh), p(d_col), p(d_op))
        lib, h = self.lib, self.h
        return {
            "means3D": d_means3D[:P], "means2D": d_means2D[:P], "cov3D": d_cov6[:P],
            "opacities": d_op[:P].reshape(P, 1),
            "shs": None if d_sh is None else d_sh[:P, : self.M],
            "colors_precomp": None if d_col is None else d_col[:P],
            "raster_xy_pix": _view(lib.orc_grad_xy_pix(h), (self.NP, 2), dt),
            "raster_conic": _view(lib.orc_grad_conic(h), (self.NP, 3), dt),
            "raster_opacity": _view(lib.orc_grad_opacity_raster(h), (self.NP,), dt),
            "raster_rgb": _view(lib.orc_grad_rgb(h), (self.NP, 3), dt),
        }


def set_parallel_backward(on: bool, dtype=np.float32) -> None:
    """Timing aid for the CPU baseline: OpenMP + atomics in the oracle's backward (non-deterministic
    summation order).  Parity tests keep the default serial, deterministic mode."""
    _lib(dtype).orc_set_parallel_backward(int(bool(on)))


def rasterize(settings: dict, *, means3D, cov3D_precomp, opacities, shs=None, colors_precomp=None,
              dtype=np.float32, spherical=False) -> OracleRasterizer:
    """Convenience: `settings` holds the GaussianRasterizationSettings fields
    (cuda_splatting.py:99-112) as numpy / python values."""
    return OracleRasterizer(
        image_height=settings["image_height"], image_width=settings["image_width"],
        tanfovx=settings["tanfovx"], tanfovy=settings["tanfovy"], bg=settings["bg"],
        viewmatrix=settings["viewmatrix"], projmatrix=settings["projmatrix"],
        sh_degree=settings["sh_degree"], campos=settings["campos"], means3D=means3D,
        cov3D_precomp=cov3D_precomp, opacities=opacities, shs=shs, colors_precomp=colors_precomp,
        dtype=dtype, spherical=spherical)
