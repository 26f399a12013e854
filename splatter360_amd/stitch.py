"""Cube -> equirectangular stitch on the MI355X (replaces Cube2Equirec, /root/reference/
src/geometry/layers.py:41-116, and fuses change_order, src/model/model_wrapper_erp.py:135-158).

The sampling grid is host-side numpy restated from layers.py:60-106 (same float32/float64 mix so
it is bit-identical; pinned by tests/golden/cube2equirec_*.npz); the gather is one HIP kernel.
"""
from __future__ import annotations

import ctypes as C
from functools import lru_cache

import numpy as np
import torch
from torch import Tensor, nn

from . import _lib

# Cube2Equirec slot order is (F R B L U D); faces rendered in the order (top, front, left, back,
# right, bottom) are mapped by the reference's change_order(): flip faces 0 and 5 on both image
# axes, then slots <- rendered [3, 4, 1, 2, 0, 5].  Bit 3 (value 8) = flipped.
CHANGE_ORDER_FACE_MAP = (3, 4, 1, 2, 0 | 8, 5 | 8)


def face_type_map(equ_h: int, equ_w: int) -> np.ndarray:
    """[H,W] int face slot per ERP pixel (0F 1R 2B 3L 4U 5D), layers.py:60-76."""
    q = equ_w // 4
    shift = 3 * equ_w // 8
    tp = np.roll(np.repeat(np.arange(4), q)[None, :].repeat(equ_h, 0), shift, axis=1)
    lon = np.linspace(-np.pi, np.pi, q) / 4
    top_rows = equ_h // 2 - np.round(np.arctan(np.cos(lon)) * equ_h / np.pi).astype(int)  # rows above -> ceiling
    col_mask = np.arange(equ_h)[:, None] < top_rows[None, :]
    mask = np.roll(np.concatenate([col_mask] * 4, 1), shift, axis=1)
    tp = tp.copy()
    tp[mask] = 4
    tp[mask[::-1]] = 5
    return tp


@lru_cache(maxsize=8)
def sample_grid_numpy(face_w: int, equ_h: int, equ_w: int) -> np.ndarray:
    """[H,W,3] float32 (u, v, face-z) = Cube2Equirec.sample_grid[0,0]  (layers.py:78-106)."""
    tp = face_type_map(equ_h, equ_w)
    f32 = np.float32
    lon = ((np.linspace(0, equ_w - 1, num=equ_w, dtype=f32) + 0.5) / equ_w - 0.5) * 2 * np.pi
    lat = -((np.linspace(0, equ_h - 1, num=equ_h, dtype=f32) + 0.5) / equ_h - 0.5) * np.pi
    lon, lat = np.meshgrid(lon, lat)
    u = np.zeros((equ_h, equ_w), f32)
    v = np.zeros((equ_h, equ_w), f32)
    for i in range(4):  # side faces: tangent-plane coordinates
        m = tp == i
        u[m] = 0.5 * np.tan(lon[m] - np.pi * i / 2)
        v[m] = -0.5 * np.tan(lat[m]) / np.cos(lon[m] - np.pi * i / 2)
    m = tp == 4
    c = 0.5 * np.tan(np.pi / 2 - lat[m])
    u[m] = c * np.sin(lon[m])
    v[m] = c * np.cos(lon[m])
    m = tp == 5
    c = 0.5 * np.tan(np.pi / 2 - np.abs(lat[m]))
    u[m] = c * np.sin(lon[m])
    v[m] = -c * np.cos(lon[m])
    u = np.clip(u, -0.5, 0.5) * 2
    v = np.clip(v, -0.5, 0.5) * 2
    z = tp.astype(f32) / 2.5 - 1
    return np.ascontiguousarray(np.stack([u, v, z], -1).astype(f32))


def _face_map_arr(face_map):
    return None if face_map is None else (C.c_int32 * 6)(*face_map)


class _Stitch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, faces, grid, face_map, strides, channels, face_w):
        if not faces.is_cuda:
            raise RuntimeError("cube->ERP stitch runs on the GPU only (no CPU path)")
        eh, ew = int(grid.shape[0]), int(grid.shape[1])
        x = faces.detach().float()
        if strides is None:
            x = x.contiguous()
        erp = torch.empty((channels, eh, ew), dtype=torch.float32, device=faces.device)
        with torch.cuda.device(faces.device):
            st = C.c_void_p(torch.cuda.current_stream(faces.device).cuda_stream)
            sarr = None if strides is None else (C.c_int64 * 3)(*strides)
            rc = _lib.lib().s360_cube2erp_forward(C.c_void_p(x.data_ptr()), C.c_void_p(grid.data_ptr()),
                                                  C.c_void_p(erp.data_ptr()), channels, face_w, eh, ew,
                                                  _face_map_arr(face_map), sarr, st)
        _lib.check(rc, "s360_cube2erp_forward")
        ctx.save_for_backward(grid)
        ctx.meta = (face_map, strides, channels, face_w, tuple(faces.shape))
        return erp

    @staticmethod
    def backward(ctx, d_erp):
        (grid,) = ctx.saved_tensors
        face_map, strides, channels, face_w, shape = ctx.meta
        eh, ew = int(grid.shape[0]), int(grid.shape[1])
        g = d_erp.detach().float().contiguous()
        d_faces = torch.empty((6, channels, face_w, face_w), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            st = C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)
            rc = _lib.lib().s360_cube2erp_backward(C.c_void_p(g.data_ptr()), C.c_void_p(grid.data_ptr()),
                                                   C.c_void_p(d_faces.data_ptr()), channels, face_w, eh, ew,
                                                   _face_map_arr(face_map), None, st)
        _lib.check(rc, "s360_cube2erp_backward")
        if strides is not None:  # input was [C, fw, 6*fw]
            d_faces = d_faces.permute(1, 2, 0, 3).reshape(shape)
        return d_faces, None, None, None, None, None


class Cube2Equirec(nn.Module):
    """Same constructor and forward contract as the reference module (layers.py:41-116):
    forward(cube_feat[B,C,fw,6*fw]) -> [B,C,equ_h,equ_w]; faces side by side in slot order F R B L U D."""

    def __init__(self, face_w: int, equ_h: int, equ_w: int):
        super().__init__()
        self.face_w, self.equ_h, self.equ_w = face_w, equ_h, equ_w
        grid = torch.from_numpy(sample_grid_numpy(face_w, equ_h, equ_w)).view(1, 1, equ_h, equ_w, 3)
        self.sample_grid = nn.Parameter(grid, requires_grad=False)

    def forward(self, cube_feat: Tensor) -> Tensor:
        bs, ch, h, w = cube_feat.shape
        assert h == self.face_w and w // 6 == self.face_w
        grid = self.sample_grid[0, 0]
        x = cube_feat.float().contiguous()
        fw = self.face_w
        strides = (fw, fw * 6 * fw, 6 * fw)
        return torch.stack([_Stitch.apply(x[b], grid, None, strides, ch, fw) for b in range(bs)])

    def stitch_rendered(self, faces: Tensor) -> Tensor:
        """faces[6,C,fw,fw] in the reference's RENDERED order (top, front, left, back, right,
        bottom) -> ERP [C,equ_h,equ_w] = Cube2Equirec(change_order(faces)) without the flip /
        permute / concat copies (model_wrapper_erp.py:393-400)."""
        return _Stitch.apply(faces, self.sample_grid[0, 0], CHANGE_ORDER_FACE_MAP, None, int(faces.shape[1]), self.face_w)
