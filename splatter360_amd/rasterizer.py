"""PyTorch surface of the MI355X rasteriser: the drop-in `GaussianRasterizationSettings` /
`GaussianRasterizer` pair the reference imports (src/model/decoder/cuda_splatting.py:5-8) and the
multi-view primitive `rasterize_views` behind it (one call = V views of one cloud).

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); all arithmetic happens in
libs360.so through the C ABI of include/s360.h.  There is no eager / CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import NamedTuple, Optional

import torch
from torch import Tensor, nn

from . import _lib

VIEW_FLOATS = 44

# Fallback switch for the one unverifiable constant table: the reference always calls its rasteriser
# fork with sh_degree = 4; if that fork turns out to stop at the public 3DGS degree 3, set this (or the environment
# variable S360_SH_DEG4_IGNORED=1) and coefficients 16..24 are ignored exactly as such a fork would ignore them.
SH_DEG4_IGNORED = bool(int(os.environ.get("S360_SH_DEG4_IGNORED", "0")))


class GaussianRasterizationSettings(NamedTuple):
    """Same fields, same order as upstream (constructed with keywords at cuda_splatting.py:99-112)."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: Tensor
    scale_modifier: float
    viewmatrix: Tensor
    projmatrix: Tensor
    sh_degree: int
    campos: Tensor
    prefiltered: bool
    debug: bool


# Lean tile lists (S360_FLAG_LEAN_LISTS): a (Gaussian, tile) instance is binned only if the splat can reach alpha >= 1/255 somewhere
# on that tile.  Images, radii and all gradients are bit-identical to the upstream-compatible lists; tiles_touched / the sorted
# lists / num_rendered / n_contrib positions are not (they describe fewer instances).  Default on; rasterize_views(lean=False),
# this switch or S360_LEAN_LISTS=0 select upstream's 3-sigma rectangles (what the integer-state parity tests compare).
LEAN_LISTS = bool(int(os.environ.get("S360_LEAN_LISTS", "1")))

# Segment-parallel compositing of long tile lists (S360_FLAG_SPLIT_LISTS, forward and backward): an 8x8 quadrant whose list is longer
# than SEG_HEAD + SEG_MIN_REST = 1 024 + 512 entries and that still holds a pixel far from saturating (T >= 1/16) after its first
# SEG_HEAD = 1 024 entries hands the rest over in SEG_LEN = 512-entry segments, one wave each, combined per pixel in list order from
# the pixel's true incoming transmittance (csrc/s360_device.h; nothing is ever replayed, no wave waits for another).  Quadrants that
# do not hand over are bit-identical either way; inside split quadrants the floating-point association changes (<= 1e-6 per pixel;
# integers, stop decisions and n_contrib stay those of the sequential walk).
# True / False (rasterize_views(split_lists=...), this switch, S360_SPLIT_LISTS=1 / 0) force the mode; both are functions of the
# call's own data.  "auto" (the default) is ADAPTIVE and therefore HISTORY-DEPENDENT: the forward reports, into the caller's pinned
# mirror, whether some quadrant was worth splitting; the flag is set for the calls that follow such a report and dropped after 16
# calls without one — a cloud that never splits (the headline's) then runs the very kernels of a build without the feature (the
# hand-over code costs the forward composite ~10 us, its second launch 3 us, the backward's segment units 12 VGPRs), and the first
# call on a cloud that needs it runs sequentially.  Identical inputs can thus give results 1e-7 apart depending on what was rendered
# before; DETERMINISTIC below removes that.
_sl = os.environ.get("S360_SPLIT_LISTS", "auto")
if _sl not in ("auto", "0", "1"):            # a typo must not surface as a ValueError from int() at import
    raise RuntimeError(f"S360_SPLIT_LISTS must be 'auto', '0' or '1', got {_sl!r}")
SPLIT_LONG_LISTS = "auto" if _sl == "auto" else bool(int(_sl))

# DETERMINISTIC (S360_DETERMINISTIC=1): results are a function of the call's inputs alone, bit for bit, whatever was rendered before —
# every adaptive ("auto") choice that can change bits is pinned: list splitting is always compiled in (S360_FLAG_SPLIT_LISTS set
# wherever "auto" could set it; the hand-over itself is decided inside the kernel from the data) with the library's worst-case
# segment storage (no history-sized max_segments), and check="lazy" calls never size their buffers below the first-call guess.
# Costs the headline step ~1.5 % (the SPLIT instances of both composites).  The binning variant (S360_FLAG_COOP_WALK) stays
# adaptive: its results are bit-identical either way (tests/test_gpu_coop_walk.py).  tests/test_gpu_determinism.py: cloud A, cloud
# B, cloud A again -> torch.equal.
DETERMINISTIC = bool(int(os.environ.get("S360_DETERMINISTIC", "0")))

# check="lazy" calls learn the instance count of their shape one call late from a pinned host word (default_capacity).  By default
# they only ever RAISE the capacity above the first-call guess 1.5 G V — a lazy call can then be truncated only by a scene beyond
# that guess, as in round 4.  LAZY_SHRINK = True (S360_LAZY_SHRINK=1; bench.py sets it and says so) sizes them at 1.25 x the largest
# count seen instead: 1.09 + 1.06 GB of workspace + backward scratch at the headline instead of 2.56 + 2.84 GB, at the price that a
# scene needing > 25 % more instances than any earlier one is truncated for one step (flagged by a RuntimeWarning at the next call
# and counted in overflow_events() — poll it after a step to redo / skip a truncated one).
LAZY_SHRINK = bool(int(os.environ.get("S360_LAZY_SHRINK", "0")))

# Opt-in (S360_FLAG_ATOMIC_GRADS): the backward composite accumulates with float32 atomics instead of the deterministic
# partial-record gather — a quarter of the backward scratch, one launch less, gradients no longer bit-reproducible run to run.
ATOMIC_GRADS = bool(int(os.environ.get("S360_ATOMIC_GRADS", "0")))

# The drop-in GaussianRasterizer reads the binning-overflow flag back after every call ("sync": one host synchronisation, like
# upstream's own scan read-back, and an automatic re-render at the exact size).  S360_DROPIN_CHECK=lazy (or this switch) removes
# that synchronisation from an unchanged reference's per-face loop; the caller then owns the check (last_state().overflowed()).
DROPIN_CHECK = os.environ.get("S360_DROPIN_CHECK", "sync")
if DROPIN_CHECK not in ("sync", "lazy"):     # a typo must not silently select the mode without the overflow re-render
    raise RuntimeError(f"S360_DROPIN_CHECK must be 'sync' or 'lazy', got {DROPIN_CHECK!r}")

DEPTH_MODES = {"depth": 0, "disparity": 1, "relative_disparity": 2, "log": 3}


def pack_views(viewmatrix: Tensor, projmatrix: Tensor, campos: Tensor, tanfovx, tanfovy, bg: Tensor,
               scale=1.0, near=0.0, far=0.0) -> Tensor:
    """-> float32 [V,44] device tensor in S360View layout.  Tensors may carry a leading view dim;
    tanfov / scale / near / far may be python floats or [V] tensors (near / far: the UNSCALED planes, only
    read when a fused depth map is requested).  No host synchronisation."""
    vm = viewmatrix.reshape(-1, 16).float()
    v = vm.shape[0]
    dev = vm.device
    pm = projmatrix.reshape(-1, 16).float().to(dev)
    cp = campos.reshape(-1, 3).float().to(dev)
    b = bg.reshape(-1, 3).float().to(dev).expand(v, 3)

    def col(t):
        if isinstance(t, Tensor):
            return t.reshape(-1, 1).float().to(dev).expand(v, 1)
        return torch.full((v, 1), float(t), dtype=torch.float32, device=dev)

    pad = torch.zeros((v, 1), dtype=torch.float32, device=dev)
    return torch.cat([vm, pm, cp.expand(v, 3), col(tanfovx), col(tanfovy), b, col(scale), col(near), col(far), pad],
                     dim=1).contiguous()


def pack_views_native(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor, background: Tensor,
                      scale_invariant: bool = True) -> Tensor:
    """s360_pack_views: N cameras -> views[N,44] in one kernel launch on the current stream (no host sync)."""
    ext = _f32c(extrinsics, "extrinsics").reshape(-1, 4, 4)
    n = int(ext.shape[0])
    dev = ext.device
    k = _f32c(intrinsics, "intrinsics").reshape(-1, 3, 3)
    nr = _f32c(near, "near").reshape(-1).expand(n).contiguous()
    fr = _f32c(far, "far").reshape(-1).expand(n).contiguous()
    bg = background.detach().float().to(dev).contiguous()
    per_view = bg.dim() == 2 and int(bg.shape[0]) == n
    if not per_view and bg.numel() != 3:
        raise RuntimeError(f"background must be [3] or [{n},3], got {tuple(bg.shape)}")
    if int(k.shape[0]) != n:
        raise RuntimeError("extrinsics / intrinsics view counts differ")
    out = torch.empty((n, VIEW_FLOATS), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = _lib.lib().s360_pack_views(_ptr(ext), _ptr(k), _ptr(nr), _ptr(fr), _ptr(bg), int(bool(per_view)), n,
                                        int(bool(scale_invariant)), _ptr(out), stream)
    _lib.check(rc, "s360_pack_views")
    return out


def pack_views_spherical(pano_c2w: Tensor, background: Tensor, scale=1.0, near=0.0, far=0.0) -> Tensor:
    """[n,4,4] panorama camera-to-world poses (the encoder's ERP frame: +z = panorama centre, +y = top row) -> views[2n,44]
    for rasterize_views(..., spherical=True): rows 2i / 2i+1 are panorama i's camera and its seam ghost (identical
    records).  `scale` multiplies the cloud and the camera centre inside the kernels (pass 1/near for the reference's
    scale-invariant convention: the radial cull sits at 0.2 scaled units).  Small host-free torch glue (one inverse)."""
    c2w = pano_c2w.reshape(-1, 4, 4).float().clone()
    n = c2w.shape[0]
    dev = c2w.device
    sc = scale if isinstance(scale, Tensor) else torch.full((n,), float(scale), device=dev)
    sc = sc.reshape(-1).float().to(dev).expand(n)
    c2w[:, :3, 3] = c2w[:, :3, 3] * sc[:, None]
    vm = torch.linalg.inv_ex(c2w, check_errors=False).inverse.transpose(1, 2).contiguous()
    one = torch.ones(n, device=dev)
    v = pack_views(vm, torch.eye(4, device=dev).expand(n, 4, 4), c2w[:, :3, 3], one, one, background, scale=sc, near=near, far=far)
    return v.repeat_interleave(2, dim=0).contiguous()


_CAPACITY_HINT: dict = {}   # (device, P, V, H, W, lean) -> LARGEST instance count any call of that shape has read back


def _hint_key(dev, p: int, v: int, h: int, w: int, lean: bool):
    d = torch.device(dev) if dev is not None else None
    return (None if d is None else (d.type, d.index), p, v, h, w, bool(lean))


_MIRRORS: dict = {}         # hint key -> pinned int64[1]: (num_instances | overflow << 32) of the latest finished call of that shape


def _mirror(key) -> Tensor:
    """The host-visible word the forward's k_tile_scan stores its instance count and overflow flag into (S360Params.header_mirror):
    one pinned int64 per (device, shape, list mode), never freed (a kernel in flight may still write it).  -1 = nothing yet."""
    m = _MIRRORS.get(key)
    if m is None:
        # [1]: "some quadrant was worth splitting" (k_render); [2]: pairs binned over more than 32 tiles (k_emit's count)
        m = torch.tensor([-1, 0, -1], dtype=torch.int64).pin_memory()
        _MIRRORS[key] = m
    return m


def _poll_mirror(key, warn: bool = True) -> None:
    """Fold what the latest FINISHED call of this shape reported into the capacity hint — a plain host read of pinned memory, no
    device synchronisation; whatever is still in flight is simply not seen yet."""
    m = _MIRRORS.get(key)
    if m is None:
        return
    word = int(m[0])
    if word < 0:
        return
    n, over, chunks = word & 0xFFFFFFFF, (word >> 32) & 1, word >> 33
    _CAPACITY_HINT[key] = max(n, _CAPACITY_HINT.get(key, 0))
    _CHUNK_HINT[key] = max(chunks, _CHUNK_HINT.get(key, 0))
    if over and warn and _OVERFLOW_WARNED.get(key) != n:
        import warnings
        _OVERFLOW_WARNED[key] = n
        _OVERFLOW_EVENTS[0] += 1
        warnings.warn(f"splatter360_amd: a check='lazy' rasteriser call needed {n} instances, more than its binning capacity — that "
                      "call's tile lists were truncated (memory-safe, image incomplete); the following calls are sized for it.  "
                      "Pass max_instances= or use check='sync' where a truncated frame is not acceptable.", RuntimeWarning, stacklevel=3)


_OVERFLOW_WARNED: dict = {}
_OVERFLOW_EVENTS = [0]


def overflow_events() -> int:
    """How many truncated check="lazy" calls have been DETECTED so far in this process (each is detected by the next lazy call of its
    shape, from the pinned mirror: no synchronisation).  A training loop that runs lazy calls with LAZY_SHRINK can compare this
    counter before and after a step's first rasteriser call to learn that the PREVIOUS step rendered a truncated frame."""
    return _OVERFLOW_EVENTS[0]

_CHUNK_HINT: dict = {}      # hint key -> most 4 096-key sort chunks of long tile lists any finished call of that shape reported
SEG_PER_CHUNK = 8           # csrc/s360_device.h: segment slots per sort chunk (4096 / S360_SEG_LEN)


_SPLIT_AGE: dict = {}       # hint key -> calls since the forward last reported a quadrant worth splitting

COOP_WALK = "auto"          # S360_FLAG_COOP_WALK: True / False force; "auto": from the previous call's count of wide rectangles
AUTO_COOP_MIN_PAIRS = 2048  # "auto": at least this many (Gaussian, view) pairs binned over more than 32 tiles in the latest finished
                            # call of the shape (the 1 M uniform cloud: tens of thousands; the encoder-like headline cloud: a few hundred)


def coop_decision(key, mode=None) -> bool:
    """S360_FLAG_COOP_WALK for the next call of this shape (`mode` None = the module switch COOP_WALK): the separately compiled
    binning kernels whose waves walk rectangles of more than 32 tiles cooperatively.  Results are bit-identical either way; the
    variant costs the headline cloud ~5 us and saves a cloud of near, screen-filling splats ~100 us, so "auto"
    follows word 2 of the pinned mirror — a plain host read, one call of delay, never a synchronisation."""
    mode = COOP_WALK if mode is None else mode
    if mode is True or mode is False:
        return mode
    m = _MIRRORS.get(key)
    return m is not None and int(m[2]) >= AUTO_COOP_MIN_PAIRS


AUTO_SPLIT_MAX_WAVES = 24576     # "auto": only where the (tile, quadrant) waves of a call fill the chip at most four times (6 144 resident waves)


def split_decision(key, mode, quadrant_waves: int = 0) -> bool:
    """S360_FLAG_SPLIT_LISTS for the next call of this shape: `mode` itself when it is True / False, else adaptive — on while a
    call of the last 16 reported (through word 1 of the pinned mirror, a plain host read) a quadrant worth splitting, and only for
    calls of at most AUTO_SPLIT_MAX_WAVES quadrant waves: the segment work runs in a second launch, i.e. after ALL tile workgroups —
    at BASELINE configs[4]'s shape (4 M Gaussians, six 512x512 faces: 24 576 waves, four rounds of residency) a long list already
    overlaps with three rounds of other tiles and splitting still measures 768 vs 792 us forward, 570 vs 593 us backward (1 % of
    the step); beyond that shape nothing was measured and the flag stays off."""
    if mode is True or mode is False:
        return mode
    if quadrant_waves > AUTO_SPLIT_MAX_WAVES:
        return False
    if DETERMINISTIC:                 # "auto" pinned: always the SPLIT instances (the hand-over is decided in the kernel, from the data)
        return True
    m = _MIRRORS.get(key)
    if m is not None and int(m[1]) != 0:
        m[1] = 0                      # (a report landing right after this clear is seen by the next call: never lost for long)
        _SPLIT_AGE[key] = 0
    else:
        _SPLIT_AGE[key] = _SPLIT_AGE.get(key, 1 << 30) + 1
    return _SPLIT_AGE[key] <= 16


def default_segments(key) -> int:
    """S360Params.max_segments for a call of this shape: twice the most long-list chunks seen so far (+ slack), in segment slots; 0
    (= the library's worst case, every list of the binning capacity long: ~48 B per instance of capacity) until a count is known.
    Quadrants whose segments do not fit are simply composited sequentially."""
    c = _CHUNK_HINT.get(key)
    return 0 if (c is None or DETERMINISTIC) else SEG_PER_CHUNK * (2 * c + 64)


def default_capacity(p: int, v: int, h: int = 0, w: int = 0, *, device=None, lean: bool = False, lazy: bool = False) -> int:
    """Capacity (instances = (Gaussian, tile) pairs) of the binning buffers and, through them, of the backward scratch
    (per instance of capacity: 24 B of keys / lists / owner table, 192 B of survivor records and 4 x 64 B of quadrant-partial
    slots in the backward scratch).  First-call guess 1.5 P V.  Once an instance count is known for this (device, shape, list
    mode) the size is 1.25 x the LARGEST count seen (a running maximum: one sparse scene never shrinks the buffers of the next,
    denser one); check="lazy" calls keep at least the first-call guess unless LAZY_SHRINK is set (module switch).  Where the count comes from: check="sync" reads it back with the overflow flag (and re-renders an overflowing
    call with the exact size); check="lazy" never synchronises — every forward also stores (count, overflow flag) into a pinned
    host word (S360Params.header_mirror), and the NEXT lazy call of the shape reads that word: the buffers follow the scene with
    one call of delay.  A lazy call whose scene outgrows 1.25 x everything seen before is truncated (flagged, memory-safe, a
    RuntimeWarning at the next call); pass max_instances= to rule that out."""
    first = (3 * p * v) // 2 + (1 << 18)
    guess = first
    key = _hint_key(device, p, v, h, w, lean)
    if lazy:
        _poll_mirror(key)
    hint = _CAPACITY_HINT.get(key)
    if hint is not None:
        guess = hint + hint // 4 + (1 << 16)
        if lazy and (DETERMINISTIC or not LAZY_SHRINK):
            guess = max(guess, first)      # lazy calls cannot be re-rendered: never below the first-call guess unless opted in
    return int(min(2**32 - 1, max(1 << 16, guess)))


class RasterState:
    """Forward workspace + layout: upstream's geomBuffer/binningBuffer/imgBuffer, addressable."""

    def __init__(self, prm: _lib.S360Params, lay: _lib.S360Layout, workspace: Tensor):
        self.prm, self.layout, self.workspace = prm, lay, workspace

    def _arr(self, off: int, count: int, dtype: torch.dtype) -> Tensor:
        nbytes = count * torch.empty((), dtype=dtype).element_size()
        return self.workspace[off:off + nbytes].view(dtype)

    def header(self) -> Tensor:
        return self._arr(self.layout.header, 64, torch.int32)

    def tensors(self) -> dict:
        """Named views of the integer / float intermediates (for parity tests and debugging)."""
        p, l = self.prm, self.layout
        npair = p.V * p.P
        gx, gy = (p.W + 15) // 16, (p.H + 15) // 16
        nt = p.V * gx * gy
        cap = p.max_instances
        return dict(
            header=self.header(),
            tiles_touched=self._tiles_touched(),
            vis_mask=self._arr(l.vis_mask, p.P, torch.uint8),
            slot_base=self._arr(l.slot_base, 2 * npair, torch.int32).view(p.V, p.P, 2)[..., 0],
            hit_mask=self._arr(l.slot_base, 2 * npair, torch.int32).view(p.V, p.P, 2)[..., 1],
            slot_pair=self._arr(l.slot_pair, cap, torch.int32),
            rec_a=self._arr(l.rec_a, npair * 12, torch.float32).view(p.V, p.P, 12)[..., 0:4],
            rec_b=self._arr(l.rec_a, npair * 12, torch.float32).view(p.V, p.P, 12)[..., 4:8],
            rec_c=self._arr(l.rec_a, npair * 12, torch.float32).view(p.V, p.P, 12)[..., 8:12],
            clamped=self._arr(l.clamped, npair, torch.uint8).view(p.V, p.P),
            depths=self._arr(l.depths, npair, torch.float32).view(p.V, p.P),
            tile_count=self._arr(l.tile_count, nt, torch.int32),
            tile_start=self._arr(l.tile_start, nt + 1, torch.int32),
            keys=self._arr(l.keys, cap, torch.int64),
            list=self._arr(l.list, cap, torch.int32),
            final_T=self._arr(l.final_T, p.V * p.H * p.W, torch.float32).view(p.V, p.H, p.W),
            n_contrib=self._arr(l.n_contrib, p.V * p.H * p.W, torch.int32).view(p.V, p.H, p.W),
            tile_max_contrib=self._arr(l.tile_max_contrib, nt, torch.int32),
            seg_flag=self._arr(l.seg_flag, nt * 4, torch.int32),       # 1: this (tile, quadrant) split its list (S360_FLAG_SPLIT_LISTS)
        )

    def _tiles_touched(self) -> Tensor:
        """[V,P] tile counts: the kernels write them for visible pairs only (vis_mask bit v of Gaussian g), the rest reads 0."""
        p, l = self.prm, self.layout
        raw = self._arr(l.tiles_touched, p.V * p.P, torch.int32).view(p.V, p.P)
        vis = self._arr(l.vis_mask, p.P, torch.uint8).to(torch.int32)
        bits = (vis[None, :] >> torch.arange(p.V, device=vis.device, dtype=torch.int32)[:, None]) & 1
        return torch.where(bits.bool(), raw, torch.zeros_like(raw))

    def _read_header(self):
        h = self.header()[:2].cpu()          # one synchronising read for both words
        n, over = int(h[0]) & 0xFFFFFFFF, bool(int(h[1]))
        p = self.prm
        key = _hint_key(self.workspace.device, p.P, p.V, p.H, p.W, bool(p.flags & _lib.FLAG_LEAN_LISTS))
        _CAPACITY_HINT[key] = max(n, _CAPACITY_HINT.get(key, 0))    # running maximum: sizes later calls (default_capacity)
        _poll_mirror(key, warn=False)                               # the call has finished: its mirror word (chunk count) is there too
        return n, over

    def count_contributions(self):
        """(contributing (pixel, entry) pairs, pairs the backward composite evaluates): s360_count_contributions on this training
        workspace (host read; a measurement aid, see include/s360.h)."""
        out = torch.zeros(2, dtype=torch.int64, device=self.workspace.device)
        with torch.cuda.device(self.workspace.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.workspace.device).cuda_stream)
            _lib.check(_lib.lib().s360_count_contributions(C.byref(self.prm), _ptr(self.workspace), self.layout.total_bytes, _ptr(out), stream),
                       "s360_count_contributions")
        a, b = out.cpu().tolist()
        return int(a), int(b)

    def count_backward_slots(self) -> dict:
        """s360_count_backward_slots on this (unsplit, training) workspace: where the backward composite's lane x pixel slots go.
        The library fills 32 words: [0..25] as include/s360.h lists them, [26] / [27] / [28] survivor records whose splat reaches the
        quadrant's upper 8x4 half / its lower half / both (the composites' own box test), [29] records, [30] 8-run iterations of a
        composite that would walk the two halves side by side on 32 + 32 lanes (round 6: counted before building it — see DESIGN.md)."""
        out = torch.zeros(32, dtype=torch.int64, device=self.workspace.device)
        with torch.cuda.device(self.workspace.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.workspace.device).cuda_stream)
            _lib.check(_lib.lib().s360_count_backward_slots(C.byref(self.prm), _ptr(self.workspace), self.layout.total_bytes, _ptr(out), stream),
                       "s360_count_backward_slots")
        c = [int(x) for x in out.cpu().tolist()]
        return dict(executed=c[0], padding=c[1], stopped=c[2], miss=c[3], contributing=c[4], skipped_runs=c[5], units=c[6], groups=c[7],
                    by_unit_hit_decile=c[8:18], runs_by_active_records=dict(zip(("0", "1-4", "5-8", "9-16", "17-24", "25-32", "33-48", "49-64"), c[18:26])),
                    records=c[29], reach_upper_half=c[26], reach_lower_half=c[27], reach_both_halves=c[28], half_wave_iterations=c[30])

    def split_errors(self) -> int:
        """header[7] (host read): 0 unless k_render_tail met a corrupt segment work item (it then skips the item; the images of such a
        call are invalid).  No wave of the split path ever waits for another one, so there is no time-out any more.  Never seen on a
        correct build; the tests and bench.py assert it stays 0."""
        return int(self.header()[7].item())

    def num_rendered(self) -> int:
        """Host read of num_instances (synchronises)."""
        return self._read_header()[0]

    def overflowed(self) -> bool:
        return self._read_header()[1]


def _ptr(t: Optional[Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32c(t: Tensor, name: str) -> Tensor:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU (hip device); the rasteriser has no CPU path")
    return t.detach().float().contiguous()


def _forward_call(prm, views, means3D, cov6, opac, shs, colors, want_radii: bool, depth_mode=None, mse=None):
    lay = _lib.layout(prm)
    dev = means3D.device
    ws = torch.empty(lay.total_bytes, dtype=torch.uint8, device=dev)
    n_img = prm.V // 2 if (prm.flags & _lib.FLAG_SPHERICAL) else prm.V     # spherical: views = (camera, seam ghost) pairs
    images = torch.empty((n_img, 3, prm.H, prm.W), dtype=torch.float32, device=dev)
    radii = torch.empty((prm.V, prm.P), dtype=torch.int32, device=dev) if want_radii else None
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    depth = None
    if mse is not None:     # (target[V,3,H,W], grad_scale): loss epilogue fused into the composite store
        target, grad_scale = mse
        if depth_mode is not None:
            depth = torch.empty((prm.V, prm.H, prm.W), dtype=torch.float32, device=dev)
        d_images = torch.empty_like(images)
        nquads = prm.V * ((prm.H + 15) // 16) * ((prm.W + 15) // 16) * 4   # one partial per wave footprint (8x8 px)
        partials = torch.empty((nquads, 2), dtype=torch.float32, device=dev)
        loss_out = torch.empty((1 + prm.V,), dtype=torch.float32, device=dev)   # loss, clipped MSE per view (same call)
        rc = _lib.lib().s360_forward_mse(C.byref(prm), _ptr(views), _ptr(means3D), _ptr(cov6), _ptr(opac), _ptr(shs),
                                         _ptr(colors), _ptr(images), _ptr(depth), DEPTH_MODES.get(depth_mode, 0), _ptr(radii),
                                         _ptr(target), C.c_float(grad_scale), _ptr(d_images), _ptr(partials), _ptr(loss_out),
                                         _ptr(ws), lay.total_bytes, stream)
        _lib.check(rc, "s360_forward_mse")
        st = RasterState(prm, lay, ws)
        st.d_images, st.mse_partials, st.mse_out = d_images, partials, loss_out
        return images, radii, st, depth
    if depth_mode is None:
        rc = _lib.lib().s360_forward(C.byref(prm), _ptr(views), _ptr(means3D), _ptr(cov6), _ptr(opac), _ptr(shs),
                                     _ptr(colors), _ptr(images), _ptr(radii), _ptr(ws), lay.total_bytes, stream)
    else:
        depth = torch.empty((n_img, prm.H, prm.W), dtype=torch.float32, device=dev)
        rc = _lib.lib().s360_forward_depth(C.byref(prm), _ptr(views), _ptr(means3D), _ptr(cov6), _ptr(opac), _ptr(shs),
                                           _ptr(colors), _ptr(images), _ptr(depth), DEPTH_MODES[depth_mode], _ptr(radii),
                                           _ptr(ws), lay.total_bytes, stream)
    _lib.check(rc, "s360_forward")
    return images, radii, RasterState(prm, lay, ws), depth


class _RasterizeViews(torch.autograd.Function):
    """autograd node of one multi-view rasterisation (forward saves inputs + workspace)."""

    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, cov6, views, cfg, mse_target=None):
        (h, w, sh_degree, shared_campos, max_instances, check, want_radii, cov9, sh_channel_major, keep_slots, depth_mode,
         defer_sh, mse_weight, mse_count, spherical, exchange, lean, mse_defer, atomic_grads, split_lists) = cfg
        if exchange is not None and (shs is None or not (shared_campos or int(views.shape[0]) == 1) or defer_sh):
            raise RuntimeError("exchange=: the chunked gradient exchange needs SH colours and views sharing one camera centre "
                               "(and replaces defer_sh)")
        ctx.exchange = exchange
        if spherical and (mse_target is not None or int(views.shape[0]) % 2):
            raise RuntimeError("spherical mode: views come in (camera, seam ghost) pairs; the fused loss epilogue is cube-face only")
        if not means3D.is_cuda:
            raise RuntimeError("means3D must live on the GPU (hip device); the rasteriser has no CPU path")
        with torch.cuda.device(means3D.device):
            m3 = _f32c(means3D, "means3D")
            c6 = _f32c(cov6, "cov3D_precomp")
            op = _f32c(opacities, "opacities").reshape(-1)
            sh = None if shs is None else _f32c(shs, "shs")
            col = None if colors_precomp is None else _f32c(colors_precomp, "colors_precomp")
            vw = _f32c(views, "views")
            p, v = int(m3.shape[0]), int(vw.shape[0])
            if v > _lib.S360_MAX_VIEWS:
                raise RuntimeError(f"at most {_lib.S360_MAX_VIEWS} views per call")
            prm = _lib.S360Params()
            prm.P, prm.V, prm.H, prm.W = p, v, int(h), int(w)
            prm.sh_degree = int(sh_degree)
            prm.M = 0 if sh is None else int(sh.shape[2] if sh_channel_major else sh.shape[1])
            if sh is not None and p == 0:
                prm.M = max(prm.M, (int(sh_degree) + 1) ** 2)
            needs_bwd = any(ctx.needs_input_grad[:6])  # (grad mode is off inside Function.forward; this reflects apply-time)
            hkey = _hint_key(m3.device, p, v, int(h), int(w), lean)
            _mirror(hkey)
            split_lists = split_decision(hkey, split_lists, v * ((int(h) + 15) // 16) * ((int(w) + 15) // 16) * 4) and not spherical
            prm.flags = (_lib.FLAG_SHARED_CAMPOS if (shared_campos or v == 1) else 0) | (
                _lib.FLAG_COV9 if cov9 else 0) | (_lib.FLAG_SH_CHANNEL_MAJOR if sh_channel_major else 0) | (
                0 if (needs_bwd or keep_slots) else _lib.FLAG_FORWARD_ONLY) | (
                _lib.FLAG_SH_DEG4_IGNORED if SH_DEG4_IGNORED else 0) | (_lib.FLAG_SPHERICAL if spherical else 0) | (
                _lib.FLAG_LEAN_LISTS if lean else 0) | (
                _lib.FLAG_DEFER_LOSS if (mse_defer and mse_target is not None and needs_bwd) else 0) | (
                _lib.FLAG_ATOMIC_GRADS if (atomic_grads and needs_bwd) else 0) | (
                _lib.FLAG_SPLIT_LISTS if (split_lists and not (atomic_grads and needs_bwd)) else 0) | (
                _lib.FLAG_COOP_WALK if coop_decision(hkey) else 0)
            prm.max_instances = int(max_instances) if max_instances else default_capacity(
                p, v, int(h), int(w), device=m3.device, lean=lean, lazy=(check != "sync"))
            # every forward reports (instance count, overflow flag, long-list chunks) into pinned host memory: how check="lazy" callers
            # size the next call
            prm.header_mirror = _mirror(hkey).data_ptr()
            prm.max_segments = default_segments(hkey) if split_lists else 0
            mse = None
            if mse_target is not None:
                tgt = _f32c(mse_target, "mse_target")
                if tuple(tgt.shape) != (v, 3, int(h), int(w)):
                    raise RuntimeError(f"mse_target must be [{v},3,{h},{w}], got {tuple(tgt.shape)}")
                n_mean = int(mse_count) if mse_count else v * 3 * int(h) * int(w)
                mse = (tgt, 2.0 * float(mse_weight) / n_mean)
            images, radii, state, depth = _forward_call(prm, vw, m3, c6, op, sh, col, want_radii, depth_mode, mse)
            if check == "sync" and state.overflowed():
                prm.max_instances = state.num_rendered()
                images, radii, state, depth = _forward_call(prm, vw, m3, c6, op, sh, col, want_radii, depth_mode, mse)
            if mse is not None:
                loss = state.mse_out[0]             # reduced by the call itself (k_mse_finish: fixed order, deterministic)
                clipped_mse = state.mse_out[1:]
            else:
                loss = torch.empty(0, dtype=torch.float32, device=images.device)
                clipped_mse = loss
        ctx.set_materialize_grads(False)
        ctx.state = state
        ctx.holder = {}                       # filled by a deferred-SH backward; reachable from the output (deferred_of)
        _RasterizeViews.last_holder = ctx.holder
        ctx.defer_sh = bool(defer_sh) and sh is not None
        ctx.has_means2D = means2D is not None
        ctx.depth_mode = depth_mode if depth is not None else None
        _RasterizeViews.last_state = state
        if radii is None:
            radii = torch.empty(0, dtype=torch.int32, device=images.device)
        if depth is None:
            depth = torch.empty(0, dtype=torch.float32, device=images.device)
            ctx.mark_non_differentiable(radii, depth, clipped_mse)
        else:   # the fused depth map is differentiable (LossDepth back-propagates through it, loss_depth.py:37-60)
            ctx.mark_non_differentiable(radii, clipped_mse)
        ctx.save_for_backward(m3, c6, op, sh, col, vw)
        return images, radii, depth, loss, clipped_mse

    @staticmethod
    def backward(ctx, grad_images, _grad_radii, grad_depth, grad_loss, _grad_clipped):
        m3, c6, op, sh, col, vw = ctx.saved_tensors
        state: RasterState = ctx.state
        prm, lay = state.prm, state.layout
        dev = m3.device
        with torch.cuda.device(dev):
            g = None if grad_images is None else grad_images.detach().float()
            g_scale = None
            if grad_loss is not None and getattr(state, "d_images", None) is not None:
                # d_images = 2w/N (image - target) from the epilogue; the scalar autograd hands back multiplies it inside
                # the composite's pixel load (dL_dimages_scale) unless an image gradient has to be added as well
                if g is None:
                    g, g_scale = state.d_images, grad_loss.detach().float().reshape(1).contiguous()
                else:
                    g = g + state.d_images * grad_loss.detach().float()
            if g is None:
                n_img = prm.V // 2 if (prm.flags & _lib.FLAG_SPHERICAL) else prm.V
                g = torch.zeros((n_img, 3, prm.H, prm.W), dtype=torch.float32, device=dev)
            g = g.contiguous()
            gd, dm = None, 0
            if grad_depth is not None and ctx.depth_mode is not None:
                gd, dm = grad_depth.detach().float().contiguous(), DEPTH_MODES[ctx.depth_mode]
            p, v = prm.P, prm.V
            d_m3 = torch.empty((p, 3), dtype=torch.float32, device=dev)
            d_c6 = torch.empty_like(c6)
            d_op = torch.empty((p,), dtype=torch.float32, device=dev)
            need = ctx.needs_input_grad
            d_m2 = torch.empty((v, p, 3), dtype=torch.float32, device=dev) if (ctx.has_means2D and need[1]) else None
            d_sh = torch.empty_like(sh) if (sh is not None and need[2]) else None
            d_col = torch.empty((p, 3), dtype=torch.float32, device=dev) if (col is not None and need[3]) else None
            bws = torch.empty(lay.backward_bytes, dtype=torch.uint8, device=dev)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if ctx.exchange is not None:
                # multi-GPU: the per-Gaussian gradients leave this node already SUMMED over the ranks — composite once, then
                # the per-Gaussian tail range by range with every range's collectives in flight behind the next one
                from . import distributed
                ex = ctx.exchange
                lib = _lib.lib()
                rc = lib.s360_backward_composite(C.byref(prm), _ptr(vw), _ptr(state.workspace), lay.total_bytes, _ptr(g), _ptr(g_scale),
                                                 _ptr(gd), dm, _ptr(bws), lay.backward_bytes, stream)
                _lib.check(rc, "s360_backward_composite")
                packed = torch.empty((p, 10), dtype=torch.float32, device=dev)
                rgb = torch.empty((p, 4), dtype=torch.float32, device=dev)
                rank = ex.rank()

                def produce(lo, hi):
                    _lib.check(lib.s360_backward_gaussians(C.byref(prm), _ptr(vw), _ptr(m3), _ptr(c6), _ptr(sh), _ptr(state.workspace),
                                                           lay.total_bytes, int(gd is not None), dm, lo, hi - lo, rank, _ptr(packed),
                                                           _ptr(d_m2), _ptr(rgb), _ptr(bws), lay.backward_bytes, stream),
                               "s360_backward_gaussians")

                slab = sh.shape[1] * sh.shape[2]

                def rebuild_sh(lo, hi, rgb_all, rep_all):
                    sub = _lib.S360Params()
                    C.memmove(C.byref(sub), C.byref(prm), C.sizeof(sub))
                    sub.P, sub.V = hi - lo, int(rep_all.shape[0])
                    st2 = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                    _lib.check(lib.s360_sh_backward(C.byref(sub), int(rgb_all.shape[0]), _ptr(rep_all.contiguous()),
                                                    C.c_void_p(m3.data_ptr() + 12 * lo), _ptr(rgb_all.contiguous()),
                                                    C.c_void_p(d_sh.data_ptr() + 4 * slab * lo), st2), "s360_sh_backward")

                def reduce_rows(lo, hi, rows):      # "gather" form: sum the ranks' rows + unpack, one pass
                    _lib.check(lib.s360_reduce_unpack_gradients(_ptr(rows), int(rows.shape[0]), lo, hi - lo, int(c6.dim() == 3), _ptr(d_m3), _ptr(d_c6),
                                                                _ptr(d_op), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                               "s360_reduce_unpack_gradients")

                # harmonics frozen (need[2] False — on every rank, it is the same model): no dL/dRGB gathers, no dL/dSH rebuild
                form = distributed.exchange_chunked(p, packed, rgb, vw[0], produce, rebuild_sh if d_sh is not None else None,
                                                    n_chunks=ex.n_chunks, group=ex.group, group_gather=ex.group_gather,
                                                    force_collectives=getattr(ex, "force_collectives", False), mode=getattr(ex, "mode", None),
                                                    reduce_rows=reduce_rows)
                if form != "gather":
                    _lib.check(lib.s360_unpack_gradients(_ptr(packed), p, int(c6.dim() == 3), _ptr(d_m3), _ptr(d_c6), _ptr(d_op), stream),
                               "s360_unpack_gradients")
                if d_m2 is not None:
                    d_m2 = d_m2.sum(0) if v > 1 else d_m2[0]
                return d_m3, d_m2, d_sh, None, d_op.view(-1, 1), d_c6, None, None, None
            if ctx.defer_sh:
                # multi-GPU factored form: no SH pass here; the caller exchanges d_rgb_sum and finishes with
                # finish_deferred_sh() (see distributed.sync_gradients_factored)
                d_rgb = torch.empty((p, 4), dtype=torch.float32, device=dev)
                rc = _lib.lib().s360_backward_split(
                    C.byref(prm), _ptr(vw), _ptr(m3), _ptr(c6), _ptr(op), _ptr(sh), _ptr(state.workspace), lay.total_bytes,
                    _ptr(g), _ptr(g_scale), _ptr(gd), dm, _ptr(d_m3), _ptr(d_m2), _ptr(d_c6), _ptr(d_op), _ptr(d_rgb),
                    _ptr(bws), lay.backward_bytes, stream)
                _lib.check(rc, "s360_backward_split")
                ctx.holder["deferred"] = _RasterizeViews.last_deferred = DeferredSH(prm, vw, m3, sh, d_rgb)
                if d_m2 is not None:
                    d_m2 = d_m2.sum(0) if v > 1 else d_m2[0]
                return d_m3, d_m2, None, None, d_op.view(-1, 1), d_c6, None, None, None
            rc = _lib.lib().s360_backward(
                C.byref(prm), _ptr(vw), _ptr(m3), _ptr(c6), _ptr(op), _ptr(sh), _ptr(col), _ptr(state.workspace),
                lay.total_bytes, _ptr(g), _ptr(g_scale), _ptr(gd), dm, _ptr(d_m3), _ptr(d_m2), _ptr(d_c6),
                _ptr(d_op), _ptr(d_sh), _ptr(d_col), _ptr(bws), lay.backward_bytes, stream)
            _lib.check(rc, "s360_backward")
        if d_m2 is not None:
            d_m2 = d_m2.sum(0) if v > 1 else d_m2[0]
        return d_m3, d_m2, d_sh, d_col, d_op.view(-1, 1), d_c6, None, None, None


class _RasterizeRaw(torch.autograd.Function):
    """The adapter tail fused into the rasteriser (s360_forward_raw / s360_backward_raw): from the encoder's raw
    outputs (depths, opacities, raw_gaussians of the context panoramas) to the rendered views of ONE target camera centre, without
    the [G,3,25] harmonics / [G,3,3] covariances the two-step path (adapter.adapter_tail, then rasterize_views) writes and reads
    back, and without the [G,3,25] dL/dSH round trip in the backward.  Differentiable w.r.t. depths, opacities, raw_gaussians."""

    @staticmethod
    def forward(ctx, depths, opacities, raw, ctx_extrinsics, sh_rot, views, cfg, mse_target=None):
        (h, w, ch, cw, per_ray, smin, smax, eps, conv, diff_means, max_instances, check, depth_mode, mse_weight, mse_count, lean, mse_defer,
         split_lists, exchange) = cfg
        ctx.exchange = exchange
        if not depths.is_cuda:
            raise RuntimeError("depths must live on the GPU (hip device); the rasteriser has no CPU path")
        dev = depths.device
        with torch.cuda.device(dev):
            dep = _f32c(depths, "depths").reshape(-1)
            op = _f32c(opacities, "opacities").reshape(-1)
            rw = _f32c(raw, "raw_gaussians")
            ext = _f32c(ctx_extrinsics, "extrinsics").reshape(-1, 4, 4)
            rot = None if sh_rot is None else _f32c(sh_rot, "sh_rotation")
            vw = _f32c(views, "views")
            nv = int(ext.shape[0])
            p, v = int(dep.shape[0]), int(vw.shape[0])
            if rw.shape[-1] != 82 or rw.numel() != p * 82 or p % max(nv, 1) or (p // max(nv, 1)) != ch * cw * per_ray:
                raise RuntimeError("rasterize_raw: raw_gaussians must be [views * h * w * per_ray, 82] (degree-4 harmonics) matching depths / extrinsics")
            if v > _lib.S360_MAX_VIEWS:
                raise RuntimeError(f"at most {_lib.S360_MAX_VIEWS} views per call")
            needs_bwd = any(ctx.needs_input_grad[:3])
            hkey = _hint_key(dev, p, v, int(h), int(w), lean)
            _mirror(hkey)
            split_lists = split_decision(hkey, split_lists, v * ((int(h) + 15) // 16) * ((int(w) + 15) // 16) * 4)
            prm = _lib.S360Params()
            prm.P, prm.V, prm.H, prm.W, prm.sh_degree, prm.M = p, v, int(h), int(w), 4, 25
            prm.flags = _lib.FLAG_SHARED_CAMPOS | _lib.FLAG_RAW_INPUTS | (0 if needs_bwd else _lib.FLAG_FORWARD_ONLY) | (
                _lib.FLAG_LEAN_LISTS if lean else 0) | (_lib.FLAG_SPLIT_LISTS if split_lists else 0) | (
                _lib.FLAG_DEFER_LOSS if (mse_defer and mse_target is not None and needs_bwd) else 0) | (
                _lib.FLAG_COOP_WALK if coop_decision(hkey) else 0)
            prm.max_instances = int(max_instances) if max_instances else default_capacity(p, v, int(h), int(w), device=dev, lean=lean,
                                                                                          lazy=(check != "sync"))
            prm.header_mirror = _mirror(hkey).data_ptr()
            prm.max_segments = default_segments(hkey) if split_lists else 0
            rin = _lib.S360RawInputs(ext.data_ptr(), dep.data_ptr(), rw.data_ptr(), None if rot is None else rot.data_ptr(), nv, p // max(nv, 1),
                                     int(ch), int(cw), int(per_ray), int(conv), float(smin), float(smax), float(eps))
            means = torch.empty((p, 3), dtype=torch.float32, device=dev)
            cov6 = torch.empty((p, 6), dtype=torch.float32, device=dev)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

            def call():
                lay = _lib.layout(prm)
                ws = torch.empty(lay.total_bytes, dtype=torch.uint8, device=dev)
                images = torch.empty((v, 3, prm.H, prm.W), dtype=torch.float32, device=dev)
                depth = torch.empty((v, prm.H, prm.W), dtype=torch.float32, device=dev) if depth_mode is not None else None
                st = RasterState(prm, lay, ws)
                tgt = d_images = partials = loss_out = None
                gs = 0.0
                if mse_target is not None:
                    tgt = _f32c(mse_target, "mse_target")
                    n_mean = int(mse_count) if mse_count else v * 3 * int(h) * int(w)
                    gs = 2.0 * float(mse_weight) / n_mean
                    d_images = torch.empty_like(images)
                    partials = torch.empty((v * ((prm.H + 15) // 16) * ((prm.W + 15) // 16) * 4, 2), dtype=torch.float32, device=dev)
                    loss_out = torch.empty((1 + v,), dtype=torch.float32, device=dev)
                    st.d_images, st.mse_partials, st.mse_out = d_images, partials, loss_out
                rc = _lib.lib().s360_forward_raw(C.byref(prm), _ptr(vw), C.byref(rin), _ptr(op), _ptr(means), _ptr(cov6), _ptr(images), _ptr(depth),
                                                 DEPTH_MODES.get(depth_mode, 0), None, _ptr(tgt), C.c_float(gs), _ptr(d_images), _ptr(partials),
                                                 _ptr(loss_out), _ptr(ws), lay.total_bytes, stream)
                _lib.check(rc, "s360_forward_raw")
                return images, depth, st

            images, depth, state = call()
            if check == "sync" and state.overflowed():
                prm.max_instances = state.num_rendered()
                images, depth, state = call()
        ctx.set_materialize_grads(False)
        ctx.state, ctx.rin_cfg, ctx.depth_mode, ctx.diff_means = state, (nv, ch, cw, per_ray, conv, smin, smax, eps), depth_mode, bool(diff_means)
        ctx.in_shapes = (tuple(depths.shape), tuple(opacities.shape), tuple(raw.shape))   # gradients go back in the callers' shapes
        _RasterizeViews.last_state = state
        loss = state.mse_out[0] if mse_target is not None else torch.empty(0, dtype=torch.float32, device=dev)
        clipped = state.mse_out[1:] if mse_target is not None else loss
        if depth is None:
            depth = torch.empty(0, dtype=torch.float32, device=dev)
            ctx.mark_non_differentiable(depth, clipped, means, cov6)
        else:
            ctx.mark_non_differentiable(clipped, means, cov6)
        ctx.save_for_backward(dep, op, rw, ext, rot, vw, means, cov6)
        return images, depth, loss, clipped, means, cov6

    @staticmethod
    def backward(ctx, grad_images, grad_depth, grad_loss, _gc, _gm, _gv):
        dep, op, rw, ext, rot, vw, means, cov6 = ctx.saved_tensors
        state: RasterState = ctx.state
        prm, lay = state.prm, state.layout
        nv, ch, cw, per_ray, conv, smin, smax, eps = ctx.rin_cfg
        dev = dep.device
        p = prm.P
        with torch.cuda.device(dev):
            g = None if grad_images is None else grad_images.detach().float()
            g_scale = None
            if grad_loss is not None and getattr(state, "d_images", None) is not None:
                if g is None:
                    g, g_scale = state.d_images, grad_loss.detach().float().reshape(1).contiguous()
                else:
                    g = g + state.d_images * grad_loss.detach().float()
            if g is None:
                g = torch.zeros((prm.V, 3, prm.H, prm.W), dtype=torch.float32, device=dev)
            g = g.contiguous()
            gd, dm = None, 0
            if grad_depth is not None and ctx.depth_mode is not None:
                gd, dm = grad_depth.detach().float().contiguous(), DEPTH_MODES[ctx.depth_mode]
            rin = _lib.S360RawInputs(ext.data_ptr(), dep.data_ptr(), rw.data_ptr(), None if rot is None else rot.data_ptr(), nv, p // max(nv, 1),
                                     int(ch), int(cw), int(per_ray), int(conv), float(smin), float(smax), float(eps))
            d_m3 = torch.empty((p, 3), dtype=torch.float32, device=dev)
            d_c6 = torch.empty((p, 6), dtype=torch.float32, device=dev)
            d_op = torch.empty((p,), dtype=torch.float32, device=dev)
            d_rgb = torch.empty((p, 4), dtype=torch.float32, device=dev)
            d_dep = torch.empty((p,), dtype=torch.float32, device=dev)
            d_raw = torch.empty((p, 82), dtype=torch.float32, device=dev)
            bws = torch.empty(lay.backward_bytes, dtype=torch.uint8, device=dev)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            if ctx.exchange is not None:
                # multi-GPU (one target panorama per rank, the same raw cloud on every rank): the rows the exchange moves — packed
                # [P,10] (dL/dmean | dL/dcov6 | dL/dopacity) and the clamp-masked dL/dRGB sums [P,4] — are exactly what k_raw_bwd takes,
                # so after the exchange ONE launch forms dL/d(raw record) from the summed rows and the N ranks' dL/dRGB factors
                from . import distributed
                ex = ctx.exchange
                lib = _lib.lib()
                _lib.check(lib.s360_backward_composite(C.byref(prm), _ptr(vw), _ptr(state.workspace), lay.total_bytes, _ptr(g), _ptr(g_scale),
                                                       _ptr(gd), dm, _ptr(bws), lay.backward_bytes, stream), "s360_backward_composite")
                packed = torch.empty((p, 10), dtype=torch.float32, device=dev)
                rgb = torch.empty((p, 4), dtype=torch.float32, device=dev)
                rank = ex.rank()
                gathered = {}

                def produce(lo, hi):
                    _lib.check(lib.s360_backward_gaussians(C.byref(prm), _ptr(vw), _ptr(means), _ptr(cov6), _ptr(rw), _ptr(state.workspace),
                                                           lay.total_bytes, int(gd is not None), dm, lo, hi - lo, rank, _ptr(packed),
                                                           None, _ptr(rgb), _ptr(bws), lay.backward_bytes, stream), "s360_backward_gaussians")

                def collect(lo, hi, rgb_all, rep_all):      # keep every range's gathered factors: k_raw_bwd runs once, over all of them
                    gathered[(lo, hi)] = rgb_all
                    gathered["rep"] = rep_all

                def reduce_rows(lo, hi, rows):      # "gather" form: sum the ranks' rows + unpack, one pass
                    _lib.check(lib.s360_reduce_unpack_gradients(_ptr(rows), int(rows.shape[0]), lo, hi - lo, 0, _ptr(d_m3), _ptr(d_c6), _ptr(d_op),
                                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "s360_reduce_unpack_gradients")

                form = distributed.exchange_chunked(p, packed, rgb, vw[0], produce, collect, n_chunks=ex.n_chunks, group=ex.group,
                                                    group_gather=ex.group_gather, force_collectives=getattr(ex, "force_collectives", False),
                                                    mode=getattr(ex, "mode", None), reduce_rows=reduce_rows)
                rep_all = gathered.pop("rep").contiguous()
                world = int(rep_all.shape[0])
                if len(gathered) == 1:
                    rgb_all = next(iter(gathered.values())).contiguous()
                else:
                    rgb_all = torch.empty((world, p, 4), dtype=torch.float32, device=dev)
                    for (lo, hi), t in gathered.items():
                        rgb_all[:, lo:hi] = t
                if form != "gather":
                    _lib.check(lib.s360_unpack_gradients(_ptr(packed), p, 0, _ptr(d_m3), _ptr(d_c6), _ptr(d_op), stream), "s360_unpack_gradients")
                _lib.check(lib.s360_backward_raw_tail(C.byref(prm), _ptr(rep_all), world, C.byref(rin), _ptr(means), _ptr(state.workspace),
                                                      lay.total_bytes, _ptr(d_m3) if ctx.diff_means else None, _ptr(d_c6), _ptr(rgb_all),
                                                      _ptr(d_dep), _ptr(d_raw), stream), "s360_backward_raw_tail")
                sd, so, sr = ctx.in_shapes
                return d_dep.view(sd), d_op.view(so), d_raw.view(sr), None, None, None, None, None
            rc = _lib.lib().s360_backward_raw(C.byref(prm), _ptr(vw), C.byref(rin), _ptr(means), _ptr(cov6), _ptr(op), _ptr(state.workspace),
                                              lay.total_bytes, _ptr(g), _ptr(g_scale), _ptr(gd), dm, int(ctx.diff_means), _ptr(d_m3), _ptr(d_c6),
                                              _ptr(d_op), _ptr(d_rgb), _ptr(d_dep), _ptr(d_raw), _ptr(bws), lay.backward_bytes, stream)
            _lib.check(rc, "s360_backward_raw")
        sd, so, sr = ctx.in_shapes
        return d_dep.view(sd), d_op.view(so), d_raw.view(sr), None, None, None, None, None


_RasterizeViews.last_state = None
_RasterizeViews.last_deferred = None
_RasterizeViews.last_holder = None


class FusedMse(NamedTuple):
    """Result of the loss epilogue: LossMse (src/loss/loss_mse.py:30-31) and the per-view clipped MSE that
    compute_psnr (src/evaluation/metrics.py:11-21) takes the log of."""
    loss: Tensor
    clipped_mse: Tensor

    def psnr(self) -> Tensor:
        m = torch.where(self.clipped_mse == 0.0, torch.full_like(self.clipped_mse, 1e-10), self.clipped_mse)
        return -10 * m.log10()


class DeferredSH:
    """What a deferred-SH backward leaves behind: the S360Params, the packed views, the inputs the SH pass
    needs and d_rgb_sum[P,4] (xyz = clamp-masked sum of dL/dRGB over this call's views, w = int32 bits of the
    first view that saw the Gaussian, -1 if none)."""

    def __init__(self, prm, views, means3D, shs, d_rgb_sum):
        self.prm, self.views, self.means3D, self.shs, self.d_rgb_sum = prm, views, means3D, shs, d_rgb_sum


def last_deferred() -> Optional[DeferredSH]:
    """DeferredSH of the most recent deferred-SH backward in this process (prefer deferred_of(images): it is tied to one
    rasteriser call and therefore safe with several defer_sh calls per step)."""
    return _RasterizeViews.last_deferred


def deferred_of(images: Tensor) -> Optional[DeferredSH]:
    """The DeferredSH that the backward of THIS rasteriser call left behind (images = the tensor returned by
    rasterize_views / render_views_fused with defer_sh=True); None before its backward has run."""
    holder = getattr(images, "s360_deferred", None)
    return None if holder is None else holder.get("deferred")


def finish_deferred_sh(prm, views: Tensor, means3D: Tensor, shs: Tensor, d_rgb_sums: Tensor) -> Tensor:
    """s360_sh_backward: views[n,44] (one representative camera per group), d_rgb_sums[n,P,4] with .w = the
    group's index into `views` (int32 bits) or -1.  Returns the summed dL/dSH (shape / layout of `shs`, which is only a
    template here: the kernel reads neither the coefficients nor any gradient buffer)."""
    d_sh = torch.empty_like(shs)
    n = int(d_rgb_sums.shape[0])
    with torch.cuda.device(shs.device):
        stream = C.c_void_p(torch.cuda.current_stream(shs.device).cuda_stream)
        rc = _lib.lib().s360_sh_backward(C.byref(prm), n, _ptr(views.contiguous()), _ptr(means3D),
                                         _ptr(d_rgb_sums.contiguous()), _ptr(d_sh), stream)
    _lib.check(rc, "s360_sh_backward")
    return d_sh


def rasterize_views(means3D: Tensor, cov6: Tensor, opacities: Tensor, shs: Optional[Tensor] = None,
                    colors_precomp: Optional[Tensor] = None, *, views: Tensor, image_height: int, image_width: int,
                    sh_degree: int = 0, shared_campos: bool = False, max_instances: Optional[int] = None,
                    check: str = "sync", want_radii: bool = True, means2D: Optional[Tensor] = None,
                    cov9: bool = False, sh_channel_major: bool = False, keep_slots: bool = False,
                    depth_mode: Optional[str] = None, defer_sh: bool = False, mse_target: Optional[Tensor] = None,
                    mse_weight: float = 1.0, mse_count: Optional[int] = None, spherical: bool = False, exchange=None,
                    lean: Optional[bool] = None, mse_defer: bool = False, atomic_grads: Optional[bool] = None,
                    split_lists: Optional[bool] = None):
    """Render V views ([V,44] packed, see pack_views) of one cloud.  cov9: cov6 is [P,3,3];
    sh_channel_major: shs is [P,3,M] (the reference's Gaussians layouts, consumed without copies).
    When no input requires grad the instance-slot tables (backward-only state) are skipped unless
    keep_slots=True.  depth_mode ("depth" | "disparity" | "relative_disparity" | "log"): also return the
    fused depth map [V,H,W] of render_depth_cuda as a third result (differentiable; needs near / far in `views`).
    defer_sh=True (views sharing one camera centre): the backward skips the SH pass, returns no gradient for
    `shs` and leaves a DeferredSH (last_deferred()) for distributed.sync_gradients_factored.  Returns (images[V,3,H,W],
    radii[V,P] int32).  opacities may be [P] or [P,1]; its gradient has the same shape.
    mse_target[V,3,H,W]: fuse the cube-face L2 loss into the composite store — an extra last result
    FusedMse(loss = mse_weight * mean((images - target)^2) (differentiable scalar; mean over mse_count elements,
    default all of this call's), clipped_mse[V] for psnr()).
    check="sync": read the overflow flag after the forward (one host sync, like upstream's own
    scan read-back) and re-run with the exact size if the binning capacity was exceeded;
    check="lazy": never synchronise — validate later via last_state().overflowed().
    exchange=distributed.ExchangeConfig(...) (views sharing one camera centre, SH colours): the gradients this node returns
    are SUMMED over the ranks of the process group — each rank renders its own views of the same replicated cloud — and the
    exchange runs range by range inside the backward (distributed.exchange_chunked).
    mse_defer=True (with mse_target, on a call that will be back-propagated): the loss reduction runs inside the backward's first
    launch instead of as a launch of its own at the end of the forward (S360_FLAG_DEFER_LOSS) — FusedMse.loss / clipped_mse hold
    their values only AFTER .backward(); for training loops that read the scalar for logging after the step.
    atomic_grads (default: module switch ATOMIC_GRADS = False): S360_FLAG_ATOMIC_GRADS — float32 atomics in the backward
    composite instead of the deterministic gather (less scratch, one launch less; gradients not bit-reproducible).
    split_lists (default: module switch SPLIT_LONG_LISTS = "auto": adaptive, history-dependent — see the switch; True / False pin it): S360_FLAG_SPLIT_LISTS — long tile lists whose pixels do not
    saturate are composited segment-parallel, forward and backward (see the switch's comment).
    lean (default: module switch LEAN_LISTS = True): bin a (Gaussian, tile) instance only where the splat can reach
    alpha >= 1/255 on that tile — same images / radii / gradients bit for bit, shorter lists; lean=False = upstream's rectangles.
    spherical=True: native equirectangular splat mode (S360_FLAG_SPHERICAL; no reference counterpart, specified by the
    oracle's geo_sph): `views` = pack_views_spherical(...) — (camera, seam ghost) pairs — and the result is one
    [H,W] equirectangular image per pair; means2D gradients are in pixel units."""
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    op2 = opacities.reshape(-1, 1)
    if depth_mode is not None and depth_mode not in DEPTH_MODES:
        raise ValueError(f"depth_mode must be one of {sorted(DEPTH_MODES)}")
    cfg = (image_height, image_width, sh_degree, shared_campos, max_instances, check, want_radii, cov9,
           sh_channel_major, keep_slots, depth_mode, defer_sh, mse_weight, mse_count, bool(spherical), exchange,
           LEAN_LISTS if lean is None else bool(lean), bool(mse_defer), ATOMIC_GRADS if atomic_grads is None else bool(atomic_grads),
           SPLIT_LONG_LISTS if split_lists is None else (split_lists if split_lists == "auto" else bool(split_lists)))
    images, radii, depth, loss, clipped = _RasterizeViews.apply(means3D, means2D, shs, colors_precomp, op2, cov6, views,
                                                                cfg, mse_target)
    images.s360_deferred = _RasterizeViews.last_holder      # see deferred_of()
    out = (images, radii) if depth_mode is None else (images, radii, depth)
    return out if mse_target is None else out + (FusedMse(loss, clipped),)


def rasterize_raw(depths: Tensor, opacities: Tensor, raw_gaussians: Tensor, context_extrinsics: Tensor, *, views: Tensor, image_height: int,
                  image_width: int, context_shape: tuple, scale_min: float, scale_max: float, sh_rotation: Optional[Tensor] = None,
                  per_ray: int = 1, eps: float = 1e-8, erp_convention: int = 0, differentiable_means: bool = False,
                  max_instances: Optional[int] = None, check: str = "sync", depth_mode: Optional[str] = None,
                  mse_target: Optional[Tensor] = None, mse_weight: float = 1.0, mse_count: Optional[int] = None, lean: Optional[bool] = None,
                  mse_defer: bool = False, split_lists: Optional[bool] = None, exchange=None):
    """Render V views sharing one camera centre ([V,44] packed) straight from the encoder's raw outputs: depths / opacities [n*h*w*per_ray]
    (view-major, ray-major), raw_gaussians [same, 82] (3 scale logits, quaternion xyzw, 3 x 25 SH coefficients), context_extrinsics
    [n,4,4], sh_rotation [n,25,25] (adapter.sh_rotation_blocks) or None.  = adapter.adapter_tail(...) followed by rasterize_views(...)
    on its result, in fewer bytes: see _RasterizeRaw.  Returns (images[V,3,H,W], means[P,3], cov6[P,6]) (+ depth maps, + FusedMse as
    in rasterize_views).  Gradients flow to depths, opacities and raw_gaussians; the means are detached like the reference's unless
    differentiable_means.  exchange=distributed.ExchangeConfig(...): multi-GPU — every rank renders its own target panorama of the same
    raw cloud and the gradients w.r.t. depths / opacities / raw_gaussians come back SUMMED over the ranks (the exchange of
    rasterize_views, with k_raw_bwd fed the N ranks' dL/dRGB factors)."""
    if depth_mode is not None and depth_mode not in DEPTH_MODES:
        raise ValueError(f"depth_mode must be one of {sorted(DEPTH_MODES)}")
    ch, cw = context_shape
    cfg = (image_height, image_width, int(ch), int(cw), int(per_ray), float(scale_min), float(scale_max), float(eps), int(erp_convention),
           bool(differentiable_means), max_instances, check, depth_mode, mse_weight, mse_count, LEAN_LISTS if lean is None else bool(lean),
           bool(mse_defer), SPLIT_LONG_LISTS if split_lists is None else (split_lists if split_lists == "auto" else bool(split_lists)), exchange)
    images, depth, loss, clipped, means, cov6 = _RasterizeRaw.apply(depths, opacities, raw_gaussians, context_extrinsics, sh_rotation, views, cfg,
                                                                    mse_target)
    out = (images, means, cov6) if depth_mode is None else (images, means, cov6, depth)
    return out if mse_target is None else out + (FusedMse(loss, clipped),)


def last_state() -> Optional[RasterState]:
    """Workspace of the most recent forward (tests / lazy overflow validation)."""
    return _RasterizeViews.last_state


def _cov6_from_scale_rotation(scales: Tensor, rotations: Tensor, scale_modifier: float) -> Tensor:
    """Upstream's computeCov3D for the (scales, rotations) input form: Sigma = R S^2 R^T with
    S = scale_modifier*scales, quaternion (r,x,y,z) not re-normalised.  Plain torch (autograd)."""
    r, x, y, z = rotations.unbind(-1)
    rot = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)
    m = rot * (scale_modifier * scales)[:, None, :]
    cov = m @ m.transpose(1, 2)
    i, j = torch.triu_indices(3, 3)
    return cov[:, i, j]


class GaussianRasterizer(nn.Module):
    """Drop-in for upstream's module (constructed at cuda_splatting.py:113, called with keywords
    at :117-124).  Returns (image[3,H,W], radii[P])."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, means2D=None, opacities=None, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        s = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if cov3D_precomp is None:
            cov3D_precomp = _cov6_from_scale_rotation(scales, rotations, s.scale_modifier)
        views = pack_views(s.viewmatrix, s.projmatrix, s.campos, s.tanfovx, s.tanfovy, s.bg)
        images, radii = rasterize_views(
            means3D, cov3D_precomp, opacities, shs, colors_precomp, views=views, image_height=s.image_height,
            image_width=s.image_width, sh_degree=s.sh_degree, shared_campos=True, means2D=means2D, check=DROPIN_CHECK)
        return images[0], radii[0]
