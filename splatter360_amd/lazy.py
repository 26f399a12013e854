"""Lazy Gaussians — the seam that lets the UNCHANGED reference reach the fused raw path (s360_forward_raw / s360_backward_raw).

In the reference the encoder ends with `self.gaussian_adapter_erp.forward(...)` (/root/reference/src/model/encoder/encoder_costvolume.py:414-427,
the module of src/model/encoder/common/gaussian_adapter_erp.py:33-119, constructed at encoder_costvolume.py:185), flattens the result
with four `rearrange` calls into the `Gaussians` container (:490-507, src/model/types.py:7-12) and the model wrapper hands that to the
decoder (src/model/model_wrapper_erp.py:217-229).  A drop-in adapter therefore has to materialise 352 B per Gaussian (means,
[.,3,3] covariances, [.,3,25] harmonics) only for the rasteriser to read them back — the round trip the raw entry points remove.

`install(adapter=True)` (plugin.py) replaces the adapter CLASS with `make_adapter_class(...)`'s: same constructor (`cfg` with
gaussian_scale_min / gaussian_scale_max / sh_degree), same forward arguments, same result container — but `means`, `covariances`,
`harmonics`, `scales`, `rotations` are `LazyField`s: torch.Tensor wrapper subclasses that carry shape / dtype / device and a
reference to the adapter's INPUTS (a `RawBundle`), no data.  A pure reshape of a LazyField (what the encoder's `rearrange`s are)
stays lazy; ANY other operation first materialises the field with the stand-alone adapter kernels (adapter.adapter_tail: one launch,
cached in the bundle, autograd-connected to the encoder's outputs) and then runs on the real tensor — every other consumer of the
Gaussians (ply export, visualisation, a different decoder) sees ordinary tensors with the reference's values.  The fused decoder
(decoder.DecoderSplattingFused, registered by install()) recognises untouched LazyFields (`bundle_of`) and calls
rasterizer.rasterize_raw on the bundle's raw tensors instead: neither harmonics nor covariances ever exist.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor

from . import adapter as _adapter

_FIELDS = {"means": (3,), "covariances": (3, 3), "scales": (3,), "rotations": (4,), "harmonics": None}


class RawBundle:
    """The adapter's inputs, kept instead of its outputs.  depths / opacities [b, v, r, srf, spp]; raw [b, v, r, srf, spp or 1, 82];
    extrinsics [b, v, 4, 4] (context panoramas, camera-to-world)."""

    def __init__(self, module, dataset_name, extrinsics, depths, opacities, raw, image_shape, eps):
        self.module, self.dataset_name, self.eps = module, dataset_name, float(eps)
        self.extrinsics, self.depths, self.opacities, self.raw = extrinsics, depths, opacities, raw
        self.image_shape = (int(image_shape[0]), int(image_shape[1]))
        self.shape5 = tuple(int(x) for x in depths.shape)
        self._materialised = None
        self._rot = None

    @property
    def device(self):
        return self.depths.device

    @property
    def per_ray(self) -> int:
        return self.shape5[3] * self.shape5[4]

    def sh_rotation(self) -> Optional[Tensor]:
        """[b * v, 25, 25] matrices of rotate_sh for the context views (None: identity), built once per bundle."""
        if self._rot is None:
            b, v = self.shape5[:2]
            self._rot = (self.module.rotation_blocks(self.extrinsics.reshape(b * v, 4, 4)),)
        return self._rot[0]

    def materialised(self):
        """The adapter's real outputs (stand-alone kernels), computed at most once."""
        if self._materialised is None:
            b, v = self.shape5[:2]
            self._materialised = self.module.forward_eager(self.dataset_name, self.extrinsics.reshape(b, v, 1, 1, 1, 4, 4), self.depths,
                                                           self.opacities, self.raw, self.image_shape, self.eps, sh_rot=self.sh_rotation())
        return self._materialised

    def field(self, name: str, shape) -> Tensor:
        return getattr(self.materialised(), name).reshape(shape)


_PURE_RESHAPES = {"reshape", "view", "_unsafe_view", "reshape_as", "view_as", "flatten", "unflatten", "unsqueeze", "squeeze", "contiguous",
                  "detach_", "requires_grad_"}
_METADATA = {"size", "dim", "numel", "nelement", "ndimension", "is_floating_point", "is_complex", "element_size", "is_contiguous", "get_device",
             "__len__", "__repr__", "__str__", "__format__", "__hash__", "is_shared", "is_pinned", "is_leaf", "_is_view", "type", "is_inference",
             "is_same_size", "stride", "storage_offset", "data_ptr"}


class LazyField(torch.Tensor):
    """A field of the adapter's result that has not been computed: metadata + (bundle, field name).  See the module docstring."""

    @staticmethod
    def __new__(cls, bundle: RawBundle, field: str, shape):
        t = torch.Tensor._make_wrapper_subclass(cls, tuple(int(x) for x in shape), dtype=torch.float32, device=bundle.device,
                                                requires_grad=False)
        t._s360_bundle, t._s360_field = bundle, field
        return t

    def materialise(self) -> Tensor:
        """The real tensor (reference values, autograd-connected to the encoder's outputs) in this field's current shape."""
        return self._s360_bundle.field(self._s360_field, tuple(self.shape))

    def __repr__(self):   # never materialise for a debugger's sake
        return f"LazyField({self._s360_field}, shape={tuple(self.shape)}, device={self.device})"

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        name = getattr(func, "__name__", "")
        me = next((a for a in args if isinstance(a, LazyField)), None)
        if name == "__get__" or name in _METADATA:      # .shape / .dtype / .device / .ndim / size() ...: the wrapper's own metadata
            if name in ("stride", "storage_offset", "data_ptr", "is_contiguous") and me is not None:
                return func(me.materialise(), *args[1:], **kwargs)
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        if me is not None and me is args[0] and name in _PURE_RESHAPES and not any(isinstance(a, LazyField) for a in args[1:]):
            if name in ("contiguous", "detach_", "requires_grad_"):
                return me
            # the new shape from a storage-less meta tensor: any pure reshape keeps the element order, hence stays lazy
            meta = torch.empty(tuple(me.shape), dtype=torch.float32, device="meta")
            margs = tuple(torch.empty(tuple(a.shape), device="meta") if isinstance(a, Tensor) else a for a in args[1:])
            new_shape = tuple(func(meta, *margs, **kwargs).shape)
            return LazyField(me._s360_bundle, me._s360_field, new_shape)
        if me is not None and name in ("permute", "transpose") and _identity_permutation(name, args, kwargs, me.dim()):
            return me
        # anything else: compute on the real tensors
        real = lambda a: a.materialise() if isinstance(a, LazyField) else a
        deep = lambda a: type(a)(deep(x) for x in a) if isinstance(a, (list, tuple)) else real(a)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*deep(tuple(args)), **{k: deep(v) for k, v in kwargs.items()})


    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        """Whatever reaches the dispatcher without passing __torch_function__ (C++ callers): compute on the real tensors."""
        from torch.utils._pytree import tree_map
        real = lambda a: a.materialise() if isinstance(a, LazyField) else a
        return func(*tree_map(real, tuple(args)), **tree_map(real, dict(kwargs or {})))


def _identity_permutation(name, args, kwargs, nd) -> bool:
    try:
        if name == "permute":
            dims = args[1] if len(args) == 2 and isinstance(args[1], (list, tuple)) else args[1:]
            return [int(d) % nd for d in dims] == list(range(nd))
        return len(args) == 3 and int(args[1]) % nd == int(args[2]) % nd
    except Exception:
        return False


def bundle_of(gaussians, d_sh: int = 25) -> Optional[RawBundle]:
    """The RawBundle behind a Gaussians container whose means / covariances / harmonics are UNTOUCHED LazyFields of one adapter call
    in the reference's flat layout ([b, G, 3], [b, G, 3, 3], [b, G, 3, 25]: src/model/types.py:7-12), else None."""
    try:
        m, c, h = gaussians.means, gaussians.covariances, gaussians.harmonics
    except AttributeError:
        return None
    if not (isinstance(m, LazyField) and isinstance(c, LazyField) and isinstance(h, LazyField)):
        return None
    bd = m._s360_bundle
    if c._s360_bundle is not bd or h._s360_bundle is not bd or (m._s360_field, c._s360_field, h._s360_field) != ("means", "covariances", "harmonics"):
        return None
    b = bd.shape5[0]
    g = math.prod(bd.shape5[1:])
    if tuple(m.shape) != (b, g, 3) or tuple(c.shape) != (b, g, 3, 3) or tuple(h.shape) != (b, g, 3, d_sh):
        return None
    op = gaussians.opacities
    if isinstance(op, LazyField) or tuple(op.shape) != (b, g):
        return None
    return bd


def make_adapter_class(base_cls=None, container_cls=None, *, sh_rotation="native", differentiable_means: bool = False, lazy: bool = True,
                       name: str = "GaussianAdapterERPFusedMI355X"):
    """A class with the reference adapter's constructor and forward contract (gaussian_adapter_erp.py:33-119) whose forward returns
    lazy fields (module docstring).  base_cls: the reference's own GaussianAdapterERP (install(adapter=True): isinstance checks and
    its `d_sh` / `d_in` properties keep working) or None; container_cls: the reference's adapter-side `Gaussians` dataclass (six
    fields, gaussian_adapter.py) or None = adapter.AdapterGaussians."""
    container = container_cls or _adapter.AdapterGaussians
    bases = (base_cls,) if base_cls is not None else (torch.nn.Module,)

    class _Adapter(*bases):
        def __init__(self, cfg) -> None:
            if base_cls is not None:
                super().__init__(cfg)                     # registers the reference's own sh_mask buffer, keeps self.cfg
            else:
                torch.nn.Module.__init__(self)
                self.cfg = cfg
                self.register_buffer("sh_mask", _adapter.sh_mask((int(cfg.sh_degree) + 1) ** 2), persistent=False)
            self._s360 = _adapter.GaussianAdapterERP(float(cfg.gaussian_scale_min), float(cfg.gaussian_scale_max), int(cfg.sh_degree),
                                                     sh_rotation=sh_rotation, differentiable_means=differentiable_means)
            self.s360_lazy = bool(lazy)

        if base_cls is None:
            @property
            def d_sh(self) -> int:
                return (int(self.cfg.sh_degree) + 1) ** 2

            @property
            def d_in(self) -> int:
                return 7 + 3 * self.d_sh

        # ---- what RawBundle calls back
        def rotation_blocks(self, ext):
            return self._s360.rotation_blocks(ext)

        def forward_eager(self, dataset_name, extrinsics, depths, opacities, raw_gaussians, image_shape, eps, sh_rot="build"):
            g = self._s360.forward(dataset_name, extrinsics, depths, opacities, raw_gaussians, image_shape, eps, sh_rot=sh_rot)
            return g

        def forward(self, dataset_name, extrinsics, depths, opacities, raw_gaussians, image_shape, eps: float = 1e-8):
            if not depths.is_cuda:
                raise RuntimeError("GaussianAdapterERP (splatter360_amd) runs on the GPU only: depths is a CPU tensor (no CPU path in the product)")
            if dataset_name not in _adapter.ERP_CONVENTIONS:
                raise Exception(f"no ERP convention for dataset {dataset_name!r} (src/geometry/utils360.py raises for it too)")
            d_sh = (int(self.cfg.sh_degree) + 1) ** 2
            if not self.s360_lazy or d_sh != 25 or depths.dim() != 5:
                g = self.forward_eager(dataset_name, extrinsics, depths, opacities, raw_gaussians, image_shape, eps)
                return container(means=g.means, covariances=g.covariances, scales=g.scales, rotations=g.rotations, harmonics=g.harmonics,
                                 opacities=g.opacities)
            b, v, r, srf, spp = (int(x) for x in depths.shape)
            bundle = RawBundle(self, dataset_name, extrinsics.reshape(b, v, 4, 4), depths, opacities, raw_gaussians, image_shape, eps)
            sh5 = (b, v, r, srf, spp)
            f = lambda nm, tail: LazyField(bundle, nm, sh5 + tail)
            return container(means=f("means", (3,)), covariances=f("covariances", (3, 3)), scales=f("scales", (3,)),
                             rotations=f("rotations", (4,)), harmonics=f("harmonics", (3, d_sh)), opacities=opacities)

    _Adapter.__name__ = _Adapter.__qualname__ = name
    return _Adapter
