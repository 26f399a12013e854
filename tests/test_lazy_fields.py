"""lazy.LazyField mechanics on the CPU (no kernels: a stub module stands in for the adapter kernels): the reference encoder's own
rearranges (/root/reference/src/model/encoder/encoder_costvolume.py:490-507) keep the fields lazy, every other operation computes on
the real tensors, `bundle_of` recognises exactly the untouched flat layout of src/model/types.py:7-12."""
from types import SimpleNamespace

import torch
from einops import rearrange

from splatter360_amd import lazy


class _StubModule:
    """Stands in for make_adapter_class(...)'s instance: counts materialisations, returns recognisable tensors."""

    def __init__(self):
        self.calls = 0

    def rotation_blocks(self, ext):
        return None

    def forward_eager(self, dataset_name, extrinsics, depths, opacities, raw, image_shape, eps, sh_rot=None):
        self.calls += 1
        sh5 = tuple(depths.shape)
        n = depths.numel()
        mk = lambda tail, k: (torch.arange(n * int(torch.tensor(tail).prod()), dtype=torch.float32) * k).reshape(*sh5, *tail) + depths.sum() * 0
        return SimpleNamespace(means=mk((3,), 1.0), covariances=mk((3, 3), 2.0), scales=mk((3,), 3.0), rotations=mk((4,), 4.0),
                               harmonics=mk((3, 25), 5.0), opacities=opacities)


def _bundle(b=2, v=2, r=6):
    mod = _StubModule()
    depths = torch.rand(b, v, r, 1, 1, requires_grad=True)
    op = torch.rand(b, v, r, 1, 1)
    raw = torch.randn(b, v, r, 1, 1, 82)
    ext = torch.eye(4).expand(b, v, 4, 4).contiguous()
    bd = lazy.RawBundle(mod, "hm3d", ext, depths, op, raw, (2, 3), 1e-8)
    sh5 = (b, v, r, 1, 1)
    g = SimpleNamespace(means=lazy.LazyField(bd, "means", sh5 + (3,)), covariances=lazy.LazyField(bd, "covariances", sh5 + (3, 3)),
                        harmonics=lazy.LazyField(bd, "harmonics", sh5 + (3, 25)), scales=lazy.LazyField(bd, "scales", sh5 + (3,)),
                        rotations=lazy.LazyField(bd, "rotations", sh5 + (4,)), opacities=op)
    return mod, bd, g


def _encoder_tail(g):
    """The four rearranges of encoder_costvolume.py:490-507."""
    return SimpleNamespace(means=rearrange(g.means, "b v r srf spp xyz -> b (v r srf spp) xyz"),
                           covariances=rearrange(g.covariances, "b v r srf spp i j -> b (v r srf spp) i j"),
                           harmonics=rearrange(g.harmonics, "b v r srf spp c d_sh -> b (v r srf spp) c d_sh"),
                           opacities=rearrange(1 * g.opacities, "b v r srf spp -> b (v r srf spp)"))


def test_the_encoders_rearranges_keep_the_fields_lazy_and_bundle_of_finds_them():
    mod, bd, g = _bundle()
    flat = _encoder_tail(g)
    assert isinstance(flat.means, lazy.LazyField) and tuple(flat.means.shape) == (2, 12, 3)
    assert isinstance(flat.covariances, lazy.LazyField) and tuple(flat.covariances.shape) == (2, 12, 3, 3)
    assert isinstance(flat.harmonics, lazy.LazyField) and tuple(flat.harmonics.shape) == (2, 12, 3, 25)
    assert flat.means.dtype == torch.float32 and flat.means.device.type == "cpu" and flat.means.dim() == 3 and flat.means.ndim == 3
    assert isinstance(flat.means, torch.Tensor)                     # what jaxtyping / beartype check on the reference's dataclass
    assert mod.calls == 0
    assert lazy.bundle_of(flat) is bd
    assert "LazyField" in repr(flat.means) and mod.calls == 0


def test_any_other_operation_materialises_once_with_the_adapters_values():
    mod, bd, g = _bundle()
    flat = _encoder_tail(g)
    want = mod.forward_eager("hm3d", None, bd.depths, bd.opacities, bd.raw, (2, 3), 1e-8)
    mod.calls = 0
    m0 = flat.means[0]                                              # indexing: a real tensor
    assert not isinstance(m0, lazy.LazyField) and mod.calls == 1
    assert torch.equal(m0, want.means.reshape(2, 12, 3)[0])
    s = (flat.covariances * 2).sum() + flat.harmonics.mean() + torch.cat([flat.means, flat.means], 1).sum()
    assert mod.calls == 1 and not isinstance(s, lazy.LazyField)     # one materialisation serves every field
    assert torch.allclose(s, (want.covariances * 2).sum() + want.harmonics.mean() + 2 * want.means.sum())
    assert torch.equal(flat.harmonics.permute(0, 1, 3, 2), want.harmonics.reshape(2, 12, 3, 25).permute(0, 1, 3, 2))
    assert torch.equal(g.scales.clone(), want.scales) and torch.equal(g.rotations + 0, want.rotations)
    assert s.requires_grad                                          # the stub ties its outputs to depths: autograd passes through


def test_bundle_of_rejects_touched_or_mixed_containers():
    mod, bd, g = _bundle()
    flat = _encoder_tail(g)
    assert lazy.bundle_of(SimpleNamespace(means=flat.means.clone(), covariances=flat.covariances, harmonics=flat.harmonics, opacities=flat.opacities)) is None
    assert lazy.bundle_of(SimpleNamespace(means=flat.means, covariances=flat.covariances, harmonics=flat.harmonics, opacities=flat.opacities[:, :5])) is None
    _, bd2, g2 = _bundle()
    assert lazy.bundle_of(SimpleNamespace(means=flat.means, covariances=_encoder_tail(g2).covariances, harmonics=flat.harmonics, opacities=flat.opacities)) is None
    assert lazy.bundle_of(SimpleNamespace(means=g.means, covariances=g.covariances, harmonics=g.harmonics, opacities=g.opacities)) is None   # not flattened
    assert lazy.bundle_of(SimpleNamespace(means=flat.harmonics, covariances=flat.covariances, harmonics=flat.harmonics, opacities=flat.opacities)) is None
    assert lazy.bundle_of(object()) is None
    assert lazy.bundle_of(flat) is bd


def test_the_adapter_class_keeps_the_reference_contract_and_has_no_cpu_path():
    import pytest
    cls = lazy.make_adapter_class()
    mod = cls(SimpleNamespace(gaussian_scale_min=0.5, gaussian_scale_max=15.0, sh_degree=4))
    assert mod.d_sh == 25 and mod.d_in == 82 and tuple(mod.sh_mask.shape) == (25,)
    assert float(mod.sh_mask[0]) == 1.0 and abs(float(mod.sh_mask[1]) - 0.025) < 1e-9 and abs(float(mod.sh_mask[24]) - 0.1 * 0.25 ** 4) < 1e-10
    with pytest.raises(RuntimeError):
        mod.forward("hm3d", torch.eye(4).reshape(1, 1, 1, 1, 1, 4, 4), torch.ones(1, 1, 8, 1, 1), torch.ones(1, 1, 8, 1, 1), torch.zeros(1, 1, 8, 1, 1, 82), (2, 4))
