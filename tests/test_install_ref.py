"""`splatter360_amd.install()` against the reference's own plugin seam (build container only: needs /root/reference).

The reference's decoder package is imported for real (its __init__ runs: DECODERS, get_decoder —
/root/reference/src/model/decoder/__init__.py:5-13) under the SURVEY Appendix-B stubs for the absent third-party modules; the
rasteriser it imports is THIS repository's drop-in `diff_gaussian_rasterization`.  Runs on CPU: construction and registry only —
the forward of the registered class is exercised on the GPU from the committed capture (tests/test_gpu_install.py)."""
import importlib
import subprocess
import sys
import textwrap
import types
from pathlib import Path
from types import SimpleNamespace

import pytest

REF = Path("/root/reference")
ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.skipif(not (REF / "src/model/decoder/__init__.py").exists(), reason="the reference is only present in the build container")

PRELUDE = textwrap.dedent(f"""
    import sys, types
    from pathlib import Path
    REF = {str(REF)!r}
    jt = types.ModuleType("jaxtyping")
    class _Ann:
        def __getitem__(self, item):
            return object
    for name in ("Float", "Int", "Int64", "Int32", "UInt8", "Bool", "Shaped"):
        setattr(jt, name, _Ann())
    sys.modules["jaxtyping"] = jt
    sys.modules["cv2"] = types.ModuleType("cv2")
    ds = types.ModuleType("src.dataset"); ds.__path__ = [REF + "/src/dataset"]; ds.DatasetCfg = object
    sys.modules["src.dataset"] = ds
    for pkg in ("src.model.encoder", "src.model.encoder.costvolume"):
        m = types.ModuleType(pkg); m.__path__ = [REF + "/" + pkg.replace(".", "/")]; sys.modules[pkg] = m
    sys.path.insert(0, REF)
    sys.path.insert(0, {str(ROOT)!r})
    from types import SimpleNamespace
    cfg, dcfg = SimpleNamespace(name="splatting_cuda"), SimpleNamespace(background_color=[0.1, 0.2, 0.3])
""")


def _run(body: str) -> str:
    r = subprocess.run([sys.executable, "-c", PRELUDE + textwrap.dedent(body)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


def test_install_registers_the_fused_decoder_under_the_reference_key():
    out = _run("""
        import splatter360_amd
        cls = splatter360_amd.install()
        from src.model.decoder import DECODERS, get_decoder
        from src.model.decoder.decoder import Decoder, DecoderOutput
        import diff_gaussian_rasterization, splatter360_amd.rasterizer as R
        assert diff_gaussian_rasterization.GaussianRasterizer is R.GaussianRasterizer     # the reference imports THIS drop-in
        dec = get_decoder(cfg, dcfg)
        assert type(dec) is cls and DECODERS["splatting_cuda"] is cls and isinstance(dec, Decoder)
        assert cls.replaced.__name__ == "DecoderSplattingCUDA"
        assert [round(float(x), 4) for x in dec.background_color] == [0.1, 0.2, 0.3]
        assert dec.cfg is cfg and dec.dataset_cfg is dcfg
        # the forward contract is the reference's (decoder.py:37-48): same parameter names, in order
        import inspect
        want = list(inspect.signature(Decoder.forward).parameters)
        assert list(inspect.signature(cls.forward).parameters) == want, want
        # no CPU path: a CPU call fails loudly instead of falling back
        import torch
        from src.model.types import Gaussians
        g = Gaussians(torch.zeros(1, 4, 3), torch.eye(3).expand(1, 4, 3, 3).contiguous(), torch.zeros(1, 4, 3, 25), torch.ones(1, 4))
        e = torch.eye(4).expand(1, 6, 4, 4).contiguous(); k = torch.eye(3).expand(1, 6, 3, 3).contiguous()
        try:
            dec.forward(g, e, k, torch.full((1, 6), 0.1), torch.full((1, 6), 10.0), (16, 16))
            raise SystemExit("a CPU call must raise")
        except RuntimeError as ex:
            assert "GPU" in str(ex) or "libs360" in str(ex), ex
        splatter360_amd.uninstall()
        assert DECODERS["splatting_cuda"].__name__ == "DecoderSplattingCUDA"
        print("ok")
    """)
    assert out.strip().endswith("ok")


def test_lazy_install_patches_when_the_reference_imports_its_decoder_package():
    out = _run("""
        import splatter360_amd
        assert splatter360_amd.install(lazy=True) is None
        assert "src.model.decoder" not in sys.modules
        from src.model.decoder import get_decoder            # the reference's own import (src/main.py:33)
        dec = get_decoder(cfg, dcfg)
        assert type(dec).__name__ == "DecoderSplattingFusedMI355X", type(dec)
        print("ok")
    """)
    assert out.strip().endswith("ok")


def test_lazy_install_survives_a_competing_finder_that_resolves_src_itself():
    """jaxtyping's install_import_hook (src/main.py:22-36) puts its own finder at sys.meta_path[0] and resolves `src.*`
    through PathFinder directly, so the lazy patcher never sees `src.model.decoder`.  Emulated here; the registry must still be
    patched by the time the imports of that `with` block are over (any non-`src` import after the package triggers it)."""
    out = _run("""
        import importlib.abc, importlib.machinery
        import splatter360_amd
        assert splatter360_amd.install(lazy=True) is None
        class Competing(importlib.abc.MetaPathFinder):          # what jaxtyping's hook does: first in line, PathFinder for src.*
            def find_spec(self, fullname, path, target=None):
                if fullname == "src" or fullname.startswith("src."):
                    return importlib.machinery.PathFinder.find_spec(fullname, path, target)
                return None
        sys.meta_path.insert(0, Competing())
        from src.model.decoder import DECODERS, get_decoder
        assert DECODERS["splatting_cuda"].__name__ == "DecoderSplattingCUDA"      # not patched yet: the patcher was bypassed
        import colorsys                                                            # any later non-src import (main.py: the encoder's)
        dec = get_decoder(cfg, dcfg)
        assert type(dec).__name__ == "DecoderSplattingFusedMI355X", type(dec)
        assert not any(type(f).__name__ == "_LazyPatcher" for f in sys.meta_path)
        print("ok")
    """)
    assert out.strip().endswith("ok")


ADAPTER_PRELUDE = textwrap.dedent("""
    # the adapter module's own imports under the Appendix-B stubs: e3nn is absent, so rotate_sh's module is a stub (its import
    # is what pulls e3nn in, src/misc/sh_rotation.py:1-8); the common package's __init__ is not executed
    import torch
    shr = types.ModuleType("src.misc.sh_rotation")
    shr.rotate_sh = lambda sh, rotations: sh
    sys.modules["src.misc.sh_rotation"] = shr
    m = types.ModuleType("src.model.encoder.common"); m.__path__ = [REF + "/src/model/encoder/common"]; sys.modules["src.model.encoder.common"] = m
""")


def test_install_adapter_replaces_the_class_the_encoder_instantiates():
    """install(adapter=True) BEFORE the reference imports its encoder: the module that defines GaussianAdapterERP
    (gaussian_adapter_erp.py:33) is patched as it loads, so the encoder's own `from .common.gaussian_adapter_erp import
    GaussianAdapterERP` (encoder_costvolume.py:19, instantiated at :185) binds the replacement — a subclass of the reference's class
    with its constructor, `d_sh` / `d_in` and `sh_mask`, returning the reference's own adapter-side container."""
    out = _run(ADAPTER_PRELUDE + textwrap.dedent("""
        import splatter360_amd
        splatter360_amd.install(lazy=True, adapter=True)
        assert "src.model.encoder.common.gaussian_adapter_erp" not in sys.modules
        # what encoder_costvolume.py:19 executes
        from src.model.encoder.common.gaussian_adapter_erp import GaussianAdapterERP, GaussianAdapterERPCfg
        amod = sys.modules["src.model.encoder.common.gaussian_adapter_erp"]
        assert GaussianAdapterERP.__name__ == "GaussianAdapterERPFusedMI355X", GaussianAdapterERP
        ref_cls = GaussianAdapterERP.replaced
        assert ref_cls.__name__ == "GaussianAdapterERP" and issubclass(GaussianAdapterERP, ref_cls)
        cfg_a = GaussianAdapterERPCfg(gaussian_scale_min=0.5, gaussian_scale_max=15.0, sh_degree=4)       # config/model/encoder/costvolume.yaml:14-16
        mod = GaussianAdapterERP(cfg_a)                                                                    # encoder_costvolume.py:185
        ref = ref_cls(cfg_a)
        assert isinstance(mod, ref_cls) and mod.cfg is cfg_a and mod.d_sh == ref.d_sh == 25 and mod.d_in == ref.d_in == 82
        assert torch.equal(mod.sh_mask, ref.sh_mask) and "sh_mask" not in mod.state_dict()                # non-persistent, like the reference's
        assert list(mod.state_dict().keys()) == list(ref.state_dict().keys())                             # checkpoints load unchanged
        import inspect
        assert list(inspect.signature(mod.forward).parameters) == list(inspect.signature(ref.forward).parameters)
        try:
            mod.forward("hm3d", torch.eye(4).reshape(1, 1, 1, 1, 1, 4, 4), torch.ones(1, 1, 8, 1, 1), torch.ones(1, 1, 8, 1, 1), torch.zeros(1, 1, 8, 1, 1, 82), (2, 4))
            raise SystemExit("a CPU call must raise")
        except RuntimeError as ex:
            assert "GPU" in str(ex), ex
        # installing again is idempotent; uninstall puts the reference's class back
        splatter360_amd.install(lazy=True, adapter=True)
        assert amod.GaussianAdapterERP is GaussianAdapterERP
        splatter360_amd.uninstall()
        assert amod.GaussianAdapterERP is ref_cls
        print("ok")
    """))
    assert out.strip().endswith("ok")


def test_install_adapter_after_the_encoder_was_imported_patches_the_bound_name_too():
    out = _run(ADAPTER_PRELUDE + textwrap.dedent("""
        import importlib
        amod = importlib.import_module("src.model.encoder.common.gaussian_adapter_erp")
        # a stand-in for the encoder module that already executed its `from .common.gaussian_adapter_erp import GaussianAdapterERP`
        enc = types.ModuleType("src.model.encoder.encoder_costvolume"); enc.GaussianAdapterERP = amod.GaussianAdapterERP
        sys.modules["src.model.encoder.encoder_costvolume"] = enc
        import splatter360_amd
        splatter360_amd.install(lazy=True, adapter=True, adapter_options=dict(sh_rotation="identity"))
        assert amod.GaussianAdapterERP.__name__ == "GaussianAdapterERPFusedMI355X" and enc.GaussianAdapterERP is amod.GaussianAdapterERP
        # the lazy fields pass through the reference's own containers: the adapter-side dataclass and src/model/types.py's Gaussians
        from splatter360_amd import lazy
        from src.model.types import Gaussians
        from einops import rearrange
        bd = lazy.RawBundle(None, "hm3d", torch.eye(4).expand(1, 2, 4, 4), torch.ones(1, 2, 8, 1, 1), torch.ones(1, 2, 8, 1, 1), torch.zeros(1, 2, 8, 1, 1, 82), (2, 4), 1e-8)
        f = lambda n, t: lazy.LazyField(bd, n, (1, 2, 8, 1, 1) + t)
        a = amod.Gaussians(means=f("means", (3,)), covariances=f("covariances", (3, 3)), scales=f("scales", (3,)), rotations=f("rotations", (4,)),
                           harmonics=f("harmonics", (3, 25)), opacities=bd.opacities)
        g = Gaussians(rearrange(a.means, "b v r srf spp xyz -> b (v r srf spp) xyz"), rearrange(a.covariances, "b v r srf spp i j -> b (v r srf spp) i j"),
                      rearrange(a.harmonics, "b v r srf spp c d_sh -> b (v r srf spp) c d_sh"), rearrange(1 * a.opacities, "b v r srf spp -> b (v r srf spp)"))   # encoder_costvolume.py:490-507
        assert lazy.bundle_of(g) is bd
        print("ok")
    """))
    assert out.strip().endswith("ok")
