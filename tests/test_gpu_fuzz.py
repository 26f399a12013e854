"""Seeded fuzz of the HIP rasteriser against the CPU oracle: random cloud sizes, image sizes that are not
multiples of the tile, random camera positions / cube-face orientations, near / far, background, SH degree
0-4 or precomputed colours, random pixel gradients.  Forward bars as in tests/test_gpu_parity.py (integer
intermediates bit-exact vs the float32 oracle, pixels <= 1e-5 mean L1).  Gradients are judged against the
float64 oracle: random clouds contain ill-conditioned splats (near-singular 2-D covariance) on which the float32
oracle itself is off by > 1e-3, so the bar is  err(HIP, f64) <= max(5e-4, 2 * err(oracle f32, f64))."""
import numpy as np
import pytest

from helpers import boundary_tensors, face_settings
from oracle import oracle
from splatter360_amd import synthetic
from test_gpu_parity import check_forward, run_hip

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 6000))
    h, w = int(rng.integers(8, 150)), int(rng.integers(8, 150))
    face = int(rng.integers(0, 6))
    pos = tuple(rng.uniform(-1.0, 1.0, 3))
    near = float(rng.choice([0.05, 0.1, 0.5, 1.0]))
    deg = int(rng.integers(0, 5))
    use_sh = bool(rng.integers(0, 4))          # 3 in 4 cases use SH
    bg = rng.uniform(0, 1, 3)
    cloud = synthetic.uniform_cloud(n, seed=seed, extent=float(rng.uniform(1.0, 4.0)),
                                    scale_range=(0.01, float(rng.uniform(0.05, 0.6))))
    S = face_settings(face, h, w, near=near, far=near * 100.0, position=pos, bg=bg)
    means, cov6, shs, opac = boundary_tensors(cloud, S["scale"])
    S["sh_degree"] = deg
    colors = None
    if not use_sh:
        colors, shs = rng.uniform(0, 1, (n, 3)).astype(np.float32), None
    gimg = rng.standard_normal((3, h, w)).astype(np.float32)
    return S, means, cov6, shs, opac, colors, gimg, (n, h, w)


@pytest.mark.parametrize("seed", range(24))
def test_fuzz_against_oracle(gpu, seed):
    S, means, cov6, shs, opac, colors, gimg, (n, h, w) = _case(seed)
    orc = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, colors_precomp=colors)
    f = orc.forward()
    og32 = orc.backward(gimg)
    o64 = oracle.rasterize(S, means3D=means, cov3D_precomp=cov6, opacities=opac, shs=shs, colors_precomp=colors, dtype=np.float64)
    o64.forward()
    og64 = o64.backward(gimg)
    hh = run_hip(S, means, cov6, shs, opac, gpu, colors=colors, grad_image=gimg)
    check_forward(hh, f, n, h, w)
    for k in ("means3D", "means2D", "cov3D", "opacities", "shs", "colors_precomp"):
        if og64.get(k) is None:
            continue
        ref = np.asarray(og64[k], np.float64).reshape(-1)
        scale = np.abs(ref).max() + 1e-30
        e_hip = np.abs(hh["grads"][k].reshape(-1) - ref).max() / scale
        e_o32 = np.abs(np.asarray(og32[k], np.float64).reshape(-1) - ref).max() / scale
        assert e_hip <= max(5e-4, 2.0 * e_o32), (k, e_hip, e_o32)
