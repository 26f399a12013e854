"""Seeded synthetic Gaussian clouds shaped like the splatter360 encoder's output.

There is no dataset or checkpoint on the build/bench machines, so the benchmark workload
(BASELINE.json configs 1-3,5) is an "encoder-like" cloud: one Gaussian per pixel of each context
panorama, un-projected along the reference's ERP ray convention
(/root/reference/src/geometry/utils360.py:93-104,148-153), with the scale / opacity / SH statistics
of the Gaussian adapter (src/model/encoder/common/gaussian_adapter_erp.py:38-47,63-77,
config/model/encoder/costvolume.yaml:14-16).  Generated with numpy on the host, deterministic
given the seed.  SURVEY.md §8(d) is the specification.
"""
from __future__ import annotations

import math

import numpy as np


def erp_ray_directions(h: int, w: int) -> np.ndarray:
    """[h,w,3] unit ray of each ERP pixel centre: theta=(0.5-(x+.5)/W)*2pi, phi=-((y+.5)/H-.5)*pi,
    dir=(cos(phi)sin(theta), sin(phi), cos(phi)cos(theta))  (utils360.py:93-104,148-153)."""
    x = (np.arange(w, dtype=np.float64) + 0.5) / w
    y = (np.arange(h, dtype=np.float64) + 0.5) / h
    theta = (0.5 - x) * 2 * math.pi
    phi = -(y - 0.5) * math.pi
    theta, phi = np.meshgrid(theta, phi)
    return np.stack([np.cos(phi) * np.sin(theta), np.sin(phi), np.cos(phi) * np.cos(theta)], -1)


def _random_rotations(rng: np.random.Generator, n: int) -> np.ndarray:
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(n, 3, 3)


def sh_band_mask(d_sh: int) -> np.ndarray:
    """1 for DC, 0.1*0.25^degree for higher bands (gaussian_adapter_erp.py:38-47)."""
    deg = int(round(math.sqrt(d_sh))) - 1
    mask = np.ones(d_sh)
    for l in range(1, deg + 1):
        mask[l * l:(l + 1) * (l + 1)] = 0.1 * 0.25 ** l
    return mask


def encoder_like_cloud(pano_h: int = 512, pano_w: int = 1024, n_context: int = 2, d_sh: int = 25,
                       seed: int = 0, depth_range=(0.5, 8.0)) -> dict:
    """Gaussians container of the reference (src/model/types.py:7-12) for batch 1, as float32
    numpy: means[G,3], covariances[G,3,3], harmonics[G,3,d_sh] (channel-major), opacities[G].
    G = n_context*pano_h*pano_w (= 1 048 576 at the default sizes)."""
    rng = np.random.default_rng(seed)
    centres = [np.array([-0.4, 0.0, 0.1]), np.array([0.4, 0.0, -0.1]), np.array([0.0, 0.3, 0.4]),
               np.array([0.1, -0.3, -0.4])]
    dirs = erp_ray_directions(pano_h, pano_w).reshape(-1, 3)
    n = dirs.shape[0]
    means, covs = [], []
    for c in range(n_context):
        depth = np.exp(rng.uniform(math.log(depth_range[0]), math.log(depth_range[1]), n))
        means.append(centres[c % 4] + dirs * depth[:, None])
        s = (0.5 + 14.5 / (1 + np.exp(-rng.standard_normal((n, 3))))) * depth[:, None] / pano_w
        r = _random_rotations(rng, n)
        covs.append(np.einsum("nij,nj,nkj->nik", r, s * s, r))
    means = np.concatenate(means)
    covs = np.concatenate(covs)
    g = means.shape[0]
    sh = rng.standard_normal((g, 3, d_sh)) * sh_band_mask(d_sh)
    sh[:, :, 0] = sh[:, :, 0] * 0.6  # DC: colours mostly inside (0,1) after the +0.5 offset
    opac = 1 / (1 + np.exp(-rng.standard_normal(g)))
    f = np.float32
    return dict(means=means.astype(f), covariances=covs.astype(f), harmonics=sh.astype(f),
                opacities=opac.astype(f))


def surface_like_cloud(pano_h: int = 512, pano_w: int = 1024, n_context: int = 2, d_sh: int = 25, seed: int = 0) -> dict:
    """What a TRAINED encoder emits rather than a random one (cf. /root/reference/src/model/encoder/encoder_costvolume.py:490-507:
    per-pixel depth from a cost volume, opacity from the matching confidence): a spatially coherent depth field per context
    panorama — a few low-frequency waves over (theta, phi) in log depth, 1.2 ... 6 units — with a little per-pixel noise, opacity
    >= 0.9, footprints of about one to two context pixels.  Same layouts and SH statistics as encoder_like_cloud.  The second
    context panorama sees (mostly) the same surface from 0.8 units away, so most of its Gaussians are hidden behind the first's:
    long lists, early saturation."""
    rng = np.random.default_rng(seed)
    centres = [np.array([-0.4, 0.0, 0.1]), np.array([0.4, 0.0, -0.1]), np.array([0.0, 0.3, 0.4]), np.array([0.1, -0.3, -0.4])]
    dirs = erp_ray_directions(pano_h, pano_w).reshape(-1, 3)
    n = dirs.shape[0]
    theta = np.arctan2(dirs[:, 0], dirs[:, 2])
    phi = np.arcsin(np.clip(dirs[:, 1], -1, 1))
    means, covs = [], []
    for c in range(n_context):
        logd = math.log(2.7) + 0.45 * np.sin(2 * theta + 0.3 * c) * np.cos(phi) + 0.25 * np.cos(3 * theta - phi) + 0.12 * np.sin(5 * phi + c)
        depth = np.exp(logd + 0.01 * rng.standard_normal(n))
        means.append(centres[c % 4] + dirs * depth[:, None])
        s = (0.8 + 1.7 / (1 + np.exp(-rng.standard_normal((n, 3))))) * depth[:, None] * (math.pi / pano_h) * 0.5
        r = _random_rotations(rng, n)
        covs.append(np.einsum("nij,nj,nkj->nik", r, s * s, r))
    means = np.concatenate(means)
    covs = np.concatenate(covs)
    g = means.shape[0]
    sh = rng.standard_normal((g, 3, d_sh)) * sh_band_mask(d_sh)
    sh[:, :, 0] = sh[:, :, 0] * 0.6
    opac = rng.uniform(0.9, 0.99, g)
    f = np.float32
    return dict(means=means.astype(f), covariances=covs.astype(f), harmonics=sh.astype(f), opacities=opac.astype(f))


def uniform_cloud(g: int, d_sh: int = 25, seed: int = 0, extent: float = 5.0,
                  scale_range=(0.01, 0.15)) -> dict:
    """Stress / small-test variant: means U[-extent,extent]^3, log-uniform scales."""
    rng = np.random.default_rng(seed)
    means = rng.uniform(-extent, extent, (g, 3))
    s = np.exp(rng.uniform(math.log(scale_range[0]), math.log(scale_range[1]), (g, 3)))
    r = _random_rotations(rng, g)
    covs = np.einsum("nij,nj,nkj->nik", r, s * s, r)
    sh = rng.standard_normal((g, 3, d_sh)) * sh_band_mask(d_sh)
    sh[:, :, 0] *= 0.6
    opac = 1 / (1 + np.exp(-rng.standard_normal(g)))
    f = np.float32
    return dict(means=means.astype(f), covariances=covs.astype(f), harmonics=sh.astype(f),
                opacities=opac.astype(f))


def target_pano_pose(position=(0.0, 0.0, 0.0)) -> np.ndarray:
    """[4,4] float32 identity-rotation panorama c2w at `position`."""
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = position
    return m
