"""rotate_sh's matrices from FIRST PRINCIPLES: the reference rotates the SH coefficient blocks with
`e3nn.o3.wigner_D(l, *matrix_to_angles(R))` (/root/reference/src/misc/sh_rotation.py:19-24, applied at
src/model/encoder/common/gaussian_adapter_erp.py:113).  e3nn is not installed here, so the product's matrices
(`s360_sh_rotation_blocks`, checked so far against a least-squares fit over this project's own SH polynomials and a handful of
recalled e3nn members) are derived here a second, independent way — all (2l+1)^2 entries of every block, l = 0..4:

  1. complex Wigner-D in the textbook (z-polar, Condon-Shortley) convention, closed form:
         D^l_{m'm}(a, b, c) = exp(-i m' a) d^l_{m'm}(b) exp(-i m c),      R = Rz(a) Ry(b) Rz(c),
         d^l_{m'm}(b) = sum_k (-1)^(k-m+m') sqrt((l+m)!(l-m)!(l+m')!(l-m')!) / ((l+m-k)! k! (l-k-m')! (k-m+m')!)
                               cos(b/2)^(2l-2k+m-m') sin(b/2)^(2k-m+m')
     (exact rational prefactors via `fractions.Fraction`, float64 trigonometry), with the transformation law
         Y_l^m(R^-1 r) = sum_m' D^l_{m'm}(R) Y_l^{m'}(r)
     VERIFIED IN THIS FILE against scipy's complex spherical harmonics on random directions — so a slip of memory in the closed
     form or in the Euler-angle extraction cannot pass;
  2. the unitary change of basis to REAL harmonics, Wikipedia's convention (= no Condon-Shortley phase in the real functions):
         Y_{l,m>0} = (-1)^m sqrt2 Re Y_l^m  ~ +P_l^m cos(m phi),   Y_{l,0} = Y_l^0,   Y_{l,m<0} = (-1)^m sqrt2 Im Y_l^|m| ~ +P_l^|m| sin(|m| phi),
     ordered m = -l..l;
  3. e3nn's axis convention, written out: polar axis y, azimuth measured from z towards x — i.e. e3nn's real harmonics are the
     textbook real harmonics evaluated in the frame (x', y', z') = (z, x, y); its l = 1 block is (x, y, z) itself, so D^1(R) = R;
     "component" normalisation, which a rotation matrix does not see.  The e3nn-basis matrix of a rotation R is therefore the
     textbook one of P R P^T with the cyclic permutation P: (x, y, z) -> (z, x, y).

Then  Y(R d) = D(R) Y(d)  fixes  D_real = U conj(D_complex(P R P^T)) U^H.   Compared entry by entry with the CPU checker
(oracle/adapter_ref.wigner_blocks, float64: <= 1e-10) and with the HIP kernel (float32 output: <= 2e-6) on random rotations, including
rotations with beta near 0 and pi (the Euler-angle singularities).  What stays unpinned is step 3 alone — that e3nn uses this basis;
its documented l <= 2 polynomials are asserted against it in tests/test_adapter_cpu.py."""
import math
from fractions import Fraction

import numpy as np
import pytest
from scipy.spatial.transform import Rotation

from oracle import adapter_ref


def _small_d(l: int, beta: float) -> np.ndarray:
    """[2l+1, 2l+1] Wigner small-d matrix, rows m' = -l..l, columns m = -l..l (closed form of the module docstring)."""
    c, s = math.cos(beta / 2), math.sin(beta / 2)
    f = math.factorial
    d = np.zeros((2 * l + 1, 2 * l + 1))
    for mp in range(-l, l + 1):
        for m in range(-l, l + 1):
            pref = Fraction(f(l + m) * f(l - m) * f(l + mp) * f(l - mp))
            tot = 0.0
            for k in range(max(0, m - mp), min(l + m, l - mp) + 1):
                den = f(l + m - k) * f(k) * f(l - k - mp) * f(k - m + mp)
                # sqrt(pref) / den, kept exact as sqrt of a rational
                coef = math.sqrt(pref / (den * den))
                tot += (-1) ** (k - m + mp) * coef * c ** (2 * l - 2 * k + m - mp) * s ** (2 * k - m + mp)
            d[mp + l, m + l] = tot
    return d


def _zyz(R: np.ndarray):
    """R = Rz(a) Ry(b) Rz(c), b in [0, pi]."""
    sb = math.hypot(R[0, 2], R[1, 2])
    b = math.atan2(sb, R[2, 2])          # (acos(R22) loses half the digits next to 0 and pi)
    if sb > 1e-12:
        a = math.atan2(R[1, 2], R[0, 2])
        c = math.atan2(R[2, 1], -R[2, 0])
    else:               # gimbal: only a +- c is defined
        a = math.atan2(R[1, 0], R[0, 0]) if R[2, 2] > 0 else math.atan2(-R[1, 0], -R[0, 0])
        c = 0.0
    return a, b, c


def _complex_D(l: int, R: np.ndarray) -> np.ndarray:
    a, b, c = _zyz(R)
    m = np.arange(-l, l + 1)
    return np.exp(-1j * m[:, None] * a) * _small_d(l, b) * np.exp(-1j * m[None, :] * c)


def _complex_sh(l: int, r: np.ndarray) -> np.ndarray:
    """[N, 2l+1] textbook complex harmonics Y_l^m(r), m = -l..l (scipy: Condon-Shortley phase, z polar)."""
    from scipy.special import sph_harm_y
    r = r / np.linalg.norm(r, axis=1, keepdims=True)
    theta = np.arccos(np.clip(r[:, 2], -1, 1))
    phi = np.arctan2(r[:, 1], r[:, 0])
    return np.stack([sph_harm_y(l, m, theta, phi) for m in range(-l, l + 1)], 1)


def _U(l: int) -> np.ndarray:
    """Real harmonics (rows m = -l..l) from complex ones (columns m = -l..l): Y_real = U Y_complex, unitary."""
    U = np.zeros((2 * l + 1, 2 * l + 1), complex)
    U[l, l] = 1.0
    for m in range(1, l + 1):
        # Y_{l,+m} = (-1)^m sqrt2 Re Y_l^m = ((-1)^m Y_l^m + Y_l^-m) / sqrt2        (Y_l^-m = (-1)^m conj Y_l^m)
        U[l + m, l + m] = (-1) ** m / math.sqrt(2)
        U[l + m, l - m] = 1 / math.sqrt(2)
        # Y_{l,-m} = (-1)^m sqrt2 Im Y_l^m = ((-1)^m Y_l^m - Y_l^-m) / (i sqrt2)
        U[l - m, l + m] = (-1) ** m / (1j * math.sqrt(2))
        U[l - m, l - m] = -1 / (1j * math.sqrt(2))
    return U


P_E3NN = np.array([[0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])      # (x, y, z) -> (x', y', z') = (z, x, y): e3nn's polar axis y becomes z'


def wigner_blocks_closed_form(rotations: np.ndarray, d_sh: int) -> np.ndarray:
    """[V,3,3] -> [V,d_sh,d_sh] block-diagonal D with Y(R d) = D(R) Y(d) in e3nn's real basis, from the closed form."""
    rot = np.asarray(rotations, np.float64).reshape(-1, 3, 3)
    out = np.zeros((rot.shape[0], d_sh, d_sh))
    for v in range(rot.shape[0]):
        Rp = P_E3NN @ rot[v] @ P_E3NN.T
        for l in range(math.isqrt(d_sh)):
            U = _U(l)
            D = U @ np.conj(_complex_D(l, Rp)) @ U.conj().T
            assert np.abs(D.imag).max() < 1e-12
            out[v, l * l:(l + 1) ** 2, l * l:(l + 1) ** 2] = D.real
    return out


def _test_rotations():
    R = list(Rotation.random(24, random_state=11).as_matrix())
    for ang in (1e-7, 1e-4, math.pi - 1e-4, math.pi - 1e-7, 0.0, math.pi):      # beta at / next to the Euler singularities (in the e3nn frame too)
        R.append(Rotation.from_euler("zyz", [0.3, ang, -1.1]).as_matrix())
        R.append(P_E3NN.T @ Rotation.from_euler("zyz", [2.0, ang, 0.4]).as_matrix() @ P_E3NN)
    R.append(np.eye(3))
    return np.stack(R)


def test_closed_form_obeys_the_textbook_transformation_law_of_scipys_complex_harmonics():
    """Step 1 self-check: Y_l^m(R^-1 r) = sum_m' D^l_{m'm}(R) Y_l^m'(r) with scipy's harmonics — pins the closed form, its index
    convention and the Euler-angle extraction independently of anything else in this repository."""
    rng = np.random.default_rng(0)
    r = rng.standard_normal((50, 3))
    for R in _test_rotations()[::3]:
        for l in range(5):
            D = _complex_D(l, R)
            lhs = _complex_sh(l, r @ R)                    # rows: Y(R^-1 r)   (r @ R = (R^T r^T)^T)
            rhs = _complex_sh(l, r) @ D                    # sum_m' Y^{m'}(r) D_{m'm}
            np.testing.assert_allclose(lhs, rhs, atol=2e-12)
            np.testing.assert_allclose(D @ D.conj().T, np.eye(2 * l + 1), atol=1e-12)


def test_real_basis_of_the_oracle_is_the_textbook_real_basis_in_the_e3nn_frame():
    """Steps 2 + 3: adapter_ref.e3nn_real_sh(l, d) = U Y_complex(P d) up to the per-degree 'component' normalisation constant
    sqrt(4 pi / (2l+1)) x sqrt(2l+1) = sqrt(4 pi) — same ORDER and SIGNS of all 2l+1 members, which is all a rotation matrix sees."""
    rng = np.random.default_rng(1)
    d = rng.standard_normal((60, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for l in range(5):
        real = (_complex_sh(l, d @ P_E3NN.T) @ _U(l).T)
        assert np.abs(real.imag).max() < 1e-12
        ours = adapter_ref.e3nn_real_sh(l, d)
        scale = np.linalg.norm(ours) / np.linalg.norm(real.real)
        np.testing.assert_allclose(real.real * scale, ours, atol=1e-12)
        np.testing.assert_allclose(scale, math.sqrt(4 * math.pi / (2 * l + 1)), rtol=1e-12)   # e3nn_real_sh: P_l(1) = 1 normalisation


def test_every_entry_of_the_cpu_checker_equals_the_closed_form():
    R = _test_rotations()
    want = wigner_blocks_closed_form(R, 25)
    got = adapter_ref.wigner_blocks(R, 25)
    np.testing.assert_allclose(got, want, atol=1e-10)
    for v in range(R.shape[0]):
        np.testing.assert_allclose(want[v, 1:4, 1:4], R[v], atol=1e-12)        # D^1 = R in e3nn's (x, y, z) basis
        np.testing.assert_allclose(want[v] @ want[v].T, np.eye(25), atol=1e-12)


@pytest.mark.gpu
def test_every_entry_of_the_hip_kernel_equals_the_closed_form(gpu):
    """All 1 + 9 + 25 + 49 + 81 entries per rotation of s360_sh_rotation_blocks (what adapter.GaussianAdapterERP applies by default,
    sh_rotation="native") against the closed form, [n,3,3] and [n,4,4] inputs."""
    import torch
    from splatter360_amd import adapter
    R = _test_rotations()
    want = wigner_blocks_closed_form(R, 25)
    got = adapter.sh_rotation_blocks(torch.tensor(R, dtype=torch.float32, device=gpu), 25).double().cpu().numpy()
    np.testing.assert_allclose(got, want, atol=3e-6)
    pose = np.tile(np.eye(4), (R.shape[0], 1, 1))
    pose[:, :3, :3] = R
    pose[:, :3, 3] = 5.0
    got4 = adapter.sh_rotation_blocks(torch.tensor(pose, dtype=torch.float32, device=gpu), 25).double().cpu().numpy()
    np.testing.assert_allclose(got4, want, atol=3e-6)
    off = np.ones((25, 25), bool)
    for l in range(5):
        off[l * l:(l + 1) ** 2, l * l:(l + 1) ** 2] = False
    assert np.abs(got[:, off]).max() == 0.0
