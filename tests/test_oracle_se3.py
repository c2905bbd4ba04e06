"""Pins the SE3 restatement (oracle/se3.hpp) with closed-form identities — the reference's Sophus dependency is not in
/root/reference, so exp/Adj values are 'parity unpinned' by reference data and pinned here instead."""
import numpy as np
import pytest

from oracle import pyoracle as po
from oracle import spec


def rand_xi(rng, scale=0.5):
    return rng.normal(0, scale, 6)


def test_exp_zero_is_identity():
    T = po.se3_exp(np.zeros(6))
    assert np.allclose(T, [0, 0, 0, 1, 0, 0, 0], atol=0)


def test_exp_matches_matrix_exponential():
    rng = np.random.default_rng(0)
    for scale in (1e-12, 1e-6, 1e-2, 0.5, 2.0):
        for _ in range(5):
            xi = rand_xi(rng, scale)
            M = spec.quat_to_mat(po.se3_exp(xi))
            assert np.allclose(M, spec.exp_se3(xi), atol=1e-12, rtol=1e-12)


def test_inverse_and_product():
    rng = np.random.default_rng(1)
    for _ in range(10):
        A, B = po.se3_exp(rand_xi(rng)), po.se3_exp(rand_xi(rng))
        MA, MB = spec.quat_to_mat(A), spec.quat_to_mat(B)
        assert np.allclose(spec.quat_to_mat(po.se3_mul(A, B)), MA @ MB, atol=1e-13)
        assert np.allclose(spec.quat_to_mat(po.se3_inverse(A)), np.linalg.inv(MA), atol=1e-13)
        a = rand_xi(rng)
        assert np.allclose(spec.quat_to_mat(po.se3_mul(po.se3_exp(a), po.se3_exp(-a))), np.eye(4), atol=1e-13)


def test_adjoint_identity():
    """T exp(xi) T^-1 == exp(Adj(T) xi)  (rightLogTransformer, se3_motion.hpp:245)."""
    rng = np.random.default_rng(2)
    for _ in range(10):
        T = po.se3_exp(rand_xi(rng))
        M = spec.quat_to_mat(T)
        Adj = po.se3_adj(T)
        assert np.allclose(Adj, spec.adjoint(M), atol=1e-13)
        xi = rand_xi(rng, 0.3)
        lhs = M @ spec.exp_se3(xi) @ np.linalg.inv(M)
        assert np.allclose(lhs, spec.exp_se3(Adj @ xi), atol=1e-11)


def test_small_angle_series_continuity():
    base = np.array([0.3, -0.2, 0.1, 1.0, 2.0, -1.0])
    for th in (1e-11, 0.9e-10, 1.1e-10, 1e-9, 1e-8):
        xi = base.copy()
        xi[3:] = base[3:] / np.linalg.norm(base[3:]) * th
        assert np.allclose(spec.quat_to_mat(po.se3_exp(xi)), spec.exp_se3(xi), atol=1e-14)
