"""initializationPoses of the tracker (src/tracker/tracker/src/monocular_tracker.cpp:136-176) and the SE3 logarithm behind its
"half motion" hypothesis.  The oracle restatement against independent statements (scipy's matrix logarithm / exponential on
4x4 matrices), then the library's host implementation against the oracle.  No device work is involved."""
import numpy as np
import pytest
from scipy.linalg import expm, logm

from dsopp_amd import synthetic as syn


def _twist_matrix(xi):
    wx, wy, wz = xi[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -wz, wy], [wz, 0, -wx], [-wy, wx, 0]]
    M[:3, 3] = xi[:3]
    return M


def _random_pose(rng, angle):
    xi = np.concatenate([rng.normal(0, 0.5, 3), rng.normal(0, 1, 3)])
    xi[3:] *= angle / np.linalg.norm(xi[3:])
    return expm(_twist_matrix(xi)), xi


def test_se3_log_matches_matrix_logarithm():
    from oracle import pyoracle as po
    rng = np.random.default_rng(3)
    for angle in (1e-12, 1e-6, 1e-3, 0.3, 1.5, 3.0):
        for _ in range(5):
            T, xi = _random_pose(rng, angle)
            got = po.se3_log(syn.mat_to_params(T))
            L = np.real(logm(T))
            want = np.array([L[0, 3], L[1, 3], L[2, 3], L[2, 1], L[0, 2], L[1, 0]])
            assert np.abs(got - want).max() <= 1e-9, (angle, got, want)
            assert np.abs(got - xi).max() <= 1e-9


def _expected(Tp, Tl, Tk):
    rel = np.linalg.inv(Tp) @ Tl
    out = [Tl @ rel, Tl @ rel @ rel, Tl @ expm(0.5 * np.real(logm(rel))), Tl, Tk]
    for deg in (1.0, 1.5, 2.0, 2.5):
        d = np.deg2rad(deg)
        for rx in (0, d, -d):
            for ry in (0, d, -d):
                for rz in (0, d, -d):
                    out.append(out[0] @ expm(_twist_matrix(np.array([0, 0, 0, rx, ry, rz]))))
    return out


@pytest.fixture(scope="module")
def poses():
    rng = np.random.default_rng(5)
    Tp, _ = _random_pose(rng, 0.4)
    step = expm(_twist_matrix(np.array([0.08, 0.01, 0.02, 0.004, 0.012, 0.003])))
    return Tp, Tp @ step, _random_pose(rng, 0.2)[0]


def test_oracle_initialization_poses(poses):
    from oracle import pyoracle as po
    Tp, Tl, Tk = poses
    got = po.initialization_poses(syn.mat_to_params(Tp), syn.mat_to_params(Tl), syn.mat_to_params(Tk))
    want = _expected(Tp, Tl, Tk)
    assert len(got) == len(want) == 113
    for k, (g, w) in enumerate(zip(got, want)):
        assert np.abs(syn.params_to_mat(g) - w).max() <= 1e-10, k
    single = po.initialization_poses(None, None, None)
    assert single.shape == (1, 7) and np.array_equal(single[0], [0, 0, 0, 1, 0, 0, 0])


def test_library_initialization_poses_match_oracle(poses):
    from dsopp_amd import capi
    from oracle import pyoracle as po
    Tp, Tl, Tk = (syn.mat_to_params(T) for T in poses)
    got, want = capi.initialization_poses(Tp, Tl, Tk), po.initialization_poses(Tp, Tl, Tk)
    assert got.shape == want.shape == (113, 7)
    for g, w in zip(got, want):
        if np.dot(g[:4], w[:4]) < 0:
            g = np.concatenate([-g[:4], g[4:]])     # q and -q are the same rotation
        assert np.abs(g - w).max() <= 1e-12
    assert capi.initialization_poses(None, None, None).shape == (1, 7)
