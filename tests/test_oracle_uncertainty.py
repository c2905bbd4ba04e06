"""Second opinion on the oracle's covariance path (row a17): the pseudo-inverse with dropped directions
(eigen_photometric_bundle_adjustment.cpp:31-45) against NumPy's eigendecomposition and numpy.linalg.pinv with the matching cut-off, and
the covariance of a relative pose (se3_motion.hpp:140-158) against a finite-difference propagation through exp / log written
independently in oracle/spec.py."""
import numpy as np
import pytest

from dsopp_amd import synthetic as syn
from oracle import spec


def _gauge_deficient_system(n, seed, null_scale):
    """symmetric positive semi-definite K x K matrix with one direction `null_scale` times weaker than the rest — the shape of a
    monocular window's reduced system, whose scale gauge is a numerical null space"""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(n, n)))
    w = np.exp(rng.uniform(np.log(1e2), np.log(1e7), n))
    w[0] = w.min() * null_scale
    return (Q * w[None, :]) @ Q.T, Q, w


@pytest.mark.parametrize("n,null_scale", [(16, 1e-9), (56, 1e-7), (96, 1e-6)])
def test_pinv_with_one_dropped_direction(n, null_scale):
    from oracle import pyoracle as po
    H, Q, w = _gauge_deficient_system(n, n, null_scale)
    P = po.pinv_drop(H, 1)
    ref = spec.pinv_drop_smallest(H, 1)
    scale = np.abs(ref).max()
    assert np.abs(P - ref).max() <= 1e-8 * scale
    # numpy.linalg.pinv with a cut-off between the dropped and the smallest kept singular value says the same
    srt = np.sort(w)
    rcond = np.sqrt(srt[0] * srt[1]) / srt[-1]
    assert np.abs(P - np.linalg.pinv(H, rcond=rcond, hermitian=True)).max() <= 1e-8 * scale
    # Moore-Penrose identities on the kept subspace, and the dropped direction is really gone
    assert np.abs(P @ H @ P - P).max() <= 1e-7 * scale
    assert np.abs(P @ Q[:, 0]).max() <= 1e-9 * scale
    # nothing dropped: the plain inverse
    H0, _, _ = _gauge_deficient_system(n, n + 1, 1.0)
    assert np.abs(po.pinv_drop(H0, 0) @ H0 - np.eye(n)).max() <= 1e-7


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_relative_pose_covariance(seed):
    from oracle import pyoracle as po
    rng = np.random.default_rng(seed)
    T1 = syn.se3_exp(rng.normal(0, 0.5, 6))
    T2 = syn.se3_exp(rng.normal(0, 0.5, 6))
    A = rng.normal(size=(12, 12))
    S = A @ A.T * 1e-4                                   # joint covariance of (eps_1, eps_2)
    s11, s22, s12 = S[:6, :6], S[6:, 6:], S[:6, 6:]
    got = po.relative_transformation_uncertainty(syn.mat_to_params(T1), syn.mat_to_params(T2), s11, s22, s12)
    want = spec.relative_pose_covariance_fd(T1, T2, s11, s22, s12)
    assert np.abs(got - want).max() <= 1e-7 * np.abs(want).max(), np.abs(got - want).max() / np.abs(want).max()
    assert np.abs(got - got.T).max() <= 1e-12 * np.abs(got).max()
    # uncorrelated, identical poses: the two covariances add
    same = po.relative_transformation_uncertainty(syn.mat_to_params(T1), syn.mat_to_params(T1), s11, s22, np.zeros((6, 6)))
    assert np.abs(same - (s11 + s22)).max() <= 1e-12 * np.abs(s11 + s22).max()


def test_window_covariances_follow_from_the_reduced_system():
    """the covariances a solved window reports == pinv (one dropped direction) of its reduced system, pushed through the relative-pose
    propagation of the spec — the whole a17 chain with only the system itself taken from the oracle"""
    from oracle import pyoracle as po
    win = syn.make_window(num_frames=4, num_points=320, width=320, height=240, seed=9)
    # (the covariance pass builds its system without the Huber weights — evaluateJacobians<..., HUBER = false>, covariance_matrix... at
    # eigen_photometric_bundle_adjustment.cpp:88-97 — so the loss is switched off for the whole window: then the stage API's linearisation
    # of the solved state is that system, and the 3rd-quartile rule behind the covariance pass flags nothing)
    o = po.OracleWindow(po.default_pba_options(first_estimate_jacobians=False, sigma_huber_loss=1e9))
    syn.load_window(o, win)
    o.solve()
    # the system of the closing linearisation (at the solved state): H_pp (with priors) - H_schur + marginal prior
    o.begin()
    o.calculate_energy()
    o.linearize()
    Hpp, _, Hsc, _ = o.get_system()
    Hm, _, _ = o.get_marginalized()
    cov = spec.pinv_drop_smallest(Hpp - Hsc + Hm, 1)
    ids = [f.frame_id for f in win.frames]
    poses = [syn.params_to_mat(o.get_pose(i)[0]) for i in ids]
    for a in range(len(ids)):
        for b in range(len(ids)):
            if a == b:
                continue
            blk = lambda i, j: cov[8 * i:8 * i + 6, 8 * j:8 * j + 6]
            want = spec.relative_pose_covariance_fd(poses[a], poses[b], blk(a, a), blk(b, b), blk(a, b))
            got = o.get_covariance(ids[a], ids[b])
            assert np.abs(got - want).max() <= 1e-4 * np.abs(want).max(), (a, b, np.abs(got - want).max() / np.abs(want).max())
