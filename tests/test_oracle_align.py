"""Second opinion on the oracle's two-frame direct alignment (oracle/pose_alignment.hpp, used as the checker of the HIP aligner and to
generate its golden vectors): an independently written NumPy statement of the problem (oracle/spec.py: align_*, from the definition in
src/energy/problems/src/eigen_pose_alignment.cpp:55-206) must give the same energy, the same 8 x 8 normal equations and — taking ONE
Levenberg-Marquardt step itself — the same pose and affine update.  The spec differentiates the geometry by finite differences and
shares no code with the oracle.  (Round-3 review: "nothing independent pins the oracle's pose alignment".)"""
import numpy as np
import pytest

from dsopp_amd import synthetic as syn
from oracle import spec


@pytest.fixture(scope="module")
def case():
    from oracle import pyoracle as po
    win = syn.make_window(num_frames=2, num_points=600, width=320, height=240, seed=21, affine_jitter=True)
    fr, ft = win.frames
    intr = win.scene.intrinsics
    rng = np.random.default_rng(3)
    u, v = fr.uv[:, 0].copy(), fr.uv[:, 1].copy()
    u[:5] = [1.0, 318.0, 150.0, 2.0, 3.0]          # a few points outside the 4-pixel border of the reference image: never valid
    idepth = fr.idepth_gt * (1 + rng.uniform(-0.01, 0.01, len(u)))
    ui, vi = u.astype(int), v.astype(int)
    intensity = fr.pixelinfo[vi, ui, 0].copy()
    intensity[7] += 400.0                           # one gross outlier: exercises the linear branch of the Huber loss
    T_w_r, T_w_t_gt = fr.T_w_c_gt, ft.T_w_c_gt
    T_w_t_init = T_w_t_gt @ syn.se3_exp(np.array([0.01, -0.008, 0.012, 0.003, -0.002, 0.004]))
    return dict(po=po, win=win, fr=fr, ft=ft, intr=np.asarray(intr, dtype=np.float64), u=u, v=v, idepth=idepth, intensity=intensity,
                T_w_r=T_w_r, T_w_t_init=T_w_t_init, e_r=1.0, e_t=1.3, ab_r=np.array([0.02, 3.0]), ab_t=np.array([-0.03, -2.0]))


def _oracle(c, max_iterations, reg, sigma=9.0, radius=1e3):
    po = c["po"]
    opt = po.default_align_options(max_iterations=max_iterations, affine_brightness_regularizer=reg, sigma_huber_loss=sigma,
                                   initial_trust_region_radius=radius)
    h, w = c["fr"].pixelinfo.shape[:2]
    return po.align_solve(opt, c["u"], c["v"], c["idepth"], c["intensity"], c["intr"], (w, h), syn.mat_to_params(c["T_w_r"]), c["e_r"], c["ab_r"],
                          c["intr"], c["ft"].pixelinfo, None, syn.mat_to_params(c["T_w_t_init"]), c["e_t"], c["ab_t"])


def _spec_state(c, T_w_t, ab_t, reg, sigma=9.0):
    h, w = c["fr"].pixelinfo.shape[:2]
    T_tr = np.linalg.inv(T_w_t) @ c["T_w_r"]
    return spec.align_normal_equations(c["ft"].pixelinfo, c["intr"], (w, h), c["intr"], T_tr, c["ab_r"], ab_t, c["e_r"], c["e_t"], c["u"], c["v"],
                                       c["idepth"], c["intensity"], sigma, reg), T_tr


@pytest.mark.parametrize("reg", [(0.0, 0.0), (1e2, 1e-1)])
def test_energy_at_the_initial_state(case, reg):
    """max_iterations = 0: the oracle reports calculateEnergy() of the initial guess"""
    c = case
    ro = _oracle(c, 0, reg)
    (H, g, ok, r), _ = _spec_state(c, c["T_w_t_init"], c["ab_t"], reg)
    e, n = spec.align_energy(r, ok, 9.0, c["ab_t"], reg)
    assert n == ro["n_valid"] and 0 < n < len(c["u"])           # some points are invalid, most are not
    assert (np.abs(r[ok]) > 9.0).sum() >= 1                       # the Huber loss is in its linear branch somewhere
    assert abs(e - ro["energy"]) <= 1e-10 * abs(e)
    assert abs(np.sqrt(e / n) - ro["rmse"]) <= 1e-10 * ro["rmse"]


@pytest.mark.parametrize("reg", [(0.0, 0.0), (1e2, 1e-1)])
def test_normal_equations_and_one_lm_step(case, reg):
    """max_iterations = 1: the oracle's H is that of the initial state, its result the initial state moved by one damped step"""
    c = case
    radius = 1e3
    ro = _oracle(c, 1, reg, radius=radius)
    assert ro["iterations"] == 1
    (H, g, ok, r), T_tr = _spec_state(c, c["T_w_t_init"], c["ab_t"], reg)
    # the oracle (as the reference) builds its system with d_state = -d r / d eps for the pose columns and +d r / d (a, b): a sign
    # pattern S = diag(-1 x 6, +1, +1) on the true Jacobian, which the normal equations carry as S H S
    S = np.diag([-1.0] * 6 + [1.0, 1.0])
    Ho = ro["H"]
    assert np.abs(Ho - S @ H @ S).max() <= 2e-6 * np.abs(H).max(), np.abs(Ho - S @ H @ S).max() / np.abs(H).max()
    # one Levenberg-Marquardt step in the TRUE parameters, taken by the spec: T_tr <- exp(delta_eps) T_tr, (a, b) <- (a, b) + delta_ab
    delta = spec.align_lm_step(H, g, 1.0 / radius)
    T_tr_new = spec.exp_se3(delta[:6]) @ T_tr
    ab_new = c["ab_t"] + delta[6:]
    T_w_t_new = c["T_w_r"] @ np.linalg.inv(T_tr_new)
    e0, _ = spec.align_energy(r, ok, 9.0, c["ab_t"], reg)
    (_, _, ok1, r1), _ = _spec_state(c, T_w_t_new, ab_new, reg)
    e1, n1 = spec.align_energy(r1, ok1, 9.0, ab_new, reg)
    assert e1 < e0                                                # the step is accepted by any LM driver
    To = syn.params_to_mat(ro["T_w_target"])
    assert np.abs(To - T_w_t_new).max() <= 1e-7, np.abs(To - T_w_t_new).max()
    assert np.abs(ro["affine_brightness"] - ab_new).max() <= 1e-6 * max(1.0, np.abs(ab_new).max())
    assert n1 == ro["n_valid"] and abs(e1 - ro["energy"]) <= 1e-6 * e1
    # and the step is not a no-op
    assert np.abs(To - c["T_w_t_init"]).max() > 1e-4


def test_pose_covariance_is_the_pseudo_inverse(case):
    """covariance = pose block of pinv(H) (eigen_pose_alignment.cpp:320-323, completeOrthogonalDecomposition().pseudoInverse())"""
    c = case
    ro = _oracle(c, 1, (1e2, 1e-1))
    P = np.linalg.pinv(ro["H"])
    assert np.abs(ro["covariance"] - P[:6, :6]).max() <= 1e-8 * np.abs(P[:6, :6]).max()
