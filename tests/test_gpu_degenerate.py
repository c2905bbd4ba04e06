"""Degenerate reduced systems (round-3 review, weak #8).  The reference solves (H_pp + priors - H_schur) delta = b with Eigen's pivoted
LDL^T (normal_linear_system.cpp:10-16,52-59), which tolerates semi-definite systems; the device solve is an unpivoted blocked Cholesky of
the Jacobi-scaled system with a guard on vanishing pivots (pba_solve_combined.hpp).  These windows have directions without any
information — a keyframe nothing reprojects into, a pair without baseline, six landmarks for sixteen unknowns, a window of fixed frames —
and the device path must reproduce the oracle's LDL^T solve: same iteration count, same valid residuals, energies 1e-7, poses 1e-7
(documented in DESIGN.md: the priors — 1e12 / 1e8 on the affine pair, 1e16 on fixed frames — and the Jacobi scaling keep every pivot of
these systems far above the 1e-30 guard; a frame with NO information at all has zero rows, whose scaled pivot is the guard's business)."""
import numpy as np
import pytest

from dsopp_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _solve_both(win, statuses=None, **opts):
    from dsopp_amd import capi
    from oracle import pyoracle as po
    g = capi.HipWindow(capi.default_pba_options(**opts))
    syn.load_window(g, win, statuses)
    o = po.OracleWindow(po.default_pba_options(**opts))
    syn.load_window(o, win, statuses)
    rg, ro = g.solve(), o.solve()
    return g, o, rg, ro


def _compare(win, g, o, rg, ro, pose_tol=1e-7, energy_rtol=1e-7):
    (eg, itg, nvg), (eo, ito, nvo) = rg, ro
    assert (itg, nvg) == (ito, nvo), (rg, ro)
    assert np.isfinite(eg) and abs(eg - eo) <= energy_rtol * max(abs(eo), 1e-12), (eg, eo)
    for f in win.frames:
        (Tg, abg), (To, abo) = g.get_pose(f.frame_id), o.get_pose(f.frame_id)
        assert np.all(np.isfinite(Tg)) and np.abs(Tg - To).max() <= pose_tol, (f.frame_id, np.abs(Tg - To).max())
        assert np.abs(abg - abo).max() <= 1e-6
        lg, lo = g.get_landmarks(f.frame_id, False), o.get_landmarks(f.frame_id)
        assert np.array_equal(lg["flags"], lo["flags"]), f.frame_id
        assert np.all(np.isfinite(lg["idepth"])) and np.abs(lg["idepth"] - lo["idepth"]).max() <= 1e-7 * max(1.0, np.abs(lo["idepth"]).max())


def test_keyframe_without_any_valid_residual():
    """the last keyframe looks the other way: every residual into and out of it is out of bounds, its pose block of the system is
    empty (only the affine prior is left on its diagonal)"""
    win = syn.make_window(num_frames=4, num_points=240, width=320, height=240, seed=17)
    f = win.frames[-1]
    flip = np.eye(4)
    flip[:3, :3] = np.diag([-1.0, 1.0, -1.0])     # 180 degrees about y
    f.T_w_c_init = f.T_w_c_init @ flip
    g, o, rg, ro = _solve_both(win)
    assert rg[2] > 0
    for h in win.frames[:-1]:
        assert np.all(g.get_residuals(h.frame_id, f.frame_id)["status"] != 0)     # nothing into the flipped frame is OK
    _compare(win, g, o, rg, ro)
    # the frame nothing constrains has not moved
    assert np.abs(g.get_pose(f.frame_id)[0] - syn.mat_to_params(f.T_w_c_init)).max() <= 1e-9
    g.close()


def test_pair_without_baseline():
    """two keyframes at the same place, differing by a rotation: inverse depths are unobservable (H_dd ~ 0 -> ill-conditioned flags or
    huge steps), translation is not"""
    win = syn.make_window(num_frames=2, num_points=160, width=320, height=240, seed=23)
    a, b = win.frames
    R = syn.se3_exp(np.array([0, 0, 0, 0.01, -0.015, 0.005]))
    b.T_w_c_gt = a.T_w_c_gt @ R
    b.T_w_c_init = a.T_w_c_init @ R @ syn.se3_exp(np.array([0, 0, 0, 1e-3, -1e-3, 5e-4]))
    # render what the rotated camera sees (same centre): resample the first image through the rotation so that the photometry is consistent
    img, depth = win.scene.render(b.T_w_c_gt, 0.0, 0.0)
    u8 = np.clip(np.rint(img), 0, 255).astype(np.uint8)
    b.image_u8, b.pixelinfo, b.depth = u8, syn.pixelinfo_from_plane(u8.astype(np.float64)), depth
    ui, vi = b.uv[:, 0].astype(int), b.uv[:, 1].astype(int)
    b.patch = np.stack([u8.astype(np.float64)[vi + int(oy), ui + int(ox)] for ox, oy in syn.PATTERN], axis=1)
    b.idepth_gt = 1.0 / depth[vi, ui]
    b.idepth_init = b.idepth_gt.copy()
    g, o, rg, ro = _solve_both(win)
    _compare(win, g, o, rg, ro, pose_tol=1e-6, energy_rtol=1e-6)
    g.close()


def test_six_landmarks_for_two_frames():
    win = syn.make_window(num_frames=2, num_points=6, width=320, height=240, seed=3)
    g, o, rg, ro = _solve_both(win)
    _compare(win, g, o, rg, ro, pose_tol=1e-6, energy_rtol=1e-6)
    g.close()


def test_every_frame_fixed():
    win = syn.make_window(num_frames=3, num_points=150, width=320, height=240, seed=31)
    for f in win.frames:
        f.fixed = True
    g, o, rg, ro = _solve_both(win)
    _compare(win, g, o, rg, ro)
    for f in win.frames:      # 1e16 on every diagonal: the poses stay where they were (to the regulariser's 1e-16 leverage)
        assert np.abs(g.get_pose(f.frame_id)[0] - syn.mat_to_params(f.T_w_c_init)).max() <= 1e-9
    g.close()


def test_all_residuals_out_of_bounds():
    """no residual at all: the LM loop ends with zero valid residuals on both sides, nothing is NaN"""
    win = syn.make_window(num_frames=3, num_points=90, width=320, height=240, seed=37)
    flip = np.eye(4)
    flip[:3, :3] = np.diag([-1.0, 1.0, -1.0])
    for k, f in enumerate(win.frames):
        if k % 2:
            f.T_w_c_init = f.T_w_c_init @ flip
    win.frames[2].T_w_c_init = win.frames[2].T_w_c_init @ syn.se3_exp(np.array([50.0, 0, 0, 0, 0, 0]))
    g, o, rg, ro = _solve_both(win)
    assert rg[2] == ro[2] == 0 and rg[1] == ro[1]
    for f in win.frames:
        assert np.all(np.isfinite(g.get_pose(f.frame_id)[0]))
        assert np.abs(g.get_pose(f.frame_id)[0] - o.get_pose(f.frame_id)[0]).max() <= 1e-9
    g.close()
