"""Restates the reference's data-free projector tests on the oracle:
test/test/energy/projector/test_reprojects.cpp:77-206 (reprojection == unproject->transform->project within 1e-3;
analytic d_u/d_v w.r.t. idepth and the LEFT perturbation exp(eps)*T_t_r == autodiff within 5e-3*|x| + 1e-1)."""
import numpy as np

from oracle import pyoracle as po
from oracle import spec

W, H = 1280, 720
INTR = np.array([448.155, 448.155, 640.0, 360.0])  # test/tools/src/solver_test_data.cpp:38-39
INTR2 = np.array([500.0, 480.0, 630.0, 350.0])


def _pattern_inside(rng, T, intr_r, intr_t):
    while True:
        uv = np.array([rng.uniform(8, W - 9), rng.uniform(8, H - 9)])
        idepth = rng.uniform(0.05, 1.0)
        pts, z = spec.project_pattern(intr_r, intr_t, spec.quat_to_mat(T), uv, idepth)
        if np.all(z > 0) and spec.in_roi(pts, W, H) and spec.in_roi(uv[None] + spec.PATTERN, W, H):
            return uv, idepth, pts


def test_pattern_reproject_matches_unproject_transform_project():
    rng = np.random.default_rng(0)
    for k in range(20):
        T = po.se3_exp(np.concatenate([rng.normal(0, 0.2, 3), rng.normal(0, 0.05, 3)]))
        intr_t = INTR if k % 2 == 0 else INTR2
        uv, idepth, pts = _pattern_inside(rng, T, INTR, intr_t)
        u, v = uv[0] + spec.PATTERN[:, 0], uv[1] + spec.PATTERN[:, 1]
        for with_j in (False, True):
            ok, tu, tv, *_ = po.reproject_pattern(INTR, (W, H), intr_t, (W, H), T, u, v, idepth, with_j)
            assert ok
            assert np.abs(tu - pts[:, 0]).max() < 1e-9 and np.abs(tv - pts[:, 1]).max() < 1e-9


def test_pattern_reproject_jacobians_vs_finite_differences():
    rng = np.random.default_rng(1)
    h = 1e-6
    for k in range(10):
        T = po.se3_exp(np.concatenate([rng.normal(0, 0.2, 3), rng.normal(0, 0.05, 3)]))
        M = spec.quat_to_mat(T)
        intr_t = INTR if k % 2 == 0 else INTR2
        uv, idepth, _ = _pattern_inside(rng, T, INTR, intr_t)
        u, v = uv[0] + spec.PATTERN[:, 0], uv[1] + spec.PATTERN[:, 1]
        ok, tu, tv, dui, dvi, duT, dvT = po.reproject_pattern(INTR, (W, H), intr_t, (W, H), T, u, v, idepth, True)
        assert ok
        pp, _ = spec.project_pattern(INTR, intr_t, M, uv, idepth + h)
        pm, _ = spec.project_pattern(INTR, intr_t, M, uv, idepth - h)
        fd = (pp - pm) / (2 * h)
        assert np.allclose(dui, fd[:, 0], rtol=1e-6, atol=1e-6) and np.allclose(dvi, fd[:, 1], rtol=1e-6, atol=1e-6)
        for c in range(6):
            e = np.zeros(6)
            e[c] = h
            pp, _ = spec.project_pattern(INTR, intr_t, spec.exp_se3(e) @ M, uv, idepth)
            pm, _ = spec.project_pattern(INTR, intr_t, spec.exp_se3(-e) @ M, uv, idepth)
            fd = (pp - pm) / (2 * h)
            assert np.allclose(duT[:, c], fd[:, 0], rtol=1e-6, atol=1e-5)
            assert np.allclose(dvT[:, c], fd[:, 1], rtol=1e-6, atol=1e-5)


def test_reproject_failure_modes():
    T = po.se3_exp(np.zeros(6))
    u, v = 100 + spec.PATTERN[:, 0], 100 + spec.PATTERN[:, 1]
    assert po.reproject_pattern(INTR, (W, H), INTR, (W, H), T, u, v, 0.5, False)[0]
    # reference pattern outside the 4-px ROI (camera_model_base.hpp:52-60)
    assert not po.reproject_pattern(INTR, (W, H), INTR, (W, H), T, u - 97, v, 0.5, False)[0]
    # invalid idepth (camera_model_base.hpp:67-74)
    assert not po.reproject_pattern(INTR, (W, H), INTR, (W, H), T, u, v, -1e-3, False)[0]
    assert not po.reproject_pattern(INTR, (W, H), INTR, (W, H), T, u, v, 1011.0, False)[0]
    # point behind the target camera
    Tb = po.se3_exp(np.array([0, 0, -5.0, 0, 0, 0]))
    assert not po.reproject_pattern(INTR, (W, H), INTR, (W, H), Tb, u, v, 1.0, True)[0]
    # ROI upper edge is inclusive at W-5 / H-5
    ue, ve = (W - 5 - 2) + spec.PATTERN[:, 0], (H - 5 - 2) + spec.PATTERN[:, 1]
    assert po.reproject_pattern(INTR, (W, H), INTR, (W, H), T, ue, ve, 0.5, False)[0]
    assert not po.reproject_pattern(INTR, (W, H), INTR, (W, H), T, ue + 0.01, ve, 0.5, False)[0]
