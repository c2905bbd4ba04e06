"""Row f-1: the depth estimator of immature landmarks (src/tracker/depth_estimators/src/depth_estimation.cpp,
src/energy/epipolar_geometry/*).  CPU: the oracle restatement against statements that do not share its code — every
epipolar-line point must reproject (explicit NumPy 4x4 chain) onto itself at its triangulated inverse depth, the line must
pass through the true correspondence, and on a synthetic scene the estimated [idepth_min, idepth_max] intervals must bracket
the ground truth and shrink with a second observation.  GPU: the HIP estimator against the oracle."""
import numpy as np
import pytest

from dsopp_amd import synthetic as syn


def _rel(T_w_t, T_w_r):
    return np.linalg.inv(T_w_t) @ T_w_r


def _mat_to_params(T):
    return syn.mat_to_params(T)


def _reproject(intr, T, uv, rho):
    fx, fy, cx, cy = intr
    d = np.array([(uv[0] - cx) / fx, (uv[1] - cy) / fy, 1.0])
    X = T[:3, :3] @ d + rho * T[:3, 3]
    return np.array([fx * X[0] / X[2] + cx, fy * X[1] / X[2] + cy]), X[2]


def _landmarks(po, frame, n=None):
    uv = frame.uv if n is None else frame.uv[:n]
    ui, vi = uv[:, 0].astype(int), uv[:, 1].astype(int)
    grad = np.stack([frame.pixelinfo[vi, ui, 1], frame.pixelinfo[vi, ui, 2]], axis=1)
    return uv, grad


@pytest.fixture(scope="module")
def scene():
    return syn.make_window(num_frames=4, num_points=4 * 150, width=320, height=240, seed=71, pose_noise=False)


def test_epipolar_segment_is_consistent(scene):
    from oracle import pyoracle as po
    win = scene
    intr = win.scene.intrinsics
    fr, ft = win.frames[0], win.frames[2]
    T = _rel(ft.T_w_c_gt, fr.T_w_c_gt)
    checked = 0
    for i in range(0, 60, 3):
        uv = fr.uv[i]
        proj, idp = po.build_epipolar_segment(320, 240, intr, _mat_to_params(T), uv)
        if len(proj) < 3:
            continue
        # (a) every point is the reprojection of the observed pixel at the point's own inverse depth
        for p, rho in zip(proj[::7], idp[::7]):
            q, z = _reproject(intr, T, uv, rho)
            assert z > 0 and np.abs(q - p).max() <= 1e-6, (i, p, q)
        # (b) idepth is monotone along the line and inside the search range
        assert np.all(np.diff(idp) >= -1e-9) or np.all(np.diff(idp) <= 1e-9)
        assert idp.min() >= -1e-9 and idp.max() <= 1000 + 1e-9
        # (c) the true correspondence lies on the polyline (within the 1-px sampling)
        q, _ = _reproject(intr, T, uv, fr.idepth_gt[i])
        if 4 <= q[0] <= 315 and 4 <= q[1] <= 235:
            assert np.min(np.hypot(*(proj - q).T)) <= 1.0, i
            checked += 1
    assert checked >= 10


def test_estimated_intervals_bracket_ground_truth(scene):
    from oracle import pyoracle as po
    win = scene
    intr = win.scene.intrinsics
    fr = win.frames[0]
    uv, grad = _landmarks(po, fr)
    direction = np.stack([(uv[:, 0] - intr[2]) / intr[0], (uv[:, 1] - intr[3]) / intr[1], np.ones(len(uv))], axis=1)
    lms = po.new_immature_landmarks(uv, direction, fr.patch, grad)
    widths = []
    for ft in (win.frames[1], win.frames[3]):
        T = _rel(ft.T_w_c_gt, fr.T_w_c_gt)
        po.estimate_depths(lms, ft.pixelinfo, None, intr, _mat_to_params(T))
        good = lms["status"] == po.IMMATURE_STATUS["good"]
        assert good.mean() > 0.5
        inside = (lms["idepth_min"][good] <= fr.idepth_gt[good] * 1.02) & (fr.idepth_gt[good] * 0.98 <= lms["idepth_max"][good])
        assert inside.mean() > 0.85, inside.mean()
        widths.append(np.median(lms["idepth_max"][good] - lms["idepth_min"][good]))
        assert np.all(lms["traced"][good] == 1)
    assert widths[1] < widths[0]   # a second, wider-baseline observation narrows the interval


@pytest.mark.gpu
def test_gpu_depth_estimation_matches_oracle(scene):
    """two consecutive observations (the second one takes the traced path with the narrowed interval): statuses and the
    traced flags identical, intervals / uniqueness / pixel intervals to 1e-8 (the device evaluates point i of the epipolar
    segment as start + i * step where the reference accumulates the step i times)"""
    import copy
    from dsopp_amd import capi
    from oracle import pyoracle as po
    win = scene
    intr = win.scene.intrinsics
    fr = win.frames[0]
    uv, grad = _landmarks(po, fr)
    direction = np.stack([(uv[:, 0] - intr[2]) / intr[0], (uv[:, 1] - intr[3]) / intr[1], np.ones(len(uv))], axis=1)
    lo = po.new_immature_landmarks(uv, direction, fr.patch, grad)
    lg = copy.deepcopy(lo)
    for step, ft in enumerate((win.frames[1], win.frames[3], win.frames[2])):
        T = _mat_to_params(_rel(ft.T_w_c_gt, fr.T_w_c_gt))
        pyr = capi.Pyramid(320, 240, 1)
        pyr.set_level(0, ft.pixelinfo)
        ab_r, ab_t = (0.01 * step, 0.5 * step), (-0.02 * step, 0.3 * step)
        po.estimate_depths(lo, ft.pixelinfo, None, intr, T, 1.0, ab_r, 1.0 + 0.1 * step, ab_t)
        capi.estimate_depths(lg, pyr, 0, intr, T, 1.0, ab_r, 1.0 + 0.1 * step, ab_t)
        pyr.close()
        assert np.array_equal(lo["status"], lg["status"]), (step, np.flatnonzero(lo["status"] != lg["status"]))
        assert np.array_equal(lo["traced"], lg["traced"]), step
        good = lo["status"] == po.IMMATURE_STATUS["good"]
        if step < 2:   # (the third observation has a short baseline: most landmarks end up skipped / ill conditioned)
            assert good.sum() > 50, step
        for k in ("idepth_min", "idepth_max", "search_pixel_interval"):
            assert np.abs(lo[k] - lg[k]).max() <= 1e-8 * max(1.0, np.abs(lo[k]).max()), (step, k)
        fin = lo["uniqueness"] < 1e300
        assert np.array_equal(fin, lg["uniqueness"] < 1e300)
        assert np.abs(lo["uniqueness"][fin] - lg["uniqueness"][fin]).max() <= 1e-7 * np.abs(lo["uniqueness"][fin]).max(), step
    # every status class that matters was seen along the way
    assert len(set(lo["status"].tolist())) >= 3
    # the device-resident set driven over the same three frames ends in the same state as the one-shot calls
    dset = capi.ImmatureSet(po.new_immature_landmarks(uv, direction, fr.patch, grad))
    for step, ft in enumerate((win.frames[1], win.frames[3], win.frames[2])):
        T = _mat_to_params(_rel(ft.T_w_c_gt, fr.T_w_c_gt))
        pyr = capi.Pyramid(320, 240, 1)
        pyr.set_level(0, ft.pixelinfo)
        dset.estimate(pyr, 0, intr, T, 1.0, (0.01 * step, 0.5 * step), 1.0 + 0.1 * step, (-0.02 * step, 0.3 * step))
        st = dset.download()   # (also orders the kernel before the pyramid goes away)
        pyr.close()
    for k in ("idepth_min", "idepth_max", "uniqueness", "search_pixel_interval", "status", "traced"):
        assert np.array_equal(st[k], lg[k]), k
    dset.close()


@pytest.mark.gpu
def test_gpu_batched_estimate_equals_per_set_calls(scene):
    """dsopp_hip_immature_sets_estimate (all keyframes' sets against the new frame in one launch) leaves every set in exactly
    the state the per-set calls leave it in; sets of different sizes, shared and separate streams"""
    import ctypes
    from dsopp_amd import capi
    from oracle import pyoracle as po
    win = scene
    intr = win.scene.intrinsics
    new = win.frames[3]
    pyr = capi.Pyramid(320, 240, 1)
    pyr.set_level(0, new.pixelinfo)
    lms, Ts = [], []
    for i, n in zip(range(3), (150, 97, 31)):
        f = win.frames[i]
        uv, grad = _landmarks(po, f, n)
        direction = np.stack([(uv[:, 0] - intr[2]) / intr[0], (uv[:, 1] - intr[3]) / intr[1], np.ones(len(uv))], axis=1)
        lms.append(po.new_immature_landmarks(uv, direction, f.patch[:n], grad))
        Ts.append(_mat_to_params(_rel(new.T_w_c_gt, f.T_w_c_gt)))
    expo, aff = np.array([1.0, 1.1, 0.9]), np.array([[0.0, 0.0], [0.01, 0.5], [-0.02, -0.3]])
    single = [capi.ImmatureSet(l) for l in lms]
    for s, T, e, a in zip(single, Ts, expo, aff):
        s.estimate(pyr, 0, intr, T, e, a, 1.05, (0.01, 0.2))
    want = [s.download() for s in single]
    hip = ctypes.CDLL("libamdhip64.so")
    shared = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(shared)) == 0
    for streams in (None, shared.value):
        batch = [capi.ImmatureSet(l, stream=streams) for l in lms]
        capi.estimate_depths_batched(batch, pyr, 0, intr, np.stack(Ts), expo, aff, 1.05, (0.01, 0.2))
        for b, w in zip(batch, want):
            got = b.download()
            for k in w:
                assert np.array_equal(got[k], w[k]), k
            b.close()
    assert sum((w["status"] == 0).sum() for w in want) > 100
    for o in single + [pyr]:
        o.close()
    hip.hipStreamDestroy(shared)
