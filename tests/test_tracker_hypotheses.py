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


# ---------------------------------------------------------------------------------------------------------------------
# estimate_pose with several initialisations per launch (dsopp_hip_aligner_set_hypothesis_width): one XCD per initialisation, the
# lowest-index success wins — the sequential loop of monocular_tracker.cpp:193-243 in one launch
# ---------------------------------------------------------------------------------------------------------------------
def _hypothesis_case(width, height, levels, seed):
    """a keyframe with device-resident reference depth maps and two later frames: the first is tracked normally (it leaves
    rmse_last_pose_estimation at its tracking values, which is what makes bad initialisations FAIL their per-level gates), the second
    gets a list of initialisations of which only a late one is near the truth"""
    import numpy as np
    from dsopp_amd import capi, synthetic as syn
    scene = syn.Scene.make(width, height, seed)
    intr = scene.intrinsics
    poses = [syn.se3_exp(0.3 * k * syn.BASE_MOTION) for k in range(3)]
    imgs = []
    for T in poses:
        img, depth = scene.render(T)
        imgs.append((np.clip(np.rint(img), 0, 255).astype(np.uint8), depth))
    pyr = []
    for u8, _ in imgs:
        p = capi.Pyramid(width, height, levels)
        p.build(u8)
        pyr.append(p)
    win = syn.make_window(num_frames=2, num_points=900, width=width, height=height, seed=seed)   # only to get landmarks for the maps
    g = capi.HipWindow(capi.default_pba_options())
    rng = np.random.default_rng(seed)
    # two keyframes at the first two poses with landmarks at their true depths -> reference depth maps of the newer one
    for k in (0, 1):
        u8, depth = imgs[k]
        info = np.zeros(u8.shape + (3,))
        info[..., 0] = u8
        uv = np.stack([rng.integers(8, width - 8, 900), rng.integers(8, height - 8, 900)], axis=1).astype(np.float64)
        ui, vi = uv[:, 0].astype(int), uv[:, 1].astype(int)
        patch = np.stack([u8.astype(np.float64)[vi + int(oy), ui + int(ox)] for ox, oy in syn.PATTERN], axis=1)
        g.push_frame(k, 1000 * (k + 1), None, None, intr, syn.mat_to_params(poses[k]), 1.0, np.zeros(2), k == 0, False, pyramid=pyr[k])
        g.set_landmarks(k, uv, 1.0 / depth[vi, ui], patch, np.zeros(900, dtype=np.uint8))
    g.set_connection(0, 1, np.zeros(900, dtype=np.uint8))
    g.set_connection(1, 0, np.zeros(900, dtype=np.uint8))
    g.solve()
    maps = g.create_reference_depth_maps(levels)
    T_ref, ab_ref = g.get_pose(1)
    return dict(capi=capi, syn=syn, intr=intr, poses=poses, pyr=pyr, g=g, maps=maps, T_ref=T_ref, ab_ref=ab_ref, levels=levels, rng=rng)


@pytest.mark.gpu
@pytest.mark.parametrize("n_bad", [7, 20])
def test_concurrent_hypotheses_equal_the_sequential_loop(n_bad):
    import time
    import numpy as np
    c = _hypothesis_case(320, 240, 3, 5)
    capi, syn = c["capi"], c["syn"]
    good = syn.mat_to_params(c["poses"][2] @ syn.se3_exp(np.array([2e-3, -1e-3, 1e-3, 5e-4, -5e-4, 2e-4])))
    # far outside the basin of the coarse-to-fine alignment (it recovers from 0.25 m / 7 degrees on this scene): 25 .. 45 degrees off
    def far():
        ax = c["rng"].normal(size=3)
        return np.concatenate([c["rng"].normal(0, 1.0, 3), ax / np.linalg.norm(ax) * c["rng"].uniform(0.45, 0.8)])
    bad = [syn.mat_to_params(c["poses"][2] @ syn.se3_exp(far())) for _ in range(n_bad)]
    hyp = np.stack(bad + [good])
    results, times = {}, {}
    for width in (1, 8):
        a = capi.HipAligner(capi.default_align_options())
        a.set_hypothesis_width(width)
        rmse_last = np.full(c["levels"], 1e10)
        # a normally tracked frame first: sets the per-level energy gates (and arms the exchange buffers)
        r0 = a.estimate_pose(2000, c["T_ref"], c["pyr"][1], c["maps"], 1.0, c["ab_ref"], 2500, c["pyr"][2], 1.0, c["intr"], good[None, :], np.zeros(2), rmse_last)
        assert r0["success"] and r0["tries"] == 1
        gates = rmse_last.copy()
        best = 1e9
        for rep in range(5):
            rl = gates.copy()
            t0 = time.perf_counter()
            r = a.estimate_pose(2000, c["T_ref"], c["pyr"][1], c["maps"], 1.0, c["ab_ref"], 3000, c["pyr"][2], 1.0, c["intr"], hyp, np.zeros(2), rl)
            best = min(best, time.perf_counter() - t0)
        results[width], times[width] = (r, rl), best
        # single-initialisation time of the same aligner, for the ratio below
        t0 = time.perf_counter()
        a.estimate_pose(2000, c["T_ref"], c["pyr"][1], c["maps"], 1.0, c["ab_ref"], 3000, c["pyr"][2], 1.0, c["intr"], good[None, :], np.zeros(2), gates.copy())
        times[(width, "single")] = time.perf_counter() - t0
        a.close()
    (rs, rls), (rp, rlp) = results[1], results[8]
    assert rs["success"] and rs["tries"] == n_bad + 1            # every bad initialisation failed a gate, the last one passed
    assert rp["success"] and rp["tries"] == rs["tries"] and rp["lm_iterations"] == rs["lm_iterations"]
    assert np.array_equal(rp["T_w_target"], rs["T_w_target"]) and np.array_equal(rp["affine_brightness"], rs["affine_brightness"])
    assert np.array_equal(rls, rlp)
    # 8 initialisations in one launch cost about what one costs; the sequential loop pays for each (n_bad = 7: the review's bar is 1.3 x)
    launches = (n_bad + 1 + 7) // 8
    assert times[8] <= 1.6 * launches * times[(8, "single")] + 2e-4, (times, launches)
    assert times[8] < 0.5 * times[1], times
    print(f"hypotheses: {n_bad + 1} tried; sequential {times[1] * 1e3:.3f} ms, 8 per launch {times[8] * 1e3:.3f} ms, single initialisation {times[(8, 'single')] * 1e3:.3f} ms")
    for p in c["pyr"]:
        p.close()
    c["g"].close()


@pytest.mark.gpu
def test_concurrent_hypotheses_all_fail_keeps_the_first():
    """every initialisation fails: the result of the FIRST one is kept and the gates are relaxed by 2.5 (monocular_tracker.cpp:229-238)
    — whether the tries ran one by one or eight at a time"""
    import numpy as np
    c = _hypothesis_case(320, 240, 3, 6)
    capi, syn = c["capi"], c["syn"]
    good = syn.mat_to_params(c["poses"][2])
    def far():
        ax = c["rng"].normal(size=3)
        return np.concatenate([c["rng"].normal(0, 1.0, 3), ax / np.linalg.norm(ax) * c["rng"].uniform(0.45, 0.8)])
    hyp = np.stack([syn.mat_to_params(c["poses"][2] @ syn.se3_exp(far())) for _ in range(11)])
    out = {}
    for width in (1, 8, 0):
        a = capi.HipAligner(capi.default_align_options())
        a.set_hypothesis_width(width)
        rmse_last = np.full(c["levels"], 1e10)
        a.estimate_pose(2000, c["T_ref"], c["pyr"][1], c["maps"], 1.0, c["ab_ref"], 2500, c["pyr"][2], 1.0, c["intr"], good[None, :], np.zeros(2), rmse_last)
        gates = rmse_last.copy()
        r = a.estimate_pose(2000, c["T_ref"], c["pyr"][1], c["maps"], 1.0, c["ab_ref"], 3000, c["pyr"][2], 1.0, c["intr"], hyp, np.zeros(2), rmse_last)
        assert not r["success"] and r["tries"] == 11
        assert np.allclose(rmse_last, 2.5 * gates, rtol=0, atol=0)
        out[width] = r
        a.close()
    for width in (8, 0):
        assert np.array_equal(out[width]["T_w_target"], out[1]["T_w_target"]) and out[width]["lm_iterations"] == out[1]["lm_iterations"]
    for p in c["pyr"]:
        p.close()
    c["g"].close()


@pytest.mark.gpu
def test_persistent_tracker_kernel_is_bitwise_reproducible():
    """estimatePose on bitwise identical inputs has to return the bitwise identical pose, per-level rmse and iteration count call after call —
    with the participants on one XCD (plain stores in that XCD's L2) and dealt over all XCDs (agent-scope stores).  A stale or early read
    anywhere in the persistent kernel (exchange buffers, LDS hand-overs, control block) shows up here as a second outcome: this is the
    test that caught a write race in the level set-up of the round-6 kernel (scripts/stress_tracker.py is the long form)."""
    import collections
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for spread in ("8", "1"):   # (read once per process by the library: one process per setting)
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "stress_tracker.py"), "250"], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, DSOPP_HIP_ALIGN_SPREAD=spread))
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [ln for ln in r.stdout.splitlines() if "distinct outcome" in ln]
        assert len(lines) == 3, r.stdout
        for ln in lines:
            assert " 1 distinct outcome(s)" in ln, ln
