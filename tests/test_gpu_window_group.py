"""dsopp_hip_window_group: the single-process multi-device form of the drop-in (one solver object in one process, as the reference's
factory builds it — src/tracker/tracker/src/fabric.cpp:58-121, src/application/dsopp_main.cpp:114-119).

A single-GPU box cannot hold one shard per device, so the group's shards all sit on device 0 and the in-process reducer
(DSOPP_HIP_TRANSPORT_LOCAL: event-ordered sum kernel) stands in for RCCL; everything else — worker threads, round-robin landmark
deal, sharded solve, interleaved read-backs — is the code a multi-device group runs.  The bar: a group solve reproduces the
single-window solve to 1e-7 (and both agree with the CPU oracle), stage by stage and over a sliding-window sequence with
marginalisation; in deterministic mode two group solves are bit-identical."""
import numpy as np
import pytest

from dsopp_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _close(a, b, rtol, atol):
    return np.abs(np.asarray(a) - np.asarray(b)).max() <= atol + rtol * np.abs(np.asarray(b)).max()


def _single_and_group(win, shards, **opts):
    from dsopp_amd import capi
    g1 = capi.HipWindow(capi.default_pba_options(**opts))
    syn.load_window(g1, win)
    gg = capi.HipWindowGroup(capi.default_pba_options(**opts), devices=[0] * shards, transport=capi.TRANSPORT_LOCAL)
    syn.load_window(gg, win)
    return g1, gg


def _compare_state(win, g1, gg, pose_tol=1e-7):
    for f in win.frames:
        (T1, ab1), (T2, ab2) = g1.get_pose(f.frame_id), gg.get_pose(f.frame_id)
        assert np.abs(T1 - T2).max() <= pose_tol and np.abs(ab1 - ab2).max() <= pose_tol, f.frame_id
        l1, l2 = g1.get_landmarks(f.frame_id, False), gg.get_landmarks(f.frame_id, False)
        assert _close(l2["idepth"], l1["idepth"], 1e-7, 1e-12), f.frame_id
        assert np.array_equal(l1["flags"], l2["flags"]) and np.array_equal(l1["n_inliers"], l2["n_inliers"]), f.frame_id
        for h in win.frames:
            if h.frame_id == f.frame_id:
                continue
            r1, r2 = g1.get_residuals(f.frame_id, h.frame_id), gg.get_residuals(f.frame_id, h.frame_id)
            assert np.array_equal(r1["status"], r2["status"]), (f.frame_id, h.frame_id)


@pytest.mark.parametrize("shards", [2, 3])
@pytest.mark.parametrize("lm_mode", [0, 1])
def test_group_solve_matches_single_window_and_oracle(shards, lm_mode):
    from oracle import pyoracle as po
    win = syn.make_window(num_frames=4, num_points=403, width=320, height=240, seed=5)  # 403: ragged shards
    g1, gg = _single_and_group(win, shards)
    assert gg.size == shards
    for f in win.frames:
        # round-robin deal: balanced to +-1, landmark j on shard j % n
        counts = [gg.shard_num_landmarks(s, f.frame_id) for s in range(shards)]
        assert sum(counts) == len(f.uv) == gg.num_landmarks(f.frame_id) and max(counts) - min(counts) <= 1
    g1.set_lm_mode(lm_mode)
    gg.set_lm_mode(lm_mode)
    e1, it1, nv1 = g1.solve()
    e2, it2, nv2 = gg.solve()
    assert (it1, nv1) == (it2, nv2) and abs(e1 - e2) <= 1e-7 * abs(e1)
    _compare_state(win, g1, gg)
    # covariances of the relative poses (pinv of the reduced system, identical on every shard)
    a, b = win.frames[1].frame_id, win.frames[2].frame_id
    assert _close(gg.get_covariance(a, b), g1.get_covariance(a, b), 1e-5, 0)
    # updateFrame read-back in the keyframe's own landmark order
    tids = [h.frame_id for h in win.frames if h.frame_id != a]
    u1, u2 = g1.get_frame_update(a, tids), gg.get_frame_update(a, tids)
    assert _close(u2["idepth"], u1["idepth"], 1e-7, 1e-12) and _close(u2["inv_hdd"], u1["inv_hdd"], 1e-6, 0)
    assert np.array_equal(u1["flags"], u2["flags"]) and np.array_equal(u1["n_inliers"], u2["n_inliers"])
    for t in tids:
        assert np.array_equal(u1["status"][t], u2["status"][t])
    # ... and against the CPU oracle
    o = po.OracleWindow(po.default_pba_options())
    syn.load_window(o, win)
    eo, ito, nvo = o.solve()
    assert (ito, nvo) == (it2, nv2) and abs(eo - e2) <= 1e-7 * abs(eo)
    for f in win.frames:
        To, _ = o.get_pose(f.frame_id)
        Tg, _ = gg.get_pose(f.frame_id)
        assert np.abs(To - Tg).max() <= 1e-7
    g1.close()
    gg.close()


def test_group_stage_api_matches_single_window():
    win = syn.make_window(num_frames=4, num_points=240, width=320, height=240, seed=3)
    g1, gg = _single_and_group(win, 2)
    for w in (g1, gg):
        w.begin()
    (e1, n1), (e2, n2) = g1.calculate_energy(), gg.calculate_energy()
    assert n1 == n2 and abs(e1 - e2) <= 1e-10 * abs(e1)
    g1.linearize()
    gg.linearize()
    for name, a, b in zip(["H_pp", "b_pp", "H_schur", "b_schur"], gg.get_system(), g1.get_system()):
        assert np.abs(a - b).max() <= 1e-9 * np.abs(b).max(), name
    s1, s2 = g1.calculate_step(1e-5), gg.calculate_step(1e-5)
    assert np.abs(s1 - s2).max() <= 1e-9
    for f in win.frames:
        l1, l2 = g1.get_landmarks(f.frame_id), gg.get_landmarks(f.frame_id)
        assert _close(l2["idepth_step"], l1["idepth_step"], 1e-7, 1e-12)
        assert _close(l2["hpib"], l1["hpib"], 1e-9, 1e-9) and _close(l2["b_d"], l1["b_d"], 1e-9, 1e-9)
    (e1, n1), (e2, n2) = g1.calculate_energy(), gg.calculate_energy()
    assert n1 == n2 and abs(e1 - e2) <= 1e-9 * abs(e1)
    a1, a2 = g1.accept_step(), gg.accept_step()
    assert abs(a1[0] - a2[0]) <= 1e-10 * a1[0] and abs(a1[1] - a2[1]) <= 1e-7 * a1[1]
    for f in win.frames:
        for x, y in zip(gg.get_frame_state(f.frame_id), g1.get_frame_state(f.frame_id)):
            assert np.abs(x - y).max() <= 1e-9
    # reject path
    g1.linearize()
    gg.linearize()
    g1.calculate_step(1e-5)
    gg.calculate_step(1e-5)
    g1.reject_step()
    gg.reject_step()
    (e1, n1), (e2, n2) = g1.calculate_energy(), gg.calculate_energy()
    assert n1 == n2 and abs(e1 - e2) <= 1e-9 * abs(e1)
    g1.close()
    gg.close()


def test_group_deterministic_mode_is_bit_reproducible():
    """ordered partial sums on every shard + the reducer's shard-order sum: two solves from the same state agree to the last bit"""
    from dsopp_amd import capi
    win = syn.make_window(num_frames=5, num_points=1500, width=320, height=240, seed=9)
    runs = []
    for _ in range(2):
        gg = capi.HipWindowGroup(capi.default_pba_options(), devices=[0, 0], transport=capi.TRANSPORT_LOCAL)
        syn.load_window(gg, win)
        gg.set_deterministic(True)
        gg.snapshot()
        out = []
        for _ in range(2):
            gg.restore()
            e, it, nv = gg.optimize()
            out.append((e, it, nv, np.concatenate([np.concatenate(gg.get_pose(f.frame_id)) for f in win.frames]),
                        np.concatenate([gg.get_landmarks(f.frame_id, False)["idepth"] for f in win.frames])))
        runs.extend(out)
        gg.close()
    e0, it0, nv0, p0, d0 = runs[0]
    for e, it, nv, p, d in runs[1:]:
        assert e == e0 and (it, nv) == (it0, nv0) and np.array_equal(p, p0) and np.array_equal(d, d0)


def test_group_appended_landmarks_keep_their_shard():
    """LocalFrame::update appends landmarks to a keyframe already in the window (PROB_SRC/photometric_bundle_adjustment.cpp:109-123):
    landmark j stays on shard j % n at local index j / n, connections grow with it, read-backs come back in keyframe order"""
    from dsopp_amd import capi
    win = syn.make_window(num_frames=3, num_points=300, width=320, height=240, seed=11)
    intr = win.scene.intrinsics
    g1 = capi.HipWindow(capi.default_pba_options())
    gg = capi.HipWindowGroup(capi.default_pba_options(), devices=[0, 0, 0], transport=capi.TRANSPORT_LOCAL)
    for w in (g1, gg):
        for i, f in enumerate(win.frames):
            w.push_frame(f.frame_id, f.timestamp, f.pixelinfo, None, intr, syn.mat_to_params(f.T_w_c_init), f.exposure, f.affine_init, f.fixed, False)
            n0 = len(f.uv) // 2 + 1   # first half now ...
            w.set_landmarks(f.frame_id, f.uv[:n0], f.idepth_init[:n0], f.patch[:n0], np.zeros(n0, dtype=np.uint8))
            for j in range(i):
                h = win.frames[j]
                w.set_connection(h.frame_id, f.frame_id, np.zeros(len(h.uv) // 2 + 1, dtype=np.uint8))
                w.set_connection(f.frame_id, h.frame_id, np.zeros(n0, dtype=np.uint8))
        w.solve()
        for f in win.frames:              # ... the rest after a solve, appended to the same keyframes
            w.set_landmarks(f.frame_id, f.uv, f.idepth_init, f.patch, np.zeros(len(f.uv), dtype=np.uint8))
            for h in win.frames:
                if h.frame_id != f.frame_id:
                    w.set_connection(f.frame_id, h.frame_id, np.zeros(len(f.uv), dtype=np.uint8))
    for f in win.frames:
        counts = [gg.shard_num_landmarks(s, f.frame_id) for s in range(3)]
        assert sum(counts) == len(f.uv) and max(counts) - min(counts) <= 1
    r1, r2 = g1.solve(), gg.solve()
    assert r1[1:] == r2[1:] and abs(r1[0] - r2[0]) <= 1e-7 * abs(r1[0])
    _compare_state(win, g1, gg)
    g1.close()
    gg.close()


def test_group_with_empty_shards():
    """fewer landmarks per keyframe than shards: some shards hold nothing at all and still take part in every collective"""
    from dsopp_amd import capi
    win = syn.make_window(num_frames=4, num_points=4 * 6, width=320, height=240, seed=33)
    g1 = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g1, win)
    gg = capi.HipWindowGroup(capi.default_pba_options(), devices=[0] * 8, transport=capi.TRANSPORT_LOCAL)
    syn.load_window(gg, win)
    assert [gg.shard_num_landmarks(s, win.frames[1].frame_id) for s in range(8)] == [1, 1, 1, 1, 1, 1, 0, 0]
    r1, r2 = g1.solve(), gg.solve()
    assert r1[1:] == r2[1:] and abs(r1[0] - r2[0]) <= 1e-6 * abs(r1[0])
    _compare_state(win, g1, gg, pose_tol=1e-6)
    g1.close()
    gg.close()


def _drive(backend, win, max_window=4):
    """keyframes through a window of at most 4 frames with marginalisation (tests/test_gpu_sliding_window.py), per-solve log"""
    intr = win.scene.intrinsics
    alive, log = [], []
    for k, f in enumerate(win.frames):
        backend.push_frame(f.frame_id, f.timestamp, f.pixelinfo, None, intr, syn.mat_to_params(f.T_w_c_init), f.exposure, f.affine_init, f.fixed, False)
        backend.set_landmarks(f.frame_id, f.uv, f.idepth_init, f.patch, np.zeros(len(f.uv), dtype=np.uint8))
        for g in alive:
            backend.set_connection(g.frame_id, f.frame_id, np.zeros(len(g.uv), dtype=np.uint8))
            backend.set_connection(f.frame_id, g.frame_id, np.zeros(len(f.uv), dtype=np.uint8))
        alive.append(f)
        if len(alive) < 2:
            continue
        e, it, nv = backend.solve()
        snap = dict(step=k, energy=e, iterations=it, n_valid=nv, poses={}, idepth={}, status={}, frame_ids=backend.frame_ids())
        for g in alive:
            snap["poses"][g.frame_id] = np.concatenate(backend.get_pose(g.frame_id))
            snap["idepth"][g.frame_id] = backend.get_landmarks(g.frame_id, False)["idepth"].copy()
            for h in alive:
                if h.frame_id != g.frame_id:
                    snap["status"][(g.frame_id, h.frame_id)] = backend.get_residuals(g.frame_id, h.frame_id)["status"].copy()
        Hm, bm, em = backend.get_marginalized()
        snap["marg"] = (Hm.copy(), bm.copy(), em)
        log.append(snap)
        if len(alive) == max_window and k + 1 < len(win.frames):
            victim = alive[1]
            for g in alive:
                flags = np.zeros(len(g.uv), dtype=np.uint8)
                flags[::4 if g is victim else 9] = 1
                backend.set_landmarks(g.frame_id, g.uv, g.idepth_init, g.patch, flags)
            backend.mark_frame_marginalized(victim.frame_id)
            alive.remove(victim)
    return log


def test_group_sliding_window_with_marginalisation():
    """the fold-in of pushFrame (updateMarginalizedLinearSystem) is a collective of its own: every shard linearises its landmarks
    flagged for marginalisation, the systems are summed, every shard reduces the same prior"""
    from dsopp_amd import capi
    win = syn.make_window(num_frames=7, num_points=7 * 300, width=320, height=240, seed=61)
    g1 = capi.HipWindow(capi.default_pba_options())
    gg = capi.HipWindowGroup(capi.default_pba_options(), devices=[0, 0], transport=capi.TRANSPORT_LOCAL)
    log1, log2 = _drive(g1, win), _drive(gg, win)
    assert len(log1) == len(log2) == 6
    for s1, s2 in zip(log1, log2):
        k = s1["step"]
        assert s1["frame_ids"] == s2["frame_ids"], k
        assert (s1["iterations"], s1["n_valid"]) == (s2["iterations"], s2["n_valid"]), k
        assert abs(s1["energy"] - s2["energy"]) <= 1e-7 * abs(s1["energy"]), k
        for fid in s1["poses"]:
            assert np.abs(s1["poses"][fid] - s2["poses"][fid]).max() <= 1e-7, (k, fid)
            assert np.abs(s1["idepth"][fid] - s2["idepth"][fid]).max() <= 1e-7 * max(1.0, np.abs(s1["idepth"][fid]).max()), (k, fid)
        for key in s1["status"]:
            assert np.array_equal(s1["status"][key], s2["status"][key]), (k, key)
        (H1, b1, e1), (H2, b2, e2) = s1["marg"], s2["marg"]
        assert H1.shape == H2.shape
        if np.abs(H1).max() > 0:
            assert np.abs(H2 - H1).max() <= 1e-7 * np.abs(H1).max() and np.abs(b2 - b1).max() <= 1e-7 * max(1.0, np.abs(b1).max()), k
            assert abs(e2 - e1) <= 1e-7 * max(1.0, abs(e1)), k
    assert np.abs(log1[-1]["marg"][0]).max() > 0  # the marginal prior is in play
    g1.close()
    gg.close()


def test_group_reference_depth_maps_match_single_window():
    """createReferenceDepthMaps over shards: the level-0 splat planes are summed across the shards before pooling / dilation"""
    win = syn.make_window(num_frames=4, num_points=800, width=320, height=240, seed=21)
    g1, gg = _single_and_group(win, 2)
    g1.solve()
    gg.solve()
    m1, m2 = g1.create_reference_depth_maps(3), gg.create_reference_depth_maps(3)
    for l in range(3):
        (i1, w1), (i2, w2) = m1.get_level(l), m2.get_level(l)
        assert np.array_equal(w1 > 0, w2 > 0), l
        assert _close(w2, w1, 1e-6, 0) and _close(i2, i1, 1e-6, 0), l
    # the tracker refills its one map object after every keyframe
    gg.refill_reference_depth_maps(m2)
    i3, w3 = m2.get_level(0)
    assert _close(w3, m1.get_level(0)[1], 1e-6, 0)
    m1.close()
    m2.close()
    g1.close()
    gg.close()


def test_group_of_one_is_a_plain_window_and_bad_arguments_are_refused():
    from dsopp_amd import capi
    win = syn.make_window(num_frames=3, num_points=150, width=160, height=120, seed=1)
    g1 = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g1, win)
    gg = capi.HipWindowGroup(capi.default_pba_options(), devices=[0])
    syn.load_window(gg, win)
    r1, r2 = g1.solve(), gg.solve()  # same kernels (the atomics' summation order is the only freedom)
    assert r1[1:] == r2[1:] and abs(r1[0] - r2[0]) <= 1e-10 * abs(r1[0])
    assert gg.transport == capi.TRANSPORT_LOCAL and gg.size == 1
    g1.close()
    gg.close()
    with pytest.raises(capi.HipError):   # RCCL refuses two ranks on one device: the group says so instead of hanging
        capi.HipWindowGroup(capi.default_pba_options(), devices=[0, 0], transport=capi.TRANSPORT_RCCL)
    with pytest.raises(capi.HipError):
        capi.HipWindowGroup(capi.default_pba_options(), devices=[capi.device_count()])
    with pytest.raises(capi.HipError):
        capi.HipWindowGroup(capi.default_pba_options(), devices=[])
    with pytest.raises(capi.HipError):
        capi.HipWindowGroup(capi.default_pba_options(), devices=[0] * 17, transport=capi.TRANSPORT_LOCAL)
    # errors of a shard surface through the group with the window's own code; the group stays usable
    gg = capi.HipWindowGroup(capi.default_pba_options(), devices=[0, 0], transport=capi.TRANSPORT_LOCAL)
    with pytest.raises(capi.HipError, match="-2"):
        gg.get_pose(42)
    with pytest.raises(capi.HipError, match="-6"):
        gg.solve()  # empty window
    syn.load_window(gg, win)
    f = win.frames[0]
    with pytest.raises(capi.HipError, match="-1"):
        gg.set_landmarks(f.frame_id, f.uv[:5], f.idepth_init[:5], f.patch[:5], np.zeros(5, dtype=np.uint8))  # landmarks only grow
    e, it, nv = gg.solve()
    assert it > 0 and nv > 0
    gg.close()


def test_group_large_window_takes_the_two_stage_sharded_path():
    """12 keyframes / 50 000 landmarks (BASELINE.json configs[4]) on two shards: each shard is above the two-stage threshold, so the
    shards build their partial systems without atomics, send the sweep's energy scalars as 64 group sums in the one collective, the
    solve launch decides from them, and the closing round sums through the group kernels — the sharded solve must reproduce the single
    window's (same iteration count, same residual count, poses and energy to rounding)"""
    from dsopp_amd import capi
    win = syn.make_window(num_frames=12, num_points=50000, width=640, height=480, seed=1)
    g1, gg = _single_and_group(win, 2)
    e1, it1, nv1 = g1.optimize()
    e2, it2, nv2 = gg.optimize()
    assert (it1, nv1) == (it2, nv2) and it1 == 7 and abs(e1 - e2) <= 1e-9 * abs(e1)
    for f in win.frames:
        (T1, ab1), (T2, ab2) = g1.get_pose(f.frame_id), gg.get_pose(f.frame_id)
        assert np.abs(T1 - T2).max() <= 1e-9 and np.abs(ab1 - ab2).max() <= 1e-9, f.frame_id
    f = win.frames[3]
    l1, l2 = g1.get_landmarks(f.frame_id, False), gg.get_landmarks(f.frame_id, False)
    assert _close(l2["idepth"], l1["idepth"], 1e-9, 1e-12)
    g1.close()
    gg.close()


_STRADDLE_SCRIPT = r"""
import numpy as np
from dsopp_amd import capi, synthetic as syn
# 4 frames x 129 landmarks on 2 shards: shard 0 holds 65 landmarks of every frame (2 chunks of 64 each: 8 chunks), shard 1 holds 64
# (4 chunks).  With the two-stage build above 5 chunks, shard 0 builds its system in two stages and shard 1 with atomics: the two
# paths must hand the ONE collective per iteration the same count and layout (round-3 advisor finding, high).
win = syn.make_window(num_frames=4, num_points=516, width=320, height=240, seed=11)
assert all(len(f.uv) == 129 for f in win.frames)
for lm_mode in (0, 2):
    g1 = capi.HipWindow(capi.default_pba_options()); syn.load_window(g1, win)
    gg = capi.HipWindowGroup(capi.default_pba_options(), devices=[0, 0], transport=capi.TRANSPORT_LOCAL); syn.load_window(gg, win)
    assert [gg.shard_num_landmarks(s, win.frames[0].frame_id) for s in (0, 1)] == [65, 64]
    g1.set_lm_mode(lm_mode); gg.set_lm_mode(lm_mode)
    e1, it1, nv1 = g1.solve()
    e2, it2, nv2 = gg.solve()
    assert (it1, nv1) == (it2, nv2) and abs(e1 - e2) <= 1e-7 * abs(e1), (lm_mode, e1, e2, it1, it2, nv1, nv2)
    for f in win.frames:
        assert np.abs(g1.get_pose(f.frame_id)[0] - gg.get_pose(f.frame_id)[0]).max() <= 1e-7
    g1.close(); gg.close()
print("straddle ok")
"""


def test_shards_straddling_the_two_stage_threshold():
    """One shard above, one below the chunk count that selects the atomic-free build (DSOPP_HIP_TWO_STAGE_MIN_CHUNKS is read once per
    process, hence the subprocess): their per-iteration collective has one size and one layout whichever path a shard takes."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSOPP_HIP_TWO_STAGE_MIN_CHUNKS="5", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", _STRADDLE_SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "straddle ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_FORCED_POOL_SCRIPT = r"""
import numpy as np
from dsopp_amd import capi, synthetic as syn
win = syn.make_window(num_frames=4, num_points=400, width=320, height=240, seed=13)
for lm_mode in (0, 1):
    g1 = capi.HipWindow(capi.default_pba_options()); syn.load_window(g1, win)
    # ONE shard on device 0 with the RCCL transport: with DSOPP_HIP_GROUP_FORCE_POOL the group does not collapse into a plain window —
    # its calls run on the shard's worker thread, the communicator is created there (ncclCommInitRank, one rank) and every Gauss-Newton
    # iteration sums its system through ncclAllReduce on the shard's stream
    gg = capi.HipWindowGroup(capi.default_pba_options(), devices=[0], transport=capi.TRANSPORT_RCCL); syn.load_window(gg, win)
    assert gg.size == 1 and gg.transport == capi.TRANSPORT_RCCL, (gg.size, gg.transport)
    g1.set_lm_mode(lm_mode); gg.set_lm_mode(lm_mode)
    e1, it1, nv1 = g1.solve()
    e2, it2, nv2 = gg.solve()
    assert (it1, nv1) == (it2, nv2) and abs(e1 - e2) <= 1e-9 * abs(e1), (lm_mode, e1, e2)
    for f in win.frames:
        assert np.abs(g1.get_pose(f.frame_id)[0] - gg.get_pose(f.frame_id)[0]).max() <= 1e-9
    a, b = win.frames[0].frame_id, win.frames[2].frame_id
    assert np.abs(g1.get_covariance(a, b) - gg.get_covariance(a, b)).max() <= 1e-6 * np.abs(g1.get_covariance(a, b)).max()
    g1.close(); gg.close()
print("forced pool ok")
"""


def test_group_of_one_through_the_worker_thread_and_rccl():
    """the RCCL transport of the group (worker thread -> ncclCommInitRank -> per-iteration ncclAllReduce) executed with one rank"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSOPP_HIP_GROUP_FORCE_POOL="1", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", _FORCED_POOL_SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "forced pool ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_P2P_SCRIPT = r"""
import sys
import numpy as np
from dsopp_amd import capi, synthetic as syn
shards = int(sys.argv[1])
win = syn.make_window(num_frames=5, num_points=1100, width=320, height=240, seed=27)
close = lambda a, b, rtol, atol: np.abs(np.asarray(a) - np.asarray(b)).max() <= atol + rtol * np.abs(np.asarray(b)).max()
for lm_mode in (0, 2):
    g1 = capi.HipWindow(capi.default_pba_options()); syn.load_window(g1, win)
    gg = capi.HipWindowGroup(capi.default_pba_options(), devices=[0] * shards, transport=capi.TRANSPORT_P2P); syn.load_window(gg, win)
    assert gg.transport == capi.TRANSPORT_P2P
    g1.set_lm_mode(lm_mode); gg.set_lm_mode(lm_mode)
    e1, it1, nv1 = g1.solve()
    e2, it2, nv2 = gg.solve()
    assert (it1, nv1) == (it2, nv2) and abs(e1 - e2) <= 1e-7 * abs(e1), (e1, e2, it1, it2)
    for f in win.frames:
        assert np.abs(g1.get_pose(f.frame_id)[0] - gg.get_pose(f.frame_id)[0]).max() <= 1e-7
        l1, l2 = g1.get_landmarks(f.frame_id, False), gg.get_landmarks(f.frame_id, False)
        assert close(l2["idepth"], l1["idepth"], 1e-7, 1e-12) and np.array_equal(l1["flags"], l2["flags"])
    a, b = win.frames[1].frame_id, win.frames[3].frame_id
    assert close(gg.get_covariance(a, b), g1.get_covariance(a, b), 1e-5, 0)
    # reference depth maps: planes larger than the receive area go through the in-process reducer of the same group
    m1, m2 = g1.create_reference_depth_maps(2), gg.create_reference_depth_maps(2)
    for lvl in range(2):
        (i1, w1), (i2, w2) = m1.get_level(lvl), m2.get_level(lvl)
        assert np.array_equal(w1 > 0, w2 > 0) and close(i2, i1, 1e-9, 1e-12)
    m1.close(); m2.close(); g1.close(); gg.close()
# deterministic mode: two solves of two groups are bit-identical (sums in shard order)
poses = []
for _ in range(2):
    gg = capi.HipWindowGroup(capi.default_pba_options(), devices=[0] * shards, transport=capi.TRANSPORT_P2P); syn.load_window(gg, win)
    gg.set_deterministic(True); gg.solve()
    poses.append(np.stack([gg.get_pose(f.frame_id)[0] for f in win.frames])); gg.close()
assert np.array_equal(poses[0], poses[1])
print("p2p ok")
"""


@pytest.mark.parametrize("shards", [2, 4])
def test_peer_to_peer_transport_matches_single_window(shards):
    """DSOPP_HIP_TRANSPORT_P2P: every shard stores its partial sums into every shard's receive area and sums its own in shard order —
    flags instead of a host barrier, one kernel per shard and collective on distinct devices.  Kernels of shards that SHARE a device
    cannot wait for each other (their streams may sit on one hardware queue: seen as a time-out — an error, not a hang — with one launch
    per shard), so there ONE launch plays every shard (a row of workgroups each): the protocol — double-buffered receive areas,
    generation-valued flags per slice, bounded system-scope waits, sums in shard order — is the one a multi-device group runs."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", _P2P_SCRIPT, str(shards)], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "p2p ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


_P2P_MULTI_DEVICE_SCRIPT = r"""
import numpy as np
from dsopp_amd import capi, synthetic as syn
n = min(capi.device_count(), 4)
win = syn.make_window(num_frames=5, num_points=1100, width=320, height=240, seed=29)
res = {}
for name, transport in (("local", capi.TRANSPORT_LOCAL), ("p2p", capi.TRANSPORT_P2P)):
    gg = capi.HipWindowGroup(capi.default_pba_options(), devices=list(range(n)), transport=transport); syn.load_window(gg, win)
    assert gg.transport == transport
    gg.set_deterministic(True)
    e, it, nv = gg.solve()
    res[name] = (e, it, nv, np.stack([gg.get_pose(f.frame_id)[0] for f in win.frames]),
                 np.concatenate([gg.get_landmarks(f.frame_id, False)["idepth"] for f in win.frames]))
    gg.close()
a, b = res["local"], res["p2p"]
assert a[:3] == b[:3], (a[:3], b[:3])
assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])  # both sum in shard order: bitwise
print("p2p multi-device ok", n)
"""


def _device_count():
    from dsopp_amd import capi
    return capi.device_count()


@pytest.mark.skipif("_device_count() < 2", reason="needs two GPUs: the distinct-device branch of the peer-to-peer all-reduce")
def test_peer_to_peer_transport_on_distinct_devices_is_opt_in_and_matches_local():
    """Across distinct devices DSOPP_HIP_TRANSPORT_P2P is refused without DSOPP_HIP_P2P_EXPERIMENTAL=1 (that branch — one kernel per shard,
    remote stores over xGMI, generation-valued flags — has never run where this library was built); with the opt-in it has to reproduce
    the in-process reducer bit for bit in deterministic mode (both add the shards' partial sums in shard order)."""
    import os
    import subprocess
    import sys
    from dsopp_amd import capi
    with pytest.raises(capi.HipError) as ei:
        capi.HipWindowGroup(capi.default_pba_options(), devices=[0, 1], transport=capi.TRANSPORT_P2P)
    assert "experimental" in str(ei.value).lower()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, DSOPP_HIP_P2P_EXPERIMENTAL="1")
    r = subprocess.run([sys.executable, "-c", _P2P_MULTI_DEVICE_SCRIPT], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "p2p multi-device ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
