"""GPU parity on the shapes the reference's solver tests and its tracker exercise beyond the 4-frame clique: minimum and
large windows (K > 64 takes the second row chunk of the solve kernel and the second target group of the sweeps), ragged
landmark counts incl. an empty frame, pre-set residual statuses and landmark flags, and the C1
bench configuration itself at full size.  Same bars as tests/test_gpu_pba.py (pose update bar of BASELINE.json: 1e-5)."""
import numpy as np
import pytest

from dsopp_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _load(backend, win, keep_pair=None, statuses=None, flags=None):
    """synthetic.load_window with an optional connection filter / per-connection statuses / per-frame landmark flags"""
    intr = win.scene.intrinsics
    for i, f in enumerate(win.frames):
        backend.push_frame(f.frame_id, f.timestamp, f.pixelinfo, None, intr, syn.mat_to_params(f.T_w_c_init), f.exposure,
                           f.affine_init, f.fixed, False)
        fl = np.zeros(len(f.uv), dtype=np.uint8) if flags is None else flags[f.frame_id]
        backend.set_landmarks(f.frame_id, f.uv, f.idepth_init, f.patch, fl)
        for j in range(i):
            g = win.frames[j]
            for (r, t) in ((g, f), (f, g)):
                if keep_pair is not None and not keep_pair(r.frame_id, t.frame_id):
                    continue
                st = None if statuses is None else statuses.get((r.frame_id, t.frame_id))
                if st is None:
                    st = np.zeros(len(r.uv), dtype=np.uint8)
                backend.set_connection(r.frame_id, t.frame_id, st)
    return backend


def _pair(win, loader=_load, **kw):
    from dsopp_amd import capi
    from oracle import pyoracle as po
    o = loader(po.OracleWindow(po.default_pba_options()), win, **kw)
    g = loader(capi.HipWindow(capi.default_pba_options()), win, **kw)
    return o, g


def _compare_solve(o, g, win, pose_tol=1e-7, keep_pair=None):
    eo, ito, nvo = o.solve()
    eg, itg, nvg = g.solve()
    assert (ito, nvo) == (itg, nvg)
    assert abs(eo - eg) <= 1e-7 * max(abs(eo), 1e-30)
    for f in win.frames:
        To, abo = o.get_pose(f.frame_id)
        Tg, abg = g.get_pose(f.frame_id)
        assert np.abs(To - Tg).max() <= pose_tol, (f.frame_id, np.abs(To - Tg).max())
        assert np.abs(abo - abg).max() <= pose_tol
        lo, lg = o.get_landmarks(f.frame_id), g.get_landmarks(f.frame_id, False)
        if len(lo["idepth"]):
            assert np.abs(lg["idepth"] - lo["idepth"]).max() <= 1e-6 * np.abs(lo["idepth"]).max() + 1e-9
            assert np.array_equal(lo["flags"] & 3, lg["flags"] & 3)
            assert np.array_equal(lo["n_inliers"], lg["n_inliers"])
    for fr in win.frames:
        for ft in win.frames:
            if fr.frame_id == ft.frame_id or (keep_pair is not None and not keep_pair(fr.frame_id, ft.frame_id)):
                continue
            if len(fr.uv):
                assert np.array_equal(o.get_residuals(fr.frame_id, ft.frame_id)["status"], g.get_residuals(fr.frame_id, ft.frame_id)["status"])


@pytest.mark.parametrize("frames,points", [(2, 64), (3, 97), (9, 450), (12, 600)])
def test_window_sizes(frames, points):
    win = syn.make_window(num_frames=frames, num_points=points, width=320, height=240, seed=20 + frames)
    o, g = _pair(win)
    _compare_solve(o, g, win)
    g.close()


def test_ragged_frames():
    """landmark counts 0 / 1 / 17 / 65 / 130: an empty frame, and no multiple of the 16-item / 64-landmark workgroups.
    (The connection graph stays a full clique: the reference indexes `residuals.at(target_frame.id)` for every frame pair,
    hessian_block_evaluation.hpp:59,203 — a sparse graph is not an input it accepts.)"""
    win = syn.make_window(num_frames=5, num_points=5 * 130, width=320, height=240, seed=31)
    for f, n in zip(win.frames, (0, 1, 17, 65, 130)):
        f.uv, f.idepth_gt, f.idepth_init, f.patch = f.uv[:n], f.idepth_gt[:n], f.idepth_init[:n], f.patch[:n]
    o, g = _pair(win)
    _compare_solve(o, g, win)
    g.close()


def test_preset_statuses_and_flags():
    """connections that start as OUTLIER / OCCLUDED / OOB, landmarks flagged marginalized or outlier at upload"""
    win = syn.make_window(num_frames=4, num_points=320, width=320, height=240, seed=37)
    rng = np.random.default_rng(5)
    statuses, flags = {}, {}
    for a in win.frames:
        flags[a.frame_id] = (rng.random(len(a.uv)) < 0.1).astype(np.uint8) * rng.integers(1, 3, len(a.uv)).astype(np.uint8)
        for b in win.frames:
            if a.frame_id != b.frame_id:
                st = np.zeros(len(a.uv), dtype=np.uint8)
                bad = rng.random(len(a.uv)) < 0.15
                st[bad] = rng.integers(1, 4, int(bad.sum()))
                statuses[(a.frame_id, b.frame_id)] = st
    o, g = _pair(win, statuses=statuses, flags=flags)
    _compare_solve(o, g, win)
    g.close()


def test_bench_configuration_full_size():
    """C1 of BASELINE.json exactly as bench.py runs it: 7 keyframes, 2000 points, 640x480, 7 LM iterations"""
    win = syn.make_window(num_frames=7, num_points=2000, width=640, height=480, seed=0)
    o, g = _pair(win)
    g.begin()
    e_init, n_init = g.calculate_energy()
    eo, ito, nvo = o.optimize()
    eg, itg, nvg = g.optimize()
    assert (ito, nvo) == (itg, nvg) and ito == 7
    assert abs(eo - eg) <= 1e-7 * abs(eo)
    for f in win.frames:
        To, abo = o.get_pose(f.frame_id)
        Tg, abg = g.get_pose(f.frame_id)
        assert np.abs(To - Tg).max() <= 1e-7 and np.abs(abo - abg).max() <= 1e-7
    # size-independent property: the photometric energy per valid residual went down (poses were perturbed from the truth)
    assert eg / nvg < e_init / n_init
    g.close()


@pytest.mark.parametrize("frames,points", [(7, 20000), (12, 50000)])
def test_large_baseline_configurations(frames, points):
    """the windows of BASELINE.json's 8-GPU configurations (C3: 7 KF / 20 000 points, C4: 12 KF / 50 000 points) solved whole on
    one GPU against the CPU oracle (16 threads): the full 7-iteration LM solve, energies 1e-7 rel,
    poses / affine 1e-7, plus the size-independent property that the energy per valid residual decreases"""
    from oracle import pyoracle as po
    win = syn.make_window(num_frames=frames, num_points=points, width=640, height=480, seed=2)
    from dsopp_amd import capi
    po.set_threads(16)
    o = _load(po.OracleWindow(po.default_pba_options()), win)
    g = _load(capi.HipWindow(capi.default_pba_options()), win)
    g.begin()
    e_init, n_init = g.calculate_energy()
    eo, ito, nvo = o.optimize()
    eg, itg, nvg = g.optimize()
    po.set_threads(1)
    assert (ito, nvo) == (itg, nvg) and ito == 7
    assert abs(eo - eg) <= 1e-7 * abs(eo)
    for f in win.frames:
        To, abo = o.get_pose(f.frame_id)
        Tg, abg = g.get_pose(f.frame_id)
        assert np.abs(To - Tg).max() <= 1e-7 and np.abs(abo - abg).max() <= 1e-7
    assert eg / nvg < e_init / n_init
    g.close()


@pytest.mark.parametrize("frames", [13, 16])
@pytest.mark.parametrize("deterministic", [False, True])
def test_window_at_capacity(frames, deterministic):
    """windows of 13 and of DSOPP_HIP_MAX_FRAMES = 16 keyframes (K = 104 / 128: four and five MFMA tiles per wave in the Schur
    build, two lanes-rows per lane in the solve's back-substitution): the full solve against the oracle in both summation modes;
    a 17th keyframe is refused with DSOPP_HIP_ERR_CAPACITY and the window stays as it was"""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    # a slower camera than the default: 16 keyframes of the default motion leave the scene
    base = syn.BASE_MOTION.copy()
    syn.BASE_MOTION[:] = base * 0.4
    try:
        win = syn.make_window(num_frames=frames + 1, num_points=(frames + 1) * 120, width=320, height=240, seed=70 + frames)
    finally:
        syn.BASE_MOTION[:] = base
    extra = win.frames.pop()
    po.set_threads(8)
    o = _load(po.OracleWindow(po.default_pba_options()), win)
    g = _load(capi.HipWindow(capi.default_pba_options()), win)
    g.set_deterministic(deterministic)
    assert g.num_frames == frames
    if frames == 16:
        intr = win.scene.intrinsics
        with pytest.raises(capi.HipError, match="-5"):
            g.push_frame(extra.frame_id, extra.timestamp, extra.pixelinfo, None, intr, syn.mat_to_params(extra.T_w_c_init), extra.exposure,
                         extra.affine_init, False, False)
        assert g.num_frames == 16 and g.frame_ids() == [f.frame_id for f in win.frames]
    _compare_solve(o, g, win)
    po.set_threads(1)
    g.close()


def test_snapshot_restore_is_idempotent(small_window):
    """restore() + optimize() must reproduce the first solve bit for bit (deterministic reductions, no atomics-order effects
    on the accepted state beyond the stated tolerance)"""
    from dsopp_amd import capi
    g = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g, small_window)
    g.snapshot()
    first = g.optimize()
    poses1 = [np.concatenate(g.get_pose(f.frame_id)) for f in small_window.frames]
    g.restore()
    second = g.optimize()
    poses2 = [np.concatenate(g.get_pose(f.frame_id)) for f in small_window.frames]
    assert first[1:] == second[1:] and abs(first[0] - second[0]) <= 1e-9 * abs(first[0])
    for a, b in zip(poses1, poses2):
        assert np.abs(a - b).max() <= 1e-9
    done, e = g.optimize_repeated(10)
    assert done == 10
    g.close()


@pytest.mark.parametrize("lm_mode", [0, 1, 2])
@pytest.mark.parametrize("fej", [1, 0])
def test_reject_path_without_force_accept(lm_mode, fej):
    """force_accept = false with enough iterations that LM steps get rejected near the optimum: the device-driven loops must
    follow the reference's reject / re-linearise bookkeeping (levenberg_marquardt_algorithm.hpp:88-128) like the oracle"""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    # seed 42: the oracle accepts 4 steps and then rejects every further one (checked with the stage API); seed 41 never rejects
    win = syn.make_window(num_frames=4, num_points=300, width=320, height=240, seed=42)
    kw = dict(force_accept=0, max_iterations=14, first_estimate_jacobians=fej)
    o = syn.load_window(po.OracleWindow(po.default_pba_options(**kw)), win)
    g = syn.load_window(capi.HipWindow(capi.default_pba_options(**kw)), win)
    g.set_lm_mode(lm_mode)
    eo, ito, nvo = o.optimize()
    eg, itg, nvg = g.optimize()
    assert (ito, nvo) == (itg, nvg), (ito, itg, nvo, nvg)
    assert abs(eo - eg) <= 1e-7 * abs(eo)
    for f in win.frames:
        To, abo = o.get_pose(f.frame_id)
        Tg, abg = g.get_pose(f.frame_id)
        assert np.abs(To - Tg).max() <= 1e-7 and np.abs(abo - abg).max() <= 1e-7
        assert np.abs(g.get_landmarks(f.frame_id, False)["idepth"] - o.get_landmarks(f.frame_id)["idepth"]).max() <= 1e-7
    for fr in win.frames:
        for ft in win.frames:
            if fr.frame_id != ft.frame_id:
                assert np.array_equal(o.get_residuals(fr.frame_id, ft.frame_id)["status"], g.get_residuals(fr.frame_id, ft.frame_id)["status"])
    g.close()


@pytest.mark.parametrize("variant", ["deterministic", "group2"])
@pytest.mark.parametrize("fej", [1, 0])
def test_reject_path_on_the_two_stage_and_the_sharded_loop(variant, fej):
    """the same reject / re-linearise bookkeeping where the LM decision is taken from group sums (atomic-free two-stage build, here
    forced by the deterministic mode) and from all-reduced sums (a window group of two landmark shards): in both the decision is the
    prologue of the solve launch and the rejected candidate's system has already been built when it is taken"""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    win = syn.make_window(num_frames=4, num_points=300, width=320, height=240, seed=42)   # rejects from the 5th step on (see above)
    kw = dict(force_accept=0, max_iterations=14, first_estimate_jacobians=fej)
    o = syn.load_window(po.OracleWindow(po.default_pba_options(**kw)), win)
    if variant == "group2":
        g = capi.HipWindowGroup(capi.default_pba_options(**kw), devices=[0, 0], transport=capi.TRANSPORT_LOCAL)
    else:
        g = capi.HipWindow(capi.default_pba_options(**kw))
    syn.load_window(g, win)
    if variant == "deterministic":
        g.set_deterministic(True)
    eo, ito, nvo = o.optimize()
    eg, itg, nvg = g.optimize()
    assert (ito, nvo) == (itg, nvg), (ito, itg, nvo, nvg)
    assert abs(eo - eg) <= 1e-7 * abs(eo)
    for f in win.frames:
        To, abo = o.get_pose(f.frame_id)
        Tg, abg = g.get_pose(f.frame_id)
        assert np.abs(To - Tg).max() <= 1e-7 and np.abs(abo - abg).max() <= 1e-7
        assert np.abs(g.get_landmarks(f.frame_id, False)["idepth"] - o.get_landmarks(f.frame_id)["idepth"]).max() <= 1e-7
    for fr in win.frames:
        for ft in win.frames:
            if fr.frame_id != ft.frame_id:
                assert np.array_equal(o.get_residuals(fr.frame_id, ft.frame_id)["status"], g.get_residuals(fr.frame_id, ft.frame_id)["status"])
    g.close()


@pytest.mark.parametrize("fej", [1, 0])
def test_affine_brightness_and_exposure(fej):
    """non-trivial photometric parameters: per-frame affine brightness (a, b) in the images and in the initial state,
    different exposure times — exercises the a / b Jacobian columns, sigma_r / b_r0 of the FEJ path and the affine priors"""
    win = syn.make_window(num_frames=4, num_points=320, width=320, height=240, seed=53, affine_jitter=True)
    rng = np.random.default_rng(8)
    for f in win.frames:
        f.exposure = float(rng.uniform(0.7, 1.3))
        if not f.fixed:
            f.affine_init = f.affine_gt + rng.normal(0, [0.01, 0.5])
    from dsopp_amd import capi
    from oracle import pyoracle as po
    # the production regularisers (1e12, 1e8) pin (a, b) to zero; weak ones let the solver really estimate them
    kw = dict(first_estimate_jacobians=fej, affine_brightness_regularizer=(1e2, 1e-2))
    o = syn.load_window(po.OracleWindow(po.default_pba_options(**kw)), win)
    g = syn.load_window(capi.HipWindow(capi.default_pba_options(**kw)), win)
    _compare_solve(o, g, win)
    assert any(abs(g.get_pose(f.frame_id)[1]).max() > 1e-3 for f in win.frames[1:])
    g.close()
    # and with the production regularisers
    o = syn.load_window(po.OracleWindow(po.default_pba_options(first_estimate_jacobians=fej)), win)
    g = syn.load_window(capi.HipWindow(capi.default_pba_options(first_estimate_jacobians=fej)), win)
    _compare_solve(o, g, win)
    g.close()


def test_float_mode_full_solve(small_window):
    """DSOPP_HIP_F32 (images and residual / Jacobian rows in fp32, fp64 accumulation — the analogue of the reference's
    -DUSE_FLOAT build) through the fused loop: fp32 round-off class against the fp64 oracle"""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    win = small_window
    o = syn.load_window(po.OracleWindow(po.default_pba_options()), win)
    g = syn.load_window(capi.HipWindow(capi.default_pba_options(dtype=capi.F32)), win)
    eo, ito, nvo = o.optimize()
    eg, itg, nvg = g.optimize()
    assert ito == itg and abs(nvo - nvg) <= 3
    assert abs(eo - eg) <= 2e-3 * abs(eo)
    for f in win.frames:
        To, abo = o.get_pose(f.frame_id)
        Tg, abg = g.get_pose(f.frame_id)
        assert np.abs(To - Tg).max() <= 2e-4, np.abs(To - Tg).max()
    g.close()


def test_frame_update_export_matches_getters(small_window):
    """dsopp_hip_window_get_frame_update (updateFrame in one transfer) == the per-array getters: served from what solve()
    prefetched behind its own synchronisation (all targets, a subset, a permuted subset), and packed on demand once another
    call has touched the window"""
    from dsopp_amd import capi
    g = syn.load_window(capi.HipWindow(capi.default_pba_options()), small_window)
    ids = [f.frame_id for f in small_window.frames]

    def check():
        for fid in ids:
            others = [t for t in ids if t != fid]
            for targets in (others, others[:1], others[::-1]):
                up = g.get_frame_update(fid, targets)
                lm = g.get_landmarks(fid, False)
                for k in ("idepth", "inv_hdd", "relative_baseline", "n_inliers", "flags"):
                    assert np.array_equal(up[k], lm[k]), k
                for t in targets:
                    assert np.array_equal(up["status"][t], g.get_residuals(fid, t)["status"])

    g.solve()
    check()                      # prefetched by solve()
    g.optimize()                 # moves idepths / statuses: the prefetched copy is stale and must not be served
    check()
    g.solve()
    g.update_point_statuses()
    check()
    g.close()


def test_async_solve_of_concurrent_windows(small_window):
    """dsopp_hip_window_optimize_async / _wait: several independent windows enqueued from one host thread, each on its own
    stream, give bit for bit what the blocking call gives; misuse is reported"""
    import ctypes
    from dsopp_amd import capi
    hip = ctypes.CDLL("libamdhip64.so")
    ref = capi.HipWindow(capi.default_pba_options())
    syn.load_window(ref, small_window)
    want = ref.optimize()
    want_poses = [np.concatenate(ref.get_pose(f.frame_id)) for f in small_window.frames]
    wins, streams = [], []
    for _ in range(4):
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        g = capi.HipWindow(capi.default_pba_options(), stream=st.value)
        syn.load_window(g, small_window)
        wins.append(g)
        streams.append(st)
    with pytest.raises(capi.HipError):
        wins[0].optimize_wait()          # nothing pending
    for g in wins:
        g.optimize_async()
    with pytest.raises(capi.HipError):
        wins[0].optimize_async()         # already pending
    for g in wins:
        got = g.optimize_wait()
        assert got[1:] == want[1:] and abs(got[0] - want[0]) <= 1e-9 * abs(want[0])
        for f, wp in zip(small_window.frames, want_poses):
            assert np.abs(np.concatenate(g.get_pose(f.frame_id)) - wp).max() <= 1e-9
    for g in wins + [ref]:
        g.close()
    for st in streams:
        hip.hipStreamDestroy(st)


def test_deterministic_mode_is_bit_reproducible_and_matches_oracle():
    """dsopp_hip_window_set_deterministic: the reduced normal equations are built without atomics (per-workgroup partial systems,
    then one ordered sum per entry), so repeated solves from the same state agree to the last bit; the default path of a small
    window sums H_schur with fp64 atomics and only agrees to rounding.  Both match the oracle."""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    win = syn.make_window(num_frames=5, num_points=1500, width=320, height=240, seed=41)
    o = syn.load_window(po.OracleWindow(po.default_pba_options()), win)
    eo, ito, nvo = o.optimize()
    g = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
    g.snapshot()
    runs = []
    for det in (True, True, True, False):
        g.set_deterministic(det)
        g.restore()
        e, it, nv = g.optimize()
        states = np.concatenate([np.concatenate(g.get_frame_state(f.frame_id)) for f in win.frames])
        idepths = np.concatenate([g.get_landmarks(f.frame_id, False)["idepth"] for f in win.frames])
        runs.append((e, it, nv, states, idepths))
    for e, it, nv, states, idepths in runs:
        assert (it, nv) == (ito, nvo) and abs(e - eo) <= 1e-7 * abs(eo)
    for k in (1, 2):   # deterministic runs: bit-identical
        assert runs[k][0] == runs[0][0]
        assert np.array_equal(runs[k][3], runs[0][3]) and np.array_equal(runs[k][4], runs[0][4])
    assert np.abs(runs[3][3] - runs[0][3]).max() <= 1e-9   # atomics path: equal to rounding
    for f in win.frames:
        To, abo = o.get_pose(f.frame_id)
        Tg, abg = g.get_pose(f.frame_id)
        assert np.abs(To - Tg).max() <= 1e-7 and np.abs(abo - abg).max() <= 1e-7
    g.close()


def test_deterministic_stage_api_systems_are_bit_reproducible():
    """under set_deterministic the stage API builds H_pp / b_pp / H_schur / b_schur through the same atomic-free two-stage path:
    two linearisations of the same state give identical arrays bit for bit, equal to the oracle's and to the atomics build
    within the usual tolerance"""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    win = syn.make_window(num_frames=5, num_points=1500, width=320, height=240, seed=43)
    o = syn.load_window(po.OracleWindow(po.default_pba_options()), win)
    o.begin()
    o.calculate_energy()
    o.linearize()
    ref = o.get_system()
    g = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
    got = []
    for det in (True, True, False):
        g.set_deterministic(det)
        g.begin()
        g.calculate_energy()
        g.linearize()
        got.append(g.get_system())
    for a, b in zip(got[0], got[1]):
        assert np.array_equal(a, b)
    for sysm in got:
        for name, a, r in zip(["H_pp", "b_pp", "H_schur", "b_schur"], sysm, ref):
            scale = max(1.0, np.abs(r).max())
            assert np.abs(a - r).max() <= 1e-9 * scale, name
            if name == "H_pp":   # the fixed frame's 1e16 prior dominates the max: the free part against its own scale too
                assert np.abs(a[8:, 8:] - r[8:, 8:]).max() <= 1e-9 * np.abs(r[8:, 8:]).max()
    for a in (got[0][0], got[0][2]):
        assert np.array_equal(a, a.T)   # written as symmetric pairs
    g.close()


@pytest.mark.parametrize("deterministic", [False, True])
def test_optimize_repeated_equals_single_solves(small_window, deterministic):
    """the bench helper enqueues its solves back to back (restore -> LM loop, results fetched per batch): the iteration count is
    exact for targets that are and are not multiples of the LM budget, and the window is left in the state a single solve from
    the snapshot with the last solve's budget leaves it in"""
    from dsopp_amd import capi
    g = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g, small_window)
    g.set_deterministic(deterministic)
    g.snapshot()
    budget = g.options.max_iterations
    for target, last in ((3 * budget, budget), (2 * budget + 3, 3), (2, 2)):
        done, e_rep = g.optimize_repeated(target)
        assert done == target
        rep = [np.concatenate(g.get_pose(f.frame_id)) for f in small_window.frames]
        g.restore()
        g.set_max_iterations(last)
        e_one, it, _ = g.optimize()
        g.set_max_iterations(budget)
        assert it == last
        one = [np.concatenate(g.get_pose(f.frame_id)) for f in small_window.frames]
        assert abs(e_rep - e_one) <= 1e-9 * abs(e_one)
        for a, b in zip(rep, one):
            assert np.abs(a - b).max() <= (0.0 if deterministic else 1e-9)
    g.close()


def test_large_window_two_stage_is_bit_reproducible():
    """windows above 192 landmark chunks take the two-stage build by themselves (C3 size: 7 frames / 20 000 points): reproducible"""
    from dsopp_amd import capi
    win = syn.make_window(num_frames=7, num_points=20000, width=640, height=480, seed=0)
    g = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
    g.snapshot()
    out = []
    for _ in range(2):
        g.restore()
        e, it, nv = g.optimize()
        out.append((e, it, nv, np.concatenate([np.concatenate(g.get_frame_state(f.frame_id)) for f in win.frames])))
    assert out[0][:3] == out[1][:3] and np.array_equal(out[0][3], out[1][3])
    g.close()


_K3_BACKSUB_SCRIPT = r"""
import sys
import numpy as np
from dsopp_amd import capi, synthetic as syn
out = {}
for name, (F, P, W, H, seed) in {"small": (5, 600, 320, 240, 3), "wide": (10, 1500, 320, 240, 5), "large": (7, 14000, 640, 480, 7),
                                   "fourteen": (14, 1800, 320, 240, 9)}.items():  # (13 - 16 keyframes: four frame slots per lane, ONE prefetched pass)
    win = syn.make_window(num_frames=F, num_points=P, width=W, height=H, seed=seed)
    g = capi.HipWindow(capi.default_pba_options()); syn.load_window(g, win)
    e, it, nv = g.solve()
    poses = np.concatenate([np.concatenate(g.get_pose(f.frame_id)) for f in win.frames])
    idepths = np.concatenate([g.get_landmarks(f.frame_id, with_hpib=False)["idepth"] for f in win.frames])
    # the same window again through a batch of back-to-back solves (every solve launch draws its own tickets / sequence number)
    g.snapshot(); g.restore()
    n, e_rep = g.optimize_repeated(21)
    out[name] = (e, it, nv, poses, idepths, n, e_rep)
    g.close()
np.savez(sys.argv[1], **{f"{k}_{i}": np.asarray(v) for k, t in out.items() for i, v in enumerate(t)})
print("k3 flow ok")
"""


def test_idepth_back_substitution_inside_the_solve_launch_equals_the_kernel_flow(tmp_path):
    """calculateIdepths by the landmark workgroups of the solve launch (default) against the round-3 flow (DSOPP_HIP_K3_BACKSUB=0, read
    once per process: a back-substitution kernel on large windows, fused into the sweep on small ones) — 256- and 512-thread solve
    kernels, atomic and two-stage Schur builds: same iterations and residual counts, energies / poses / inverse depths to rounding."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for flow in ("1", "0"):
        path = str(tmp_path / f"flow{flow}.npz")
        env = dict(os.environ, DSOPP_HIP_K3_BACKSUB=flow, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", _K3_BACKSUB_SCRIPT, path], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "k3 flow ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        res[flow] = np.load(path)
    a, b = res["1"], res["0"]
    for name in ("small", "wide", "large", "fourteen"):
        assert int(a[f"{name}_1"]) == int(b[f"{name}_1"]) and int(a[f"{name}_2"]) == int(b[f"{name}_2"]), name
        assert abs(float(a[f"{name}_0"]) - float(b[f"{name}_0"])) <= 1e-9 * abs(float(b[f"{name}_0"])), name
        assert np.abs(a[f"{name}_3"] - b[f"{name}_3"]).max() <= 1e-9, name
        assert np.abs(a[f"{name}_4"] - b[f"{name}_4"]).max() <= 1e-8 * max(1.0, np.abs(b[f"{name}_4"]).max()), name
        assert int(a[f"{name}_5"]) == int(b[f"{name}_5"]) and abs(float(a[f"{name}_6"]) - float(b[f"{name}_6"])) <= 1e-9 * abs(float(b[f"{name}_6"])), name


_COMB_COPIES_SCRIPT = r"""
import sys
import numpy as np
from dsopp_amd import capi, synthetic as syn
out = {}
for name, (F, P, W, H, seed) in {"small": (5, 600, 320, 240, 3), "seven": (7, 2000, 320, 240, 11), "mid": (7, 6000, 640, 480, 7)}.items():
    win = syn.make_window(num_frames=F, num_points=P, width=W, height=H, seed=seed)
    g = capi.HipWindow(capi.default_pba_options()); syn.load_window(g, win)
    e, it, nv = g.solve()
    poses = np.concatenate([np.concatenate(g.get_pose(f.frame_id)) for f in win.frames])
    idepths = np.concatenate([g.get_landmarks(f.frame_id, with_hpib=False)["idepth"] for f in win.frames])
    g.snapshot(); g.restore()
    n, e_rep = g.optimize_repeated(14)
    out[name] = (e, it, nv, poses, idepths, n, e_rep)
    g.close()
np.savez(sys.argv[1], **{f"{k}_{i}": np.asarray(v) for k, t in out.items() for i, v in enumerate(t)})
print("comb copies ok")
"""


def test_combined_system_accumulated_in_several_copies_equals_the_single_copy(tmp_path):
    """The reduction launch of an unsharded window on the atomics path may spread its f64 atomics over several copies of the combined
    system, which the solve launch adds while loading (pba.hip: comb_copies_active; by default from 80 chunks of 64 landmarks up to the
    two-stage threshold, at most 7 keyframes).  Forced on for every window (DSOPP_HIP_COMB_COPIES_MIN_CHUNKS=1, read once per process)
    with 4 and with 2 copies against the single copy: same iterations and residual counts, energies / poses / inverse depths to rounding."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for copies in ("1", "2", "4"):
        path = str(tmp_path / f"copies{copies}.npz")
        env = dict(os.environ, DSOPP_HIP_COMB_COPIES=copies, DSOPP_HIP_COMB_COPIES_MIN_CHUNKS="1", PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", _COMB_COPIES_SCRIPT, path], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "comb copies ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
        res[copies] = np.load(path)
    b = res["1"]
    for copies in ("2", "4"):
        a = res[copies]
        for name in ("small", "seven", "mid"):
            assert int(a[f"{name}_1"]) == int(b[f"{name}_1"]) and int(a[f"{name}_2"]) == int(b[f"{name}_2"]), (copies, name)
            assert abs(float(a[f"{name}_0"]) - float(b[f"{name}_0"])) <= 1e-9 * abs(float(b[f"{name}_0"])), (copies, name)
            assert np.abs(a[f"{name}_3"] - b[f"{name}_3"]).max() <= 1e-9, (copies, name)
            assert np.abs(a[f"{name}_4"] - b[f"{name}_4"]).max() <= 1e-8 * max(1.0, np.abs(b[f"{name}_4"]).max()), (copies, name)
            assert int(a[f"{name}_5"]) == int(b[f"{name}_5"]) and abs(float(a[f"{name}_6"]) - float(b[f"{name}_6"])) <= 1e-9 * abs(float(b[f"{name}_6"])), (copies, name)


def test_residual_list_that_ends_inside_a_landmark_batch():
    """The device keeps every appended batch of landmarks in its own spatial order (pba.hip: HostFrame::to_internal) under the invariant
    that the device's first n landmarks are the caller's first n wherever a residual list ends.  Here lists end INSIDE a batch (60 and
    then 130 of 200 landmarks), landmarks are appended in a second batch and the lists are completed afterwards: every step re-orders
    or extends the device arrays (splitBatchAt), and the finished window must solve exactly like the oracle's, with every getter
    answering in the caller's order."""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    win = syn.make_window(num_frames=4, num_points=4 * 260, width=320, height=240, seed=43)
    rng = np.random.default_rng(7)
    statuses = {}
    for a in win.frames:
        for b in win.frames:
            if a.frame_id != b.frame_id:
                st = np.zeros(len(a.uv), dtype=np.uint8)
                bad = rng.random(len(a.uv)) < 0.1
                st[bad] = rng.integers(1, 4, int(bad.sum()))
                statuses[(a.frame_id, b.frame_id)] = st
    o = _load(po.OracleWindow(po.default_pba_options()), win, statuses=statuses)
    g = capi.HipWindow(capi.default_pba_options())
    intr = win.scene.intrinsics
    first = 200  # landmarks of the first batch; the rest arrive later
    for f in win.frames:
        g.push_frame(f.frame_id, f.timestamp, f.pixelinfo, None, intr, syn.mat_to_params(f.T_w_c_init), f.exposure, f.affine_init, f.fixed, False)
        g.set_landmarks(f.frame_id, f.uv[:first], f.idepth_init[:first], f.patch[:first], np.zeros(first, dtype=np.uint8))
    cuts = (60, 130)
    for a in win.frames:
        for k, b in enumerate(x for x in win.frames if x.frame_id != a.frame_id):
            g.set_connection(a.frame_id, b.frame_id, statuses[(a.frame_id, b.frame_id)][:cuts[k % 2]])   # ends inside the batch
    for f in win.frames:
        n = len(f.uv)
        g.set_landmarks(f.frame_id, f.uv, f.idepth_init, f.patch, np.zeros(n, dtype=np.uint8))            # second batch appended
    for a in win.frames:
        for b in win.frames:
            if a.frame_id != b.frame_id:
                g.set_connection(a.frame_id, b.frame_id, statuses[(a.frame_id, b.frame_id)])                # lists completed
    # before anything runs: the residual statuses come back in the caller's order
    a, b = win.frames[1], win.frames[2]
    assert np.array_equal(g.get_residuals(a.frame_id, b.frame_id)["status"], statuses[(a.frame_id, b.frame_id)])
    _compare_solve(o, g, win)
    g.close()
