"""Row a21: createReferenceDepthMaps (src/tracker/tracker/src/create_depth_maps.cpp:18-147).
CPU: the oracle restatement against an independently written NumPy statement of the same definition (explicit 4x4
unproject -> transform -> project per landmark, reshape-sum pooling, loop dilation).
GPU: the device path (dsopp_hip_window_create_reference_depth_maps, atomics) against the oracle after a full solve, and the
device-side depth-map scan of the aligner against the host scan."""
import numpy as np
import pytest

from dsopp_amd import synthetic as syn


def _se3_matrix(p):
    x, y, z, w = p[:4]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = p[4:]
    return T


def _numpy_depth_maps(sources, T_w_newest, intr, W, H, levels):
    fx, fy, cx, cy = intr
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    ids, wgt = np.zeros((H, W)), np.zeros((H, W))
    Tn_inv = np.linalg.inv(_se3_matrix(T_w_newest))
    for s in sources:
        T = Tn_inv @ _se3_matrix(s["T_w"])
        for (u, v), rho, var, skip, st in zip(s["uv"], s["idepth"], s["variance"], s["skip"], s["status"]):
            if st != 0 or skip:
                continue
            if not (-1e-4 < rho < 1 / 0.001 + 10) or not (4 <= u <= W - 5 and 4 <= v <= H - 5):
                continue
            d = np.array([(u - cx) / fx, (v - cy) / fy, 1.0])   # bearing vector, z = 1
            X = T[:3, :3] @ d + rho * T[:3, 3]                   # point / depth in the newest frame
            if X[2] <= 0:
                continue
            p = K @ (X / X[2])
            if not (4 <= p[0] <= W - 5 and 4 <= p[1] <= H - 5):
                continue
            ix, iy = int(np.floor(p[0] + 0.5)), int(np.floor(p[1] + 0.5))
            w = np.sqrt(1e-3 / (var + 1e-12))
            ids[iy, ix] += rho / X[2] * w   # idepth in the newest frame = rho / z
            wgt[iy, ix] += w
    out = [(ids, wgt)]
    for _ in range(1, levels):
        a, b = out[-1]
        h2, w2 = a.shape[0] // 2, a.shape[1] // 2
        pool = lambda m: m[:2 * h2, :2 * w2].reshape(h2, 2, w2, 2).sum(axis=(1, 3))
        out.append((pool(a), pool(b)))
    res = []
    for lvl, (a, b) in enumerate(out):
        a2, b2 = a.copy(), b.copy()
        offs = [(1, 0), (-1, 0), (0, 1), (0, -1)] if lvl > 1 else [(1, 1), (-1, -1), (1, -1), (-1, 1)]
        for y in range(1, a.shape[0] - 1):
            for x in range(1, a.shape[1] - 1):
                if b[y, x] > 0:
                    continue
                nb = [(a[y + oy, x + ox], b[y + oy, x + ox]) for ox, oy in offs if b[y + oy, x + ox] > 0]
                if nb:
                    a2[y, x] = sum(n[0] for n in nb) / len(nb)
                    b2[y, x] = sum(n[1] for n in nb) / len(nb)
        res.append((a2, b2))
    return res


def _sources_from_window(win_obj, win, rng=None):
    """what createReferenceDepthMaps reads from the keyframes, taken from a solver window (oracle or HIP: same getters)"""
    newest = win.frames[-1]
    sources = []
    for f in win.frames[:-1]:
        lm = win_obj.get_landmarks(f.frame_id)
        T, _ = win_obj.get_pose(f.frame_id)
        idepth = lm["idepth"].copy()
        skip = ((lm["flags"] & 3) != 0) | (idepth < 0)          # isOutlier | isMarginalized | (idepth < 0 -> outlier, updateFrame)
        idepth[np.abs(idepth) < 1e-8] = 0
        sources.append(dict(T_w=T, uv=f.uv, idepth=idepth, variance=lm["inv_hdd"], skip=skip.astype(np.uint8),
                            status=win_obj.get_residuals(f.frame_id, newest.frame_id)["status"]))
    T_newest, _ = win_obj.get_pose(newest.frame_id)
    return sources, T_newest


def test_oracle_matches_numpy_statement():
    from oracle import pyoracle as po
    win = syn.make_window(num_frames=4, num_points=400, width=160, height=120, seed=11)
    rng = np.random.default_rng(2)
    sources = []
    for f in win.frames[:-1]:
        n = len(f.uv)
        sources.append(dict(T_w=syn.mat_to_params(f.T_w_c_gt), uv=f.uv, idepth=f.idepth_gt * (1 + rng.uniform(-0.01, 0.01, n)),
                            variance=rng.uniform(1e-7, 1e-3, n), skip=(rng.random(n) < 0.1).astype(np.uint8),
                            status=(rng.random(n) < 0.15).astype(np.uint8) * rng.integers(1, 4, n).astype(np.uint8)))
    T_newest = syn.mat_to_params(win.frames[-1].T_w_c_gt)
    got = po.create_reference_depth_maps(sources, T_newest, win.scene.intrinsics, 160, 120, 4)
    want = _numpy_depth_maps(sources, T_newest, win.scene.intrinsics, 160, 120, 4)
    assert sum((w > 0).sum() for _, w in got) > 500
    for lvl, ((a, b), (c, d)) in enumerate(zip(got, want)):
        assert a.shape == c.shape
        assert np.array_equal(b > 0, d > 0), lvl
        assert np.abs(b - d).max() <= 1e-12 * max(1.0, np.abs(d).max()), lvl
        assert np.abs(a - c).max() <= 1e-9 * max(1.0, np.abs(c).max()), lvl   # x/z via the fused 3x4 product vs explicit matrices


def _numpy_flow(ids, wgt, intr, T):
    """calculateMeanSquareOpticalFlow (monocular_tracker.cpp:104-134) with explicit bearing vectors"""
    fx, fy, cx, cy = intr
    H, W = ids.shape
    tot, n = 0.0, 0
    for y in range(4, H - 4):
        for x in range(4, W - 4):
            if wgt[y, x] <= 0:
                continue
            rho = ids[y, x] / wgt[y, x]
            if rho < 1e-6 or not rho < 1010:
                continue
            d = np.array([(x - cx) / fx, (y - cy) / fy, 1.0])
            X = T[:3, :3] @ d + rho * T[:3, 3]
            if X[2] <= 0:
                continue
            q = X / X[2]                                     # bearing of the reprojection, z = 1
            u, v = fx * q[0] + cx, fy * q[1] + cy
            if not (4 <= u <= W - 5 and 4 <= v <= H - 5):
                continue
            tot += np.sum((d[:2] - q[:2]) ** 2)
            n += 1
    return np.sqrt(tot / n), n


def _flow_case():
    from oracle import pyoracle as po
    win = syn.make_window(num_frames=4, num_points=900, width=160, height=120, seed=17)
    sources = [dict(T_w=syn.mat_to_params(f.T_w_c_gt), uv=f.uv, idepth=f.idepth_gt, variance=np.full(len(f.uv), 1e-5),
                    skip=np.zeros(len(f.uv), dtype=np.uint8), status=np.zeros(len(f.uv), dtype=np.uint8)) for f in win.frames[:-1]]
    maps = po.create_reference_depth_maps(sources, syn.mat_to_params(win.frames[-1].T_w_c_gt), win.scene.intrinsics, 160, 120, 2)
    Ts = [syn.se3_exp(np.array([0.05, -0.02, 0.03, 0.01, -0.02, 0.005])), syn.se3_exp(np.array([0.05, -0.02, 0.03, 0, 0, 0])),
          np.eye(4), syn.se3_exp(np.array([0.3, 0.1, -0.2, 0.05, 0.08, -0.03]))]
    return win, maps, Ts


def test_oracle_optical_flow_matches_numpy_statement():
    from oracle import pyoracle as po
    win, maps, Ts = _flow_case()
    for lvl in (0, 1):
        ids, wgt = maps[lvl]
        intr = win.scene.intrinsics / (1 << lvl)
        for k, T in enumerate(Ts):
            want, n = _numpy_flow(ids, wgt, intr, T)
            got = po.mean_square_optical_flow(ids, wgt, intr, syn.mat_to_params(T))
            assert n > 100
            assert abs(got - want) <= 1e-10 * max(want, 1e-3), (lvl, k, got, want)
            if k == 2:
                assert got < 1e-12     # identity: no flow
    assert np.isnan(po.mean_square_optical_flow(np.zeros((120, 160)), np.zeros((120, 160)), win.scene.intrinsics, syn.mat_to_params(np.eye(4))))


@pytest.mark.gpu
def test_gpu_optical_flow_matches_oracle():
    from dsopp_amd import capi
    from oracle import pyoracle as po
    W, H, L = 320, 240, 2
    win = syn.make_window(num_frames=4, num_points=1200, width=W, height=H, seed=13)
    g = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
    g.solve()
    maps = g.create_reference_depth_maps(L)
    _, _, Ts = _flow_case()
    for lvl in range(L):
        ids, wgt = maps.get_level(lvl)
        intr = win.scene.intrinsics / (1 << lvl)
        want = np.array([po.mean_square_optical_flow(ids, wgt, intr, syn.mat_to_params(T)) for T in Ts])
        got = maps.mean_square_optical_flow(lvl, intr, [syn.mat_to_params(T) for T in Ts])
        assert np.all(np.abs(got - want) <= 1e-12 * np.maximum(want, 1e-3)), (lvl, got, want)
        one = maps.mean_square_optical_flow(lvl, intr, [syn.mat_to_params(Ts[0])])
        assert one[0] == got[0]    # deterministic reduction, independent of the batch
    with pytest.raises(capi.HipError):
        maps.mean_square_optical_flow(0, win.scene.intrinsics, [syn.mat_to_params(np.eye(4))] * 5)
    # refilling the same object (what the tracker does after every keyframe) reproduces a fresh creation bit for bit
    before = [maps.get_level(l) for l in range(L)]
    g.refill_reference_depth_maps(maps)
    for l in range(L):
        a, b = maps.get_level(l)
        assert np.array_equal(a > 0, before[l][0] > 0) and np.abs(a - before[l][0]).max() <= 1e-12 * np.abs(a).max()   # (atomics order)
        assert np.abs(b - before[l][1]).max() <= 1e-12 * np.abs(b).max()
    maps.close()
    g.close()


@pytest.mark.gpu
def test_gpu_depth_maps_and_device_scan():
    from dsopp_amd import capi
    from oracle import pyoracle as po
    W, H, L = 320, 240, 4
    win = syn.make_window(num_frames=4, num_points=600, width=W, height=H, seed=13)
    o = syn.load_window(po.OracleWindow(po.default_pba_options()), win)
    g = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
    o.solve()
    g.solve()
    sources, T_newest = _sources_from_window(o, win)
    want = po.create_reference_depth_maps(sources, T_newest, win.scene.intrinsics, W, H, L)
    maps = g.create_reference_depth_maps(L)
    for lvl in range(L):
        ids, wgt = maps.get_level(lvl)
        wi, ww = want[lvl]
        assert ids.shape == wi.shape
        assert np.array_equal(wgt > 0, ww > 0), lvl
        assert np.abs(wgt - ww).max() <= 1e-6 * np.abs(ww).max(), lvl   # weights carry H_dd^-1 of the GPU / CPU solves (1e-9 .. 1e-7 apart)
        assert np.abs(ids - wi).max() <= 1e-6 * np.abs(wi).max(), lvl
    assert (want[0][1] > 0).sum() > 300
    # tracker side: device scan of a level == host scan of the same (downloaded) level
    newest = win.frames[-1]
    pyr = capi.Pyramid(W, H, L)
    pyr.build(newest.image_u8)
    tgt = capi.Pyramid(W, H, L)
    tgt.build(win.frames[-2].image_u8)
    T_ref, _ = g.get_pose(newest.frame_id)
    T_init = syn.mat_to_params(win.frames[-2].T_w_c_init)
    # the optical-flow measure while no reference points exist for these maps: the dense pass over the planes
    _, _, Ts = _flow_case()
    flow_dense = {lvl: maps.mean_square_optical_flow(lvl, win.scene.intrinsics / (1 << lvl), [syn.mat_to_params(T) for T in Ts]) for lvl in (2, 0)}
    for lvl in (2, 0):
        intr = win.scene.intrinsics / (1 << lvl)
        ids, wgt = maps.get_level(lvl)
        res = []
        for device in (True, False):
            a = capi.HipAligner(capi.default_align_options())
            a.reset()
            if device:
                a.push_reference_depth_maps(5000, T_ref, pyr, lvl, intr, maps, 1.0, np.zeros(2))
            else:
                a.push_reference_depth_map(5000, T_ref, pyr, lvl, intr, ids, wgt, 1.0, np.zeros(2))
            n = a.num_points()
            a.push_target(6000, T_init, tgt, lvl, intr, 1.0, np.zeros(2))
            res.append((n, a.solve()))
            a.close()
        (n_dev, r_dev), (n_host, r_host) = res
        assert n_dev == n_host and n_dev > 50
        assert r_dev["iterations"] == r_host["iterations"] and r_dev["n_valid"] == r_host["n_valid"]
        assert r_dev["energy"] == r_host["energy"] and np.array_equal(r_dev["T_w_target"], r_host["T_w_target"])
        # ... and once the tracker has extracted the level's reference points the measure walks that list (the same pixels, one launch):
        # equal to the dense pass up to summation order, and to the CPU checker
        flow_points = maps.mean_square_optical_flow(lvl, intr, [syn.mat_to_params(T) for T in Ts])
        want = np.array([po.mean_square_optical_flow(ids, wgt, intr, syn.mat_to_params(T)) for T in Ts])
        assert np.all(np.abs(flow_points - flow_dense[lvl]) <= 1e-12 * np.maximum(flow_dense[lvl], 1e-3)), (lvl, flow_points, flow_dense[lvl])
        assert np.all(np.abs(flow_points - want) <= 1e-12 * np.maximum(want, 1e-3)), (lvl, flow_points, want)
        assert maps.mean_square_optical_flow(lvl, intr, [syn.mat_to_params(Ts[0])])[0] == flow_points[0]
    # a refill makes the lists stale: the measure falls back to the dense pass on the new planes
    g.refill_reference_depth_maps(maps)
    again = maps.mean_square_optical_flow(0, win.scene.intrinsics, [syn.mat_to_params(T) for T in Ts])
    assert np.all(np.abs(again - flow_dense[0]) <= 1e-10 * np.maximum(flow_dense[0], 1e-3))
    for obj in (maps, pyr, tgt, g):
        obj.close()


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(320, 240, 4), (1280, 1024, 4), (1280, 1024, 5)])
def test_gpu_estimate_pose_matches_oracle_chain(size):
    """the tracker's coarse-to-fine estimatePose (monocular_tracker.cpp:179-245) in one C call against the same chain
    stepped by hand through the oracle: window solve -> reference depth maps -> per level {scan, align} — also at the full C2 size
    of BASELINE.json (1280 x 1024) with the 4 pyramid levels the production Camera requests (src/sensors/camera/src/camera.cpp:43-45)
    and with the 5 that kMaxPyramidDepth allows (src/features/include/features/camera/pixel_data_frame.hpp:26)"""
    from dsopp_amd import capi
    from oracle import pyoracle as po
    W, H, L = size
    win = syn.make_window(num_frames=4, num_points=800 if W == 320 else 2000, width=W, height=H, seed=17)
    o = syn.load_window(po.OracleWindow(po.default_pba_options()), win)
    g = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
    o.solve()
    g.solve()
    newest, target = win.frames[-1], win.frames[-2]
    intr = win.scene.intrinsics
    T_init = syn.mat_to_params(target.T_w_c_init)
    # oracle chain
    sources, T_ref_o = _sources_from_window(o, win)
    maps_o = po.create_reference_depth_maps(sources, T_ref_o, intr, W, H, L)
    infos_ref, _ = po.build_pyramid(newest.image_u8, levels=L)
    infos_tgt, _ = po.build_pyramid(target.image_u8, levels=L)
    _, ab_ref_o = o.get_pose(newest.frame_id)
    T, ab, its_o, rmse_o = T_init, np.zeros(2), 0, []
    for lvl in range(L - 1, -1, -1):
        u, v, idp, inten = po.points_from_depth_map(infos_ref[lvl], *maps_o[lvl])
        h, w = infos_ref[lvl].shape[:2]
        r = po.align_solve(po.default_align_options(), u, v, idp, inten, intr / (1 << lvl), (w, h), T_ref_o, 1.0, ab_ref_o, intr / (1 << lvl),
                           infos_tgt[lvl], None, T, 1.0, ab)
        T, ab = r["T_w_target"], r["affine_brightness"]
        its_o += r["iterations"]
        rmse_o.append(r["rmse"])
    # HIP: one call
    maps_g = g.create_reference_depth_maps(L)
    pr, pt = capi.Pyramid(W, H, L), capi.Pyramid(W, H, L)
    pr.build(newest.image_u8)
    pt.build(target.image_u8)
    T_ref_g, ab_ref_g = g.get_pose(newest.frame_id)
    a = capi.HipAligner(capi.default_align_options())
    rmse_last = np.full(L, 1e10)
    res = a.estimate_pose(newest.timestamp, T_ref_g, pr, maps_g, 1.0, ab_ref_g, newest.timestamp + 1, pt, 1.0, intr, T_init[None, :], np.zeros(2),
                          rmse_last)
    assert res["success"] and res["tries"] == 1
    assert res["lm_iterations"] == its_o
    assert np.abs(res["T_w_target"] - T).max() <= 1e-6, np.abs(res["T_w_target"] - T).max()
    assert np.abs(res["affine_brightness"] - ab).max() <= 1e-5
    assert np.allclose(rmse_last[::-1], rmse_o, rtol=1e-6)   # per-level rmse (rmse_last is indexed by level, the chain ran coarse -> fine)
    # it tracks: the rotation ends closer to the ground truth than the initialisation (the translation is only defined up to the
    # window's monocular scale gauge, which a bundle-adjusted window fixed at frame 0 alone is free to move by a per cent or two)
    gt = syn.mat_to_params(target.T_w_c_gt)
    assert np.abs(res["T_w_target"][:4] - gt[:4]).max() < 0.5 * np.abs(T_init[:4] - gt[:4]).max()
    # a hopeless rmse bound rejects every initialisation: result of the first one is returned, bounds are relaxed by 2.5
    rl = np.full(L, 1e-9)
    res2 = a.estimate_pose(newest.timestamp, T_ref_g, pr, maps_g, 1.0, ab_ref_g, newest.timestamp + 1, pt, 1.0, intr,
                           np.stack([T_init, T_init]), np.zeros(2), rl)
    assert not res2["success"] and res2["tries"] == 2 and np.allclose(rl, 2.5e-9)
    for obj in (a, maps_g, pr, pt, g):
        obj.close()


@pytest.mark.gpu
def test_tracker_full_size_identity_property():
    """C2 of BASELINE.json at full size (1280x1024, 5 pyramid levels), checked through a size-independent property: tracking
    the newest keyframe's OWN image against its reference depth maps from a perturbed initialisation must return the
    keyframe's own pose (photometric error zero at the identity), whatever the scene."""
    from dsopp_amd import capi
    W, H, L = 1280, 1024, 5
    win = syn.make_window(num_frames=4, num_points=1200, width=W, height=H, seed=23)
    g = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
    g.solve()
    maps = g.create_reference_depth_maps(L)
    kf = win.frames[-1]
    pyr = capi.Pyramid(W, H, L)
    pyr.build(kf.image_u8)
    T_ref, ab_ref = g.get_pose(kf.frame_id)
    Tm = np.eye(4)
    Tm[:3, 3] = [0.01, -0.006, 0.004]
    T_init = syn.mat_to_params(syn.params_to_mat(T_ref) @ Tm) if hasattr(syn, "params_to_mat") else T_ref + np.array([0, 0, 0, 0, 0.01, -0.006, 0.004])
    a = capi.HipAligner(capi.default_align_options())
    rmse_last = np.full(L, 1e10)
    res = a.estimate_pose(kf.timestamp, T_ref, pyr, maps, 1.0, ab_ref, kf.timestamp + 1, pyr, 1.0, win.scene.intrinsics, T_init[None, :], ab_ref, rmse_last)
    assert res["success"] and res["lm_iterations"] > 5
    assert np.abs(res["T_w_target"] - T_ref).max() < 2e-4, np.abs(res["T_w_target"] - T_ref).max()
    assert rmse_last[0] < 1.0   # grey levels: the level-0 residual of a frame against itself
    for obj in (a, maps, pyr, g):
        obj.close()


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(320, 240, 4), (1280, 1024, 5)])
def test_estimate_pose_persistent_launch_equals_launch_per_iteration(size):
    """estimatePose as ONE persistent launch over all pyramid levels (lm_path 0: resident workgroups meet at a device-scope
    counter once per LM iteration) against the launch-per-iteration path (lm_path 1): same state machine and arithmetic, so the
    iteration counts and per-level rmse must be identical and the pose equal to rounding — at the tracker's full C2 size too.
    Several hypotheses: the first one is hopeless (fails a level on the device), the second succeeds."""
    from dsopp_amd import capi
    W, H, L = size
    win = syn.make_window(num_frames=4, num_points=1200, width=W, height=H, seed=29)
    g = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
    g.solve()
    maps = g.create_reference_depth_maps(L)
    newest, target = win.frames[-1], win.frames[-2]
    pr, pt = capi.Pyramid(W, H, L), capi.Pyramid(W, H, L)
    pr.build(newest.image_u8)
    pt.build(target.image_u8)
    T_ref, ab_ref = g.get_pose(newest.frame_id)
    T_good = syn.mat_to_params(target.T_w_c_init)
    T_bad = syn.mat_to_params(target.T_w_c_gt @ syn.se3_exp(np.array([1.5, -1.0, 0.8, 0.3, -0.4, 0.25])))
    out = []
    for path in (0, 1):
        a = capi.HipAligner(capi.default_align_options())
        a.set_lm_path(path)
        frames = []
        rmse_last = np.full(L, 1e10)
        # frame 1 sets the per-level energy bounds, frame 2 runs against them: [bad, good] -> the bad hypothesis must be rejected
        frames.append(a.estimate_pose(newest.timestamp, T_ref, pr, maps, 1.0, ab_ref, newest.timestamp + 1, pt, 1.0, win.scene.intrinsics,
                                      T_good[None, :], np.zeros(2), rmse_last))
        frames.append(dict(rmse=rmse_last.copy()))
        frames.append(a.estimate_pose(newest.timestamp, T_ref, pr, maps, 1.0, ab_ref, newest.timestamp + 2, pt, 1.0, win.scene.intrinsics,
                                      np.stack([T_bad, T_good]), np.zeros(2), rmse_last))
        frames.append(dict(rmse=rmse_last.copy()))
        out.append(frames)
        a.close()
    (f0, r0, f1, r1), (h0, s0, h1, s1) = out
    assert f0["success"] and h0["success"] and f0["tries"] == h0["tries"] == 1
    assert f1["success"] and h1["success"] and f1["tries"] == h1["tries"] == 2, (f1["tries"], h1["tries"])
    for fa, fb in ((f0, h0), (f1, h1)):
        assert fa["lm_iterations"] == fb["lm_iterations"], (fa["lm_iterations"], fb["lm_iterations"])
        assert np.abs(fa["T_w_target"] - fb["T_w_target"]).max() <= 1e-12
        assert np.abs(fa["affine_brightness"] - fb["affine_brightness"]).max() <= 1e-10
    assert np.allclose(r0["rmse"], s0["rmse"], rtol=1e-12) and np.allclose(r1["rmse"], s1["rmse"], rtol=1e-12)
    for obj in (maps, pr, pt, g):
        obj.close()


@pytest.mark.gpu
def test_persistent_launches_with_changing_participant_counts():
    """The exchange buffers of the persistent alignment launch stay armed from launch to launch (the pass counter continues, a
    workgroup re-arms by rotation; the host fills them once).  The number of participating workgroups follows the depth maps' point
    counts, so ONE aligner is driven alternately against a sparse and a dense window's maps — few, many, few, many participants,
    failed hypotheses in between — and every call must reproduce the launch-per-iteration path of a fresh aligner."""
    from dsopp_amd import capi
    W, H, L = 640, 480, 4
    scenes = []
    for n_points, seed in ((160, 31), (6000, 32)):
        win = syn.make_window(num_frames=4, num_points=n_points, width=W, height=H, seed=seed)
        g = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
        g.solve()
        maps = g.create_reference_depth_maps(L)
        newest, target = win.frames[-1], win.frames[-2]
        pr, pt = capi.Pyramid(W, H, L), capi.Pyramid(W, H, L)
        pr.build(newest.image_u8)
        pt.build(target.image_u8)
        T_ref, ab_ref = g.get_pose(newest.frame_id)
        T_good = syn.mat_to_params(target.T_w_c_init)
        T_bad = syn.mat_to_params(target.T_w_c_gt @ syn.se3_exp(np.array([1.5, -1.0, 0.8, 0.3, -0.4, 0.25])))
        scenes.append(dict(win=win, g=g, maps=maps, pr=pr, pt=pt, T_ref=T_ref, ab_ref=ab_ref, newest=newest, hyp=np.stack([T_bad, T_good])))

    def call(a, sc, k):
        rmse_last = np.full(L, 1e10)
        r = a.estimate_pose(sc["newest"].timestamp, sc["T_ref"], sc["pr"], sc["maps"], 1.0, sc["ab_ref"], sc["newest"].timestamp + 1 + k, sc["pt"], 1.0,
                            sc["win"].scene.intrinsics, sc["hyp"][1:] if k % 2 == 0 else sc["hyp"], np.zeros(2), rmse_last)
        return r, rmse_last

    persistent = capi.HipAligner(capi.default_align_options())
    persistent.set_lm_path(0)
    order = [0, 1, 0, 0, 1, 1, 0, 1]
    for k, which in enumerate(order):
        sc = scenes[which]
        ra, rmse_a = call(persistent, sc, k)
        fresh = capi.HipAligner(capi.default_align_options())
        fresh.set_lm_path(1)
        rb, rmse_b = call(fresh, sc, k)
        fresh.close()
        assert ra["success"] == rb["success"] and ra["tries"] == rb["tries"] and ra["lm_iterations"] == rb["lm_iterations"], (k, which, ra, rb)
        assert np.abs(ra["T_w_target"] - rb["T_w_target"]).max() <= 1e-12, (k, which)
        assert np.allclose(rmse_a, rmse_b, rtol=1e-12), (k, which)
    persistent.close()
    for sc in scenes:
        for obj in (sc["maps"], sc["pr"], sc["pt"], sc["g"]):
            obj.close()
