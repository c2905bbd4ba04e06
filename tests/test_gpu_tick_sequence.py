"""MonocularTracker::tick as a sequence (src/tracker/tracker/src/monocular_tracker.cpp:398-525), driven entirely through the
C-ABI over a synthetic 12-frame sequence, every stage checked against the CPU oracle fed with the same inputs:

  per frame   : pyramid -> initializationPoses -> estimatePose (coarse-to-fine against the device-resident depth maps of the
                last keyframe) -> calculateMeanSquareOpticalFlow -> DepthEstimation::estimate of every keyframe's immature set
  per keyframe: LandmarksActivator::activate -> the activated landmarks join the bundle adjustment -> pushFrame ->
                solve -> marginalisation of the oldest keyframe (window of 3) -> createReferenceDepthMaps (refill)

Feature extraction, the keyframe strategy and the marginalisation strategy are outside the hot path: candidate pixels are
random high-gradient pixels, every third frame is a keyframe, the oldest free keyframe leaves.  A parallel OracleWindow
receives exactly the same calls (sliding-window style), the per-frame stages are compared stage by stage."""
import copy

import numpy as np
import pytest

from dsopp_amd import synthetic as syn

pytestmark = pytest.mark.gpu

W, H, L = 320, 240, 3
N_BOOT, N_IMM, KF_EVERY, N_FRAMES, MAX_KF = 260, 220, 3, 12, 3


def _pick_pixels(rng, pixelinfo, n):
    grad = np.hypot(pixelinfo[..., 1], pixelinfo[..., 2])
    uv = np.zeros((0, 2))
    while len(uv) < n:
        cand = np.stack([rng.integers(8, W - 8, 4 * n), rng.integers(8, H - 8, 4 * n)], axis=1)
        uv = np.concatenate([uv, cand[grad[cand[:, 1], cand[:, 0]] > 4.0].astype(np.float64)])
    return uv[:n]


def _patch(plane, uv):
    ui, vi = uv[:, 0].astype(int), uv[:, 1].astype(int)
    return np.stack([plane[vi + int(oy), ui + int(ox)] for ox, oy in syn.PATTERN], axis=1)


class KF:
    """what track::ActiveKeyframe holds for one keyframe, plus the device objects"""

    def __init__(self, fid, ts, frame, capi, po, rng, intr):
        self.id, self.ts, self.frame = fid, ts, frame
        self.pyr = frame["pyr"]
        self.pix = frame["infos"][0]
        plane = frame["u8"].astype(np.float64)
        self.uv, self.idepth, self.patch = np.zeros((0, 2)), np.zeros(0), np.zeros((0, 8))
        uv = _pick_pixels(rng, self.pix, N_IMM)
        ui, vi = uv[:, 0].astype(int), uv[:, 1].astype(int)
        grad = np.stack([self.pix[vi, ui, 1], self.pix[vi, ui, 2]], axis=1)
        direction = np.stack([(uv[:, 0] - intr[2]) / intr[0], (uv[:, 1] - intr[3]) / intr[1], np.ones(len(uv))], axis=1)
        self.imm = po.new_immature_landmarks(uv, direction, _patch(plane, uv), grad)
        self.dset = capi.ImmatureSet(self.imm)


def test_tick_sequence():
    from dsopp_amd import capi
    from oracle import pyoracle as po
    rng = np.random.default_rng(7)
    scene = syn.Scene.make(W, H, 41)
    intr = scene.intrinsics
    frames = []
    for k in range(N_FRAMES):
        T = syn.se3_exp(0.45 * k * syn.BASE_MOTION)
        img, depth = scene.render(T)
        u8 = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        infos, _ = po.build_pyramid(u8, levels=L)
        pyr = capi.Pyramid(W, H, L)
        pyr.build(u8)
        frames.append(dict(k=k, ts=1000 * (k + 1), T_gt=T, u8=u8, depth=depth, infos=infos, pyr=pyr))
    g, o = capi.HipWindow(capi.default_pba_options()), po.OracleWindow(po.default_pba_options())
    aligner = capi.HipAligner(capi.default_align_options())
    alive, poses, stats = [], {}, dict(tracked=0, activated=0, solves=0, marginalised=0)

    def push_keyframe(kf, T_w, affine, fixed):
        for b in (g, o):
            if b is g:
                b.push_frame(kf.id, kf.ts, None, None, intr, T_w, 1.0, affine, fixed, False, pyramid=kf.pyr)
            else:
                b.push_frame(kf.id, kf.ts, kf.pix, None, intr, T_w, 1.0, affine, fixed, False)
            b.set_landmarks(kf.id, kf.uv, kf.idepth, kf.patch, np.zeros(len(kf.uv), dtype=np.uint8))
            for h in alive:
                b.set_connection(h.id, kf.id, np.zeros(len(h.uv), dtype=np.uint8))
                b.set_connection(kf.id, h.id, np.zeros(len(kf.uv), dtype=np.uint8))
        alive.append(kf)

    def solve_and_compare(tag):
        eg, itg, nvg = g.solve()
        eo, ito, nvo = o.solve()
        assert (itg, nvg) == (ito, nvo), tag
        assert abs(eg - eo) <= 1e-6 * abs(eo), tag
        for kf in alive:
            (Tg, abg), (To, abo) = g.get_pose(kf.id), o.get_pose(kf.id)
            assert np.abs(Tg - To).max() <= 1e-6 and np.abs(abg - abo).max() <= 1e-6, (tag, kf.id)
            lg, lo = g.get_landmarks(kf.id), o.get_landmarks(kf.id)
            if not len(lo["idepth"]):
                continue   # a fresh keyframe has no active landmarks yet
            assert np.abs(lg["idepth"] - lo["idepth"]).max() <= 1e-6 * max(1.0, np.abs(lo["idepth"]).max()), (tag, kf.id)
            assert np.array_equal(lg["flags"] & 3, lo["flags"] & 3), (tag, kf.id)
        stats["solves"] += 1

    def oracle_depth_maps():
        newest = alive[-1]
        sources = []
        for kf in alive[:-1]:
            lm = o.get_landmarks(kf.id)
            idepth = lm["idepth"].copy()
            skip = ((lm["flags"] & 3) != 0) | (idepth < 0)
            idepth[np.abs(idepth) < 1e-8] = 0
            sources.append(dict(T_w=o.get_pose(kf.id)[0], uv=kf.uv, idepth=idepth, variance=lm["inv_hdd"], skip=skip.astype(np.uint8),
                                status=o.get_residuals(kf.id, newest.id)["status"]))
        return po.create_reference_depth_maps(sources, o.get_pose(newest.id)[0], intr, W, H, L)

    # ---- bootstrap (the reference's initializer is outside the hot path): two keyframes with active landmarks near the truth
    for k in (0, KF_EVERY):
        f = frames[k]
        kf = KF(k, f["ts"], f, capi, po, rng, intr)
        kf.uv = _pick_pixels(rng, kf.pix, N_BOOT)
        ui, vi = kf.uv[:, 0].astype(int), kf.uv[:, 1].astype(int)
        kf.idepth = 1.0 / f["depth"][vi, ui] * (1 + rng.uniform(-2e-3, 2e-3, N_BOOT))
        kf.patch = _patch(f["u8"].astype(np.float64), kf.uv)
        T0 = f["T_gt"] if k == 0 else f["T_gt"] @ syn.se3_exp(np.concatenate([rng.normal(0, 5e-3, 3), rng.normal(0, 1e-3, 3)]))
        push_keyframe(kf, syn.mat_to_params(T0), np.zeros(2), k == 0)
    solve_and_compare("bootstrap")
    maps = g.create_reference_depth_maps(L)
    for k in range(KF_EVERY + 1):
        poses[k] = syn.mat_to_params(frames[k]["T_gt"])
    poses[KF_EVERY] = g.get_pose(KF_EVERY)[0]
    rmse_last = np.full(L, 1e10)
    affine_prev = np.zeros(2)
    min_distance = 2.0   # LandmarksActivator::min_distance_to_neighbor_, landmarks_activator.hpp:51

    for k in range(KF_EVERY + 1, N_FRAMES):
        f = frames[k]
        ref = alive[-1]
        T_ref, ab_ref = g.get_pose(ref.id)
        # -- initializationPoses
        hyp = capi.initialization_poses(poses[k - 2], poses[k - 1], T_ref)
        hyp_o = po.initialization_poses(poses[k - 2], poses[k - 1], T_ref)
        assert hyp.shape == (113, 7) and np.abs(np.abs(hyp) - np.abs(hyp_o)).max() <= 1e-12
        # -- estimatePose against the device-resident maps; the oracle steps the same chain from the same inputs by hand
        rl = rmse_last.copy()
        res = aligner.estimate_pose(ref.ts, T_ref, ref.pyr, maps, 1.0, ab_ref, f["ts"], f["pyr"], 1.0, intr, hyp, affine_prev, rl)
        assert res["success"] and res["tries"] == 1, k
        T_o, ab_o, its_o = hyp[0], affine_prev, 0
        for lvl in range(L - 1, -1, -1):
            ids, wgt = maps.get_level(lvl)
            u, v, idp, inten = po.points_from_depth_map(ref.frame["infos"][lvl], ids, wgt)
            r = po.align_solve(po.default_align_options(), u, v, idp, inten, intr / (1 << lvl), (W >> lvl, H >> lvl), T_ref, 1.0, ab_ref,
                               intr / (1 << lvl), f["infos"][lvl], None, T_o, 1.0, ab_o)
            T_o, ab_o = r["T_w_target"], r["affine_brightness"]
            its_o += r["iterations"]
        assert res["lm_iterations"] == its_o, k
        assert np.abs(res["T_w_target"] - T_o).max() <= 1e-6 and np.abs(res["affine_brightness"] - ab_o).max() <= 1e-5, k
        gt = syn.mat_to_params(f["T_gt"])
        assert np.abs(res["T_w_target"][4:] - gt[4:]).max() < 0.02 and np.abs(res["T_w_target"][:4] - gt[:4]).max() < 5e-3, k   # it tracks
        rmse_last[:] = rl
        poses[k], affine_prev = res["T_w_target"], res["affine_brightness"]
        stats["tracked"] += 1
        # -- calculateMeanSquareOpticalFlow (with and without rotation, monocular_tracker.cpp:474-479)
        T_t_r = np.linalg.inv(syn.params_to_mat(poses[k])) @ syn.params_to_mat(T_ref)
        T_nr = T_t_r.copy()
        T_nr[:3, :3] = np.eye(3)
        flow = maps.mean_square_optical_flow(0, intr, [syn.mat_to_params(T_t_r), syn.mat_to_params(T_nr)])
        ids, wgt = maps.get_level(0)
        for fl, T in zip(flow, (T_t_r, T_nr)):
            want = po.mean_square_optical_flow(ids, wgt, intr, syn.mat_to_params(T))
            assert abs(fl - want) <= 1e-12 * max(want, 1e-3), k
        # -- estimateDepths: every keyframe's immature landmarks against the new frame (monocular_tracker.cpp:74-102)
        for kf in alive:
            Tkf, abkf = g.get_pose(kf.id)
            T_new_kf = syn.mat_to_params(np.linalg.inv(syn.params_to_mat(poses[k])) @ syn.params_to_mat(Tkf))
            kf.dset.estimate(f["pyr"], 0, intr, T_new_kf, 1.0, abkf, 1.0, affine_prev)
            po.estimate_depths(kf.imm, f["infos"][0], None, intr, T_new_kf, 1.0, abkf, 1.0, affine_prev)
            st = kf.dset.download()
            assert np.array_equal(st["status"], kf.imm["status"]) and np.array_equal(st["traced"], kf.imm["traced"]), (k, kf.id)
            for key in ("idepth_min", "idepth_max", "search_pixel_interval"):
                assert np.abs(st[key] - kf.imm[key]).max() <= 1e-8 * max(1.0, np.abs(kf.imm[key]).max()), (k, kf.id, key)
            for key in ("idepth_min", "idepth_max", "uniqueness", "search_pixel_interval"):   # keep both sides on the same state
                kf.imm[key] = st[key]
        if k % KF_EVERY:
            continue
        # ================= new keyframe =================
        new = KF(k, f["ts"], f, capi, po, rng, intr)
        # -- LandmarksActivator::activate (before the keyframe enters the bundle adjustment, monocular_tracker.cpp:495)
        ofr = []
        for kf in alive:
            lm = g.get_landmarks(kf.id)
            Tkf, abkf = g.get_pose(kf.id)
            ofr.append(dict(pixelinfo=kf.pix, mask=None, T_w=Tkf, exposure=1.0, affine=abkf, active_uv=kf.uv, active_idepth=lm["idepth"],
                            active_skip=((lm["flags"] & 3) != 0).astype(np.uint8), immature=copy.deepcopy(kf.imm)))
        ofr.append(dict(pixelinfo=new.pix, mask=None, T_w=poses[k], exposure=1.0, affine=affine_prev))
        st_o, n_act_o, dist_o = po.activate_landmarks(ofr, intr, 20.0, 2 * N_BOOT, min_distance, refine=True)
        st_g, idp_g, ares = g.activate_landmarks([kf.id for kf in alive], [kf.dset for kf in alive], new.pyr, poses[k], 1.0, affine_prev,
                                                 2 * N_BOOT, min_distance, True)
        assert ares["number_of_active_points"] == n_act_o and abs(ares["min_distance_to_neighbor"] - dist_o) < 1e-12
        min_distance = ares["min_distance_to_neighbor"]
        for kf, a, b, idp, fo in zip(alive, st_o, st_g, idp_g, ofr):
            assert np.array_equal(a, b), (k, kf.id, np.flatnonzero(a != b))
            mid = 0.5 * fo["immature"]["idepth_min"] + 0.5 * fo["immature"]["idepth_max"]
            assert np.abs(mid - idp).max() <= 1e-9 * max(1.0, np.abs(mid).max())
            kf.imm = fo["immature"]                              # applyImmatureLandmarkActivationStatuses on the host copy
            kf.imm["idepth_min"], kf.imm["idepth_max"] = kf.dset.download()["idepth_min"], kf.dset.download()["idepth_max"]
            act = b == 0
            stats["activated"] += int(act.sum())
            if act.any():   # the activated landmarks become active landmarks of their keyframe: appended in both backends
                kf.uv = np.concatenate([kf.uv, kf.imm["projection"][act]])
                kf.idepth = np.concatenate([kf.idepth, idp[act]])
                kf.patch = np.concatenate([kf.patch, kf.imm["patch"][act]])
                for bk in (g, o):
                    cur = bk.get_landmarks(kf.id)
                    n_old = len(cur["idepth"])
                    old = {h.id: bk.get_residuals(kf.id, h.id)["status"] for h in alive if h is not kf}
                    bk.set_landmarks(kf.id, kf.uv, np.concatenate([cur["idepth"], idp[act]]), kf.patch,
                                     np.concatenate([cur["flags"] & 3, np.zeros(int(act.sum()), dtype=np.uint8)]))
                    for hid, st_old in old.items():
                        bk.set_connection(kf.id, hid, np.concatenate([st_old[:n_old], np.zeros(len(kf.uv) - n_old, dtype=np.uint8)]))
        # -- pushFrame + solve (refinePoses), then the oldest free keyframe leaves (window of MAX_KF)
        push_keyframe(new, poses[k], affine_prev, False)
        solve_and_compare(("keyframe", k))
        poses[k] = g.get_pose(new.id)[0]
        if len(alive) > MAX_KF:
            victim = alive[1]
            for bk in (g, o):
                for kf in alive:
                    cur = bk.get_landmarks(kf.id)
                    flags = cur["flags"] & 3       # what the keyframe reports: bit0 isMarginalized, bit1 isOutlier
                    if kf is victim:
                        flags = flags | 1          # every landmark of the leaving keyframe is marginalised
                    bk.set_landmarks(kf.id, kf.uv, cur["idepth"], kf.patch, flags.astype(np.uint8))
                bk.mark_frame_marginalized(victim.id)
            alive.remove(victim)
            victim.dset.close()
            stats["marginalised"] += 1
        # -- createReferenceDepthMaps into the tracker's map object
        g.refill_reference_depth_maps(maps)
        want = oracle_depth_maps()
        for lvl in range(L):
            ids, wgt = maps.get_level(lvl)
            assert np.array_equal(wgt > 0, want[lvl][1] > 0), (k, lvl)
            assert np.abs(ids - want[lvl][0]).max() <= 1e-5 * max(1.0, np.abs(want[lvl][0]).max()), (k, lvl)
        rmse_last = np.full(L, 1e10)   # new reference keyframe
    assert stats["tracked"] == N_FRAMES - KF_EVERY - 1 and stats["solves"] == 3 and stats["activated"] > 100 and stats["marginalised"] >= 1, stats


def test_native_sequence_driver_reproduces_the_python_driven_run():
    """dsopp_amd/host/tick_sequence.cpp — MonocularTracker::tick (monocular_tracker.cpp:425-525) in C++ over the host mirror — on an
    exported 60-frame sequence: same keyframes, same activations, poses equal to the Python-driven HIP run of scripts/tick_sequence.py
    (both make the same C-ABI calls on the same inputs; the native one adds the reference's updateFrame / updateLocalFrame traffic)."""
    import os
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    import tick_sequence
    out = tick_sequence.run(torch, syn, width=320, height=240, levels=3, frames=60, cpu_frames=0, n_boot=500, n_immature=700, desired_points=900,
                            max_keyframes=5, no_cpu=True, native=True)
    nat = out["native"]
    assert "error" not in nat, nat
    assert nat["frames"] == out["hip"]["frames"] and nat["keyframes"] == out["hip"]["keyframes"] and nat["keyframes"] >= 6
    assert nat["activated"] == out["hip"]["activated"] and nat["marginalised"] == out["hip"]["marginalised"] and nat["marginalised"] >= 1
    assert nat["frames_compared"] >= 55 and nat["pose_difference_to_the_python_driven_run_max"] <= 1e-9
