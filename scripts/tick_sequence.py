"""BASELINE.json's second metric — frame-tracking ms/frame — over a tracked SEQUENCE: MonocularTracker::tick as the reference runs it
(src/tracker/tracker/src/monocular_tracker.cpp:425-525), driven through the C-ABI only.

  per frame   : image -> pyramid -> initializationPoses -> estimatePose (coarse-to-fine against the device-resident reference depth maps)
                -> calculateMeanSquareOpticalFlow (with / without rotation) -> DepthEstimation::estimate of every keyframe's immature set
                -> keyframe decision (the reference's flow rule, mean_square_optical_flow_and_rmse_keyframe_strategy.cpp:14-48)
  per keyframe: LandmarksActivator::activate -> activated landmarks join the bundle adjustment -> pushFrame -> solve (refinePoses)
                -> the oldest free keyframe is marginalised once the window holds more than `max_keyframes` -> createReferenceDepthMaps

Outside the hot path and outside the per-frame time: rendering of the synthetic images, the choice of candidate pixels
(random high-gradient pixels stand in for the feature extractor), the marginalisation strategy (oldest free keyframe).

The same driver runs the CPU port (oracle/, restatement of the reference's CPU path) on a prefix of the same frames; both are compared
with the synthetic ground truth (translation / rotation drift).  Used by bench.py (`tick_sequence`), runnable on its own:
    python scripts/tick_sequence.py [--width 640 --height 480 --frames 200 --cpu-frames 30]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def render_frames(torch, scene, poses, device="cuda"):
    """synthetic.Scene.render on the GPU (torch is plumbing here: the data source, not the product): list of (u8 image, depth) on the host"""
    W, H = scene.width, scene.height
    dt = torch.float64
    uu, vv = torch.meshgrid(torch.arange(W, dtype=dt, device=device), torch.arange(H, dtype=dt, device=device), indexing="xy")
    rx, ry = (uu - scene.cx) / scene.fx, (vv - scene.cy) / scene.fy
    c = lambda a: torch.tensor(np.asarray(a), dtype=dt, device=device)
    d_amp, d_fu, d_fv, d_ph = c(scene.depth_amp), c(scene.depth_fu), c(scene.depth_fv), c(scene.depth_ph)
    t_amp, t_ku, t_kv, t_ph = c(scene.tex_amp), c(scene.tex_ku), c(scene.tex_kv), c(scene.tex_ph)

    def depth0(u0, v0):
        z = torch.full_like(u0, 6.0)
        for i in range(len(d_amp)):
            z = z + d_amp[i] * torch.cos(2 * np.pi * (d_fu[i] * u0 / W + d_fv[i] * v0 / H) + d_ph[i])
        return z

    out = []
    for T in poses:
        R, t = T[:3, :3], T[:3, 3]
        d = torch.full((H, W), 6.0, dtype=dt, device=device)

        def world(d):
            X = R[0, 0] * rx * d + R[0, 1] * ry * d + R[0, 2] * d + t[0]
            Y = R[1, 0] * rx * d + R[1, 1] * ry * d + R[1, 2] * d + t[1]
            Z = R[2, 0] * rx * d + R[2, 1] * ry * d + R[2, 2] * d + t[2]
            return scene.fx * X / Z + scene.cx, scene.fy * Y / Z + scene.cy, Z

        for _ in range(12):
            u0, v0, Z = world(d)
            d = d * depth0(u0, v0) / Z
        u0, v0, _ = world(d)
        img = torch.full_like(u0, 127.5)
        for i in range(len(t_amp)):
            img = img + t_amp[i] * torch.cos(2 * np.pi * (t_ku[i] * u0 + t_kv[i] * v0) + t_ph[i])
        out.append((torch.clamp(torch.round(img), 0, 255).to(torch.uint8).cpu().numpy(), d.cpu().numpy()))
    return out


def _pick_pixels(rng, pixelinfo, n):
    H, W = pixelinfo.shape[:2]
    grad = np.hypot(pixelinfo[..., 1], pixelinfo[..., 2])
    uv = np.zeros((0, 2))
    while len(uv) < n:
        cand = np.stack([rng.integers(8, W - 8, 4 * n), rng.integers(8, H - 8, 4 * n)], axis=1)
        uv = np.concatenate([uv, cand[grad[cand[:, 1], cand[:, 0]] > 4.0].astype(np.float64)])
    return uv[:n]


def _patch(plane, uv, pattern):
    ui, vi = uv[:, 0].astype(int), uv[:, 1].astype(int)
    return np.stack([plane[vi + int(oy), ui + int(ox)] for ox, oy in pattern], axis=1)


def _pixelinfo0(u8):
    """level-0 (I, Ix, Iy) on the host: what the candidate selection looks at (central differences, as the pyramid builds them)"""
    p = u8.astype(np.float64)
    info = np.zeros(p.shape + (3,))
    info[..., 0] = p
    info[1:-1, 1:-1, 1] = 0.5 * (p[1:-1, 2:] - p[1:-1, :-2])
    info[1:-1, 1:-1, 2] = 0.5 * (p[2:, 1:-1] - p[:-2, 1:-1])
    return info


def candidate_pixels(seed, k, u8, n):
    """the stand-in for the feature extractor: n random high-gradient pixels of frame k.  A generator of its own per frame, so that every
    driver of the sequence (this one on either backend, dsopp_amd/host/tick_sequence.cpp through the exported file) sees the same
    candidates whichever frames it turns into keyframes"""
    return _pick_pixels(np.random.default_rng([seed, 7, k]), _pixelinfo0(u8), n)


def bootstrap_keyframe(seed, k, u8, depth, pose_gt, n_boot, syn):
    """active landmarks near the truth + a slightly perturbed pose for one of the two bootstrap keyframes (the reference's initializer is
    outside the hot path)"""
    rng = np.random.default_rng([seed, 11, k])
    uv = _pick_pixels(rng, _pixelinfo0(u8), n_boot)
    ui, vi = uv[:, 0].astype(int), uv[:, 1].astype(int)
    idepth = 1.0 / depth[vi, ui] * (1 + rng.uniform(-2e-3, 2e-3, n_boot))
    T0 = pose_gt if k == 0 else pose_gt @ syn.se3_exp(np.concatenate([rng.normal(0, 5e-3, 3), rng.normal(0, 1e-3, 3)]))
    return uv, idepth, syn.mat_to_params(T0)


def export_sequence(path, frames_u8, depths, poses_gt, scene, syn, *, levels, n_boot, n_immature, desired_points, max_keyframes, kf_factor, seed=7,
                    first_kf_gap=3):
    """the sequence as dsopp_amd/host/tick_sequence.cpp reads it (little endian): "DSOPTICK" | 9 x int32 (W, H, levels, frames, n_boot,
    n_immature, desired_points, max_keyframes, first_kf_gap) | kf_factor f64 | intrinsics 4 x f64 | frames x (H x W) u8 | frames x pose
    7 x f64 | 2 x (boot uv, boot idepth, boot pose) | frames x (n_immature x 2) f64 candidate pixels"""
    n = len(frames_u8)
    with open(path, "wb") as fh:
        fh.write(b"DSOPTICK")
        fh.write(np.array([scene.width, scene.height, levels, n, n_boot, n_immature, desired_points, max_keyframes, first_kf_gap], dtype="<i4").tobytes())
        fh.write(np.array([kf_factor], dtype="<f8").tobytes())
        fh.write(np.asarray(scene.intrinsics, dtype="<f8").tobytes())
        for im in frames_u8:
            fh.write(np.ascontiguousarray(im, dtype=np.uint8).tobytes())
        for T in poses_gt:
            fh.write(np.asarray(syn.mat_to_params(T), dtype="<f8").tobytes())
        for k in (0, first_kf_gap):
            uv, idepth, T0 = bootstrap_keyframe(seed, k, frames_u8[k], depths[k], poses_gt[k], n_boot, syn)
            fh.write(np.ascontiguousarray(uv, dtype="<f8").tobytes() + np.ascontiguousarray(idepth, dtype="<f8").tobytes() + np.asarray(T0, dtype="<f8").tobytes())
        for k in range(n):
            fh.write(np.ascontiguousarray(candidate_pixels(seed, k, frames_u8[k], n_immature), dtype="<f8").tobytes())


def run_native(path, exe=None, timeout=600):
    """builds (g++, seconds) and runs the C++ driver on an exported sequence: (its JSON line, {frame: pose})"""
    import subprocess
    import tempfile
    lib = os.path.join(ROOT, "dsopp_amd", "lib")
    tmp = tempfile.mkdtemp(prefix="dsopp_tick_")
    exe = exe or os.path.join(tmp, "tick_sequence")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "dsopp_amd", "host", "tick_sequence.cpp"), f"-L{lib}", "-ldsopp_hip",
                               f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", exe])
    poses_path = os.path.join(tmp, "poses.txt")
    r = subprocess.run([exe, path, poses_path], capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"native tick sequence failed ({r.returncode}): {r.stdout[-500:]} {r.stderr[-1500:]}")
    if os.environ.get("DSOPP_HIP_HOST_TIMES"):   # the library's host-time table of the NATIVE process (tuning aid)
        print("\n".join("[native] " + ln for ln in r.stderr.splitlines() if "[host times]" in ln), file=sys.stderr)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    poses = {}
    for ln in open(poses_path):
        v = ln.split()
        poses[int(v[0])] = np.array([float(x) for x in v[1:]])
    return json.loads(line), poses


class _Keyframe:
    def __init__(self, fid, ts):
        self.id, self.ts = fid, ts
        self.uv, self.idepth, self.patch = np.zeros((0, 2)), np.zeros(0), np.zeros((0, 8))


def run_sequence(backend, frames_u8, depths, poses_gt, scene, syn, *, levels, n_boot, n_immature, desired_points, max_keyframes, kf_factor, seed=7,
                 first_kf_gap=3, max_frames=None, threads=None):
    """backend: "hip" (dsopp_amd.capi) or "cpu" (oracle.pyoracle).  Returns timings per frame / per keyframe and the estimated poses."""
    hip = backend == "hip"
    if hip:
        from dsopp_amd import capi  # (the HIP leg never imports the CPU checker)
        po = None
    else:
        from oracle import pyoracle as po
        if threads:
            po.set_threads(threads)
    W, H = scene.width, scene.height
    intr = scene.intrinsics
    n_frames = len(frames_u8) if max_frames is None else min(max_frames, len(frames_u8))
    win = capi.HipWindow(capi.default_pba_options()) if hip else po.OracleWindow(po.default_pba_options())
    aligner = capi.HipAligner(capi.default_align_options()) if hip else None
    alive, est, retired = [], {}, []
    t_frame, t_keyframe, kf_ids, tries_hist, lm_its = [], [], [], [], []
    stats = dict(activated=0, marginalised=0, solves=0)

    def make_frame(k):
        """what arrives per frame: the 8-bit image; its pyramid is part of the per-frame time"""
        f = dict(k=k, ts=1000 * (k + 1), u8=frames_u8[k])
        if hip:
            f["pyr"] = capi.Pyramid(W, H, levels)
            f["pyr"].build(f["u8"])
        else:
            f["infos"], _ = po.build_pyramid(f["u8"], levels=levels)
        return f

    def new_keyframe(f):
        kf = _Keyframe(f["k"], f["ts"])
        kf.frame = f
        info0 = _pixelinfo0(f["u8"])
        uv = candidate_pixels(seed, f["k"], f["u8"], n_immature)
        ui, vi = uv[:, 0].astype(int), uv[:, 1].astype(int)
        grad = np.stack([info0[vi, ui, 1], info0[vi, ui, 2]], axis=1)
        direction = np.stack([(uv[:, 0] - intr[2]) / intr[0], (uv[:, 1] - intr[3]) / intr[1], np.ones(len(uv))], axis=1)
        kf.imm = syn.new_immature_landmarks(uv, direction, _patch(f["u8"].astype(np.float64), uv, syn.PATTERN), grad)
        kf.dset = capi.ImmatureSet(kf.imm) if hip else None
        return kf

    def push_keyframe(kf, T_w, affine, fixed):
        if hip:
            win.push_frame(kf.id, kf.ts, None, None, intr, T_w, 1.0, affine, fixed, False, pyramid=kf.frame["pyr"])
        else:
            win.push_frame(kf.id, kf.ts, kf.frame["infos"][0], None, intr, T_w, 1.0, affine, fixed, False)
        win.set_landmarks(kf.id, kf.uv, kf.idepth, kf.patch, np.zeros(len(kf.uv), dtype=np.uint8))
        for h in alive:
            win.set_connection(h.id, kf.id, np.zeros(len(h.uv), dtype=np.uint8))
            win.set_connection(kf.id, h.id, np.zeros(len(kf.uv), dtype=np.uint8))
        alive.append(kf)

    def cpu_depth_maps():
        newest = alive[-1]
        sources = []
        for kf in alive[:-1]:
            lm = win.get_landmarks(kf.id)
            idepth = lm["idepth"].copy()
            skip = ((lm["flags"] & 3) != 0) | (idepth < 0)
            idepth[np.abs(idepth) < 1e-8] = 0
            sources.append(dict(T_w=win.get_pose(kf.id)[0], uv=kf.uv, idepth=idepth, variance=lm["inv_hdd"], skip=skip.astype(np.uint8),
                                status=win.get_residuals(kf.id, newest.id)["status"]))
        return po.create_reference_depth_maps(sources, win.get_pose(newest.id)[0], intr, W, H, levels)

    # ---- bootstrap (the reference's initializer is outside the hot path): two keyframes with active landmarks near the truth
    boot = {}
    for k in (0, first_kf_gap):
        f = make_frame(k)
        boot[k] = f
        kf = new_keyframe(f)
        kf.uv, kf.idepth, T0 = bootstrap_keyframe(seed, k, f["u8"], depths[k], poses_gt[k], n_boot, syn)
        kf.patch = _patch(f["u8"].astype(np.float64), kf.uv, syn.PATTERN)
        push_keyframe(kf, T0, np.zeros(2), k == 0)
    win.solve()
    maps = win.create_reference_depth_maps(levels) if hip else cpu_depth_maps()
    for k in range(first_kf_gap + 1):
        est[k] = syn.mat_to_params(poses_gt[k])
    est[first_kf_gap] = win.get_pose(first_kf_gap)[0]
    rmse_last = np.full(levels, 1e10)
    affine_prev = np.zeros(2)
    min_distance = 2.0
    rmse_ref = -1.0

    # the window's keyframe poses only change in a solve: read once per keyframe (as the tracker's frame objects hold them), not per frame
    kf_pose = {kf.id: win.get_pose(kf.id) for kf in alive}
    for k in range(first_kf_gap + 1, n_frames):
        t0 = time.perf_counter()
        f = make_frame(k)
        ref = alive[-1]
        T_ref, ab_ref = kf_pose[ref.id]
        if hip:
            hyp = capi.initialization_poses(est[k - 2], est[k - 1], T_ref)
            res = aligner.estimate_pose(ref.ts, T_ref, ref.frame["pyr"], maps, 1.0, ab_ref, f["ts"], f["pyr"], 1.0, intr, hyp, affine_prev, rmse_last)
            T_new, ab_new, tries, its = res["T_w_target"], res["affine_brightness"], res["tries"], res["lm_iterations"]
            if not res["success"]:
                raise RuntimeError(f"frame {k}: tracking lost")
            rmse0 = rmse_last[0]
        else:
            hyp = po.initialization_poses(est[k - 2], est[k - 1], T_ref)
            T_new, ab_new, its = hyp[0], affine_prev, 0
            for lvl in range(levels - 1, -1, -1):     # (first hypothesis only: the HIP run asserts that it needed no other)
                ids, wgt = maps[lvl]
                u, v, idp, inten = po.points_from_depth_map(ref.frame["infos"][lvl], ids, wgt)
                r = po.align_solve(po.default_align_options(), u, v, idp, inten, intr / (1 << lvl), (W >> lvl, H >> lvl), T_ref, 1.0, ab_ref,
                                   intr / (1 << lvl), f["infos"][lvl], None, T_new, 1.0, ab_new)
                T_new, ab_new = r["T_w_target"], r["affine_brightness"]
                its += r["iterations"]
                rmse_last[lvl] = r["rmse"]
            tries, rmse0 = 1, rmse_last[0]
        est[k], affine_prev = T_new, ab_new
        # calculateMeanSquareOpticalFlow, with and without rotation
        T_t_r = np.linalg.inv(syn.params_to_mat(T_new)) @ syn.params_to_mat(T_ref)
        T_nr = T_t_r.copy()
        T_nr[:3, :3] = np.eye(3)
        if hip:
            flow, flow_nr = maps.mean_square_optical_flow(0, intr, [syn.mat_to_params(T_t_r), syn.mat_to_params(T_nr)])
        else:
            flow = po.mean_square_optical_flow(maps[0][0], maps[0][1], intr, syn.mat_to_params(T_t_r))
            flow_nr = po.mean_square_optical_flow(maps[0][0], maps[0][1], intr, syn.mat_to_params(T_nr))
        # estimateDepths: every keyframe's immature landmarks against the new frame
        rel = [syn.mat_to_params(np.linalg.inv(syn.params_to_mat(T_new)) @ syn.params_to_mat(kf_pose[kf.id][0])) for kf in alive]
        abs_kf = [kf_pose[kf.id][1] for kf in alive]
        if hip:
            capi.estimate_depths_batched([kf.dset for kf in alive], f["pyr"], 0, intr, np.stack(rel), np.ones(len(alive)), np.stack(abs_kf), 1.0, affine_prev)
            alive[0].dset.sync()
        else:
            for kf, T_new_kf, abkf in zip(alive, rel, abs_kf):
                po.estimate_depths(kf.imm, f["infos"][0], None, intr, T_new_kf, 1.0, abkf, 1.0, affine_prev)
        # keyframe decision: mean_square_optical_flow_and_rmse_keyframe_strategy.cpp:14-48 (exposures are 1 in this sequence)
        if rmse_ref < 0:
            rmse_ref = rmse0
        need_kf = kf_factor * (4.5 * flow + 9.0 * flow_nr + 2.0 * abs(affine_prev[0] - ab_ref[0])) > 1.0 or rmse0 / rmse_ref > 4.0
        t_frame.append(time.perf_counter() - t0)
        tries_hist.append(tries)
        lm_its.append(its)
        if not need_kf:
            if hip:
                f["pyr"].close()
            continue
        # ================= new keyframe =================
        rmse_ref = -1.0
        t1 = time.perf_counter()
        new = new_keyframe(f)
        t_candidates = time.perf_counter() - t1      # candidate pixels: the feature extractor's job, not counted
        if hip:
            st, idp_act, ares = win.activate_landmarks([kf.id for kf in alive], [kf.dset for kf in alive], new.frame["pyr"], est[k], 1.0, affine_prev,
                                                       desired_points, min_distance, True)
            min_distance = ares["min_distance_to_neighbor"]
        else:
            ofr = []
            for kf in alive:
                lm = win.get_landmarks(kf.id)
                Tkf, abkf = win.get_pose(kf.id)
                ofr.append(dict(pixelinfo=kf.frame["infos"][0], mask=None, T_w=Tkf, exposure=1.0, affine=abkf, active_uv=kf.uv, active_idepth=lm["idepth"],
                                active_skip=((lm["flags"] & 3) != 0).astype(np.uint8), immature=kf.imm))
            ofr.append(dict(pixelinfo=new.frame["infos"][0], mask=None, T_w=est[k], exposure=1.0, affine=affine_prev))
            st, _, min_distance = po.activate_landmarks(ofr, intr, 20.0, desired_points, min_distance, refine=True)
            idp_act = [0.5 * fo["immature"]["idepth_min"] + 0.5 * fo["immature"]["idepth_max"] for fo in ofr[:-1]]
        for kf, s, idp in zip(alive, st, idp_act):
            act = s == 0
            stats["activated"] += int(act.sum())
            if not act.any():
                continue
            proj = kf.imm["projection"][act]
            kf.uv = np.concatenate([kf.uv, proj])
            kf.patch = np.concatenate([kf.patch, kf.imm["patch"][act]])
            # (flags + inverse depths only, in one transfer on the HIP side: dsopp_hip_window_get_frame_update — not the n x K Schur rows)
            cur = win.get_frame_update(kf.id, []) if hip else win.get_landmarks(kf.id)
            win.set_landmarks(kf.id, kf.uv, np.concatenate([cur["idepth"], idp[act]]), kf.patch,
                              np.concatenate([cur["flags"] & 3, np.zeros(int(act.sum()), dtype=np.uint8)]))
            # set_connection APPENDS the entries [current size, n) of a residual list (photometric_bundle_adjustment.cpp:109-123): the new
            # landmarks start as kOk = 0, what is passed for the existing ones is not read — no need to fetch their statuses first
            # (also towards a keyframe marginalised at the previous keyframe: it stays in the solver, with its connections, until the
            # pushFrame below folds it into the prior — LocalFrame::update gives the new landmarks a residual in every connection)
            fresh = np.zeros(len(kf.uv), dtype=np.uint8)
            for h in alive + retired:
                if h is not kf:
                    win.set_connection(kf.id, h.id, fresh)
        push_keyframe(new, est[k], affine_prev, False)
        retired.clear()
        win.solve()
        stats["solves"] += 1
        est[k] = win.get_pose(new.id)[0]
        if len(alive) > max_keyframes:
            victim = alive[1]
            for kf in alive:
                cur = win.get_frame_update(kf.id, []) if hip else win.get_landmarks(kf.id)
                flags = cur["flags"] & 3
                if kf is victim:
                    flags = flags | 1
                win.set_landmarks(kf.id, kf.uv, cur["idepth"], kf.patch, flags.astype(np.uint8))
            win.mark_frame_marginalized(victim.id)
            alive.remove(victim)
            # both backends BORROW the marginalised frame's image until the next pushFrame has folded it into the prior: keep it that long
            retired[:] = [victim]
            if hip:
                victim.dset.close()
            stats["marginalised"] += 1
        if hip:
            win.refill_reference_depth_maps(maps)
        else:
            maps = cpu_depth_maps()
        rmse_last = np.full(levels, 1e10)
        kf_pose = {kf.id: win.get_pose(kf.id) for kf in alive}
        t_keyframe.append(time.perf_counter() - t1 - t_candidates)
        kf_ids.append(k)

    # drift against the synthetic ground truth (frame 0 is fixed at the truth, the bootstrap depths fix the scale: no alignment)
    ks = sorted(est)
    dt, dr = [], []
    for k in ks:
        T = syn.params_to_mat(est[k])
        E = np.linalg.inv(poses_gt[k]) @ T
        dt.append(np.linalg.norm(E[:3, 3]))
        dr.append(np.degrees(np.arccos(np.clip((np.trace(E[:3, :3]) - 1) / 2, -1, 1))))
    path = float(sum(np.linalg.norm(poses_gt[k][:3, 3] - poses_gt[k - 1][:3, 3]) for k in ks[1:]))
    # A monocular window has no scale of its own: the two bootstrap keyframes set it (depths drawn within 0.2 % of the truth against a pose
    # drawn 5 mm off a 48 mm baseline), and whatever they settle on stays — per seed that is 0.02 % .. 6 % of the path, for the CPU port and
    # the HIP path alike (they agree to 1e-12).  The tracking error proper is what remains after the one scale factor is fitted, as the
    # monocular benchmarks of the reference's data sets do (Sim(3) alignment; frame 0 is fixed at the truth, so only the scale is free).
    te = np.array([syn.params_to_mat(est[k])[:3, 3] - poses_gt[0][:3, 3] for k in ks])
    tg = np.array([poses_gt[k][:3, 3] - poses_gt[0][:3, 3] for k in ks])
    scale = float((te * tg).sum() / max((te * te).sum(), 1e-30))
    dts = np.linalg.norm(scale * te - tg, axis=1)
    tf, tk = np.array(t_frame) * 1e3, np.array(t_keyframe) * 1e3
    if hip:
        win.close()
    return dict(frames=len(t_frame), keyframes=len(t_keyframe), ms_per_frame_mean=float(tf.mean()), ms_per_frame_median=float(np.median(tf)),
                ms_per_frame_p95=float(np.quantile(tf, 0.95)), ms_per_keyframe_mean=float(tk.mean()) if len(tk) else None,
                ms_per_keyframe_p95=float(np.quantile(tk, 0.95)) if len(tk) else None,
                ms_per_frame_including_keyframe_work=float((tf.sum() + tk.sum()) / len(tf)),
                lm_iterations_per_frame=float(np.mean(lm_its)), hypotheses_tried_max=int(max(tries_hist)), keyframe_every=float(len(t_frame) / max(1, len(t_keyframe))),
                translation_error_final=float(dt[-1]), translation_error_rmse=float(np.sqrt(np.mean(np.square(dt)))), rotation_error_final_deg=float(dr[-1]),
                rotation_error_rmse_deg=float(np.sqrt(np.mean(np.square(dr)))), path_length=path, drift_percent_of_path=float(100 * dt[-1] / max(path, 1e-12)),
                scale_of_the_estimate=scale, translation_error_final_scale_aligned=float(dts[-1]),
                drift_percent_of_path_scale_aligned=float(100 * dts[-1] / max(path, 1e-12)),
                window_landmarks_end=int(sum(len(kf.uv) for kf in alive)), **stats), est


def run(torch, syn, width=640, height=480, levels=4, frames=200, cpu_frames=30, step=0.2, n_boot=1000, n_immature=1500, desired_points=2000,
        max_keyframes=7, kf_factor=None, cpu_threads=None, no_cpu=False, native=True):
    """the bench's entry point: renders the sequence, runs the HIP tracker over all of it and the CPU port over its first `cpu_frames` frames"""
    scene = syn.Scene.make(width, height, 41)
    poses = [syn.se3_exp(step * k * syn.BASE_MOTION) for k in range(frames)]
    rendered = render_frames(torch, scene, poses)
    u8 = [r[0] for r in rendered]
    depths = {0: rendered[0][1], 3: rendered[3][1]}
    # the reference reads the strategy's factor from its configuration; 5 gives a keyframe every ~4 frames of this motion
    kf_factor = kf_factor if kf_factor is not None else 5.0
    kw = dict(levels=levels, n_boot=n_boot, n_immature=n_immature, desired_points=desired_points, max_keyframes=max_keyframes, kf_factor=kf_factor)
    out = {"workload": f"{frames} synthetic frames {width}x{height}, {levels} pyramid levels, forward-dominant motion ({step} x BASE_MOTION per frame), window of "
                       f"{max_keyframes} keyframes, {n_immature} candidate pixels per keyframe, {desired_points} desired active points, keyframes by the reference's "
                       f"flow rule (factor {kf_factor})",
           "not_in_the_per_frame_time": "rendering, candidate-pixel selection (feature extractor), marginalisation strategy"}
    hip, est_hip = run_sequence("hip", u8, depths, poses, scene, syn, **kw)
    out["hip"] = hip
    if native:
        # the same sequence, the same calls, from C++ (dsopp_amd/host/tick_sequence.cpp over the host mirror): no interpreter, no NumPy
        # marshalling between the C-ABI calls — what the reference's MonocularTracker::tick would pay for this backend
        import tempfile
        path = os.path.join(tempfile.mkdtemp(prefix="dsopp_seq_"), "sequence.bin")
        try:
            export_sequence(path, u8, depths, poses, scene, syn, **kw)
            nat, est_nat = run_native(path)
            common = sorted(set(est_hip) & set(est_nat))
            nat["pose_difference_to_the_python_driven_run_max"] = float(max(np.abs(est_hip[k] - est_nat[k]).max() for k in common))
            nat["frames_compared"] = len(common)
            out["native"] = nat
        except Exception as exc:  # noqa: BLE001 — reported, never fatal for the bench line
            out["native"] = {"error": repr(exc)[:600]}
        finally:
            if os.path.exists(path):
                os.remove(path)
    if not no_cpu and cpu_frames > 0:
        import multiprocessing
        threads = cpu_threads or max(1, min(multiprocessing.cpu_count(), 8) - 1)    # the reference's pool: min(hw, 8) - 1 (dsopp_main.cpp:114-119)
        cpu, est_cpu = run_sequence("cpu", u8, depths, poses, scene, syn, max_frames=cpu_frames, threads=threads, **kw)
        cpu["threads"] = threads
        out["cpu_port"] = cpu
        common = sorted(set(est_hip) & set(est_cpu))
        out["hip_vs_cpu_port_pose_difference_max"] = float(max(np.abs(est_hip[k] - est_cpu[k]).max() for k in common))
        out["speedup_per_frame"] = cpu["ms_per_frame_mean"] / hip["ms_per_frame_mean"]
        if cpu["ms_per_keyframe_mean"] and hip["ms_per_keyframe_mean"]:
            out["speedup_per_keyframe"] = cpu["ms_per_keyframe_mean"] / hip["ms_per_keyframe_mean"]
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--cpu-frames", type=int, default=30)
    ap.add_argument("--kf-factor", type=float, default=None)
    ap.add_argument("--step", type=float, default=0.2)
    a = ap.parse_args()
    import torch
    from dsopp_amd import synthetic as syn
    print(json.dumps(run(torch, syn, a.width, a.height, a.levels, a.frames, a.cpu_frames, step=a.step, kf_factor=a.kf_factor)))
