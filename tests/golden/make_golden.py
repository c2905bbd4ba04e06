#!/usr/bin/env python
"""Generates tests/golden/pba_tiny.npz and tests/golden/tracker_tiny.npz — committed input/output vectors of the hot path
(bundle adjustment, pyramid, alignment) and of the rows around it (depth estimation, landmark activation, depth maps'
optical flow, pose hypotheses).

The reference ships NO golden vectors for this path and cannot be built or imported in this environment,
so these vectors do not come from the reference: the expected outputs are produced by the CPU oracle (oracle/*.hpp) and,
for the residuals, independently by the NumPy spec (oracle/spec.py); the two are asserted to agree before anything is
written.  They pin the oracle against regressions and give the GPU tests a fixed target that does not depend on the
oracle being rebuilt.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from dsopp_amd import synthetic as syn  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from oracle import spec  # noqa: E402


def main():
    win = syn.make_window(num_frames=3, num_points=60, width=160, height=120, seed=5)
    out = {"intrinsics": win.scene.intrinsics, "num_frames": 3}
    for i, f in enumerate(win.frames):
        out[f"image_u8_{i}"] = f.image_u8
        out[f"uv_{i}"] = f.uv
        out[f"idepth_{i}"] = f.idepth_init
        out[f"patch_{i}"] = f.patch
        out[f"T_init_{i}"] = syn.mat_to_params(f.T_w_c_init)
        out[f"T_gt_{i}"] = syn.mat_to_params(f.T_w_c_gt)
    o = po.OracleWindow(po.default_pba_options())
    syn.load_window(o, win)
    o.begin()
    e0, n0 = o.calculate_energy()
    # independent cross-check of the residual energies by the NumPy spec before committing anything
    sf = {f.frame_id: spec.SpecFrame(syn.mat_to_params(f.T_w_c_init), f.affine_init, np.zeros(8), f.pixelinfo, win.scene.intrinsics)
          for f in win.frames}
    e_spec = 0.0
    for fr in win.frames:
        for ft in win.frames:
            if fr.frame_id == ft.frame_id:
                continue
            for k in range(len(fr.uv)):
                ok, r, _, _ = spec.residual8(sf[fr.frame_id], sf[ft.frame_id], fr.uv[k], fr.idepth_init[k], fr.patch[k])
                if ok:
                    e_spec += spec.huber(r, 20.0)[0]
    assert abs(e_spec - e0) <= 1e-10 * e0, (e_spec, e0)
    o.linearize()
    Hpp, bpp, Hsc, bsc = o.get_system()
    step = o.calculate_step(1e-5)
    e1, n1 = o.calculate_energy()
    out.update(energy0=e0, n_valid0=n0, H_pp=Hpp, b_pp=bpp, H_schur=Hsc, b_schur=bsc, step=step, energy1=e1, n_valid1=n1)
    o2 = po.OracleWindow(po.default_pba_options())
    syn.load_window(o2, win)
    e, it, nv = o2.solve()
    out.update(solve_energy=e, solve_iterations=it, solve_n_valid=nv)
    for i, f in enumerate(win.frames):
        T, ab = o2.get_pose(f.frame_id)
        out[f"T_final_{i}"] = T
        out[f"idepth_final_{i}"] = o2.get_landmarks(f.frame_id)["idepth"]
    # pyramid + alignment vectors
    infos, _ = po.build_pyramid(win.frames[0].image_u8, levels=3)
    infos_t, _ = po.build_pyramid(win.frames[1].image_u8, levels=3)
    for l in range(3):
        out[f"pyramid0_level{l}"] = infos[l]
    f0, f1 = win.frames[0], win.frames[1]
    rng = np.random.default_rng(1)
    h, w = infos[1].shape[:2]
    idsum, wgt = np.zeros((h, w)), np.zeros((h, w))
    xs, ys = rng.integers(0, w, 400), rng.integers(0, h, 400)
    idsum[ys, xs] = 1.0 / f0.depth[ys * 2, xs * 2]
    wgt[ys, xs] = 1.0
    u, v, idp, inten = po.points_from_depth_map(infos[1], idsum, wgt)
    intr1 = win.scene.intrinsics / 2
    ra = po.align_solve(po.default_align_options(), u, v, idp, inten, intr1, (w, h), syn.mat_to_params(f0.T_w_c_gt), 1.0, np.zeros(2), intr1,
                        infos_t[1], None, syn.mat_to_params(f1.T_w_c_init), 1.0, np.zeros(2))
    out.update(align_idepth_sum=idsum, align_weight=wgt, align_T=ra["T_w_target"], align_rmse=ra["rmse"], align_iterations=ra["iterations"],
               align_n_valid=ra["n_valid"], align_affine=ra["affine_brightness"])
    path = os.path.join(ROOT, "tests", "golden", "pba_tiny.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def tracker_rows():
    """a 4-keyframe window (the 4th is the new keyframe): immature landmarks traced in two frames, then activated"""
    W, H = 160, 120
    win = syn.make_window(num_frames=4, num_points=4 * 110, width=W, height=H, seed=29, pose_noise=False)
    intr = win.scene.intrinsics
    out = {"intrinsics": intr, "num_frames": 4}
    frames = []
    for i, f in enumerate(win.frames):
        out[f"image_u8_{i}"] = f.image_u8
        out[f"T_{i}"] = syn.mat_to_params(f.T_w_c_gt)
        d = dict(pixelinfo=f.pixelinfo, mask=None, T_w=syn.mat_to_params(f.T_w_c_gt), exposure=1.0, affine=np.zeros(2))
        if i < 3:
            na = 40
            uv = f.uv[na:]
            ui, vi = uv[:, 0].astype(int), uv[:, 1].astype(int)
            grad = np.stack([f.pixelinfo[vi, ui, 1], f.pixelinfo[vi, ui, 2]], axis=1)
            direction = np.stack([(uv[:, 0] - intr[2]) / intr[0], (uv[:, 1] - intr[3]) / intr[1], np.ones(len(uv))], axis=1)
            lms = po.new_immature_landmarks(uv, direction, f.patch[na:], grad)
            out.update({f"active_uv_{i}": f.uv[:na], f"active_idepth_{i}": f.idepth_init[:na], f"active_patch_{i}": f.patch[:na],
                        f"imm_uv_{i}": uv, f"imm_direction_{i}": direction, f"imm_patch_{i}": f.patch[na:], f"imm_gradient_{i}": grad})
            for step, j in enumerate((i + 1, 3 if i < 2 else 0)):   # (never the same frame twice: that is a tie generator)
                g = win.frames[j]
                po.estimate_depths(lms, g.pixelinfo, None, intr, syn.mat_to_params(np.linalg.inv(g.T_w_c_gt) @ f.T_w_c_gt))
                for k in ("idepth_min", "idepth_max", "uniqueness", "search_pixel_interval", "status", "traced"):
                    out[f"depth_{i}_obs{step}_{k}"] = np.array(lms[k])
            d.update(active_uv=f.uv[:na], active_idepth=f.idepth_init[:na], active_skip=np.zeros(na, dtype=np.uint8), immature=lms)
        frames.append(d)
    st, n_act, dist = po.activate_landmarks(frames, intr, 20.0, 120, 1.5, refine=True)
    out.update(activation_desired=120, activation_distance_in=1.5, activation_distance_out=dist, activation_active_points=n_act)
    for i in range(3):
        out[f"activation_status_{i}"] = st[i]
        out[f"activation_idepth_{i}"] = 0.5 * frames[i]["immature"]["idepth_min"] + 0.5 * frames[i]["immature"]["idepth_max"]
        out[f"activation_set_status_{i}"] = frames[i]["immature"]["status"]
    # reference depth maps of keyframe 2 from keyframes 0, 1 and the optical flow of two relative poses
    sources = [dict(T_w=syn.mat_to_params(f.T_w_c_gt), uv=f.uv, idepth=f.idepth_gt, variance=np.full(len(f.uv), 1e-5),
                    skip=np.zeros(len(f.uv), dtype=np.uint8), status=np.zeros(len(f.uv), dtype=np.uint8)) for f in win.frames[:2]]
    maps = po.create_reference_depth_maps(sources, syn.mat_to_params(win.frames[2].T_w_c_gt), intr, W, H, 2)
    T_rel = np.linalg.inv(win.frames[3].T_w_c_gt) @ win.frames[2].T_w_c_gt
    T_norot = T_rel.copy()
    T_norot[:3, :3] = np.eye(3)
    out.update(maps_idepth_sum_0=maps[0][0], maps_weight_0=maps[0][1], flow_T=np.stack([syn.mat_to_params(T_rel), syn.mat_to_params(T_norot)]),
               flow=np.array([po.mean_square_optical_flow(maps[0][0], maps[0][1], intr, syn.mat_to_params(T)) for T in (T_rel, T_norot)]))
    out["hypotheses"] = po.initialization_poses(out["T_1"], out["T_2"], out["T_0"])
    path = os.path.join(ROOT, "tests", "golden", "tracker_tiny.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
    tracker_rows()
