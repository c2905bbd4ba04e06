#!/usr/bin/env python
"""Generates tests/golden/pba_tiny.npz — committed input/output vectors of the hot path.

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


if __name__ == "__main__":
    main()
