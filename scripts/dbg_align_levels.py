"""per-level LM iteration counts of estimatePose: persistent kernel (stamps build, DSOPP_HIP_TRACE=1 prints them) against the oracle chain"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from dsopp_amd import capi, synthetic as syn
from oracle import pyoracle as po
from test_depth_maps import _sources_from_window
W, H, L = 1280, 1024, 5
win = syn.make_window(num_frames=4, num_points=2000, width=W, height=H, seed=17)
o = syn.load_window(po.OracleWindow(po.default_pba_options()), win)
g = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
o.solve(); g.solve()
newest, target = win.frames[-1], win.frames[-2]
intr = win.scene.intrinsics
T_init = syn.mat_to_params(target.T_w_c_init)
sources, T_ref_o = _sources_from_window(o, win)
maps_o = po.create_reference_depth_maps(sources, T_ref_o, intr, W, H, L)
infos_ref, _ = po.build_pyramid(newest.image_u8, levels=L)
infos_tgt, _ = po.build_pyramid(target.image_u8, levels=L)
_, ab_ref_o = o.get_pose(newest.frame_id)
T, ab = T_init, np.zeros(2)
for lvl in range(L - 1, -1, -1):
    u, v, idp, inten = po.points_from_depth_map(infos_ref[lvl], *maps_o[lvl])
    h, w = infos_ref[lvl].shape[:2]
    r = po.align_solve(po.default_align_options(), u, v, idp, inten, intr / (1 << lvl), (w, h), T_ref_o, 1.0, ab_ref_o, intr / (1 << lvl), infos_tgt[lvl], None, T, 1.0, ab)
    T, ab = r["T_w_target"], r["affine_brightness"]
    print(f"oracle level {lvl}: {len(u)} points, {r['iterations']} iterations, rmse {r['rmse']:.12g}", file=sys.stderr)
maps_g = g.create_reference_depth_maps(L)
pr, pt = capi.Pyramid(W, H, L), capi.Pyramid(W, H, L)
pr.build(newest.image_u8); pt.build(target.image_u8)
T_ref_g, ab_ref_g = g.get_pose(newest.frame_id)
a = capi.HipAligner(capi.default_align_options())
rmse_last = np.full(L, 1e10)
res = a.estimate_pose(newest.timestamp, T_ref_g, pr, maps_g, 1.0, ab_ref_g, newest.timestamp + 1, pt, 1.0, intr, T_init[None, :], np.zeros(2), rmse_last)
print(res["lm_iterations"], res["success"], rmse_last, file=sys.stderr)
