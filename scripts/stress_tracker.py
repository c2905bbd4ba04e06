"""Race detector for the persistent tracker kernel: estimatePose N times on bitwise identical inputs (the window solved once, maps and pyramids
resident) — every call has to return the bitwise identical pose, per-level rmse and iteration count.  Prints the distinct outcomes."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dsopp_amd import capi, synthetic as syn
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for (W, H, L, P) in ((1280, 1024, 5, 2000), (320, 240, 4, 800), (1280, 1024, 4, 2000)):
    win = syn.make_window(num_frames=4, num_points=P, width=W, height=H, seed=17)
    g = syn.load_window(capi.HipWindow(capi.default_pba_options()), win)
    g.solve()
    newest, target = win.frames[-1], win.frames[-2]
    intr = win.scene.intrinsics
    T_init = syn.mat_to_params(target.T_w_c_init)
    maps_g = g.create_reference_depth_maps(L)
    pr, pt = capi.Pyramid(W, H, L), capi.Pyramid(W, H, L)
    pr.build(newest.image_u8); pt.build(target.image_u8)
    T_ref_g, ab_ref_g = g.get_pose(newest.frame_id)
    a = capi.HipAligner(capi.default_align_options())
    if os.environ.get("STRESS_WARM_FALLBACK"):
        # fill every cache of the call (level points of the depth maps, ...) through the launch-per-iteration path first
        a.set_lm_path(1)
        a.estimate_pose(newest.timestamp, T_ref_g, pr, maps_g, 1.0, ab_ref_g, newest.timestamp + 1, pt, 1.0, intr, T_init[None, :], np.zeros(2), np.full(L, 1e10))
        a.set_lm_path(0)
    seen = collections.Counter()
    for i in range(N):
        rmse_last = np.full(L, 1e10)
        res = a.estimate_pose(newest.timestamp, T_ref_g, pr, maps_g, 1.0, ab_ref_g, newest.timestamp + 1, pt, 1.0, intr, T_init[None, :], np.zeros(2), rmse_last)
        key = (res["lm_iterations"], bool(res["success"]), res["T_w_target"].tobytes(), rmse_last.tobytes())
        seen[key] += 1
    print(f"{W}x{H} x {L} levels: {N} calls, {len(seen)} distinct outcome(s):", sorted(((k[0], k[1], v) for k, v in seen.items()), key=lambda t: -t[2]))
    for o in (a, maps_g, pr, pt, g):
        o.close()
