import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dsopp_amd import capi, synthetic as syn
win = syn.make_window(num_frames=12, num_points=12 * 286, width=640, height=480, seed=61)
intr = win.scene.intrinsics
g = capi.HipWindow(capi.default_pba_options())
alive = []
pyramids = []
for f in win.frames:   # the tracker built every keyframe's device pyramid when the frame arrived: not part of the keyframe step
    p = capi.Pyramid(640, 480, 1)
    p.set_level(0, f.pixelinfo)
    pyramids.append(p)
for k, f in enumerate(win.frames):
    t0 = time.perf_counter()
    g.push_frame(f.frame_id, f.timestamp, None, None, intr, syn.mat_to_params(f.T_w_c_init), f.exposure, f.affine_init, f.fixed, False, pyramid=pyramids[k])
    t1 = time.perf_counter()
    g.set_landmarks(f.frame_id, f.uv, f.idepth_init, f.patch, np.zeros(len(f.uv), dtype=np.uint8))
    for a in alive:
        g.set_connection(a.frame_id, f.frame_id, np.zeros(len(a.uv), dtype=np.uint8))
        g.set_connection(f.frame_id, a.frame_id, np.zeros(len(f.uv), dtype=np.uint8))
    alive.append(f)
    t2 = time.perf_counter()
    if len(alive) < 2: continue
    g.solve()
    t3 = time.perf_counter()
    for a in alive:
        g.get_pose(a.frame_id)
        g.get_frame_update(a.frame_id, [b.frame_id for b in alive if b is not a])
    t4 = time.perf_counter()
    if len(alive) == 7 and k + 1 < len(win.frames):
        victim = alive[1]
        for a in alive:
            fl = np.zeros(len(a.uv), dtype=np.uint8); fl[::4 if a is victim else 9] = 1
            g.set_landmarks(a.frame_id, a.uv, a.idepth_init, a.patch, fl)
        g.mark_frame_marginalized(victim.frame_id)
        alive.remove(victim)
    t5 = time.perf_counter()
    print(f"kf {k}: frames {len(alive)} push_frame(incl. fold-in of marginalised frames) {1e3*(t1-t0):.2f} ms, landmarks+connections {1e3*(t2-t1):.2f}, solve {1e3*(t3-t2):.2f}, read-back {1e3*(t4-t3):.2f}, marginalise flags {1e3*(t5-t4):.2f}")
