import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, copy
from dsopp_amd import capi, synthetic as syn
from oracle import pyoracle as po
import test_depth_estimation as t
win = syn.make_window(num_frames=4, num_points=4 * 150, width=320, height=240, seed=71, pose_noise=False)
intr = win.scene.intrinsics
fr = win.frames[0]
uv, grad = t._landmarks(po, fr)
direction = np.stack([(uv[:, 0] - intr[2]) / intr[0], (uv[:, 1] - intr[3]) / intr[1], np.ones(len(uv))], axis=1)
lo = po.new_immature_landmarks(uv, direction, fr.patch, grad)
lg = copy.deepcopy(lo)
for step, ft in enumerate((win.frames[1], win.frames[3], win.frames[2])):
    T = t._mat_to_params(t._rel(ft.T_w_c_gt, fr.T_w_c_gt))
    pyr = capi.Pyramid(320, 240, 1)
    pyr.set_level(0, ft.pixelinfo)
    po.estimate_depths(lo, ft.pixelinfo, None, intr, T)
    capi.estimate_depths(lg, pyr, 0, intr, T)
    print(step, "oracle", np.bincount(lo["status"], minlength=7), "gpu", np.bincount(lg["status"], minlength=7))
    d = np.flatnonzero(lo["status"] != lg["status"])
    print("  status diffs", d[:10], "max d idmin", np.abs(lo["idepth_min"] - lg["idepth_min"]).max(), "idmax", np.abs(lo["idepth_max"] - lg["idepth_max"]).max(),
          "spi", np.abs(lo["search_pixel_interval"] - lg["search_pixel_interval"]).max())
    fin = (lo["uniqueness"] < 1e300) & (lg["uniqueness"] < 1e300)
    print("  uniq rel", (np.abs(lo["uniqueness"][fin] - lg["uniqueness"][fin]) / np.abs(lo["uniqueness"][fin])).max(), "fin mismatch", int(((lo["uniqueness"] < 1e300) != (lg["uniqueness"] < 1e300)).sum()))
