"""kernel timeline of one dsopp_hip_window_solve (LM loop + relinearise + covariances + point statuses) on the C1 window,
for `rocprofv3 --kernel-trace` (tuning aid)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (its HIP runtime first)
from dsopp_amd import capi, synthetic as syn

win = syn.make_window(7, 2000, 640, 480, seed=0)
g = capi.HipWindow(capi.default_pba_options())
syn.load_window(g, win)
g.snapshot()
for _ in range(4):
    g.restore()
    g.solve()
g.close()
