import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsopp_amd import capi, synthetic as syn
F, P = int(sys.argv[1]), int(sys.argv[2])
win = syn.make_window(F, P, 640, 480, seed=1)
g = capi.HipWindow(capi.default_pba_options()); syn.load_window(g, win); g.snapshot(); g.restore()
print(F, P, "schur", round(g.time_kernel("schur", 30), 1), "loop sweep", round(g.time_kernel("sweep_linearize_loop", 10), 1))
