import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from dsopp_amd import capi, synthetic as syn
from oracle import pyoracle as po
import test_gpu_sliding_window as t
win = syn.make_window(num_frames=9, num_points=9 * 300, width=320, height=240, seed=61)
lo = t._drive(po.OracleWindow(po.default_pba_options()), win)
g = capi.HipWindow(capi.default_pba_options())
mode = int(os.environ.get("LM_MODE", "0"))
g.set_lm_mode(mode)
lg = t._drive(g, win)
for so, sg in zip(lo, lg):
    dp = max(np.abs(so["poses"][f] - sg["poses"][f]).max() for f in so["poses"])
    di = max(np.abs(so["idepth"][f] - sg["idepth"][f]).max() for f in so["idepth"])
    ds = sum(int((so["status"][k] != sg["status"][k]).sum()) for k in so["status"])
    print(so["step"], "it", so["iterations"], sg["iterations"], "nv", so["n_valid"], sg["n_valid"], "dE/E %.2e" % (abs(so["energy"] - sg["energy"]) / abs(so["energy"])),
          "dpose %.2e didepth %.2e status diffs %d" % (dp, di, ds))
