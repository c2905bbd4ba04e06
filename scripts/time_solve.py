import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dsopp_amd import capi, synthetic as syn
win = syn.make_window(7, 2000, 640, 480, seed=0)
for unc in (1, 0):
    g = capi.HipWindow(capi.default_pba_options(estimate_uncertainty=unc))
    syn.load_window(g, win)
    g.snapshot()
    ts = []
    for _ in range(8):
        g.restore()
        t0 = time.perf_counter(); g.solve(); ts.append(time.perf_counter() - t0)
    to = []
    for _ in range(8):
        g.restore()
        t0 = time.perf_counter(); g.optimize(); to.append(time.perf_counter() - t0)
    print(f"estimate_uncertainty={unc}: solve() {np.median(ts)*1e3:.3f} ms   optimize() {np.median(to)*1e3:.3f} ms")
    g.close()
from oracle import pyoracle as po
po.set_threads(7)
o = po.OracleWindow(po.default_pba_options()); syn.load_window(o, win)
t0 = time.perf_counter(); o.solve(); print(f"oracle solve() {1e3*(time.perf_counter()-t0):.1f} ms")
