import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
from dsopp_amd import capi, synthetic as syn
win = syn.make_window(7, 2000, 640, 480, seed=0)
g = capi.HipWindow(capi.default_pba_options()); syn.load_window(g, win)
out = (C.c_longlong*16)()
g.snapshot()
for lin in (1, 0):
    for rep in range(2):
        capi.lib().dsopp_hip_debug_sweep_stamps(g._h, lin, out)
        g.restore(); g.set_max_iterations(1); g.optimize()
        capi.lib().dsopp_hip_debug_sweep_stamps(g._h, lin, out)
        st = np.array(list(out), dtype=np.int64)
        if lin: print("   inside 3 -> 4: texels + residual + Huber", (st[10]-st[3])/100.0, " Jacobians", (st[11]-st[10])/100.0, " gram rows to LDS", (st[12]-st[11])/100.0, " butterflies + row store", (st[13]-st[12])/100.0, " bookkeeping", (st[14]-st[13])/100.0, " gram contraction", (st[4]-st[14])/100.0)
        print("lin" if lin else "energy", "phase us:", np.diff(st[:7]) / 100.0, "block total", (st[6]-st[0])/100.0, "kernel span", (st[9]-st[8])/100.0, "mid block start offset", (st[0]-st[8])/100.0)
