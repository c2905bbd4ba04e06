import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
from dsopp_amd import capi, synthetic as syn
win = syn.make_window(7, 2000, 640, 480, seed=0)
g = capi.HipWindow(capi.default_pba_options()); syn.load_window(g, win)
out = (C.c_longlong*48)()
capi.lib().dsopp_hip_debug_solve_stamps(g._h, out)
g.snapshot()
for _ in range(3):
    g.restore(); g.optimize()
capi.lib().dsopp_hip_debug_solve_stamps(g._h, out)
st = np.array(list(out), dtype=np.int64)
print("solve phase us:", np.diff(st[:7]) / 100.0, "total", (st[6]-st[0])/100.0)
print("reduceSchur (wg 1) phase us:", np.diff(st[8:14]) / 100.0, "total", (st[13]-st[8])/100.0)
print("prologue: loads+lds", (st[14]-st[8])/100.0, "tree+decide", (st[15]-st[14])/100.0, "apply", (st[9]-st[15])/100.0)
ch = st[16:16+3*7+1]
print("chol: factor0", (st[16]-st[2])/100.0)
for kb in range(7):
    print(f"  kb={kb}: colupdate+bar {(ch[1+3*kb]-ch[3*kb])/100.0:5.2f}  factor+panel(w0) {(ch[2+3*kb]-ch[1+3*kb])/100.0:5.2f}  wait-bar {(ch[3+3*kb]-ch[2+3*kb])/100.0:5.2f}")
