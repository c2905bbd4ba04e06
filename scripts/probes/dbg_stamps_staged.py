"""Phase stamps of the staged two-stage Schur kernel (workgroup 1, its first two sub-chunks).  Needs a -DDSOPP_HIP_STAMPS build."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
from dsopp_amd import capi, synthetic as syn
F = int(sys.argv[1]) if len(sys.argv) > 1 else 12
P = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
win = syn.make_window(F, P, 640, 480, seed=0)
g = capi.HipWindow(capi.default_pba_options()); syn.load_window(g, win)
out = (C.c_longlong * 64)()
assert capi.lib().dsopp_hip_debug_solve_stamps(g._h, out) == 0, capi.lib().dsopp_hip_last_error()
g.snapshot()
for _ in range(3):
    g.restore(); g.optimize()
    capi.lib().dsopp_hip_debug_solve_stamps(g._h, out)
    st = np.array(list(out), dtype=np.int64) / 100.0
    ts = st[32:48]
    print("staged (wg 1): 0 descriptors in LDS | per sub-chunk: 1 before wait, 2 rows landed, 3 staging read, 4 next requested, 5 finalised, 6 barrier, 7 SYRK done | 15 end")
    print("  sub 0:", np.round(ts[1:8] - ts[0], 2), " sub 1:", np.round(ts[8:15] - ts[0], 2), " end:", round(float(ts[15] - ts[0]), 2),
          " last workgroup ends:", round(float(st[49] - ts[0]), 2), " longest workgroup:", round(float(st[50]), 2))
