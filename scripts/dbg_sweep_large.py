"""phase stamps (wall_clock64, 100 MHz) of one sweep workgroup at a given window size — needs a -DDSOPP_HIP_STAMPS build (DSOPP_HIP_LIB)"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
from dsopp_amd import capi, synthetic as syn
F, P = int(sys.argv[1]), int(sys.argv[2])
win = syn.make_window(F, P, 640, 480, seed=1)
g = capi.HipWindow(capi.default_pba_options()); syn.load_window(g, win)
out = (C.c_longlong * 16)()
g.snapshot()
for lin in (1,):
    for rep in range(3):
        capi.lib().dsopp_hip_debug_sweep_stamps(g._h, lin, out)
        g.restore(); g.set_max_iterations(2); g.optimize()
        capi.lib().dsopp_hip_debug_sweep_stamps(g._h, lin, out)
        st = np.array(list(out), dtype=np.int64)
        print("stamps rel. to kernel start of the block (us): 0 start", [(int(x) - int(st[0])) / 100.0 for x in st[1:8]], "group starts", [(int(x) - int(st[0])) / 100.0 for x in st[10:16]],
              "shader clock GHz over the block", round(float(st[15]) / max(1.0, float(st[7] - st[0])) / 10.0, 3), "block start offset", (st[0] - st[8]) / 100.0)
