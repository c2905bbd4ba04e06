import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsopp_amd import capi, synthetic as syn
F, P = int(sys.argv[1]), int(sys.argv[2])
win = syn.make_window(F, P, 640, 480, seed=1)
g = capi.HipWindow(capi.default_pba_options()); syn.load_window(g, win); g.snapshot(); g.restore()
t = {k: round(g.time_kernel(k, 20), 1) for k in ("sweep_linearize", "sweep_energy", "schur", "assemble_solve")}
g.optimize_repeated(7)
t0 = time.perf_counter(); n, _ = g.optimize_repeated(28); dt = time.perf_counter() - t0
print(F, P, os.environ.get("DSOPP_HIP_SWEEP_GROUPS"), t, "us per GN iteration: %.1f" % (dt / n * 1e6))
