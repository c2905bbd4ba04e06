"""Soak of the solve launch's in-kernel hand-over (tickets, sentinel slots): many back-to-back solves at several window sizes, then several
windows in flight on one device — a rare scheduling accident would show as the bounded wait's trap (the process aborts)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from dsopp_amd import capi, synthetic as syn

t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 60.0
wins = {}
for name, (F, P) in {"c1": (7, 2000), "c3": (7, 20000), "wide": (10, 6000), "c4": (12, 50000)}.items():
    win = syn.make_window(F, P, 640, 480, seed=3)
    g = capi.HipWindow(capi.default_pba_options()); syn.load_window(g, win); g.snapshot()
    wins[name] = g
total = 0
ref = {}
while time.time() < t_end:
    for name, g in wins.items():
        n, e = g.optimize_repeated(70 if name == "c1" else 21)
        total += n
        if name in ref:
            assert abs(e - ref[name]) <= 1e-9 * abs(ref[name]), (name, e, ref[name])
        ref[name] = e
    # several windows in flight (one stream each)
    for g in wins.values():
        g.restore()
    for g in wins.values():
        g.optimize_async()
    for name, g in wins.items():
        g.optimize_wait()
print("soak ok:", total, "Gauss-Newton iterations, energies reproduced", {k: float(v) for k, v in ref.items()})
