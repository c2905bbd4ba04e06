"""Aggregate Gauss-Newton throughput of several independent C1 windows solved concurrently on ONE GPU (one host thread and one
HIP stream per window): the serving-side view — a single window is latency-bound (3 dependent launches per iteration), so
independent windows overlap almost freely until the machine fills."""
import ctypes
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dsopp_amd import capi, synthetic as syn  # noqa: E402


def run(n_windows, iters=700):
    win = syn.make_window(num_frames=7, num_points=2000, width=640, height=480, seed=0)
    gs = []
    hip = ctypes.CDLL("libamdhip64.so")
    for _ in range(n_windows):
        st = ctypes.c_void_p()   # one stream per window, shared by the window's pyramids (HIP maps streams to hardware queues round-robin)
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        g = capi.HipWindow(capi.default_pba_options(), stream=st.value)
        syn.load_window(g, win)
        g.snapshot()
        g.optimize_repeated(14)
        gs.append(g)
    barrier = threading.Barrier(n_windows + 1)
    done = [0] * n_windows

    def worker(i):
        barrier.wait()
        done[i], _ = gs[i].optimize_repeated(iters)
        barrier.wait()

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(n_windows)]
    for t in ts:
        t.start()
    barrier.wait()
    t0 = time.perf_counter()
    barrier.wait()
    dt = time.perf_counter() - t0
    for t in ts:
        t.join()
    for g in gs:
        g.close()
    return sum(done) / dt


if __name__ == "__main__":
    for n in (1, 2, 4, 8, 16):
        print(n, "windows:", round(run(n)), "GN it/s aggregate", flush=True)
