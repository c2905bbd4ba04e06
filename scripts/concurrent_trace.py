"""What caps the aggregate rate of several independent C1 windows on one GPU (round-4 review, weak #8)?
    python scripts/concurrent_trace.py run N [threads|async] [iters]     -> prints the aggregate GN it/s (run it under rocprofv3 --kernel-trace)
    python scripts/concurrent_trace.py analyse <kernel_trace.csv>        -> queues used, kernels in flight over time, device-side gaps
The analysis reads a rocprofv3 kernel trace: per hardware queue the busy time and the mean gap between consecutive kernels of the SAME
queue (device-side dispatch cost), and over all queues the time-weighted number of kernels in flight."""
import csv
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(n_windows, mode="threads", iters=700):
    import ctypes
    from dsopp_amd import capi, synthetic as syn
    win = syn.make_window(num_frames=7, num_points=2000, width=640, height=480, seed=0)
    hip = ctypes.CDLL("libamdhip64.so")
    gs = []
    for _ in range(n_windows):
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
        g = capi.HipWindow(capi.default_pba_options(), stream=st.value)
        syn.load_window(g, win)
        g.snapshot()
        g.optimize_repeated(14)
        gs.append(g)
    if mode == "threads":
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
        total = sum(done)
    else:
        t0 = time.perf_counter()
        total = 0
        for _ in range(iters // 7):
            for g in gs:
                g.restore()
                g.optimize_async()
            for g in gs:
                total += g.optimize_wait()[1]
        dt = time.perf_counter() - t0
    for g in gs:
        g.close()
    return total / dt


def analyse(path):
    rows = list(csv.DictReader(open(path)))
    ev = []
    by_queue = {}
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = r.get("Queue_Id", "?")
        by_queue.setdefault(q, []).append((s, e, r["Kernel_Name"]))
        ev.append((s, 1))
        ev.append((e, -1))
    # steady state: the middle half of the trace
    t_lo, t_hi = min(s for s, _ in ev), max(s for s, _ in ev)
    a, b = t_lo + (t_hi - t_lo) // 4, t_hi - (t_hi - t_lo) // 4
    ev.sort()
    level, last, area, hist = 0, None, 0.0, {}
    for t, d in ev:
        if last is not None and a <= last and t <= b:
            area += level * (t - last)
            hist[level] = hist.get(level, 0) + (t - last)
        level += d
        last = t
    span = b - a
    print(f"{len(rows)} kernel launches on {len(by_queue)} hardware queues; steady-state window {span / 1e6:.1f} ms")
    print(f"kernels in flight, time-weighted mean {area / span:.2f}; share of time at each level: " +
          ", ".join(f"{k}: {100 * v / span:.0f} %" for k, v in sorted(hist.items())))
    for q, ks in sorted(by_queue.items()):
        ks.sort()
        ks = [k for k in ks if a <= k[0] <= b]
        if len(ks) < 2:
            continue
        busy = sum(e - s for s, e, _ in ks)
        gaps = [ks[i + 1][0] - ks[i][1] for i in range(len(ks) - 1)]
        gaps_pos = [g for g in gaps if g > 0]
        print(f"  queue {q}: {len(ks)} kernels, busy {100 * busy / span:.0f} % of the window, mean kernel {busy / len(ks) / 1e3:.2f} us, "
              f"mean gap to the next kernel of the queue {sum(gaps_pos) / max(1, len(gaps_pos)) / 1e3:.2f} us (median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f})")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        n = int(sys.argv[2])
        mode = sys.argv[3] if len(sys.argv) > 3 else "threads"
        iters = int(sys.argv[4]) if len(sys.argv) > 4 else 700
        print(f"{n} windows, {mode}: {run(n, mode, iters):.0f} GN it/s aggregate", flush=True)
    else:
        analyse(sys.argv[2])
