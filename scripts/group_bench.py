"""The single-process multi-device form of the drop-in (dsopp_hip_window_group, row g-1) on the devices given: one solver object,
one landmark shard per device, worker thread per shard, ONE collective per Gauss-Newton iteration (RCCL between distinct devices, the
in-process reducer when ids repeat or --transport local).

    python scripts/group_bench.py --devices 0,1,2,3 [--workload c3|c4|c1] [--transport auto|rccl|local|p2p] [--also p2p] [--blocks 9]

Prints ONE JSON line: whole-window GN iterations/s of the group, the same window on the first device alone, their ratio, and the
largest pose difference between the two solves (parity of the sharded solve on real hardware).  bench.py --gpus N runs this once from
rank 0 (in a subprocess with a timeout, after its own timed part) and reports the line as `window_group`."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402  (torch's HIP runtime first)

torch.cuda.init()
from dsopp_amd import capi, synthetic as syn  # noqa: E402

WORKLOADS = {"c1": (7, 2000, 0), "c3": (7, 20000, 0), "c4": (12, 50000, 1)}


def rate(g, blocks, steps):
    g.snapshot()
    g.optimize_repeated(7)
    ts = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        done, _ = g.optimize_repeated(steps)
        ts.append((time.perf_counter() - t0) / max(done, 1))
    return 1.0 / float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default="0")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--transport", default="auto", choices=("auto", "rccl", "local", "p2p"))
    ap.add_argument("--also", default="", help="a second transport measured on the same window after the first (e.g. p2p: the one-shot all-reduce "
                                               "beside RCCL); reported under `also`, its failure does not fail the run")
    ap.add_argument("--blocks", type=int, default=9)
    ap.add_argument("--steps", type=int, default=14)
    args = ap.parse_args()
    devices = [int(d) for d in args.devices.split(",") if d != ""]
    F, P, seed = WORKLOADS[args.workload]
    win = syn.make_window(num_frames=F, num_points=P, width=640, height=480, seed=seed)
    codes = {"auto": capi.TRANSPORT_AUTO, "rccl": capi.TRANSPORT_RCCL, "local": capi.TRANSPORT_LOCAL, "p2p": capi.TRANSPORT_P2P}
    names = {capi.TRANSPORT_RCCL: "rccl", capi.TRANSPORT_LOCAL: "local (in-process reducer)", capi.TRANSPORT_P2P: "p2p (one-shot all-reduce)"}
    transport = codes[args.transport]

    single = capi.HipWindow(capi.default_pba_options(), device=devices[0])
    syn.load_window(single, win)
    single_rate = rate(single, args.blocks, args.steps)
    single.restore()
    e1, it1, nv1 = single.optimize()

    group = capi.HipWindowGroup(capi.default_pba_options(), devices=devices, transport=transport)
    syn.load_window(group, win)
    group_rate = rate(group, args.blocks, args.steps)
    group.restore()
    e2, it2, nv2 = group.optimize()
    diff = 0.0
    for f in win.frames:
        (T1, ab1), (T2, ab2) = single.get_pose(f.frame_id), group.get_pose(f.frame_id)
        diff = max(diff, float(np.abs(T1 - T2).max()), float(np.abs(ab1 - ab2).max()))
    out = {"what": "dsopp_hip_window_group: ONE process, one landmark shard per device, one collective per GN iteration",
           "workload": args.workload, "frames": F, "total_points": P, "devices": devices,
           "transport": names.get(group.transport, str(group.transport)),
           "value": group_rate, "unit": "GN iterations/s", "same_workload_1gpu": single_rate, "speedup": group_rate / single_rate,
           "us_per_iteration": 1e6 / group_rate, "iterations": [int(it1), int(it2)], "valid_residuals": [int(nv1), int(nv2)],
           "relative_energy_difference": abs(e1 - e2) / abs(e1), "max_pose_difference_vs_single_window": diff}
    group.close()
    if args.also:
        try:
            g2 = capi.HipWindowGroup(capi.default_pba_options(), devices=devices, transport=codes[args.also])
            syn.load_window(g2, win)
            r2 = rate(g2, args.blocks, args.steps)
            g2.restore()
            e3, it3, nv3 = g2.optimize()
            d2 = max(float(np.abs(single.get_pose(f.frame_id)[0] - g2.get_pose(f.frame_id)[0]).max()) for f in win.frames)
            out["also"] = {"transport": names.get(g2.transport, str(g2.transport)), "value": r2, "us_per_iteration": 1e6 / r2, "speedup": r2 / single_rate,
                           "iterations": int(it3), "max_pose_difference_vs_single_window": d2}
            g2.close()
        except Exception as exc:  # noqa: BLE001
            out["also"] = {"transport": args.also, "error": repr(exc)}
    single.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
