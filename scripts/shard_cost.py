"""Per-rank cost of the landmark-sharded Gauss-Newton iteration at every shard size of BASELINE.json's configs[3] / [4], measured on
ONE GPU: the window's shard of rank 0 at world sizes 1 / 2 / 4 / 8 is loaded alone and solved through the sharded code path with a
one-rank native communicator (accumulate -> ncclAllReduce of one rank -> decide -> solve), so every kernel of the path runs at
the size it has on an N-GPU node; what a one-GPU box cannot supply is the collective's cross-device latency (the one-rank
ncclAllReduce is a local copy kernel).  rocprofv3 around this script gives the per-kernel split.

    python scripts/shard_cost.py [c3|c4] > gpurun_out/shard_cost_<w>.json
The DESIGN.md table "predicted strong scaling" is  t_N = measured(N) - t_allreduce_local + t_allreduce(N)  with t_allreduce(N) left open."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402  (torch's HIP runtime first)

torch.cuda.init()
from dsopp_amd import capi, distributed, synthetic as syn  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
F, total, seed = (7, 20000, 0) if which == "c3" else (12, 50000, 1)


def per_iteration_us(g, blocks=9, steps=14):
    g.snapshot()
    g.optimize_repeated(7)
    ts = []
    for _ in range(blocks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        done, _ = g.optimize_repeated(steps)
        ts.append((time.perf_counter() - t0) / done)
    ts.sort()
    return ts[len(ts) // 2] * 1e6


out = {"workload": which, "frames": F, "total_points": total, "rows": []}
comm = capi.Comm(0, 1, 0, lambda raw: raw)
for world in (1, 2, 4, 8):
    win = syn.make_window(num_frames=F, num_points=total, width=640, height=480, seed=seed)
    distributed.shard_window(win, 0, world)
    g = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g, win)
    row = {"world": world, "points_on_rank": win.num_points}
    if world == 1:
        row["unsharded_us"] = per_iteration_us(g)   # the plain single-GPU path (no collective, decision fused into the build)
    g.set_comm(comm)
    row["sharded_path_us"] = per_iteration_us(g)    # the path an N-rank job runs, with a one-rank (local) collective
    g.set_comm(None)
    g.close()
    out["rows"].append(row)
    print(row, file=sys.stderr)
comm.close()
print(json.dumps(out))
