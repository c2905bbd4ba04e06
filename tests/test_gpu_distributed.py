"""N > 1 path of the HIP library on a real GPU: two processes share cuda:0, each owns one landmark shard of the window and
the C-ABI's all-reduce callback runs torch.distributed over gloo on the device buffers (the bench uses the same callback
over nccl = RCCL with one GPU per rank; two ranks cannot share one device under RCCL).  The sharded solve must reproduce
the single-window solve: the K x K sums, energies and the global outlier threshold are all combined across ranks."""
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lm_mode, out):
    import torch
    import torch.distributed as dist
    from dsopp_amd import capi, distributed, synthetic as syn
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        win = syn.make_window(num_frames=4, num_points=400, width=320, height=240, seed=5)
        ref = None
        if rank == 0:
            g0 = capi.HipWindow(capi.default_pba_options())
            syn.load_window(g0, win)
            g0.set_lm_mode(lm_mode)
            ref = (g0.solve(), [g0.get_pose(f.frame_id) for f in win.frames])
            g0.close()
        distributed.shard_window(win, rank, world)
        stream = torch.cuda.Stream()
        g = capi.HipWindow(capi.default_pba_options(), device=0, stream=stream.cuda_stream)
        syn.load_window(g, win)
        g.set_allreduce(distributed.make_device_allreduce(dist, torch, stream, 0), rank, world)
        g.set_lm_mode(lm_mode)
        res = g.solve()
        poses = [g.get_pose(f.frame_id) for f in win.frames]
        # every rank must hold the same frame states
        flat = np.concatenate([np.concatenate([T, ab]) for T, ab in poses])
        summed = distributed.allreduce_numpy(dist, torch, flat)
        same = np.abs(summed - world * flat).max() <= 1e-12
        if rank == 0:
            (e0, it0, nv0), poses0 = ref
            e, it, nv = res
            ok = same and it == it0 and nv == nv0 and abs(e - e0) <= 1e-7 * abs(e0)
            for (T, ab), (T0, ab0) in zip(poses, poses0):
                ok = ok and np.abs(T - T0).max() <= 1e-7 and np.abs(ab - ab0).max() <= 1e-7
            out.put((bool(ok), float(e), float(e0), int(it), int(it0), int(nv), int(nv0)))
        g.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lm_mode", [0, 1])
def test_sharded_solve_matches_single_window(lm_mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lm_mode, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    res = out.get(timeout=5)
    assert res[0] is True, res


def test_bench_script_runs_with_two_ranks():
    """bench.py's own N > 1 path (sharding, collective callback, MAX-over-ranks timing, rank-0 JSON line) end to end: two
    ranks on this one GPU over gloo (DSOPP_BENCH_SINGLE_DEVICE=1); the driver runs the same script over RCCL with one GPU per rank"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DSOPP_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "28", "--warmup", "7"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 28 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["total_points"] == 4000 and "roofline" in d
