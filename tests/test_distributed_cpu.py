"""N > 1 path on CPU (world_size 2, gloo): the landmark sharding + sum-collective scheme the GPU path uses.

The HIP kernels cannot run here, so the per-rank partial systems come from the CPU oracle (checker stand-in); what is
tested is the scheme itself — shard bounds, linearity of every quantity the C-ABI all-reduces (reduced normal equations,
energy / valid counts), and the variable-length gather behind the global outlier threshold."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _priors(K):
    """evaluateLinearSystemPrior for the synthetic window: frame 0 fixed (1e16), affine priors on the others (ab = 0)"""
    P = np.zeros((K, K))
    P[:8, :8] = 1e16 * np.eye(8)
    for f in range(1, K // 8):
        P[8 * f + 6, 8 * f + 6] = 1e12
        P[8 * f + 7, 8 * f + 7] = 1e8
    return P


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    from dsopp_amd import distributed, synthetic as syn
    from oracle import pyoracle as po
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        po.set_threads(1)
        win = syn.make_window(num_frames=3, num_points=150, width=160, height=120, seed=9)
        full = None
        if rank == 0:
            wf = po.OracleWindow(po.default_pba_options())
            syn.load_window(wf, win)
            wf.begin()
            e_full, n_full = wf.calculate_energy()
            wf.linearize()
            full = (e_full, n_full) + wf.get_system()
            energies_full = np.concatenate([wf.get_residuals(a.frame_id, b.frame_id)["energy"] for a in win.frames for b in win.frames
                                            if a.frame_id != b.frame_id])
        n_before = [len(f.uv) for f in win.frames]
        distributed.shard_window(win, rank, world)
        for f, n in zip(win.frames, n_before):
            lo, hi = distributed.shard_bounds(n, rank, world)
            assert len(f.uv) == hi - lo
        w = po.OracleWindow(po.default_pba_options())
        syn.load_window(w, win)
        w.begin()
        e, n = w.calculate_energy()
        w.linearize()
        Hpp, bpp, Hsc, bsc = w.get_system()
        K = len(bpp)
        packed = np.concatenate([(Hpp - _priors(K)).ravel(), bpp, Hsc.ravel(), bsc, [e, n]])
        summed = distributed.allreduce_numpy(dist, torch, packed)
        energies = np.concatenate([w.get_residuals(a.frame_id, b.frame_id)["energy"] for a in win.frames for b in win.frames
                                   if a.frame_id != b.frame_id])
        gathered = distributed.gather_variable(dist, torch, energies, rank, world)
        if rank == 0:
            e_full, n_full, Hpp_f, bpp_f, Hsc_f, bsc_f = full
            o = 0
            Hpp_s = summed[o:o + K * K].reshape(K, K) + _priors(K)
            o += K * K
            bpp_s = summed[o:o + K]
            o += K
            Hsc_s = summed[o:o + K * K].reshape(K, K)
            o += K * K
            bsc_s = summed[o:o + K]
            o += K
            ok = (np.abs(Hpp_s - Hpp_f).max() <= 1e-9 * np.abs(Hpp_f).max() and np.abs(bpp_s - bpp_f).max() <= 1e-9 * np.abs(bpp_f).max()
                  and np.abs(Hsc_s - Hsc_f).max() <= 1e-9 * np.abs(Hsc_f).max() and np.abs(bsc_s - bsc_f).max() <= 1e-9 * np.abs(bsc_f).max()
                  and abs(summed[o] - e_full) <= 1e-10 * e_full and int(round(summed[o + 1])) == n_full
                  and len(gathered) == len(energies_full) and np.allclose(np.sort(gathered), np.sort(energies_full), rtol=1e-12, atol=0))
            out.put(bool(ok))
    finally:
        dist.destroy_process_group()


def test_sharded_sums_equal_full_window():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    assert out.get(timeout=5) is True


def test_shard_bounds_cover_everything():
    from dsopp_amd.distributed import shard_bounds
    for n in (0, 1, 7, 285, 2000, 20001):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
