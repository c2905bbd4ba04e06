#!/usr/bin/env python
"""bench.py — Gauss-Newton iterations/s of the sliding-window photometric bundle adjustment on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
  step      = one Gauss-Newton iteration of the LM loop = linearize + calculateStep + calculateEnergy + accept
              (levenberg_marquardt_algorithm.hpp:77-128 of the reference)
  workload  = C1 of BASELINE.json (configs[1]): 7-keyframe window, 2000 active points per GPU, 640x480, full clique,
              production solver settings (7 LM iterations, lambda = 1e-5, Huber 20, force_accept), synthetic scene.
              All inputs (images, landmarks, statuses) are resident in HBM before the timed region starts.
  N > 1     = landmarks sharded across ranks, frames and images replicated; ONE RCCL all-reduce per GN iteration over
              [combined system | energy scalars], enqueued by the library itself (dsopp_hip_comm: ncclAllReduce on the
              window's stream; --comm torch routes it through a torch.distributed callback instead).
              Headline at N > 1 = STRONG scaling on BASELINE.json configs[3] (C3: 7 KF, 20 000 points in TOTAL, sharded): value =
              GN iterations of the whole window / wall time, "scaling": "strong"; rank 0 first times the SAME window alone on its
              GPU ("same_workload_1gpu") and the line carries speedup = value / same_workload_1gpu.  C1 weak scaling (2000 points
              PER GPU) and C4 strong scaling (12 KF / 50 000) ride along as extras ("weak_scaling_c1", "strong_scaling"), and so
              does "window_group": the same C3 window driven by ONE process over all N devices through dsopp_hip_window_group
              (scripts/group_bench.py, started by rank 0 after the timed part) — the form the reference's single solver object takes.
              `--workload c1|c3|c4` / `--scaling` select another headline explicitly.  Without a launcher `--gpus N` spawns its own
              ranks (torch.distributed.run) and refuses to run when the node has fewer than N GPUs.
Extra objects on the same line: roofline (linearisation sweep kernel, measured live with HIP events on the library's
stream) and cpu_baseline (the oracle = CPU port of the reference algorithm, timed on the host cores of this box).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_linearize(P, F, s):
    """per point 12 words in; per (point,target) 8 px x 4 texels x 3 ch = 96 words gathered + 1 status in
    + 3 out; per point K+2 words out."""
    T, K = F - 1, 8 * F
    return s * P * (12 + T * 100 + (K + 2))


def algorithmic_bytes_energy(P, F, s):
    return s * P * (12 + (F - 1) * 35)


WORKLOADS = {
    # name: (frames, total points or None = per-GPU count, scaling, description) — BASELINE.json configs[1], [3], [4]
    "c1": (7, None, "weak", "C1: 7-KF window, 2000 active points per GPU"),
    "c3": (7, 20000, "strong", "C3: 7-KF window, 20000 active points in total, landmark-sharded"),
    "c4": (12, 50000, "strong", "C4: 12-KF window, 50000 active points in total, landmark-sharded"),
}


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU the way the driver does
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...) and pass its exit code on.
    Fails loudly when the node has fewer GPUs than ranks — never a silent single-GPU run."""
    import socket
    import subprocess
    import torch
    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_dev < args.gpus and os.environ.get("DSOPP_BENCH_SINGLE_DEVICE") != "1":
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {n_dev} GPU(s); one rank per GPU is required "
                         "(RCCL refuses two ranks on one device) — not running a smaller job under the same label")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


class Job:
    """rank / world / process group / native communicator of this process"""

    def __init__(self, args, torch):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # self-test hook (tests/test_gpu_distributed.py): all ranks on one device + gloo, to run the N > 1 plumbing of this script
        # on a single-GPU box (RCCL refuses two ranks on one device)
        self.single_device = os.environ.get("DSOPP_BENCH_SINGLE_DEVICE") == "1"
        self.force_dist = os.environ.get("DSOPP_BENCH_FORCE_DIST") == "1"  # collective path with a single rank (self-test)
        if self.single_device:
            self.local_rank = 0
        n_dev = torch.cuda.device_count()
        if self.local_rank >= n_dev:
            raise SystemExit(f"rank {self.rank}: local rank {self.local_rank} but only {n_dev} GPU(s) visible")
        torch.cuda.set_device(self.local_rank)
        self.dist = None
        self.comm = None
        self.transport = "none"
        if self.world > 1 or self.force_dist:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            if self.single_device:
                dist.init_process_group(backend="gloo")
            else:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.local_rank))
            self.dist = dist
            self.torch = torch
            if not self.single_device and args.comm == "native":
                from dsopp_amd import capi

                def exchange(raw):
                    box = [raw]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
                # multi-rank RCCL cannot be exercised on the builder's single-GPU boxes: if the native communicator fails to come
                # up on ANY rank, every rank falls back to the torch.distributed callback (still RCCL) and the line says so
                try:
                    self.comm = capi.Comm(self.rank, self.world, self.local_rank, exchange)
                    ok = 1.0
                except Exception as exc:  # noqa: BLE001
                    print(f"[bench] rank {self.rank}: native communicator failed ({exc}); using the torch.distributed callback", file=sys.stderr)
                    self.comm, ok = None, 0.0
                flag = torch.tensor([ok], device="cuda", dtype=torch.float64)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if float(flag.item()) < 1.0:
                    if self.comm is not None:
                        self.comm.close()
                        self.comm = None
                    self.transport = "torch.distributed callback (nccl = RCCL; native communicator unavailable)"
                else:
                    self.transport = "native ncclAllReduce (dsopp_hip_comm, RCCL)"
            else:
                self.transport = "torch.distributed callback (gloo)" if self.single_device else "torch.distributed callback (nccl = RCCL)"

    def attach(self, g, stream):
        if self.dist is None:
            return
        if self.comm is not None:
            g.set_comm(self.comm)
        else:
            from dsopp_amd import distributed
            g.set_allreduce(distributed.make_device_allreduce(self.dist, self.torch, stream, self.local_rank), self.rank, self.world)

    def barrier(self, torch):
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, torch, x):
        if self.dist is None:
            return x
        t = torch.tensor([x], device="cpu" if self.single_device else "cuda", dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.comm is not None:
            self.comm.close()
        if self.dist is not None:
            self.dist.destroy_process_group()


def timed_blocks(job, torch, run_iterations, steps, min_seconds=0.2, max_blocks=200):
    """Times blocks of EXACTLY `steps` GN iterations, each bracketed by barrier + torch.cuda.synchronize() on both sides, max
    over ranks per block; blocks are repeated until min_seconds of timed work have accumulated (one block at the driver's
    --steps 20 is 1 ms: a single sample).  Returns (median block seconds, all block seconds)."""
    times = []
    n_blocks = 1
    while len(times) < n_blocks:
        job.barrier(torch)
        t0 = time.perf_counter()
        done = run_iterations(steps)
        job.barrier(torch)
        dt = job.max_over_ranks(torch, time.perf_counter() - t0)
        assert done == steps
        times.append(dt)
        if len(times) == 1:   # every rank derives the same count from the all-reduced first block
            n_blocks = int(min(max_blocks, max(1, np.ceil(min_seconds / max(dt, 1e-6)))))
    return float(np.median(times)), times


def run_sharded_workload(job, torch, capi, syn, distributed, name, args, dtype, scaling=None, solo_first=False):
    """BASELINE.json configs[3] / [4] with the TOTAL landmark count fixed and sharded over the ranks (strong: whole-window GN
    iterations / s at this world size), or C1 with 2000 points per rank (weak).  solo_first: rank 0 times the same whole window
    alone on its GPU before the sharded run, while the other ranks wait — the same-workload single-GPU rate the speedup refers to."""
    F, total, scaling_w, desc = WORKLOADS[name]
    scaling = scaling or scaling_w
    total = total if scaling == "strong" and total else (total or 2000) * (job.world if scaling == "weak" else 1)
    full = syn.make_window(num_frames=F, num_points=total, width=640, height=480, seed=1 if name == "c4" else 0)
    solo = None
    if solo_first and job.world > 1:
        if job.rank == 0:
            g1 = capi.HipWindow(capi.default_pba_options(dtype=dtype), device=job.local_rank)
            syn.load_window(g1, full)
            g1.snapshot()
            g1.optimize_repeated(7)
            ts = []
            for _ in range(9):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                done, _ = g1.optimize_repeated(14)
                ts.append((time.perf_counter() - t0) / done)
            g1.close()
            solo = 1.0 / float(np.median(ts))
        job.barrier(torch)
    win = distributed.shard_window(full, job.rank, job.world)
    stream = torch.cuda.Stream()
    g = capi.HipWindow(capi.default_pba_options(dtype=dtype), device=job.local_rank, stream=stream.cuda_stream)
    syn.load_window(g, win)
    job.attach(g, stream)
    g.snapshot()

    def run(n):
        return g.optimize_repeated(n)[0]
    run(7)
    med, times = timed_blocks(job, torch, run, 14, min_seconds=0.1, max_blocks=50)
    per_job = job.world if scaling == "weak" else 1
    out = {"workload": f"{desc}, 640x480, {win.num_points} on this rank", "frames": F, "total_points": total, "n_gpus": job.world,
           "gn_iterations_per_s": per_job * 14 / med, "ms_per_iteration": med / 14 * 1e3, "timed_blocks": len(times), "scaling": scaling}
    if solo is not None:
        out["same_workload_1gpu"] = solo
        out["speedup"] = out["gn_iterations_per_s"] / solo
    g.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=140)
    ap.add_argument("--warmup", type=int, default=14)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="c1 (default at --gpus 1: the metric's configuration), c3 (default at --gpus N > 1: strong scaling), c4")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: --points per GPU (default for c1); strong: the window's total is fixed and sharded (default for c3 / c4)")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--points", type=int, default=None, help="active points per GPU (weak) or in total (strong)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--comm", default="native", choices=["native", "torch"],
                    help="N > 1 exchange: native = the library's own ncclAllReduce (dsopp_hip_comm), torch = torch.distributed callback")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the large-window roofline, the tracker (C2) timing and the other extras")
    ap.add_argument("--no-group", action="store_true", help="N > 1: skip the single-process window-group measurement (row g-1)")
    ap.add_argument("--group-timeout", type=int, default=150, help="seconds the window-group process may take")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)

    # stdout carries exactly ONE line, the JSON of rank 0: everything else any library of this process writes to file descriptor 1
    # (RCCL prints a five-line version banner from C stdio when a communicator comes up, flushed at exit, i.e. BEHIND the JSON
    # line) is sent to stderr; the JSON goes to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from dsopp_amd import capi, distributed, synthetic as syn

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}: refusing to report one under the other's label")
    job = Job(args, torch)
    world, rank = job.world, job.rank
    if args.workload is None:
        # N = 1: the metric's own configuration.  N > 1: north_star asks for STRONG scaling — BASELINE.json configs[3], the 7-KF /
        # 20 000-point window sharded over the ranks, with the same window's single-GPU rate measured beside it
        args.workload = "c1" if world == 1 or args.scaling == "weak" else "c3"

    Fw, total_w, scaling_w, desc = WORKLOADS[args.workload]
    F = args.frames or Fw
    scaling = args.scaling or scaling_w
    if scaling == "weak":
        P = args.points or 2000
        total_points = P * world
    else:
        total_points = args.points or total_w or 2000
        P = total_points // world
    full_win = syn.make_window(num_frames=F, num_points=total_points, width=args.width, height=args.height, seed=0)
    dtype = capi.F64 if args.dtype == "f64" else capi.F32
    same_workload_1gpu = None
    if world > 1 and scaling == "strong":
        # rank 0 alone, the whole window on its one GPU, no collective: the rate the sharded run is compared with (same process,
        # same box, same build); the other ranks wait at the barrier
        if rank == 0:
            g1 = capi.HipWindow(capi.default_pba_options(dtype=dtype), device=job.local_rank)
            syn.load_window(g1, full_win)
            g1.snapshot()
            g1.optimize_repeated(max(args.warmup, 7))
            ts = []
            for _ in range(15):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                done1, _ = g1.optimize_repeated(args.steps)
                ts.append((time.perf_counter() - t0) / done1)
            g1.close()
            same_workload_1gpu = 1.0 / float(np.median(ts))
        job.barrier(torch)
    win = distributed.shard_window(full_win, job.rank, job.world)
    P_local = win.num_points

    stream = torch.cuda.Stream()
    opts = capi.default_pba_options(dtype=dtype)
    g = capi.HipWindow(opts, device=job.local_rank, stream=stream.cuda_stream)
    syn.load_window(g, win)
    job.attach(g, stream)
    g.snapshot()

    def run_iterations(n_target):
        """executes exactly n_target GN iterations as repeated LM loops from the snapshot ({restore; optimize} per solve,
        every solve with its own read-back + stream sync; the loop itself runs inside the library so that the gap between
        two solves is not Python's call overhead); returns iterations done"""
        done, _ = g.optimize_repeated(n_target)
        if done < n_target:
            raise RuntimeError("LM loop made no progress")
        return done

    warmup = max(args.warmup, 7)
    run_iterations(warmup)
    block_s, block_times = timed_blocks(job, torch, run_iterations, args.steps)
    steps_done = args.steps

    # ---- live per-kernel timing (HIP events on the library's stream) -> roofline of the dominant kernel
    g.set_profiling(True)
    run_iterations(35)
    prof = g.get_profile()
    g.set_profiling(False)
    s_bytes = 8 if dtype == capi.F64 else 4
    # dominant HBM-bound kernel: average over 200 back-to-back launches inside ONE HIP event pair on the library's stream
    # (an event pair around a single ~6 us launch adds ~5 us; the per-class numbers in "kernels" carry that overhead).
    # sweep_linearize_loop is the variant the LM loop actually runs (back-substitution of the pending step fused in);
    # sweep_linearize is the plain linearisation that opens a solve.
    g.restore()
    isolated = {k: g.time_kernel(k, 200) for k in ("sweep_linearize_loop", "sweep_linearize", "sweep_energy", "schur", "assemble_solve")}
    lin_avg_s = isolated["sweep_linearize_loop"] * 1e-6
    b_lin = algorithmic_bytes_linearize(P_local, F, s_bytes)
    b_en = algorithmic_bytes_energy(P_local, F, s_bytes)
    achieved = b_lin / lin_avg_s / 1e9 if lin_avg_s > 0 else 0.0
    kernels = {k: {"avg_us": (v[0] / v[1] * 1e3 if v[1] else 0.0), "launches": v[1]} for k, v in prof.items() if v[1]}
    dominant = max(kernels.items(), key=lambda kv: kv[1]["avg_us"] * kv[1]["launches"])[0]

    # HBM traffic of the same launch from the TCC counters: collected by scripts/pmc_traffic.py (rocprofv3, FETCH_SIZE and
    # WRITE_SIZE in separate --pmc passes, calibrated on copies AND on a 64-byte-segment gather in the same pass) and committed
    # under profiles/
    traffic, traffic_detail = load_pmc_traffic(F, P, args)

    extras = {}
    if not args.no_extras and args.workload == "c1" and scaling == "weak":
        # BASELINE.json configs[3] / [4] at THIS world size (total landmark count fixed, sharded): the strong-scaling figures
        extras["strong_scaling"] = {name: run_sharded_workload(job, torch, capi, syn, distributed, name, args, dtype, solo_first=True)
                                    for name in ("c3", "c4")}
    if not args.no_extras and world > 1 and args.workload == "c3" and scaling == "strong":
        # the other two configurations at this world size: C1 weak (2000 points per GPU, the N = 1 headline's window on every rank)
        # and C4 strong (12 KF / 50 000 points) with its own same-workload single-GPU rate
        extras["weak_scaling_c1"] = run_sharded_workload(job, torch, capi, syn, distributed, "c1", args, dtype, scaling="weak")
        extras["strong_scaling"] = {"c4": run_sharded_workload(job, torch, capi, syn, distributed, "c4", args, dtype, solo_first=True)}
    if world > 1 and not args.no_group:
        extras_group = run_window_group(job, args)
        if extras_group is not None:
            extras["window_group"] = extras_group
    if rank == 0 and world == 1 and not args.no_extras:
        g.restore()
        extras["stages"] = run_stage_table(g, win, syn, args)
        # the whole per-keyframe call of the drop-in: solve() = LM loop + relinearizeSystem + pose covariances (pinv of the
        # K x K system on the host) + updatePointStatuses (device-side radix select of the 3rd quartile)
        ts = []
        for _ in range(10):
            g.restore()
            t0 = time.perf_counter()
            g.solve()
            ts.append(time.perf_counter() - t0)
        extras["full_solve"] = {"gpu_ms": float(np.median(ts) * 1e3), "what": "dsopp_hip_window_solve on the C1 window (7 LM iterations + uncertainty + point statuses)"}
        g.restore()
        extras["roofline_large"] = run_large_window_roofline(capi, syn, dtype, s_bytes)
        extras["f32_mode"] = run_f32_mode(capi, syn, win, F, P_local)
        # ... and where fp32 texels halve the gathered bytes: the 12 KF / 50 000-point window (not a headline: the reference's scalar is double)
        f32_large = run_large_window_roofline(capi, syn, capi.F32, 4)
        extras["f32_mode"]["large_window"] = {k: f32_large[k] for k in ("workload", "kernels_isolated_avg_us", "algorithmic_bytes_per_launch", "achieved", "frac",
                                                                      "gn_iterations_per_s")}
        extras["tracker"] = run_tracker_timing(capi, syn, torch, no_cpu=args.no_cpu)
        # the same with the 4 pyramid levels the production camera requests (src/sensors/camera/src/camera.cpp:43-45)
        t4 = run_tracker_timing(capi, syn, torch, no_cpu=args.no_cpu, levels=4)
        extras["tracker_4_levels"] = {k: t4[k] for k in ("metric", "ms_per_frame", "pyramid_ms", "lm_iterations_per_frame", "success", "rmse_per_level",
                                                         "cpu_port_ms_per_frame", "cpu_port_pose_difference") if k in t4}
        extras["depth_estimation"] = run_depth_estimation_timing(capi, syn, args)
        extras["landmark_activation"] = run_landmark_activation_timing(capi, syn, args)
        extras["concurrent_windows"] = run_concurrent_windows(capi, syn, torch, win)
        extras["keyframe_step"] = run_keyframe_step_timing(capi, syn)
        extras["tick_sequence"] = run_tick_sequences(torch, syn, args)
        extras["dense_window"] = run_dense_window(capi, syn)
        extras["roofline_large_fullres"] = run_fullres_windows(capi, syn)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu_baseline = run_cpu_baseline(args, F, P, win, syn)

    if rank == 0:
        # the committed rocprofv3 figure of the kernel AS THE LOOP RUNS IT (kernel_stats csv of the same command under profiles/) rides
        # along as `frac_profile`; `roofline.frac` itself is the event-bracketed measurement of this run
        prof_roof = profile_roofline(b_lin) if (F, P_local, args.dtype) == (7, 2000, "f64") else None
        in_loop = (prof_roof or {}).get("in_loop")
        frac_events = achieved / HBM_PEAK_GBS
        ms_per_step = block_s / steps_done * 1e3
        per_job = world if scaling == "weak" else 1   # weak: every rank advances its own 2000-point share of the window
        line = {
            "metric": "Gauss-Newton iters/sec (7-KF window, 2k active pts)",
            "value": per_job * steps_done / block_s,
            "unit": "GN iterations/s",
            "n_gpus": world,
            "steps": steps_done,
            "warmup": args.warmup,  # as given; a warm-up shorter than one LM solve (7 iterations) is rounded up to one:
            "warmup_effective": warmup,
            "ms_per_step": ms_per_step,
            "timed_region_s": block_s,   # the median timed block: `steps` iterations between two barrier + synchronize pairs
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "timing": {"what": f"median over {len(block_times)} timed blocks of {steps_done} GN iterations each (barrier + synchronize on both sides "
                               "of every block, max over ranks per block); inside a block the LM solves (7 iterations each, the "
                               "window restored in front of every solve) are enqueued back to back, their results fetched with one copy at the "
                               "end of the block: no host synchronisation between solves",
                       "blocks": len(block_times), "block_ms_min": min(block_times) * 1e3, "block_ms_median": block_s * 1e3,
                       "block_ms_max": max(block_times) * 1e3},
            "config": {"workload": f"{desc} ({total_points} total, {P_local} on rank 0), {args.width}x{args.height}, full clique, "
                                   "production LM settings", "name": args.workload,
                       "frames": F, "points_per_gpu": P, "total_points": total_points,
                       "parallelism": f"landmark-sharded x{world}" if world > 1 else "single GPU",
                       "exchange": job.transport, "ranks": world,
                       # the launcher's world size above; what the communicator itself counts (ncclCommCount) when the native one is in use
                       "ranks_comm": job.comm.size() if job.comm is not None else None},
            # `frac` / `achieved` / `avg_launch_us` are THIS run's measurement: HIP events on the library's stream around 200 back-to-back
            # launches of the kernel as the loop configures it.  `frac_profile` is the committed rocprofv3 --kernel-trace --stats average of
            # the kernel inside the bench loop (profiles/, file and commit named) — the cross-check, never the headline figure.
            "roofline": {"bound": "hbm", "kernel": "sweepKernel<double,LIN,FEJ,HUBER> (linearisation sweep as the LM loop launches it)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac_events,
                         "frac_source": "HIP events of this run on the library's stream, 200 back-to-back launches",
                         "frac_profile": in_loop["frac"] if in_loop else None,
                         "profile_source": in_loop["source"] if in_loop else None,
                         "profile_commit": in_loop["committed_at"] if in_loop else None,
                         "traffic": traffic, "traffic_detail": traffic_detail,
                         "algorithmic_bytes_per_launch": b_lin, "avg_launch_us": lin_avg_s * 1e6,
                         "mfma": load_mfma_utilisation(),
                         # whole Gauss-Newton iteration against the same roof: SURVEY.md §8d's B_gn = B_lin + B_en over the
                         # driver-timed time per iteration (launch gaps, reduction and dense solve included)
                         "iteration_frac": (b_lin + b_en) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "iteration_algorithmic_bytes": b_lin + b_en,
                         # the same 200 isolated launches under `rocprofv3 --kernel-trace --stats` (scripts/profile_target.py c1_isolated,
                         # committed csv): rocprof's per-dispatch duration of back-to-back launches includes each launch's ramp, which
                         # overlaps the previous kernel's tail, so its average sits ~10 % above the event-bracketed wall time per launch
                         "profile": prof_roof,
                         "what_the_fraction_is": "algorithmic bytes (no texel reuse counted) over kernel time against the 8 TB/s HBM3E peak; "
                                                 "the C1 working set (69 MB of texels) fits the 256 MB Infinity Cache, whose hits the TCC_EA "
                                                 "counters behind `traffic` include: a latency-bound gather, not DRAM streaming",
                         "opening_linearisation": {"avg_launch_us": isolated["sweep_linearize"],
                                                   "frac": b_lin / (isolated["sweep_linearize"] * 1e-6) / 1e9 / HBM_PEAK_GBS},
                         "energy_sweep": {"algorithmic_bytes_per_launch": b_en, "avg_launch_us": isolated["sweep_energy"]}},
            "kernels": kernels,
            "kernels_isolated_avg_us": isolated,
            "dominant_kernel_by_total_time": dominant,
        }
        if same_workload_1gpu is not None:
            # strong scaling: the same window on ONE GPU of this node (rank 0 alone, before the sharded run) and the ratio
            line["same_workload_1gpu"] = {"value": same_workload_1gpu, "unit": "GN iterations/s",
                                          "what": f"{desc}: the whole window ({total_points} points) on rank 0's GPU alone, no collective, "
                                                  f"median of 15 blocks of {steps_done} iterations"}
            line["speedup"] = line["value"] / same_workload_1gpu
        line.update(extras)
        if cpu_baseline is not None:
            line["cpu_baseline"] = cpu_baseline
            line["speedup_vs_cpu_port"] = line["value"] / cpu_baseline["value"]
        headline, extras_path = emit(line)
        os.write(json_fd, (json.dumps(headline, separators=(",", ":")) + "\n").encode())
    g.close()
    job.close()


HEADLINE_MAX_BYTES = 4096   # the driver keeps an 8 KB tail of stdout: the ONE stdout line must fit it with room to spare

HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "timed_region_s", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "speedup", "speedup_vs_cpu_port")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us",
                 "frac_source", "frac_profile", "profile_source", "profile_commit", "iteration_frac")


def make_headline(line):
    """The compact form of the result that goes to stdout: the driver's contract keys + roofline + cpu_baseline and nothing else.  Everything
    the run measured beside the headline (stage table, tracker, tick sequences, large windows, ...) stays in `line`, which emit() writes to
    bench_extras.json.  Raises when the result would not fit HEADLINE_MAX_BYTES: a line the driver cannot parse is worth nothing."""
    head = {k: line[k] for k in HEADLINE_KEYS if k in line}
    cfg = line.get("config") or {}
    head["config"] = {k: cfg[k] for k in ("workload", "name", "frames", "points_per_gpu", "total_points", "parallelism", "exchange", "ranks") if k in cfg}
    roof = line.get("roofline") or {}
    head["roofline"] = {k: roof[k] for k in ROOFLINE_KEYS if k in roof}
    tm = line.get("timing") or {}
    if tm:
        head["timing"] = {k: tm[k] for k in ("blocks", "block_ms_min", "block_ms_median", "block_ms_max") if k in tm}
    if "same_workload_1gpu" in line:
        head["same_workload_1gpu"] = line["same_workload_1gpu"].get("value")
    cpu = line.get("cpu_baseline")
    if cpu is not None:
        head["cpu_baseline"] = {k: cpu[k] for k in ("value", "unit", "cores", "kind", "sample") if k in cpu}
        head["cpu_baseline"]["sample"] = str(head["cpu_baseline"].get("sample", ""))[:400]
    # BASELINE.json's second metric, when the run measured it (N = 1 with extras): frame-tracking ms / frame at 1280x1024
    trk = {}
    for key, name in (("tracker", "5_levels"), ("tracker_4_levels", "4_levels")):
        if isinstance(line.get(key), dict) and "ms_per_frame" in line[key]:
            trk[name] = line[key]["ms_per_frame"]
    if trk:
        head["frame_tracking_ms_per_frame_1280x1024"] = trk
    if "extras_file" in line:
        head["extras_file"] = line["extras_file"]
    text = json.dumps(head, separators=(",", ":"))
    if len(text) > HEADLINE_MAX_BYTES:
        raise RuntimeError(f"bench headline is {len(text)} bytes (> {HEADLINE_MAX_BYTES}): trim it, the driver keeps an 8 KB tail only")
    return head


def emit(line):
    """writes the full result to bench_extras.json (repo root; also gpurun_out/ when that exists, so that it travels back from a GPU box) and to
    stderr, and returns (headline, path)"""
    line["extras_file"] = "bench_extras.json"
    head = make_headline(line)
    full = json.dumps(line, indent=1)
    path = os.path.join(ROOT, "bench_extras.json")
    for p in (path, os.path.join(ROOT, "gpurun_out", "bench_extras.json")):
        try:
            if os.path.isdir(os.path.dirname(p)):
                with open(p, "w") as fh:
                    fh.write(full + "\n")
        except OSError as exc:
            print(f"[bench] could not write {p}: {exc}", file=sys.stderr)
    print("[bench] full result (also in bench_extras.json):", file=sys.stderr)
    print(json.dumps(line), file=sys.stderr)
    return head, path


def run_window_group(job, args):
    """Row g-1 on this node: after the timed part rank 0 starts ONE more process that drives all N devices through the single-process
    window group (scripts/group_bench.py: one solver object, worker thread per device, RCCL between them) on the C3 window, while the
    other ranks wait on the rendezvous store — a host-side wait, their GPUs stay idle.  Reported next to the headline, never instead
    of it; a failure or a timeout is reported as such and does not fail the bench."""
    import datetime
    import subprocess
    key = "dsopp_window_group_done"
    try:
        store = job.dist.distributed_c10d._get_default_store()
    except Exception:  # noqa: BLE001
        store = None
    out = None
    if job.rank == 0:
        devices = ",".join("0" if job.single_device else str(d) for d in range(job.world))
        cmd = [sys.executable, os.path.join(ROOT, "scripts", "group_bench.py"), "--devices", devices, "--workload", "c3", "--blocks", "7", "--also", "p2p"]
        t0 = time.perf_counter()
        try:
            # (the one-shot peer-to-peer all-reduce across distinct devices is an opt-in experiment of the library: this measurement opts in)
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.group_timeout, env=dict(os.environ, DSOPP_HIP_P2P_EXPERIMENTAL="1"))
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and lines:
                out = json.loads(lines[-1])
            else:
                out = {"error": f"exit code {r.returncode}", "stderr_tail": r.stderr[-600:]}
        except subprocess.TimeoutExpired:
            out = {"error": f"no result within {args.group_timeout} s (process killed)"}
        except Exception as exc:  # noqa: BLE001
            out = {"error": repr(exc)}
        out["command"] = " ".join(cmd[1:])
        out["wall_s"] = time.perf_counter() - t0
        if store is not None:
            store.set(key, "1")
    elif store is not None:
        try:
            store.wait([key], datetime.timedelta(seconds=args.group_timeout + 120))
        except Exception:  # noqa: BLE001
            pass
    if store is None:
        job.barrier(job.torch)
    return out


PROFILE_ROUNDS = ("r06", "r05", "r04")   # newest first: a figure read from profiles/ always names the file (and its commit) it came from


def newest_profile(name):
    """relative path (under profiles/) of the newest round's copy of `name`, or None"""
    for rnd in PROFILE_ROUNDS:
        if os.path.exists(os.path.join(ROOT, "profiles", rnd, name)):
            return os.path.join(rnd, name)
    return None


def load_profile_kernel_avg_us(csv_name, needle):
    """average duration (us) of the kernel whose name contains `needle` in a committed rocprofv3 kernel_stats csv under profiles/"""
    import csv
    path = os.path.join(ROOT, "profiles", csv_name)
    if not os.path.exists(path):
        return None
    try:
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if needle in row["Name"]:
                    return float(row["AverageNs"]) / 1e3, int(row["Calls"])
    except (OSError, KeyError, ValueError):
        pass
    return None


def _committed_at(rel_path):
    """commit and date a tracked profile file was last written at (None outside a git checkout, e.g. on the GPU box): a reader can see how
    old the figure `roofline.frac` is priced with is — the kernel may have changed since"""
    import subprocess
    # the profile scripts stamp the commit of the build they measured next to their output (there is no .git on the GPU box)
    stamp = os.path.join(ROOT, os.path.dirname(rel_path), "BUILD_COMMIT.txt")
    if os.path.exists(stamp):
        with open(stamp) as fh:
            return fh.read().strip() or None
    try:
        r = subprocess.run(["git", "log", "-1", "--format=%h %cI", "--", rel_path], cwd=ROOT, capture_output=True, text=True, timeout=10)
        return r.stdout.strip() or None
    except Exception:  # noqa: BLE001
        return None


def profile_roofline(b_lin):
    out = {}
    for key, csv_name in (("isolated_launches", "c1_isolated_kernel_stats.csv"), ("in_loop", "c1_kernel_stats.csv")):
        # newest first: since the back-substitution moved into the solve launch the loop runs the plain linearisation variant; before
        # that the BACKSUB variant (whose template argument list lost an argument in round 4)
        for alt, needle in ((os.path.join("r06", csv_name), "sweepKernel<double, true, true, true, false, false>"),
                            (os.path.join("r05", csv_name), "sweepKernel<double, true, true, true, false, false>"),
                            (os.path.join("r04", csv_name), "sweepKernel<double, true, true, true, false, false>"),
                            (os.path.join("r04", csv_name), "sweepKernel<double, true, true, true, true, false>"),
                            (os.path.join("r03", "c_" + csv_name), "sweepKernel<double, true, true, true, true, false, false>")):
            r = load_profile_kernel_avg_us(alt, needle)
            if r:
                out[key] = {"source": f"profiles/{alt}", "avg_us": r[0], "calls": r[1], "frac": b_lin / (r[0] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                            "committed_at": _committed_at(os.path.join("profiles", alt))}
                break
    return out or None


def load_mfma_utilisation():
    """f64 matrix-core figures of the kernels that use them, from the committed SQ counter passes (scripts/mfma_utilisation.py ->
    profiles/r04/mfma_utilisation.json): instruction counts, busy cycles of the matrix pipe against the kernel's cycles, achieved
    TFLOP/s against AMD's 78.6 TFLOP/s fp64-matrix figure for MI355X (MI355X_MICROARCH.md lists no fp64 row)"""
    rel = newest_profile("mfma_utilisation.json")
    if rel is None:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", rel)) as fh:
            d = json.load(fh)
        d["source"] = f"profiles/{rel}"
        d["committed_at"] = _committed_at(os.path.join("profiles", rel))
        return d
    except (OSError, ValueError):
        return None


def load_pmc_traffic(F, P, args):
    """per-launch HBM bytes of the in-loop linearisation sweep from the committed counter run (profiles/, newest round first)"""
    if (F, P, args.width, args.height, args.dtype, args.workload) != (7, 2000, 640, 480, "f64", "c1"):
        return None, None
    for name in (os.path.join("r06", "pmc_traffic_c1.json"), os.path.join("r05", "pmc_traffic_c1.json"), os.path.join("r04", "pmc_traffic_c1.json"), os.path.join("r03", "pmc_traffic_c1.json"), "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        pmc_file = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(pmc_file):
            continue
        try:
            with open(pmc_file) as fh:
                pmc = json.load(fh)
            per = pmc["per_launch_bytes"]
            k = per.get("sweep_linearize_loop") or per["sweep_linearize"]
            en = per.get("sweep_energy")
            detail = {"fetch_bytes": k["fetch"], "write_bytes": k["write"], "source": f"profiles/{name}",
                      "energy_sweep_bytes": None if en is None else en["total"],
                      "kernel": "sweep_linearize_loop" if "sweep_linearize_loop" in per else "sweep_linearize",
                      "fetch_bytes_per_count": (pmc.get("calibration") or {}).get("fetch_gather_bytes_per_count",
                                                                                  pmc["passes"]["FETCH_SIZE"]["bytes_per_count"]),
                      "write_bytes_per_count": pmc["passes"]["WRITE_SIZE"]["bytes_per_count"],
                      "calibration": pmc.get("calibration"),
                      "note": "per launch of the same kernel on the same window (scripts/pmc_traffic.py); TCC_EA0 counters = "
                              "XCD-L2 fabric requests, Infinity-Cache hits included; fetch counts converted with the unit a texel "
                              "gather of known geometry measured in the same pass (a wide stream is tallied at half that)"}
            return k["total"], detail
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def run_stage_table(g, win, syn, args, repeats=30):
    """Per-stage wall times of the blocking stage entry points, mirroring the stage list of the reference's benchmark
    (test/performance/benchmarks/energy/photometric_bundle_adjustment_benchmark.cpp:252-264: Linearize, CalculateStep,
    CalculateEnergy, AcceptStep, RejectStep), GPU (every call ends with its read-back + synchronisation) beside the CPU
    port on the same C1 window.  In the fused solve loop these stages are not separate calls (see "kernels")."""
    def timed(fn, setup=None, n=repeats):
        tot = 0.0
        for _ in range(n):
            if setup:
                setup()
            t0 = time.perf_counter()
            fn()
            tot += time.perf_counter() - t0
        return tot / n * 1e6

    def table(w, n):
        w.begin()
        w.calculate_energy()
        out = {"calculate_energy": timed(w.calculate_energy, n=n)}
        out["linearize"] = timed(w.linearize, n=n)
        w.linearize()
        out["calculate_step"] = timed(lambda: w.calculate_step(1e-5), n=n)

        def prep():
            w.calculate_step(1e-5)
            w.calculate_energy()
        out["reject_step"] = timed(w.reject_step, setup=prep, n=n)
        out["accept_step"] = timed(w.accept_step, setup=lambda: (w.linearize(), prep()), n=max(3, n // 6))
        return out

    stages = {"gpu_us": table(g, repeats)}
    if not args.no_cpu:
        from oracle import pyoracle as po
        hw = os.cpu_count() or 1
        po.set_threads(max(1, min(hw, 8) - 1))
        o = po.OracleWindow(po.default_pba_options())
        syn.load_window(o, win)
        stages["cpu_port_us"] = table(o, 5)
        o2 = po.OracleWindow(po.default_pba_options())
        syn.load_window(o2, win)
        t0 = time.perf_counter()
        o2.solve()
        stages["cpu_port_full_solve_ms"] = (time.perf_counter() - t0) * 1e3
        stages["cpu_threads"] = max(1, min(hw, 8) - 1)
    return stages


def run_f32_mode(capi, syn, win, F, P):
    """the same C1 loop with images and residual / Jacobian rows in fp32 (fp64 accumulation): the analogue of the reference's
    -DUSE_FLOAT build.  Not the headline (the reference's default scalar is double); texels are 16 B instead of 32 B."""
    g = capi.HipWindow(capi.default_pba_options(dtype=capi.F32))
    syn.load_window(g, win)
    g.snapshot()
    g.optimize_repeated(14)
    t0 = time.perf_counter()
    done, _ = g.optimize_repeated(140)
    dt = time.perf_counter() - t0
    g.restore()
    t_lin = g.time_kernel("sweep_linearize", 200)
    b_lin = algorithmic_bytes_linearize(P, F, 4)
    out = {"gn_iterations_per_s": done / dt, "sweep_linearize_us": t_lin, "algorithmic_bytes_per_launch": b_lin,
           "achieved_GBs": b_lin / (t_lin * 1e-6) / 1e9, "frac_of_hbm_peak": b_lin / (t_lin * 1e-6) / 1e9 / HBM_PEAK_GBS}
    g.close()
    return out


def run_large_window_roofline(capi, syn, dtype, s_bytes):
    """C4-sized window on one GPU (12 keyframes, 50 000 points, 640x480): the configuration where the sweep moves enough
    bytes for the HBM fraction to mean something.  Isolated kernel, 50 back-to-back launches per event pair."""
    F, P = 12, 50000
    win = syn.make_window(num_frames=F, num_points=P, width=640, height=480, seed=1)
    g = capi.HipWindow(capi.default_pba_options(dtype=dtype))
    syn.load_window(g, win)
    g.snapshot()
    g.restore()
    out = {"workload": f"{F}-KF window, {P} points, 640x480, single GPU", "kernels_isolated_avg_us": {}}
    for k in ("sweep_linearize", "sweep_energy", "schur", "assemble_solve"):
        out["kernels_isolated_avg_us"][k] = g.time_kernel(k, 50)
    b_lin = algorithmic_bytes_linearize(P, F, s_bytes)
    t = out["kernels_isolated_avg_us"]["sweep_linearize"] * 1e-6
    out.update({"bound": "hbm", "kernel": "sweep_linearize", "algorithmic_bytes_per_launch": b_lin, "achieved": b_lin / t / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b_lin / t / 1e9 / HBM_PEAK_GBS})
    g.restore()
    g.optimize_repeated(7)
    t0 = time.perf_counter()
    n, _ = g.optimize_repeated(28)
    dt = time.perf_counter() - t0
    out["us_per_iteration"] = dt / n * 1e6
    out["gn_iterations_per_s"] = n / dt
    g.close()
    # The generator lists a frame's landmarks in the order random pixels were drawn in; since round 5 the library keeps its own order on the
    # device (every appended batch sorted into 32 x 32-pixel tiles), so this IS the spatially ordered sweep the round-4 line
    # reported as an extra.  The loop's own sweep launches (rocprofv3 timeline of one solve, committed):
    rel = newest_profile("large_loop_one_solve_timeline.csv") or os.path.join("r05", "large_loop_one_solve_timeline.csv")
    try:
        durs = []
        with open(os.path.join(ROOT, "profiles", rel)) as fh:
            for line in fh:
                if line.startswith('"sweepKernel<double, true, true, true, false, false>"'):
                    durs.append(float(line.rsplit('",', 1)[1].split(",")[0]))
        if durs:
            t_loop = sum(durs) / len(durs) * 1e-6
            out["in_loop"] = {"source": f"profiles/{rel}", "launches": len(durs), "avg_us": t_loop * 1e6, "achieved": b_lin / t_loop / 1e9,
                              "frac": b_lin / t_loop / 1e9 / HBM_PEAK_GBS, "committed_at": _committed_at(os.path.join("profiles", rel))}
    except OSError:
        pass
    out["landmark_order"] = ("internal: 32 x 32-pixel tiles per appended batch behind the C-ABI (caller order random); "
                             "A/B against the caller's order: profiles/r05/time_landmark_order_ab.txt")
    return out


def run_tracker_timing(capi, syn, torch, frames=20, no_cpu=False, levels=5):
    """C2 of BASELINE.json: coarse-to-fine direct image alignment of a new 1280x1024 frame against the last keyframe,
    5 pyramid levels — estimatePose of the tracker (monocular_tracker.cpp:179-245) on its real inputs: a 7-keyframe /
    2000-point window is bundle-adjusted, createReferenceDepthMaps runs on the device, then every new frame costs
    one pyramid build (8-bit image resident in HBM) + one dsopp_hip_aligner_estimate_pose call (all levels)."""
    W, H, L = 1280, 1024, levels  # 5 = PixelDataFrame::kMaxPyramidDepth (BASELINE's "5-level"); 4 = what the production Camera requests
    win = syn.make_window(num_frames=8, num_points=2288, width=W, height=H, seed=3)
    new_frame = win.frames.pop()  # the frame to track; the other 7 are the keyframe window (2002 points)
    g = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g, win)
    g.solve()
    maps = g.create_reference_depth_maps(L)
    kf = win.frames[-1]
    pr, pt = capi.Pyramid(W, H, L), capi.Pyramid(W, H, L)
    pr.build(kf.image_u8)
    img_dev = torch.from_numpy(new_frame.image_u8.copy()).cuda()
    torch.cuda.synchronize()
    T_ref, ab_ref = g.get_pose(kf.frame_id)
    T_init = syn.mat_to_params(new_frame.T_w_c_init)
    a = capi.HipAligner(capi.default_align_options())
    a.set_lm_path(int(os.environ.get("DSOPP_ALIGN_LM_PATH", "0")))
    rmse_last = np.full(L, 1e10)

    rl_final = rmse_last.copy()

    def one_frame():
        pt.build_device(img_dev.data_ptr())
        rl = rmse_last.copy()
        r = a.estimate_pose(kf.timestamp, T_ref, pr, maps, 1.0, ab_ref, new_frame.timestamp, pt, 1.0, win.scene.intrinsics, T_init[None, :],
                            np.zeros(2), rl)
        rl_final[:] = rl
        return r

    res = one_frame()
    torch.cuda.synchronize()
    its = 0
    per_frame = []
    for _ in range(3 * frames):   # every frame timed on its own (estimate_pose returns the pose: it is a blocking call); median
        t0 = time.perf_counter()
        res = one_frame()
        per_frame.append(time.perf_counter() - t0)
        its += res["lm_iterations"]
    torch.cuda.synchronize()
    its /= 3
    ms = float(np.median(per_frame)) * 1e3
    t0 = time.perf_counter()
    for _ in range(frames):
        pt.build_device(img_dev.data_ptr())
    torch.cuda.synchronize()
    pyr_ms = (time.perf_counter() - t0) / frames * 1e3
    m2 = g.create_reference_depth_maps(L)
    ts = []
    for _ in range(5):   # the tracker keeps one reference_frame_depth_map_ and refills it after every keyframe
        t0 = time.perf_counter()
        g.refill_reference_depth_maps(m2)
        ts.append(time.perf_counter() - t0)
    dm_ms = float(np.median(ts) * 1e3)
    t_tr = [syn.mat_to_params(np.linalg.inv(new_frame.T_w_c_gt) @ kf.T_w_c_gt)] * 2   # the tracker asks for two poses per frame
    flow = m2.mean_square_optical_flow(0, win.scene.intrinsics, t_tr)                 # (first call allocates the scratch)
    ts = []
    for _ in range(9):
        t0 = time.perf_counter()
        flow = m2.mean_square_optical_flow(0, win.scene.intrinsics, t_tr)
        ts.append(time.perf_counter() - t0)
    flow_ms = float(np.median(ts) * 1e3)
    n0 = int((m2.get_level(0)[1] > 0).sum())
    out = {"metric": f"frame-tracking ms/frame (1280x1024, {L} pyramid levels, coarse-to-fine alignment)", "ms_per_frame": ms,
           "ms_per_frame_min_mean_max": [float(np.min(per_frame)) * 1e3, float(np.mean(per_frame)) * 1e3, float(np.max(per_frame)) * 1e3],
           "pyramid_ms": pyr_ms, "lm_iterations_per_frame": its / frames, "success": bool(res["success"]),
           "reference_depth_maps_ms_per_keyframe": dm_ms, "depth_map_cells_level0": n0,
           "mean_square_optical_flow_ms_per_frame": flow_ms, "mean_square_optical_flow": float(flow[0]),
           "rmse_per_level": [float(x) for x in rl_final],
           "data": "synthetic 7-keyframe window + 1 new frame, target image resident in HBM"}
    # re-localisation (monocular_tracker.cpp:136-176,193-243): the initialisations are tried until one passes the per-level energy gates —
    # here every one but the last is far off (25 .. 45 degrees), with the gates a normally tracked frame leaves.  One per launch (the
    # sequential loop) against eight per launch (one XCD each, dsopp_hip_aligner_set_hypothesis_width; the default switches to it once a
    # first try has failed)
    rng = np.random.default_rng(11)

    def far():
        ax = rng.normal(size=3)
        return np.concatenate([rng.normal(0, 1.0, 3), ax / np.linalg.norm(ax) * rng.uniform(0.45, 0.8)])

    hyp_out = {}
    for n_hyp in (8, 113):
        hyp = np.stack([syn.mat_to_params(new_frame.T_w_c_init @ syn.se3_exp(far())) for _ in range(n_hyp - 1)] + [T_init])
        row = {}
        for width, name in ((1, "one_per_launch_ms"), (8, "eight_per_launch_ms")):
            a.set_hypothesis_width(width)
            ts, r = [], None
            for _ in range(3 if n_hyp > 8 and width == 1 else 7):
                rl = rl_final.copy()
                t0 = time.perf_counter()
                r = a.estimate_pose(kf.timestamp, T_ref, pr, maps, 1.0, ab_ref, new_frame.timestamp, pt, 1.0, win.scene.intrinsics, hyp, np.zeros(2), rl)
                ts.append(time.perf_counter() - t0)
            row[name] = float(np.median(ts) * 1e3)
            row.setdefault("tries", []).append(int(r["tries"]))
            row.setdefault("success", []).append(bool(r["success"]))
            row.setdefault("pose", []).append(r["T_w_target"])
        row["identical_result"] = bool(np.array_equal(row["pose"][0], row["pose"][1]) and row["tries"][0] == row["tries"][1])
        del row["pose"]
        hyp_out[f"{n_hyp}_initialisations_last_one_good"] = row
    a.set_hypothesis_width(0)
    out["relocalisation"] = hyp_out
    if not no_cpu:
        # the same frame through the CPU port: pyramid of the new frame + the coarse-to-fine chain stepped level by level
        # (scan of the depth map, alignment) from the same initialisation, against the same (downloaded) depth maps
        from oracle import pyoracle as po
        infos_ref, _ = po.build_pyramid(kf.image_u8, levels=L)
        t0 = time.perf_counter()
        infos_tgt, _ = po.build_pyramid(new_frame.image_u8, levels=L)
        cpu_pyr_ms = (time.perf_counter() - t0) * 1e3
        T, ab, its = T_init, np.zeros(2), 0
        t0 = time.perf_counter()
        for lvl in range(L - 1, -1, -1):
            ids, wgt = maps.get_level(lvl)
            u, v, idp, inten = po.points_from_depth_map(infos_ref[lvl], ids, wgt)
            r = po.align_solve(po.default_align_options(), u, v, idp, inten, win.scene.intrinsics / (1 << lvl), (W >> lvl, H >> lvl), T_ref, 1.0,
                               ab_ref, win.scene.intrinsics / (1 << lvl), infos_tgt[lvl], None, T, 1.0, ab)
            T, ab = r["T_w_target"], r["affine_brightness"]
            its += r["iterations"]
        out["cpu_port_ms_per_frame"] = cpu_pyr_ms + (time.perf_counter() - t0) * 1e3
        out["cpu_port_pyramid_ms"] = cpu_pyr_ms
        out["cpu_port_lm_iterations"] = its
        out["cpu_port_pose_difference"] = float(np.abs(T - res["T_w_target"]).max())
        out["cpu_port_threads"] = 1
    for o in (a, maps, m2, pr, pt, g):
        o.close()
    return out


def run_depth_estimation_timing(capi, syn, args, repeats=10):
    """row f-1: DepthEstimation::estimate for the immature landmarks of the window's keyframes against a new 640x480 frame
    (monocular_tracker.cpp:74-102 runs it for every frame over all keyframes): 7 keyframes x 2000 immature landmarks.
    GPU call = upload of the landmark arrays + kernel + download (host buffers at the boundary); CPU port beside it."""
    W, H, KF, N = 640, 480, 7, 2000
    win = syn.make_window(num_frames=KF + 1, num_points=(KF + 1) * N, width=W, height=H, seed=5, pose_noise=False)
    new = win.frames[-1]
    intr = win.scene.intrinsics
    pyr = capi.Pyramid(W, H, 1)
    pyr.set_level(0, new.pixelinfo)
    sets, Ts = [], []
    for f in win.frames[:KF]:
        uv = f.uv
        ui, vi = uv[:, 0].astype(int), uv[:, 1].astype(int)
        grad = np.stack([f.pixelinfo[vi, ui, 1], f.pixelinfo[vi, ui, 2]], axis=1)
        direction = np.stack([(uv[:, 0] - intr[2]) / intr[0], (uv[:, 1] - intr[3]) / intr[1], np.ones(len(uv))], axis=1)
        # ImmatureTrackingLandmark constructor defaults (immature_tracking_landmark.hpp:93-106), as struct-of-arrays
        sets.append(dict(projection=uv.copy(), direction=direction, patch=f.patch.copy(), gradient=grad, idepth_min=np.zeros(N),
                         idepth_max=np.full(N, 1000.0), uniqueness=np.full(N, np.finfo(np.float64).max),
                         search_pixel_interval=np.full(N, np.finfo(np.float64).max), status=np.full(N, 5, dtype=np.uint8),
                         traced=np.zeros(N, dtype=np.uint8)))
        Ts.append(syn.mat_to_params(np.linalg.inv(new.T_w_c_gt) @ f.T_w_c_gt))
    import copy
    t_gpu, t_rb = [], []
    good = 0
    for rep in range(repeats + 1):
        dsets = [capi.ImmatureSet(l) for l in sets]   # device-resident: the landmarks persist over frames, only their state moves
        for ds in dsets:
            ds.sync()                                 # (their uploads are not part of the per-frame call)
        t0 = time.perf_counter()
        capi.estimate_depths_batched(dsets, pyr, 0, intr, np.stack(Ts), np.ones(KF), np.zeros((KF, 2)))   # one launch over all keyframes
        dsets[0].sync()
        t_est = time.perf_counter() - t0
        states = [ds.download() for ds in dsets]      # (the tracker reads the states at keyframe time: activation / marginalisation)
        if rep:
            t_gpu.append(t_est)
            t_rb.append(time.perf_counter() - t0 - t_est)
        good = int(sum((s["status"] == 0).sum() for s in states))
        for ds in dsets:
            ds.close()
    out = {"workload": f"{KF} keyframes x {N} immature landmarks against one {W}x{H} frame", "gpu_ms_per_frame": float(np.median(t_gpu) * 1e3),
           "landmarks": KF * N, "good_after_first_observation": good, "state_read_back_ms": float(np.median(t_rb) * 1e3),
           "what": "dsopp_hip_immature_sets_estimate: one launch over the 7 device-resident landmark sets, then a stream sync; "
                   "state_read_back_ms = download of all estimator states (needed at keyframe time only)"}
    if not args.no_cpu:   # the CPU port is imported by this leg only
        from oracle import pyoracle as po
        hw = os.cpu_count() or 1
        po.set_threads(max(1, min(hw, 8) - 1))
        work = copy.deepcopy(sets)
        t0 = time.perf_counter()
        for lms, T in zip(work, Ts):
            po.estimate_depths(lms, new.pixelinfo, None, intr, T)
        out["cpu_port_ms_per_frame"] = (time.perf_counter() - t0) * 1e3
        out["cpu_port_threads"] = 1
    pyr.close()
    return out


def run_landmark_activation_timing(capi, syn, args, repeats=10):
    """row f-3: LandmarksActivator::activate when a new 640x480 keyframe arrives (monocular_tracker.cpp:495): 6 window
    keyframes x (286 active + 1500 immature landmarks), sparsity selection at pyramid level 1 + 3-iteration idepth refinement
    of every selected landmark over the 6 other keyframes.  The immature landmarks carry the estimator state two device
    depth-estimation passes leave.  GPU call = three launches + one packed read-back of statuses and inverse depths."""
    W, H, KF, NA, NI = 640, 480, 6, 286, 1500
    win = syn.make_window(num_frames=KF + 1, num_points=(KF + 1) * (NA + NI), width=W, height=H, seed=9, pose_noise=False)
    intr = win.scene.intrinsics
    new = win.frames[-1]
    pyramids = []
    for f in win.frames:
        p = capi.Pyramid(W, H, 2)
        p.set_level(0, f.pixelinfo)
        pyramids.append(p)
    g = capi.HipWindow(capi.default_pba_options())
    dsets, states = [], []
    fmax = float(np.finfo(np.float64).max)
    for i, f in enumerate(win.frames[:KF]):
        g.push_frame(f.frame_id, f.timestamp, None, None, intr, syn.mat_to_params(f.T_w_c_gt), 1.0, np.zeros(2), i == 0, False, pyramid=pyramids[i])
        g.set_landmarks(f.frame_id, f.uv[:NA], f.idepth_init[:NA], f.patch[:NA], np.zeros(NA, dtype=np.uint8))
        uv = f.uv[NA:]
        ui, vi = uv[:, 0].astype(int), uv[:, 1].astype(int)
        n = len(uv)
        lms = dict(projection=uv, direction=np.stack([(uv[:, 0] - intr[2]) / intr[0], (uv[:, 1] - intr[3]) / intr[1], np.ones(n)], axis=1),
                   patch=f.patch[NA:], gradient=np.stack([f.pixelinfo[vi, ui, 1], f.pixelinfo[vi, ui, 2]], axis=1), status=np.full(n, 5, dtype=np.uint8))
        ds = capi.ImmatureSet(lms)
        for j in (i + 1, KF):   # traced in the next keyframe and in the new one
            t = win.frames[j]
            ds.estimate(pyramids[j], 0, intr, syn.mat_to_params(np.linalg.inv(t.T_w_c_gt) @ f.T_w_c_gt))
        states.append(ds.download())
        dsets.append(ds)
    for a in win.frames[:KF]:
        for b in win.frames[:KF]:
            if a is not b:
                g.set_connection(a.frame_id, b.frame_id, np.zeros(NA, dtype=np.uint8))
    ids = [f.frame_id for f in win.frames[:KF]]
    t_gpu, res, st = [], None, None
    for rep in range(repeats + 1):
        for ds, s0 in zip(dsets, states):
            ds.upload(s0)
        for ds in dsets:
            ds.sync()   # (restoring the pre-activation state is not part of the per-keyframe call)
        t0 = time.perf_counter()
        st, _, res = g.activate_landmarks(ids, dsets, pyramids[KF], syn.mat_to_params(new.T_w_c_gt), 1.0, (0, 0), KF * NA, 2.0, True)
        if rep:
            t_gpu.append(time.perf_counter() - t0)
    out = {"workload": f"{KF} keyframes x ({NA} active + {NI} immature landmarks) against a new {W}x{H} keyframe, refinement on",
           "gpu_ms_per_keyframe": float(np.median(t_gpu) * 1e3), "activated": res["n_activated"], "skipped": res["n_skipped"], "deleted": res["n_deleted"],
           "selection_rounds": res["selection_rounds"], "min_distance_to_neighbor": res["min_distance_to_neighbor"],
           "what": "project + grid-hash greedy selection + wave-per-landmark LM refinement + read-back"}
    if not args.no_cpu:
        from oracle import pyoracle as po
        frames = []
        for i, f in enumerate(win.frames):
            d = dict(pixelinfo=f.pixelinfo, mask=None, T_w=syn.mat_to_params(f.T_w_c_gt), exposure=1.0, affine=np.zeros(2))
            if i < KF:
                lm = {k: np.array(v) for k, v in states[i].items()}
                lm["projection"], lm["patch"] = f.uv[NA:], f.patch[NA:]
                d.update(active_uv=f.uv[:NA], active_idepth=f.idepth_init[:NA], active_skip=np.zeros(NA, dtype=np.uint8), immature=lm)
            frames.append(d)
        t0 = time.perf_counter()
        st_o, _, _ = po.activate_landmarks(frames, intr, 20.0, KF * NA, 2.0, refine=True)
        out["cpu_port_ms_per_keyframe"] = (time.perf_counter() - t0) * 1e3
        out["cpu_port_threads"] = 1
        out["statuses_identical_to_cpu_port"] = bool(all(np.array_equal(a, b) for a, b in zip(st, st_o)))
    for ds in dsets:
        ds.close()
    g.close()
    for p in pyramids:
        p.close()
    return out


def run_concurrent_windows(capi, syn, torch, win, counts=(1, 2, 4, 8), solves=60):
    """Serving-side view: several INDEPENDENT C1 windows on one GPU, each on its own stream, all driven from one host thread
    through dsopp_hip_window_optimize_async / _wait ({restore; optimize} per solve like the headline loop).  A single
    window is latency-bound (3 dependent launches per iteration), so independent windows overlap until the machine
    fills.  Aggregate GN iterations/s per window count; the headline `value` stays the single-window figure."""
    out = {}
    for n in counts:
        streams = [torch.cuda.Stream() for _ in range(n)]
        gs = []
        for st in streams:
            g = capi.HipWindow(capi.default_pba_options(), stream=st.cuda_stream)
            syn.load_window(g, win)
            g.snapshot()
            g.optimize()
            gs.append(g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        its = 0
        for _ in range(solves):
            for g in gs:
                g.restore()
                g.optimize_async()
            for g in gs:
                its += g.optimize_wait()[1]
        dt = time.perf_counter() - t0
        out[str(n)] = its / dt
        for g in gs:
            g.close()
    # ... and with one host thread PER window (each runs its window's pipelined solves, as the headline loop does): the single enqueueing
    # thread above is bound by its own launch rate (~3 us per launch, 25 launches per solve); with a thread each the cap is the number of
    # hardware queues HIP maps the streams onto (GPU_MAX_HW_QUEUES, 4 by default: kernels of one queue run in order, so windows that
    # share a queue take turns) — profiles/r05/concurrent_windows_rates.txt
    import threading
    threaded = {}
    for n in counts:
        streams = [torch.cuda.Stream() for _ in range(n)]
        gs = []
        for st in streams:
            g = capi.HipWindow(capi.default_pba_options(), stream=st.cuda_stream)
            syn.load_window(g, win)
            g.snapshot()
            g.optimize_repeated(14)
            gs.append(g)
        torch.cuda.synchronize()
        barrier = threading.Barrier(n + 1)
        done = [0] * n

        def worker(i):
            barrier.wait()
            done[i], _ = gs[i].optimize_repeated(7 * solves)
            barrier.wait()

        ts = [threading.Thread(target=worker, args=(i,)) for i in range(n)]
        for t in ts:
            t.start()
        barrier.wait()
        t0 = time.perf_counter()
        barrier.wait()
        dt = time.perf_counter() - t0
        for t in ts:
            t.join()
        threaded[str(n)] = sum(done) / dt
        for g in gs:
            g.close()
    return {"workload": "n independent C1 windows, one stream each, one host thread (async enqueue, then wait)",
            "gn_iterations_per_s_by_window_count": out,
            "one_host_thread_per_window": {"gn_iterations_per_s_by_window_count": threaded,
                                           "what": "each window's solves enqueued by its own thread (dsopp_hip_window_optimize_repeated): the aggregate "
                                                   "rate follows the number of hardware queues in use (4 by default), see profiles/r05/concurrent_windows_rates.txt"}}


def run_keyframe_step_timing(capi, syn):
    """One keyframe step as the tracker drives the backend (monocular_tracker.cpp:497-507), in steady state: a 7-frame
    window of 286 landmarks per frame, 12 keyframes streamed through it.  pushFrame (incl. the fold-in of the frame that was
    marginalised, updateMarginalizedLinearSystem) -> landmarks + connections of the new keyframe -> solve() -> updateFrame
    read-back of every keyframe -> marginalisation flags.  Device pyramids exist already (the tracker built them when the
    frames arrived).  Median of the steady-state keyframes, ms per call group."""
    W, H = 640, 480
    win = syn.make_window(num_frames=12, num_points=12 * 286, width=W, height=H, seed=61)
    intr = win.scene.intrinsics
    g = capi.HipWindow(capi.default_pba_options())
    pyramids = []
    for f in win.frames:
        p = capi.Pyramid(W, H, 1)
        p.set_level(0, f.pixelinfo)
        pyramids.append(p)
    alive, rows = [], []
    for k, f in enumerate(win.frames):
        t0 = time.perf_counter()
        g.push_frame(f.frame_id, f.timestamp, None, None, intr, syn.mat_to_params(f.T_w_c_init), f.exposure, f.affine_init, f.fixed, False,
                     pyramid=pyramids[k])
        t1 = time.perf_counter()
        g.set_landmarks(f.frame_id, f.uv, f.idepth_init, f.patch, np.zeros(len(f.uv), dtype=np.uint8))
        for a in alive:
            g.set_connection(a.frame_id, f.frame_id, np.zeros(len(a.uv), dtype=np.uint8))
            g.set_connection(f.frame_id, a.frame_id, np.zeros(len(f.uv), dtype=np.uint8))
        alive.append(f)
        t2 = time.perf_counter()
        if len(alive) < 2:
            continue
        g.solve()
        t3 = time.perf_counter()
        for a in alive:
            g.get_pose(a.frame_id)
            g.get_frame_update(a.frame_id, [b.frame_id for b in alive if b is not a])
        t4 = time.perf_counter()
        if len(alive) == 7 and k + 1 < len(win.frames):
            victim = alive[1]
            for a in alive:
                fl = np.zeros(len(a.uv), dtype=np.uint8)
                fl[::4 if a is victim else 9] = 1
                g.set_landmarks(a.frame_id, a.uv, a.idepth_init, a.patch, fl)
            g.mark_frame_marginalized(victim.frame_id)
            alive.remove(victim)
        t5 = time.perf_counter()
        if k >= 8:
            rows.append([t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4])
    med = np.median(np.array(rows), axis=0) * 1e3
    g.close()
    for p in pyramids:
        p.close()
    return {"workload": "7-frame window, 286 landmarks per keyframe, steady state with one marginalisation per keyframe",
            "push_frame_incl_fold_in_ms": float(med[0]), "landmarks_and_connections_ms": float(med[1]), "solve_ms": float(med[2]),
            "update_frame_read_back_ms": float(med[3]), "marginalisation_flags_ms": float(med[4]), "total_ms": float(med.sum())}


def run_fullres_windows(capi, syn):
    """The sweep where its footprint lives in HBM (round-4 review, missing #4): the reference's dense configuration runs at
    resize_ratio 1 on 1280 x 1024 TUM-mono images (test/test_data/tummono/dense.yaml:21-22,35,42).  12 KF / 50 000 points and 15 KF /
    5000 points at that size, f64 and f32 texels: 12 x 42 MB = 503 MB (15: 629 MB) of f64 texels — beyond the 256 MiB Infinity Cache that
    holds every 640 x 480 window of this bench.  Isolated kernels by events + the fused loop's time per iteration; fractions against the
    8 TB/s spec and the guide's 6.3 TB/s achievable."""
    out = {"resolution": "1280x1024", "achievable_GBs": 6300.0}
    for name, F, P in (("12kf_50k", 12, 50000), ("15kf_5k", 15, 5000)):
        win = syn.make_window(num_frames=F, num_points=P, width=1280, height=1024, seed=1, render_device="cuda")
        entry = {"workload": f"{F} keyframes, {P} points, 1280x1024, full clique", "texel_bytes": {}}
        for dname, dtype, s_bytes in (("f64", capi.F64, 8), ("f32", capi.F32, 4)):
            g = capi.HipWindow(capi.default_pba_options(dtype=dtype))
            syn.load_window(g, win)
            g.snapshot()
            g.optimize_repeated(7)
            t0 = time.perf_counter()
            done, _ = g.optimize_repeated(28)
            dt = time.perf_counter() - t0
            g.restore()
            iso = {k: g.time_kernel(k, 30) for k in ("sweep_linearize", "sweep_linearize_loop", "sweep_energy", "schur", "assemble_solve")}
            b_lin, b_en = algorithmic_bytes_linearize(P, F, s_bytes), algorithmic_bytes_energy(P, F, s_bytes)
            t = iso["sweep_linearize_loop"] * 1e-6
            entry[dname] = {"us_per_iteration": dt / done * 1e6, "gn_iterations_per_s": done / dt, "kernels_isolated_avg_us": iso,
                            "algorithmic_bytes_per_launch": b_lin, "achieved_GBs": b_lin / t / 1e9, "frac_of_spec": b_lin / t / 1e9 / HBM_PEAK_GBS,
                            "frac_of_achievable": b_lin / t / 1e9 / 6300.0,
                            "energy_sweep": {"algorithmic_bytes_per_launch": b_en, "achieved_GBs": b_en / (iso["sweep_energy"] * 1e-6) / 1e9}}
            entry["texel_bytes"][dname] = F * 1280 * 1024 * 4 * s_bytes
            g.close()
        out[name] = entry
        del win
    return out


def run_dense_window(capi, syn):
    """the window shape of the reference's dense configuration (test/test_data/tummono/dense.yaml:35,42: 5000 desired points, up to 15
    keyframes): K = 120, the 5-tiles-per-wave variant of the two-stage Schur kernel when the deterministic build is on"""
    F, P = 15, 5000
    win = syn.make_window(num_frames=F, num_points=P, width=640, height=480, seed=2)
    out = {"workload": f"{F} keyframes, {P} points, 640x480, full clique (dense.yaml: number_of_desired_points 5000, maximum_size 15)"}
    for name, det in (("default", False), ("deterministic", True)):
        g = capi.HipWindow(capi.default_pba_options())
        syn.load_window(g, win)
        if det:
            g.set_deterministic(True)
        g.snapshot()
        g.optimize_repeated(14)
        t0 = time.perf_counter()
        done, _ = g.optimize_repeated(140)
        dt = time.perf_counter() - t0
        g.restore()
        out[name] = {"gn_iterations_per_s": done / dt, "us_per_iteration": dt / done * 1e6,
                     "kernels_isolated_avg_us": {k: g.time_kernel(k, 50) for k in ("sweep_linearize", "sweep_energy", "schur", "assemble_solve")}}
        g.close()
    return out


def run_tick_sequences(torch, syn, args):
    """BASELINE.json's second metric over tracked SEQUENCES (scripts/tick_sequence.py: MonocularTracker::tick through the C-ABI, 200 synthetic
    frames, keyframes by the reference's flow rule, a 7-keyframe window with marginalisation), at the bundle adjustment's image size and at the
    tracker configuration's, with the CPU port on a prefix of the same frames and both compared with the synthetic ground truth"""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import tick_sequence
    out = {}
    for name, (w, h, levels, cpu_frames) in {"640x480": (640, 480, 4, 30), "1280x1024": (1280, 1024, 5, 16)}.items():
        try:
            out[name] = tick_sequence.run(torch, syn, w, h, levels, 200, cpu_frames, no_cpu=args.no_cpu)
        except Exception as exc:  # noqa: BLE001 — an extra must not take the headline down
            out[name] = {"error": repr(exc)}
    return out


def run_cpu_baseline(args, F, P, win, syn):
    """the oracle (CPU port of the reference algorithm: materialised ResidualPoint AoS, Kahan accumulators, same stage
    structure) on the same window, with the reference's thread cap clamp(hw,1,8)-1 (dsopp_main.cpp:114-119)."""
    from oracle import pyoracle as po
    hw = os.cpu_count() or 1
    threads = max(1, min(hw, 8) - 1)
    po.set_threads(threads)
    o = po.OracleWindow(po.default_pba_options())
    syn.load_window(o, win)
    init = [(f.frame_id, syn.mat_to_params(f.T_w_c_init), f.affine_init, f.idepth_init) for f in win.frames]

    def reset():
        for fid, T, ab, idp in init:
            o.reset_state(fid, T, ab, idp)

    reset()
    o.optimize()
    its, t_used, solves = 0, 0.0, 0
    while t_used < args.cpu_seconds and solves < 200:
        reset()
        t0 = time.perf_counter()
        _, it, _ = o.optimize()
        t_used += time.perf_counter() - t0
        its += it
        solves += 1
    # second figure: all host cores (the reference caps its pool at 7 worker threads; this is the uncapped restatement)
    all_cores = None
    if hw > threads:
        hw = min(hw, 16)  # the pool oversubscribes badly beyond a few dozen threads on this problem size
        po.set_threads(hw)
        reset()
        o.optimize()
        its2, t2, n2 = 0, 0.0, 0
        while t2 < min(args.cpu_seconds, 6.0) and n2 < 100:
            reset()
            t0 = time.perf_counter()
            _, it, _ = o.optimize()
            t2 += time.perf_counter() - t0
            its2 += it
            n2 += 1
        all_cores = {"value": its2 / t2, "cores": hw}
        po.set_threads(threads)
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": its / t_used, "unit": "GN iterations/s", "cores": threads, "kind": "port", "all_cores": all_cores,
            "sample": f"{solves} LM solves ({its} GN iterations, {t_used:.1f} s) of the same C1 window ({F} KF, {P} points), "
                      f"oracle = restatement of the reference CPU path, {threads} threads (reference cap), host {cpu_model} ({hw} hw threads)"}


if __name__ == "__main__":
    main()
