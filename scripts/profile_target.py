"""One workload per process for `rocprofv3 --kernel-trace --stats` (scripts/collect_profiles.sh): the kernel_stats csv of each run
is what DESIGN.md's per-kernel figures are recomputed from.
   c1         the bench's C1 loop (7 KF / 2000 points): 140 GN iterations
   c1_isolated  the roofline kernel alone, as bench.py times it: 200 back-to-back launches of the in-loop linearisation sweep
              (sweepKernel<double, LIN, FEJ, HUBER, BACKSUB = false>: since the back-substitution moved into the solve launch the loop
              runs the plain linearisation variant) on the C1 window — so the csv's average for it IS the figure
              `roofline.avg_launch_us` quotes (in the c1 csv the same kernel runs inside the loop, between dependent launches)
   large      12 KF / 50 000 points on one GPU: 3 LM solves + isolated kernel launches
   large_loop the same window, fused loop only (5 solves)
   c3_loop    7 KF / 20 000 points, fused loop only
   large_loop_tile32  as large_loop, every frame's landmarks in 32 x 32-pixel tiles in raster order (a grid-cell extractor's order)
   tracker    C2: 1280x1024, 5 levels, 20 frames of pyramid + estimatePose
   depth      7 x 2000 immature landmarks against one 640x480 frame
   activation 6 x (286 active + 1500 immature) landmarks against a new keyframe"""
import argparse
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsopp_amd import capi, synthetic as syn

what = sys.argv[1]
args = argparse.Namespace(no_cpu=True)
if what == "c1":
    win = syn.make_window(7, 2000, 640, 480, seed=0)
    g = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g, win)
    g.snapshot()
    g.optimize_repeated(14)
    g.optimize_repeated(140)
    g.close()
elif what == "c1_isolated":
    win = syn.make_window(7, 2000, 640, 480, seed=0)
    g = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g, win)
    g.snapshot()
    g.restore()
    print("sweep_linearize_loop", g.time_kernel("sweep_linearize_loop", 200))
    g.close()
elif what == "large":
    win = syn.make_window(12, 50000, 640, 480, seed=1)
    g = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g, win)
    g.snapshot()
    g.optimize_repeated(21)
    g.restore()
    for k in ("sweep_linearize", "sweep_linearize_loop", "sweep_energy", "schur", "assemble_solve"):
        print(k, g.time_kernel(k, 20))
    g.close()
elif what == "c3_loop":
    win = syn.make_window(7, 20000, 640, 480, seed=0)
    g = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g, win)
    g.snapshot()
    g.optimize_repeated(7)
    g.optimize_repeated(28)
    g.close()
elif what == "large_loop":
    # only the fused loop (for a kernel-by-kernel timeline of one solve at this size: scripts/one_solve_timeline.py)
    win = syn.make_window(12, 50000, 640, 480, seed=1)
    g = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g, win)
    g.snapshot()
    g.optimize_repeated(7)
    g.optimize_repeated(28)
    g.close()
elif what == "large_loop_tile32":
    # the large window with spatially ordered landmarks (32 x 32-pixel tiles in raster order): fused loop only
    win = syn.make_window(12, 50000, 640, 480, seed=1, order="tile32")
    g = capi.HipWindow(capi.default_pba_options())
    syn.load_window(g, win)
    g.snapshot()
    g.optimize_repeated(7)
    g.optimize_repeated(28)
    g.close()
elif what == "tracker":
    print(bench.run_tracker_timing(capi, syn, torch, no_cpu=True))
elif what == "depth":
    print(bench.run_depth_estimation_timing(capi, syn, args))
elif what == "activation":
    print(bench.run_landmark_activation_timing(capi, syn, args))
else:
    raise SystemExit(what)
