"""Workload of the PMC passes (scripts/pmc_traffic.py runs it under `rocprofv3 --pmc ...`):
  1. calibration on known byte counts, in the same pass as the kernels they calibrate:
       copy      — 512 MiB elementwise copy (wide coalesced stream; beyond the 256 MiB Infinity Cache)
       gather_*  — dsopp_hip_debug_gather_calibration: lanes read whole 32-byte texels at random positions of a 1 GiB buffer,
                   the access shape of the sweeps' bilinear sampling:
                     isolated : one texel per lane, every lane in its own 128-byte line
                     pair     : two adjacent texels per lane starting at an EVEN texel (one 64-byte segment per lane)
                     footprint: 2 x 2 texels per lane at a random (odd or even) column, second row one image row below
  2. back-to-back launches of the bundle-adjustment kernels on the C1 window and on the 12-KF / 50 000-point window
     (the same launches bench.py times).
Usage: pmc_target.py [c1|large|fullres|fullres_f32]"""
import ctypes as C
import os
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from dsopp_amd import capi, synthetic as syn

which = sys.argv[1] if len(sys.argv) > 1 else "c1"

# ---- calibration 1: streaming copy
n = 128 * 1024 * 1024
src = torch.ones(n, dtype=torch.float32, device="cuda")
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()
del src, dst

# ---- calibration 2: texel gathers over 1 GiB (32 Mi texels of 32 B), 2 Mi lanes each
n_tex = 32 * 1024 * 1024
buf = torch.zeros(n_tex * 4, dtype=torch.float64, device="cuda")
scratch = torch.zeros(8, dtype=torch.float64, device="cuda")
rng = np.random.default_rng(0)
n_lanes = 2 * 1024 * 1024
W = 640  # image row of the footprint pattern (texels)
# distinct 128-byte lines (4 texels) on EVEN image rows only: the footprint's second row (one row below) then never shares
# a line with another lane
lines_per_row = W // 4
n_pairs = n_tex // W // 2
pos = rng.permutation(n_pairs * lines_per_row)[:n_lanes]
base = ((pos // lines_per_row) * 2 * W + (pos % lines_per_row) * 4).astype(np.int64)   # first texel of the line
patterns = {
    "isolated": (base, 1, 0),
    "pair": (base + 2 * rng.integers(0, 2, n_lanes), 2, 0),
    "footprint": (base + rng.integers(0, 3, n_lanes), 2, W),
}
tools = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dsopp_amd", "lib", "libdsopp_hip_tools.so"))  # not part of the product library
fn = tools.dsopp_hip_debug_gather_calibration
for name, (idx, per_lane, stride) in patterns.items():
    d_idx = torch.from_numpy(idx.astype(np.int64)).to(torch.int32).cuda()  # texel indices < 2^25
    torch.cuda.synchronize()
    for _ in range(2):
        rc = fn(C.c_void_p(buf.data_ptr()), C.c_void_p(d_idx.data_ptr()), C.c_size_t(n_lanes), C.c_int(per_lane), C.c_size_t(stride),
                C.c_void_p(scratch.data_ptr()), C.c_void_p(0))
        assert rc == 0
    torch.cuda.synchronize()
del buf

# ---- the kernels
if which == "c1":
    win = syn.make_window(7, 2000, 640, 480, seed=0)
elif which.startswith("fullres"):
    # 12 KF / 50 000 points on 1280 x 1024 images (the resolution the reference's dense configuration runs at, dense.yaml:21-22): 503 MB of
    # f64 texels — beyond the 256 MiB Infinity Cache, so the counters see DRAM traffic (fullres_f32: 16-byte texels)
    win = syn.make_window(12, 50000, 1280, 1024, seed=1, render_device="cuda")
else:
    win = syn.make_window(12, 50000, 640, 480, seed=1)
g = capi.HipWindow(capi.default_pba_options(dtype=capi.F32) if which.endswith("_f32") else capi.default_pba_options())
syn.load_window(g, win)
g.snapshot()
g.restore()
g.optimize()
g.restore()
for k in ("sweep_linearize", "sweep_linearize_loop", "sweep_energy", "schur", "assemble_solve"):
    g.time_kernel(k, 20)
g.close()
