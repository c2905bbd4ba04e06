"""Workload of the PMC passes (scripts/pmc_traffic.py runs it under `rocprofv3 --pmc ...`): a calibration copy of known
size, then back-to-back launches of the bundle-adjustment kernels on the C1 window (the same launches bench.py times)."""
import sys
sys.path.insert(0, ".")
import torch
from dsopp_amd import capi, synthetic as syn

# calibration: elementwise copy of 512 MiB (beyond the 256 MiB Infinity Cache): 512 MiB read + 512 MiB written
n = 128 * 1024 * 1024
src = torch.ones(n, dtype=torch.float32, device="cuda")
dst = torch.empty_like(src)
for _ in range(3):
    dst.copy_(src)
torch.cuda.synchronize()

win = syn.make_window(7, 2000, 640, 480, seed=0)
g = capi.HipWindow(capi.default_pba_options())
syn.load_window(g, win)
g.snapshot()
g.restore()
g.optimize()
g.restore()
for k in ("sweep_linearize", "sweep_energy", "schur", "assemble_solve"):
    g.time_kernel(k, 20)
g.close()
