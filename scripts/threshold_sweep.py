"""tuning aid: per-iteration time of a window for the current DSOPP_HIP_TWO_STAGE_MIN_CHUNKS / DSOPP_HIP_BACKSUB_SPLIT_MIN_CHUNKS
    python scripts/threshold_sweep.py F P [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from dsopp_amd import capi, synthetic as syn
F, P = int(sys.argv[1]), int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
win = syn.make_window(num_frames=F, num_points=P, width=640, height=480, seed=seed)
g = capi.HipWindow(capi.default_pba_options())
syn.load_window(g, win)
g.snapshot()
g.optimize_repeated(7)
ts = []
for _ in range(9):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done, _ = g.optimize_repeated(14)
    ts.append((time.perf_counter() - t0) / done)
ts.sort()
print(f"F={F} P={P} two_stage_min={os.environ.get('DSOPP_HIP_TWO_STAGE_MIN_CHUNKS','-')} backsub_split_min={os.environ.get('DSOPP_HIP_BACKSUB_SPLIT_MIN_CHUNKS','-')}: {ts[len(ts)//2]*1e6:.1f} us per iteration")
