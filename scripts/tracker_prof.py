import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dsopp_amd import capi, synthetic as syn
import bench
out = bench.run_tracker_timing(capi, syn, torch, frames=20)
print(out)
