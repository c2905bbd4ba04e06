"""host-side breakdown of the batched depth-estimation call (DSOPP_HIP_TRACE=1): enqueue time inside the library, time to the
stream synchronisation, Python wrapper on top (tuning aid)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import numpy as np
import torch  # noqa: F401
import bench
from dsopp_amd import capi, synthetic as syn

print(bench.run_depth_estimation_timing(capi, syn, argparse.Namespace(no_cpu=True), repeats=3))
