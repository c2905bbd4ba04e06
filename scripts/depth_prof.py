import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dsopp_amd import capi, synthetic as syn
args = argparse.Namespace(no_cpu=True)
print(bench.run_depth_estimation_timing(capi, syn, args))
