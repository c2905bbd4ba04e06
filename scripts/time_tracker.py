"""BASELINE's second metric in isolation: frame-tracking ms / frame at 1280x1024 with 5 and 4 pyramid levels (bench.run_tracker_timing without
the CPU port).  DSOPP_HIP_LIB selects the library (A/B against another build); with a stamps build and DSOPP_HIP_TRACE=1 the persistent
kernel's phase stamps of one pass go to stderr."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from dsopp_amd import capi, synthetic as syn  # noqa: E402
import bench  # noqa: E402

out = {}
for levels in (5, 4):
    t = bench.run_tracker_timing(capi, syn, torch, frames=20, no_cpu=True, levels=levels)
    out[f"{levels}_levels"] = {k: t[k] for k in ("ms_per_frame", "ms_per_frame_min_mean_max", "pyramid_ms", "lm_iterations_per_frame", "success", "rmse_per_level")}
    out[f"{levels}_levels"]["relocalisation_8"] = t["relocalisation"]["8_initialisations_last_one_good"]
print(json.dumps({"lib": os.environ.get("DSOPP_HIP_LIB", "default"), **out}))
