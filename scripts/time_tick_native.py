"""the native C++ tick driver on the bench's 200-frame sequences (scripts/tick_sequence.py: run with native=True, no CPU leg): per-frame and
per-keyframe times with their phase split.  DSOPP_HIP_LIB selects the library build."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from dsopp_amd import synthetic as syn  # noqa: E402
import tick_sequence  # noqa: E402

sizes = {"1280x1024": (1280, 1024, 5), "640x480": (640, 480, 4)}
for name in (sys.argv[1:] or ["1280x1024"]):
    w, h, levels = sizes[name]
    out = tick_sequence.run(torch, syn, w, h, levels, 200, 0, no_cpu=True)
    n = out["native"]
    keep = ("frames", "keyframes", "ms_per_frame_mean", "ms_per_frame_median", "ms_per_keyframe_mean", "ms_per_frame_including_keyframe_work", "lm_iterations_per_frame",
            "ms_per_frame_by_phase", "ms_per_keyframe_by_phase", "pose_difference_to_the_python_driven_run_max", "error")
    print(json.dumps({"size": name, **{k: n[k] for k in keep if k in n}}))
