"""exports a short 1280x1024 tick sequence and builds dsopp_amd/host/tick_sequence.cpp next to it: the inputs of a rocprofv3 --hip-trace run of
the native driver (scripts/gpu_r6_keyframe_trace.sh)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch  # noqa: E402
from dsopp_amd import synthetic as syn  # noqa: E402
import tick_sequence as ts  # noqa: E402

out = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 80
os.makedirs(out, exist_ok=True)
W, H, L = 1280, 1024, 5
scene = syn.Scene.make(W, H, 41)
poses = [syn.se3_exp(0.2 * k * syn.BASE_MOTION) for k in range(frames)]
rendered = ts.render_frames(torch, scene, poses)
u8 = [r[0] for r in rendered]
depths = {0: rendered[0][1], 3: rendered[3][1]}
ts.export_sequence(os.path.join(out, "sequence.bin"), u8, depths, poses, scene, syn, levels=L, n_boot=1000, n_immature=1500, desired_points=2000, max_keyframes=7,
                   kf_factor=5.0)
lib = os.path.join(ROOT, "dsopp_amd", "lib")
subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(ROOT, "dsopp_amd", "host", "tick_sequence.cpp"), f"-L{lib}", "-ldsopp_hip", f"-Wl,-rpath,{lib}",
                       "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-o", os.path.join(out, "tick_sequence")])
print("prepared", out)
