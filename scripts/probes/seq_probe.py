import sys, json, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts"))
import torch
from dsopp_amd import synthetic as syn
import tick_sequence as ts
for f in (1.0, 5.0, 20.0, 100.0):
    r = ts.run(torch, syn, 640, 480, 4, 60, 0, kf_factor=f, no_cpu=True)
    h = r["hip"]
    print(f, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in h.items()})
