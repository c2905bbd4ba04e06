"""A slice of scripts/stress_parity.py inside the suite the driver runs: 20 randomised small windows (random size, seed, first-estimate Jacobians,
force_accept, iteration budget, fused / host-driven loop, deterministic build on / off, one in three through the single-process window group with
1-4 shards) and 4 large ones (up to 12 keyframes / 16 000 points at 640 x 480: two-stage Schur build, several groups per sweep workgroup) —
the production solve() of the HIP library against the CPU checker, tolerances of tests/test_gpu_pba*.py.  The long form (150 + 150 + 30 cases)
is run by hand after changes to the solve path; its logs are under profiles/."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("args,cases", [(["20", "6"], 20), (["4", "6", "big"], 4)])
def test_randomised_solve_parity_slice(args, cases):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stress_parity.py")] + args, capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-(cases + 2):])
    assert r.returncode == 0, tail + "\n" + r.stderr[-1500:]
    assert f"{cases}/{cases} cases agree" in r.stdout, tail
    assert "MISMATCH" not in r.stdout
