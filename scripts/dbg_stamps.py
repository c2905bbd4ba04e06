"""Phase stamps of the fused loop's solve kernel (solveCombinedKernel) and reduction kernel on the C1 window.
Needs a library built with -DDSOPP_HIP_STAMPS:  DSOPP_HIP_EXTRA_FLAGS=-DDSOPP_HIP_STAMPS bash dsopp_amd/csrc/build.sh"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
from dsopp_amd import capi, synthetic as syn
F = int(sys.argv[1]) if len(sys.argv) > 1 else 7
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
win = syn.make_window(F, P, 640, 480, seed=0)
g = capi.HipWindow(capi.default_pba_options()); syn.load_window(g, win)
out = (C.c_longlong * 64)()
assert capi.lib().dsopp_hip_debug_solve_stamps(g._h, out) == 0, capi.lib().dsopp_hip_last_error()
g.snapshot()
for _ in range(3):
    g.restore(); g.optimize()
capi.lib().dsopp_hip_debug_solve_stamps(g._h, out)
st = np.array(list(out), dtype=np.int64) / 100.0   # wall_clock64 ticks at 100 MHz -> us
print(f"solve: entry -> operands + decision {st[0]-st[7]:.2f}  loads+assemble {st[1]-st[0]:.2f}  cholesky {st[2]-st[1]:.2f}  back-substitution {st[6]-st[2]:.2f}  step out {st[3]-st[6]:.2f}  pair refresh {st[4]-st[3]:.2f}  prior energy {st[5]-st[4]:.2f}  "
      f"total {st[5]-st[0]:.2f}")
rs = st[24:]
print(f"solve head: entry -> control block here {st[8]-st[7]:.2f} -> decided {st[9]-st[8]:.2f} -> states stored (stamp 0) {st[0]-st[9]:.2f} -> first barrier {st[10]-st[0]:.2f} -> system stored {st[11]-st[10]:.2f} -> rhs + barrier {st[1]-st[11]:.2f}")
print(f"cholesky, panel wave, summed over the block steps: column update {st[12]:.2f}  barrier {st[13]:.2f}  factor + panel {st[14]:.2f}  barrier behind it {st[15]:.2f}")
print("reduceSchur (wg 1) stamps us:", np.round(rs[:8] - rs[0], 2))
ts = st[32:40]
print("schurTwoStage (wg 1, first chunk) stamps us:", np.round(ts - ts[0], 2), "(0 chunk start, 1 rows cleared + flags, 2 phase 1 done, 3 barrier, 4 MFMA done, 5 b_schur done, 6 all chunks done, 7 partial written)")
