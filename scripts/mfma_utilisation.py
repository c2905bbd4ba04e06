"""f64 matrix-core utilisation of the kernels that use v_mfma_f64_16x16x4_f64, from the committed counter passes and kernel statistics:
   python scripts/mfma_utilisation.py profiles/r04 -> profiles/r04/mfma_utilisation.json
Inputs: sq_<target>.json (scripts/sq_counters.sh: SQ_INSTS_MFMA, SQ_INSTS_VALU_MFMA_MOPS_F64, SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE per
launch) and <target>_kernel_stats.csv (rocprofv3 --kernel-trace --stats: average duration).  One v_mfma_f64_16x16x4_f64 is 2 x 16 x 16 x 4 =
2048 flop and 4 MOPS counts (512 flop each) and keeps the SIMD's matrix pipe busy for 64 cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
Peak: AMD quotes 78.6 TFLOP/s fp64 matrix for MI355X (= 256 CUs x 4 SIMDs x 2.4 GHz x 32 flop / cycle); the microarchitecture guide of
this repo lists no fp64 row, so that spec figure is the denominator."""
import csv
import json
import os
import re
import sys

PEAK_TFLOPS = 78.6
d = sys.argv[1]
out = {"peak_tflops_fp64_matrix": PEAK_TFLOPS, "peak_source": "AMD MI355X spec (fp64 matrix); MI355X_MICROARCH.md has no fp64 row", "kernels": {}}
for target, stats in (("c1", "c1_kernel_stats.csv"), ("large_loop", "large_loop_kernel_stats.csv"), ("large", "large_kernel_stats.csv"),
                      ("tracker", "tracker_kernel_stats.csv")):
    sq_path = os.path.join(d, f"sq_{target}.json")
    if not os.path.exists(sq_path):
        continue
    sq = json.load(open(sq_path))["kernels"]
    dur = {}
    sp = os.path.join(d, stats)
    if os.path.exists(sp):
        for row in csv.DictReader(open(sp)):
            n = re.sub(r"\(.*", "", row["Name"].replace("(anonymous namespace)::", "")).replace("dsopp_hip::", "").replace("void ", "")
            dur[n] = float(row["AverageNs"])
    for name, c in sq.items():
        n_mfma = c.get("SQ_INSTS_MFMA", 0)
        if not n_mfma:
            continue
        flops = c.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) * 512.0
        cycles = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
        e = {"window": target, "mfma_instructions_per_launch": n_mfma, "flop_per_launch": flops,
             "matrix_pipe_busy_cycles_per_launch": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), "kernel_cycles": cycles,
             "matrix_pipe_utilisation": (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cycles * 1024.0)) if cycles else None}
        if name in dur:
            e["avg_duration_us"] = dur[name] / 1e3
            e["achieved_tflops"] = flops / (dur[name] * 1e-9) / 1e12
            e["frac_of_peak"] = e["achieved_tflops"] / PEAK_TFLOPS
        out["kernels"][f"{target}: {name}"] = e
json.dump(out, open(os.path.join(d, "mfma_utilisation.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    print(k[:80], {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("matrix_pipe_utilisation", "achieved_tflops", "frac_of_peak", "avg_duration_us")})
