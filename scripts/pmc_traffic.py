"""HBM traffic of the hot kernels from the TCC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
SEPARATE rocprofv3 --pmc passes (they do not fit one pass), no trace domains besides --kernel-trace, and a calibration on
a known byte count (512 MiB elementwise copy) in the same pass.  Writes gpurun_out/pmc_traffic.json.
Run on the GPU box:  python scripts/pmc_traffic.py"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "pmc")


def run_pass(counter):
    d = os.path.join(OUT, counter)
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           sys.executable, os.path.join(ROOT, "scripts", "pmc_target.py")]
    subprocess.run(cmd, cwd=ROOT, env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    per_kernel = {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            per_kernel.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return per_kernel


def pick(per_kernel, needle):
    for name, vals in per_kernel.items():
        if needle in name:
            return name, vals
    return None, []


def main():
    res = {}
    calib_bytes = 128 * 1024 * 1024 * 4
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        pk = run_pass(counter)
        # calibration kernel: the torch elementwise copy (largest counter value in the pass)
        cal_name, cal_vals = max(pk.items(), key=lambda kv: max(kv[1]))
        cal = max(cal_vals)
        entry = {"calibration_kernel": cal_name[:80], "calibration_counter_value": cal, "calibration_bytes": calib_bytes,
                 "bytes_per_count": calib_bytes / cal, "kernels": {}}
        for label, needle in (("sweep_linearize", "sweepKernel<double, true, true, true, false>"),
                              ("sweep_linearize_backsub", "sweepKernel<double, true, true, true, true>"),
                              ("sweep_energy", "sweepKernel<double, false, true, true, false>"),
                              ("schur", "reduceSchurKernel"), ("assemble_solve", "assembleSolveKernel")):
            name, vals = pick(pk, needle)
            if vals:
                tail = vals[-20:]  # the back-to-back launches of time_kernel
                avg = sum(tail) / len(tail)
                entry["kernels"][label] = {"launches": len(tail), "counter_avg": avg, "bytes_avg": avg * entry["bytes_per_count"]}
        res[counter] = entry
    out = {"source": "rocprofv3 --kernel-trace --pmc <C> -- python scripts/pmc_target.py (one pass per counter)",
           "note": "bytes = counter x bytes_per_count, bytes_per_count calibrated on a 512 MiB elementwise copy in the same pass "
                   "(MI355X_MICROARCH.md: FETCH_SIZE under-reports wide streaming reads 2x on gfx950; WRITE_SIZE uncalibrated)",
           "passes": res, "per_launch_bytes": {}}
    for label in res["FETCH_SIZE"]["kernels"]:
        f = res["FETCH_SIZE"]["kernels"][label]["bytes_avg"]
        w = res["WRITE_SIZE"]["kernels"].get(label, {}).get("bytes_avg", 0.0)
        out["per_launch_bytes"][label] = {"fetch": f, "write": w, "total": f + w}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "pmc_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["per_launch_bytes"], indent=1))
    print({c: (res[c]["calibration_kernel"], res[c]["bytes_per_count"]) for c in res})


if __name__ == "__main__":
    main()
