"""HBM traffic of the hot kernels from the TCC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in
SEPARATE rocprofv3 --pmc passes (they do not fit one pass), no trace domains besides --kernel-trace, and calibrations on
known byte counts in the same pass: a 512 MiB streaming copy AND texel gathers with the access shape of the sweeps
(scripts/pmc_target.py).  Writes gpurun_out/pmc_traffic.json (+ the raw counter csv files under gpurun_out/pmc/).
Run on the GPU box:  python scripts/pmc_traffic.py [c1|large|fullres|fullres_f32]"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WHICH = next((a for a in sys.argv[1:] if not a.startswith("--")), "c1")
OUT = os.path.join(ROOT, "gpurun_out", "pmc_" + WHICH)

N_LANES = 2 * 1024 * 1024
# known byte counts of the gather calibrations (scripts/pmc_target.py): useful = bytes the lanes consume, seg64 / line128 =
# bytes of the distinct 64-byte segments / 128-byte lines they touch
GATHER = {
    "isolated": {"useful": N_LANES * 32, "seg64": N_LANES * 64, "line128": N_LANES * 128},
    "pair": {"useful": N_LANES * 64, "seg64": N_LANES * 64, "line128": N_LANES * 128},
    # 2 rows x {texels c, c+1}, c uniform in {0, 1, 2} of a 4-texel line: c = 1 straddles two 64-byte segments of one line
    "footprint": {"useful": N_LANES * 128, "seg64": int(N_LANES * 2 * (1 + 1 / 3) * 64), "line128": N_LANES * 2 * 128},
}


PARSE_ONLY = "--parse-only" in sys.argv  # re-derive the json from csv files of an earlier run


def run_pass(counter):
    d = os.path.join(OUT, counter)
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           sys.executable, os.path.join(ROOT, "scripts", "pmc_target.py"), WHICH]
    if not PARSE_ONLY:
        subprocess.run(cmd, cwd=ROOT, env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=1200)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                rows.append((int(row.get("Dispatch_Id", 0)), row["Kernel_Name"], float(row["Counter_Value"])))
    rows.sort()
    return rows


def main():
    res = {}
    copy_bytes = 128 * 1024 * 1024 * 4
    # (the loop's sweep is the plain linearisation variant since the back-substitution moved into the solve launch: one kernel, two labels)
    S = "float" if WHICH.endswith("_f32") else "double"
    labels = (("sweep_linearize", f"sweepKernel<{S}, true, true, true, false"),
              ("sweep_linearize_loop", f"sweepKernel<{S}, true, true, true, false"),
              ("sweep_energy", f"sweepKernel<{S}, false, true, true, false"),
              ("schur", "reduceSchurKernel"), ("assemble_solve", "assembleSolveKernel"))
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = run_pass(counter)
        by_kernel = {}
        for _, name, v in rows:
            by_kernel.setdefault(name, []).append(v)
        # calibration 1: the 512 MiB device-to-device copy (torch's copy_ of a contiguous tensor: __amd_rocclr_copyBuffer, or an
        # elementwise copy kernel, depending on the torch build) — the first launches of the pass
        cal_name, cal_vals = next(((n, v) for n, v in by_kernel.items() if "copyBuffer" in n or "direct_copy" in n or "CopyFunctor" in n),
                                  max(((n, v) for n, v in by_kernel.items() if "gatherCalibration" not in n), key=lambda kv: max(kv[1])))
        cal = max(cal_vals)
        entry = {"copy_kernel": cal_name[:80], "copy_counter_value": cal, "copy_bytes": copy_bytes, "bytes_per_count": copy_bytes / cal,
                 "kernels": {}, "gather": {}}
        # calibration 2: the gather kernel's launches come in pairs per pattern, in the order isolated, pair, footprint
        gvals = [v for _, name, v in rows if "gatherCalibrationKernel" in name]
        for k, pat in enumerate(("isolated", "pair", "footprint")):
            if len(gvals) >= 2 * k + 2:
                c = gvals[2 * k + 1]
                known = GATHER[pat]
                entry["gather"][pat] = {"counter_value": c, **known,
                                        "bytes_per_count_if_seg64": known["seg64"] / c if c else None,
                                        "bytes_per_count_if_line128": known["line128"] / c if c else None,
                                        "bytes_per_count_if_useful": known["useful"] / c if c else None}
        for label, needle in labels:
            vals = next((v for name, v in by_kernel.items() if needle in name), [])
            if vals:
                tail = vals[-20:]  # the back-to-back launches of time_kernel
                entry["kernels"][label] = {"launches": len(tail), "counter_avg": sum(tail) / len(tail)}
        res[counter] = entry
    # Conversion.  The counter's unit is 1 KiB of 64-byte fabric requests.  The ISOLATED gather (one 32-byte texel per lane, each
    # in its own 128-byte line) pins the unit for gathers: counter x 1024 = lanes x 64 B within a few per cent, i.e. a texel
    # access fetches its 64-byte half line and the counter tallies it exactly (a wide stream is tallied at HALF its bytes, the
    # copy calibration).  The pair / footprint patterns read MORE than their distinct segments (x1.2 / x1.5: the several 16-byte
    # loads of a lane touch a segment at different times and part of them miss again) — that excess is real traffic and the
    # kernel times confirm it (all three patterns run at the same ~3.3 TB/s of counted bytes).  So: sweeps (texel gathers) are
    # converted with the isolated-gather unit, the small dense kernels with the streaming unit, writes with the copy's write unit.
    iso = res["FETCH_SIZE"]["gather"].get("isolated", {})
    gather_bpc = iso.get("bytes_per_count_if_seg64") or 1024.0
    out = {"source": f"rocprofv3 --kernel-trace --pmc <C> -- python scripts/pmc_target.py {WHICH} (one pass per counter)",
           "workload": {"c1": "C1: 7 KF / 2000 points / 640x480", "fullres": "12 KF / 50 000 points / 1280x1024, f64 texels (503 MB)",
                        "fullres_f32": "12 KF / 50 000 points / 1280x1024, f32 texels (252 MB)"}.get(WHICH, "12 KF / 50 000 points / 640x480"),
           "note": "bytes = counter x bytes_per_count.  Streaming calibration: 512 MiB elementwise copy (MI355X_MICROARCH.md: FETCH_SIZE "
                   "under-reports wide streaming reads 2x on gfx950).  Gather calibration: texel gathers of known geometry in the same pass; "
                   "the sweeps' fetches are converted with the unit the isolated gather pins (~993 B per count: one 64-byte request per texel access), the small dense kernels with the streaming unit.",
           "passes": res, "calibration": {"fetch_streaming_bytes_per_count": res["FETCH_SIZE"]["bytes_per_count"],
                                          "fetch_gather_bytes_per_count": gather_bpc,
                                          "write_bytes_per_count": res["WRITE_SIZE"]["bytes_per_count"]},
           "per_launch_bytes": {}}
    for label in res["FETCH_SIZE"]["kernels"]:
        is_sweep = label.startswith("sweep")
        f = res["FETCH_SIZE"]["kernels"][label]["counter_avg"] * (gather_bpc if is_sweep else res["FETCH_SIZE"]["bytes_per_count"])
        w = res["WRITE_SIZE"]["kernels"].get(label, {}).get("counter_avg", 0.0) * res["WRITE_SIZE"]["bytes_per_count"]
        out["per_launch_bytes"][label] = {"fetch": f, "write": w, "total": f + w,
                                          "fetch_if_streaming_calibration": res["FETCH_SIZE"]["kernels"][label]["counter_avg"] * res["FETCH_SIZE"]["bytes_per_count"]}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    name = {"c1": "pmc_traffic.json", "large": "pmc_traffic_large.json"}.get(WHICH, f"pmc_traffic_{WHICH}.json")
    with open(os.path.join(ROOT, "gpurun_out", name), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["per_launch_bytes"], indent=1))
    print(json.dumps(out["calibration"], indent=1))
    print(json.dumps(res["FETCH_SIZE"]["gather"], indent=1))


if __name__ == "__main__":
    main()
