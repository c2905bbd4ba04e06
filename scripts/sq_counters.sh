#!/bin/bash
# SQ counters of the kernels of one scripts/profile_target.py workload, one rocprofv3 --pmc pass per counter set (kernel trace only:
# the pool refuses --pmc next to the hip / hsa trace domains), summarised per kernel into gpurun_out/<out>.json.
#   usage: sq_counters.sh <target> <out> [name filter regex]     env DSOPP_HIP_LIB selects another build of the library
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
target=$1; out=$2; filt=${3:-.}
mkdir -p gpurun_out
rm -f /tmp/sq_*.json
i=0
if [ -n "$COUNTER_SETS" ]; then IFS=';' read -ra SETS <<< "$COUNTER_SETS"; else SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_CYCLES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"); fi
for set in "${SETS[@]}"; do
  d=/tmp/pmc_$i
  rm -rf $d
  (cd /tmp && timeout ${PASS_TIMEOUT:-150} rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o pmc -- python $GRAFT_REPO_ROOT/scripts/profile_target.py $target > /tmp/pmc.log 2>&1) || { echo "set [$set] failed: $(grep -i -m2 "error\|invalid\|not" /tmp/pmc.log)"; i=$((i+1)); continue; }
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "set [$set]: no counter file"; i=$((i+1)); continue; }
  python - "$f" "$filt" /tmp/sq_$i.json <<'PY'
import csv, sys, collections, re, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
filt = re.compile(sys.argv[2])
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "")).replace("dsopp_hip::", "").replace("void ", "")
    if not filt.search(n): continue
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
json.dump({n: {c: {"avg": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for n, cs in acc.items()}, open(sys.argv[3], "w"))
PY
  i=$((i+1))
done
python - "$out" "$target" <<'PY'
import glob, json, sys, collections, os
merged = collections.defaultdict(dict)
for f in sorted(glob.glob("/tmp/sq_*.json")):
    for n, cs in json.load(open(f)).items():
        merged[n].update({c: round(v["avg"], 1) for c, v in cs.items()})
        merged[n]["launches"] = max(merged[n].get("launches", 0), max(v["launches"] for v in cs.values()))
for n, c in merged.items():
    wc, busy = c.get("SQ_WAVE_CYCLES"), c.get("SQ_BUSY_CYCLES")
    d = {}
    if wc:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS"):
            if k in c: d[k + "/WAVE_CYCLES"] = round(c[k] / wc, 4)
    if c.get("SQ_WAVES"):
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_MFMA_MOPS_F64"):
            if k in c: d[k + "/wave"] = round(c[k] / c["SQ_WAVES"], 2)
    c["derived"] = d
res = {"target": sys.argv[2], "library": os.environ.get("DSOPP_HIP_LIB", "dsopp_amd/lib/libdsopp_hip.so"), "kernels": merged,
       "units": "raw rocprofv3 counter values summed over the device's SQs, averaged over the launches of the kernel; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave (MI355X_MICROARCH.md)"}
json.dump(res, open(f"gpurun_out/{sys.argv[1]}.json", "w"), indent=1)
for n, c in merged.items():
    print(n[:90]); print("   ", {k: v for k, v in c.items() if k != "derived"}); print("   ", c["derived"])
PY
