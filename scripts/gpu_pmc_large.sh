#!/bin/bash
# what bounds the large-window kernels: derived SQ / TA / TCP counters of the sweep, the two-stage build and the solve (12 KF / 50k)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for set in "VALUBusy" "MemUnitBusy" "MemUnitStalled VALUUtilization" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "LDSBankConflict L2CacheHit"; do
  d=/tmp/pmc_$(echo $set | tr ' ' '_')
  rm -rf $d
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -o pmc -- python $GRAFT_REPO_ROOT/scripts/profile_target.py large_loop > /tmp/pmc.log 2>&1) || { echo "set [$set] failed: $(tail -2 /tmp/pmc.log | head -1)"; continue; }
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "set [$set]: no counter file"; continue; }
  python - "$f" <<'PY'
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("dsopp_hip::", "").replace("void ", "")[:60]
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    if not any(k in n for k in ("sweepKernel<double, true, true, true, false, false", "sweepKernel<double, false", "schurTwoStage", "solveCombined", "backsub", "combineSystem")): continue
    print(n, {c: round(sum(v) / len(v), 2) for c, v in cs.items()})
PY
done
