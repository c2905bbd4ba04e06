#!/bin/bash
# same-box A/B of the round's second half (launch prologues, solve-launch head, copies of the combined system) against the library built from
# the sources at d8ab8f0 (dsopp_amd/lib_exp_r5start: git worktree add /tmp/wt d8ab8f0; DSOPP_HIP_OUT=... bash dsopp_amd/csrc/build.sh):
# one-solve timelines under rocprofv3 --kernel-trace for C1 / C3 / C4 and the host-timed rate per iteration, two passes, alternating.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
out=$O/second_half_ab.txt
: > $out
for pass in 1 2; do
  for lib in lib_exp_r5start lib; do
    for what in c1 c3_loop large_loop; do
      d=/tmp/prof_ab; rm -rf $d
      (cd /tmp && DSOPP_HIP_LIB=$GRAFT_REPO_ROOT/dsopp_amd/$lib/libdsopp_hip.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d $d -o prof -- python $GRAFT_REPO_ROOT/scripts/profile_target.py $what > /dev/null 2>&1)
      t=$(find $d -name '*kernel_trace.csv' | head -1)
      [ -n "$t" ] && python scripts/one_solve_timeline.py "$t" > /tmp/tl.csv && python - "$pass" "$lib" "$what" <<'PY' | tee -a $out
import csv, sys
rows = list(csv.reader(open("/tmp/tl.csv")))
tot = rows[-1][1]
agg = {}
for k, d, g in rows[1:-1][4:]:
    agg.setdefault(k.split("(")[0][:34], []).append(float(d))
print("pass", sys.argv[1], f"{sys.argv[2]:16s}", f"{sys.argv[3]:10s}", "one solve", tot, "us |", "  ".join(f"{k} {sum(v)/len(v):.2f}" for k, v in agg.items()))
PY
    done
  done
done
for w in "7 2000" "7 6000" "7 10000" "7 20000" "12 50000" "15 5000"; do
  for lib in lib_exp_r5start lib; do
    r=$(DSOPP_HIP_LIB=$PWD/dsopp_amd/$lib/libdsopp_hip.so timeout 300 python scripts/threshold_sweep.py $w 2>/dev/null | grep "us per" | sed 's/.*: //')
    echo "host-timed  $lib  $w: $r" | tee -a $out
  done
done
