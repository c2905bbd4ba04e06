#!/bin/bash
# round-4 measurement pass on the GPU box: everything DESIGN.md quotes lands under gpurun_out/r04/ (copied into profiles/r04/).
#   usage: bash scripts/gpu_round4.sh [quick]     quick: skip the test suite and the r03-library counter pass
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
if [ "$1" != "quick" ]; then
  timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -2
fi
# rocprofv3 --kernel-trace --stats per workload
for what in c1 c1_isolated large large_loop large_loop_tile32 c3_loop tracker depth activation; do
  d=/tmp/prof_$what; rm -rf $d
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o prof -- python $GRAFT_REPO_ROOT/scripts/profile_target.py $what > $GRAFT_REPO_ROOT/$O/$what.log 2>&1)
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $O/${what}_kernel_stats.csv || echo "no stats for $what"
  if [ "$what" = c1 ] || [ "$what" = large_loop ] || [ "$what" = large_loop_tile32 ] || [ "$what" = c3_loop ]; then
    t=$(find $d -name '*kernel_trace.csv' | head -1)
    [ -n "$t" ] && python scripts/one_solve_timeline.py "$t" > $O/${what}_one_solve_timeline.csv
  fi
  rm -rf $d $O/$what.log
done
# SQ counters (this build; the round-3 build beside it when its library travelled along)
PASS_TIMEOUT=90 bash scripts/sq_counters.sh c1 r04/sq_c1 "sweepKernel|reduceSchur|solveCombined" > $O/sq_c1.txt 2>&1
PASS_TIMEOUT=90 bash scripts/sq_counters.sh large_loop r04/sq_large_loop "sweepKernel|schurTwoStage|backsub|combineSystem|solveCombined" > $O/sq_large_loop.txt 2>&1
if [ "$1" != "quick" ] && [ -f dsopp_amd/lib_r03/libdsopp_hip.so ]; then
  DSOPP_HIP_LIB=$GRAFT_REPO_ROOT/dsopp_amd/lib_r03/libdsopp_hip.so PASS_TIMEOUT=90 bash scripts/sq_counters.sh large_loop r04/sq_large_loop_round3_build "sweepKernel|schurTwoStage|backsub" > $O/sq_large_loop_round3_build.txt 2>&1
fi
# texture-path counters of the large-window sweep (TA / TCP / TD busy and stall cycles, L2 request latency)
PASS_TIMEOUT=60 COUNTER_SETS="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum;TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum;TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum;TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum;TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN2_sum;TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum;TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum;TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum;TD_TD_BUSY_sum TD_TC_STALL_sum;GRBM_GUI_ACTIVE" \
  bash scripts/sq_counters.sh large_loop r04/texture_path_large_loop "sweepKernel<double, true, true, true, false, false" > $O/texture_path_large_loop.txt 2>&1
# gather-rate probe (what the texture path / fabric charge for the sweep's footprint pattern, no arithmetic)
[ -x scripts/probes/bin/gather_rate_probe ] && timeout 120 scripts/probes/bin/gather_rate_probe > $O/gather_rate_probe.txt 2>&1
# TCC traffic
timeout 600 python scripts/pmc_traffic.py c1 > $O/pmc_c1.log 2>&1; cp gpurun_out/pmc_traffic.json $O/pmc_traffic_c1.json 2>/dev/null
timeout 600 python scripts/pmc_traffic.py large > $O/pmc_large.log 2>&1; cp gpurun_out/pmc_traffic_large.json $O/pmc_traffic_large.json 2>/dev/null
# bench lines
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu 2>$O/bench_steps20.err | grep "^{" > $O/bench_steps20.json
python bench.py 2>$O/bench.err | grep "^{" > $O/bench.json
python - <<'PY'
import json
for f in ("bench", "bench_steps20"):
    try:
        d = json.load(open(f"gpurun_out/r04/{f}.json"))
        print(f, round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 5), "roofline", d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
PY
ls $O
