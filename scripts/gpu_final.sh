#!/bin/bash
# end-of-round measurement pass on the GPU box: tests, rocprofv3 kernel statistics per workload, one-solve timelines, counters,
# shard costs, bench lines.  Everything lands under gpurun_out/final/ (copy what is to be judged into profiles/).
cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
timeout 2000 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
bash scripts/collect_profiles.sh final > $O/collect.log 2>&1; cp gpurun_out/profiles_final/*_kernel_stats.csv $O/ 2>/dev/null
for w in c1 c3_loop large_loop; do bash scripts/gpu_timeline.sh $w > /dev/null 2>&1; cp gpurun_out/${w}_one_solve_timeline.csv $O/; done
python scripts/shard_cost.py c3 2>/dev/null | grep "^{" > $O/shard_cost_c3.json
python scripts/shard_cost.py c4 2>/dev/null | grep "^{" > $O/shard_cost_c4.json
timeout 900 python scripts/pmc_traffic.py c1 > $O/pmc_c1.log 2>&1; cp gpurun_out/pmc_traffic.json $O/pmc_traffic_c1.json 2>/dev/null
timeout 900 python scripts/pmc_traffic.py large > $O/pmc_large.log 2>&1; cp gpurun_out/pmc_traffic_large.json $O/pmc_traffic_large.json 2>/dev/null
python scripts/group_bench.py --devices 0,0 --workload c3 --blocks 7 2>/dev/null | grep "^{" > $O/group_bench_c3_two_shards_one_gpu.json
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu 2>$O/bench_steps20.err | grep "^{" > $O/bench_steps20.json
DSOPP_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --no-cpu 2>$O/bench_2rank.err | grep "^{" > $O/bench_2rank_one_gpu.json
python bench.py 2>$O/bench.err | grep "^{" > $O/bench.json
python - <<'PY'
import json
for f in ("bench", "bench_steps20", "bench_2rank_one_gpu"):
    try:
        d = json.load(open(f"gpurun_out/final/{f}.json"))
        print(f, round(d["value"], 1), d["unit"], "ms/step", round(d["ms_per_step"], 5), "roofline", d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
PY
ls -la $O
