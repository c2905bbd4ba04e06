#!/bin/bash
# rocprofv3 kernel-trace statistics of every workload DESIGN.md quotes a kernel time for.  Run on the GPU box from the repo root:
#   bash scripts/collect_profiles.sh [tag]      -> gpurun_out/profiles_<tag>/<workload>_kernel_stats.csv (copy into profiles/)
set -uo pipefail
TAG="${1:-r03}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/profiles_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
for what in c1 c1_isolated large tracker depth activation; do
  d="$OUT/raw_$what"
  rm -rf "$d"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o prof -- python "$ROOT/scripts/profile_target.py" "$what" > "$OUT/$what.log" 2>&1)
  f=$(find "$d" -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" "$OUT/${what}_kernel_stats.csv"; else echo "no stats for $what" >&2; tail -5 "$OUT/$what.log" >&2; fi
  rm -rf "$d"
done
ls -la "$OUT"
