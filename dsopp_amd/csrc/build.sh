#!/bin/bash
# Builds the HIP library for gfx950 in-tree: dsopp_amd/lib/libdsopp_hip.so (+ libdsopp_hip_tools.so: counter-calibration kernels)
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function ${DSOPP_HIP_EXTRA_FLAGS:-}"
pids=()
for src in pyramid pba align depth_estimation comm calibration window_group; do
  if [ -f "$HERE/$src.hip" ]; then
    $HIPCC $FLAGS -c "$HERE/$src.hip" -o "$OUT/$src.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
objs=()
for src in pyramid pba align depth_estimation comm window_group; do [ -f "$OUT/$src.o" ] && objs+=("$OUT/$src.o"); done
$HIPCC --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o "$OUT/libdsopp_hip.so"
# measurement aids that are not part of the product (gather kernels of known geometry for the counter calibration, scripts/pmc_target.py)
[ -f "$OUT/calibration.o" ] && $HIPCC --offload-arch=gfx950 -shared -fPIC "$OUT/calibration.o" -o "$OUT/libdsopp_hip_tools.so"
echo "built $OUT/libdsopp_hip.so"
