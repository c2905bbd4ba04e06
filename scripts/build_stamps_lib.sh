#!/bin/bash
# tuning aid: the library with phase stamps compiled in (-DDSOPP_HIP_STAMPS) into dsopp_amd/lib_stamps/, selected with
# DSOPP_HIP_LIB=$PWD/dsopp_amd/lib_stamps/libdsopp_hip.so (scripts/dbg_*.py); the shipped library carries no stamps
set -euo pipefail
HERE="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$HERE/dsopp_amd/lib_stamps"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -DDSOPP_HIP_STAMPS ${DSOPP_HIP_EXTRA_FLAGS:-}"
pids=()
for src in pyramid pba align depth_estimation comm window_group; do
  $HIPCC $FLAGS -c "$HERE/dsopp_amd/csrc/$src.hip" -o "$OUT/$src.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC "$OUT"/{pyramid,pba,align,depth_estimation,comm,window_group}.o -o "$OUT/libdsopp_hip.so"
echo "built $OUT/libdsopp_hip.so"
