#!/bin/bash
# Builds the HIP library for gfx950 in-tree: dsopp_amd/lib/libdsopp_hip.so (+ libdsopp_hip_tools.so: counter-calibration kernels)
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="${DSOPP_HIP_OUT:-$HERE/../lib}"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -amdgpu-kernarg-preload-count=16: the dispatcher hands the first 16 argument words to every wave in scalar registers (gfx940+), so a
# kernel whose first loads depend on its leading pointer arguments alone starts them in its first cycles instead of behind an
# argument fetch (the sweep, reduction and solve launches are laid out for this: profiles/r05/prologue_kernel_stats.txt)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16 ${DSOPP_HIP_EXTRA_FLAGS:-}"
# align.hip: the persistent tracker kernel's control step is ONE wave running a ~600-instruction dependency chain (8 x 8 LDL^T, SE3 exp): the
# default scheduler orders for register pressure and puts whole FMA chains in front of every pivot; max-ilp interleaves the independent
# updates with the reciprocal chains (profiles/r06/tracker_*: control step 1.76 -> see DESIGN.md section 4)
FLAGS_align="${DSOPP_HIP_ALIGN_SCHED--mllvm -amdgpu-sched-strategy=max-ilp}"
pids=()
for src in pyramid pba align depth_estimation comm calibration window_group; do
  if [ -f "$HERE/$src.hip" ]; then
    extra="FLAGS_$src"
    $HIPCC $FLAGS ${!extra:-} -c "$HERE/$src.hip" -o "$OUT/$src.o" &
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
