#!/bin/bash
# experiment: where the kernel arguments live (HIP_FORCE_DEV_KERNARG) and whether the first 16 argument dwords are preloaded into
# scalar registers by the dispatcher (-mllvm -amdgpu-kernarg-preload-count=16: dsopp_amd/lib_exp_kp, built by
#   DSOPP_HIP_OUT=$PWD/dsopp_amd/lib_exp_kp DSOPP_HIP_EXTRA_FLAGS="-mllvm -amdgpu-kernarg-preload-count=16" bash dsopp_amd/csrc/build.sh)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
out=$O/time_kernarg_ab.txt
: > $out
for rep in 1 2; do
  for lib in lib lib_exp_kp; do
    for dk in unset 0 1; do
      if [ $dk = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$dk; fi
      r=$(DSOPP_HIP_LIB=$PWD/dsopp_amd/$lib/libdsopp_hip.so timeout 300 python scripts/threshold_sweep.py 7 2000 2>/dev/null | grep "us per" | sed 's/.*: //')
      echo "rep $rep  $lib  HIP_FORCE_DEV_KERNARG=$dk  7 KF / 2000: $r" | tee -a $out
    done
  done
done
unset HIP_FORCE_DEV_KERNARG
for lib in lib lib_exp_kp; do
  r=$(DSOPP_HIP_LIB=$PWD/dsopp_amd/$lib/libdsopp_hip.so timeout 300 python scripts/threshold_sweep.py 12 50000 2>/dev/null | grep "us per" | sed 's/.*: //')
  echo "$lib  12 KF / 50000: $r" | tee -a $out
  r=$(DSOPP_HIP_LIB=$PWD/dsopp_amd/$lib/libdsopp_hip.so timeout 300 python scripts/threshold_sweep.py 15 5000 2>/dev/null | grep "us per" | sed 's/.*: //')
  echo "$lib  15 KF / 5000: $r" | tee -a $out
done
