#!/bin/bash
# experiment: launch prologues (argument preload + one burst of scalar loads per kernel head + the solve launch's ticket not waited for)
#   lib          = build under test, default flags           lib_exp_kp  = round-5 sources at d8ab8f0 + argument preload
#   lib_exp_pro  = build under test + argument preload
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
out=$O/time_prologue_ab.txt
: > $out
for rep in 1 2 3; do
  for lib in ${LIBS:-lib lib_exp_kp lib_exp_pro}; do
    [ -f dsopp_amd/$lib/libdsopp_hip.so ] || continue
    r=$(DSOPP_HIP_LIB=$PWD/dsopp_amd/$lib/libdsopp_hip.so timeout 300 python scripts/threshold_sweep.py 7 2000 2>/dev/null | grep "us per" | sed 's/.*: //')
    echo "rep $rep  $lib  7 KF / 2000: $r" | tee -a $out
  done
done
for lib in ${LIBS:-lib lib_exp_kp lib_exp_pro}; do
  [ -f dsopp_amd/$lib/libdsopp_hip.so ] || continue
  for w in "7 20000" "12 50000" "15 5000"; do
    r=$(DSOPP_HIP_LIB=$PWD/dsopp_amd/$lib/libdsopp_hip.so timeout 300 python scripts/threshold_sweep.py $w 2>/dev/null | grep "us per" | sed 's/.*: //')
    echo "$lib  $w: $r" | tee -a $out
  done
done
if [ -n "$TESTLIB" ]; then
  DSOPP_HIP_LIB=$PWD/dsopp_amd/$TESTLIB/libdsopp_hip.so timeout 1500 python -m pytest ${TESTS:-tests/test_gpu_pba.py tests/test_gpu_pba_edge.py tests/test_gpu_degenerate.py} -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_prologue.log
fi
