#!/bin/bash
# tuning aid: chunks of 64 landmarks per two-stage Schur workgroup (DSOPP_HIP_SCHUR_CHUNKS) against the per-iteration time
cd $GRAFT_REPO_ROOT
for c in 0 1 2 3 4 6; do echo -n "chunks/wg $c: "; DSOPP_HIP_SCHUR_CHUNKS=$c python scripts/threshold_sweep.py 12 50000 1 2>/dev/null | grep "us per"; done
for c in 0 1 2 3; do echo -n "chunks/wg $c: "; DSOPP_HIP_SCHUR_CHUNKS=$c python scripts/threshold_sweep.py 7 20000 2>/dev/null | grep "us per"; done
