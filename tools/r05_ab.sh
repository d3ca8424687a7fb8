#!/bin/bash
# A/B of library builds on ONE box: small-batch operating points per lib, and the pass's sub-phase ticks for the *_t2 (timing2) builds.
# usage: tools/r05_ab.sh <tag> "<lib suffixes for latency, '' = default>" "<timing2 lib suffixes>"
TAG=$1; LIBS=$2; TLIBS=$3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for L in $LIBS; do
  [ "$L" = "default" ] && S="" || S="_$L"
  echo "== latency, lib '$L'"
  PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip$S.so timeout 300 python tools/latency_sweep.py --batches ${AB_BATCHES:-1,8,64} --steps 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:d[k] for k in ('B','align_us_back_to_back','poseopt_us_back_to_back','step_us_back_to_back','gn_iters_mean','gn_iters_max')})
" | tee -a $O/latency_$L.log
done
for L in $TLIBS; do
  echo "== phase ticks, lib '$L' (slots: setup | rest of pass | after the pass | A | B | C)"
  for B in 1 8; do
    PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip_$L.so TIMING_BATCH=$B timeout 300 python tools/gpu_phase_timing.py 2>/dev/null | tee -a $O/ticks_${L}_b$B.log
  done
done
