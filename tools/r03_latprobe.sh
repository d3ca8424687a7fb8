#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for L in "$@"; do
  echo "== lib '$L'"
  PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip$L.so timeout 300 python tools/latency_sweep.py --batches 1,8 --steps 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:d[k] for k in ('B','align_us_back_to_back','align_kernel_us_hipevent','gn_iters_mean','gn_iters_max')})
"
done
