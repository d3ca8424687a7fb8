#!/bin/bash
# latency A/B of library builds + a parity subset on the default build.  usage: tools/ab_latency.sh <tag> [lib suffixes...]
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "matches_oracle or batch_equals_single or adversarial or long_lines or fewer_patches or static or seed_sweep" > $O/pytest_subset.log 2>&1; tail -4 $O/pytest_subset.log
for L in "$@"; do
  echo "== lib '$L'"
  PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip$L.so timeout 300 python tools/latency_sweep.py --batches 1,8,64 --steps 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:d[k] for k in ('B','align_us_back_to_back','align_kernel_us_hipevent','step_us_back_to_back','gn_iters_mean','gn_iters_max')})
"
done
