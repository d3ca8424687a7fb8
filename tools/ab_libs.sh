#!/bin/bash
# A/B of experiment builds of libplsvo_hip (PLSVO_HIP_LIB): parity smoke + latency sweep per library.  usage: tools/ab_libs.sh "<batches>" lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
BATCHES=$1; shift
for L in "$@"; do
  echo "== $L"
  PLSVO_HIP_LIB=$R/pl-svo_amd/$L timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "matches_oracle or batch_equals or long_lines or edge" 2>&1 | grep -E "passed|failed" | tail -1
  PLSVO_HIP_LIB=$R/pl-svo_amd/$L timeout 900 python tools/latency_sweep.py --batches $BATCHES --threads 0 --steps 20 2>&1 | grep "^{" | cut -c1-200
done
