#!/bin/bash
# round-3 GPU call: full GPU suite (no -x: every failure is reported) + A/B of library builds on the default bench command
# usage: tools/ab_suite.sh <tag> [lib suffixes...]     -> gpurun_out/<tag>/
TAG=${1:-r03a}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log
cp gpurun_out/parity_seed_sweep_*.json $O/ 2>/dev/null
for L in "$@"; do
  LIB=$R/pl-svo_amd/libplsvo_hip$L.so
  for rep in 1 2; do
    PLSVO_HIP_LIB=$LIB timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/bench$L.$rep.json 2> $O/bench$L.$rep.err
    python - <<PY
import json
try:
    d = json.load(open("$O/bench$L.$rep.json"))
    print("$L rep $rep", d["value"], d["kernel_ms_per_step"], d.get("chi2_ties"))
except Exception as e:
    print("$L rep $rep failed", e)
PY
  done
done
