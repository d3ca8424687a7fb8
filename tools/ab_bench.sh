#!/bin/bash
# A/B of library builds on the default bench command (no tests).  usage: tools/ab_bench.sh <tag> <reps> [lib suffixes...]
TAG=$1; REPS=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for rep in $(seq 1 $REPS); do
for L in "$@"; do
  LIB=$R/pl-svo_amd/libplsvo_hip$L.so
  PLSVO_HIP_LIB=$LIB timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency $BENCH_ARGS > $O/bench$L.$rep.json 2> $O/bench$L.$rep.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench$L.$rep.json"))
    print("$L rep $rep", d["value"], d["kernel_ms_per_step"], (d.get("chi2_ties") or {}).get("decided_on_exact_float_sums"))
except Exception as e:
    print("$L rep $rep failed", e)
PY
done
done
