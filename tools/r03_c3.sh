#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() { # label lib extra-args
  PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip$2.so timeout 300 python bench.py $3 --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', d['value'], d['kernel_ms_per_step'], (d.get('chi2_ties') or {}).get('decided_on_exact_float_sums'))
"
}
run c3_tiled128 "" "--config 3"
run c3_linear128 _t64 "--config 3"
run c2b8192_tiled128 "" "--batch 8192"
run c2b8192_linear128 _t64 "--batch 8192"
