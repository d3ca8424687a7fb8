#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/r03s; mkdir -p $O
PLSVO_SWEEP_SEEDS=30 timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "seed_sweep or matches_oracle or launch_shape or batch_equals_single or adversarial or long_lines or static" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 60 python tools/latency_sweep.py --batches 1,8 --steps 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:d[k] for k in ('B','align_us_back_to_back','step_us_back_to_back','gn_iters_max')})
"
timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('default', d['value'], d['kernel_ms_per_step'], d['chi2_ties']['decided_on_exact_float_sums'])
"
timeout 100 python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config3', d['value'], d['kernel_ms_per_step'])
"
