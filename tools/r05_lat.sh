#!/bin/bash
# Round-5 latency iteration: parity subset on the current build, phase table of the alignment kernel at B = 1 / 8 (instrumented build),
# small-batch operating points, the drop-in's per-call split.   usage: tools/r05_lat.sh <tag> [pytest -k expression]   -> gpurun_out/<tag>/
TAG=${1:-r05x}
K=${2:-"(every_launch_shape and sparse) or matches_oracle or batch_equals_single or near_tie or long_lines or adversarial or edge_cases or holes"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
PLSVO_SWEEP_SEEDS=30 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "$K" > $O/pytest_subset.log 2>&1; tail -3 $O/pytest_subset.log
for B in 1 8; do
  TIMING_BATCH=$B timeout 300 python tools/gpu_phase_timing.py > $O/phase_ticks_b$B.log 2>&1; tail -3 $O/phase_ticks_b$B.log
done
TIMING_BATCH=1 TIMING_THREADS=256 timeout 300 python tools/gpu_poseopt_timing.py > $O/poseopt_ticks_b1.log 2>&1; tail -1 $O/poseopt_ticks_b1.log
timeout 600 python tools/latency_sweep.py --batches 1,8,64,512 --threads 0 --steps 100 --out $O/latency.json > $O/latency.log 2>&1; tail -4 $O/latency.log | cut -c1-420
timeout 300 python tools/adapter_latency.py 200 > $O/adapter_latency.log 2>&1; tail -4 $O/adapter_latency.log
