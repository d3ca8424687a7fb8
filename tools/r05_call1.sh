#!/bin/bash
# Round-5 GPU call 1: (a) the new parity cases -- config 3 at one wave per frame (sweep, tie case), every launch shape at 1280x720 and at a
# frame whose coarsest level is a partial tile in both dimensions, the pose-record argument checks; (b) the phase table of the alignment
# kernel at B = 1 and B = 8 (instrumented build) and of the pose optimiser; (c) the small-batch operating points and the drop-in's
# per-call split on THIS box, as the baseline the round's latency work is compared with.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05a
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sequence.py -m gpu -q -x -k "(every_launch_shape and sparse) or near_tie or (seed_sweep and config3-one) or grid_rule" > $O/pytest_new.log 2>&1; tail -3 $O/pytest_new.log
for B in 1 8; do
  TIMING_BATCH=$B timeout 300 python tools/gpu_phase_timing.py > $O/phase_ticks_b$B.log 2>&1; cat $O/phase_ticks_b$B.log | tail -4
done
TIMING_BATCH=1 TIMING_THREADS=256 timeout 300 python tools/gpu_poseopt_timing.py > $O/poseopt_ticks_b1.log 2>&1; tail -2 $O/poseopt_ticks_b1.log
timeout 600 python tools/latency_sweep.py --batches 1,8,64 --threads 0 --steps 100 --out $O/latency_baseline.json > $O/latency_baseline.log 2>&1; tail -4 $O/latency_baseline.log | cut -c1-600
timeout 300 python tools/adapter_latency.py 200 > $O/adapter_latency.log 2>&1; tail -5 $O/adapter_latency.log
