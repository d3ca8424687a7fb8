#!/bin/bash
# Round-6 GPU call 11: closing checks of the final tree -- the -m gpu suite as the driver runs it, smoke(), determinism of the two-workgroup shape,
# configs[4] with the final pose-optimiser kernel, and the pose optimiser's small-batch shapes at 256 against 512 threads per frame
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06j
mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider ) > $O/gputest.log 2>&1; grep -E "passed|failed|real" $O/gputest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python tools/pair_determinism.py > $O/pair_determinism.log 2>&1; cat $O/pair_determinism.log | grep distinct
timeout 400 python bench.py --config 5 > $O/bench_config5.json 2> $O/bench_config5.err; cut -c1-260 $O/bench_config5.json
for B in 1 8 64; do TIMING_BATCH=$B TIMING_THREADS=256,512 timeout 200 python tools/gpu_poseopt_timing.py 2>&1 | grep "T=" ; done | tee $O/poseopt_phase_ticks_256_512.log
