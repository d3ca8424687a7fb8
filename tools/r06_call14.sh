#!/bin/bash
# Round-6 GPU call 14: the static-pivot-order 6x6 solve (the latency shapes' form) in the throughput shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06l
mkdir -p $O
cd $R
( PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip_ssolve.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "one-wave or near_tie or every_launch_shape or matches_oracle or full-motion or fewer_patches or adversarial or ldlt" ) > $O/parity_ssolve.log 2>&1
echo "parity ssolve: $(tail -1 $O/parity_ssolve.log)"
export PLSVO_BENCH_LAUNCH_ORDER=staged
bash tools/ab_bench.sh r06l 2 "" _ssolve
