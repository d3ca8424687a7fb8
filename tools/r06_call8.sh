#!/bin/bash
# Round-6 GPU call 8: per-slot record and 3-D point arrays as planes per job (coalesced) against the array-of-structures layout
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06g
mkdir -p $O
cd $R
( PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip_soa.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "one-wave or near_tie or every_launch_shape or matches_oracle or full-motion" ) > $O/parity_soa.log 2>&1
echo "parity soa: $(tail -1 $O/parity_soa.log)"
export PLSVO_BENCH_LAUNCH_ORDER=staged
bash tools/ab_bench.sh r06g 2 "" _soa
BENCH_ARGS="--config 3" bash tools/ab_bench.sh r06g_c3 1 "" _soa
