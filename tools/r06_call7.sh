#!/bin/bash
# Round-6 GPU call 7: float rows (192 B per slot, no rebuild) against the 64-byte record + rebuild in the one-wave-per-frame shape
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06f
mkdir -p $O
cd $R
( PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip_frows.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "one-wave or near_tie or every_launch_shape or matches_oracle or full-motion" ) > $O/parity_frows.log 2>&1
echo "parity frows: $(tail -1 $O/parity_frows.log)"
export PLSVO_BENCH_LAUNCH_ORDER=staged
bash tools/ab_bench.sh r06f 2 "" _frows
BENCH_ARGS="--config 3" bash tools/ab_bench.sh r06f_c3 1 "" _frows
