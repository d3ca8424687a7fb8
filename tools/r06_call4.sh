#!/bin/bash
# Round-6 GPU call 4: A/B of two small changes to align_fused_kernel<64>'s round (staged launch order): the reference-patch rebuild hoisted
# in front of the first use of the window (hoist), an L2 prefetch of the next round's windows from the previous iteration's positions (pf)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06c
mkdir -p $O
cd $R
( PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip_hoistpf.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "one-wave or near_tie or every_launch_shape or matches_oracle or full-motion or long_lines" ) > $O/parity_hoistpf.log 2>&1
tail -2 $O/parity_hoistpf.log
export PLSVO_BENCH_LAUNCH_ORDER=staged
bash tools/ab_bench.sh r06c 2 "" _hoist _pf _hoistpf
