#!/bin/bash
# Round-6 GPU call 6: the packed slot layout (first-fit decreasing into 64-slot rounds) and, on top of it, the column-strip mirror
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06e
mkdir -p $O
cd $R
for L in "" _strip; do
( PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip$L.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "one-wave or near_tie or every_launch_shape or matches_oracle or full-motion or long_lines or two_workgroups or halfsample or upload or batch_equals" ) > $O/parity$L.log 2>&1
echo "parity $L: $(tail -1 $O/parity$L.log)"
done
export PLSVO_BENCH_LAUNCH_ORDER=staged
bash tools/ab_bench.sh r06e 2 "" _strip
BENCH_ARGS="--config 3" bash tools/ab_bench.sh r06e_c3 1 "" _strip
