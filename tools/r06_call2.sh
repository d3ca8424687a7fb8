#!/bin/bash
# Round-6 GPU call 2: LDS-DMA semantics probe, parity of the LDS-DMA alignment shape, A/B against the register-parked prefetch (NO_DMA build)
# in both launch-order policies.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06b
mkdir -p $O
cd $R
./tools/micro/glds_probe > $O/glds_probe.log 2>&1; cat $O/glds_probe.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "one-wave or near_tie or every_launch_shape or matches_oracle or full-motion or long_lines or adversarial or batch_equals or mixed_batch or border or full_size or launch_order" ) > $O/parity_dma.log 2>&1
tail -3 $O/parity_dma.log
for ORDER in staged refresh; do
  export PLSVO_BENCH_LAUNCH_ORDER=$ORDER
  echo "== launch order $ORDER"
  bash tools/ab_bench.sh r06b_$ORDER 2 "" _nodma
done
