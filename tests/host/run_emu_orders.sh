#!/bin/bash
# The emulated GPU parity subset in the emulator's OTHER lane order (WAVE_EMU_ORDER=asc0last: lanes 1..63 ascending, lane 0 of every wave
# last; the default is descending).  Between two rendezvous a lane runs alone, so an exchange through LDS / global memory between lanes that
# is not separated by a fence or barrier sees different values in the two orders and the comparison with the oracle fails in one of them.
# On the hardware such code works only as long as the compiler keeps the store in front of the load -- which it need not when it can prove
# that a lane's own two addresses differ -- so a pass in both orders is the check that every such exchange has its fence.
# Round 3: 74 passed in both orders.
R=$(cd $(dirname $0)/../.. && pwd)
OUT=$(mktemp -d /tmp/plsvo_emu_orders.XXXX)
$R/tests/host/build_emu.sh $OUT || exit 1
K="(test_gpu_parity and (halfsample and shape2 or matches_oracle and not config3 or every_launch_shape or long_lines or edge_cases or fewer_patches or border_features or single_linearisation or pose_optimizer and not seed_sweep or adversarial)) or test_golden or test_depth_filter or test_structure_opt or test_match_direct or test_reproject_trajectory or (test_sequence and (matches_the_oracle_chain or grid_rule))"
cd $R
for ORDER in desc asc0last; do
  echo "== WAVE_EMU_ORDER=$ORDER"
  WAVE_EMU_ORDER=$ORDER OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 PLSVO_HIP_LIB=$OUT/libplsvo_hip_emu.so \
    python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_depth_filter.py tests/test_structure_opt.py tests/test_match_direct.py \
      tests/test_reproject_trajectory.py tests/test_sequence.py -m gpu -q -n 6 -p no:cacheprovider -k "$K" 2>&1 | tail -2
done
