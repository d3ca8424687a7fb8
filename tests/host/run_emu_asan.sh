#!/bin/bash
# The GPU parity tests against an AddressSanitizer build of the host emulation (tests/host/build_emu.sh): every load and store of the
# kernels and of the C ABI checked against the bounds of the hipMalloc'ed buffers.  ~9 minutes on 8 cores; not part of the pytest suite.
# Round 3, HEAD of that day (ucontext fibres; the hand-written switch is annotated the same way): 82 passed, no AddressSanitizer report (the one failure is test_hip_resident_chain_equals_the_per_call_chain,
# whose 1e-9 tolerance is tuned to the gfx950 arithmetic: tests/test_emu_parity.py leaves it out for the same reason).
# usage: tests/host/run_emu_asan.sh [pytest -k expression]
R=$(cd $(dirname $0)/../.. && pwd)
CXX=${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
OUT=$(mktemp -d /tmp/plsvo_emu_asan.XXXX)
$R/tests/host/build_emu.sh $OUT "" -fsanitize=address -fno-omit-frame-pointer -g || exit 1
K=${1:-"(test_gpu_parity and (halfsample or matches_oracle and not config3 or every_launch_shape or long_lines or edge_cases or fewer_patches or border_features or single_linearisation or pose_optimizer and not seed_sweep or adversarial or batch_equals or mixed_batch)) or test_golden or test_depth_filter or test_structure_opt or test_match_direct or test_reproject_trajectory or (test_sequence and (matches_the_oracle_chain or grid_rule or resident))"}
cd $R
LD_PRELOAD=$($CXX -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 \
  OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 PLSVO_HIP_LIB=$OUT/libplsvo_hip_emu.so \
  python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_depth_filter.py tests/test_structure_opt.py tests/test_match_direct.py \
    tests/test_reproject_trajectory.py tests/test_sequence.py -m gpu -q -n 6 -p no:cacheprovider -k "$K" 2>&1 | tee $OUT/run.log | tail -5
echo "AddressSanitizer reports: $(grep -c 'ERROR: AddressSanitizer' $OUT/run.log)   (log: $OUT/run.log)"
