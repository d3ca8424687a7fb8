#!/bin/bash
# The GPU parity tests against an AddressSanitizer build of the host emulation (tests/host/build_emu.sh): every load and store of the
# kernels and of the C ABI checked against the bounds of the hipMalloc'ed buffers.  ~6 minutes on 8 cores; not part of the pytest suite.
# Round 3: 92 passed (all three seed sweeps at full length included), no AddressSanitizer report.  Round 4: 100 passed, round 5: 117 passed, round 6: 118 passed (60-seed sweeps), no report.
# usage: tests/host/run_emu_asan.sh [pytest -k expression]
R=$(cd $(dirname $0)/../.. && pwd)
CXX=${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
OUT=$(mktemp -d /tmp/plsvo_emu_asan.XXXX)
$R/tests/host/build_emu.sh $OUT "" -fsanitize=address -fno-omit-frame-pointer -g || exit 1
K=${1:-"not rccl and not config4 and not bench_distributed and not test_gpu_adapter and not full_size and not resident_chain_equals and not every_float"}
cd $R
LD_PRELOAD=$($CXX -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1 \
  OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 PLSVO_HIP_LIB=$OUT/libplsvo_hip_emu.so \
  python -m pytest tests -m gpu -q -n 6 -p no:cacheprovider -k "$K" 2>&1 | tee $OUT/run.log | tail -5
echo "AddressSanitizer reports: $(grep -c 'ERROR: AddressSanitizer' $OUT/run.log)   (log: $OUT/run.log)"
