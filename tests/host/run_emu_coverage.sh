#!/bin/bash
# Line coverage of the DEVICE sources by the `-m gpu` tests, measured on the host emulation build (clang --coverage, read with gcov):
# which lines of the kernels and of the C ABI the GPU suite never executes.  ~1 minute.  Prints one line per source file and writes
# the annotated sources to <out>/<object>/*.gcov ("#####" marks a line that never ran).
# Round 3 (before the tests this run asked for were added): kernels 97-100 % (align_kernels.hip 554 of 557 lines, the three being the
# inconsistent-host-layout bail-out and the launcher's bad-thread-count default), plsvo_capi.hip 84 % (argument checks and the
# benchmark's utility entry points); it found a device branch no test reached -- a CONVERGED line seed -- now in
# test_hip_update_seeds_edge_cases, and led to test_abi_utility_entry_points / test_every_entry_point_rejects_malformed_arguments.
# usage: tests/host/run_emu_coverage.sh [out dir]
R=$(cd $(dirname $0)/../.. && pwd)
OUT=${1:-$(mktemp -d /tmp/plsvo_emu_cov.XXXX)}
EMU_LDFLAGS=--coverage $R/tests/host/build_emu.sh $OUT/build "" --coverage -Xclang "-coverage-version=B14*" || exit 1
cd $R
OMP_NUM_THREADS=1 OPENBLAS_NUM_THREADS=1 PLSVO_SWEEP_SEEDS=12 PLSVO_HIP_LIB=$OUT/build/libplsvo_hip_emu.so \
  python -m pytest tests -m gpu -q -n 6 -p no:cacheprovider \
    -k "not rccl and not config4 and not bench_distributed and not test_gpu_adapter and not full_size and not resident_chain_equals and not every_float" 2>&1 | tail -2
for f in align_kernels poseopt_kernels chain_kernels match_kernels seeds_kernels structopt_kernels pyramid_kernels plsvo_capi; do
  mkdir -p $OUT/$f
  (cd $R && gcov-11 -o $OUT/build/$f.gcda $OUT/build/$f.gcda > $OUT/$f.log 2>&1; mv *.gcov $OUT/$f/ 2>/dev/null)
  grep -A1 "File 'pl-svo_amd/csrc/$f" $OUT/$f.log | grep -E "File|Lines" | paste - -
done
echo "annotated sources under $OUT"
