#!/bin/bash
# GPU call for library VARIANTS that have never run (round 3 left four: _bc = byte-record cache, _li = coarse level of the current image in
# LDS, _bcli = both, _dpp = DPP near-tie chains; all bit-identical to the default build under host emulation, tests/test_emu_parity.py).
# For each suffix: the parity subset THROUGH that library (PLSVO_HIP_LIB), then two repetitions of the default bench command and
# config 3, interleaved with the default build so that box-to-box differences cancel; then the single-frame latency sweep.
# Build the variants in the container first (the .so files travel with the snapshot):
#   make -C pl-svo_amd/csrc byte_cache lds_img bc_lds_img tie_recompute      (_tr: parity variant, rebuilds the chi2 terms of unarmed near ties)
#   tools/build_patched.sh tools/patches/slot_parallel_exact_sum_dpp.patch dpp
# usage: tools/ab_variants.sh <tag> _bc _li _bcli _dpp        -> gpurun_out/<tag>/
TAG=${1:-ab}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for L in "$@"; do
  LIB=$R/pl-svo_amd/libplsvo_hip$L.so
  [ -f $LIB ] || { echo "missing $LIB"; continue; }
  PLSVO_HIP_LIB=$LIB PLSVO_SWEEP_SEEDS=30 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sequence.py -m gpu -q \
    -k "matches_oracle or batch_equals_single or adversarial or long_lines or fewer_patches or static or seed_sweep or launch_shape or chain" > $O/pytest$L.log 2>&1
  echo "== parity through libplsvo_hip$L.so: $(grep -E 'passed|failed|error' $O/pytest$L.log | tail -1)"
  # the two frames whose near tie falls on an unarmed iteration: the default build parts from the oracle there, _tr must not
  for A in "4373 64" "5348 128 config3"; do
    PLSVO_HIP_LIB=$LIB timeout 120 python tests/host/emu_tie_case_runner.py $A 2>/dev/null | tail -1 | cut -c1-220
  done
done
for rep in 1 2; do
  for L in "" "$@"; do
    LIB=$R/pl-svo_amd/libplsvo_hip$L.so
    for CFG in 2 3; do
      PLSVO_HIP_LIB=$LIB timeout 600 python bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/bench_c$CFG$L.$rep.json 2> $O/bench_c$CFG$L.$rep.err
      python - <<PY
import json
try:
    d = json.load(open("$O/bench_c$CFG$L.$rep.json"))
    print("config $CFG lib '$L' rep $rep: %.0f frames/s" % d["value"], d["kernel_ms_per_step"], d.get("chi2_ties", {}).get("decided_on_exact_float_sums"))
except Exception as e:
    print("config $CFG lib '$L' rep $rep failed", e)
PY
    done
  done
done
for L in "" "$@"; do
  echo "== latency, lib '$L'"
  PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip$L.so timeout 300 python tools/latency_sweep.py --batches 1,8,64 --steps 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:d[k] for k in ('B','align_us_back_to_back','align_kernel_us_hipevent','step_us_back_to_back','gn_iters_mean')})
"
done
