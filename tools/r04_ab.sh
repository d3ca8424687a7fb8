#!/bin/bash
# Round-4 A/B call on one MI355X box: variant libraries (PLSVO_HIP_LIB) interleaved with the default build so box-to-box differences cancel.
# usage: tools/r04_ab.sh <tag> "<parity libs>" "<bench-config-2 libs>" "<bench-config-3 libs>" "<latency libs>"   ('.' = the default build)
# -> gpurun_out/<tag>/   (summary lines on stdout)
TAG=${1:-r04a}
PARITY=${2:-}
B2=${3:-}
B3=${4:-}
LAT=${5:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
lib() { if [ "$1" = "." ]; then echo $R/pl-svo_amd/libplsvo_hip.so; else echo $R/pl-svo_amd/libplsvo_hip$1.so; fi; }
for L in $PARITY; do
  LIB=$(lib $L)
  [ -f $LIB ] || { echo "missing $LIB"; continue; }
  for A in "4373 64" "5348 128 config3"; do
    echo "tie case $A lib '$L': $(PLSVO_HIP_LIB=$LIB timeout 120 python tests/host/emu_tie_case_runner.py $A 2>/dev/null | tail -1 | cut -c1-260)"
  done
done
for L in $PARITY; do
  [ "$L" = "." ] && continue
  LIB=$(lib $L)
  PLSVO_HIP_LIB=$LIB PLSVO_SWEEP_SEEDS=${SWEEP:-30} timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sequence.py -m gpu -q -x \
    -k "matches_oracle or batch_equals_single or adversarial or long_lines or fewer_patches or static or seed_sweep or launch_shape or chain" > $O/pytest$L.log 2>&1
  echo "== parity through libplsvo_hip$L.so: $(grep -E 'passed|failed|error' $O/pytest$L.log | tail -1)"
done
bench() {  # config lib rep extra...
  CFG=$1; L=$2; rep=$3; shift 3
  PLSVO_HIP_LIB=$(lib $L) timeout 600 python bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-latency "$@" > $O/bench_c$CFG$L.$rep.json 2> $O/bench_c$CFG$L.$rep.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_c$CFG$L.$rep.json"))
    print("config $CFG $* lib '$L' rep $rep: %.0f frames/s" % d["value"], d["kernel_ms_per_step"], d.get("chi2_ties"), "frac", d["roofline"]["frac"])
except Exception as e:
    print("config $CFG lib '$L' rep $rep failed", e)
PY
}
for rep in ${REPS:-1}; do
  for L in $B2; do bench 2 $L $rep; done
done
for L in $B3; do bench 3 $L 1; bench 3 $L 16k --batch 16384; done
for L in $LAT; do
  echo "== latency, lib '$L'"
  PLSVO_HIP_LIB=$(lib $L) timeout 300 python tools/latency_sweep.py --batches 1,8,64 --steps 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:d[k] for k in ('B','align_us_back_to_back','align_kernel_us_hipevent','step_us_back_to_back','gn_iters_mean')})
"
done
