#!/bin/bash
# Round-4 GPU call 3: the one-lane-per-slot alignment kernel (default build; _v4a = without the cache-row prefetch) against round 3's
# lane-pair kernel (_v3), and the row-per-frame pose optimiser (default) against the wave-per-frame one (PLSVO_POSEOPT_THREADS=64).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04c
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== parity through the default build"
PLSVO_SWEEP_SEEDS=40 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not rccl and not config4 and not bench_distributed and not full_size" > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
lib() { if [ "$1" = "." ]; then echo $R/pl-svo_amd/libplsvo_hip.so; else echo $R/pl-svo_amd/libplsvo_hip$1.so; fi; }
bench() {  # lib tag config extra...
  L=$1; T=$2; CFG=$3; shift 3
  PLSVO_HIP_LIB=$(lib $L) timeout 600 python bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-latency "$@" > $O/bench_c$CFG$L.$T.json 2> $O/bench_c$CFG$L.$T.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_c$CFG$L.$T.json"))
    print("config $CFG lib '$L' $T $*: %.0f frames/s" % d["value"], d["kernel_ms_per_step"], "gn_iters", d.get("chi2_ties", {}).get("gn_iterations_per_step"), "unarmed", d.get("chi2_ties", {}).get("near_ties_without_terms"), "frac", d["roofline"]["frac"])
except Exception as e:
    print("config $CFG lib '$L' $T failed", e)
PY
}
for L in _v3 . _v4a . _v3; do bench $L ab 2; done
PLSVO_POSEOPT_THREADS=64 bench . po64 2
for L in _v3 .; do bench $L ab 3; done
bench . ab 5; PLSVO_POSEOPT_THREADS=64 bench . po64 5
for L in _v3 .; do
  echo "== latency, lib '$L'"
  PLSVO_HIP_LIB=$(lib $L) timeout 300 python tools/latency_sweep.py --batches 1,8,64,512 --steps 100 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print({k:d[k] for k in ('B','align_us_back_to_back','align_kernel_us_hipevent','step_us_back_to_back','gn_iters_mean')})
"
done
