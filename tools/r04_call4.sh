#!/bin/bash
# Round-4 GPU call 4: A/B builds on the one-lane-per-slot kernel (byte records, staged level, gather a round ahead and combinations), then the whole GPU suite.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04d
mkdir -p $O
cd $R
export TMPDIR=/tmp
lib() { if [ "$1" = "." ]; then echo $R/pl-svo_amd/libplsvo_hip.so; else echo $R/pl-svo_amd/libplsvo_hip$1.so; fi; }
bench() {  # lib tag config extra...
  L=$1; T=$2; CFG=$3; shift 3
  PLSVO_HIP_LIB=$(lib $L) timeout 600 python bench.py --config $CFG --steps 4 --warmup 1 --no-cpu-baseline --no-latency "$@" > $O/bench_c$CFG$L.$T.json 2> $O/bench_c$CFG$L.$T.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_c$CFG$L.$T.json"))
    print("config $CFG lib '$L' $T $*: %.0f frames/s" % d["value"], d["kernel_ms_per_step"], "gn_iters", d.get("chi2_ties", {}).get("gn_iterations_per_step"), "frac", d["roofline"]["frac"])
except Exception as e:
    print("config $CFG lib '$L' $T failed", e)
PY
}
for L in . _li _bc _bcli _ga _bcga _bcliga .; do bench $L ab 2; done
for L in _bcliga _bcli; do bench $L ab 3; done
echo "== the whole GPU suite through the default build"
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
