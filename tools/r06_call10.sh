#!/bin/bash
# Round-6 GPU call 10: the alignment launch of a large batch as S slices on streams of falling priority (whole-frame or per-level launches)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06i
mkdir -p $O
cd $R
export PLSVO_BENCH_LAUNCH_ORDER=staged
run() {   # tag, env assignments...
  TAG=$1; shift
  env "$@" timeout 600 python bench.py --steps 5 --warmup 2 --cpu-seconds 2 --no-latency > $O/bench_$TAG.json 2> $O/bench_$TAG.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench_$TAG.json"))
    oc = (d.get("chi2_ties") or {}).get("oracle_check", {})
    print("$TAG", d["value"], d["kernel_ms_per_step"], (d.get("chi2_ties") or {}).get("decided_on_exact_float_sums"), "oracle:", oc.get("frames"), oc.get("outside_bar_T_f_w"), oc.get("outside_bar_T_cur_from_ref"), oc.get("different_iteration_counts"))
except Exception as e:
    print("$TAG failed", e)
PY
}
run base A=1
run perlevel PLSVO_ALIGN_PER_LEVEL=1
run s2 PLSVO_ALIGN_SLICES=2
run s2pl PLSVO_ALIGN_SLICES=2 PLSVO_ALIGN_SLICES_PER_LEVEL=1
run s4pl PLSVO_ALIGN_SLICES=4 PLSVO_ALIGN_SLICES_PER_LEVEL=1
run s4 PLSVO_ALIGN_SLICES=4
run s8pl PLSVO_ALIGN_SLICES=8 PLSVO_ALIGN_SLICES_PER_LEVEL=1
