#!/bin/bash
# Round-4 GPU call 2: robust-weight exhaustive check, the new tie-case tests, bottleneck experiment builds (tools/patches/bottleneck_experiment.patch:
# fixed 8 iterations per level, frozen pose, one memory stream at a time replaced by arithmetic), weight / LDS-image A/B, PMC groups.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04b
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== robust weight, every float in [0,256]"; timeout 120 tools/robust_weight_exhaustive | tee $O/robust_weight_exhaustive.json
echo "== tie cases + sweeps through the default build"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "near_tie or seed_sweep or matches_oracle or batch_equals_single" > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
lib() { if [ "$1" = "." ]; then echo $R/pl-svo_amd/libplsvo_hip.so; else echo $R/pl-svo_amd/libplsvo_hip$1.so; fi; }
bench() {  # lib tag extra...
  L=$1; T=$2; shift 2
  PLSVO_HIP_LIB=$(lib $L) timeout 600 python bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline --no-latency "$@" > $O/bench$L.$T.json 2> $O/bench$L.$T.err
  python - <<PY
import json
try:
    d = json.load(open("$O/bench$L.$T.json"))
    print("lib '$L' $T $*: %.0f frames/s" % d["value"], d["kernel_ms_per_step"], "gn_iters", d.get("chi2_ties", {}).get("gn_iterations_per_step"), "unarmed", d.get("chi2_ties", {}).get("near_ties_without_terms"))
except Exception as e:
    print("lib '$L' $T failed", e)
PY
}
echo "== A/B: default (tie recompute + exact weight), float-only weight, LDS image"
for L in . _w32 _li . _w32 _li; do bench $L ab; done
echo "== bottleneck experiments (8 fixed iterations per level, frozen pose): 1 baseline, +2 no image gather, +4 no cache rows, +8 no chi stores, +16 no xyz, +32 no solve"
for E in 1 3 5 7 9 17 33 63 1; do bench _exp$E exp; done
echo "== counters available"; rocprofv3 -L > $O/rocprofv3_counters.txt 2>&1; grep -c . $O/rocprofv3_counters.txt
cd /tmp
CMD="python $R/bench.py --batch 32768 --steps 2 --warmup 1 --no-cpu-baseline --no-latency"
g=0
for G in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" "TCC_HIT_sum TCC_MISS_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
  g=$((g+1)); rm -rf /tmp/pmcg_$g
  timeout 300 rocprofv3 --kernel-trace --pmc $G -d /tmp/pmcg_$g -- $CMD > $O/pmcg_$g.log 2>&1
  DB=$(find /tmp/pmcg_$g -name "*results.db" | paste -sd, -)
  if [ -n "$DB" ]; then python $R/tools/rocpd_summary.py --counters "$DB" $O/pmcg_$g.csv "bench.py --batch 32768 --steps 2 --warmup 1 ($G)"; grep "align_fused" $O/pmcg_$g.csv | tail -4; else tail -3 $O/pmcg_$g.log; fi
done
