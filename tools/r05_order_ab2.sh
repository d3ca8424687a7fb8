#!/bin/bash
# A/B on ONE box of the pose optimiser's launch-order refresh (PLSVO_POSEOPT_NO_REORDER=1 switches it off): default bench, config 5 in both
# kernel shapes; then the full GPU suite + smoke() and the default bench line inside a rocprofv3 kernel trace.   -> gpurun_out/r05y/
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05y; mkdir -p $O; cd $R
show() { python -c "import json,sys;d=json.load(open('$1'));print(d['value'],d['ms_per_step'],d.get('kernel_ms_per_step'))"; }
for V in 1 0; do
  echo "== PLSVO_POSEOPT_NO_REORDER=$V default"
  PLSVO_POSEOPT_NO_REORDER=$V timeout 600 python bench.py --no-cpu-baseline --no-latency --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/default_$V.json; show $O/default_$V.json
done
for T in 64 16; do for V in 1 0; do
  echo "== PLSVO_POSEOPT_NO_REORDER=$V config 5, PLSVO_POSEOPT_THREADS=$T"
  PLSVO_POSEOPT_THREADS=$T PLSVO_POSEOPT_NO_REORDER=$V timeout 300 python bench.py --config 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/c5_${T}_$V.json; show $O/c5_${T}_$V.json
done; done
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gpu_suite_full.log 2>&1; grep -E "passed|failed|error" $O/gpu_suite_full.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -E "smoke ok|Error|error" | tee $O/smoke.log
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py > $O/bench_stdout.txt 2> $O/bench_default.err
grep '^{"metric"' $O/bench_stdout.txt | tail -1 > $O/bench_default.json
cut -c1-300 $O/bench_default.json
DB=$(find /tmp/kt -name "*results.db" | paste -sd, -)
python $R/tools/rocpd_summary.py "$DB" $O/kernel_trace_stats.csv "python bench.py (default: 32768 streams, 20 steps + 3 warm-up; MI355X); same run as bench_default.json"
