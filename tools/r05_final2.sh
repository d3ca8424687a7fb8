#!/bin/bash
# Round-5 closing evidence after the launch-order refresh: full GPU suite + smoke(), the default bench line inside a rocprofv3 kernel trace
# (same run), config 3 and config 5 at their default batches.  (The PMC passes of tools/r05_final.sh are not repeated: align_fused_kernel<64>
# differs from the profiled build by one 4-byte store per frame.)      usage: tools/r05_final2.sh     -> gpurun_out/r05z/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05z
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee $O/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $O/gpu_suite.log
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py > $O/bench_stdout.txt 2> $O/bench_default.err
grep '^{"metric"' $O/bench_stdout.txt | tail -1 > $O/bench_default.json
cut -c1-300 $O/bench_default.json
DB=$(find /tmp/kt -name "*results.db" | paste -sd, -)
python $R/tools/rocpd_summary.py "$DB" $O/kernel_trace_stats.csv "python bench.py (default: 32768 streams, 20 steps + 3 warm-up; MI355X); same run as bench_default.json"
head -6 $O/kernel_trace_stats.csv
cd $R
timeout 400 python bench.py --config 3 --no-latency > $O/bench_config3.json 2> $O/bench_config3.err; cut -c1-200 $O/bench_config3.json
