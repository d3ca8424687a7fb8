#!/bin/bash
# the default bench line inside a rocprofv3 kernel trace (same run)      -> gpurun_out/r05w/
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05w; mkdir -p $O
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py > $O/bench_stdout.txt 2> $O/bench_default.err
grep '^{"metric"' $O/bench_stdout.txt | tail -1 > $O/bench_default.json
cut -c1-300 $O/bench_default.json
DB=$(find /tmp/kt -name "*results.db" | paste -sd, -)
python $R/tools/rocpd_summary.py "$DB" $O/kernel_trace_stats.csv "python bench.py (default: 32768 streams, 20 steps + 3 warm-up; MI355X); same run as bench_default.json"
