#!/bin/bash
# The default bench line and its rocprofv3 kernel trace from ONE run of the SAME command (`python bench.py`).
# usage (on the GPU box): tools/profile_bench_same_run.sh <tag>   -> gpurun_out/<tag>/{bench_default.json,kernel_trace_stats.csv}
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/kt_same
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_same -- python $R/bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt
grep '^{"metric"' $O/bench_stdout.txt | tail -1 > $O/bench_default.json
DB=$(find /tmp/kt_same -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py "$DB" $O/kernel_trace_stats.csv "python bench.py   (MI355X; the bench line of this very run is bench_default.json)"
cat $O/bench_default.json
head -8 $O/kernel_trace_stats.csv
