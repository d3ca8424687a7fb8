#!/bin/bash
# Round-6 evidence call.  (1) FETCH_SIZE / WRITE_SIZE passes of align_fused_kernel at the DEFAULT batches of configs 2 and 3 (counters
# restricted to the kernel) -> hbm_traffic*.json, copied into profiles/ ON THE BOX so that the bench runs that follow report the traffic
# of this very build (bench.py checks the kernel-source hash); (2) the default bench line inside a rocprofv3 kernel trace (same run);
# (3) configs 3 / 4 / 5 and the one-rank run of the N > 1 code path; (4) the whole -m gpu suite with the full seed sweeps.
# usage: tools/r06_final.sh <commit>      -> gpurun_out/r06/
COMMIT=${1:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
bash $R/tools/r06_pmc.sh $COMMIT "2 3"
cp $O/hbm_traffic.json $R/profiles/hbm_traffic.json
cp $O/hbm_traffic_config3.json $R/profiles/hbm_traffic_config3.json
cd /tmp
rm -rf /tmp/kt
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py > $O/bench_stdout.txt 2> $O/bench_default.err
grep '^{"metric"' $O/bench_stdout.txt | tail -1 > $O/bench_default.json
cut -c1-300 $O/bench_default.json
DB=$(find /tmp/kt -name "*results.db" | paste -sd, -)
python $R/tools/rocpd_summary.py "$DB" $O/kernel_trace_stats.csv "python bench.py (default: 32768 streams, 20 steps + 3 warm-up; MI355X); same run as bench_default.json"
head -8 $O/kernel_trace_stats.csv
cd $R
timeout 600 python bench.py --config 3 --no-latency > $O/bench_config3.json 2> $O/bench_config3.err; cut -c1-200 $O/bench_config3.json
timeout 300 python bench.py --config 4 > $O/bench_config4.json 2> $O/bench_config4.err; cut -c1-200 $O/bench_config4.json
timeout 400 python bench.py --config 5 > $O/bench_config5.json 2> $O/bench_config5.err; cut -c1-200 $O/bench_config5.json
timeout 400 python bench.py --dist-selftest --no-cpu-baseline --no-latency > $O/bench_dist_selftest.json 2> $O/bench_dist_selftest.err; cut -c1-200 $O/bench_dist_selftest.json
( time PLSVO_SWEEP_FULL=1 timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 -p no:cacheprovider ) > $O/gputest_full.log 2>&1
tail -4 $O/gputest_full.log
