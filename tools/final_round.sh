#!/bin/bash
# End-of-round GPU call: the full GPU suite + smoke, then the bench lines at HEAD (default inside a rocprofv3 kernel trace; configs 3, 4, 5)
# and config 3's PMC passes (its launch shape changed after profile_round.sh ran; the default workload's 64-thread kernel did not).
# usage: tools/final_round.sh <tag> <commit>
TAG=${1:-r03f}; COMMIT=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
cp gpurun_out/parity_seed_sweep_*.json gpurun_out/poseopt_seed_sweep.json $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
CMD="python $R/bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-latency"
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -- $CMD > $O/pmc_c3_$C.log 2>&1
  DB=$(find /tmp/pmc_$C -name "*results.db" | paste -sd, -)
  python $R/tools/rocpd_summary.py --counters "$DB" $O/pmc_c3_$C.csv "python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-latency (MI355X)"
done
python $R/tools/hbm_traffic.py $O/pmc_c3_FETCH_SIZE.csv $O/pmc_c3_WRITE_SIZE.csv 8192 $O/hbm_traffic_config3.json "python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-latency" $COMMIT \
       $R/profiles/r03_calib_FETCH_SIZE.csv $R/profiles/r03_calib_WRITE_SIZE.csv $R/profiles/r03_calib_known_bytes.json > $O/hbm_traffic_c3.log 2>&1; tail -c 200 $O/hbm_traffic_c3.log; echo
cp $O/hbm_traffic_config3.json $R/profiles/hbm_traffic_config3.json
rm -rf /tmp/kt
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py > $O/bench_stdout.txt 2> $O/bench_default.err
grep '^{"metric"' $O/bench_stdout.txt | tail -1 > $O/bench_default.json; cut -c1-260 $O/bench_default.json
DB=$(find /tmp/kt -name "*results.db" | paste -sd, -)
python $R/tools/rocpd_summary.py "$DB" $O/kernel_trace_stats.csv "python bench.py (default: 32768 streams, 20 steps + 3 warm-up; MI355X); same run as bench_default.json"
cd $R
timeout 600 python bench.py --config 3 > $O/bench_config3.json 2> $O/bench_config3.err; cut -c1-200 $O/bench_config3.json
timeout 300 python bench.py --config 4 > $O/bench_config4.json 2> $O/bench_config4.err; cut -c1-200 $O/bench_config4.json
timeout 300 python bench.py --config 5 > $O/bench_config5.json 2> $O/bench_config5.err; cut -c1-200 $O/bench_config5.json
head -9 $O/kernel_trace_stats.csv | tail -4
