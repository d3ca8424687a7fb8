#!/bin/bash
# Runs on the GPU box (through gpurun): the bench lines of every workload (default = BASELINE configs[1] with the CPU baseline, the
# latency, host-fed and frame-chain legs; configs 3, 4, 5), a rocprofv3 kernel trace of the default bench command (same run as the JSON
# line it is stored with), and the FETCH_SIZE / WRITE_SIZE PMC passes (--kernel-trace only, as the pool requires) on the bench command
# of configs 2 AND 3 and on the calibration kernels (tools/pmc_calib), reduced to hbm_traffic*.json.
# usage: tools/profile_round.sh <tag> <commit>      -> everything lands in gpurun_out/<tag>/
TAG=${1:-r03}
COMMIT=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 600 python bench.py --config 4 > $O/bench_config4.json 2> $O/bench_config4.err; cut -c1-200 $O/bench_config4.json
timeout 600 python bench.py --config 5 > $O/bench_config5.json 2> $O/bench_config5.err; cut -c1-200 $O/bench_config5.json
# default bench line + kernel trace of that same run
cd /tmp
rm -rf /tmp/kt
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py > $O/bench_stdout.txt 2> $O/bench_default.err
grep '^{"metric"' $O/bench_stdout.txt | tail -1 > $O/bench_default.json
cut -c1-300 $O/bench_default.json
DB=$(find /tmp/kt -name "*results.db" | paste -sd, -)
python $R/tools/rocpd_summary.py "$DB" $O/kernel_trace_stats.csv "python bench.py (default: 32768 streams, 20 steps + 3 warm-up; MI355X); same run as bench_default.json"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$C
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/cal_$C -- $R/tools/pmc_calib > $O/calib_$C.log 2>&1
  DB=$(find /tmp/cal_$C -name "*results.db" | paste -sd, -)
  python $R/tools/rocpd_summary.py --counters "$DB" $O/calib_$C.csv "tools/pmc_calib (MI355X)" "%calib_%"
done
grep "^{" $O/calib_FETCH_SIZE.log | tail -1 > $O/calib_known_bytes.json
for CFG in 2 3; do
  B=32768; [ $CFG = 3 ] && B=8192
  CMD="python $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-latency"
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$C
    timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -- $CMD > $O/pmc_c${CFG}_$C.log 2>&1
    DB=$(find /tmp/pmc_$C -name "*results.db" | paste -sd, -)
    python $R/tools/rocpd_summary.py --counters "$DB" $O/pmc_c${CFG}_$C.csv "python bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-latency (MI355X)"
  done
  OUTJ=$O/hbm_traffic.json; [ $CFG = 3 ] && OUTJ=$O/hbm_traffic_config3.json
  python $R/tools/hbm_traffic.py $O/pmc_c${CFG}_FETCH_SIZE.csv $O/pmc_c${CFG}_WRITE_SIZE.csv $B $OUTJ "python bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-latency" $COMMIT \
       $O/calib_FETCH_SIZE.csv $O/calib_WRITE_SIZE.csv $O/calib_known_bytes.json > $O/hbm_traffic_c$CFG.log 2>&1; tail -c 400 $O/hbm_traffic_c$CFG.log; echo
done
# config 3's own line, now that its traffic figure exists next to it
cd $R
mkdir -p profiles; cp $O/hbm_traffic.json profiles/hbm_traffic.json; cp $O/hbm_traffic_config3.json profiles/hbm_traffic_config3.json
timeout 900 python bench.py --config 3 > $O/bench_config3.json 2> $O/bench_config3.err; cut -c1-300 $O/bench_config3.json
head -8 $O/kernel_trace_stats.csv
