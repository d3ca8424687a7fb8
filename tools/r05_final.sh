#!/bin/bash
# Round-5 evidence call: the default bench line inside a rocprofv3 kernel trace (same run), FETCH_SIZE / WRITE_SIZE PMC passes of the
# alignment launch for configs 2 AND 3 at the one-wave-per-frame shape both benchmarks run (8192 streams: under the profiler the
# synthetic-input kernels of a larger batch cost minutes per pass), configs 3 / 4 / 5 at their default batches, the occupancy probe.
# usage: tools/r05_final.sh <commit>      -> gpurun_out/r05/
COMMIT=${1:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py > $O/bench_stdout.txt 2> $O/bench_default.err
grep '^{"metric"' $O/bench_stdout.txt | tail -1 > $O/bench_default.json
cut -c1-300 $O/bench_default.json
DB=$(find /tmp/kt -name "*results.db" | paste -sd, -)
python $R/tools/rocpd_summary.py "$DB" $O/kernel_trace_stats.csv "python bench.py (default: 32768 streams, 20 steps + 3 warm-up; MI355X); same run as bench_default.json"
head -6 $O/kernel_trace_stats.csv
for CFG in 2 3; do
  CMD="python $R/bench.py --config $CFG --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-latency"
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$C
    PLSVO_ALIGN_THREADS=64 timeout 480 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -- $CMD > $O/pmc_c${CFG}_$C.log 2>&1
    DB=$(find /tmp/pmc_$C -name "*results.db" | paste -sd, -)
    python $R/tools/rocpd_summary.py --counters "$DB" $O/pmc_c${CFG}_$C.csv "PLSVO_ALIGN_THREADS=64 python bench.py --config $CFG --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-latency (MI355X)"
  done
  OUTJ=$O/hbm_traffic.json; [ $CFG = 3 ] && OUTJ=$O/hbm_traffic_config3.json
  python $R/tools/hbm_traffic.py $O/pmc_c${CFG}_FETCH_SIZE.csv $O/pmc_c${CFG}_WRITE_SIZE.csv 8192 $OUTJ "PLSVO_ALIGN_THREADS=64 python bench.py --config $CFG --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-latency" $COMMIT \
       $R/profiles/r03_calib_FETCH_SIZE.csv $R/profiles/r03_calib_WRITE_SIZE.csv $R/profiles/r03_calib_known_bytes.json > $O/hbm_traffic_c$CFG.log 2>&1; tail -c 200 $O/hbm_traffic_c$CFG.log; echo
done
cd $R
timeout 300 python bench.py --config 5 > $O/bench_config5.json 2> $O/bench_config5.err; cut -c1-200 $O/bench_config5.json
timeout 400 python bench.py --config 3 --no-latency > $O/bench_config3.json 2> $O/bench_config3.err; cut -c1-200 $O/bench_config3.json
timeout 300 python bench.py --config 4 > $O/bench_config4.json 2> $O/bench_config4.err; cut -c1-200 $O/bench_config4.json
[ -f $R/pl-svo_amd/libplsvo_hip_probe_w2.so ] && bash $R/tools/r05_w3_probe.sh run
