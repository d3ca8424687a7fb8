#!/bin/bash
# Round-4 final GPU call: parity subset on the final build, the default bench line inside a rocprofv3 kernel trace (same run), the
# FETCH_SIZE / WRITE_SIZE PMC passes of the alignment launch (8192 streams at the benchmark's one-wave-per-frame shape: the synthetic-input
# kernels of a 32768-stream run cost seven minutes per pass under the profiler), configs 5, 3, 4.
# usage: tools/r04_final.sh <commit>      -> gpurun_out/r04/
COMMIT=${1:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04
mkdir -p $O
export TMPDIR=/tmp
cd $R
PLSVO_SWEEP_SEEDS=30 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "near_tie or seed_sweep or matches_oracle or batch_equals_single or launch_shape or mixed_batch or adversarial" > $O/pytest_parity.log 2>&1; tail -2 $O/pytest_parity.log
cd /tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $R/bench.py > $O/bench_stdout.txt 2> $O/bench_default.err
grep '^{"metric"' $O/bench_stdout.txt | tail -1 > $O/bench_default.json
cut -c1-400 $O/bench_default.json
DB=$(find /tmp/kt -name "*results.db" | paste -sd, -)
python $R/tools/rocpd_summary.py "$DB" $O/kernel_trace_stats.csv "python bench.py (default: 32768 streams, 20 steps + 3 warm-up; MI355X); same run as bench_default.json"
head -8 $O/kernel_trace_stats.csv
CMD="python $R/bench.py --config 2 --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-latency"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  PLSVO_ALIGN_THREADS=64 timeout 420 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -- $CMD > $O/pmc_c2_$C.log 2>&1
  DB=$(find /tmp/pmc_$C -name "*results.db" | paste -sd, -)
  python $R/tools/rocpd_summary.py --counters "$DB" $O/pmc_c2_$C.csv "PLSVO_ALIGN_THREADS=64 python bench.py --config 2 --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-latency (MI355X)"
done
python $R/tools/hbm_traffic.py $O/pmc_c2_FETCH_SIZE.csv $O/pmc_c2_WRITE_SIZE.csv 8192 $O/hbm_traffic.json "PLSVO_ALIGN_THREADS=64 python bench.py --config 2 --batch 8192 --steps 3 --warmup 1 --no-cpu-baseline --no-latency" $COMMIT \
     $R/profiles/r03_calib_FETCH_SIZE.csv $R/profiles/r03_calib_WRITE_SIZE.csv $R/profiles/r03_calib_known_bytes.json > $O/hbm_traffic_c2.log 2>&1; tail -c 300 $O/hbm_traffic_c2.log; echo
cd $R
timeout 300 python bench.py --config 5 > $O/bench_config5.json 2> $O/bench_config5.err; cut -c1-200 $O/bench_config5.json
timeout 300 python bench.py --config 3 --no-latency > $O/bench_config3.json 2> $O/bench_config3.err; cut -c1-200 $O/bench_config3.json
timeout 300 python bench.py --config 4 > $O/bench_config4.json 2> $O/bench_config4.err; cut -c1-200 $O/bench_config4.json
