#!/bin/bash
# Runs on the GPU box (through gpurun): GPU test suite, smoke, the default bench line, a rocprofv3 kernel trace of the
# bench command and the two PMC passes (FETCH_SIZE, WRITE_SIZE; --kernel-trace only, as the pool requires).
# usage: tools/profile_round.sh <tag>      -> everything lands in gpurun_out/<tag>/
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json
timeout 300 python tools/bench_match.py > $O/match_microbench.json 2> $O/match_microbench.err; cat $O/match_microbench.json
timeout 300 python tools/bench_structopt.py > $O/structopt_microbench.json 2> /dev/null
timeout 300 python tools/bench_seeds.py > $O/seeds_microbench.json 2> $O/seeds_microbench.err; cat $O/seeds_microbench.json
CMD="python $R/bench.py --batch 32768 --steps 3 --warmup 1 --no-cpu-baseline"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- $CMD > $O/kt.log 2>&1
DB=$(find /tmp/kt -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py "$DB" $O/kernel_trace_stats.csv "python bench.py --batch 32768 --steps 3 --warmup 1 --no-cpu-baseline (MI355X)"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -- $CMD > $O/pmc_$C.log 2>&1
  DB=$(find /tmp/pmc_$C -name "*results.db" | head -1)
  python $R/tools/rocpd_summary.py --counters "$DB" $O/pmc_$C.csv "python bench.py --batch 32768 --steps 3 --warmup 1 --no-cpu-baseline (MI355X)"
done
python $R/tools/hbm_traffic.py $O/pmc_FETCH_SIZE.csv $O/pmc_WRITE_SIZE.csv 32768 $O/hbm_traffic.json "python bench.py --batch 32768 --steps 3 --warmup 1 --no-cpu-baseline"
head -8 $O/kernel_trace_stats.csv
