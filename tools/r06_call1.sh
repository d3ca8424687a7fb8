#!/bin/bash
# Round-6 GPU call 1: the -m gpu suite with per-test durations, the default bench line (with the moving-inputs leg), the counter passes
# at the default batch.  usage: tools/r06_call1.sh <commit>
COMMIT=${1:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=60 -p no:cacheprovider ) > $O/gputest.log 2>&1
tail -5 $O/gputest.log
timeout 900 python bench.py > $O/bench_stdout.txt 2> $O/bench_default.err
grep '^{"metric"' $O/bench_stdout.txt | tail -1 > $O/bench_default.json
cut -c1-400 $O/bench_default.json
bash tools/r06_pmc.sh $COMMIT "2"
