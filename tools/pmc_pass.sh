#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counter> [<counter> ...]   -- one rocprofv3 --pmc pass per counter (kernel-trace only)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
CMD="python $R/bench.py --batch ${PMC_BATCH:-4096} --steps 3 --warmup 1 --no-cpu-baseline --no-latency"
for C in "$@"; do
  rm -rf /tmp/pmc_$C
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -- $CMD > $O/pmc_$C.log 2>&1
  DB=$(find /tmp/pmc_$C -name "*results.db" | paste -sd, -)
  if [ -n "$DB" ]; then python $R/tools/rocpd_summary.py --counters "$DB" $O/pmc_$C.csv "bench.py --batch ${PMC_BATCH:-4096} --steps 3 --warmup 1 --no-cpu-baseline --no-latency"; grep align_fused $O/pmc_$C.csv | tail -2; else tail -3 $O/pmc_$C.log; fi
done
