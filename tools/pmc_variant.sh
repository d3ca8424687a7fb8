#!/bin/bash
# Memory-side traffic of ONE library build (default or an A/B variant) on the bench command of config 2: the FETCH_SIZE / WRITE_SIZE
# PMC passes (--kernel-trace only, as the pool requires), corrected with the calibration kernels exactly as tools/profile_round.sh does.
# usage: tools/pmc_variant.sh <tag> <lib suffix, e.g. _bc or ""> [commit]      -> gpurun_out/<tag>/hbm_traffic<suffix>.json
TAG=${1:?tag}; SUF=$2; COMMIT=${3:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
export PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip$SUF.so
cd /tmp
if [ ! -f $O/calib_known_bytes.json ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/cal_$C
    timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/cal_$C -- $R/tools/pmc_calib > $O/calib_$C.log 2>&1
    DB=$(find /tmp/cal_$C -name "*results.db" | paste -sd, -)
    python $R/tools/rocpd_summary.py --counters "$DB" $O/calib_$C.csv "tools/pmc_calib (MI355X)" "%calib_%"
  done
  grep "^{" $O/calib_FETCH_SIZE.log | tail -1 > $O/calib_known_bytes.json
fi
CMD="python $R/bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline --no-latency"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -- $CMD > $O/pmc${SUF}_$C.log 2>&1
  DB=$(find /tmp/pmc_$C -name "*results.db" | paste -sd, -)
  python $R/tools/rocpd_summary.py --counters "$DB" $O/pmc${SUF}_$C.csv "PLSVO_HIP_LIB=libplsvo_hip$SUF.so $CMD (MI355X)"
done
python $R/tools/hbm_traffic.py $O/pmc${SUF}_FETCH_SIZE.csv $O/pmc${SUF}_WRITE_SIZE.csv 32768 $O/hbm_traffic$SUF.json "PLSVO_HIP_LIB=libplsvo_hip$SUF.so $CMD" $COMMIT \
     $O/calib_FETCH_SIZE.csv $O/calib_WRITE_SIZE.csv $O/calib_known_bytes.json > $O/hbm_traffic$SUF.log 2>&1
tail -c 400 $O/hbm_traffic$SUF.log; echo
