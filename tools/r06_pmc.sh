#!/bin/bash
# Round-6 counter passes: FETCH_SIZE / WRITE_SIZE of align_fused_kernel at the DEFAULT batch (32768 streams: not scaled) -- counter
# collection restricted to that kernel (--kernel-include-regex), so the torch kernels that render the synthetic inputs run at full
# speed -- reduced to profiles-ready hbm_traffic*.json carrying the kernel-source hash bench.py checks.  Config 3 at its default
# batch (16384) likewise.  usage: tools/r06_pmc.sh <commit> [configs, default "2 3"]      -> gpurun_out/r06/
COMMIT=${1:-unknown}
CFGS=${2:-2 3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for CFG in $CFGS; do
  B=32768; [ $CFG = 3 ] && B=16384
  CMD="python $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-latency"
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$C
    T0=$SECONDS
    timeout 900 rocprofv3 --kernel-trace --pmc $C --kernel-include-regex "align_fused|pose_opt" -d /tmp/pmc_$C -- $CMD > $O/pmc_c${CFG}_$C.log 2>&1
    echo "pmc pass config $CFG $C: $((SECONDS - T0)) s"
    DB=$(find /tmp/pmc_$C -name "*results.db" | paste -sd, -)
    python $R/tools/rocpd_summary.py --counters "$DB" $O/pmc_c${CFG}_$C.csv "--kernel-include-regex align_fused -- python bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-latency (MI355X, $B streams)"
  done
  OUTJ=$O/hbm_traffic.json; [ $CFG = 3 ] && OUTJ=$O/hbm_traffic_config3.json
  python $R/tools/hbm_traffic.py $O/pmc_c${CFG}_FETCH_SIZE.csv $O/pmc_c${CFG}_WRITE_SIZE.csv $B $OUTJ "rocprofv3 --kernel-include-regex align_fused -- python bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-latency" $COMMIT \
       $R/profiles/r03_calib_FETCH_SIZE.csv $R/profiles/r03_calib_WRITE_SIZE.csv $R/profiles/r03_calib_known_bytes.json > $O/hbm_traffic_c$CFG.log 2>&1; tail -c 300 $O/hbm_traffic_c$CFG.log; echo
done
