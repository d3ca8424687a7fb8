#!/bin/bash
# A/B on ONE box: the pose optimiser's throughput shapes at two (default), three and four waves per SIMD (-DPLSVO_POSEOPT_WAVES=3 / 4:
# VGPR cap 168 / 128; the feature loops stay spill-free at three)         -> gpurun_out/r05v/
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05v; mkdir -p $O; cd $R
show() { python -c "import json,sys;d=json.load(open('$1'));print(d['value'],d['ms_per_step'],d.get('kernel_ms_per_step'))"; }
for L in default pow3 pow4; do
  [ "$L" = "default" ] && LIB=$R/pl-svo_amd/libplsvo_hip.so || LIB=$R/pl-svo_amd/libplsvo_hip_$L.so
  echo "== $L: config 5 (wave per frame), config 5 rows, config 2"
  PLSVO_HIP_LIB=$LIB timeout 200 python bench.py --config 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/c5_$L.json; show $O/c5_$L.json
  PLSVO_HIP_LIB=$LIB PLSVO_POSEOPT_THREADS=16 timeout 200 python bench.py --config 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/c5rows_$L.json; show $O/c5rows_$L.json
  PLSVO_HIP_LIB=$LIB timeout 300 python bench.py --no-cpu-baseline --no-latency --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/c2_$L.json; show $O/c2_$L.json
done
PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip_pow3.so timeout 200 python -m pytest tests -m gpu -x -q -k "poseopt or pose_opt" 2>&1 | grep -E "passed|failed"
