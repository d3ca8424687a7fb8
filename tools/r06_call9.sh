#!/bin/bash
# Round-6 GPU call 9: pose optimiser -- the wave-per-frame shape with the pose re-read from LDS per feature round (58 -> 14 spilled registers at
# three waves per SIMD, none at two) against the round-5 build; per-phase cycles of the small-batch shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06h
mkdir -p $O
cd $R
for L in po_base po_base2 po_lds3 po_lds2; do
  for ORD in 0 1; do
    echo "== $L PLSVO_POSEOPT_NO_REORDER=$ORD"
    PLSVO_POSEOPT_NO_REORDER=$ORD PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip_$L.so timeout 300 python tools/bench_poseopt.py 2>&1 | grep features
  done
done 2>&1 | tee $O/poseopt_ab.log
PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip_po_lds3.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "pose_optimizer" 2>&1 | tail -1
for B in 1 8; do TIMING_BATCH=$B TIMING_THREADS=256 timeout 200 python tools/gpu_poseopt_timing.py 2>&1 | grep "T=" ; done | tee $O/poseopt_phase_ticks.log
