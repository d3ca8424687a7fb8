#!/bin/bash
# how small a staged batch gains from the launch-order refresh: both refreshes off (PLSVO_ALIGN_NO_REORDER=PLSVO_POSEOPT_NO_REORDER=1) and on (default:
# alignment when the batch has more frames than resident workgroups, pose optimiser rows above 4 frames per CU)        -> gpurun_out/r05x/
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05x; mkdir -p $O; cd $R
show() { python -c "import json,sys;d=json.load(open('$1'));print(d['value'],d['ms_per_step'],d.get('kernel_ms_per_step'))"; }
for B in ${AB_BATCHES:-600 1500 4096 12000}; do for V in 1 0; do
  echo "== batch $B, NO_REORDER=$V"
  PLSVO_ALIGN_NO_REORDER=$V PLSVO_POSEOPT_NO_REORDER=$V timeout 300 python bench.py --batch $B --no-cpu-baseline --no-latency --steps 40 --warmup 3 2>/dev/null | tail -1 > $O/b${B}_$V.json; show $O/b${B}_$V.json
done; done
timeout 300 python -m pytest tests -m gpu -x -q -k "launch_order" 2>&1 | grep -E "passed|failed"
