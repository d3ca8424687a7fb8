#!/bin/bash
# A/B on ONE box of the device-side launch-order refresh (PLSVO_ALIGN_NO_REORDER=1 switches it off): default bench and config 3.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r05_order; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -x -q -k "launch_order or batch_equals_single or full_size" 2>&1 | tail -2
for rep in 1 2; do
for V in 1 0; do
  echo "== PLSVO_ALIGN_NO_REORDER=$V default"
  PLSVO_ALIGN_NO_REORDER=$V timeout 600 python bench.py --no-cpu-baseline --no-latency --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/default_$V.json
  python -c "import json;d=json.load(open('$O/default_$V.json'));print(d['value'],d['ms_per_step'],d.get('kernel_ms'))"
done
done
for V in 1 0; do
  echo "== PLSVO_ALIGN_NO_REORDER=$V config 3"
  PLSVO_ALIGN_NO_REORDER=$V timeout 900 python bench.py --config 3 --no-cpu-baseline --no-latency --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/c3_$V.json
  python -c "import json;d=json.load(open('$O/c3_$V.json'));print(d['value'],d['ms_per_step'],d.get('kernel_ms'))"
done
