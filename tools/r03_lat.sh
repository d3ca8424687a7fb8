#!/bin/bash
# small-batch latency check: quick parity tests of the single-frame path + the latency leg.  usage: tools/r03_lat.sh <tag>
TAG=${1:-r03l}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "matches_oracle or batch_equals_single or adversarial or long_lines or fewer_patches or static" > $O/pytest_subset.log 2>&1; tail -4 $O/pytest_subset.log
timeout 900 python bench.py --batch 4096 --steps 5 --warmup 2 --cpu-seconds 4 > $O/bench_b4096.json 2> $O/bench.err; python - <<PY
import json
d=json.load(open("$O/bench_b4096.json"))
print("value", d["value"], d["kernel_ms_per_step"])
print("latency", {k:(v.get("frames_per_s"), v.get("us_per_step"), v.get("align_kernel_us")) for k,v in d.get("latency",{}).items() if isinstance(v,dict) and "frames_per_s" in v}, {k:d["latency"]["adapter_per_call"].get(k) for k in ("run_us_median","poseopt_us_median")})
print("host_fed", {k:d.get("host_fed",{}).get(k) for k in ("frames_per_s","h2d_GBps","median_rot_err_vs_truth_rad","error")})
PY
tail -2 $O/bench.err
