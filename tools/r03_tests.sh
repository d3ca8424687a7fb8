#!/bin/bash
# round-3 GPU call: full GPU suite (no -x) + smoke + config-4 bench line.  usage: tools/r03_tests.sh <tag>
TAG=${1:-r03t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
cp gpurun_out/parity_seed_sweep_*.json $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python bench.py --config 4 > $O/bench_config4.json 2> $O/bench_config4.err; cut -c1-600 $O/bench_config4.json; tail -3 $O/bench_config4.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<PY
import json
d=json.load(open("$O/bench_default.json"))
print("value", d["value"], d["kernel_ms_per_step"], "frac", d["roofline"]["frac"], "req", d["roofline"]["kernel_requested_GBps"])
print("latency", {k:(v.get("frames_per_s"), v.get("us_per_step")) for k,v in d.get("latency",{}).items() if isinstance(v,dict) and "frames_per_s" in v}, d.get("latency",{}).get("adapter_per_call"))
print("host_fed", d.get("host_fed"))
print("frame_chain", d.get("frame_chain"))
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("cpu_baseline",{}).get("single_thread_value"))
PY
tail -3 $O/bench_default.err
