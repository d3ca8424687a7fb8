#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace of the widened rows' kernels (match_direct, reproject, structure
# optimisation, seed update, half-sampler) driven by their micro-benchmarks, at the round-1 launch sizes and at chip-filling ones.
# usage: tools/profile_f_kernels.sh <tag>      -> gpurun_out/<tag>/
TAG=${1:-r02f}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
run() {  # name, env assignments..., script
  local name=$1; shift
  rm -rf /tmp/kt_$name
  env "${@:1:$#-1}" timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$name -- python $R/tools/${@: -1} > $O/$name.stdout 2> $O/$name.err
  grep "^{" $O/$name.stdout | tail -1 > $O/$name.json
  DB=$(find /tmp/kt_$name -name "*results.db" | paste -sd, -)
  python $R/tools/rocpd_summary.py --per-kernel "$DB" $O/${name}_kernel_trace_stats.csv "python tools/${@: -1} (${@:1:$#-1}; MI355X)"
  cut -c1-400 $O/$name.json; head -6 $O/${name}_kernel_trace_stats.csv
}
run seeds_64 SEED_SEQS=16 SEED_REPLICATE=4 bench_seeds.py
run seeds_512 SEED_SEQS=16 SEED_REPLICATE=32 bench_seeds.py
run match MATCH_PAIRS=512 bench_match.py
run structopt STRUCT_FRAMES=4096 bench_structopt.py
