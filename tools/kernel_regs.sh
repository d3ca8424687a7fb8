#!/bin/bash
# Register / scratch / LDS usage of every kernel of a translation unit, from the gfx950 assembly hipcc emits (cross-compiles without a GPU).
# usage: tools/kernel_regs.sh align_kernels [extra hipcc flags ...]        -> one line per kernel; the .s stays in /tmp/plsvo_isa/
U=${1:-align_kernels}; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/plsvo_isa
EXTRA=""
case $U in structopt_kernels|match_kernels|seeds_kernels) EXTRA="-ffp-contract=off";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only $EXTRA "$@" $R/pl-svo_amd/csrc/$U.hip -o /tmp/plsvo_isa/$U.s || exit 1
python3 - /tmp/plsvo_isa/$U.s <<'EOF'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
    blk = m.group(0)
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>3s} sgpr {g('sgpr_count'):>4s} vspill {g('vgpr_spill_count'):>3s} "
          f"sspill {g('sgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
EOF
