#!/bin/bash
# Host emulation build of the WHOLE library: every .hip of pl-svo_amd/csrc/ compiled as C++ for the host against tests/host/emu/
# (device language + runtime on a lock-step wave emulator), linked into <out>/libplsvo_hip_emu.so.  Test infrastructure only: the
# library is never installed next to the product, exports the marker `plsvo_emu_build`, and runs a frame in seconds, not microseconds.
# usage: tests/host/build_emu.sh <out dir> [source dir (default pl-svo_amd/csrc)] [extra compiler flags, e.g. -DPLSVO_BYTE_CACHE=1]
set -e
R=$(cd $(dirname $0)/../.. && pwd)
OUT=${1:?out dir}; SRC=${2:-$R/pl-svo_amd/csrc}; shift; shift || true
CXX=${EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
mkdir -p $OUT
FLAGS="-x c++ -D__HIPCC__ -std=c++17 -O1 -fPIC -ffp-contract=off -Wno-unknown-pragmas -Wno-unknown-attributes -Wno-ignored-attributes -Wno-unused-value -I $R/tests/host/emu -I $SRC $*"
OBJS=""
# EMU_CONTRACT=fast: fma contraction (-ffp-contract=fast-honor-pragmas -mfma: hipcc's own default mode) for the translation units the device Makefile compiles with hipcc's default
# (contraction allowed), off for the three it compiles with -ffp-contract=off -- the host compiler's choice of WHICH products to fuse is
# its own, so this is a probe of how much the results depend on that choice, not a model of the gfx950 code.  Default: off everywhere.
for f in align_kernels poseopt_kernels pyramid_kernels structopt_kernels match_kernels seeds_kernels chain_kernels plsvo_capi; do
  C=""
  case $f in structopt_kernels|match_kernels|seeds_kernels) ;; *) [ "$EMU_CONTRACT" = fast ] && C="-ffp-contract=fast-honor-pragmas -mfma" ;; esac
  $CXX $FLAGS $C -c $SRC/$f.hip -o $OUT/$f.o &
  OBJS="$OBJS $OUT/$f.o"
done
$CXX $FLAGS -c $R/tests/host/emu_runtime.cpp -o $OUT/emu_runtime.o &
wait
$CXX -shared -fPIC -pthread -o $OUT/libplsvo_hip_emu.so $OBJS $OUT/emu_runtime.o -ldl $EMU_LDFLAGS   # (EMU_LDFLAGS: e.g. -fsanitize=address, --coverage)
echo "built $OUT/libplsvo_hip_emu.so"
