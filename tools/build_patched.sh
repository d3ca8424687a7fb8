#!/bin/bash
# Builds pl-svo_amd/libplsvo_hip_<suffix>.so from the tree with one of tools/patches/*.patch applied to a scratch copy of csrc
# (the tree itself is not modified), for an A/B through PLSVO_HIP_LIB (tools/ab_latency.sh <tag> "" _<suffix>).
# usage: tools/build_patched.sh tools/patches/slot_parallel_exact_sum_dpp.patch dpp
#        tools/build_patched.sh tools/patches/slot_parallel_exact_sum_dpp.patch next "-DPLSVO_BYTE_CACHE=1 -DPLSVO_LDS_IMG=1 -DPLSVO_TIE_RECOMPUTE=1"
#        (third argument: extra compiler flags for every translation unit -- the candidate default of the next round)
set -e
PATCH=$(realpath $1); SUF=$2; XFLAGS=$3
R=$(cd $(dirname $0)/.. && pwd)
W=$(mktemp -d /tmp/plsvo_patched.XXXX)
mkdir -p $W/pl-svo_amd $W/include
cp -r $R/pl-svo_amd/csrc $W/pl-svo_amd/csrc
cp $R/include/plsvo_hip.h $W/include/
(cd $W && patch -p1 -s < $PATCH)
rm -f $W/pl-svo_amd/csrc/*.o   # (objects copied from the tree were built with the tree's flags)
make -s -C $W/pl-svo_amd/csrc -j8 OUT=$W/lib.so EXTRA="$XFLAGS"
cp $W/lib.so $R/pl-svo_amd/libplsvo_hip_$SUF.so
echo "built pl-svo_amd/libplsvo_hip_$SUF.so"
rm -rf $W
