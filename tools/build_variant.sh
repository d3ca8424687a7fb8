#!/bin/bash
# Builds pl-svo_amd/libplsvo_hip_<suffix>.so from a scratch copy of csrc with extra compiler flags (A/B builds through PLSVO_HIP_LIB).
# usage: tools/build_variant.sh <suffix> "<flags>" [make target: all | timing]
set -e
SUF=$1; XFLAGS=$2; TGT=${3:-all}
R=$(cd $(dirname $0)/.. && pwd)
W=$(mktemp -d /tmp/plsvo_variant.XXXX)
mkdir -p $W/pl-svo_amd $W/include
cp -r $R/pl-svo_amd/csrc $W/pl-svo_amd/csrc
cp $R/include/plsvo_hip.h $W/include/
rm -f $W/pl-svo_amd/csrc/*.o
make -s -C $W/pl-svo_amd/csrc -j8 EXTRA="$XFLAGS" $TGT
case $TGT in all) cp $W/pl-svo_amd/libplsvo_hip.so $R/pl-svo_amd/libplsvo_hip_$SUF.so ;; timing) cp $W/pl-svo_amd/libplsvo_hip_timing.so $R/pl-svo_amd/libplsvo_hip_$SUF.so ;; esac
echo "built pl-svo_amd/libplsvo_hip_$SUF.so"
rm -rf $W
