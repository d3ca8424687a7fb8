#!/bin/bash
# VERDICT r04 item 3, the measured answer: what would a third wave per SIMD buy the one-wave-per-frame alignment kernel if its lane-private
# accumulators were gone?  PROBE builds (wrong results by construction, same memory streams and -- to a few instructions -- the same
# instruction stream): the 27 accumulators of the THROUGHPUT pass collapse onto 6 registers (acc[k & 3], acc[4 + (i & 1)]), which frees
# ~44 VGPRs; the probe is then compiled for two and for three waves per SIMD.  Equal code, different occupancy: the ratio of the two
# launch times is the ceiling of what any accumulators-out-of-the-lane design (LDS [k][lane], per-round reduce-scatter) could gain,
# before its own LDS traffic.  usage: tools/r05_w3_probe.sh build   (here)   |   tools/r05_w3_probe.sh run   (on the GPU box)
R=$(cd $(dirname $0)/.. && pwd)
if [ "$1" = build ]; then
  for W in 2 3; do
    S=$(mktemp -d /tmp/plsvo_probe.XXXX)
    mkdir -p $S/pl-svo_amd $S/include; cp -r $R/pl-svo_amd/csrc $S/pl-svo_amd/csrc; cp $R/include/plsvo_hip.h $S/include/; rm -f $S/pl-svo_amd/csrc/*.o
    F=$S/pl-svo_amd/csrc/align_kernels.hip
    # only the throughput path's expansion (the second of the two textual copies)
    python3 - "$F" "$W" <<'PY'
import sys
p, w = sys.argv[1], sys.argv[2]
s = open(p).read()
a = "for (int jj = i; jj < 6; ++jj) { acc[k] += J[i] * v0[jj] + J[6 + i] * v1[jj]; ++k; }"
b = "for (int i = 0; i < 6; ++i) acc[21 + i] -= jD * J[i] + jE * J[6 + i];"
assert s.count(a) == 2 and s.count(b) == 2
i = s.rindex(a); s = s[:i] + a.replace("acc[k] +=", "acc[k & 3] +=") + s[i + len(a):]
i = s.rindex(b); s = s[:i] + b.replace("acc[21 + i] -=", "acc[4 + (i & 1)] -=") + s[i + len(b):]
s = s.replace("constexpr int kMinWavesPerSimd = 2;", "constexpr int kMinWavesPerSimd = %s;" % w)
open(p, "w").write(s)
PY
    make -s -C $S/pl-svo_amd/csrc -j8 2>&1 | grep -v warning | tail -2
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$S/pl-svo_amd/csrc -S --cuda-device-only -o $S/a.s $F 2>/dev/null
    echo "probe, $W waves per SIMD: $(grep -A12 'align_fused_kernelILi64' $S/a.s | grep -m3 'vgpr_count\|vgpr_spill\|private_segment_fixed' | tr -d '\n')"
    cp $S/pl-svo_amd/libplsvo_hip.so $R/pl-svo_amd/libplsvo_hip_probe_w$W.so
    rm -rf $S
  done
  exit 0
fi
cd $R; O=gpurun_out/r05_w3_probe; mkdir -p $O
for L in _probe_w2 _probe_w3 ""; do
  echo "== lib$L" | tee -a $O/log.txt
  PLSVO_HIP_LIB=$R/pl-svo_amd/libplsvo_hip$L.so PLSVO_ALIGN_THREADS=64 python bench.py --batch ${PROBE_BATCH:-8192} --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('align_fused ms per launch', d['kernel_ms_per_step']['align_fused'], ' frames/s', d['value'])
" | tee -a $O/log.txt
done
