# Round-4 last A/B: the final alignment kernel forced to three waves per SIMD (kMinWavesPerSimd = 3 -> libplsvo_hip_w3.so) against the default build,
# 8192 streams at the one-wave-per-frame shape.  Result: profiles/r04j_three_waves_ab.log (2.2x slower: spills).
mkdir -p gpurun_out/r04j
for L in "" _w3 ""; do
  PLSVO_ALIGN_THREADS=64 PLSVO_HIP_LIB=$PWD/pl-svo_amd/libplsvo_hip$L.so timeout 40 python bench.py --config 2 --batch 8192 --steps 5 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith(chr(123)):
        d = json.loads(l); print('lib [$L]', d['value'], d['kernel_ms_per_step'])
"
done 2>&1 | tee gpurun_out/r04j/summary.log
