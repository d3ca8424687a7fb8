"""Per-phase cycle breakdown of align_fused_kernel (needs the instrumented build: make -C pl-svo_amd/csrc timing).
Runs BASELINE config 2 streams one pyramid level at a time and prints s_memtime ticks per phase, per iteration."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PLSVO_HIP_LIB", os.path.join(ROOT, "pl-svo_amd", "libplsvo_hip_timing.so"))
sys.path.insert(0, ROOT)
import numpy as np
P = importlib.import_module("pl-svo_amd")
B = int(os.environ.get("TIMING_BATCH", "256"))
ctx = P.capi.Context(0)
streams = [P.synth.make_align_stream(1234 + i, 640, 480, 200, 80, max_level=3) for i in range(B)]
imgs = P.synth.render_streams(streams, device="cuda")
import torch
torch.cuda.synchronize()   # the library enqueues on its own stream
ctx.config_pyramids(2 * B, 640, 480, 4)
ctx.build_pyramids_dev(0, 2 * B, imgs.data_ptr(), 640, 640 * 480, 0)
ctx.synchronize()
names = ["setup+precompute", "fused pass", "reduce", "rows finish", "solve6", "update", "barrier", "chi2 near-tie test + exact sums"]
L = ctx.L
L.plsvo_align_phase_ticks.restype = C.c_int
L.plsvo_align_phase_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
if os.environ.get("SETUP") == "1":   # a -DPLSVO_TIMING=3 build (tools/build_variant.sh <suffix> "-DPLSVO_TIMING=3" timing): the level set-up split into its steps
    names = ["before the level (kernel start / previous level's tail)", "s_meta reset + barrier", "slot table (thread 0's features)", "barrier after the table",
             "reference patches (thread 0's slots)", "pose matrix + barrier", "iterations after the first", "FIRST iteration of the launch"]
for threads in [os.environ.get("PLSVO_ALIGN_THREADS", "default")]:   # (the library reads the override once, at context creation)
    for level in (3, 2, 1):
        jobs = [P.align_job_from_stream(s, level, level, ref_slot=2 * i, cur_slot=2 * i + 1) for i, s in enumerate(streams)]
        ctx.align_stage(jobs)
        ctx.align_run(); ctx.synchronize()
        ctx.set_profiling(True); ctx.reset_profiling()
        ctx.align_run(); ctx.synchronize()
        ms, n = ctx.kernel_time(P.abi.K_ALIGN_LEVEL)
        ctx.set_profiling(False)
        res = ctx.align_fetch()
        iters = sum(r.iters_per_level[level] for r in res)
        pl, pi = ctx.align_work()
        t = (C.c_uint64 * 8)()
        L.plsvo_align_phase_ticks(ctx.h, t)
        t = np.array(t[:8], dtype=np.float64)
        per_iter = t.copy(); per_iter[0] /= B; per_iter[1:] /= max(iters, 1)
        if os.environ.get("SETUP") == "1":
            per_iter = t / B
        print(f"T={threads} level {level}: kernel {ms:.3f} ms, B={B}, mean iters {iters / B:.2f}, patches/frame {pl / B:.0f}, "
              f"ticks: " + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, per_iter)) +
              f" | per-iteration total {per_iter[1:].sum():.0f} ticks, setup {per_iter[0]:.0f} (per frame)")
