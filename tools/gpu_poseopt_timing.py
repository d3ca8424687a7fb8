"""Per-phase cycle breakdown of pose_opt_kernel (needs the instrumented build: make -C pl-svo_amd/csrc timing)."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PLSVO_HIP_LIB", os.path.join(ROOT, "pl-svo_amd", "libplsvo_hip_timing.so"))
sys.path.insert(0, ROOT)
import numpy as np
P = importlib.import_module("pl-svo_amd")
B = int(os.environ.get("TIMING_BATCH", "8"))
ctx = P.capi.Context(0)
jobs = [P.poseopt_job_from_frame(P.synth.make_poseopt_frame(1234 + i, 200, 80, 640, 480)) for i in range(B)]
ctx.poseopt_stage(jobs)
L = ctx.L
L.plsvo_poseopt_phase_ticks.restype = C.c_int
L.plsvo_poseopt_phase_ticks.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
names = ["init+scale pass", "scale medians", "GN loop", "cov+cull", "final medians"]
for threads in os.environ.get("TIMING_THREADS", "64,256,512").split(","):
    ctx.set_launch_shapes(poseopt_threads=int(threads))
    ctx.poseopt_run(); ctx.synchronize()
    ctx.set_profiling(True); ctx.reset_profiling()
    ctx.poseopt_run(); ctx.synchronize()
    ms, n = ctx.kernel_time(P.abi.K_POSEOPT)
    ctx.set_profiling(False)
    res = ctx.poseopt_fetch()
    iters = sum(r.iters for r in res)
    t = (C.c_uint64 * 8)()
    L.plsvo_poseopt_phase_ticks(ctx.h, t)
    t = np.array(t[:5], dtype=np.float64) / B
    print(f"T={threads}: kernel {ms * 1e3:.1f} us, B={B}, mean GN iters {iters / B:.2f}, ticks per frame: " + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, t)) + f" | total {t.sum():.0f}")
