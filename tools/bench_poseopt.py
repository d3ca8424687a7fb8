"""Pose optimiser alone, 32768 frames resident: ms per launch at 200 + 80 features (the headline step's share; row-per-frame shape) and at
500 + 200 (BASELINE configs[4]; wave-per-frame shape).  usage: python tools/bench_poseopt.py [frames]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P = importlib.import_module("pl-svo_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
ctx = P.capi.Context(0)
for npts, nseg in ((200, 80), (500, 200)):
    base = [P.poseopt_job_from_frame(P.synth.make_poseopt_frame(1234 + i, npts, nseg, 640, 480)) for i in range(256)]
    jobs = [base[i % 256] for i in range(B)]
    ctx.poseopt_stage(jobs)
    for _ in range(3):
        ctx.poseopt_run()
    ctx.synchronize()
    ctx.set_profiling(True); ctx.reset_profiling()
    for _ in range(10):
        ctx.poseopt_run()
    ctx.synchronize()
    ms, n = ctx.kernel_time(P.abi.K_POSEOPT)
    ctx.set_profiling(False)
    print(f"{npts}+{nseg} features, {B} frames: {ms / n:.3f} ms per launch ({B / (ms / n) * 1e3 / 1e6:.2f} M frames/s)", flush=True)
