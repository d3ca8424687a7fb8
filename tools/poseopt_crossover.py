import importlib, os, sys
sys.path.insert(0, os.getcwd())
P = importlib.import_module("pl-svo_amd")
ctx = P.capi.Context(0)
B = 16384
for npts, nseg in ((300, 120), (400, 150), (450, 170), (500, 200)):
    base = [P.poseopt_job_from_frame(P.synth.make_poseopt_frame(1234 + i, npts, nseg, 1280, 720)) for i in range(128)]
    ctx.poseopt_stage([base[i % 128] for i in range(B)])
    for thr in (16, 64):
        ctx.set_launch_shapes(poseopt_threads=thr)
        for _ in range(3): ctx.poseopt_run()
        ctx.synchronize(); ctx.set_profiling(True); ctx.reset_profiling()
        for _ in range(8): ctx.poseopt_run()
        ctx.synchronize(); ms, n = ctx.kernel_time(P.abi.K_POSEOPT); ctx.set_profiling(False)
        print(f"{npts}+{nseg}: threads {thr}: {ms / n:.3f} ms per {B}-frame launch", flush=True)
    ctx.set_launch_shapes(poseopt_threads=0)
# small batches: 256 vs 512 threads per frame
for B in (1, 8):
    jobs = [P.poseopt_job_from_frame(P.synth.make_poseopt_frame(1234 + i, 200, 80, 640, 480)) for i in range(B)]
    ctx.poseopt_stage(jobs)
    for thr in (256, 512):
        ctx.set_launch_shapes(poseopt_threads=thr)
        for _ in range(5): ctx.poseopt_run()
        ctx.synchronize(); ctx.set_profiling(True); ctx.reset_profiling()
        for _ in range(50): ctx.poseopt_run()
        ctx.synchronize(); ms, n = ctx.kernel_time(P.abi.K_POSEOPT); ctx.set_profiling(False)
        print(f"B={B}: threads {thr}: {1e3 * ms / n:.1f} us per launch", flush=True)
