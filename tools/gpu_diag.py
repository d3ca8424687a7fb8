"""First-contact diagnostics on the GPU box: runs every kernel once against the oracle and prints/dumps
what differs.  Not a test; output goes to gpurun_out/diag.json."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
P = importlib.import_module("pl-svo_amd")
from oracle import binding as ob
import helpers as Hh

out = {}
ctx = P.capi.Context(0)
print("device:", ctx.device_info())
ob.build()

def section(name):
    print("\n==== " + name, flush=True)

# ---- half-sampler ----
section("halfsample")
rng = np.random.default_rng(0)
img = rng.integers(0, 256, (480, 640), dtype=np.uint8)
for rounding in (0, 1):
    ctx.config_pyramids(2, 640, 480, 5)
    ctx.build_pyramid(0, img, rounding)
    dev = ctx.download_pyramid(0)
    orc = ob.build_pyramid(img, 5, rounding)
    print("rounding", rounding, [int(np.abs(d.astype(int) - o.astype(int)).max()) for d, o in zip(dev, orc)])

def run_case(tag, seed, W, H, npts, nseg, nlev, maxl, minl, n_iter=30):
    section(f"align {tag}")
    st, ref, cur, job = Hh.make_case(ob, seed, W, H, npts, nseg, nlev, maxl, minl, n_iter)
    res_o, log_o = ob.sparse_align(job, ref, cur, max_log=200)
    ctx.config_pyramids(2, W, H, nlev)
    ctx.upload_pyramid(0, ref); ctx.upload_pyramid(1, cur)
    ctx.align_set_trace(200)
    t0 = time.time()
    res_d = ctx.sparse_align(job)
    dt = time.time() - t0
    log_d = ctx.align_fetch_trace(0)
    print("oracle iters", res_o.iters_per_level[:nlev], "n_meas", res_o.n_meas, "alive", int(res_o.seg_alive.sum()), "chi2", res_o.chi2)
    print("device iters", res_d.iters_per_level[:nlev], "n_meas", res_d.n_meas, "alive", int(res_d.seg_alive.sum()), "chi2", res_d.chi2, "status", res_d.status, "wall %.1f ms" % (dt * 1e3))
    ang, tr, ok = Hh.pose_close(res_d.T, res_o.T)
    print("pose diff rot %.3e rad, trans rel %.3e -> %s" % (ang, tr, "OK" if ok else "FAIL"), "| oracle vs truth", P.synth.se3_log_angle_dist(res_o.T, st.T_true))
    try:
        n, worst = Hh.compare_align_logs(log_o, log_d)
        print("compared", n, "of", len(log_o), "/", len(log_d), "iterations; worst rel diffs", {k: "%.2e" % v for k, v in worst.items()})
    except AssertionError as e:
        print("LOG MISMATCH:", e)
    for a, b in list(zip(log_o, log_d))[:4]:
        print("  L%d it%d acc %d/%d n_meas %d/%d chi2 %.6f/%.6f |x| %.3e/%.3e H00 %.6e/%.6e" % (a["level"], a["iter"], a["accepted"], b["accepted"], a["n_meas"], b["n_meas"], a["new_chi2"], b["new_chi2"], np.abs(a["x"]).max(), np.abs(b["x"]).max(), a["H"][0, 0], b["H"][0, 0]))
    print("work (patch_levels, patch_iters):", ctx.align_work(), "alive equal:", bool(np.array_equal(res_o.seg_alive, res_d.seg_alive)))
    out[tag] = dict(rot=ang, trans=tr, ok=bool(ok), iters_o=res_o.iters_per_level, iters_d=res_d.iters_per_level)
    return st, ref, cur, job

run_case("small 160x120 pts only", 1234, 160, 120, 24, 0, 3, 2, 0)
run_case("small 160x120 pts+lines", 1235, 160, 120, 24, 10, 3, 2, 0)
run_case("config1 640x480 100pts L2-0", 1236, 640, 480, 100, 0, 3, 2, 0)
run_case("config2 640x480 200+80 L3-1", 1237, 640, 480, 200, 80, 4, 3, 1)
run_case("config3 1280x720 400+150 L4-2", 1238, 1280, 720, 400, 150, 5, 4, 2)

# ---- pose optimiser ----
for tag, npts, nseg, nref in (("poseopt 500+200 (9-arg)", 500, 200, -1), ("poseopt 200+80 (10-arg, 5 ref)", 200, 80, 5), ("poseopt points only", 100, 0, -1)):
    section(tag)
    fr = P.synth.make_poseopt_frame(77, npts, nseg)
    job = P.poseopt_job_from_frame(fr, n_iter_ref=nref)
    ro, lo = ob.pose_optimize(job, max_log=40)
    ctx.poseopt_set_trace(40)
    rd = ctx.pose_optimize(job)
    ld = ctx.poseopt_fetch_trace(0)
    ang, tr, ok = Hh.pose_close(rd.T, ro.T)
    print("iters", ro.iters, rd.iters, "ref", ro.iters_ref, rd.iters_ref, "obs", (ro.num_obs_pt, ro.num_obs_ls), (rd.num_obs_pt, rd.num_obs_ls))
    print("pose diff rot %.3e trans rel %.3e %s" % (ang, tr, "OK" if ok else "FAIL"), "scale", ro.estimated_scale, rd.estimated_scale, "err_init", ro.error_init, rd.error_init, "err_final", ro.error_final, rd.error_final)
    print("keep equal:", bool(np.array_equal(ro.pt_keep, rd.pt_keep)), bool(np.array_equal(ro.seg_keep, rd.seg_keep)), "cov rel", Hh.rel(rd.cov, ro.cov), "vs truth", P.synth.se3_log_angle_dist(ro.T, fr.T_true))
    for a, b in list(zip(lo, ld))[:3]:
        print("  ph%d it%d acc %d/%d chi2 %.9e/%.9e A00 %.6e/%.6e |dT| %.3e/%.3e" % (a["phase"], a["iter"], a["accepted"], b["accepted"], a["new_chi2"], b["new_chi2"], a["A"][0, 0], b["A"][0, 0], np.abs(a["dT"]).max(), np.abs(b["dT"]).max()))
    out[tag] = dict(rot=ang, trans=tr, ok=bool(ok))

# ---- quick batch timing (config 2) ----
section("batch timing config 2")
import torch
B = int(os.environ.get("DIAG_BATCH", "256"))
streams = [P.synth.make_align_stream(1234 + i, 640, 480, 200, 80, max_level=3) for i in range(B)]
t0 = time.time(); imgs = P.synth.render_streams(streams, device="cuda"); torch.cuda.synchronize(); print("render %.2fs" % (time.time() - t0))
ctx.config_pyramids(2 * B, 640, 480, 4)
ctx.build_pyramids_dev(0, 2 * B, imgs.data_ptr(), 640, 640 * 480, 0)
ctx.synchronize()
jobs = [P.align_job_from_stream(s, 3, 1, ref_slot=2 * i, cur_slot=2 * i + 1) for i, s in enumerate(streams)]
ctx.align_set_trace(0)
ctx.align_stage(jobs)
ctx.set_profiling(True)
for rep in range(3):
    ctx.reset_profiling()
    t0 = time.time(); ctx.align_run(); ctx.synchronize(); dt = time.time() - t0
    ms, n = ctx.kernel_time(P.abi.K_ALIGN_LEVEL)
    pl, pi = ctx.align_work()
    print("rep %d: wall %.2f ms, level kernels %.3f ms over %d launches, %.0f frames/s, patch_levels %d patch_iters %d -> %.1f GB/s algorithmic" % (rep, dt * 1e3, ms, n, B / dt, pl, pi, (pl * 497 + pi * 485) / (ms * 1e-3) / 1e9))
res = ctx.align_fetch()
errs = [P.synth.se3_log_angle_dist(r.T, s.T_true) for r, s in zip(res, streams)]
print("vs truth: median rot %.2e trans %.2e; mean iters/level" % (np.median([e[0] for e in errs]), np.median([e[1] for e in errs])), np.mean([r.iters_per_level[:4] for r in res], axis=0), "mean alive", np.mean([r.seg_alive.sum() for r in res]))
for T in (256, 512, 1024):
    os.environ["PLSVO_ALIGN_THREADS"] = str(T)
    ctx.reset_profiling(); t0 = time.time(); ctx.align_run(); ctx.synchronize(); dt = time.time() - t0
    ms, n = ctx.kernel_time(P.abi.K_ALIGN_LEVEL)
    print("threads %d: wall %.2f ms level kernels %.3f ms -> %.0f frames/s" % (T, dt * 1e3, ms, B / dt))
del os.environ["PLSVO_ALIGN_THREADS"]
ctx.reset_profiling(); t0 = time.time(); ctx.align_run(); ctx.synchronize(); dt = time.time() - t0
print("no LDS image: wall %.2f ms -> %.0f frames/s" % (dt * 1e3, B / dt))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w"), indent=1, default=float)
print("\nDIAG DONE")
