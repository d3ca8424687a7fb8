"""bench.py -- sparse-align + pose-opt frames/sec on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of B independent synthetic streams resident in HBM:
  plsvo::SparseImgAlign::run (3 pyramid levels, <=30 GN iterations each) followed by
  plsvo::pose_optimizer::optimizeGaussNewton (<=10 iterations), per stream, through the C ABI.
Default workload = BASELINE.json configs[1]: 640x480, 200 points + 80 line segments, 4-image pyramid
(alignment levels 3..1).  Per-GPU batch is fixed as N grows ("weak" scaling); streams are independent, so
ranks exchange nothing on the data path -- the only collective is the RCCL all-gather of the per-stream
result records (96-byte plsvo_pose_record: pose + n_tracked, num_obs_pt, num_obs_ls, status), once per step (pl-svo_amd/dist.py::timed_sharded_steps is the timed region).

  python bench.py --gpus 1 --steps K --warmup W            (single process)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline`, `cpu_baseline` and -- default
workload, one GPU -- `latency`: the small-batch operating points (1 / 8 / 64 streams resident, what a live camera or
BASELINE config 4's 8 streams per GPU see) and the C++ drop-in's per-call wall time including the PCIe upload.
Other workloads (extra lines for profiles/, the driver's invocation is the default one):
  --config 3   BASELINE configs[2]: 1280x720, 400 points + 150 segments, 5-image pyramid, levels 4..2
  --config 4   BASELINE configs[3]: 64 streams in eight shards of 8 (one shard per GPU at --gpus 8; on fewer GPUs a rank runs its
               shards back to back), pose all-gather through the C ABI (plsvo_gather_poses) every step
  --config 5   BASELINE configs[4]: pose_optimizer only, 500 points + 200 segments, 10 iterations
"""
import argparse
import math
import importlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_PATCH_LEVEL = 497     # precompute, per patch-level
BYTES_PER_PATCH_ITER = 485      # residual/Jacobian, per patch-iteration
BYTES_PER_POINT_ITER = 24       # pose-opt, per point-iteration
BYTES_PER_SEG_ITER = 40         # pose-opt, per segment-iteration
# what align_fused_kernel itself requests: per patch-iteration the 64-byte record of the reference patch + 24 B 3-D point
# + 5 rows x 2 aligned dwords of the current image (+ 64 B of chi2 terms written per POINT patch-iteration); per
# patch-level 7 rows x 3 dwords of the reference image read, 64 + 24 + 8 B written
OWN_BYTES_PER_PATCH_ITER = 64 + 24 + 40
OWN_BYTES_PER_PATCH_LEVEL = 84 + 64 + 24 + 8
CHI_BYTES_PER_POINT_ITER = 64
# the bytes THIS formulation cannot avoid moving (the bounded figure beside `roofline.frac`): per patch-iteration the 5x5 u8
# window of the current image, the reference patch's 64-byte record (7x7 image bytes + two fractions) and the 3-D point; per point
# patch-iteration also the 16 chi2 terms while armed; per patch-level the 7x7 u8 window of the reference image read, record + point written
MIN_BYTES_PER_PATCH_ITER = 25 + 64 + 24
MIN_BYTES_PER_PATCH_LEVEL = 49 + 64 + 24

METRIC = "sparse-align+pose-opt frames/sec, 640×480, ~200 pts+80 lines; 1/2/4/8 GPU"   # BASELINE.json's metric, verbatim
CONFIGS = {
    2: dict(W=640, H=480, pts=200, seg=80, pyr=4, maxl=3, minl=1, batch=32768, pose_pts=200, pose_seg=80, metric=METRIC,
            workload="BASELINE configs[1]: 640x480, 200 points + 80 line segments, 4-image pyramid (levels 3..1), "
                     "sparse_img_align (<=30 GN it/level) + pose_optimizer (<=10 it, Tukey/MAD)"),
    # (16384 streams: 64 per CU selects the one-wave-per-frame launch shape with the tiled pyramid mirror -- measured 1.10 M frames/s against
    #  0.98 M at 8192 streams / two waves per frame; 80 GB of pyramids + mirror of the 288)
    3: dict(W=1280, H=720, pts=400, seg=150, pyr=5, maxl=4, minl=2, batch=16384, pose_pts=400, pose_seg=150,
            metric="sparse-align+pose-opt frames/sec, 1280×720, 400 pts+150 lines (BASELINE configs[2])",
            workload="BASELINE configs[2]: 1280x720, 400 points + 150 line segments, 5-image pyramid (levels 4..2 = the reference's "
                     "defaults, src/config.cpp:98-99), sparse_img_align (<=30 GN it/level) + pose_optimizer (<=10 it, Tukey/MAD)"),
    4: dict(W=640, H=480, pts=200, seg=80, pyr=4, maxl=3, minl=1, batch=8, shards=8, pose_pts=200, pose_seg=80,
            metric="sparse-align+pose-opt frames/sec, 64 streams of 640×480 (~200 pts+80 lines) in 8 shards of 8 (BASELINE configs[3])",
            workload="BASELINE configs[3]: 64 independent 640x480 streams (200 points + 80 line segments, levels 3..1) in eight shards of 8 streams, "
                     "one shard per GPU on an 8-GPU node (with fewer GPUs a rank runs its shards back to back), pose all-gather per step"),
    5: dict(W=640, H=480, pts=0, seg=0, pyr=0, maxl=0, minl=0, batch=32768, pose_pts=500, pose_seg=200,
            metric="pose-opt frames/sec, 500 pts+200 lines, 10 GN iterations (BASELINE configs[4])",
            workload="BASELINE configs[4]: pose_optimizer::optimizeGaussNewton only, 500 points + 200 line segments, <=10 iterations, "
                     "Tukey weights / MAD scale, outlier cull, covariance, medians"),
}
ORACLE_FLAGS = ("gcc -O3 -march=x86-64-v3 -fno-signed-zeros -fno-math-errno -funroll-loops -ffp-contract=off -fno-tree-slp-vectorize "
                "(the reference's Release flags, CMakeLists.txt:25-36, with -march=native -> x86-64-v3 and fma contraction off: oracle/Makefile)")


def host_cores():
    """threads worth starting for the CPU baseline: the affinity mask, capped by the container's CPU quota (cgroup v2/v1)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(math.ceil(int(q) / int(p)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(math.ceil(q / p))))
        except (OSError, ValueError):
            pass
    return n


KERNEL_SOURCES = ("align_kernels.hip", "plsvo_dev.hpp", "plsvo_math.hpp", "plsvo_wave.hpp", "align_refpatch.hpp", "robust_weight.hpp")


def kernel_source_sha():
    """sha256 (first 16 hex digits) over the sources align_fused_kernel is compiled from.  The PMC passes (tools/hbm_traffic.py) store
    it beside the traffic they measured; `roofline.traffic` is taken from them only while the hash still matches -- i.e. the counters
    describe THIS kernel, whatever else was committed since."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        try:
            h.update(open(os.path.join(ROOT, "pl-svo_amd", "csrc", f), "rb").read())
        except OSError:
            h.update(b"<missing " + f.encode() + b">")
    return h.hexdigest()[:16]


def offline_traffic(B, config=2):
    """HBM-side traffic of the dominant kernel from the committed PMC passes (profiles/hbm_traffic.json for the default workload,
    profiles/hbm_traffic_config3.json for --config 3): measured OFFLINE with rocprofv3 on this same command (counters cannot be
    read from inside a process).  Returns a dict: bytes per launch (None without a usable file), its source, the uncorrected bytes,
    `same_kernel` (the file's source hash equals this tree's) and `same_batch` (measured at this very batch size: not scaled)."""
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json" if config == 2 else f"hbm_traffic_config{config}.json")
    out = {"bytes": None, "source": None, "raw": None, "same_kernel": False, "same_batch": False}
    try:
        d = json.load(open(tfile))
        per_stream = d.get("align_fused_kernel_bytes_per_stream")
        if per_stream is None:
            return out
        out["same_kernel"] = d.get("kernel_source_sha") == kernel_source_sha()
        out["same_batch"] = int(d.get("batch", -1)) == int(B)
        out["source"] = (f"offline rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, kernel-trace only), profiles/{os.path.basename(tfile)}: measured at "
                         f"{d.get('measured_at', 'commit unknown')}, kernel sources {d.get('kernel_source_sha', 'not recorded')} "
                         f"({'= this tree' if out['same_kernel'] else 'NOT this tree: ' + kernel_source_sha()}), "
                         f"{d.get('batch')} streams ({'this batch, not scaled' if out['same_batch'] else f'scaled to {B}'})")
        raw = d.get("align_fused_kernel_bytes_per_stream_uncorrected")
        out["bytes"] = int(round(per_stream * B))
        out["raw"] = int(round(raw * B)) if raw is not None else None
        return out
    except (OSError, ValueError, TypeError):
        return out


# Launch order of the timed steps.  The timed region re-runs ONE staged batch, so a launch order learned from the
# previous launch is learned from bit-identical inputs; whether it survives inputs that CHANGE is what the moving-inputs leg
# (moving_leg below, profiles/r06_launch_order_moving_inputs.log) measures.  The policy here is set from that measurement and is the
# same at every --gpus N; the line reports the figure of the other policy beside `value`.
# Measured (MI355X, 8192 streams x 8 images, profiles/r06_launch_order_moving_inputs.log): with independently drawn motions the refreshed
# order keeps 18 % of the gain the same-image order has (4.13 ms staged, 4.04 refreshed, 3.61 ideal) -- less than half, so the timed steps
# run in the STAGE CALL'S order; the repeat-input figure is reported beside `value` (launch_order.value_other_policy).
LAUNCH_ORDER_REFRESH = False


_DRY = False   # set by main() for the CPU dry run on the emulated library (tests/test_emu_parity.py): a handful of steps instead of hundreds


def latency_leg(P, ctx, align_jobs, pose_jobs, streams, pose_frames, cfg, single_thread_fps):
    """Small-batch operating points on one GPU (inputs resident in HBM, as in the headline number) + the drop-in's per-call time."""
    abi = P.abi
    out = {"note": "b streams resident in HBM; step = plsvo_align_run + plsvo_poseopt_run; frames_per_s / us_per_step: 60 steps enqueued without "
                   "host synchronisation; *_synced: one step + hipStreamSynchronize, per step; mean over the groups (min / max beside it)"}
    def measure(lo, hi):
        ctx.align_stage(align_jobs[lo:hi])
        ctx.poseopt_stage(pose_jobs[lo:hi])
        ctx.synchronize()

        def step():
            ctx.align_run()
            ctx.poseopt_run()
        for _ in range(1 if _DRY else 5):
            step()
        ctx.synchronize()
        K = 2 if _DRY else 60
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        ctx.synchronize()
        bb = (time.perf_counter() - t0) / K
        t0 = time.perf_counter()
        for _ in range(K // 2):
            step()
            ctx.synchronize()
        sy = (time.perf_counter() - t0) / (K // 2)
        ctx.set_profiling(True)
        ctx.reset_profiling()
        for _ in range(2 if _DRY else 20):
            step()
        ctx.synchronize()
        ctx.set_profiling(False)
        ka, na = ctx.kernel_time(abi.K_ALIGN_LEVEL)
        kp, npo = ctx.kernel_time(abi.K_POSEOPT)
        return bb, sy, 1e3 * ka / max(na, 1), 1e3 * kp / max(npo, 1)

    for b in (1, 8, 64):
        # a launch lasts as long as its slowest frame (3 levels x up to 30 GN iterations), so a small batch depends on which
        # streams are in it: every size is measured on up to 8 disjoint groups of the timed batch's first streams
        groups = [(g * b, (g + 1) * b) for g in range(8) if (g + 1) * b <= min(len(align_jobs), 64 if b < 64 else 512)]
        if not groups:
            continue
        if _DRY:
            groups = groups[:1]
        rows = [measure(lo, hi) for lo, hi in groups]
        bb = np.array([r[0] for r in rows]); sy = np.array([r[1] for r in rows])
        out[f"B{b}"] = {"groups": len(rows), "frames_per_s": round(float(np.mean(b / bb)), 1), "frames_per_s_min": round(float(np.min(b / bb)), 1),
                        "frames_per_s_max": round(float(np.max(b / bb)), 1), "us_per_step": round(float(np.mean(bb)) * 1e6, 1),
                        "frames_per_s_synced": round(float(np.mean(b / sy)), 1), "us_per_step_synced": round(float(np.mean(sy)) * 1e6, 1),
                        "align_kernel_us": round(float(np.mean([r[2] for r in rows])), 1), "poseopt_kernel_us": round(float(np.mean([r[3] for r in rows])), 1)}
    # the C++ drop-in (pl-svo_amd/host/plsvo/hip_adapter.hpp) called like FrameHandlerMono::processFrame does: every call
    # flattens the feature lists, uploads the NEW frame's pyramid over PCIe (the previous frame's is cached on the device),
    # launches, synchronises and writes the results back
    drv = os.path.join(ROOT, "pl-svo_amd", "host", "adapter_driver")
    if os.path.exists(drv):
        try:
            ref = ctx.download_pyramid(0)
            cur = ctx.download_pyramid(1)
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "in.bin")
                P.adapter_io.write_adapter_input(path, streams[0], ref, cur, pose_frames[0], cfg["pyr"], cfg["maxl"], cfg["minl"])
                r = subprocess.run([drv, "--bench", "200", path], check=True, capture_output=True, text=True, timeout=300)
            a = json.loads(r.stdout.strip().splitlines()[-1])
            a["what"] = ("adapter_driver --bench 200: wall time per SparseImgAlign::run / optimizeGaussNewton call of the C++ drop-in, "
                         "feature flattening + PCIe upload of one 408 KB pyramid + launch + synchronise + write-back included")
            if single_thread_fps:
                a["cpu_single_thread_us_per_frame"] = round(1e6 / single_thread_fps, 1)
            out["adapter_per_call"] = a
        except Exception as e:   # the latency leg must never take the headline line down
            out["adapter_per_call"] = {"error": str(e)[:200]}
    return out


def host_fed_leg(P, torch, dev, stream, streams, cfg, n_streams=1024, steps=12):
    """Host-fed throughput (never `value`): every step, every stream receives a NEW level-0 image from pinned host memory.
    Double-buffered: while step t's kernels run, step t+1's images cross PCIe on a second stream.  Per step on the compute stream:
    wait for the upload; cur -> ref on the device (plsvo_hip_copy_slots); device half-sampler builds the new cur pyramids
    (plsvo_hip_build_pyramids_dev); plsvo_align_run; plsvo_poseopt_run.  One 640x480 frame = 307 200 B over PCIe."""
    capi, synth = P.capi, P.synth
    n = min(n_streams, len(streams))
    W, H = cfg["W"], cfg["H"]
    sub = streams[:n]
    ctx = capi.Context(dev.index or 0, stream=stream.cuda_stream)
    try:
        ctx.config_pyramids(2 * n, W, H, cfg["pyr"])               # ref frames in slots [0, n), cur frames in [n, 2n)
        host = []
        for c0 in range(0, n, 256):
            imgs = synth.render_streams(sub[c0:c0 + 256], device=dev)          # [b, 2, H, W] u8
            ref0 = imgs[:, 0].contiguous()
            ctx.build_pyramids_dev(c0, ref0.shape[0], ref0.data_ptr(), W, W * H, 0)
            ctx.synchronize()
            host.append(imgs.cpu())
            del imgs, ref0
        both = torch.cat(host)
        # the frames the "camera" delivers: each stream's two images alternately (B, A, B, A, ..), so that every step aligns a
        # genuinely different image pair (forward motion on even steps, the reverse motion on odd ones)
        host_img = [both[:, 1].contiguous().pin_memory(), both[:, 0].contiguous().pin_memory()]
        del both
        jobs = [P.align_job_from_stream(s_, cfg["maxl"], cfg["minl"], ref_slot=i, cur_slot=n + i) for i, s_ in enumerate(sub)]
        frames = [synth.make_poseopt_frame(1234 + i, cfg["pose_pts"], cfg["pose_seg"], W, H) for i in range(n)]
        pjobs = [P.poseopt_job_from_frame(f) for f in frames]
        ctx.align_stage(jobs)
        ctx.poseopt_stage(pjobs)
        staging = [torch.empty((n, H, W), dtype=torch.uint8, device=dev) for _ in range(2)]
        copy_stream = torch.cuda.Stream(dev)
        uploaded = [torch.cuda.Event() for _ in range(2)]
        consumed = [torch.cuda.Event() for _ in range(2)]

        def upload(k):                                             # k = step parity = which of the two images is due
            with torch.cuda.stream(copy_stream):
                copy_stream.wait_event(consumed[k])                # the staging buffer is free again
                staging[k].copy_(host_img[k], non_blocking=True)
                uploaded[k].record(copy_stream)

        def step(t):
            k = t & 1
            stream.wait_event(uploaded[k])
            if t > 0:
                ctx.copy_slots(0, n, n)                            # the frame just tracked becomes the reference
            ctx.build_pyramids_dev(n, n, staging[k].data_ptr(), W, W * H, 0)
            consumed[k].record(stream)
            ctx.align_run()
            ctx.poseopt_run()

        for k in range(2):
            consumed[k].record(stream)
        upload(0)
        for t in range(3):                                         # warm-up
            upload((t + 1) & 1)
            step(t)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for t in range(3, 3 + steps):                              # `steps` even: the last step (t = steps + 2) is a forward (A -> B) one
            upload((t + 1) & 1)
            step(t)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        res = ctx.align_fetch()
        errs = np.array([synth.se3_log_angle_dist(r.T, s_.T_true) for r, s_ in zip(res[:32], sub[:32])])
        return {"frames_per_s": round(n * steps / dt, 1), "streams": n, "steps": steps, "ms_per_step": round(1e3 * dt / steps, 3),
                "pcie_bytes_per_frame": W * H, "h2d_GBps": round(n * steps * W * H / dt / 1e9, 2),
                "median_rot_err_vs_truth_rad": float(np.median(errs[:, 0])),
                "what": "host-fed pipeline, NOT the headline value: every step uploads one new level-0 image per stream from pinned host memory "
                        "(second stream, double-buffered), copies cur->ref on the device, builds the new pyramids with the device half-sampler, then "
                        "plsvo_align_run + plsvo_poseopt_run; steady state over the timed steps"}
    finally:
        ctx.close()


def moving_leg(P, torch, dev, stream, streams, cfg, n_streams=8192, n_images=8, models=("independent", "smooth")):
    """Does a launch order learned from the PREVIOUS launch survive inputs that change?  (Never `value`.)

    The timed steps of this benchmark re-run one staged batch, so align_reorder_kernel sorts every launch by the work the previous
    launch measured on bit-identical inputs.  Here every launch sees a NEW current image: n streams, each with its reference frame and
    n_images current images (same scene, same features, same initial pose) rendered at
      "independent"  independently drawn motions (synth.stream_motion): the harshest case -- only what the scene and the feature
                     set decide about a frame's cost carries over from one image to the next; THIS model decides the policy;
      "smooth"       motions within ~15 % of one per-stream motion: a camera on a smooth path.
    Before every launch the next image's pyramid is copied into the streams' current-frame slots on the device
    (plsvo_hip_copy_slots), then plsvo_align_run; the launch shape is the headline's (one wave per frame).  Three launch orders,
    hipEvent time of the alignment launch per image:
      staged    PLSVO_OPT_ALIGN_REORDER = 0: the stage call's order (most patches first) for every launch
      refresh   the library's default: longest-first by the work of the previous launch -- which ran on a DIFFERENT image
      ideal     the same image launched a second time: the order comes from its own work (what a repeat-input benchmark step sees)
    gain_kept = (staged - refresh) / (staged - ideal)."""
    capi, synth, abi = P.capi, P.synth, P.abi
    n, K = min(n_streams, len(streams)), n_images
    W, H = cfg["W"], cfg["H"]
    sub = streams[:n]
    ctx = capi.Context(dev.index or 0, stream=stream.cuda_stream)
    try:
        ctx.set_launch_shapes(align_threads=64)
        ctx.config_pyramids((2 + K) * n, W, H, cfg["pyr"])      # ref frames [0, n), working current frames [n, 2n), image k at [(2 + k) n, (3 + k) n)
        chunk = 256
        for c0 in range(0, n, chunk):
            part = sub[c0:c0 + chunk]
            img = synth.render_views(part, [None] * len(part), device=dev)
            ctx.build_pyramids_dev(c0, len(part), img.data_ptr(), W, W * H, 0)
            ctx.synchronize()
            del img
        jobs = [P.align_job_from_stream(s_, cfg["maxl"], cfg["minl"], ref_slot=i, cur_slot=n + i) for i, s_ in enumerate(sub)]
        ctx.align_stage(jobs)

        def timed_run():
            ctx.reset_profiling()
            ctx.align_run()
            ctx.synchronize()
            ms, nl = ctx.kernel_time(abi.K_ALIGN_LEVEL)
            return ms / max(nl, 1)

        def sweep(refresh, twice):
            ctx.set_launch_order_refresh(align=refresh)
            # the launch before the sweep's first one ran on the LAST image: image 0 never meets an order learned on itself
            ctx.copy_slots(n, (2 + K - 1) * n, n)
            ctx.align_run()
            first, second = [], []
            for k in range(K):
                ctx.copy_slots(n, (2 + k) * n, n)
                first.append(timed_run())
                if twice:
                    second.append(timed_run())
            return first, second

        out = {"streams": n, "images_per_stream": K, "launch_shape_threads": 64,
               "what": "every launch aligns a NEW current image (same features and initial pose): hipEvent time of the alignment launch in the stage call's "
                       "order, in the order refreshed from the previous launch (another image), and re-launched on the same image (order from its own "
                       "work); `independent`: motions drawn independently per image (decides the policy), `smooth`: within ~15 % of one motion per stream; "
                       "NOT the headline value"}
        for model in models:
            motions = [[synth.stream_motion(s_, k, model=model) for s_ in sub] for k in range(K)]
            for c0 in range(0, n, chunk):
                part = sub[c0:c0 + chunk]
                for k in range(K):
                    img = synth.render_views(part, motions[k][c0:c0 + chunk], device=dev, noise_tag=k + 1)
                    ctx.build_pyramids_dev((2 + k) * n + c0, len(part), img.data_ptr(), W, W * H, 0)
                    ctx.synchronize()
                    del img
            ctx.set_profiling(True)
            sweep(True, False)                      # warm-up of everything (code, order buffers)
            staged, _ = sweep(False, False)
            refresh, ideal = sweep(True, True)
            ctx.set_profiling(False)
            res = ctx.align_fetch()                 # the last launch ran on image K - 1
            errs = np.array([synth.se3_log_angle_dist(r.T, T) for r, T in zip(res[:64], motions[K - 1][:64])])
            ms_s, ms_r, ms_i = float(np.mean(staged)), float(np.mean(refresh)), float(np.mean(ideal))
            kept = (ms_s - ms_r) / (ms_s - ms_i) if ms_s - ms_i > 1e-9 else None
            out[model] = {"align_launch_ms": {"staged_order": round(ms_s, 4), "refresh_from_previous_image": round(ms_r, 4), "ideal_same_image": round(ms_i, 4)},
                          "align_launch_ms_per_image": {"staged_order": [round(x, 4) for x in staged], "refresh_from_previous_image": [round(x, 4) for x in refresh],
                                                        "ideal_same_image": [round(x, 4) for x in ideal]},
                          "gain_ideal_pct": round(100.0 * (ms_s - ms_i) / ms_s, 2), "gain_refresh_pct": round(100.0 * (ms_s - ms_r) / ms_s, 2),
                          "gain_kept": round(kept, 3) if kept is not None else None,
                          "median_rot_err_vs_truth_rad": float(np.median(errs[:, 0]))}
        return out
    finally:
        ctx.close()


def chain_leg(P, torch, dev, stream, streams, cfg, n_streams=4096, steps=10, cpu_frames=6):
    """Resident frame step (never `value`): alignment -> pose composition -> reprojection of the stream's landmarks -> direct matching ->
    selection -> pose optimisation as ONE enqueue per step for n_streams streams (plsvo_chain_run), everything staying in HBM; beside
    it the same chain on the CPU oracle, one stream at a time on one thread (Python between the four calls included)."""
    capi, synth, abi = P.capi, P.synth, P.abi
    n = min(n_streams, len(streams))
    W, H = cfg["W"], cfg["H"]
    sub = streams[:n]
    ctx = capi.Context(dev.index or 0, stream=stream.cuda_stream)
    try:
        ctx.config_pyramids(2 * n, W, H, cfg["pyr"])
        for c0 in range(0, n, 256):
            imgs = synth.render_streams(sub[c0:c0 + 256], device=dev)
            ctx.build_pyramids_dev(2 * c0, 2 * imgs.shape[0], imgs.data_ptr(), W, W * H, 0)
            ctx.synchronize()
            del imgs

        def chain_job(i, s_):
            aj = P.align_job_from_stream(s_, cfg["maxl"], cfg["minl"], ref_slot=2 * i, cur_slot=2 * i + 1)
            pos = np.concatenate([s_.pt_pos_w, s_.seg_spos_w, s_.seg_epos_w])
            return abi.ChainJob(aj, s_.T_ref_w, s_.T_ref_w, 2 * i, len(s_.pt_pos_w), len(s_.seg_spos_w), pos,
                                np.concatenate([s_.pt_px, s_.seg_spx, s_.seg_epx]), np.concatenate([s_.pt_f, s_.seg_sf, s_.seg_ef]))
        jobs = [chain_job(i, s_) for i, s_ in enumerate(sub)]
        ctx.chain_stage(jobs, sub[0].cam, n_pyr_levels=cfg["pyr"] - 1)
        for _ in range(2):
            ctx.chain_run()
        ctx.synchronize()
        ctx.set_profiling(True)             # hipEvent pairs on the launch stream around the three kernel families of a step
        ctx.reset_profiling()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.chain_run()
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / steps
        fam_ms = {name: ctx.kernel_time(k)[0] / steps for name, k in (("align_fused", abi.K_ALIGN_LEVEL), ("reproject_match_select", abi.K_MATCH), ("pose_opt", abi.K_POSEOPT))}
        ctx.set_profiling(False)
        res = ctx.chain_fetch()
        errs = np.array([synth.se3_log_angle_dist(r.pose.T, synth.se3_mul(s_.T_true, s_.T_ref_w)) for r, s_ in zip(res[:32], sub[:32])])
        out = {"frames_per_s": round(n / dt, 1), "streams": n, "ms_per_step": round(1e3 * dt, 3),
               "candidates_per_frame": jobs[0].n_cand, "matched_points_median": int(np.median([len(r.sel_pt) for r in res[:64]])),
               "median_rot_err_vs_truth_rad": float(np.median(errs[:, 0])),
               "kernel_ms_per_step": {k_: round(v_, 4) for k_, v_ in fam_ms.items()},
               "what": "resident frame step, NOT the headline value: plsvo_chain_run = alignment, pose composition, reprojection of the stream's "
                       "landmarks, direct matching, selection, pose optimisation, one enqueue per step, nothing leaves HBM"}
        try:   # the same chain on the CPU oracle (checker only; one thread, Python between the calls)
            from oracle import binding as ob
            ob.build()
            t_cpu, ang_max, passes, matched = 0.0, 0.0, 0, 0
            for i in range(min(cpu_frames, n)):
                s_, cj = sub[i], jobs[i]
                ref, cur = ctx.download_pyramid(2 * i), ctx.download_pyramid(2 * i + 1)
                t1 = time.perf_counter()
                ar, _ = ob.sparse_align(cj.align_job, ref, cur)
                T_k = synth.se3_mul(ar.T, s_.T_ref_w)
                rp = ob.reproject(abi.ReprojectJob(s_.cam, np.stack([s_.T_ref_w, T_k]), np.ones(cj.n_cand, np.int32), cj.pos, cell_size=30))
                vis = rp["cell"] >= 0
                npt, nsg = cj.n_cand_pt, cj.n_cand_seg
                sv = vis[npt:npt + nsg] & vis[npt + nsg:]
                vis[npt:npt + nsg] = sv; vis[npt + nsg:] = sv
                idx = np.nonzero(vis)[0]
                m = len(idx)
                mr = ob.match_direct(abi.MatchJob(s_.cam, np.stack([s_.T_ref_w, T_k]), np.array([0, 1], np.int32), np.ones(m, np.int32), np.zeros(m, np.int32),
                                                  cj.ref_px[idx], cj.ref_f[idx], np.zeros(m, np.int32), np.zeros(m, np.uint8), np.zeros((m, 2)), cj.pos[idx],
                                                  rp["px"][idx], cfg["pyr"] - 1, 10), [ref, cur])
                passes += int(np.sum(mr["n_iter"])); matched += m
                found = np.zeros(cj.n_cand, bool); found[idx] = mr["found"].astype(bool)
                px_new = rp["px"].copy(); px_new[idx] = mr["px_cur"]
                level = np.zeros(cj.n_cand, np.int32); level[idx] = np.maximum(mr["search_level"], 0)
                pt_i = np.nonzero(found[:npt])[0]
                seg_i = np.nonzero(found[npt:npt + nsg] & found[npt + nsg:])[0]
                brg = lambda px: (lambda r: r / np.linalg.norm(r, axis=1, keepdims=True))(np.stack([(px[:, 0] - s_.cam[2]) / s_.cam[0], (px[:, 1] - s_.cam[3]) / s_.cam[1], np.ones(len(px))], axis=1))
                sf, ef = brg(px_new[npt + seg_i]), brg(px_new[npt + nsg + seg_i])
                line = np.cross(sf, ef)
                line = line / np.sqrt(line[:, 0:1] ** 2 + line[:, 1:2] ** 2) if len(seg_i) else np.zeros((0, 3))
                po, _ = ob.pose_optimize(abi.PoseOptJob(T_k, abs(s_.cam[0]), 2.0, 10, brg(px_new[pt_i]), cj.pos[pt_i], level[pt_i], line,
                                                        cj.pos[npt + seg_i], cj.pos[npt + nsg + seg_i], level[npt + seg_i]))
                t_cpu += time.perf_counter() - t1
                ang_max = max(ang_max, synth.se3_log_angle_dist(po.T, res[i].pose.T)[0])
            k = min(cpu_frames, n)
            out["cpu_oracle_chain"] = {"frames_per_s": round(k / t_cpu, 2), "frames": k, "threads": 1,
                                       "max_rot_diff_vs_device_rad": ang_max,
                                       "note": "oracle/libplsvo_oracle.so through its Python binding, one stream at a time"}
            out["speedup_vs_cpu_oracle_chain_1thread"] = round(out["frames_per_s"] / (k / t_cpu), 1)
            # Accounting of the step's dominant kernel family (match_direct_kernel + the three glue kernels share the hipEvent pair): per
            # matched candidate the reference's findMatchDirect touches, algorithmically, the 11x11 source region of the 10x10 affine warp
            # (121 B of the keyframe level; warpAffine itself issues 4 byte reads per pixel), the 9x9 window of the new frame once per
            # alignment pass (81 B), 85 B of candidate record and 21 B of result; and executes ~20 float operations per warped pixel and
            # ~17 per pixel of an 8x8 alignment pass.  Passes per candidate come from the CPU oracle on the first frames (the device
            # executes the same passes: its results are bit-identical).  The kernel is bound by instruction issue, not by HBM.
            mean_passes = passes / max(matched, 1)
            cands = n * jobs[0].n_cand
            b_cand, f_cand = 121 + 81 * mean_passes + 85 + 21, 100 * 20 + 64 * 17 * mean_passes
            t_m = fam_ms["reproject_match_select"] * 1e-3
            if t_m > 0:
                out["match_direct_roofline"] = {"bound": "hbm", "achieved": round(cands * b_cand / t_m / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                                "frac": round(cands * b_cand / t_m / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                                                "candidates_per_step": int(cands), "mean_alignment_passes": round(mean_passes, 2),
                                                "algorithmic_bytes_per_candidate": round(b_cand, 1), "float_ops_per_candidate": round(f_cand, 1),
                                                "achieved_GFLOPs_f32": round(cands * f_cand / t_m / 1e9, 1),
                                                "note": "reproject + active + match_direct + select kernels under one hipEvent pair (match_direct_kernel is >95 % of it); "
                                                        "one lane per candidate, sequential float sums as in the reference (bit-exact): issue-bound"}
        except Exception as e:
            out["cpu_oracle_chain"] = {"error": str(e)[:200]}
        return out
    finally:
        ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configs[] number (1-based): 2 = the metric's workload")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("PLSVO_BENCH_BATCH", "0")), help="streams per GPU (0 = the config's default)")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="CPU-baseline budget (rank 0, N=1 only): half single-thread, half all cores")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the small-batch / per-call latency leg")
    ap.add_argument("--dist-selftest", action="store_true", help="run the N>1 code path (process group, RCCL communicator, pose copy + "
                    "plsvo_gather_poses every step) with a single rank: what a 1-GPU box can exercise of it")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    pose_only = args.config == 5

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # DRY RUN (tests/test_emu_parity.py only): PLSVO_BENCH_DRY_RUN=1 together with a host emulation build of the library runs this very
    # script on the CPU -- same staging, same calls, same JSON assembly, tiny batch -- so that a mistake in it shows up in the CPU suite
    # and not at the end of a round.  Its line says what it is; without the emulated library the variable has no effect.
    dry = os.environ.get("PLSVO_BENCH_DRY_RUN") == "1" and hasattr(importlib.import_module("pl-svo_amd").capi.lib(), "plsvo_emu_build")
    if dry:
        global _DRY
        _DRY = True
        import contextlib
        import types
        torch.cuda.set_device = lambda *_a, **_k: None
        torch.cuda.synchronize = lambda *_a, **_k: None
        _noop = lambda *_a, **_k: None
        torch.cuda.Stream = lambda *_a, **_k: types.SimpleNamespace(cuda_stream=None, wait_event=_noop)
        torch.cuda.Event = lambda *_a, **_k: types.SimpleNamespace(record=_noop)
        torch.cuda.stream = lambda *_a, **_k: contextlib.nullcontext()
        torch.Tensor.pin_memory = lambda self, *_a, **_k: self
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.dist_selftest
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if dry:   # the CPU dry run of the N > 1 path: gloo ranks, torch.distributed's all-gather in place of the C ABI's RCCL one
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    dev = torch.device("cpu") if dry else torch.device("cuda", local_rank)

    P = importlib.import_module("pl-svo_amd")
    capi, synth, abi, D = P.capi, P.synth, P.abi, P.dist
    if hasattr(capi.lib(), "plsvo_emu_build") and not dry:   # tests/host/build_emu.sh: the device sources on a CPU wave emulator -- a test vehicle, never a measurement
        raise SystemExit("bench.py: PLSVO_HIP_LIB names a host emulation build of the library; the benchmark runs the gfx950 library only")
    B = args.batch if args.batch > 0 else cfg["batch"]
    W, H = cfg["W"], cfg["H"]
    # shards: blocks of B streams with their own context.  Normally one per rank; BASELINE configs[3] fixes eight of them, so with
    # fewer than eight ranks a rank owns several and runs them back to back.
    shards_total = cfg.get("shards", world)
    if shards_total % world:
        raise SystemExit(f"--config {args.config} has {shards_total} shards: --gpus must divide it")
    local_shards = shards_total // world

    # ONE stream for everything: the library enqueues on it (plsvo_hip_create_on_stream), it is torch's current stream while the
    # benchmark runs, so torch.cuda.synchronize and the RCCL all-gather (plsvo_gather_poses, same stream) are ordered after the
    # library's kernels and copies
    order_refresh = {"staged": False, "refresh": True}.get(os.environ.get("PLSVO_BENCH_LAUNCH_ORDER", ""), LAUNCH_ORDER_REFRESH)
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        shard = []      # per local shard: dict(ctx, streams, align_jobs, pose_frames, pose_jobs)
        for v in range(local_shards):
            c = capi.Context(local_rank, stream=stream.cuda_stream)
            c.set_launch_order_refresh(align=order_refresh, poseopt=order_refresh)
            seeds = D.rank_seeds(rank * local_shards + v, shards_total, B)          # 1234 + global stream index
            streams, align_jobs = [], []
            if not pose_only:
                streams = [synth.make_align_stream(s_, W, H, cfg["pts"], cfg["seg"], max_level=cfg["maxl"]) for s_ in seeds]
                c.config_pyramids(2 * B, W, H, cfg["pyr"])
                chunk = 256 if W <= 640 else 64
                for c0 in range(0, B, chunk):
                    sub = streams[c0:c0 + chunk]
                    imgs = synth.render_streams(sub, device=dev)                      # [b, 2, H, W] u8 in HBM
                    c.build_pyramids_dev(2 * c0, 2 * len(sub), imgs.data_ptr(), W, W * H, 0)   # device half-sampler
                    c.synchronize()
                    del imgs
                align_jobs = [P.align_job_from_stream(s_, cfg["maxl"], cfg["minl"], ref_slot=2 * i, cur_slot=2 * i + 1) for i, s_ in enumerate(streams)]
                c.align_stage(align_jobs)      # features + job descriptors -> HBM; the timed region only launches kernels
            pose_frames = [synth.make_poseopt_frame(s_, cfg["pose_pts"], cfg["pose_seg"], W, H) for s_ in seeds]
            pose_jobs = [P.poseopt_job_from_frame(f) for f in pose_frames]
            c.poseopt_stage(pose_jobs)
            c.synchronize()
            shard.append(dict(ctx=c, streams=streams, align_jobs=align_jobs, pose_frames=pose_frames, pose_jobs=pose_jobs))
        ctx, streams, align_jobs, pose_frames, pose_jobs = (shard[0][k] for k in ("ctx", "streams", "align_jobs", "pose_frames", "pose_jobs"))

        def step_local():
            for sh in shard:
                if not pose_only:
                    sh["ctx"].align_run()
                sh["ctx"].poseopt_run()

        n_local = local_shards * B
        # what a rank publishes per stream: the 96-byte plsvo_pose_record (pose + n_tracked, num_obs_pt, num_obs_ls, status),
        # packed on the device from the resident state of the two launches, on their stream
        REC = P.abi.POSE_RECORD_BYTES
        local_poses = torch.empty((n_local, REC), dtype=torch.uint8, device=dev)

        def copy_local(t):
            for v, sh in enumerate(shard):
                sh["ctx"].pack_pose_records(t.data_ptr() + v * B * REC)

        comm = P.rccl.comm_over_process_group() if (use_dist and not dry) else None       # the gather is the C ABI's, on the library's stream
        def gather(local, out):
            ctx.gather_poses(comm, local.data_ptr(), n_local, out.data_ptr())

        def timers_on():                        # hipEvent pairs around every launch of the timed steps only
            for sh in shard:
                sh["ctx"].set_profiling(True)
                sh["ctx"].reset_profiling()
        elapsed, gathered = D.timed_sharded_steps(step_local, copy_local, local_poses, args.steps, args.warmup,
                                                  device_sync=lambda: torch.cuda.synchronize(dev), before_timed=timers_on,
                                                  gather=gather if (use_dist and not dry) else None, force_gather=args.dist_selftest)
        for sh in shard:
            sh["ctx"].set_profiling(False)
        other_policy = None
        if world == 1 and not use_dist:
            # the same K steps under the OTHER launch-order policy (same protocol, one process: no barrier needed), reported beside `value`
            for sh in shard:
                sh["ctx"].set_launch_order_refresh(align=not order_refresh, poseopt=not order_refresh)
            for _ in range(max(args.warmup, 2)):
                step_local()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step_local()
            torch.cuda.synchronize(dev)
            other_policy = time.perf_counter() - t0
            for sh in shard:
                sh["ctx"].set_launch_order_refresh(align=order_refresh, poseopt=order_refresh)
            step_local()                        # the resident results / work counters read below are those of the line's own policy
            torch.cuda.synchronize(dev)
        gather_us = None
        if comm is not None:
            # latency of the pose all-gather alone (plsvo_gather_poses = ncclAllGather on the library's stream): a baseline for the
            # first multi-GPU run to compare with.  8 records = BASELINE configs[3]'s per-GPU shard, n_local = this run's.
            torch.cuda.synchronize(dev)
            gather_us = {}
            for nrec in sorted({min(8, n_local), n_local}):
                outb = torch.empty((world * nrec, REC), dtype=torch.uint8, device=dev)
                for _ in range(20):
                    ctx.gather_poses(comm, local_poses.data_ptr(), nrec, outb.data_ptr())
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(200):
                    ctx.gather_poses(comm, local_poses.data_ptr(), nrec, outb.data_ptr())
                torch.cuda.synchronize(dev)
                gather_us[str(nrec)] = round((time.perf_counter() - t0) / 200 * 1e6, 2)
                del outb
            torch.cuda.synchronize(dev)
            P.rccl.comm_destroy(comm)

        # ---- roofline of the dominant kernel, from live hipEvent timings on the launch stream ----
        lvl_ms, lvl_launches, pose_ms, pose_launches = 0.0, 0, 0.0, 0
        for sh in shard:
            if not pose_only:
                a_ms, a_n = sh["ctx"].kernel_time(abi.K_ALIGN_LEVEL)
                lvl_ms += a_ms; lvl_launches += a_n
            p_ms, p_n = sh["ctx"].kernel_time(abi.K_POSEOPT)
            pose_ms += p_ms; pose_launches += p_n
        res = [] if pose_only else ctx.align_fetch()
        pres = ctx.poseopt_fetch()
        result = None
        if rank == 0:
            frames = world * n_local * args.steps
            value = frames / elapsed
            if pose_only:
                pt_it, seg_it = ctx.poseopt_work()
                alg_bytes = pt_it * BYTES_PER_POINT_ITER + seg_it * BYTES_PER_SEG_ITER
                avg_ms = pose_ms / max(pose_launches, 1)
                achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
                roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                            "traffic": None, "kernel": "pose_opt_kernel", "avg_launch_ms": round(avg_ms, 4), "launches": int(pose_launches),
                            "algorithmic_bytes_per_launch": int(alg_bytes),
                            "definition": "SURVEY.md 8(d) algorithmic bytes (24 B per point-iteration, 40 B per segment-iteration, device-side counters) / "
                                          "hipEvent time of the launch; informational: the kernel is latency-bound by construction (DESIGN.md 3.2)"}
            else:
                patch_levels = patch_iters = 0                      # counted on the device, per run of the staged batch
                for sh in shard:
                    a_, b_ = sh["ctx"].align_work()
                    patch_levels += a_; patch_iters += b_
                pt_iters = sum(sh["ctx"].align_work_points() for sh in shard)
                flags = (capi.lib().plsvo_hip_build_flags() or b"").decode().split() if hasattr(capi.lib(), "plsvo_hip_build_flags") else []
                cache_b = 64
                OWN_ITER, OWN_LEVEL = OWN_BYTES_PER_PATCH_ITER, OWN_BYTES_PER_PATCH_LEVEL
                MIN_ITER, MIN_LEVEL = MIN_BYTES_PER_PATCH_ITER, MIN_BYTES_PER_PATCH_LEVEL
                survey_bytes = patch_levels * BYTES_PER_PATCH_LEVEL + patch_iters * BYTES_PER_PATCH_ITER
                own_bytes = patch_levels * OWN_LEVEL + patch_iters * OWN_ITER + pt_iters * CHI_BYTES_PER_POINT_ITER
                min_bytes = patch_levels * MIN_LEVEL + patch_iters * MIN_ITER + pt_iters * CHI_BYTES_PER_POINT_ITER
                launches_per_step = max(lvl_launches, 1) / max(args.steps, 1)          # one per shard
                avg_ms = lvl_ms / max(lvl_launches, 1)
                per_launch = lambda nbytes: nbytes / launches_per_step
                rate = lambda nbytes: per_launch(nbytes) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
                # `achieved` / `frac` are a BANDWIDTH, bounded by the peak: the memory-side bytes of the PMC passes (when profiles/ holds a pass
                # of THIS kernel -- same source hash -- else the bytes this formulation cannot avoid moving) / the launch's hipEvent time
                # on the launch stream.  SURVEY.md 8(d)'s reference-layout bytes (485 B per patch-iteration, incl. the 384-B per-pixel Jacobian
                # cache this kernel never materialises: five sums per patch instead) x the device-counted units / that time is a WORK RATE in
                # the reference's units -- it passes the HBM peak -- and is reported as such, under its own name.
                tr = offline_traffic(n_local, args.config) if (args.config in (2, 3) and not flags) else {"bytes": None, "source": None, "raw": None, "same_kernel": False, "same_batch": False}
                traffic = tr["bytes"] if tr["same_kernel"] else None
                basis_bytes = traffic if traffic else per_launch(min_bytes)
                achieved = basis_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
                work_rate = rate(survey_bytes)
                roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                            "traffic": traffic,
                            "basis": ("pmc_traffic: memory-side bytes per launch of the FETCH_SIZE / WRITE_SIZE passes on this kernel (profiles/, offline) / live hipEvent launch time"
                                      if traffic else
                                      "formulation_min: the bytes this formulation cannot avoid moving (no PMC pass of this kernel's sources in profiles/) / live hipEvent launch time"),
                            "traffic_source": tr["source"], "traffic_uncorrected": tr["raw"] if tr["same_kernel"] else None,
                            "traffic_scaled_from_other_batch": (not tr["same_batch"]) if traffic else None,
                            "traffic_of_other_kernel_sources": (tr["bytes"] if (tr["bytes"] and not tr["same_kernel"]) else None),
                            "kernel": "align_fused_kernel", "kernel_source_sha": kernel_source_sha(),
                            "avg_launch_ms": round(avg_ms, 4), "launches": int(lvl_launches),
                            "algorithmic_bytes_per_launch": int(basis_bytes),      # the bytes `achieved` divides by the launch time (see `basis`)
                            "formulation_min_bytes_per_launch": int(per_launch(min_bytes)),
                            "formulation_min_GBps": round(rate(min_bytes), 1), "formulation_min_frac": round(rate(min_bytes) / HBM_PEAK_GBPS, 4),
                            "formulation_min_definition": ("bytes THIS formulation cannot avoid moving: per patch-iteration 25 B window of the current image + %d B "
                                                           "record of the reference patch + 24 B 3-D point, + 64 B of chi2 terms per POINT patch-iteration "
                                                           "written while armed; per patch-level 49 B reference window + %d B cache and point written") % (cache_b, cache_b + 24),
                            "kernel_requested_bytes_per_launch": int(per_launch(own_bytes)),
                            "kernel_requested_GBps": round(rate(own_bytes), 1),
                            "traffic_over_requested": round(traffic / per_launch(own_bytes), 3) if traffic else None,
                            "traffic_over_formulation_min": round(traffic / per_launch(min_bytes), 3) if traffic else None,
                            "traffic_frac": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if traffic and avg_ms > 0 else None,
                            # SURVEY 8(d)'s units: a work rate, NOT a bandwidth (can pass the peak)
                            "survey_algorithmic_bytes_per_launch": int(per_launch(survey_bytes)),
                            "work_rate_survey_units_GBps": round(work_rate, 1), "work_rate_over_hbm_peak": round(work_rate / HBM_PEAK_GBPS, 4),
                            "work_rate_definition": ("SURVEY.md 8(d) reference-layout bytes (485 B per patch-iteration: 25 B window + 64 B cached reference intensity + "
                                                     "384 B cached per-pixel Jacobian + 12 B point; 497 B per patch-level) x the units counted on the device / the launch time: "
                                                     "what the reference's data layout would have to stream to do this work -- this kernel keeps 5 patch sums instead of the "
                                                     "Jacobian cache, so it is not bytes moved and not bounded by the peak"),
                            "patch_levels_per_step": int(patch_levels), "patch_iters_per_step": int(patch_iters), "point_patch_iters_per_step": int(pt_iters)}
            result = {
                "metric": cfg["metric"],
                "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64",
                "data": "synthetic" if not dry else "synthetic -- DRY RUN on the host emulation build of the library: NOT a measurement",
                "config": {"workload": cfg["workload"], "streams_per_gpu": n_local, "global_batch": world * n_local,
                           "shards": shards_total, "streams_per_shard": B,
                           "parallelism": (f"streams sharded x{world}, all-gather of 96-byte pose records through plsvo_gather_poses (RCCL)" if world > 1 else
                                           ("single GPU, N>1 code path with one rank (--dist-selftest)" if args.dist_selftest else "single GPU"))},
                "roofline": roofline,
                "kernel_ms_per_step": {"align_fused": round(lvl_ms / args.steps, 4), "pose_opt": round(pose_ms / args.steps, 4)},
            }
            result["launch_order"] = {
                "policy": "refresh" if order_refresh else "staged",
                "what": ("refresh: a re-run of the staged batch starts its frames longest-first by the work the previous launch measured (align_reorder_kernel; "
                         "PLSVO_OPT_ALIGN_REORDER / _POSEOPT_REORDER = 1, the library's default); staged: every launch keeps the stage call's order.  The timed "
                         "steps re-run ONE staged batch, i.e. identical inputs: `moving_inputs` (N = 1, default workload) measures what is left of the refresh "
                         "when every launch sees a new image; the policy constant bench.py::LAUNCH_ORDER_REFRESH was set from that measurement "
                         "(profiles/r06_launch_order_moving_inputs.log)"),
                "value_other_policy": round(frames / other_policy, 1) if other_policy else None,
                "other_policy": ("staged" if order_refresh else "refresh") if other_policy else None}
            if gather_us is not None:
                result["pose_gather_latency_us"] = {"records_to_us_per_call": gather_us, "ranks": world,
                                                    "what": "plsvo_gather_poses alone (ncclAllGather of 96-byte records on the library's stream), 200 calls enqueued back to back, "
                                                            "one synchronisation: with one rank the collective's fixed cost, the baseline a multi-GPU run adds its transport to"}
            if os.environ.get("PLSVO_HIP_LIB"):   # an A/B build of the library (tools/ab_variants.sh): the line says which one it measured
                result["config"]["library"] = os.path.basename(os.environ["PLSVO_HIP_LIB"])
                if hasattr(capi.lib(), "plsvo_hip_build_flags"):
                    result["config"]["build_flags"] = (capi.lib().plsvo_hip_build_flags() or b"").decode().strip()
            if local_shards > 1:
                # every shard alone: K steps, then a host synchronisation (what one GPU of the 8-GPU deployment would do per step)
                per = []
                for sh in shard:
                    c = sh["ctx"]
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        if not pose_only:
                            c.align_run()
                        c.poseopt_run()
                    c.synchronize()
                    per.append((time.perf_counter() - t0) / args.steps)
                result["per_shard"] = {"us_per_step": [round(1e6 * t, 1) for t in per], "us_per_step_mean": round(1e6 * float(np.mean(per)), 1),
                                       "frames_per_s_of_one_shard": round(B / float(np.mean(per)), 1),
                                       "note": f"{local_shards} shards of {B} streams run back to back on this GPU; with one shard per GPU the step "
                                               "time of the job is the slowest shard's (plus the pose all-gather): that run is the driver's --gpus 8"}
            if not pose_only and hasattr(ctx.L, "plsvo_align_chi2_ties"):
                gn_it = gn_ties = gn_un = 0
                for sh in shard:
                    a_, b_, c_ = sh["ctx"].align_chi2_ties()
                    gn_it += a_; gn_ties += b_; gn_un += c_
                result["chi2_ties"] = {"gn_iterations_per_step": int(gn_it), "decided_on_exact_float_sums": int(gn_ties), "near_ties_without_terms": int(gn_un),
                                       "what": "iterations whose `new_chi2 > chi2_` decision was taken on the reference's sequential float sums "
                                               "(the two values closer than the sums' own rounding noise)"}
            if not pose_only:
                errs = np.array([synth.se3_log_angle_dist(r.T, s.T_true) for r, s in zip(res[:64], streams[:64])])
                result["accuracy_vs_truth"] = {"median_rot_rad": float(np.median(errs[:, 0])), "median_trans_m": float(np.median(errs[:, 1]))}
            # ---- CPU baseline: the oracle on a bounded sample of the same streams, on this box's host cores ----
            single_fps = None
            if world == 1 and not args.no_cpu_baseline:
                from oracle import binding as ob
                ob.build()
                n_s = min(B, 64)
                what = 2 if pose_only else 3
                pyrs = [] if pose_only else [(ctx.download_pyramid(2 * i), ctx.download_pyramid(2 * i + 1)) for i in range(n_s)]
                # warm-up, and a parity check of the timed batch's first n_s frames against the oracle: the bar of BASELINE.json (1e-4 rad, 1e-4
                # relative translation) on cur_frame->T_f_w_ (src/sparse_img_align.cpp:92) AND on T_cur_from_ref itself (relative to the
                # inter-frame translation); frames outside it are COUNTED in the line (a float-tie flip, INTEGRATION.md 3, can leave a frame
                # outside the second bar), a gross disagreement stops the benchmark
                chk = {"frames": 0, "outside_bar_T_f_w": 0, "outside_bar_T_cur_from_ref": 0, "worst_rot_rad": 0.0, "worst_trans_rel_T_f_w": 0.0,
                       "worst_trans_rel_T_cur_from_ref": 0.0, "different_iteration_counts": 0}
                for i in range(n_s):
                    if not pose_only:
                        ro, _ = ob.sparse_align(align_jobs[i], pyrs[i][0], pyrs[i][1])
                        ang, dist = synth.se3_log_angle_dist(ro.T, res[i].T)
                        assert ang < 1e-3, "bench batch disagrees with the oracle (alignment)"
                        fo, fd = synth.se3_mul(np.asarray(ro.T, float), streams[i].T_ref_w), synth.se3_mul(np.asarray(res[i].T, float), streams[i].T_ref_w)
                        ang_f, dist_f = synth.se3_log_angle_dist(fo, fd)
                        rel_f = dist_f / max(float(np.linalg.norm(fo[4:])), 1e-2)
                        rel_i = dist / max(float(np.linalg.norm(np.asarray(ro.T)[4:])), 1e-3)
                        chk["frames"] += 1
                        chk["outside_bar_T_f_w"] += int(not (ang_f <= 1e-4 and rel_f <= 1e-4))
                        chk["outside_bar_T_cur_from_ref"] += int(not (ang <= 1e-4 and rel_i <= 1e-4))
                        chk["worst_rot_rad"] = max(chk["worst_rot_rad"], float(ang))
                        chk["worst_trans_rel_T_f_w"] = max(chk["worst_trans_rel_T_f_w"], float(rel_f))
                        chk["worst_trans_rel_T_cur_from_ref"] = max(chk["worst_trans_rel_T_cur_from_ref"], float(rel_i))
                        chk["different_iteration_counts"] += int(list(ro.iters_per_level) != list(res[i].iters_per_level))
                    if i < 4:
                        po, _ = ob.pose_optimize(pose_jobs[i])
                        ang2, _ = synth.se3_log_angle_dist(po.T, pres[i].T)
                        assert ang2 < 1e-4, "bench batch disagrees with the oracle (pose optimisation)"
                if chk["frames"] and "chi2_ties" in result:
                    chk["what"] = (f"the first {chk['frames']} frames of the timed batch against the CPU oracle on the same inputs: frames outside 1e-4 rad / 1e-4 relative "
                                   "translation on cur_frame->T_f_w_ and on T_cur_from_ref (relative to the inter-frame translation)")
                    result["chi2_ties"]["oracle_check"] = chk
                # the timed loops run inside the oracle library (POSIX threads, no Python between frames)
                half = 0.5 * args.cpu_seconds
                rp, cp = [p[0] for p in pyrs], [p[1] for p in pyrs]
                done1, tc1, lat = ob.bench(align_jobs[:n_s], rp, cp, pose_jobs[:n_s], 1, half, what=what, latencies=True)
                cores = host_cores()
                doneN, tcN = ob.bench(align_jobs[:n_s], rp, cp, pose_jobs[:n_s], cores, half, what=what)
                pinned = ob.bench_threads_pinned()
                single_fps = done1 / tc1
                result["cpu_baseline"] = {
                    "value": round(doneN / tcN, 2), "unit": "frames/s", "cores": cores, "kind": "port",
                    "sample": f"{doneN} frames on {cores} threads in {tcN:.1f} s (independent streams, round-robin over the first "
                              f"{n_s} streams of the timed batch), oracle/libplsvo_oracle.so, timed inside the library; the reference itself is "
                              f"single-threaded on this path; {pinned} of {cores} threads pinned one per allowed CPU; the host has "
                              f"{os.cpu_count()} logical CPUs, the container's affinity mask / CPU quota allows {cores}",
                    "host_logical_cpus": os.cpu_count(), "threads_pinned": pinned,
                    "scaling_over_one_thread": round((doneN / tcN) / single_fps, 2),
                    "single_thread_value": round(single_fps, 2),
                    "single_thread_sample": f"{done1} frames in {tc1:.1f} s on one thread",
                    "single_thread_median_us_per_frame": round(float(np.median(lat)), 1) if len(lat) else None,
                    "build_flags": ORACLE_FLAGS}
                result["speedup_vs_cpu_all_cores"] = round(value / (doneN / tcN), 1)
                result["speedup_vs_cpu_1core"] = round(value / single_fps, 1)
            # ---- small batches and the drop-in's per-call latency (default workload, one GPU) ----
            if world == 1 and args.config == 2 and not args.no_latency:
                try:
                    result["latency"] = latency_leg(P, ctx, align_jobs, pose_jobs, streams, pose_frames, cfg, single_fps)
                    cb = result.get("cpu_baseline")
                    if cb and "B8" in result["latency"]:
                        result["latency"]["B8_vs_cpu_all_cores"] = round(result["latency"]["B8"]["frames_per_s"] / cb["value"], 2)
                except Exception as e:   # an auxiliary leg never takes the headline line down
                    result["latency"] = {"error": str(e)[:300]}
            if world == 1 and args.config == 2 and not args.no_latency and not args.dist_selftest:
                try:
                    result["launch_order"]["moving_inputs"] = moving_leg(P, torch, dev, stream, streams, cfg, **({"n_streams": 8, "n_images": 2} if dry else {}))
                except Exception as e:   # never take the headline line down
                    result["launch_order"]["moving_inputs"] = {"error": str(e)[:300]}
                try:
                    result["host_fed"] = host_fed_leg(P, torch, dev, stream, streams, cfg, **({"steps": 2} if dry else {}))
                except Exception as e:   # never take the headline line down
                    result["host_fed"] = {"error": str(e)[:300]}
                try:
                    result["frame_chain"] = chain_leg(P, torch, dev, stream, streams, cfg, **({"steps": 2, "cpu_frames": 2} if dry else {}))
                except Exception as e:
                    result["frame_chain"] = {"error": str(e)[:300]}
            print(json.dumps(result), flush=True)
        for sh in shard:
            sh["ctx"].close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
