"""Low-batch operating points of the hot path on one MI355X (VERDICT r01 "missing" #1): B streams resident in HBM,
K back-to-back steps (no host sync between steps) and K synchronised single steps, per launch shape.
BASELINE config 2 streams (640x480, 200 points + 80 segments, levels 3..1).

  python tools/latency_sweep.py [--batches 1,8,64,512] [--threads 0,128,256,512,1024] [--steps 50] [--out file.json]

threads = 0 means "library default" (no PLSVO_ALIGN_THREADS override)."""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="1,8,64,512")
    ap.add_argument("--threads", default="0")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--out", default="")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--pts", type=int, default=200)
    ap.add_argument("--segs", type=int, default=80)
    ap.add_argument("--pyr", type=int, default=4)
    ap.add_argument("--max-level", type=int, default=3)
    ap.add_argument("--min-level", type=int, default=1)
    ap.add_argument("--same-images", action="store_true", help="experiment: every job reads pyramid slots 0/1 (image gathers always cache-hot)")
    args = ap.parse_args()
    import torch
    P = importlib.import_module("pl-svo_amd")
    on_gpu = torch.cuda.is_available()     # (without a GPU: a dry run against a host emulation build named by PLSVO_HIP_LIB, tests/host/)
    dev = torch.device("cuda", 0) if on_gpu else torch.device("cpu")
    ctx = P.capi.Context(0)
    rows = []
    W, H = args.width, args.height
    for B in [int(x) for x in args.batches.split(",")]:
        streams = [P.synth.make_align_stream(1234 + i, W, H, args.pts, args.segs, max_level=args.max_level) for i in range(B)]
        ctx.config_pyramids(2 * B, W, H, args.pyr)
        for c0 in range(0, B, 256):
            sub = streams[c0:c0 + 256]
            imgs = P.synth.render_streams(sub, device=dev)
            if on_gpu:
                torch.cuda.synchronize()  # the library enqueues on its own stream: the rendered images must be complete before it reads them
            ctx.build_pyramids_dev(2 * c0, 2 * len(sub), imgs.data_ptr(), W, W * H, 0)
            ctx.synchronize()
        jobs = [P.align_job_from_stream(s, args.max_level, args.min_level, ref_slot=0 if args.same_images else 2 * i,
                                        cur_slot=1 if args.same_images else 2 * i + 1) for i, s in enumerate(streams)]
        pjobs = [P.poseopt_job_from_frame(P.synth.make_poseopt_frame(1234 + i, args.pts, args.segs, W, H)) for i in range(B)]
        ctx.align_stage(jobs)
        ctx.poseopt_stage(pjobs)
        ctx.synchronize()
        for T in [int(x) for x in args.threads.split(",")]:
            ctx.set_launch_shapes(align_threads=T)

            def timed(fn, K):
                for _ in range(3):
                    fn()
                ctx.synchronize()
                t0 = time.perf_counter()
                for _ in range(K):
                    fn()
                ctx.synchronize()
                back_to_back = (time.perf_counter() - t0) / K
                t0 = time.perf_counter()
                for _ in range(K):
                    fn()
                    ctx.synchronize()
                synced = (time.perf_counter() - t0) / K
                return back_to_back * 1e6, synced * 1e6

            def both():
                ctx.align_run()
                ctx.poseopt_run()
            a_bb, a_sy = timed(ctx.align_run, args.steps)
            p_bb, p_sy = timed(ctx.poseopt_run, args.steps)
            s_bb, s_sy = timed(both, args.steps)
            ctx.set_profiling(True)
            ctx.reset_profiling()
            for _ in range(10):
                both()
            ctx.synchronize()
            ctx.set_profiling(False)
            k_align = ctx.kernel_time(P.abi.K_ALIGN_LEVEL)
            k_pose = ctx.kernel_time(P.abi.K_POSEOPT)
            res = ctx.align_fetch()
            iters = [sum(r.iters_per_level) for r in res]
            row = {"B": B, "threads": T, "align_us_back_to_back": round(a_bb, 1), "align_us_synced": round(a_sy, 1),
                   "poseopt_us_back_to_back": round(p_bb, 1), "poseopt_us_synced": round(p_sy, 1),
                   "step_us_back_to_back": round(s_bb, 1), "step_us_synced": round(s_sy, 1),
                   "frames_per_s_back_to_back": round(B / (s_bb * 1e-6), 1), "frames_per_s_synced": round(B / (s_sy * 1e-6), 1),
                   "align_kernel_us_hipevent": round(1e3 * k_align[0] / max(k_align[1], 1), 1),
                   "poseopt_kernel_us_hipevent": round(1e3 * k_pose[0] / max(k_pose[1], 1), 1),
                   "gn_iters_mean": round(sum(iters) / len(iters), 2), "gn_iters_max": max(iters)}
            rows.append(row)
            print(json.dumps(row), flush=True)
    if args.out:
        json.dump({"what": "low-batch operating points, BASELINE config-2 streams, one MI355X (tools/latency_sweep.py)", "rows": rows},
                  open(args.out, "w"), indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
