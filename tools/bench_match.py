"""Micro-benchmark of direct feature matching (next row #2): candidates/s on the GPU (kernel time from hipEvents, and
end-to-end through the synchronous C ABI including the PCIe copies) next to the CPU oracle on one core.
MATCH_PAIRS (keyframe, current frame) pairs of 640x480 images, 120 points + 40 segments (= 200 candidates) each."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
P = importlib.import_module("pl-svo_amd")
from oracle import binding as ob

pairs = int(os.environ.get("MATCH_PAIRS", "512"))
W, H = 640, 480
ctx = P.capi.Context(0)
ctx.config_pyramids(2 * pairs, W, H, 4)
streams, ds = [], []
for k in range(pairs):
    st, d = P.synth.make_match_batch(9000 + k, W, H, 120, 40, zoom=0.3 * (k % 3 == 0))
    streams.append(st)
    ds.append(d)
for c0 in range(0, pairs, 128):
    imgs = P.synth.render_streams(streams[c0:c0 + 128], device="cuda:0")
    ctx.build_pyramids_dev(2 * c0, 2 * (min(c0 + 128, pairs) - c0), imgs.data_ptr(), W, W * H, 0)
    ctx.synchronize()
    if c0 == 0:
        host_imgs = imgs[:8].cpu().numpy()
    del imgs
cat = lambda k: np.concatenate([d[k] for d in ds])
big = dict(cam=ds[0]["cam"], frame_T=cat("frame_T"), frame_slot=np.arange(2 * pairs, dtype=np.int32),
           cur_frame=np.concatenate([d["cur_frame"] + 2 * k for k, d in enumerate(ds)]),
           ref_frame=np.concatenate([d["ref_frame"] + 2 * k for k, d in enumerate(ds)]),
           **{k: cat(k) for k in ("ref_px", "ref_f", "ref_level", "ref_type", "ref_grad", "pos", "px_cur")})
job = P.match_job_from_batch(big)
ctx.match_direct(job)
ctx.set_profiling(True); ctx.reset_profiling()
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    rd = ctx.match_direct(job)
wall = (time.perf_counter() - t0) / reps
ms, n = ctx.kernel_time(P.abi.K_MATCH)
kern = ms / n * 1e-3
# CPU oracle on the first 8 pairs (bit-exactness + 1-core rate)
exact, cpu_t, cpu_n = True, 0.0, 0
for k in range(8):
    fr = [ob.build_pyramid(host_imgs[k, 0], 4), ob.build_pyramid(host_imgs[k, 1], 4)]
    j = P.match_job_from_batch(ds[k])
    t0 = time.perf_counter(); ro = ob.match_direct(j, fr); cpu_t += time.perf_counter() - t0
    cpu_n += j.n
    sl = slice(200 * k, 200 * (k + 1))
    same = (np.array_equal(ro["found"], rd["found"][sl]) and np.array_equal(ro["n_iter"], rd["n_iter"][sl]) and
            np.array_equal(np.nan_to_num(ro["px_cur"], nan=-1.0), np.nan_to_num(rd["px_cur"][sl], nan=-1.0)))
    exact = exact and bool(same)
passes = int(rd["n_iter"].sum())
# algorithmic bytes: 400 keyframe bytes per warped candidate + 81 current-image bytes per residual pass + 130 B of candidate record
warped = int((rd["search_level"] >= 0).sum())
alg = warped * 400 + passes * 81 + job.n * 130
print(json.dumps({"candidates": job.n, "pairs": pairs, "found_frac": round(float(rd["found"].mean()), 3), "residual_passes": passes,
                  "kernel_ms": round(kern * 1e3, 4), "candidates_per_s_kernel": round(job.n / kern),
                  "abi_call_ms_incl_pcie": round(wall * 1e3, 3), "candidates_per_s_abi": round(job.n / wall),
                  "cpu_oracle_candidates_per_s_1core": round(cpu_n / cpu_t), "bit_exact_vs_oracle_first_8_pairs": exact,
                  "algorithmic_bytes": alg, "algorithmic_GBps_kernel": round(alg / kern / 1e9, 2),
                  "float_ops_estimate_GFLOPs": round(passes * 64 * 17 / kern / 1e9, 1)}))
