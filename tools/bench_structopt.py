"""Micro-benchmark of structure optimisation (next row #3): landmarks/s on the GPU (kernel time from hipEvents, and
end-to-end through the synchronous C ABI including the PCIe copies) next to the CPU oracle on one core."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
P = importlib.import_module("pl-svo_amd")
from oracle import binding as ob

frames = int(os.environ.get("STRUCT_FRAMES", "4096"))      # 20 points + 20 segments per frame (src/config.cpp defaults)
d = P.synth.make_structure_batch(7, 20 * frames, 20 * frames, 64)
job = P.structopt_job_from_batch(d, 5, 5)
ctx = P.capi.Context(0)
ctx.structure_optimize(job)
ctx.set_profiling(True); ctx.reset_profiling()
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    rd = ctx.structure_optimize(job)
wall = (time.perf_counter() - t0) / reps
ms, n = ctx.kernel_time(P.abi.K_STRUCTOPT)
kern = ms / n * 1e-3
sub = P.structopt_job_from_batch(P.synth.make_structure_batch(7, 20 * 64, 20 * 64, 64), 5, 5)
t0 = time.perf_counter(); ro = ob.structure_optimize(sub); cpu = time.perf_counter() - t0
full = ob.structure_optimize(job)
exact = bool(np.array_equal(full["pt_pos"], rd["pt_pos"]) and np.array_equal(full["seg_spos"], rd["seg_spos"]) and np.array_equal(full["seg_epos"], rd["seg_epos"]))
lm = job.n_pts + job.n_seg
obs_it = int((np.diff(job.pt_obs_off) * rd["pt_iters"]).sum() + 2 * (np.diff(job.seg_obs_off) * rd["seg_iters"]).sum())
print(json.dumps({"landmarks": lm, "frames_equivalent": frames, "kernel_ms": round(kern * 1e3, 4), "landmarks_per_s_kernel": round(lm / kern),
                  "abi_call_ms_incl_pcie": round(wall * 1e3, 3), "landmarks_per_s_abi": round(lm / wall),
                  "cpu_oracle_landmarks_per_s_1core": round((sub.n_pts + sub.n_seg) / cpu), "bit_exact_vs_oracle": exact,
                  "observation_iterations": obs_it, "algorithmic_GBps_kernel": round(obs_it * 80 / kern / 1e9, 1)}))
