"""Diagnostic: pose optimiser on degenerate inputs, HIP path next to the CPU oracle (traces printed)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
P = importlib.import_module("pl-svo_amd")
from oracle import binding as ob
np.set_printoptions(precision=6, linewidth=200)
ctx = P.capi.Context(0)
for npts, nseg in [(0, 0), (1, 0), (0, 1)]:
    fr = P.synth.make_poseopt_frame(83, npts, nseg)
    job = P.poseopt_job_from_frame(fr)
    ro, lo = ob.pose_optimize(job, max_log=40)
    ctx.poseopt_set_trace(40)
    rd = ctx.pose_optimize(job)
    ld = ctx.poseopt_fetch_trace(0)
    print("==", npts, nseg)
    for name, r, l in (("oracle", ro, lo), ("device", rd, ld)):
        print(name, "T", r.T, "iters", r.iters, "obs", r.num_obs_pt, r.num_obs_ls, "err", r.error_init, r.error_final, "scale", r.estimated_scale,
              "keep", r.pt_keep, r.seg_keep)
        for k, rec in enumerate(l):
            print("  rec", k, {kk: (vv if not isinstance(vv, np.ndarray) else (vv.diagonal() if vv.ndim == 2 else vv)) for kk, vv in rec.items()})
