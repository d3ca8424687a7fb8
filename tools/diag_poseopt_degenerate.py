"""Diagnostic: pose optimiser on noise-free data at the true pose, HIP path next to the CPU oracle (traces printed)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
P = importlib.import_module("pl-svo_amd")
from oracle import binding as ob
np.set_printoptions(precision=6, linewidth=200)
ctx = P.capi.Context(0)
for npts, nseg in [(60, 20), (60, 0)]:
    fr = P.synth.make_poseopt_frame(95, npts, nseg, noise_px=0.0, outlier_frac=0.0, pert_t=0.0, pert_r=0.0)
    job = P.poseopt_job_from_frame(fr)
    ro, lo = ob.pose_optimize(job, max_log=40)
    ctx.poseopt_set_trace(40)
    rd = ctx.pose_optimize(job)
    ld = ctx.poseopt_fetch_trace(0)
    print("==", npts, nseg)
    for name, r, l in (("oracle", ro, lo), ("device", rd, ld)):
        print(name, "iters", r.iters, "obs", r.num_obs_pt, r.num_obs_ls, "err", r.error_init, r.error_final, "scale", r.estimated_scale, "cov diag", np.diag(np.asarray(r.cov).reshape(6, 6)))
        for k, rec in enumerate(l):
            print("  rec", k, "chi2", rec["new_chi2"], "acc", rec["accepted"], "Adiag", np.diag(rec["A"]), "b", rec["b"], "dT", rec["dT"])
