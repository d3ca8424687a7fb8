"""diagnostic: GN-path differences HIP vs oracle on lines-only / config-2 cases (prints iteration counts and pose deltas)"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as Hh
from oracle import binding as ob
P = importlib.import_module("pl-svo_amd")
ctx = P.capi.Context(0)
for (npts, nseg, maxl, minl, nlev) in ((0, 60, 2, 1, 3), (200, 80, 3, 1, 4)):
    print("case", npts, nseg)
    for seed in range(13, 33):
        st, ref, cur, job = Hh.make_case(ob, seed, 640, 480, npts, nseg, nlev, maxl, minl)
        ro, lo = ob.sparse_align(job, ref, cur, max_log=200)
        ctx.config_pyramids(2, 640, 480, nlev); ctx.upload_pyramid(0, ref); ctx.upload_pyramid(1, cur); ctx.align_set_trace(200)
        rd = ctx.sparse_align(job); ld = ctx.align_fetch_trace(0)
        ang, tr, ok = Hh.pose_close(Hh.frame_pose(rd.T, st), Hh.frame_pose(ro.T, st))
        same = Hh.same_path(lo, ld)
        k = Hh.common_prefix(lo, ld)
        extra = ""
        if not same and k < min(len(lo), len(ld)):
            a, b = lo[k - 1], ld[k - 1]
            extra = f" first diff after rec {k}: oracle chi2 {a['new_chi2']:.9g} dev {b['new_chi2']:.9g} acc {a['accepted']}/{b['accepted']}"
        print(f"  seed {seed}: rot {ang:.2e} trans {tr:.2e} {'same path' if same else 'DIFF path'} iters o={ro.iters_per_level[:4]} d={rd.iters_per_level[:4]}{extra}")
