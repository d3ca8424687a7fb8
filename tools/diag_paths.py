"""Diagnostic: per-record agreement of the HIP path's Gauss-Newton trace with the oracle's on BASELINE config-2 streams."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
P = importlib.import_module("pl-svo_amd")
from oracle import binding as ob
import helpers as Hh
ctx = P.capi.Context(0)
for seed in (4000, 4001, 4002, 4003):
    st, ref, cur, job = Hh.make_case(ob, seed, 640, 480, 200, 80, 4, 3, 1)
    ro, lo = ob.sparse_align(job, ref, cur, max_log=120)
    ctx.config_pyramids(2, 640, 480, 4)
    ctx.upload_pyramid(0, ref); ctx.upload_pyramid(1, cur)
    ctx.align_set_trace(120)
    rd = ctx.sparse_align(job)
    ld = ctx.align_fetch_trace(0)
    rows = []
    for a, b in zip(lo, ld):
        if (a["level"], a["iter"]) != (b["level"], b["iter"]): break
        rows.append((a["level"], a["iter"], Hh.rel(b["H"], a["H"]), Hh.rel(b["Jres"], a["Jres"]), abs(a["new_chi2"] - b["new_chi2"]) / abs(a["new_chi2"]),
                     float(np.max(np.abs(a["x"] - b["x"])) / np.max(np.abs(a["x"])))))
    print("seed", seed, "records", len(lo), len(ld), "same path", Hh.same_path(lo, ld), "pose", Hh.pose_close(rd.T, ro.T)[:2])
    for r in rows[:4] + rows[-2:]:
        print("   L%d it%2d  H %.2e  Jres %.2e  chi2 %.2e  x %.2e" % r)
