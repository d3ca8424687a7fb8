"""Diagnostic: static camera (cur == ref, T = I) through the HIP path and the oracle: first-iteration chi2 at the top level per feature subset."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
P = importlib.import_module("pl-svo_amd")
from oracle import binding as ob
import helpers as Hh
np.set_printoptions(precision=6, linewidth=220)
st, ref, cur, _ = Hh.make_case(ob, 36, 320, 240, 40, 12, 3, 2, 0)
I = np.array([0, 0, 0, 1, 0, 0, 0.0])
ctx = P.capi.Context(0)
ctx.config_pyramids(2, 320, 240, 3)
ctx.upload_pyramid(0, ref); ctx.upload_pyramid(1, ref)
def run(tag, pts, segs, lvl=2):
    job = P.abi.AlignJob(st.cam, lvl, lvl, 1, 1e-6, I, st.pt_px[pts], st.pt_xyz_ref[pts], st.seg_spx[segs], st.seg_epx[segs], st.seg_len[segs], st.seg_p_ref[segs], st.seg_q_ref[segs])
    ro, lo = ob.sparse_align(job, ref, ref, max_log=4)
    ctx.align_set_trace(4)
    rd = ctx.sparse_align(job)
    ld = ctx.align_fetch_trace(0)
    print(tag, "oracle chi2", lo[0]["new_chi2"], "n", lo[0]["n_meas"], "| device chi2", ld[0]["new_chi2"], "n", ld[0]["n_meas"], "Jres0", ld[0]["Jres"][0], "H00", ld[0]["H"][0, 0])
none = np.zeros(0, int)
run("points only", np.arange(40), none)
for p in range(40):
    job = None
for s in range(12):
    run(f"seg {s} len {st.seg_len[s]:.1f} spx {st.seg_spx[s]} epx {st.seg_epx[s]}", none, np.array([s]))
bad = []
for p in range(40):
    job = P.abi.AlignJob(st.cam, 2, 2, 1, 1e-6, I, st.pt_px[[p]], st.pt_xyz_ref[[p]], st.seg_spx[:0], st.seg_epx[:0], st.seg_len[:0], st.seg_p_ref[:0], st.seg_q_ref[:0])
    ctx.align_set_trace(4); rd = ctx.sparse_align(job); ld = ctx.align_fetch_trace(0)
    if ld and ld[0]["new_chi2"] != 0.0: bad.append((p, ld[0]["new_chi2"]))
print("points with nonzero device chi2:", bad)
