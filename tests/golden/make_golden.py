"""Generates the committed golden fixtures tests/golden/*.npz.

The reference ships no golden vectors and cannot be built here (DESIGN.md "oracle"), so these are produced
by this repo's CPU oracle (oracle/plsvo_oracle.c) on small synthetic inputs; they pin the oracle against
regressions and give the GPU box inputs+expected outputs that do not depend on regenerating anything.
Each file holds the exact inputs (level-0 images, flattened features, parameters) and the oracle's
outputs, including the per-iteration trace.   Run:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
P = importlib.import_module("pl-svo_amd")
from oracle import binding as ob  # noqa: E402
import helpers as Hh  # noqa: E402

ALIGN = [("align_small", 9001, 160, 120, 20, 8, 3, 2, 0), ("align_cfg1_crop", 9002, 320, 240, 60, 0, 3, 2, 0),
         ("align_lines", 9003, 320, 240, 30, 16, 4, 3, 1)]
POSE = [("pose_small", 9101, 40, 16, -1), ("pose_tenarg", 9102, 60, 20, 3), ("pose_cfg5", 9103, 500, 200, -1)]


def main():
    ob.build()
    for tag, seed, W, H, npts, nseg, nlev, maxl, minl in ALIGN:
        st, ref, cur, job = Hh.make_case(ob, seed, W, H, npts, nseg, nlev, maxl, minl)
        res, log = ob.sparse_align(job, ref, cur, max_log=200)
        np.savez_compressed(
            os.path.join(HERE, tag + ".npz"), ref0=ref[0], cur0=cur[0], n_levels=nlev, max_level=maxl, min_level=minl, n_iter=30,
            eps=1e-6, cam=np.array(st.cam, float), T_init=st.T_init, T_true=st.T_true, T_ref_w=st.T_ref_w, pt_px=st.pt_px,
            pt_xyz_ref=st.pt_xyz_ref, seg_spx=st.seg_spx, seg_epx=st.seg_epx, seg_len=st.seg_len, seg_p_ref=st.seg_p_ref,
            seg_q_ref=st.seg_q_ref, out_T=res.T, out_n_meas=res.n_meas, out_iters=np.array(res.iters_per_level),
            out_alive=res.seg_alive, out_chi2=res.chi2, out_H=res.H, out_status=res.status,
            log_level=np.array([r["level"] for r in log]), log_iter=np.array([r["iter"] for r in log]),
            log_accepted=np.array([r["accepted"] for r in log]), log_n_meas=np.array([r["n_meas"] for r in log]),
            log_chi2=np.array([r["new_chi2"] for r in log]), log_H=np.array([r["H"] for r in log]),
            log_Jres=np.array([r["Jres"] for r in log]), log_x=np.array([r["x"] for r in log]),
            log_T=np.array([r["T_after"] for r in log]))
        print(tag, "iters", res.iters_per_level[:nlev], "n_meas", res.n_meas)
    for tag, seed, npts, nseg, nref in POSE:
        fr = P.synth.make_poseopt_frame(seed, npts, nseg)
        res, log = ob.pose_optimize(P.poseopt_job_from_frame(fr, n_iter_ref=nref), max_log=40)
        np.savez_compressed(
            os.path.join(HERE, tag + ".npz"), T_init=fr.T_init, fx=fr.fx, reproj_thresh=2.0, n_iter=10, n_iter_ref=nref,
            pt_f=fr.pt_f, pt_pos=fr.pt_pos, pt_level=fr.pt_level, seg_line=fr.seg_line, seg_spos=fr.seg_spos, seg_epos=fr.seg_epos,
            seg_level=fr.seg_level, out_T=res.T, out_cov=res.cov, out_scale=res.estimated_scale, out_error_init=res.error_init,
            out_error_final=res.error_final, out_num_obs=np.array([res.num_obs_pt, res.num_obs_ls]), out_pt_keep=res.pt_keep,
            out_seg_keep=res.seg_keep, out_iters=np.array([res.iters, res.iters_ref]),
            log_chi2=np.array([r["new_chi2"] for r in log]), log_A=np.array([r["A"] for r in log]),
            log_dT=np.array([r["dT"] for r in log]), log_T=np.array([r["T_after"] for r in log]))
        print(tag, "iters", res.iters, res.iters_ref)


if __name__ == "__main__":
    main()
