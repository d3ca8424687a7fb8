"""Shared helpers for the parity tests (test infrastructure)."""
import importlib

import numpy as np

P = importlib.import_module("pl-svo_amd")
synth = P.synth

# parity bar from BASELINE.json north_star: 1e-4 rad, 1e-4 relative translation
ROT_TOL = 1e-4
TRANS_REL_TOL = 1e-4


def pose_close(Ta, Tb, scale=None, rot_tol=ROT_TOL, trans_tol=TRANS_REL_TOL):
    """rotation angle and translation error between two poses; translation relative to `scale`
    (defaults to max(|t|, 1e-3 m... of the reference pose))."""
    ang, dist = synth.se3_log_angle_dist(np.asarray(Ta), np.asarray(Tb))
    ref = scale if scale is not None else max(float(np.linalg.norm(np.asarray(Tb)[4:])), 1e-2)
    return ang, dist / ref, (ang <= rot_tol and dist / ref <= trans_tol)


def frame_pose(T_cur_from_ref, st):
    """cur_frame->T_f_w_ as SparseImgAlign::run writes it (src/sparse_img_align.cpp:92)"""
    return synth.se3_mul(np.asarray(T_cur_from_ref, float), st.T_ref_w)


def make_case(ob, seed, W, H, n_pts, n_seg, n_levels, max_level, min_level, n_iter=30, motion_scale=0.5):
    """One synthetic alignment case: stream, pyramids (built by the oracle's half-sampler), job."""
    st = synth.make_align_stream(seed, W, H, n_pts, n_seg, max_level=max_level, motion_scale=motion_scale)
    imgs = synth.render_streams([st]).numpy()
    ref = ob.build_pyramid(imgs[0, 0], n_levels)
    cur = ob.build_pyramid(imgs[0, 1], n_levels)
    job = P.align_job_from_stream(st, max_level, min_level, n_iter=n_iter)
    return st, ref, cur, job


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    d = np.max(np.abs(a - b))
    s = max(np.max(np.abs(b)), 1e-300)
    return d / s


def compare_align_logs(log_ref, log_dev, h_tol=1e-6, chi_tol=1e-4):
    """Per-iteration comparison while both traces follow the same path.  Returns (n_compared, worst dict)."""
    worst = dict(H=0.0, Jres=0.0, chi2=0.0, x=0.0)
    n = 0
    # the gradient vanishes at the optimum: normalise its difference by the largest gradient of the level
    jscale = {}
    for a in log_ref:
        jscale[a["level"]] = max(jscale.get(a["level"], 0.0), float(np.max(np.abs(a["Jres"]))))
    for a, b in zip(log_ref, log_dev):
        if (a["level"], a["iter"]) != (b["level"], b["iter"]):
            break
        assert a["n_meas"] == b["n_meas"], f"n_meas differs at level {a['level']} iter {a['iter']}: {a['n_meas']} vs {b['n_meas']}"
        worst["H"] = max(worst["H"], rel(b["H"], a["H"]))
        worst["Jres"] = max(worst["Jres"], float(np.max(np.abs(a["Jres"] - b["Jres"]))) / max(jscale[a["level"]], 1e-300))
        worst["chi2"] = max(worst["chi2"], abs(a["new_chi2"] - b["new_chi2"]) / max(abs(a["new_chi2"]), 1e-300))
        worst["x"] = max(worst["x"], float(np.max(np.abs(a["x"] - b["x"]))))
        n += 1
        if a["accepted"] != b["accepted"]:
            break
    return n, worst
