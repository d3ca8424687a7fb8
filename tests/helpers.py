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


def make_case(ob, seed, W, H, n_pts, n_seg, n_levels, max_level, min_level, n_iter=30, motion_scale=0.5, seg_len_range=None):
    """One synthetic alignment case: stream, pyramids (built by the oracle's half-sampler), job."""
    st = synth.make_align_stream(seed, W, H, n_pts, n_seg, max_level=max_level, motion_scale=motion_scale, seg_len_range=seg_len_range)
    imgs = synth.render_streams([st]).numpy()
    ref = ob.build_pyramid(imgs[0, 0], n_levels)
    cur = ob.build_pyramid(imgs[0, 1], n_levels)
    job = P.align_job_from_stream(st, max_level, min_level, n_iter=n_iter)
    return st, ref, cur, job


def common_prefix(log_ref, log_dev):
    """number of leading trace records in which both paths evaluate the same (level, iteration) and take the same decision"""
    n = 0
    for a, b in zip(log_ref, log_dev):
        if (a["level"], a["iter"]) != (b["level"], b["iter"]):
            break
        n += 1
        if a["accepted"] != b["accepted"]:
            break
    return n


def same_path(log_ref, log_dev):
    """True when the two Gauss-Newton paths visit the same iterations with the same accept / roll-back decisions"""
    return len(log_ref) == len(log_dev) and all((a["level"], a["iter"], a["accepted"]) == (b["level"], b["iter"], b["accepted"])
                                                for a, b in zip(log_ref, log_dev))


def se3_matrix4(T):
    """4x4 homogeneous matrix of a (qx, qy, qz, qw, tx, ty, tz) pose"""
    M = np.eye(4)
    M[:3, :3] = synth.quat_to_R(np.asarray(T, float)[:4])
    M[:3, 3] = np.asarray(T, float)[4:]
    return M


def hat6(u):
    """4x4 twist matrix of a tangent vector (upsilon, omega), Sophus ordering"""
    u = np.asarray(u, float)
    w = u[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = u[:3]
    return M


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
