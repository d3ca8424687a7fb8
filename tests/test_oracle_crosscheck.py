"""The C oracle against the independent NumPy restatement (tests/np_restatement.py), per Gauss-Newton
iteration.  With no reference tests/golden vectors to pin against, two separately written restatements
of /root/reference's sources agreeing to ~1e-12 is what stands in."""
import numpy as np
import pytest

import helpers as Hh
import np_restatement as npr

ALIGN_CASES = [("points", 101, 160, 120, 20, 0, 3, 2, 0), ("points+lines", 102, 160, 120, 16, 8, 3, 2, 0),
               ("lines", 103, 200, 150, 0, 10, 3, 2, 1), ("one-level", 104, 160, 120, 12, 4, 2, 1, 1)]


@pytest.mark.parametrize("order", ["reference", "device"])
@pytest.mark.parametrize("case", ALIGN_CASES, ids=[c[0] for c in ALIGN_CASES])
def test_sparse_align_oracle_vs_numpy(P, ob, case, order):
    tag, seed, W, H, npts, nseg, nlev, maxl, minl = case
    st, ref, cur, job = Hh.make_case(ob, seed, W, H, npts, nseg, nlev, maxl, minl)
    ro, lo = ob.sparse_align(job, ref, cur, max_log=200)
    a = npr.SparseAlignNP(st.cam, maxl, minl, 30, 1e-6, order=order)
    r = a.run(st.T_init, ref, cur, st.pt_px, st.pt_xyz_ref, st.seg_spx, st.seg_epx, st.seg_len, st.seg_p_ref, st.seg_q_ref)
    assert len(lo) == len(a.log)
    for x, y in zip(lo, a.log):
        assert (x["level"], x["iter"], x["accepted"], x["n_meas"]) == (y["level"], y["iter"], y["accepted"], y["n_meas"])
        assert Hh.rel(y["H"], x["H"]) < 1e-9
        assert x["new_chi2"] == pytest.approx(y["new_chi2"], rel=1e-6, nan_ok=True)   # 0 measurements -> 0/0
    assert [ro.iters_per_level[l] for l in range(maxl, minl - 1, -1)] == [r["iters"][l] for l in range(maxl, minl - 1, -1)]
    assert np.array_equal(ro.seg_alive.astype(bool), r["alive"])
    assert ro.n_meas == r["n_meas"]
    ang, dist = P.synth.se3_log_angle_dist(ro.T, r["T"])
    assert ang < 1e-9 and dist < 1e-9


POSE_CASES = [("pts+lines", 201, 80, 30, -1), ("ten-arg", 202, 60, 20, 4), ("points", 203, 50, 0, -1), ("lines", 204, 0, 25, -1)]


@pytest.mark.parametrize("case", POSE_CASES, ids=[c[0] for c in POSE_CASES])
def test_pose_optimizer_oracle_vs_numpy(P, ob, case):
    tag, seed, npts, nseg, nref = case
    fr = P.synth.make_poseopt_frame(seed, npts, nseg)
    ro, lo = ob.pose_optimize(P.poseopt_job_from_frame(fr, n_iter_ref=nref), max_log=40)
    r = npr.pose_optimize_np(fr.T_init, fr.fx, 2.0, 10, fr.pt_f, fr.pt_pos, fr.pt_level, fr.seg_line, fr.seg_spos, fr.seg_epos,
                             fr.seg_level, n_iter_ref=nref)
    assert (ro.iters, ro.iters_ref) == (r["iters"], r["iters_ref"])
    for x, y in zip(lo, r["log"]):
        assert x["accepted"] == y["accepted"]
        assert Hh.rel(y["A"], x["A"]) < 1e-11
    assert np.array_equal(ro.pt_keep.astype(bool), r["pt_keep"]) and np.array_equal(ro.seg_keep.astype(bool), r["seg_keep"])
    assert ro.estimated_scale == pytest.approx(r["estimated_scale"], rel=1e-12)
    assert ro.error_init == pytest.approx(r["error_init"], rel=1e-12)
    assert ro.error_final == pytest.approx(r["error_final"], rel=1e-9)
    assert (ro.num_obs_pt, ro.num_obs_ls) == (r["num_obs_pt"], r["num_obs_ls"])
    ang, dist = P.synth.se3_log_angle_dist(ro.T, r["T"])
    assert ang < 1e-12 and dist < 1e-12
    assert Hh.rel(ro.cov, r["cov"]) < 1e-8


def test_known_answer_alignment_recovers_motion(P, ob):
    """behaviour, not just agreement: the aligned pose must land near the synthetic ground truth"""
    st, ref, cur, job = Hh.make_case(ob, 301, 640, 480, 150, 40, 4, 3, 1)
    ro, _ = ob.sparse_align(job, ref, cur)
    a0, d0 = P.synth.se3_log_angle_dist(st.T_init, st.T_true)
    a1, d1 = P.synth.se3_log_angle_dist(ro.T, st.T_true)
    assert a1 < 0.2 * a0 and d1 < 0.2 * d0
    assert ro.n_tracked > 0 and ro.status == 0


def test_known_answer_pose_optimizer(P, ob):
    fr = P.synth.make_poseopt_frame(302, 150, 0, noise_px=1e-3, outlier_frac=0.0)
    ro, _ = ob.pose_optimize(P.poseopt_job_from_frame(fr))
    ang, dist = P.synth.se3_log_angle_dist(ro.T, fr.T_true)
    assert ang < 1e-6 and dist < 1e-5 and ro.pt_keep.all()
    fr = P.synth.make_poseopt_frame(303, 300, 0, noise_px=0.3, outlier_frac=0.1, outlier_px=30.0)
    ro, _ = ob.pose_optimize(P.poseopt_job_from_frame(fr))
    assert not ro.pt_keep[fr.pt_outlier].any() and ro.pt_keep[~fr.pt_outlier].mean() > 0.95
    # no observations at all: early return, nothing written (src/pose_optimizer.cpp:88-89)
    fr = P.synth.make_poseopt_frame(304, 0, 0)
    ro, _ = ob.pose_optimize(P.poseopt_job_from_frame(fr))
    assert ro.status == 1 and np.allclose(ro.T, fr.T_init)


def test_oracle_edge_cases(P, ob):
    # no features: pose untouched, run() == 0
    st, ref, cur, job = Hh.make_case(ob, 305, 160, 120, 0, 0, 3, 2, 0)
    ro, _ = ob.sparse_align(job, ref, cur)
    assert ro.n_tracked == 0 and np.array_equal(ro.T, st.T_init)
    # line patches with exactly zero residual: H += H_*w/0 -> NaN -> solve fails -> stop flag, rollback (Q2)
    st, ref, cur, job = Hh.make_case(ob, 306, 160, 120, 0, 6, 3, 1, 1)
    flat = [np.full_like(l, 77) for l in ref]
    ro, log = ob.sparse_align(job, flat, flat, max_log=10)
    assert ro.status == 1 and log[0]["accepted"] == 0
    assert np.array_equal(ro.T, st.T_init)


def test_ldlt_oracle_against_the_independent_eigen32_restatement(ob):
    """oracle/plsvo_oracle.c follows Eigen's in-place code (swaps, partially updated columns); tests/np_restatement.py states the
    same factorisation from the mathematics (sort the original diagonal, permute, factor without pivoting).  Bitwise agreement on
    full-rank, rank-deficient, tied, indefinite, tiny, zero and badly scaled systems."""
    rng = np.random.default_rng(31)
    ob.set_ldlt_flavour(320)
    cases = []
    for cond_pow in (0, 2, 5, 8, 11):
        for _ in range(6):
            A = rng.normal(0, 1, (20, 6)) * np.logspace(0, -cond_pow / 2.0, 6)[rng.permutation(6)]
            cases.append((A.T @ A, rng.normal(0, 1, 6)))
    for rank in (1, 2, 3, 4, 5):                                    # rank-deficient J^T J, right-hand side in its range
        for _ in range(6):
            J = rng.normal(0, 1, (rank, 6))
            cases.append((J.T @ J, J.T @ rng.normal(0, 1, rank)))
    for n_obs in (1, 2, 3):                                         # point Jacobians: structural zeros and H00 == H11 ties
        for _ in range(6):
            Js = [ob.jacobian_xyz2uv(p) for p in rng.uniform([-1, -1, 2], [1, 1, 6], (n_obs, 3))]
            cases.append((sum(J.T @ J for J in Js), sum(J.T @ rng.normal(0, 1e-2, 2) for J in Js)))
    D = np.diag([3.0, 3.0, 3.0, 1.0, 1.0, 1.0])                     # exact ties everywhere
    cases.append((D, np.arange(6.0)))
    Q, _ = np.linalg.qr(rng.normal(0, 1, (6, 6)))
    cases.append((Q @ np.diag([1.0, -2.0, 3.0, 4.0, -5.0, 6.0]) @ Q.T, np.arange(6.0)))      # indefinite
    cases.append((np.zeros((6, 6)), np.ones(6)))
    cases.append((1e-300 * (np.eye(6) + 0.1), np.ones(6)))
    cases.append((np.diag([1.0, 1e-20, 1.0, 1e-17, 1.0, 1.0]), np.ones(6)))                   # diagonals below the cutoff
    for H, b in cases:
        H = 0.5 * (H + H.T)
        x_c = ob.ldlt_solve6(H, b)
        x_np = npr.eigen32_ldlt_solve(H, b)
        assert np.array_equal(x_c, x_np, equal_nan=True), (H, b, x_c, x_np)
