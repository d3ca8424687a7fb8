"""CPU unit tests of the oracle's building blocks: analytic known answers for every piece of arithmetic the
hot path relies on (tier "Unit").  These pin the restated third-party semantics
(Sophus SE3, Eigen LDLT / inverse, vikit robust cost and camera)."""
import math

import numpy as np
import pytest


def test_se3_exp_small_angle_and_identities(ob):
    T = ob.se3_exp(np.zeros(6))
    assert np.allclose(T, [0, 0, 0, 1, 0, 0, 0])
    u = np.array([0.3, -0.2, 0.1, 0.4, -0.5, 0.2])
    T = ob.se3_exp(u)
    Ti = ob.se3_inv(T)
    assert np.allclose(ob.se3_mul(T, Ti), [0, 0, 0, 1, 0, 0, 0], atol=1e-15)
    assert abs(np.linalg.norm(T[:4]) - 1) < 1e-15
    # exp(u) exp(-u) = I
    assert np.allclose(ob.se3_mul(T, ob.se3_exp(-u)), [0, 0, 0, 1, 0, 0, 0], atol=1e-15)
    # rotation part matches Rodrigues
    R, t = ob.se3_matrix(T)
    w = u[3:]
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]) / th
    Rr = np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K
    assert np.allclose(R, Rr, atol=1e-15)
    # pure translation
    assert np.allclose(ob.se3_exp([1, 2, 3, 0, 0, 0])[4:], [1, 2, 3])
    # tiny rotation uses the series branch and stays finite/unit
    T = ob.se3_exp([0, 0, 0, 1e-12, 0, 0])
    assert abs(np.linalg.norm(T[:4]) - 1) < 1e-15 and abs(T[0] - 0.5e-12) < 1e-24


def test_se3_act_matches_matrix(ob):
    rng = np.random.default_rng(1)
    for _ in range(20):
        T = ob.se3_exp(rng.normal(0, 1, 6))
        p = rng.normal(0, 3, 3)
        R, t = ob.se3_matrix(T)
        assert np.allclose(ob.se3_act(T, p), R @ p + t, atol=1e-14)
        A, B = ob.se3_exp(rng.normal(0, 1, 6)), ob.se3_exp(rng.normal(0, 1, 6))
        assert np.allclose(ob.se3_act(ob.se3_mul(A, B), p), ob.se3_act(A, ob.se3_act(B, p)), atol=1e-13)


def test_jacobian_xyz2uv_is_minus_projection_derivative(ob):
    """Frame::jacobian_xyz2uv = - d proj(exp(xi) p) / d xi at xi = 0 (include/plsvo/frame.h:138-160)"""
    rng = np.random.default_rng(2)
    for _ in range(10):
        p = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(2, 6)])
        J = ob.jacobian_xyz2uv(p)
        h = 1e-6
        num = np.zeros((2, 6))
        for k in range(6):
            d = np.zeros(6)
            d[k] = h
            qp, qm = ob.se3_act(ob.se3_exp(d), p), ob.se3_act(ob.se3_exp(-d), p)
            num[:, k] = (qp[:2] / qp[2] - qm[:2] / qm[2]) / (2 * h)
        assert np.allclose(J, -num, atol=1e-8)
    assert ob.jacobian_xyz2uv([0.3, 0.2, 2.0])[0, 1] == 0.0 and ob.jacobian_xyz2uv([0.3, 0.2, 2.0])[1, 0] == 0.0


def test_ldlt_solve_and_inverse(ob):
    rng = np.random.default_rng(3)
    for _ in range(20):
        A = rng.normal(0, 1, (12, 6))
        H = A.T @ A + 1e-3 * np.eye(6)
        b = rng.normal(0, 1, 6)
        assert np.allclose(ob.ldlt_solve6(H, b), np.linalg.solve(H, b), rtol=1e-9, atol=1e-12)
        assert np.allclose(ob.inv6(H), np.linalg.inv(H), rtol=1e-8, atol=1e-10)
    # all-zero system: Eigen's LDLT returns 0 (zero measurements must leave the pose alone)
    assert np.array_equal(ob.ldlt_solve6(np.zeros((6, 6)), np.ones(6)), np.zeros(6))
    # NaN propagates into x[0] (solve() failure path, src/sparse_img_align.cpp:700)
    H = np.eye(6)
    H[2, 3] = H[3, 2] = np.nan
    assert np.isnan(ob.ldlt_solve6(H, np.ones(6))[0])
    # indefinite but non-singular: pivoted LDLT still solves it
    H = np.diag([1.0, -2.0, 3.0, 4.0, -5.0, 6.0])
    H[0, 1] = H[1, 0] = 0.5
    assert np.allclose(ob.ldlt_solve6(H, np.arange(6.0)), np.linalg.solve(H, np.arange(6.0)))


def test_setup_sampling_table(ob):
    """LineFeat::setupSampling (src/feature.cpp:160-173): N = max(1, length / (2*4*corr)), corr = 2 sqrt(1+sin^2)"""
    n, dif = ob.setup_sampling([0, 0], [160, 0], 160.0)          # horizontal: corr = 2 -> N = 160/16
    assert n == 10 and np.allclose(dif, [160, 0])
    n, _ = ob.setup_sampling([0, 0], [0, 97], 97.0)              # vertical, truncation
    assert n == 6
    L = 100 * math.sqrt(2)
    n, _ = ob.setup_sampling([0, 0], [100, 100], L)              # 45 deg: sin = 1/sqrt2, corr = 2 sqrt(1.5)
    assert n == int(L / (8 * 2 * math.sqrt(1.5)))
    n, _ = ob.setup_sampling([5, 5], [8, 6], math.hypot(3, 1))   # short segment: one sample
    assert n == 1
    n, _ = ob.setup_sampling([5, 5], [5, 5], 0.0)                # degenerate: NaN inside, std::max gives 1
    assert n == 1
    # per-level reduction N_l = 1 + (N-1) / 2^l is integer arithmetic (src/sparse_img_align.cpp:320)
    assert [1 + (10 - 1) // (1 << l) for l in range(4)] == [10, 5, 3, 2]


def test_line_normal(ob):
    sf = np.array([0.1, 0.2, 1.0]) / np.linalg.norm([0.1, 0.2, 1.0])
    ef = np.array([-0.3, 0.25, 1.0]) / np.linalg.norm([-0.3, 0.25, 1.0])
    l = ob.line_normal(sf, ef)
    assert abs(math.hypot(l[0], l[1]) - 1) < 1e-15
    assert abs(l @ (sf / sf[2])) < 1e-15 and abs(l @ (ef / ef[2])) < 1e-15   # both end points lie on the line


def test_robust_cost_known_answers(ob):
    assert ob.tukey(0.0) == 1.0
    assert ob.tukey(4.6851) == 0.0 or ob.tukey(4.6851) < 1e-12
    assert ob.tukey(5.0) == 0.0
    assert ob.tukey(float("inf")) == 0.0 and ob.tukey(float("nan")) == 0.0
    x = 2.0
    b2 = np.float32(4.6851) ** 2
    assert ob.tukey(x) == pytest.approx(float((np.float32(1) - np.float32(4) / b2) ** 2), rel=1e-6)
    assert ob.mad_scale([1, 2, 3, 4, 5]) == pytest.approx(1.48 * 3, rel=1e-6)
    assert ob.mad_scale([4, 1, 3, 2]) == pytest.approx(1.48 * 3, rel=1e-6)      # upper median, no averaging
    assert ob.median_f64([0.5, 0.1, 0.9, 0.3]) == 0.5


def test_halfsample_known_answers(ob):
    img = np.array([[10, 11, 200, 202], [12, 13, 201, 203], [0, 1, 2, 3], [255, 255, 255, 255]], dtype=np.uint8)
    sse = ob.halfsample(img, 0)     # avg(avg(a,c),avg(b,d)) with rounding averages
    sca = ob.halfsample(img, 1)     # (a+b+c+d)/4 truncating
    assert sse.tolist() == [[12, 202], [128, 129]]
    assert sca.tolist() == [[11, 201], [127, 128]]
    odd = ob.halfsample(np.zeros((5, 7), np.uint8), 0)
    assert odd.shape == (2, 3)


def test_camera_roundtrip(ob, P):
    cam = P.abi.Pinhole(416.0, 416.0, 320.0, 240.0, 640, 480)
    import ctypes as C
    px = np.array([100.5, 300.25])
    f = np.empty(3)
    ob.lib().plsvo_oracle_cam2world(C.byref(cam), px.ctypes.data_as(P.abi.c_double_p), f.ctypes.data_as(P.abi.c_double_p))
    assert abs(np.linalg.norm(f) - 1) < 1e-15
    back = np.empty(2)
    xyz = f * 3.7
    ob.lib().plsvo_oracle_world2cam(C.byref(cam), xyz.ctypes.data_as(P.abi.c_double_p), back.ctypes.data_as(P.abi.c_double_p))
    assert np.allclose(back, px, atol=1e-12)


def test_cpu_baseline_harness_runs_the_two_hot_functions_on_threads(P, ob):
    """bench.py's cpu_baseline leg: frames are counted inside the oracle library, on POSIX threads"""
    import helpers as Hh
    cases = [Hh.make_case(ob, 300 + i, 160, 120, 20, 6, 4, 3, 1) for i in range(3)]
    pj = [P.poseopt_job_from_frame(P.synth.make_poseopt_frame(300 + i, 40, 10)) for i in range(3)]
    d1, t1 = ob.bench([c[3] for c in cases], [c[1] for c in cases], [c[2] for c in cases], pj, 1, 0.3)
    d2, t2 = ob.bench([c[3] for c in cases], [c[1] for c in cases], [c[2] for c in cases], pj, 2, 0.3)
    assert d1 > 0 and d2 > 0 and 0.25 < t1 < 30.0 and 0.25 < t2 < 30.0


# ---- independent pins of the restated third-party pieces (nothing below comes from this repository's own code) ----

def _hat6(u):
    """4x4 twist matrix of a tangent vector (upsilon, omega), Sophus ordering"""
    w = u[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = u[:3]
    return M


def test_se3_exp_against_scipy_expm(ob):
    """Sophus::SE3::exp == matrix exponential of the 4x4 twist (scipy.linalg.expm), from GN-update-sized to large angles"""
    from scipy.linalg import expm
    rng = np.random.default_rng(11)
    for scale in (1e-9, 1e-6, 1e-3, 0.05, 0.5, 2.5):
        for _ in range(10):
            u = rng.normal(0, scale, 6)
            R, t = ob.se3_matrix(ob.se3_exp(u))
            E = expm(_hat6(u))
            tol = 5e-15 if scale <= 0.5 else 5e-13      # expm's own Pade error grows with the norm of the twist
            assert np.allclose(R, E[:3, :3], atol=tol, rtol=0), scale
            assert np.allclose(t, E[:3, 3], atol=tol, rtol=0), scale


def test_se3_mul_inv_against_4x4_matrices(ob):
    rng = np.random.default_rng(12)
    for _ in range(20):
        A, B = ob.se3_exp(rng.normal(0, 0.7, 6)), ob.se3_exp(rng.normal(0, 0.7, 6))
        def M(T):
            R, t = ob.se3_matrix(T)
            X = np.eye(4); X[:3, :3] = R; X[:3, 3] = t
            return X
        assert np.allclose(M(ob.se3_mul(A, B)), M(A) @ M(B), atol=1e-14)
        assert np.allclose(M(ob.se3_inv(A)), np.linalg.inv(M(A)), atol=1e-14)


def test_ldlt_against_scipy_ldl_pivot_order_and_solution(ob):
    """Eigen's LDLT pivots on the largest remaining |diagonal|; scipy.linalg.ldl (LAPACK sytrf, Bunch-Kaufman) is a different
    pivoting of the same factorisation: both must reproduce H and solve H x = b to the conditioning of H."""
    from scipy.linalg import ldl, solve
    rng = np.random.default_rng(13)
    for cond_pow in (0, 3, 6, 9):
        A = rng.normal(0, 1, (40, 6)) * np.logspace(0, -cond_pow / 2.0, 6)
        H = A.T @ A
        b = H @ rng.normal(0, 1, 6)
        L, Dm, perm = ldl(H)
        assert np.allclose(L @ Dm @ L.T, H, rtol=1e-12, atol=1e-14 * np.abs(H).max())
        x_ref = solve(H, b, assume_a="sym")
        x = ob.ldlt_solve6(H, b)
        c = np.linalg.cond(H)
        assert np.linalg.norm(x - x_ref) <= 50 * c * 2.2e-16 * np.linalg.norm(x_ref) + 1e-300, cond_pow
        assert np.linalg.norm(H @ x - b) <= 1e-12 * np.linalg.norm(b)


def test_tukey_and_mad_closed_forms(ob):
    """vk::robust_cost: Tukey weight (1 - x^2/b^2)^2 inside b = 4.6851, 0 outside (float arithmetic); MAD scale = 1.48 * upper median"""
    b = np.float32(4.6851)
    for x in (0.0, 0.1, 1.0, 2.5, 4.0, 4.68, 4.69, 10.0):
        xf = np.float32(x)
        expect = float((np.float32(1.0) - xf * xf / (b * b)) ** 2) if xf * xf <= b * b else 0.0
        assert ob.tukey(x) == pytest.approx(expect, rel=2e-7, abs=1e-12)
    rng = np.random.default_rng(14)
    for n in (1, 2, 5, 8, 101):
        v = np.abs(rng.normal(0, 1, n)).astype(np.float32)
        assert ob.mad_scale(v) == pytest.approx(1.48 * float(np.sort(v)[n // 2]), rel=1e-6)
        assert ob.median_f64(v.astype(np.float64)) == float(np.sort(v.astype(np.float64))[n // 2])


def test_ldlt_zero_pivot_rules_of_the_two_eigen_releases(ob, P):
    """PL-SVO does not pin an Eigen release (README.md:27 names Ubuntu 12.04 / 14.04 / 16.04), and ldlt().solve() treats zero pivots
    differently in 3.2 (relative cutoff, eps * largest diagonal) and 3.3 (exact zero only).  The oracle restates both (default 320):
      * on full-rank systems -- everything this path produces with three or more point observations -- the two are the same
        arithmetic bit for bit, so neither the fixtures nor the parity tests depend on the choice;
      * on rank-deficient normal equations the 3.2 rule returns a basic solution: exact zeros for the directions whose pivots are
        rounding residue, and the remaining components solve the reduced system."""
    rng = np.random.default_rng(21)
    assert ob.set_ldlt_flavour(320) in (320, 330)
    try:
        for cond_pow in (0, 0, 3, 6, 9, 12):
            A = rng.normal(0, 1, (30, 6)) * np.logspace(0, -cond_pow / 2.0, 6)
            H = A.T @ A
            b = rng.normal(0, 1, 6)
            ob.set_ldlt_flavour(320); x320 = ob.ldlt_solve6(H, b)
            ob.set_ldlt_flavour(330); x330 = ob.ldlt_solve6(H, b)
            assert np.array_equal(x320, x330), cond_pow
        # rank 2 (one point observation) and rank 4 with an exact diagonal tie (two point observations: H00 == H11 = sum 1/z^2)
        for n_obs, rank in ((1, 2), (2, 4)):
            pts = rng.uniform([-1, -1, 2], [1, 1, 6], (n_obs, 3))
            Js = [ob.jacobian_xyz2uv(p) for p in pts]
            H = sum(J.T @ J for J in Js)
            b = sum(J.T @ rng.normal(0, 1e-2, 2) for J in Js)
            if n_obs == 2:
                assert H[0, 0] == H[1, 1]
            ob.set_ldlt_flavour(320)
            x = ob.ldlt_solve6(H, b)
            v = np.flatnonzero(x != 0.0)
            assert len(v) == rank, (n_obs, x)
            assert np.allclose(H[np.ix_(v, v)] @ x[v], b[v], rtol=1e-9, atol=1e-14)
            assert np.linalg.norm(H @ x - b) <= 1e-9 * np.linalg.norm(b)      # b is in the range of H: a basic solution is exact
        # a whole frame: identical under both rules when it is full rank
        fr = P.synth.make_poseopt_frame(78, 60, 20)
        job = P.poseopt_job_from_frame(fr)
        ob.set_ldlt_flavour(320); r320, _ = ob.pose_optimize(job)
        ob.set_ldlt_flavour(330); r330, _ = ob.pose_optimize(job)
        assert np.array_equal(r320.T, r330.T) and np.array_equal(r320.pt_keep, r330.pt_keep) and r320.iters == r330.iters
        # one point observation: the 3.2 rule keeps the step inside the two observable directions
        ob.set_ldlt_flavour(320)
        job = P.poseopt_job_from_frame(P.synth.make_poseopt_frame(83, 1, 0))
        r, log = ob.pose_optimize(job, max_log=10)
        assert all(np.count_nonzero(rec["dT"]) <= 2 for rec in log[:3]) and r.error_final < 1e-9
    finally:
        ob.set_ldlt_flavour(320)
