"""Structure optimisation (hot-path contract row (f) "next" #3): plsvo::Point::optimize / LineSeg::optimize,
src/feature3D_impl.cpp:36-174.  CPU: the C oracle against a NumPy restatement and known answers.
GPU: the HIP kernel is compiled without fma contraction and must reproduce the oracle BIT FOR BIT."""
import numpy as np
import pytest

import np_restatement as npr


def _np_point_optimize(frame_T, pos, obs_frame, obs_f, n_iter):
    """NumPy restatement of Point::optimize (independent of oracle/plsvo_oracle.c)"""
    pos = np.array(pos, float)
    old = pos.copy()
    chi2, iters = 0.0, 0
    for it in range(n_iter):
        A, b, new_chi2 = np.zeros((3, 3)), np.zeros(3), 0.0
        iters += 1
        for fr, f in zip(obs_frame, obs_f):
            T = frame_T[fr]
            p = npr.se3_act(T, pos)
            x, y, z, w = T[:4]
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                          [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            zi = 1.0 / p[2]
            P = np.array([[zi, 0, -p[0] * zi * zi], [0, zi, -p[1] * zi * zi]])
            J = (-P) @ R
            e = f[:2] / f[2] - p[:2] / p[2]
            new_chi2 += float(e @ e)
            A += J.T @ J
            b -= J.T @ e
        dp = np.linalg.solve(A, b) if abs(np.linalg.det(A)) > 0 else np.zeros(3)
        if (it > 0 and new_chi2 > chi2) or np.isnan(dp[0]):
            pos = old.copy()
            break
        old = pos.copy()
        pos = pos + dp
        chi2 = new_chi2
        if np.max(np.abs(dp)) <= 1e-10:
            break
    return pos, iters


def test_oracle_points_match_numpy_restatement(P, ob):
    d = P.synth.make_structure_batch(1, n_pts=12, n_seg=0)
    res = ob.structure_optimize(P.structopt_job_from_batch(d))
    for i in range(12):
        o0, o1 = d["pt_obs_off"][i], d["pt_obs_off"][i + 1]
        pos, iters = _np_point_optimize(d["frame_T"], d["pt_pos"][i], d["pt_obs_frame"][o0:o1], d["pt_obs_f"][o0:o1], 5)
        assert iters == res["pt_iters"][i]
        assert np.allclose(pos, res["pt_pos"][i], rtol=0, atol=1e-11)


def test_oracle_segments_are_two_coupled_point_problems(P, ob):
    """LineSeg::optimize runs the point update on both end points but breaks / rolls back jointly (:139-146, :169)"""
    d = P.synth.make_structure_batch(2, n_pts=0, n_seg=10)
    res = ob.structure_optimize(P.structopt_job_from_batch(d))
    for i in range(10):
        o0, o1 = d["seg_obs_off"][i], d["seg_obs_off"][i + 1]
        n = int(res["seg_iters"][i])
        ps, its = _np_point_optimize(d["frame_T"], d["seg_spos"][i], d["seg_obs_frame"][o0:o1], d["seg_obs_sf"][o0:o1], n)
        pe, ite = _np_point_optimize(d["frame_T"], d["seg_epos"][i], d["seg_obs_frame"][o0:o1], d["seg_obs_ef"][o0:o1], n)
        if its == n and ite == n and n < 5:      # both end points ran the same number of steps without a private early stop
            continue
        if its == n and ite == n:
            assert np.allclose(ps, res["seg_spos"][i], atol=1e-10) and np.allclose(pe, res["seg_epos"][i], atol=1e-10)


def test_known_answer_triangulation(P, ob):
    """noise-free bearings from >= 3 views: the landmark must move (close) to its true position"""
    d = P.synth.make_structure_batch(3, n_pts=20, n_seg=20, noise_px=0.0, pert=0.05, obs_range=(3, 6))
    res = ob.structure_optimize(P.structopt_job_from_batch(d, 10, 10))
    e0 = np.linalg.norm(d["pt_pos"] - d["pt_true"], axis=1)
    e1 = np.linalg.norm(res["pt_pos"] - d["pt_true"], axis=1)
    assert np.median(e1) < 1e-6 and np.all(e1 < e0 + 1e-12)
    es = np.linalg.norm(res["seg_spos"] - d["seg_s_true"], axis=1)
    assert np.median(es) < 1e-6


def test_edge_cases(P, ob):
    # a landmark without observations: A = 0 -> dp = 0 -> unchanged after one evaluation
    d = P.synth.make_structure_batch(4, n_pts=3, n_seg=0)
    d["pt_obs_off"] = np.array([0, 0, 0, 0], np.int32)
    res = ob.structure_optimize(P.structopt_job_from_batch(d))
    assert np.array_equal(res["pt_pos"], d["pt_pos"]) and list(res["pt_iters"]) == [1, 1, 1]
    # zero iterations: untouched
    d = P.synth.make_structure_batch(5, n_pts=4, n_seg=4)
    res = ob.structure_optimize(P.structopt_job_from_batch(d, 0, 0))
    assert np.array_equal(res["pt_pos"], d["pt_pos"]) and np.array_equal(res["seg_epos"], d["seg_epos"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(11, 20, 20, 6), (12, 500, 300, 12), (13, 64, 0, 4), (14, 0, 33, 5), (15, 5000, 5000, 40)])
def test_hip_structure_optimize_is_bit_exact(P, ob, gpu_ctx, case):
    seed, npts, nseg, nfr = case
    d = P.synth.make_structure_batch(seed, npts, nseg, nfr)
    job = P.structopt_job_from_batch(d)
    ro = ob.structure_optimize(job)
    rd = gpu_ctx.structure_optimize(job)
    assert np.array_equal(rd["pt_iters"], ro["pt_iters"]) and np.array_equal(rd["seg_iters"], ro["seg_iters"])
    assert np.array_equal(rd["pt_pos"], ro["pt_pos"]), np.abs(rd["pt_pos"] - ro["pt_pos"]).max()
    assert np.array_equal(rd["seg_spos"], ro["seg_spos"]) and np.array_equal(rd["seg_epos"], ro["seg_epos"])


@pytest.mark.gpu
def test_hip_structure_optimize_edge_cases(P, ob, gpu_ctx):
    d = P.synth.make_structure_batch(21, n_pts=3, n_seg=2)
    d["pt_obs_off"] = np.array([0, 0, 0, 0], np.int32)
    job = P.structopt_job_from_batch(d)
    ro, rd = ob.structure_optimize(job), gpu_ctx.structure_optimize(job)
    assert np.array_equal(rd["pt_pos"], ro["pt_pos"]) and np.array_equal(rd["seg_spos"], ro["seg_spos"])
    # landmarks with ONE observation: rank-2 3x3 normal equations -- A.ldlt().solve(b) returns a zero component for the depth
    # direction (Eigen 3.2's zero-pivot rule, oracle flavour 320) and the device must return the same bits
    d = P.synth.make_structure_batch(23, n_pts=6, n_seg=0)
    d["pt_obs_off"] = np.arange(7, dtype=np.int32)
    job1 = P.structopt_job_from_batch(d)
    ro, rd = ob.structure_optimize(job1), gpu_ctx.structure_optimize(job1)
    assert np.all(np.isfinite(ro["pt_pos"]))
    assert np.array_equal(rd["pt_pos"], ro["pt_pos"]) and np.array_equal(rd["pt_iters"], ro["pt_iters"])
    job0 = P.structopt_job_from_batch(P.synth.make_structure_batch(22, 4, 4), 0, 0)
    rd = gpu_ctx.structure_optimize(job0)
    assert np.array_equal(rd["pt_pos"], job0.pt_pos)
