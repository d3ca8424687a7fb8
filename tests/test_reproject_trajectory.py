"""Reprojection of landmarks into a frame (Reprojector::reproject, src/reprojector.cpp:389-423) -- the producer of the
direct matcher's candidates -- and the trajectory wire format of the reference's harness (app/run_pipeline.cpp:425-451).
CPU: oracle against NumPy and known answers.  GPU: bit-exact against the oracle."""
import numpy as np
import pytest

import np_restatement as npr


def _job(P, seed, n=400, n_frames=3, W=640, H=480, cell=30):
    rng = np.random.default_rng(seed)
    cam = (0.65 * W, 0.65 * W, W / 2.0, H / 2.0, W, H)
    frame_T = np.stack([P.synth.se3_exp(np.concatenate([rng.uniform(-0.3, 0.3, 3), rng.uniform(-0.1, 0.1, 3)])) for _ in range(n_frames)])
    pos = np.stack([rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(-1.0, 8.0, n)], axis=1)   # some behind the camera
    pos[0] = [0.0, 0.0, 0.0]                                                                             # z ~ t_z: may divide by ~0
    frame = rng.integers(0, n_frames, n).astype(np.int32)
    return P.abi.ReprojectJob(cam, frame_T, frame, pos, cell_size=cell), cam


def test_oracle_reproject_matches_numpy(P, ob):
    job, cam = _job(P, 1)
    r = ob.reproject(job)
    fx, fy, cx, cy, W, H = cam
    n_in = 0
    for i in range(job.n):
        c = npr.se3_act(job.frame_T[job.frame[i]], job.pos[i])
        px = np.array([fx * c[0] / c[2] + cx, fy * c[1] / c[2] + cy])
        assert np.allclose(r["px"][i], px, rtol=1e-12, atol=1e-9)
        ox, oy = int(r["px"][i][0]), int(r["px"][i][1])
        inside = 8 <= ox < W - 8 and 8 <= oy < H - 8
        cell = int(r["px"][i][1] / 30) * 22 + int(r["px"][i][0] / 30) if inside else -1     # ceil(640/30) = 22 columns
        assert r["cell"][i] == cell
        n_in += inside
    assert 20 < n_in < job.n


def test_trajectory_record_and_format(P, ob):
    T = P.synth.se3_exp(np.array([0.3, -0.2, 1.5, 0.05, -0.1, 0.2]))
    cov = np.full((6, 6), 1e-6)
    ok, rec = ob.trajectory_record(T, cov)
    Tinv = P.synth.se3_inv(T)
    assert ok and np.allclose(rec[:3], Tinv[4:], atol=1e-15) and np.allclose(rec[3:], Tinv[:4], atol=1e-15)
    # skip rules: a zero / huge covariance entry, and the exact identity pose
    bad = cov.copy(); bad[2, 3] = 0.0
    assert not ob.trajectory_record(T, bad)[0]
    bad[2, 3] = 1e17
    assert not ob.trajectory_record(T, bad)[0]
    assert ob.trajectory_record(np.array([0, 0, 0, 1.0, 0, 0, 0]), cov)[0] is False
    # the product's host-only record equals the oracle's bit for bit (no device needed), and the text format is %g
    ok2, rec2 = P.capi.trajectory_record(T, cov)
    assert ok2 and np.array_equal(rec, rec2)
    line = P.trajectory.tum_line("1403636579.763555", T, cov)
    fields = line.split()
    assert fields[0] == "1403636579.763555" and len(fields) == 8
    assert [float(x) for x in fields[1:]] == [float("%g" % v) for v in rec]
    assert P.trajectory.tum_line("0", T, bad) is None


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n", [(11, 1000), (12, 1), (13, 100000)])
def test_hip_reproject_is_bit_exact(P, ob, gpu_ctx, seed, n):
    job, _ = _job(P, seed, n=n)
    ro, rd = ob.reproject(job), gpu_ctx.reproject(job)
    assert np.array_equal(ro["cell"], rd["cell"])
    assert np.array_equal(np.nan_to_num(ro["px"], nan=-1, posinf=-2, neginf=-3), np.nan_to_num(rd["px"], nan=-1, posinf=-2, neginf=-3))


@pytest.mark.gpu
def test_hip_reproject_feeds_the_matcher(P, ob, gpu_ctx):
    """reproject -> match_direct on the device gives the oracle's chain bit for bit"""
    st, d = P.synth.make_match_batch(51, 320, 240, 60, 12)
    imgs = P.synth.render_streams([st]).numpy()[0]
    frames = [ob.build_pyramid(imgs[0], 4), ob.build_pyramid(imgs[1], 4)]
    gpu_ctx.config_pyramids(2, 320, 240, 4)
    gpu_ctx.build_pyramid(0, frames[0][0], 0); gpu_ctx.build_pyramid(1, frames[1][0], 0)
    rj = P.abi.ReprojectJob(d["cam"], d["frame_T"], d["cur_frame"], d["pos"], cell_size=30)
    pr_o, pr_d = ob.reproject(rj), gpu_ctx.reproject(rj)
    assert np.array_equal(pr_o["px"], pr_d["px"]) and np.array_equal(pr_o["cell"], pr_d["cell"])
    keep = pr_d["cell"] >= 0
    assert keep.sum() > 20
    for k in ("cur_frame", "ref_frame", "ref_px", "ref_f", "ref_level", "ref_type", "ref_grad", "pos"):
        d[k] = d[k][keep]
    d["px_cur"] = pr_d["px"][keep]
    mj = P.match_job_from_batch(d)
    mo, md = ob.match_direct(mj, frames), gpu_ctx.match_direct(mj)
    assert np.array_equal(mo["found"], md["found"])
    assert np.array_equal(np.nan_to_num(mo["px_cur"], nan=-1), np.nan_to_num(md["px_cur"], nan=-1))
    assert mo["found"].mean() > 0.3
