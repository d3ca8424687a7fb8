"""The C++ drop-in adapter (pl-svo_amd/host/plsvo/hip_adapter.hpp): frames with std::lists of features go in,
the reference's two calls are made as FrameHandlerMono::processFrame makes them
(src/frame_handler_mono.cpp:272-274, 327-329), and every mutation the reference performs is checked against
the oracle: cur_frame->T_f_w_, LineFeat::feat3D = NULL, frame->T_f_w_, Cov_, the scalar outputs, culled features."""
import os
import subprocess

import numpy as np
import pytest

import helpers as Hh

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "pl-svo_amd", "host", "adapter_driver")


def test_adapter_reproduces_reference_mutations(P, ob, tmp_path):
    assert os.path.exists(DRIVER), "build it with __graft_entry__.build()"
    W, H, nlev, maxl, minl, npts, nseg, ndead = 320, 240, 4, 3, 1, 60, 24, 3
    st, ref, cur, _ = Hh.make_case(ob, 777, W, H, npts, nseg, nlev, maxl, minl)
    fr = P.synth.make_poseopt_frame(778, 90, 30, W, H)
    path = tmp_path / "in.bin"
    P.adapter_io.write_adapter_input(path, st, ref, cur, fr, nlev, maxl, minl, n_dead_seg=ndead)
    # landmarks with observation lists for the structure-optimisation step (src/frame_handler_mono.cpp:340)
    sb = P.synth.make_structure_batch(779, 12, 9, 5)
    spath = tmp_path / "struct.bin"
    with open(spath, "wb") as f:
        np.array([5, 12, 9, 5], float).tofile(f)
        sb["frame_T"].astype(np.float64).tofile(f)
        for i in range(12):
            o0, o1 = sb["pt_obs_off"][i], sb["pt_obs_off"][i + 1]
            np.concatenate([sb["pt_pos"][i], [o1 - o0]]).astype(np.float64).tofile(f)
            for o in range(o0, o1):
                np.concatenate([[sb["pt_obs_frame"][o]], sb["pt_obs_f"][o]]).astype(np.float64).tofile(f)
        for i in range(9):
            o0, o1 = sb["seg_obs_off"][i], sb["seg_obs_off"][i + 1]
            np.concatenate([sb["seg_spos"][i], sb["seg_epos"][i], [o1 - o0]]).astype(np.float64).tofile(f)
            for o in range(o0, o1):
                np.concatenate([[sb["seg_obs_frame"][o]], sb["seg_obs_sf"][o], sb["seg_obs_ef"][o]]).astype(np.float64).tofile(f)
    out = tmp_path / "out.txt"
    subprocess.run([DRIVER, str(path), str(out), str(spath)], check=True, timeout=120)
    lines = open(out).read().strip().splitlines()
    got = {l.split()[0]: l.split()[1:] for l in lines if not l.startswith(("spt", "sseg"))}
    so = ob.structure_optimize(P.structopt_job_from_batch(sb))
    spt = np.array([[float(x) for x in l.split()[1:]] for l in lines if l.startswith("spt")])
    sseg = np.array([[float(x) for x in l.split()[1:]] for l in lines if l.startswith("sseg")])
    assert np.array_equal(spt, so["pt_pos"]), "Point::pos_ after the adapter must equal the oracle bit for bit"
    assert np.array_equal(sseg[:, :3], so["seg_spos"]) and np.array_equal(sseg[:, 3:], so["seg_epos"])

    # oracle on the same flattened inputs (first `ndead` segments have no landmark)
    alive_in = np.ones(nseg, np.uint8)
    alive_in[:ndead] = 0
    job = P.abi.AlignJob(st.cam, maxl, minl, 30, 1e-6, st.T_init, st.pt_px, st.pt_xyz_ref, st.seg_spx, st.seg_epx, st.seg_len,
                         st.seg_p_ref, st.seg_q_ref, seg_alive_in=alive_in)
    ro, _ = ob.sparse_align(job, ref, cur)
    T_cur = np.array(got["T_cur"], float)
    ang, tr, ok = Hh.pose_close(T_cur, Hh.frame_pose(ro.T, st))
    assert ok, (ang, tr)
    # run()'s return value: the C++ adapter and the Python binding drive the same kernel with the same inputs -> equal, exactly;
    # against the oracle it is equal whenever the two Gauss-Newton paths are the same path (traces compared record by record)
    ctx = P.capi.Context(0)
    try:
        ctx.config_pyramids(2, W, H, nlev)
        ctx.upload_pyramid(0, ref)
        ctx.upload_pyramid(1, cur)
        ctx.align_set_trace(200)
        rd = ctx.sparse_align(job)
        ld = ctx.align_fetch_trace(0)
    finally:
        ctx.close()
    _, lo = ob.sparse_align(job, ref, cur, max_log=200)
    assert int(got["n_tracked"][0]) == rd.n_tracked
    Hh.compare_align_logs(lo, ld)                       # n_meas equal on every shared iteration
    if Hh.same_path(lo, ld):
        assert int(got["n_tracked"][0]) == ro.n_tracked
    assert [int(x) for x in got["alive"]] == list(ro.seg_alive)
    assert float(got["fisher00"][0]) == pytest.approx(ro.H[0, 0] / (5e-4 * 255 * 255), rel=1e-4)

    po, _ = ob.pose_optimize(P.poseopt_job_from_frame(fr))
    assert Hh.pose_close(np.array(got["T_opt"], float), po.T)[2]
    sc = got["scalars"]
    assert float(sc[0]) == pytest.approx(po.estimated_scale, rel=1e-6)
    assert float(sc[1]) == pytest.approx(po.error_init, rel=1e-9) and float(sc[2]) == pytest.approx(po.error_final, rel=1e-6)
    assert (int(sc[3]), int(sc[4])) == (po.num_obs_pt, po.num_obs_ls)
    assert float(sc[5]) == pytest.approx(po.cov[0, 0], rel=1e-6)
    assert [int(x) for x in got["pt_keep"]] == list(po.pt_keep) and [int(x) for x in got["seg_keep"]] == list(po.seg_keep)


def test_direct_matcher_adapter_reproduces_find_match_direct(P, ob, tmp_path):
    """plsvo::DirectMatcher: PointFeat / LineFeat objects in, one batched run, Matcher::findMatchDirect's outputs
    (return value, px_cur / spx_cur / epx_cur, search_level_) out -- bit for bit the oracle's, twice (the second pass
    is served from the adapter's keyframe-pyramid cache)."""
    driver = os.path.join(ROOT, "pl-svo_amd", "host", "match_driver")
    assert os.path.exists(driver), "build it with __graft_entry__.build()"
    W, H, nlev, npts, nseg = 320, 240, 4, 50, 14
    st, d = P.synth.make_match_batch(881, W, H, npts, nseg, zoom=0.2, edgelet_frac=0.3)
    imgs = P.synth.render_streams([st]).numpy()[0]
    frames = [ob.build_pyramid(imgs[0], nlev), ob.build_pyramid(imgs[1], nlev)]
    path = tmp_path / "match.bin"
    s0, e0 = npts, npts + nseg
    with open(path, "wb") as f:
        np.array([W, H, nlev, npts, nseg, 3], float).tofile(f)
        np.array(st.cam[:4], float).tofile(f)
        d["frame_T"].astype(np.float64).tofile(f)
        for pyr in frames:
            for l in pyr:
                np.ascontiguousarray(l, np.uint8).tofile(f)
        np.hstack([d["ref_px"][:npts], d["ref_f"][:npts], d["ref_level"][:npts, None].astype(float), d["ref_type"][:npts, None].astype(float),
                   d["ref_grad"][:npts], d["pos"][:npts], d["px_cur"][:npts]]).astype(np.float64).tofile(f)
        # a segment's two end points share the observation's level (Feature::level)
        d["ref_level"][e0:] = d["ref_level"][s0:e0]
        np.hstack([d["ref_px"][s0:e0], d["ref_px"][e0:], d["ref_f"][s0:e0], d["ref_f"][e0:], d["ref_level"][s0:e0, None].astype(float),
                   d["pos"][s0:e0], d["pos"][e0:], d["px_cur"][s0:e0], d["px_cur"][e0:]]).astype(np.float64).tofile(f)
    out = tmp_path / "match.txt"
    subprocess.run([driver, str(path), str(out)], check=True, timeout=120)
    ro = ob.match_direct(P.match_job_from_batch(d), frames)
    rows = [l.split() for l in open(out).read().strip().splitlines()]
    for p in ("0", "1"):
        pt = np.array([[float(x) for x in r[1:]] for r in rows if r[0] == "pt" + p])
        sg = np.array([[float(x) for x in r[1:]] for r in rows if r[0] == "seg" + p])
        assert pt.shape == (npts, 4) and sg.shape == (nseg, 6)
        assert np.array_equal(pt[:, 0], ro["found"][:npts]) and np.array_equal(pt[:, 1], ro["search_level"][:npts])
        assert np.array_equal(np.nan_to_num(pt[:, 2:], nan=-1), np.nan_to_num(ro["px_cur"][:npts], nan=-1))
        assert np.array_equal(sg[:, 0], ro["found"][s0:e0] & ro["found"][e0:]) and np.array_equal(sg[:, 1], ro["search_level"][e0:])
        assert np.array_equal(np.nan_to_num(sg[:, 2:4], nan=-1), np.nan_to_num(ro["px_cur"][s0:e0], nan=-1))
        assert np.array_equal(np.nan_to_num(sg[:, 4:6], nan=-1), np.nan_to_num(ro["px_cur"][e0:], nan=-1))
    assert ro["found"].mean() > 0.3


def test_depth_filter_adapter_reproduces_update_seeds(P, ob, tmp_path):
    """plsvo::depth_filter::updateSeeds: std::lists of PointSeed / LineSeed in, the reference's mutations out -- stale seeds
    erased, a/b/mu/sigma2 updated, converged seeds reported with their world position and erased -- against the oracle."""
    driver = os.path.join(ROOT, "pl-svo_amd", "host", "match_driver")
    W, H, nlev, npts, nseg = 320, 240, 4, 40, 10
    st, d = P.synth.make_match_batch(882, W, H, npts, nseg, zoom=0.0, motion_scale=2.0, edgelet_frac=0.2, levels=(0,), level_p=(1.0,))
    imgs = P.synth.render_streams([st]).numpy()[0]
    frames = [ob.build_pyramid(imgs[0], nlev), ob.build_pyramid(imgs[1], nlev)]
    s0, e0 = npts, npts + nseg
    path = tmp_path / "match.bin"
    with open(path, "wb") as f:
        np.array([W, H, nlev, npts, nseg, 3], float).tofile(f)
        np.array(st.cam[:4], float).tofile(f)
        d["frame_T"].astype(np.float64).tofile(f)
        for pyr in frames:
            for l in pyr:
                np.ascontiguousarray(l, np.uint8).tofile(f)
        np.hstack([d["ref_px"][:npts], d["ref_f"][:npts], d["ref_level"][:npts, None].astype(float), d["ref_type"][:npts, None].astype(float),
                   d["ref_grad"][:npts], d["pos"][:npts], d["px_cur"][:npts]]).astype(np.float64).tofile(f)
        np.hstack([d["ref_px"][s0:e0], d["ref_px"][e0:], d["ref_f"][s0:e0], d["ref_f"][e0:], d["ref_level"][s0:e0, None].astype(float),
                   d["pos"][s0:e0], d["pos"][e0:], d["px_cur"][s0:e0], d["px_cur"][e0:]]).astype(np.float64).tofile(f)
    # seeds: uninformed priors around the scene depth; seeds 1, 5, 9, .. are nearly converged, seed 2 is too old (erased unseen)
    ref_pos = P.synth.se3_inv(d["frame_T"][0])[4:]
    depth = np.linalg.norm(d["pos"] - ref_pos, axis=1)
    dmean, dmin = float(depth.mean()), 0.8 * float(depth.min())
    f32 = lambda v: float(np.float32(v))
    pt_rows = [[i, 5, 10.0, 10.0, f32(1 / dmean), f32(1 / dmin), f32((1 / dmin) ** 2 / 36)] for i in range(npts)]
    for i in range(1, npts, 4):                                 # every fourth seed is nearly converged around the true depth
        pt_rows[i][4], pt_rows[i][6] = f32(1 / depth[i]), f32(1e-6)
    pt_rows[2][1] = 1                                           # batch_counter - batch_id = 4 > max_n_kfs = 3
    mid = 0.5 * (d["ref_px"][s0:e0] + d["ref_px"][e0:])
    fx, fy, cx, cy = st.cam[:4]
    fm = np.stack([(mid[:, 0] - cx) / fx, (mid[:, 1] - cy) / fy, np.ones(nseg)], axis=1)
    fm /= np.linalg.norm(fm, axis=1, keepdims=True)
    seg_rows = [[i, 5, 10.0, 10.0, f32(1 / dmean), f32(1 / dmean), f32(1 / dmin), f32(1 / dmin), f32((1 / dmin) ** 2 / 36), f32((1 / dmin) ** 2 / 36),
                 mid[i, 0], mid[i, 1], fm[i, 0], fm[i, 1], fm[i, 2]] for i in range(nseg)]
    spath = tmp_path / "seeds.bin"
    with open(spath, "wb") as f:
        np.array([npts, nseg, 5], float).tofile(f)
        np.array(pt_rows, float).tofile(f)
        np.array(seg_rows, float).tofile(f)
    out = tmp_path / "out.txt"
    subprocess.run([driver, str(path), str(out), str(spath)], check=True, timeout=120)
    rows = [l.split() for l in open(out).read().strip().splitlines()]
    # oracle on the same seeds (without the stale one)
    keep = np.array([i for i in range(npts) if i != 2])
    pr = np.array(pt_rows)[keep]
    pt = dict(ref_frame=np.zeros(len(keep), np.int32), cur_frame=np.ones(len(keep), np.int32), px=d["ref_px"][keep], f=d["ref_f"][keep],
              level=d["ref_level"][keep], type=d["ref_type"][keep], grad=d["ref_grad"][keep], a=pr[:, 2], b=pr[:, 3], mu=pr[:, 4], z_range=pr[:, 5],
              sigma2=pr[:, 6])
    sr = np.array(seg_rows)
    seg = dict(ref_frame=np.zeros(nseg, np.int32), cur_frame=np.ones(nseg, np.int32), px=mid, f=fm, sf=d["ref_f"][s0:e0], ef=d["ref_f"][e0:],
               level=d["ref_level"][s0:e0], a=sr[:, 2], b=sr[:, 3], mu_s=sr[:, 4], mu_e=sr[:, 5], z_range_s=sr[:, 6], z_range_e=sr[:, 7],
               sigma2_s=sr[:, 8], sigma2_e=sr[:, 9])
    ro = ob.update_seeds(P.abi.SeedsJob(st.cam, d["frame_T"], np.array([0, 1]), pt, seg), frames)
    got_p = {int(r[1]): [float(x) for x in r[2:]] for r in rows if r[0] == "pseed"}
    got_pc = {int(r[1]): [float(x) for x in r[2:]] for r in rows if r[0] == "pconv"}
    got_s = {int(r[1]): [float(x) for x in r[2:]] for r in rows if r[0] == "sseed"}
    got_sc = {int(r[1]) for r in rows if r[0] == "sconv"}
    assert 2 not in got_p and 2 not in got_pc, "the stale seed must be erased before the update"
    conv = {int(i) for j, i in enumerate(keep) if ro["pt_status"][j] == P.abi.SEED_CONVERGED}
    assert conv and set(got_pc) == conv, (conv, set(got_pc))
    for j, i in enumerate(keep):
        stt = ro["pt_status"][j]
        if stt == P.abi.SEED_CONVERGED:
            assert i in got_pc and np.allclose(got_pc[i][:3], ro["pt_xyz_world"][j], rtol=1e-6)
        elif stt == P.abi.SEED_NAN:
            assert i not in got_p and i not in got_pc
        else:
            exp = [ro["pt_a"][j], ro["pt_b"][j], ro["pt_mu"][j], ro["pt_sigma2"][j]]
            assert i in got_p and np.allclose(got_p[i], exp, rtol=2e-3), (i, got_p.get(i), exp)
            assert np.allclose(got_p[i][2], exp[2], rtol=2e-6)
    for i in range(nseg):
        if ro["seg_status"][i] == P.abi.SEED_CONVERGED:
            assert i in got_sc
        elif ro["seg_status"][i] != P.abi.SEED_NAN:
            exp = [ro[k][i] for k in ("seg_a", "seg_b", "seg_mu_s", "seg_mu_e", "seg_sigma2_s", "seg_sigma2_e")]
            assert i in got_s and np.allclose(got_s[i], exp, rtol=2e-3), (i, got_s.get(i), exp)
    assert (ro["pt_status"] >= 2).sum() > 10


def test_adapter_under_sanitizers(P, ob, tmp_path):
    """the same self-test with the adapter's host code built -fsanitize=address,undefined (pl-svo_amd/host/Makefile): list
    walking, flattening, slot cache and write-back must be clean"""
    drv = DRIVER + "_san"
    if not os.path.exists(drv):
        pytest.skip("adapter_driver_san not built")
    W, H, nlev, maxl, minl = 320, 240, 4, 3, 1
    st, ref, cur, _ = Hh.make_case(ob, 881, W, H, 50, 20, nlev, maxl, minl)
    fr = P.synth.make_poseopt_frame(882, 80, 25, W, H)
    path = tmp_path / "in.bin"
    P.adapter_io.write_adapter_input(path, st, ref, cur, fr, nlev, maxl, minl, n_dead_seg=2)
    env = dict(os.environ)
    # protect_shadow_gap=0: the ROCm runtime maps its SVM aperture where ASan would otherwise keep a guard gap
    env["ASAN_OPTIONS"] = "detect_leaks=0:protect_shadow_gap=0:abort_on_error=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    r = subprocess.run([drv, str(path), str(tmp_path / "out.txt")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-2000:]
    r = subprocess.run([drv, "--bench", "5", str(path)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "run_us_mean" in r.stdout, r.stderr[-2000:]


def test_adapter_verbose_prints_the_reference_lines(P, ob, tmp_path):
    """verbose = true: the level banner of SparseImgAlign::run (src/sparse_img_align.cpp:88-89), vikit's per-iteration solver
    lines and the pose optimiser's (src/pose_optimizer.cpp:175-190, :252-256), reconstructed from the device trace"""
    W, H, nlev, maxl, minl = 320, 240, 4, 3, 1
    st, ref, cur, job = Hh.make_case(ob, 883, W, H, 50, 20, nlev, maxl, minl)
    fr = P.synth.make_poseopt_frame(884, 80, 25, W, H)
    path = tmp_path / "in.bin"
    P.adapter_io.write_adapter_input(path, st, ref, cur, fr, nlev, maxl, minl)
    env = dict(os.environ, PLSVO_DRIVER_VERBOSE="1")
    r = subprocess.run([DRIVER, str(path), str(tmp_path / "out.txt")], env=env, capture_output=True, text=True, timeout=120, check=True)
    out = r.stdout
    for level in (3, 2, 1):
        assert f"PYRAMID LEVEL {level}\n---------------" in out
    assert "It. 0\t Success\t new_chi2 = " in out and "n_meas = " in out and "x_norm = " in out
    assert "it 0\t Success \t new_chi2 = " in out and "norm(dT) = " in out
    assert "n deleted obs = " in out and "error init = " in out
    # one solver line per Gauss-Newton iteration the device ran
    ro, lo = ob.sparse_align(job, ref, cur, max_log=200)
    n_lines = sum(1 for l in out.splitlines() if l.startswith("It. "))
    assert abs(n_lines - len(lo)) <= 3          # the oracle's path may differ by an iteration per level (float chi2 decisions)
