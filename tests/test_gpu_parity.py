"""GPU parity tests: the HIP path (through the C ABI, libplsvo_hip.so) against the CPU oracle on the
same seeded inputs.  Bar (BASELINE.json north_star): SE(3) poses within 1e-4 rad / 1e-4 relative
translation of the CPU path; integer/byte results (half-sampler, alive/keep masks, counts) bit-exact."""
import numpy as np
import pytest

import helpers as Hh

pytestmark = pytest.mark.gpu


def _run_both(P, ob, ctx, seed, W, H, npts, nseg, nlev, maxl, minl, n_iter=30, trace=200, motion_scale=0.5):
    st, ref, cur, job = Hh.make_case(ob, seed, W, H, npts, nseg, nlev, maxl, minl, n_iter, motion_scale=motion_scale)
    res_o, log_o = ob.sparse_align(job, ref, cur, max_log=trace)
    ctx.config_pyramids(2, W, H, nlev)
    ctx.upload_pyramid(0, ref)
    ctx.upload_pyramid(1, cur)
    ctx.align_set_trace(trace)
    res_d = ctx.sparse_align(job)
    log_d = ctx.align_fetch_trace(0)
    return st, res_o, log_o, res_d, log_d


@pytest.mark.parametrize("rounding", [0, 1])
@pytest.mark.parametrize("shape", [(640, 480, 5), (1280, 720, 5), (162, 122, 3), (100, 75, 3)])
def test_halfsample_bit_exact(P, ob, gpu_ctx, rounding, shape):
    W, H, nlev = shape
    img = np.random.default_rng(W + rounding).integers(0, 256, (H, W), dtype=np.uint8)
    gpu_ctx.config_pyramids(1, W, H, nlev)
    gpu_ctx.build_pyramid(0, img, rounding)
    dev = gpu_ctx.download_pyramid(0)
    orc = ob.build_pyramid(img, nlev, rounding)
    for l, (d, o) in enumerate(zip(dev, orc)):
        assert d.shape == o.shape
        assert np.array_equal(d, o), f"level {l} differs"


CASES = [
    # tag, seed, W, H, points, segments, pyramid images, max_level, min_level
    ("tiny-points", 11, 160, 120, 24, 0, 3, 2, 0),
    ("tiny-points-lines", 12, 160, 120, 24, 10, 3, 2, 0),
    ("config1", 1234, 640, 480, 100, 0, 3, 2, 0),          # BASELINE configs[0]
    ("config2", 1235, 640, 480, 200, 80, 4, 3, 1),         # BASELINE configs[1]
    ("config3", 1236, 1280, 720, 400, 150, 5, 4, 2),       # BASELINE configs[2] (reference default levels)
    ("config3-b", 5001, 1280, 720, 400, 150, 5, 4, 2),     # a second seed set of the same workload
    # segments only.  The reference's segment objective on its own does not converge (the oracle's trajectory: one accepted
    # step, then a chi2 increase and the roll-back to the initial pose; with the full synthetic motion the first step is a
    # 0.5 rad jump and every segment is culled).  Both are paths the device must follow: the small-motion one through the
    # ordinary bar, the diverging one with the bar scaled to the size of the (meaningless) step both sides take.
    ("lines-only", 13, 640, 480, 0, 60, 3, 2, 1, 0.1),
    ("lines-only-diverging", 13, 640, 480, 0, 60, 3, 2, 1, 0.5),
    ("level0", 14, 320, 240, 60, 20, 3, 2, 0),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_sparse_align_matches_oracle(P, ob, gpu_ctx, case):
    tag, seed, W, H, npts, nseg, nlev, maxl, minl = case[:9]
    motion_scale = case[9] if len(case) > 9 else 0.5
    st, res_o, log_o, res_d, log_d = _run_both(P, ob, gpu_ctx, seed, W, H, npts, nseg, nlev, maxl, minl, motion_scale=motion_scale)
    # per-iteration linearisation while the two GN paths coincide
    n, worst = Hh.compare_align_logs(log_o, log_d)
    assert n >= 1
    # measured: <= 7e-11 on the BASELINE workloads (6e-8 in the last shared record before two paths part); segments-only inputs
    # are dominated by the per-line float sum of |res| (tree order on the device, sequential in the reference): 1e-6
    assert worst["H"] < (1e-5 if tag.startswith("lines-only") else 1e-6) and worst["Jres"] < 1e-3 and worst["chi2"] < 1e-4, worst
    # final pose: the parity bar, on the pose run() writes back (cur_frame->T_f_w_, :92)
    ang, tr, ok = Hh.pose_close(Hh.frame_pose(res_d.T, st), Hh.frame_pose(res_o.T, st))
    # a diverged alignment (the oracle itself ends > 0.1 rad from where it started): the bar scales with the step both sides took
    step_rot, step_trans = P.synth.se3_log_angle_dist(res_o.T, st.T_init)
    diverged = step_rot > 0.1
    if diverged:
        assert tag.endswith("diverging"), f"{tag}: the oracle's own alignment diverged ({step_rot:.2f} rad): not a parity case"
        assert Hh.same_path(log_o, log_d)
        assert ang <= Hh.ROT_TOL * step_rot / 0.01 and tr <= Hh.TRANS_REL_TOL * step_rot / 0.01, f"{tag}: rot {ang:.3e} rad, trans rel {tr:.3e}"
    else:
        assert ok, f"{tag}: rot {ang:.3e} rad, trans rel {tr:.3e}"
        # and on the inter-frame motion itself, T_cur_from_ref (src/sparse_img_align.cpp:80), relative to ITS translation: the
        # same bar.  (The reference's GN termination is decided by comparisons of float chi2 sums; the device takes those
        # decisions on the same float sums whenever they are close, so the two paths are the same path.)
        ang2, tr2, _ = Hh.pose_close(res_d.T, res_o.T)
        assert ang2 < 1e-4 and tr2 < 1e-4, f"{tag}: inter-frame rot {ang2:.3e} rad, trans rel {tr2:.3e}"
        if Hh.same_path(log_o, log_d):
            # same Gauss-Newton path: what is left is summation order, two orders of magnitude inside the bar
            # (1e-8 on the BASELINE workloads, 2e-7 on the 24-point 160x120 case)
            assert ang < 1e-8 and tr < 1e-7 and ang2 < 1e-8 and tr2 < 1e-5, f"{tag}: same path but rot {ang:.3e} / {ang2:.3e}, trans {tr:.3e} / {tr2:.3e}"
    # culled segments (LineFeat::feat3D = NULL) and tracked count
    assert np.array_equal(res_d.seg_alive, res_o.seg_alive)
    assert res_d.status == res_o.status
    # n_meas_ / n_tracked: equal at EVERY iteration the two paths share, the last common one included (compare_align_logs
    # asserts it record by record; n is the number of shared records) ...
    k = Hh.common_prefix(log_o, log_d) - 1
    assert k >= 0 and log_o[k]["n_meas"] == log_d[k]["n_meas"]
    # ... each side returns the n_meas_ of ITS last computeResiduals ...
    assert res_d.n_meas == log_d[-1]["n_meas"] and res_d.n_tracked == res_d.n_meas // 16
    # ... and when the two Gauss-Newton paths are the same path, so are the returned counts
    same = Hh.same_path(log_o, log_d)
    assert same, f"{tag}: the device left the oracle's Gauss-Newton path at record {Hh.common_prefix(log_o, log_d)}"
    PATH_STATS["cases"] += 1
    PATH_STATS["different_paths"] += 0 if same else 1
    if same:
        assert res_d.n_meas == res_o.n_meas and res_d.n_tracked == res_o.n_tracked
        assert res_d.iters_per_level == res_o.iters_per_level


PATH_STATS = {"cases": 0, "different_paths": 0}


SWEEPS = [
    # tag, first seed, seeds, W, H, points, segments, pyramid images, max_level, min_level, threads per frame (0 = automatic: 512 here)
    ("config2", 4000, 150, 640, 480, 200, 80, 4, 3, 1, 0),
    ("config3", 5000, 60, 1280, 720, 400, 150, 5, 4, 2, 0),
    # the benchmark's launch shape: one wave per frame, tiled pyramid mirror, chi2 terms in HBM planes kept only while the steps are small
    # (seeds 4300..4449: the window holds seed 4373, whose near tie falls on an iteration whose terms were not kept)
    ("config2-one-wave-per-frame", 4300, 150, 640, 480, 200, 80, 4, 3, 1, 64),
    # config 3 at ITS benchmark shape (8192 streams -> two waves per frame: row-major slab, HBM planes, arming); the window holds seed 5348
    ("config3-two-waves-per-frame", 5320, 40, 1280, 720, 400, 150, 5, 4, 2, 128),
    # config 3 at the shape `bench.py --config 3` runs since round 4 (16384 streams -> ONE wave per frame: tiled mirror, whose level 4 is
    # 80 x 45 pixels = a partial tile row, HBM planes, arming); same window
    ("config3-one-wave-per-frame", 5320, 40, 1280, 720, 400, 150, 5, 4, 2, 64),
    # SURVEY.md 8(d)'s FULL inter-frame motion (motion_scale = 1.0; every other BASELINE-sized case runs half of it, tests/helpers.py) at the
    # benchmark's launch shape: the coarse levels start two to three times further from the optimum, most segments fail the `res_ >= 200`
    # / out-of-image test of src/sparse_img_align.cpp:588-594, 648 at the first coarse iterations and are culled for good (:687-688) -- the
    # mass-cull path at size -- and the Gauss-Newton paths are longer
    ("config2-full-motion-one-wave-per-frame", 6000, 40, 640, 480, 200, 80, 4, 3, 1, 64, 1.0),
]


TIE_CASES = [
    # seed, threads per frame, (W, H, points, segments, pyramid images, max_level, min_level): frames on which a near tie of the solver's
    # `new_chi2 > chi2_` decision falls on an iteration whose per-pixel chi2 terms were not kept (HBM planes are written only while the
    # steps are small).  Round 3's build decided those on the rounded-once sums and left the oracle's path: 1.6e-3 / 9.5e-3 of the
    # inter-frame translation, 1.9e-4 / 3.2e-4 on T_f_w -- outside the bar.  The kernel now rebuilds the missing terms before deciding.
    (4373, 64, (640, 480, 200, 80, 4, 3, 1)),
    (5348, 128, (1280, 720, 400, 150, 5, 4, 2)),
    (5348, 64, (1280, 720, 400, 150, 5, 4, 2)),     # the same frame at one wave per frame (config 3's benchmark shape since round 4)
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", TIE_CASES, ids=[f"seed{c[0]}-{c[1]}threads" for c in TIE_CASES])
def test_near_tie_on_an_unarmed_iteration_follows_the_oracle(P, ob, gpu_ctx, case):
    """The two frames the 1000-seed sweeps found at the launch shapes the benchmark runs (config 2 at one wave per frame, config 3 at two):
    same Gauss-Newton path as the oracle, record for record, the bar of BASELINE.json on T_f_w AND on T_cur_from_ref, and no near tie
    decided without the reference's float sums (src/sparse_img_align.cpp:171,192,484,683 + the vikit loop)."""
    seed, threads, (W, H, npts, nseg, nlev, maxl, minl) = case
    st, ref, cur, job = Hh.make_case(ob, seed, W, H, npts, nseg, nlev, maxl, minl)
    res_o, log_o = ob.sparse_align(job, ref, cur, max_log=200)
    gpu_ctx.set_launch_shapes(align_threads=threads)
    try:
        gpu_ctx.config_pyramids(2, W, H, nlev)
        gpu_ctx.upload_pyramid(0, ref)
        gpu_ctx.upload_pyramid(1, cur)
        gpu_ctx.align_set_trace(200)
        res_d = gpu_ctx.sparse_align(job)
        log_d = gpu_ctx.align_fetch_trace(0)
        iters, ties, unarmed = gpu_ctx.align_chi2_ties()
    finally:
        gpu_ctx.set_launch_shapes(align_threads=0)
    assert Hh.same_path(log_o, log_d), (res_o.iters_per_level[:5], res_d.iters_per_level[:5])
    assert unarmed == 0 and ties >= 2            # both frames have a near tie on an armed AND on an unarmed iteration
    ang, tr, ok = Hh.pose_close(Hh.frame_pose(res_d.T, st), Hh.frame_pose(res_o.T, st))
    ang2, tr2, ok2 = Hh.pose_close(res_d.T, res_o.T)
    assert ok and ok2 and ang2 < 1e-8 and tr2 < 1e-7, (ang, tr, ang2, tr2)
    assert res_d.n_meas == res_o.n_meas and res_d.iters_per_level == res_o.iters_per_level and np.array_equal(res_d.seg_alive, res_o.seg_alive)


@pytest.mark.parametrize("sweep", SWEEPS, ids=[c[0] for c in SWEEPS])
def test_seed_sweep_follows_the_oracle_path_on_every_seed(P, ob, gpu_ctx, sweep):
    """210 seeds of BASELINE configs 2 and 3.  On EVERY seed the parity bar of BASELINE.json holds on cur_frame->T_f_w_ AND on
    T_cur_from_ref itself: 1e-4 rad, 1e-4 of the inter-frame translation.  The device takes the oracle's Gauss-Newton path record for
    record (same iterations, same accept / roll-back decisions) because the chi2 comparison is taken on the reference's own sequential
    float sums whenever the two values are close (align_kernels.hip::exact_chi2_pair) -- except where the reference's decision hangs on
    the LAST bits of its input: the device's pose differs from the oracle's by ~1e-11 (summation order of H in double), which now and
    then moves the float pixel position of one patch by one ulp and chi2 by a few ulps; a comparison closer than that (<= 8 float ulps
    between the two chi2 values, or ||x||_inf within 10 % of eps) may still go the other way.  Measured: 3 of 210 seeds (round 2, when
    the comparison was made on exactly-rounded sums: 6 of 40).  Worst cases go to gpurun_out/ for profiles/."""
    import json, os
    tag, seed0, n_seeds, W, H, npts, nseg, nlev, maxl, minl, threads = sweep[:11]
    motion = sweep[11] if len(sweep) > 11 else 0.5
    # The driver's `-m gpu` run has a wall-clock limit (704 s of 1200): the 150-seed windows run their first 60 seeds there
    # and all of them under PLSVO_SWEEP_FULL=1 (the builder's own GPU calls, tools/verify_round.sh); PLSVO_SWEEP_SEEDS fixes any count
    # (quick checks of a kernel change: fewer seeds; long emulated runs: more)
    if n_seeds > 60 and os.environ.get("PLSVO_SWEEP_FULL") != "1":
        n_seeds = 60
    n_seeds = int(os.environ.get("PLSVO_SWEEP_SEEDS", n_seeds))
    gpu_ctx.set_launch_shapes(align_threads=threads)
    try:
        _seed_sweep_body(P, ob, gpu_ctx, tag, seed0, n_seeds, W, H, npts, nseg, nlev, maxl, minl, motion)
    finally:
        gpu_ctx.set_launch_shapes(align_threads=0)


def _seed_sweep_body(P, ob, gpu_ctx, tag, seed0, n_seeds, W, H, npts, nseg, nlev, maxl, minl, motion=0.5):
    import json, os
    worst = {"rot_rad": 0.0, "trans_rel": 0.0, "inter_rot_rad": 0.0, "inter_trans_rel": 0.0, "inter_trans_abs_m": 0.0, "min_inter_translation_m": 1e9}
    different, failures, iters_d, iters_o = [], [], 0, 0
    worst_lin = {"H": 0.0, "x": 0.0, "chi2": 0.0}
    chunk = 30
    ties = its = unarmed = 0
    for c0 in range(0, n_seeds, chunk):
        seeds = list(range(seed0 + c0, seed0 + min(c0 + chunk, n_seeds)))
        cases = [Hh.make_case(ob, sd, W, H, npts, nseg, nlev, maxl, minl, motion_scale=motion) for sd in seeds]
        gpu_ctx.config_pyramids(2 * len(seeds), W, H, nlev)
        jobs = []
        for k, (st, ref, cur, job) in enumerate(cases):
            gpu_ctx.upload_pyramid(2 * k, ref)
            gpu_ctx.upload_pyramid(2 * k + 1, cur)
            jobs.append(P.align_job_from_stream(st, maxl, minl, ref_slot=2 * k, cur_slot=2 * k + 1))
        gpu_ctx.align_set_trace(200)
        res_dev = gpu_ctx.sparse_align_batch(jobs)
        a, b, c_ = gpu_ctx.align_chi2_ties()
        its += a; ties += b; unarmed += c_
        for k, seed in enumerate(seeds):
            st, ref, cur, job = cases[k]
            res_o, log_o = ob.sparse_align(job, ref, cur, max_log=200)
            res_d, log_d = res_dev[k], gpu_ctx.align_fetch_trace(k)
            iters_d += len(log_d); iters_o += len(log_o)
            ang, tr, ok = Hh.pose_close(Hh.frame_pose(res_d.T, st), Hh.frame_pose(res_o.T, st))
            t_inter = float(np.linalg.norm(np.asarray(res_o.T)[4:]))
            ang2, dist2 = P.synth.se3_log_angle_dist(np.asarray(res_d.T), np.asarray(res_o.T))
            tr2 = dist2 / max(t_inter, 1e-3)
            worst["min_inter_translation_m"] = min(worst["min_inter_translation_m"], t_inter)
            for key, val in (("rot_rad", ang), ("trans_rel", tr), ("inter_rot_rad", ang2), ("inter_trans_rel", tr2), ("inter_trans_abs_m", dist2)):
                if val > worst[key]:
                    worst[key], worst["seed_" + key] = val, seed
            if not (ok and ang2 <= Hh.ROT_TOL and tr2 <= Hh.TRANS_REL_TOL):
                failures.append({"seed": seed, "rot": ang, "trans_rel": tr, "inter_rot": ang2, "inter_trans_rel": tr2})
            if not np.array_equal(res_d.seg_alive, res_o.seg_alive):
                failures.append({"seed": seed, "what": "seg_alive differs"})
            n, w = Hh.compare_align_logs(log_o, log_d)      # asserts n_meas equality on every shared iteration
            worst_lin["H"] = max(worst_lin["H"], float(w["H"])); worst_lin["x"] = max(worst_lin["x"], float(w["x"]))
            worst_lin["chi2"] = max(worst_lin["chi2"], float(w["chi2"]))
            # (1e-11 while the poses are equal to rounding; a few 1e-6 once one patch position has moved by a float ulp)
            if not (n >= 1 and w["H"] < 1e-4):
                failures.append({"seed": seed, "what": "linearisation differs on the shared records", "worst": {k_: float(v_) for k_, v_ in w.items()}})
            if Hh.same_path(log_o, log_d):
                assert res_d.n_meas == res_o.n_meas and res_d.n_tracked == res_o.n_tracked, seed
                assert res_d.iters_per_level == res_o.iters_per_level, seed
            else:
                k_ = Hh.common_prefix(log_o, log_d) - 1
                ra, rb = log_o[k_], log_d[k_]
                info = {"seed": seed, "oracle_iters": res_o.iters_per_level[:5], "device_iters": res_d.iters_per_level[:5],
                        "shared_records": k_ + 1, "level": ra["level"], "iter": ra["iter"], "accepted": [ra["accepted"], rb["accepted"]],
                        "new_chi2": [ra["new_chi2"], rb["new_chi2"]],
                        "x_norm": [float(np.max(np.abs(ra["x"]))), float(np.max(np.abs(rb["x"])))]}
                if k_ >= 1:
                    info["prev_chi2"] = [log_o[k_ - 1]["new_chi2"], log_d[k_ - 1]["new_chi2"]]
                # the two paths may part only on a decision that hangs on the last bits (see the docstring)
                if ra["accepted"] != rb["accepted"]:
                    gaps = [abs(info["new_chi2"][i] - info["prev_chi2"][i]) / info["prev_chi2"][i] for i in range(2)] if k_ >= 1 else [1.0, 1.0]
                    # (one patch whose float pixel position moves by an ulp shifts chi2 by a few float ulps -- up to ~30 for a strong gradient
                    #  under a large residual; the sweeps of rounds 3 and 4 met 1.1 ... 4.7 ulps on the MI355X and the emulated device)
                    info["kind"], info["chi2_gap_rel"] = "chi2 within 8 float ulps", gaps
                    if not max(gaps) <= 8 * 1.2e-7:
                        failures.append({"seed": seed, "what": "paths part on a chi2 comparison that is not a last-bit tie", "info": info})
                else:
                    # (the step at convergence is the quotient of two nearly cancelling sums: once one patch position has moved by a float ulp
                    #  it differs by a few per cent between the two paths.  Largest over 1000 emulated seeds: 8 %; seed 4413 on the MI355X: 8 %)
                    info["kind"] = "||x|| within 10 % of eps"
                    if not all(abs(v - 1e-6) < 1e-7 for v in info["x_norm"]):
                        failures.append({"seed": seed, "what": "paths part without a near tie of either stopping rule", "info": info})
                different.append(info)
    out = {"what": f"{n_seeds} seeds ({seed0}..{seed0 + n_seeds - 1}) of BASELINE {tag} ({W}x{H}, {npts} points + {nseg} segments, levels {maxl}..{minl}): "
                   "HIP path vs CPU oracle",
           "bar": {"rot_rad": Hh.ROT_TOL, "trans_rel": Hh.TRANS_REL_TOL,
                   "applies_to": "cur_frame->T_f_w_ (relative to |t|) and T_cur_from_ref (relative to the inter-frame translation)"},
           "worst": worst, "worst_per_record_while_paths_coincide": worst_lin, "seeds_with_different_gn_path": len(different), "different": different, "outside_the_bar": failures,
           "gn_iterations_device": iters_d, "gn_iterations_oracle": iters_o,
           "iterations_decided_on_exact_float_chi2": ties, "near_ties_without_kept_terms": unarmed,
           "iterations_counted_by_the_device": its}
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(root, "gpurun_out", f"parity_seed_sweep_{tag}.json"), "w"), indent=1)
    print(json.dumps(out))
    assert not failures, failures
    assert unarmed == 0, unarmed      # every near tie was decided on the reference's float sums (missing terms are rebuilt first)
    # measured on the MI355X, rounds 3-5: 3 / 2 of 150 (config 2, small-batch / one-wave shape), 1 of 60 and 0 of 40 (config 3); 1000 emulated
    # seeds: 17-23.  The allowance does not grow with what a rewrite of the kernel happens to meet: 2 % of the seeds, at least 3
    assert len(different) <= max(3, n_seeds // 50), different


def test_device_trace_against_scipy(P, ob, gpu_ctx):
    """Independent pins of the device's own solve and update, through its trace: every accepted step satisfies
    H x = Jres (numpy LU) and T_after = T_before * expm(hat(-x)) (scipy's matrix exponential); the pose optimiser's
    A dT = b and T_after = expm(hat(dT)) * T_before likewise."""
    from scipy.linalg import expm
    st, res_o, log_o, res_d, log_d = _run_both(P, ob, gpu_ctx, 1235, 640, 480, 200, 80, 4, 3, 1)
    T_prev = T_old = Hh.se3_matrix4(st.T_init)     # model_, old_model_ of the vikit solver
    checked = rolled_back = 0
    for r in log_d:
        H, g, x = np.asarray(r["H"], float).reshape(6, 6), np.asarray(r["Jres"], float), np.asarray(r["x"], float)
        x_ref = np.linalg.solve(H, g)
        assert np.linalg.norm(x - x_ref) <= 1e3 * np.linalg.cond(H) * 2.2e-16 * np.linalg.norm(x_ref) + 1e-18
        assert np.linalg.norm(H @ x - g) <= 1e-11 * np.linalg.norm(g) + 1e-300
        T_new = Hh.se3_matrix4(r["T_after"])
        if r["accepted"]:
            assert np.allclose(T_new, T_prev @ expm(Hh.hat6(-x)), atol=2e-15, rtol=0)
            T_old, T_prev = T_prev, T_new
            checked += 1
        else:                                       # chi2 went up: model_ = old_model_ (the step before is undone), level ends
            assert np.array_equal(T_new, T_old)
            T_prev = T_old
            rolled_back += 1
    assert checked >= 5 and rolled_back >= 1
    fr = P.synth.make_poseopt_frame(77, 500, 200)
    job = P.poseopt_job_from_frame(fr)
    gpu_ctx.poseopt_set_trace(40)
    gpu_ctx.pose_optimize(job)
    ld = gpu_ctx.poseopt_fetch_trace(0)
    T_prev = T_old = Hh.se3_matrix4(fr.T_init)
    for r in ld:
        A, b, dT = np.asarray(r["A"], float).reshape(6, 6), np.asarray(r["b"], float), np.asarray(r["dT"], float)
        assert np.linalg.norm(A @ dT - b) <= 1e-11 * np.linalg.norm(b) + 1e-300
        T_new = Hh.se3_matrix4(r["T_after"])
        if r["accepted"]:
            assert np.allclose(T_new, expm(Hh.hat6(dT)) @ T_prev, atol=2e-15, rtol=0)
            T_old, T_prev = T_prev, T_new
        else:
            assert np.array_equal(T_new, T_old)
            T_prev = T_old


def test_sparse_align_long_lines_two_pass_levels(P, ob, gpu_ctx):
    """segments with more than 64 samples at a level (a 1150..1800 px line on a 1920x1080 frame: N = length / 16..23) do not fit one
    wave-round: the kernel runs such a level in two passes (residual sums, workgroup barrier, expansion) -- same results, same parity bar"""
    st, ref, cur, job = Hh.make_case(ob, 41, 1920, 1080, 60, 14, 3, 1, 0, seg_len_range=(1150.0, 1800.0))
    n0 = [ob.setup_sampling(s_, e_, L)[0] for s_, e_, L in zip(st.seg_spx, st.seg_epx, st.seg_len)]   # samples at level 0
    assert max(n0) > 64 and min(n0) <= 128, "the case must contain a line with more than 64 samples at level 0 (two-pass) and a level without one"
    assert P.capi.align_slot_layout(job, 0)[3] and not P.capi.align_slot_layout(job, 1)[3]
    res_o, log_o = ob.sparse_align(job, ref, cur, max_log=200)
    gpu_ctx.config_pyramids(2, 1920, 1080, 3)
    gpu_ctx.upload_pyramid(0, ref)
    gpu_ctx.upload_pyramid(1, cur)
    gpu_ctx.align_set_trace(200)
    res_d = gpu_ctx.sparse_align(job)
    log_d = gpu_ctx.align_fetch_trace(0)
    n, worst = Hh.compare_align_logs(log_o, log_d)
    assert n >= 1 and worst["H"] < 2e-5 and worst["Jres"] < 1e-3 and worst["chi2"] < 1e-4, worst
    ang, tr, ok = Hh.pose_close(Hh.frame_pose(res_d.T, st), Hh.frame_pose(res_o.T, st))
    assert ok, (ang, tr)
    assert np.array_equal(res_d.seg_alive, res_o.seg_alive)
    # the same frame through a batch (other launch shape) must agree with the single call bit for bit
    gpu_ctx.align_set_trace(0)
    batch = gpu_ctx.sparse_align_batch([job, job])
    single = gpu_ctx.sparse_align(job)
    assert np.array_equal(batch[0].T, single.T) and np.array_equal(batch[1].T, single.T)


def test_sparse_align_single_linearisation(P, ob, gpu_ctx):
    """one computeResiduals at a given pose (n_iter = 1): H, Jres, chi2, n_meas against the oracle"""
    st, res_o, log_o, res_d, log_d = _run_both(P, ob, gpu_ctx, 21, 640, 480, 200, 80, 4, 2, 2, n_iter=1)
    assert len(log_o) == 1 and len(log_d) == 1
    a, b = log_o[0], log_d[0]
    assert a["n_meas"] == b["n_meas"]
    # same pose on both sides: the only differences are the summation order of the per-line mean |res|
    # (float) and of chi2 (the oracle sums thousands of floats sequentially)
    assert Hh.rel(b["H"], a["H"]) < 1e-9          # (measured 1e-12 ... 6e-12: the float residuals and gradients are the oracle's bit for bit)
    assert Hh.rel(b["Jres"], a["Jres"]) < 1e-7
    assert abs(a["new_chi2"] - b["new_chi2"]) <= 5e-6 * abs(a["new_chi2"])
    assert np.max(np.abs(a["x"] - b["x"])) < 1e-8


def test_sparse_align_edge_cases(P, ob, gpu_ctx):
    # no features at all: run() returns 0 and leaves the pose alone (src/sparse_img_align.cpp:58-62)
    st, ref, cur, job = Hh.make_case(ob, 31, 160, 120, 0, 0, 3, 2, 0)
    gpu_ctx.config_pyramids(2, 160, 120, 3)
    gpu_ctx.upload_pyramid(0, ref)
    gpu_ctx.upload_pyramid(1, cur)
    gpu_ctx.align_set_trace(0)
    r = gpu_ctx.sparse_align(job)
    assert r.n_tracked == 0 and np.array_equal(r.T, job.c.T_cur_from_ref[:])
    # features present but none visible (all in the 3-pixel border at every level): zero measurements
    st, ref, cur, job = Hh.make_case(ob, 32, 160, 120, 8, 0, 3, 2, 1)
    job.pt_px[:] = [[1.0, 1.0]] * 8
    ro, _ = ob.sparse_align(job, ref, cur)
    rd = gpu_ctx.sparse_align(job)
    assert ro.n_meas == 0 and rd.n_meas == 0
    assert Hh.pose_close(rd.T, ro.T)[2]
    # segments dead on entry stay dead and contribute nothing
    st, ref, cur, job = Hh.make_case(ob, 33, 320, 240, 40, 12, 3, 2, 1)
    alive_in = np.ones(12, np.uint8)
    alive_in[::2] = 0
    job2 = P.abi.AlignJob(st.cam, 2, 1, 30, 1e-6, st.T_init, st.pt_px, st.pt_xyz_ref, st.seg_spx, st.seg_epx, st.seg_len,
                          st.seg_p_ref, st.seg_q_ref, seg_alive_in=alive_in)
    gpu_ctx.config_pyramids(2, 320, 240, 3)
    gpu_ctx.upload_pyramid(0, ref)
    gpu_ctx.upload_pyramid(1, cur)
    ro, _ = ob.sparse_align(job2, ref, cur)
    rd = gpu_ctx.sparse_align(job2)
    assert np.array_equal(rd.seg_alive, ro.seg_alive) and not rd.seg_alive[::2].any()
    assert Hh.pose_close(rd.T, ro.T)[2]
    # a segment sample leaving the current image culls the whole line (:588-594)
    # and large residuals (mean |res| >= 200/16 per pixel) cull it too (:648)
    st, ref, cur, job = Hh.make_case(ob, 34, 320, 240, 40, 12, 3, 2, 1, motion_scale=4.0)
    gpu_ctx.upload_pyramid(0, ref)
    gpu_ctx.upload_pyramid(1, cur)
    ro, _ = ob.sparse_align(job, ref, cur)
    rd = gpu_ctx.sparse_align(job)
    assert not ro.seg_alive.all()
    assert np.array_equal(rd.seg_alive, ro.seg_alive)


@pytest.mark.parametrize("npts", [1, 2, 3])
def test_sparse_align_fewer_patches_than_unknowns(P, ob, gpu_ctx, npts):
    """a 4x4 patch constrains two directions of the pose (its 16 Jacobian rows span r0, r1 of the 2x6 projection Jacobian): one or
    two point patches give rank-2 / rank-4 normal equations.
      * one patch: Eigen's LDLT (3.2 rule, fixed pivot order -- plsvo_wave.hpp) returns zero components for the four directions it
        cannot see; the device must return the same components and walk the same path;
      * two patches: H is a double sum of 32 rank-1 terms, its rounding residue outside the rank-4 span is ~eps * max -- right AT
        Eigen's cutoff -- so the reference's own step is rounding noise divided by rounding noise under either release's rule
        (the oracle shows 5 or 6 non-zero components from iteration to iteration).  Checked: same linearisation, finite result;
      * three patches: full rank, ordinary parity."""
    st, ref, cur, job = Hh.make_case(ob, 35, 320, 240, npts, 0, 3, 2, 0)
    ro, lo = ob.sparse_align(job, ref, cur, max_log=120)
    gpu_ctx.config_pyramids(2, 320, 240, 3)
    gpu_ctx.upload_pyramid(0, ref)
    gpu_ctx.upload_pyramid(1, cur)
    gpu_ctx.align_set_trace(120)
    rd = gpu_ctx.sparse_align(job)
    ld = gpu_ctx.align_fetch_trace(0)
    assert np.all(np.isfinite(rd.T)) and len(ld) >= 1
    a, b = lo[0], ld[0]
    assert a["n_meas"] == b["n_meas"] == 16 * npts
    assert Hh.rel(b["H"], a["H"]) < 1e-9 and Hh.rel(b["Jres"], a["Jres"]) < 1e-7
    if npts == 2:
        return
    n = Hh.common_prefix(lo, ld)
    assert n >= 2
    for a, b in list(zip(lo, ld))[:n]:
        assert a["n_meas"] == b["n_meas"]
        if npts == 1:
            assert np.count_nonzero(a["x"]) <= 2, a["x"]
        assert np.array_equal(a["x"] == 0.0, b["x"] == 0.0), (a["x"], b["x"])
        assert np.allclose(b["x"], a["x"], rtol=1e-3, atol=1e-4 * float(np.max(np.abs(lo[0]["x"])))), (a["x"], b["x"])
    if Hh.same_path(lo, ld):
        assert Hh.pose_close(rd.T, ro.T)[2], Hh.pose_close(rd.T, ro.T)
        assert rd.n_meas == ro.n_meas


def _adversarial_jobs(P, st):
    """(tag, job, images identical?) -- inputs on which the reference's arithmetic leaves the well-behaved range"""
    I = np.array([0, 0, 0, 1, 0, 0, 0.0])

    def job_of(max_level=2, min_level=0, n_iter=30, **over):
        a = dict(pt_px=st.pt_px, pt_xyz_ref=st.pt_xyz_ref, seg_spx=st.seg_spx, seg_epx=st.seg_epx, seg_len=st.seg_len,
                 seg_p_ref=st.seg_p_ref, seg_q_ref=st.seg_q_ref, T_init=st.T_init)
        a.update(over)
        return P.abi.AlignJob(st.cam, max_level, min_level, n_iter, 1e-6, a["T_init"], a["pt_px"], a["pt_xyz_ref"], a["seg_spx"], a["seg_epx"],
                              a["seg_len"], a["seg_p_ref"], a["seg_q_ref"])
    out = []
    # a static camera: cur == ref and T = I.  Every residual is exactly 0, a line's mean |res| is 0 and `H += H_line * w / res_`
    # (:681) divides by it: H becomes inf/NaN, x[0] is NaN, the solver sets stop_ (:700) and leaves the pose alone
    out.append(("static-camera", job_of(T_init=I), True))
    out.append(("static-camera-points-only", P.abi.AlignJob(st.cam, 2, 0, 30, 1e-6, I, st.pt_px, st.pt_xyz_ref, st.seg_spx[:0], st.seg_epx[:0],
                                                            st.seg_len[:0], st.seg_p_ref[:0], st.seg_q_ref[:0]), True))
    x = st.pt_xyz_ref.copy(); x[:5, 2] *= -1
    out.append(("points-behind-the-camera", job_of(pt_xyz_ref=x), False))
    x = st.pt_xyz_ref.copy(); x[:5] = 0
    out.append(("zero-depth-points", job_of(pt_xyz_ref=x), False))
    x = st.pt_xyz_ref.copy(); x[3, 1] = np.nan
    out.append(("nan-point", job_of(pt_xyz_ref=x), False))
    T = st.T_init.copy(); T[5] = np.nan
    out.append(("nan-pose", job_of(T_init=T), False))
    e = st.seg_epx.copy(); e[:3] = st.seg_spx[:3]
    L = st.seg_len.copy(); L[:3] = 0
    out.append(("zero-length-segments", job_of(seg_epx=e, seg_len=L), False))
    e = st.seg_epx.copy(); e[:3] = st.seg_spx[:3] + [2.0, 1.0]
    out.append(("three-pixel-segments", job_of(seg_epx=e, seg_len=np.linalg.norm(e - st.seg_spx, axis=1)), False))
    out.append(("no-iterations", job_of(n_iter=0), False))
    out.append(("one-iteration", job_of(n_iter=1), False))
    out.append(("level-0-only", job_of(max_level=0, min_level=0), False))
    T = st.T_init.copy(); T[4:] += [3.0, 0, 0]
    out.append(("start-3-m-off", job_of(T_init=T), False))
    p = st.pt_px.copy(); p[:4] = [[-50, 20], [400, 100], [100, -3], [100, 900]]
    out.append(("feature-pixels-outside-the-image", job_of(pt_px=p), False))
    return out


def test_sparse_align_adversarial_inputs(P, ob, gpu_ctx):
    """inputs that drive the reference's arithmetic through inf / NaN / empty sets: the device must take the same decisions (stop_
    flag path, culls, measurement counts) and return the same pose -- NaN where the reference returns NaN"""
    st, ref, cur, _ = Hh.make_case(ob, 36, 320, 240, 40, 12, 3, 2, 0)
    gpu_ctx.config_pyramids(2, 320, 240, 3)
    gpu_ctx.upload_pyramid(0, ref)
    for tag, job, static in _adversarial_jobs(P, st):
        gpu_ctx.upload_pyramid(1, ref if static else cur)
        ro, lo = ob.sparse_align(job, ref, ref if static else cur, max_log=200)
        gpu_ctx.align_set_trace(200)
        rd = gpu_ctx.sparse_align(job)
        ld = gpu_ctx.align_fetch_trace(0)
        To, Td = np.asarray(ro.T, float), np.asarray(rd.T, float)
        assert np.array_equal(np.isnan(To), np.isnan(Td)), (tag, To, Td)
        assert np.array_equal(rd.seg_alive, ro.seg_alive), tag
        n = Hh.common_prefix(lo, ld)
        assert n == min(len(lo), len(ld)) or n >= 1, (tag, n, len(lo), len(ld))
        for a, b in list(zip(lo, ld))[:n]:
            assert a["n_meas"] == b["n_meas"] and a["accepted"] == b["accepted"], (tag, a["level"], a["iter"])
            assert np.array_equal(np.isnan(a["x"]), np.isnan(b["x"])) or not a["accepted"], (tag, a["x"], b["x"])
        if Hh.same_path(lo, ld):
            assert (rd.n_meas, rd.n_tracked, rd.iters_per_level) == (ro.n_meas, ro.n_tracked, ro.iters_per_level), tag
            ok = ~np.isnan(To)
            if ok.all():
                assert Hh.pose_close(Td, To)[2], (tag, Hh.pose_close(Td, To))
        else:
            assert tag not in ("static-camera", "static-camera-points-only", "nan-pose", "no-iterations", "one-iteration"), tag
            assert Hh.pose_close(Td, To)[2], (tag, Hh.pose_close(Td, To))


def test_sparse_align_border_features_leave_holes_in_the_slot_table(P, ob, gpu_ctx):
    """points / segment end points inside the 3-pixel border of a COARSE level but not of a fine one (`:216-219`, `:299-301`): their
    slots are holes at the coarse levels and live at the fine ones; points that project outside the current image; segments
    without a landmark on entry.  Per-iteration n_meas, culls and the pose must follow the oracle through all of it."""
    W, H = 640, 480
    st = P.synth.make_align_stream(61, W, H, 120, 40, max_level=3)
    imgs = P.synth.render_streams([st]).numpy()
    ref, cur = ob.build_pyramid(imgs[0, 0], 4), ob.build_pyramid(imgs[0, 1], 4)
    px = st.pt_px.copy()
    rng = np.random.default_rng(5)
    # 3 px at level 3 = 24 px at level 0, 12 px at level 2, 6 px at level 1: spread 40 points over those bands, all four sides
    for i in range(40):
        band = (4.0, 7.0, 13.0, 25.0)[i % 4] + rng.uniform(0.0, 2.0)
        side = (i // 4) % 4
        if side == 0: px[i] = [band, rng.uniform(40, H - 40)]
        elif side == 1: px[i] = [W - 1 - band, rng.uniform(40, H - 40)]
        elif side == 2: px[i] = [rng.uniform(40, W - 40), band]
        else: px[i] = [rng.uniform(40, W - 40), H - 1 - band]
    spx, epx = st.seg_spx.copy(), st.seg_epx.copy()
    for s in range(8):                                  # segment end points in the same bands
        spx[s] = [(5.0, 9.0, 15.0, 27.0)[s % 4], 100.0 + 20 * s]
    alive_in = np.ones(40, np.uint8)
    alive_in[[3, 17, 39]] = 0
    job = P.abi.AlignJob(st.cam, 3, 1, 30, 1e-6, st.T_init, px, st.pt_xyz_ref, spx, epx, np.linalg.norm(epx - spx, axis=1),
                         st.seg_p_ref, st.seg_q_ref, seg_alive_in=alive_in)
    ro, lo = ob.sparse_align(job, ref, cur, max_log=200)
    gpu_ctx.config_pyramids(2, W, H, 4)
    gpu_ctx.upload_pyramid(0, ref)
    gpu_ctx.upload_pyramid(1, cur)
    gpu_ctx.align_set_trace(200)
    rd = gpu_ctx.sparse_align(job)
    ld = gpu_ctx.align_fetch_trace(0)
    n, worst = Hh.compare_align_logs(lo, ld)            # asserts n_meas equality per shared iteration
    assert n >= 3 and worst["H"] < 1e-6 and worst["chi2"] < 1e-4, worst
    per_level = {r["level"]: r["n_meas"] for r in lo}
    assert per_level[3] < per_level[2] < per_level[1], "the case must gain measurements from level to level (border bands)"
    assert np.array_equal(rd.seg_alive, ro.seg_alive) and not rd.seg_alive[[3, 17, 39]].any()
    assert Hh.pose_close(Hh.frame_pose(rd.T, st), Hh.frame_pose(ro.T, st))[2]
    if Hh.same_path(lo, ld):
        assert rd.n_meas == ro.n_meas


def test_sparse_align_batch_equals_single(P, ob, gpu_ctx):
    """streams are independent: a batch must reproduce the single-job results bit for bit"""
    B, W, H = 6, 320, 240
    streams = [P.synth.make_align_stream(500 + i, W, H, 50 + 5 * i, 10 + i, max_level=3) for i in range(B)]
    imgs = P.synth.render_streams(streams).numpy()
    gpu_ctx.config_pyramids(2 * B, W, H, 4)
    for i in range(B):
        gpu_ctx.build_pyramid(2 * i, imgs[i, 0], 0)
        gpu_ctx.build_pyramid(2 * i + 1, imgs[i, 1], 0)
    jobs = [P.align_job_from_stream(s, 3, 1, ref_slot=2 * i, cur_slot=2 * i + 1) for i, s in enumerate(streams)]
    gpu_ctx.align_set_trace(0)
    batch = gpu_ctx.sparse_align_batch(jobs)
    for i, j in enumerate(jobs):
        single = gpu_ctx.sparse_align(j)
        assert np.array_equal(single.T, batch[i].T)
        assert single.n_meas == batch[i].n_meas and single.iters_per_level == batch[i].iters_per_level
        assert np.array_equal(single.seg_alive, batch[i].seg_alive)
        ref = gpu_ctx.download_pyramid(2 * i)
        cur = gpu_ctx.download_pyramid(2 * i + 1)
        ro, _ = ob.sparse_align(j, ref, cur)
        assert Hh.pose_close(batch[i].T, ro.T)[2]


def test_sparse_align_mixed_batch_with_degenerate_jobs(P, ob, gpu_ctx):
    """a batch that mixes ordinary jobs with a job without features, a one-patch job (rank-2 H), a static-camera job (infinite H,
    stop_ path) and a job whose features are all invisible: every job's result equals its single-job result bit for bit"""
    W, H = 320, 240
    streams = [P.synth.make_align_stream(520 + i, W, H, 40, 10, max_level=2) for i in range(3)]
    imgs = P.synth.render_streams(streams).numpy()
    gpu_ctx.config_pyramids(6, W, H, 3)
    for i in range(3):
        gpu_ctx.build_pyramid(2 * i, imgs[i, 0], 0)
        gpu_ctx.build_pyramid(2 * i + 1, imgs[i, 1], 0)
    st = streams[0]
    I = np.array([0, 0, 0, 1, 0, 0, 0.0])
    none2, none3, none1 = np.zeros((0, 2)), np.zeros((0, 3)), np.zeros(0)

    def job(s, ref_slot, cur_slot, T=None, pts=slice(None), segs=slice(None), pt_px=None):
        return P.abi.AlignJob(s.cam, 2, 0, 30, 1e-6, s.T_init if T is None else T, s.pt_px[pts] if pt_px is None else pt_px, s.pt_xyz_ref[pts],
                              s.seg_spx[segs], s.seg_epx[segs], s.seg_len[segs], s.seg_p_ref[segs], s.seg_q_ref[segs],
                              ref_slot=ref_slot, cur_slot=cur_slot)
    jobs = [job(streams[0], 0, 1),
            job(st, 0, 1, pts=slice(0, 0), segs=slice(0, 0)),               # no features
            job(streams[1], 2, 3),
            job(st, 0, 1, pts=slice(0, 1), segs=slice(0, 0)),               # one patch
            job(st, 0, 0, T=I),                                             # static camera: cur slot == ref slot
            job(st, 0, 1, segs=slice(0, 0), pt_px=np.full((40, 2), 1.0)),   # every point inside the border: no measurement
            job(streams[2], 4, 5)]
    gpu_ctx.align_set_trace(0)
    batch = gpu_ctx.sparse_align_batch(jobs)
    for i, j in enumerate(jobs):
        single = gpu_ctx.sparse_align(j)
        assert np.array_equal(single.T, batch[i].T, equal_nan=True), i
        assert single.n_meas == batch[i].n_meas and single.iters_per_level == batch[i].iters_per_level, i
        assert np.array_equal(single.seg_alive, batch[i].seg_alive), i
    assert batch[1].n_meas == 0 and np.array_equal(batch[1].T, st.T_init)
    assert batch[5].n_meas == 0
    assert np.array_equal(batch[4].T, I)                                   # stop_ at the first iteration of every level: pose untouched


def test_errors_are_reported_not_swallowed(P, gpu_ctx):
    """the C ABI's error behaviour: bad levels, slots out of range, a camera that does not match the pyramids, more features than the
    slot tables can hold -- each one is an error code + message at the call that can detect it, and the context stays usable"""
    W, H = 320, 240
    st = P.synth.make_align_stream(530, W, H, 40, 10, max_level=2)
    imgs = P.synth.render_streams([st]).numpy()
    gpu_ctx.config_pyramids(2, W, H, 3)
    gpu_ctx.build_pyramid(0, imgs[0, 0], 0)
    gpu_ctx.build_pyramid(1, imgs[0, 1], 0)
    good = P.align_job_from_stream(st, 2, 0)

    def expect(code, fn):
        with pytest.raises(P.capi.PlsvoError) as e:
            fn()
        assert e.value.code == code, (e.value.code, str(e.value))
        assert len(str(e.value)) > 20
    expect(P.abi.E_INVALID, lambda: gpu_ctx.sparse_align(P.align_job_from_stream(st, 5, 0)))             # level 5 of a 3-level pyramid
    expect(P.abi.E_INVALID, lambda: gpu_ctx.sparse_align(P.align_job_from_stream(st, 1, 2)))             # max_level < min_level
    expect(P.abi.E_CAPACITY, lambda: gpu_ctx.sparse_align(P.align_job_from_stream(st, 2, 0, ref_slot=0, cur_slot=7)))
    wrong_cam = (st.cam[0], st.cam[1], st.cam[2], st.cam[3], 640, 480)
    expect(P.abi.E_INVALID, lambda: gpu_ctx.sparse_align(P.abi.AlignJob(wrong_cam, 2, 0, 30, 1e-6, st.T_init, st.pt_px, st.pt_xyz_ref, st.seg_spx,
                                                                        st.seg_epx, st.seg_len, st.seg_p_ref, st.seg_q_ref)))
    n = 40000                                                                                             # 40000 patch slots: 480 KB of tables
    rng = np.random.default_rng(1)
    px = np.stack([rng.uniform(20, W - 20, n), rng.uniform(20, H - 20, n)], axis=1)
    xyz = np.concatenate([(px - [st.cam[2], st.cam[3]]) / [st.cam[0], st.cam[1]], np.ones((n, 1))], axis=1) * 3.0
    big = P.abi.AlignJob(st.cam, 2, 0, 30, 1e-6, st.T_init, px, xyz, st.seg_spx[:0], st.seg_epx[:0], st.seg_len[:0], st.seg_p_ref[:0], st.seg_q_ref[:0])
    expect(P.abi.E_CAPACITY, lambda: gpu_ctx.sparse_align(big))
    r = gpu_ctx.sparse_align(good)                                                                        # the context still works
    assert r.n_meas > 0 and np.all(np.isfinite(r.T))


# frame sizes of the launch-shape test: BASELINE configs 2 and 3 (1280 x 720 at five levels: level 4 is 80 x 45, a partial tile ROW of the
# 16 x 8-pixel tiled mirror the one-wave shape reads) and a size whose coarsest level, 62 x 37, is a multiple of the tile in NEITHER dimension
LAUNCH_SHAPE_FRAMES = [(640, 480, 4, 3, 1), (1280, 720, 5, 4, 2), (1000, 600, 5, 4, 2)]


@pytest.mark.parametrize("frame", LAUNCH_SHAPE_FRAMES, ids=[f"{f[0]}x{f[1]}" for f in LAUNCH_SHAPE_FRAMES])
@pytest.mark.parametrize("threads", [64, 128, 256, 512])
def test_sparse_align_every_launch_shape(P, ob, gpu_ctx, threads, frame):
    """the library picks 64 / 128 / 256 / 512 threads per frame from the batch size (64: one wave per frame, no workgroup barrier
    at all -- the shape of the 32768-frame benchmark); every shape must meet the bar on its own, follow the oracle's per-iteration
    trace, and give the same result wherever a job sits in the batch"""
    gpu_ctx.set_launch_shapes(align_threads=threads)
    try:
        _every_launch_shape_body(P, ob, gpu_ctx, threads, *frame)
    finally:
        gpu_ctx.set_launch_shapes(align_threads=0)


def _every_launch_shape_body(P, ob, gpu_ctx, threads, W=640, H=480, nlev=4, maxl=3, minl=1):
    B = 5
    k = 2 if W > 640 else 1      # config 3 has twice the features of config 2
    streams = [P.synth.make_align_stream(700 + i, W, H, k * (200 - 30 * i), k * (80 - 10 * i) - (5 if k == 2 else 0), max_level=maxl) for i in range(B)]
    imgs = P.synth.render_streams(streams).numpy()
    gpu_ctx.config_pyramids(2 * B, W, H, nlev)
    pyr = []
    for i in range(B):
        gpu_ctx.build_pyramid(2 * i, imgs[i, 0], 0)
        gpu_ctx.build_pyramid(2 * i + 1, imgs[i, 1], 0)
        pyr.append((gpu_ctx.download_pyramid(2 * i), gpu_ctx.download_pyramid(2 * i + 1)))
    jobs = [P.align_job_from_stream(s, maxl, minl, ref_slot=2 * i, cur_slot=2 * i + 1) for i, s in enumerate(streams)]
    gpu_ctx.align_set_trace(200)
    batch = gpu_ctx.sparse_align_batch(jobs)
    logs = [gpu_ctx.align_fetch_trace(i) for i in range(B)]
    for i, (st, j) in enumerate(zip(streams, jobs)):
        ro, lo = ob.sparse_align(j, pyr[i][0], pyr[i][1], max_log=200)
        n, worst = Hh.compare_align_logs(lo, logs[i])
        assert n >= 1 and worst["H"] < 1e-6 and worst["chi2"] < 1e-4, (threads, i, worst)
        ang, tr, ok = Hh.pose_close(Hh.frame_pose(batch[i].T, st), Hh.frame_pose(ro.T, st))
        assert ok, (threads, i, ang, tr)
        assert np.array_equal(batch[i].seg_alive, ro.seg_alive)
    gpu_ctx.align_set_trace(0)
    single = gpu_ctx.sparse_align(jobs[3])
    assert np.array_equal(single.T, batch[3].T) and single.n_meas == batch[3].n_meas


@pytest.mark.parametrize("threads", [16, 64, 256, 512])
def test_pose_optimizer_every_launch_shape(P, ob, gpu_ctx, threads):
    gpu_ctx.set_launch_shapes(poseopt_threads=threads)
    try:
        _poseopt_launch_shape_body(P, ob, gpu_ctx)
    finally:
        gpu_ctx.set_launch_shapes(poseopt_threads=0)


def _poseopt_launch_shape_body(P, ob, gpu_ctx):
    for seed, npts, nseg, nref in ((77, 500, 200, -1), (79, 300, 100, 5), (82, 7, 3, -1)):
        job = P.poseopt_job_from_frame(P.synth.make_poseopt_frame(seed, npts, nseg), n_iter_ref=nref)
        ro, _ = ob.pose_optimize(job)
        gpu_ctx.poseopt_set_trace(0)
        rd = gpu_ctx.pose_optimize(job)
        assert Hh.pose_close(rd.T, ro.T)[2]
        assert np.array_equal(rd.pt_keep, ro.pt_keep) and np.array_equal(rd.seg_keep, ro.seg_keep)
        assert (rd.num_obs_pt, rd.num_obs_ls) == (ro.num_obs_pt, ro.num_obs_ls)
        assert rd.error_init == pytest.approx(ro.error_init, rel=1e-9) and rd.error_final == pytest.approx(ro.error_final, rel=1e-6)
        assert Hh.rel(rd.cov, ro.cov) < 1e-6


POSE_CASES = [("config5", 77, 500, 200, -1), ("config5-b", 1077, 500, 200, -1), ("frame-200-80", 78, 200, 80, -1), ("ten-arg", 79, 300, 100, 5),
              ("points-only", 80, 120, 0, -1), ("lines-only", 81, 0, 60, -1), ("tiny", 82, 7, 3, -1)]


@pytest.mark.parametrize("case", POSE_CASES, ids=[c[0] for c in POSE_CASES])
def test_pose_optimizer_matches_oracle(P, ob, gpu_ctx, case):
    tag, seed, npts, nseg, nref = case
    fr = P.synth.make_poseopt_frame(seed, npts, nseg)
    job = P.poseopt_job_from_frame(fr, n_iter_ref=nref)
    ro, lo = ob.pose_optimize(job, max_log=40)
    gpu_ctx.poseopt_set_trace(40)
    rd = gpu_ctx.pose_optimize(job)
    ld = gpu_ctx.poseopt_fetch_trace(0)
    assert Hh.pose_close(rd.T, ro.T)[2], Hh.pose_close(rd.T, ro.T)
    assert np.array_equal(rd.pt_keep, ro.pt_keep) and np.array_equal(rd.seg_keep, ro.seg_keep)
    assert (rd.num_obs_pt, rd.num_obs_ls) == (ro.num_obs_pt, ro.num_obs_ls)
    assert rd.estimated_scale == pytest.approx(ro.estimated_scale, rel=1e-6)
    assert rd.error_init == pytest.approx(ro.error_init, rel=1e-9)
    assert rd.error_final == pytest.approx(ro.error_final, rel=1e-6)
    assert Hh.rel(rd.cov, ro.cov) < 1e-6
    a, b = lo[0], ld[0]
    assert Hh.rel(b["A"], a["A"]) < 1e-9 and Hh.rel(b["b"], a["b"]) < 1e-7
    assert abs(a["new_chi2"] - b["new_chi2"]) <= 1e-9 * abs(a["new_chi2"])


def test_pose_optimizer_row_per_frame_shape_on_a_mixed_batch(P, ob, gpu_ctx):
    """The large-batch shape of the pose optimiser (a 16-lane DPP row per frame, four frames per wave: pose_opt_rows_kernel) on ONE batch
    that mixes everything the other tests feed one frame at a time -- config 5, the 10-argument overload, points only, lines only, tiny,
    empty, rank-deficient, noise-free, far outliers, NaN pose, identical points, zero iterations, 2000 + 600 features -- so that the four
    frames of a wave differ in feature counts, iteration counts, early returns and NaN patterns; 19 frames (the last wave has an idle
    row).  Every frame against the oracle as in its own test, and against the same frame alone at the default shape."""
    import copy
    frames = [(P.synth.make_poseopt_frame(seed, npts, nseg), dict(n_iter_ref=nref)) for _, seed, npts, nseg, nref in POSE_CASES]
    frames += [(P.synth.make_poseopt_frame(83, npts, nseg), {}) for _, npts, nseg in POSE_DEGENERATE]
    frames.append((P.synth.make_poseopt_frame(95, 60, 20, noise_px=0.0, outlier_frac=0.0, pert_t=0.0, pert_r=0.0), {}))
    fr = P.synth.make_poseopt_frame(96, 60, 20)
    f3 = copy.copy(fr); f3.pt_pos = fr.pt_pos.copy(); f3.pt_pos[:5] = 1e6
    f4 = copy.copy(fr); f4.T_init = fr.T_init.copy(); f4.T_init[6] = np.nan
    f5 = copy.copy(fr); f5.pt_pos = np.repeat(fr.pt_pos[:1], len(fr.pt_pos), 0); f5.pt_f = np.repeat(fr.pt_f[:1], len(fr.pt_f), 0)
    frames += [(f3, {}), (f4, {}), (f5, {}), (fr, {"n_iter": 0})]
    assert len(frames) == 19
    jobs = [P.poseopt_job_from_frame(f, **kw) for f, kw in frames]
    gpu_ctx.poseopt_set_trace(0)
    alone = [gpu_ctx.pose_optimize(j) for j in jobs]            # default shape for one frame: a workgroup per frame
    gpu_ctx.set_launch_shapes(poseopt_threads=16)
    try:
        rows = gpu_ctx.pose_optimize_batch(jobs)
        recs = gpu_ctx.fetch_pose_records(len(jobs))
    finally:
        gpu_ctx.set_launch_shapes(poseopt_threads=0)
    for k, (job, rd, ra) in enumerate(zip(jobs, rows, alone)):
        ro, _ = ob.pose_optimize(job)
        assert np.array_equal(np.isnan(rd.T), np.isnan(ro.T)), (k, rd.T, ro.T)
        assert np.array_equal(rd.pt_keep, ro.pt_keep) and np.array_equal(rd.seg_keep, ro.seg_keep), k
        assert (rd.num_obs_pt, rd.num_obs_ls, rd.iters, rd.iters_ref, rd.status) == (ro.num_obs_pt, ro.num_obs_ls, ro.iters, ro.iters_ref, ro.status), k
        if not np.any(np.isnan(ro.T)):
            assert Hh.pose_close(rd.T, ro.T)[2], (k, Hh.pose_close(rd.T, ro.T))
            # the two launch shapes sum the normal equations in different orders: equal to rounding, not bitwise
            assert Hh.pose_close(rd.T, ra.T, rot_tol=1e-9, trans_tol=1e-9)[2], (k, Hh.pose_close(rd.T, ra.T))
            if k < len(POSE_CASES) + len(POSE_DEGENERATE):     # (noise-free data at the true pose: every error is projection rounding)
                assert rd.error_init == pytest.approx(ro.error_init, rel=1e-9, abs=1e-300), k
            if k < len(POSE_CASES):      # (rank-deficient and noise-free frames: the reference's own result hangs on rounding)
                assert rd.error_final == pytest.approx(ro.error_final, rel=1e-6) and Hh.rel(rd.cov, ro.cov) < 1e-6, k
                assert rd.estimated_scale == pytest.approx(ro.estimated_scale, rel=1e-6), k
        assert (rd.num_obs_pt, rd.num_obs_ls, rd.iters) == (ra.num_obs_pt, ra.num_obs_ls, ra.iters), k
        assert np.array_equal(recs[k]["T_f_w"], np.asarray(rd.T), equal_nan=True) and int(recs[k]["num_obs_pt"]) == rd.num_obs_pt, k
    assert int(recs[7]["status"]) & P.abi.REC_POSEOPT_EMPTY      # the empty frame


POSE_DEGENERATE = [("empty", 0, 0), ("one-point", 1, 0), ("two-points", 2, 0), ("one-line", 0, 1), ("three-points", 3, 0), ("two-and-two", 2, 2),
                   ("large-2000-600", 2000, 600)]


@pytest.mark.parametrize("case", POSE_DEGENERATE, ids=[c[0] for c in POSE_DEGENERATE])
def test_pose_optimizer_degenerate_and_large_sizes(P, ob, gpu_ctx, case):
    """no observations at all, fewer observations than unknowns (rank-deficient normal equations: the pivoted LDLT's zero pivots
    give zero components, as Eigen's solve does), and a frame several times the benchmark's size"""
    tag, npts, nseg = case
    fr = P.synth.make_poseopt_frame(83, npts, nseg)
    job = P.poseopt_job_from_frame(fr)
    ro, lo = ob.pose_optimize(job, max_log=40)
    gpu_ctx.poseopt_set_trace(40)
    rd = gpu_ctx.pose_optimize(job)
    ld = gpu_ctx.poseopt_fetch_trace(0)
    assert np.all(np.isfinite(rd.T))
    assert np.array_equal(rd.pt_keep, ro.pt_keep) and np.array_equal(rd.seg_keep, ro.seg_keep)
    assert (rd.num_obs_pt, rd.num_obs_ls) == (ro.num_obs_pt, ro.num_obs_ls)
    assert len(ld) == len(lo) and rd.iters == ro.iters
    assert Hh.pose_close(rd.T, ro.T)[2], (Hh.pose_close(rd.T, ro.T), rd.T, ro.T)
    assert rd.error_init == pytest.approx(ro.error_init, rel=1e-9, abs=1e-300)
    for a, b in zip(lo, ld):     # rank-deficient steps stay inside the observable directions: the same zero components, the rest to rounding
        assert np.array_equal(a["dT"] == 0.0, b["dT"] == 0.0), (tag, a["dT"], b["dT"])
        assert np.allclose(b["dT"], a["dT"], rtol=1e-6, atol=1e-9 * max(1e-300, float(np.max(np.abs(lo[0]["dT"]))))), (tag, a["dT"], b["dT"])
    if npts + nseg == 0:
        assert np.array_equal(rd.T, np.asarray(job.c.T_f_w[:], dtype=np.float64))   # nothing to optimise: the pose comes back untouched


def test_pose_optimizer_adversarial_inputs(P, ob, gpu_ctx):
    """noise-free data at the true pose (every error ~1e-14, the MAD scale with it), NaN pose, far outliers, identical
    observations, zero iterations: same masks, counts, iteration counts and NaN pattern as the reference"""
    import copy
    cases = []
    cases.append(("exact-data-exact-start", P.synth.make_poseopt_frame(95, 60, 20, noise_px=0.0, outlier_frac=0.0, pert_t=0.0, pert_r=0.0), {}))
    cases.append(("exact-points-only", P.synth.make_poseopt_frame(95, 60, 0, noise_px=0.0, outlier_frac=0.0, pert_t=0.0, pert_r=0.0), {}))
    cases.append(("exact-data-start-off", P.synth.make_poseopt_frame(95, 60, 20, noise_px=0.0, outlier_frac=0.0), {}))
    fr = P.synth.make_poseopt_frame(96, 60, 20)
    # (NaN observations are not compared: vk::getMedian runs std::nth_element over floats that contain NaN -- not a strict weak
    #  ordering, the reference's scale is undefined there)
    f3 = copy.copy(fr); f3.pt_pos = fr.pt_pos.copy(); f3.pt_pos[:5] = 1e6
    cases.append(("far-outliers", f3, {}))
    f4 = copy.copy(fr); f4.T_init = fr.T_init.copy(); f4.T_init[6] = np.nan
    cases.append(("nan-pose", f4, {}))
    f5 = copy.copy(fr); f5.pt_pos = np.repeat(fr.pt_pos[:1], len(fr.pt_pos), 0); f5.pt_f = np.repeat(fr.pt_f[:1], len(fr.pt_f), 0)
    cases.append(("identical-points", f5, {}))
    cases.append(("zero-iterations", fr, {"n_iter": 0}))
    gpu_ctx.poseopt_set_trace(0)
    for tag, f, kw in cases:
        job = P.poseopt_job_from_frame(f, **kw)
        ro, _ = ob.pose_optimize(job)
        rd = gpu_ctx.pose_optimize(job)
        assert np.array_equal(np.isnan(rd.T), np.isnan(ro.T)), (tag, rd.T, ro.T)
        assert np.array_equal(rd.pt_keep, ro.pt_keep) and np.array_equal(rd.seg_keep, ro.seg_keep), tag
        assert (rd.num_obs_pt, rd.num_obs_ls, rd.iters) == (ro.num_obs_pt, ro.num_obs_ls, ro.iters), tag
        if not np.isnan(ro.T).any():
            assert Hh.pose_close(rd.T, ro.T)[2], (tag, Hh.pose_close(rd.T, ro.T))
        if tag.startswith("exact-"):
            # every error is rounding residue of the projection (1e-14 and below; the device rotates with a matrix, Sophus with the
            # quaternion), the MAD scale with it, and the reference's line Jacobian divides by e.norm() (:156-157): whether a line's
            # float ds, de come out as 1e-17 or as exact zeros -- 0/0, a NaN step, a rolled-back iteration, a NaN covariance --
            # is decided by the last bit.  Pose, masks, counts and iteration counts (checked above) do not depend on it.
            continue
        for a, b in ((rd.error_init, ro.error_init), (rd.error_final, ro.error_final), (rd.estimated_scale, ro.estimated_scale)):
            assert (np.isnan(a) and np.isnan(b)) or a == pytest.approx(b, rel=1e-5), (tag, a, b)
        assert np.array_equal(np.isnan(rd.cov), np.isnan(ro.cov)), tag
        if not np.isnan(ro.cov).any():
            assert Hh.rel(rd.cov, ro.cov) < 1e-6, tag


def test_pose_optimizer_seed_sweep(P, ob, gpu_ctx):
    """40 seeds of a 200-point + 80-segment frame: keep masks, counts and iteration counts bit-equal on every seed, poses to 1e-9"""
    import json, os
    worst = {"rot_rad": 0.0, "trans_rel": 0.0}
    gpu_ctx.poseopt_set_trace(0)
    for seed in range(5000, 5040):
        fr = P.synth.make_poseopt_frame(seed, 200, 80)
        job = P.poseopt_job_from_frame(fr)
        ro, _ = ob.pose_optimize(job, max_log=0)
        rd = gpu_ctx.pose_optimize(job)
        assert np.array_equal(rd.pt_keep, ro.pt_keep) and np.array_equal(rd.seg_keep, ro.seg_keep), seed
        assert (rd.num_obs_pt, rd.num_obs_ls, rd.iters) == (ro.num_obs_pt, ro.num_obs_ls, ro.iters), seed
        ang, tr, ok = Hh.pose_close(rd.T, ro.T)
        assert ok and ang < 1e-9 and tr < 1e-9, (seed, ang, tr)
        assert rd.estimated_scale == pytest.approx(ro.estimated_scale, rel=1e-6)
        worst["rot_rad"] = max(worst["rot_rad"], ang); worst["trans_rel"] = max(worst["trans_rel"], tr)
    out = {"what": "40 seeds (5000..5039), 200 points + 80 segments, pose_optimizer::optimizeGaussNewton: HIP path vs CPU oracle; keep masks, "
                   "observation counts and iteration counts equal on every seed", "worst": worst}
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(root, "gpurun_out", "poseopt_seed_sweep.json"), "w"), indent=1)
    print(json.dumps(out))


def test_pose_optimizer_known_answer(P, gpu_ctx):
    """noise-free observations, no outliers: the optimiser must return T_true and keep everything"""
    fr = P.synth.make_poseopt_frame(90, 200, 0, noise_px=1e-3, outlier_frac=0.0)
    rd = gpu_ctx.pose_optimize(P.poseopt_job_from_frame(fr))
    ang, dist = P.synth.se3_log_angle_dist(rd.T, fr.T_true)
    assert ang < 1e-6 and dist < 1e-5
    assert rd.pt_keep.all()
    # with segments the reference's line Jacobian (one factor for both rows, :156-157) converges only
    # linearly: 10 iterations leave ~4e-5 rad -- the device must land where the reference lands
    fr = P.synth.make_poseopt_frame(90, 200, 60, noise_px=1e-3, outlier_frac=0.0)
    rd = gpu_ctx.pose_optimize(P.poseopt_job_from_frame(fr))
    ang, dist = P.synth.se3_log_angle_dist(rd.T, fr.T_true)
    assert ang < 1e-4 and dist < 1e-3
    assert rd.pt_keep.all() and rd.seg_keep.all()
    # gross outliers are culled
    fr = P.synth.make_poseopt_frame(91, 300, 0, noise_px=0.3, outlier_frac=0.1, outlier_px=30.0)
    rd = gpu_ctx.pose_optimize(P.poseopt_job_from_frame(fr))
    assert not rd.pt_keep[fr.pt_outlier].any()
    assert rd.pt_keep[~fr.pt_outlier].mean() > 0.95


def test_pose_optimizer_batch_equals_single(P, gpu_ctx):
    frames = [P.synth.make_poseopt_frame(300 + i, 100 + 10 * i, 30 + 3 * i) for i in range(5)]
    jobs = [P.poseopt_job_from_frame(f) for f in frames]
    gpu_ctx.poseopt_set_trace(0)
    batch = gpu_ctx.pose_optimize_batch(jobs)
    for j, rb in zip(jobs, batch):
        rs = gpu_ctx.pose_optimize(j)
        assert np.array_equal(rs.T, rb.T) and np.array_equal(rs.pt_keep, rb.pt_keep) and rs.iters == rb.iters


def test_full_size_properties(P, gpu_ctx):
    """BASELINE config 2 at batch size, checked through size-independent properties instead of the oracle:
    (i) re-running a staged batch is idempotent, (ii) the aligned pose lands near the synthetic ground
    truth, (iii) alignment started AT the converged pose stays there."""
    import torch
    B, W, H = 32, 640, 480
    streams = [P.synth.make_align_stream(2000 + i, W, H, 200, 80, max_level=3) for i in range(B)]
    imgs = P.synth.render_streams(streams, device="cuda")
    torch.cuda.synchronize()   # the library enqueues on its own stream: the rendered images must be complete before it reads them
    gpu_ctx.config_pyramids(2 * B, W, H, 4)
    gpu_ctx.build_pyramids_dev(0, 2 * B, imgs.data_ptr(), W, W * H, 0)
    gpu_ctx.synchronize()
    jobs = [P.align_job_from_stream(s, 3, 1, ref_slot=2 * i, cur_slot=2 * i + 1) for i, s in enumerate(streams)]
    gpu_ctx.align_set_trace(0)
    gpu_ctx.align_stage(jobs)
    gpu_ctx.align_run()
    r1 = gpu_ctx.align_fetch()
    gpu_ctx.align_run()
    r2 = gpu_ctx.align_fetch()
    for a, b in zip(r1, r2):
        assert np.array_equal(a.T, b.T) and a.n_meas == b.n_meas
    errs = np.array([P.synth.se3_log_angle_dist(r.T, s.T_true) for r, s in zip(r1, streams)])
    assert np.median(errs[:, 0]) < 1e-3 and np.median(errs[:, 1]) < 1e-2
    jobs2 = []
    for s, r, j in zip(streams, r1, jobs):
        jobs2.append(P.abi.AlignJob(s.cam, 3, 1, 30, 1e-6, r.T, s.pt_px, s.pt_xyz_ref, s.seg_spx, s.seg_epx, s.seg_len,
                                    s.seg_p_ref, s.seg_q_ref, seg_alive_in=r.seg_alive, ref_slot=j.c.ref_slot, cur_slot=j.c.cur_slot))
    r3 = gpu_ctx.sparse_align_batch(jobs2)
    for a, c in zip(r1, r3):
        ang, dist = P.synth.se3_log_angle_dist(a.T, c.T)
        assert ang < 2e-4 and dist < 2e-3


@pytest.mark.gpu
def test_every_entry_point_rejects_malformed_arguments(P, gpu_ctx):
    """Negative counts, missing arrays and impossible parameters come back as PLSVO_E_INVALID with a message naming the entry point --
    before anything is enqueued -- and leave the context usable."""
    import ctypes as C
    A, L, h = P.abi, gpu_ctx.L, gpu_ctx.h
    gpu_ctx.config_pyramids(2, 160, 120, 3)

    def rejected(rc, name):
        msg = (L.plsvo_hip_last_error(h) or b"").decode()
        assert rc != 0 and name in msg, (name, rc, msg)

    so = A.StructOptIn(); so.n_pts = -1
    rejected(L.plsvo_structure_optimize(h, C.byref(so), C.byref(A.StructOptOut())), "structure_optimize")
    so = A.StructOptIn(); so.n_pts = 3                       # landmarks without their arrays
    rejected(L.plsvo_structure_optimize(h, C.byref(so), C.byref(A.StructOptOut())), "structure_optimize")
    mi = A.MatchIn()                                         # n_pyr_levels = 0
    rejected(L.plsvo_match_direct(h, C.byref(mi), C.byref(A.MatchOut())), "match_direct")
    mi = A.MatchIn(); mi.n = 4; mi.n_frames = 1; mi.n_pyr_levels = 3
    rejected(L.plsvo_match_direct(h, C.byref(mi), C.byref(A.MatchOut())), "match_direct")
    rejected(L.plsvo_reproject(h, C.byref(A.ReprojectIn()), C.byref(A.ReprojectOut())), "reproject")          # cell_size = 0
    rejected(L.plsvo_chain_stage(h, 0, None, None), "chain_stage")
    cp = A.ChainParams()                                     # cell_size = 0
    rejected(L.plsvo_chain_stage(h, 1, C.byref(A.ChainIn()), C.byref(cp)), "chain_stage")
    rejected(L.plsvo_frame_step_batch(h, 1, C.byref(A.ChainIn()), C.byref(cp), C.byref(A.ChainOut())), "chain_stage")
    rejected(L.plsvo_update_seeds(h, C.byref(A.SeedsIn()), C.byref(A.SeedsOut())), "update_seeds")            # n_pyr_levels = 0
    si = A.SeedsIn(); si.n_pyr_levels = 3; si.n_pt = 2; si.n_frames = 1; si.cam.width, si.cam.height = 160, 120
    rejected(L.plsvo_update_seeds(h, C.byref(si), C.byref(A.SeedsOut())), "update_seeds")                     # seeds without their arrays
    po = A.PoseOptIn(); po.n_pts = -2
    rejected(L.plsvo_pose_optimize(h, C.byref(po), C.byref(A.PoseOptOut())), "poseopt_stage")
    po = A.PoseOptIn(); po.n_pts = 5
    rejected(L.plsvo_pose_optimize(h, C.byref(po), C.byref(A.PoseOptOut())), "poseopt_stage")
    rejected(L.plsvo_poseopt_copy_poses(h, None), "poseopt_copy_poses")
    rejected(L.plsvo_hip_kernel_time(h, 99, None, None), "kernel_time")
    # still usable
    img = np.random.default_rng(5).integers(0, 256, (120, 160), dtype=np.uint8)
    gpu_ctx.build_pyramid(0, img, 0)
    assert np.array_equal(gpu_ctx.download_level(0, 0), img)


@pytest.mark.gpu
def test_upload_pyramid_frees_the_callers_buffers_and_refreshes_tiles_lazily(P, ob, gpu_ctx):
    """plsvo_hip_upload_pyramid (round 5): the levels cross PCIe as ONE copy from a pinned image of the slot and the call does not wait
    for it -- so the caller may overwrite its buffers the moment the call returns (the reference's cv::Mat pyramids are the caller's) --
    and the TILED mirror of an uploaded slot is refreshed only before a launch that reads it: the one-wave-per-frame shape, also through
    a slot copy made while the mirror was stale.  Three uploads in a row exercise both pinned images and the event that guards them."""
    W, H, nlev = 640, 480, 4
    st, ref, cur, job = Hh.make_case(ob, 1235, W, H, 200, 80, nlev, 3, 1)
    res_o, _ = ob.sparse_align(job, ref, cur)
    gpu_ctx.config_pyramids(6, W, H, nlev)
    scratch = [l.copy() for l in ref]
    gpu_ctx.upload_pyramid(0, scratch)
    for l in scratch:
        l[...] = 0                                   # the caller's buffers are its own again
    junk = [np.full_like(l, 77) for l in cur]
    gpu_ctx.upload_pyramid(1, junk)                  # second pinned image
    gpu_ctx.upload_pyramid(1, cur)                   # first one again: its previous DMA is awaited, not overwritten
    for l in range(nlev):
        assert np.array_equal(gpu_ctx.download_level(0, l), ref[l]) and np.array_equal(gpu_ctx.download_level(1, l), cur[l])
    gpu_ctx.copy_slots(2, 0, 2)                      # copies made while the mirrors of slots 0 and 1 are stale
    moved = P.align_job_from_stream(st, 3, 1, ref_slot=2, cur_slot=3)
    for threads, jb in ((0, job), (64, job), (64, moved)):   # default (row-major slab), then the shape that reads the tiles
        gpu_ctx.set_launch_shapes(align_threads=threads)
        try:
            r = gpu_ctx.sparse_align(jb)
        finally:
            gpu_ctx.set_launch_shapes(align_threads=0)
        ang, tr, ok = Hh.pose_close(r.T, res_o.T)
        assert ok and ang < 1e-8 and r.n_meas == res_o.n_meas and r.iters_per_level == res_o.iters_per_level, (threads, ang, tr)


@pytest.mark.gpu
def test_two_workgroups_per_frame_shape(P, ob, gpu_ctx):
    """Small batches run TWO workgroups per frame (rank 0 the point slots, rank 1 the segments' samples, 32 partial sums exchanged per
    Gauss-Newton iteration: align_kernels.hip).  The shape must be deterministic call to call, independent of where a frame sits in the
    batch and of how many frames there are (1 .. 20: several groups of sixteen blocks, a partly filled last group), and every frame must
    follow the oracle; frames without segments / without points leave one of the two workgroups without slots."""
    W, H = 640, 480
    kinds = [(200, 80), (120, 0), (0, 60), (200, 80), (30, 5)]
    streams = [P.synth.make_align_stream(900 + i, W, H, *kinds[i % len(kinds)], max_level=3, motion_scale=0.1 if kinds[i % len(kinds)][0] == 0 else 0.5)
               for i in range(20)]
    imgs = P.synth.render_streams(streams).numpy()
    gpu_ctx.config_pyramids(2 * len(streams), W, H, 4)
    pyr = []
    for i in range(len(streams)):
        gpu_ctx.build_pyramid(2 * i, imgs[i, 0], 0)
        gpu_ctx.build_pyramid(2 * i + 1, imgs[i, 1], 0)
        pyr.append((gpu_ctx.download_pyramid(2 * i), gpu_ctx.download_pyramid(2 * i + 1)))
    jobs = [P.align_job_from_stream(s, 3, 1, ref_slot=2 * i, cur_slot=2 * i + 1) for i, s in enumerate(streams)]
    gpu_ctx.align_set_trace(200)
    batch = gpu_ctx.sparse_align_batch(jobs)
    logs = [gpu_ctx.align_fetch_trace(i) for i in range(len(jobs))]
    gpu_ctx.align_set_trace(0)
    again = gpu_ctx.sparse_align_batch(jobs)
    work = gpu_ctx.align_work()
    for i, (st, j) in enumerate(zip(streams, jobs)):
        assert np.array_equal(batch[i].T, again[i].T) and batch[i].n_meas == again[i].n_meas, i              # deterministic
        ro, lo = ob.sparse_align(j, pyr[i][0], pyr[i][1], max_log=200)
        n, worst = Hh.compare_align_logs(lo, logs[i])
        # (1e-11 while the poses are equal to rounding; a few 1e-6 once one patch position has moved by a float ulp -- a 35-patch frame
        #  shows it at 6e-6, with one workgroup per frame as with two)
        assert n >= 1 and worst["H"] < 1e-4 and worst["chi2"] < 1e-4, (i, worst)
        assert np.array_equal(batch[i].seg_alive, ro.seg_alive), i
        if Hh.same_path(lo, logs[i]):
            assert batch[i].n_meas == ro.n_meas and batch[i].iters_per_level == ro.iters_per_level, i
            ang, tr, ok = Hh.pose_close(Hh.frame_pose(batch[i].T, st), Hh.frame_pose(ro.T, st))
            assert ok, (i, ang, tr)
    for k in (0, 7, 19):                                                                                       # alone == inside the batch
        single = gpu_ctx.sparse_align(jobs[k])
        assert np.array_equal(single.T, batch[k].T) and single.n_meas == batch[k].n_meas, k
    # the work counters of the two workgroups add up (each counts its own slots, the pair exchanges the counts): the one-workgroup shape's
    gpu_ctx.set_launch_shapes(align_threads=256)      # (four frames per CU at most: the latency shape, but > cu_count / 2 ... forced here by
    try:                                              #  the environment-free route: 256 threads at 20 frames still pairs, so compare totals only)
        gpu_ctx.sparse_align_batch(jobs)
        work256 = gpu_ctx.align_work()
    finally:
        gpu_ctx.set_launch_shapes(align_threads=0)
    assert work[0] == work256[0] > 0 and abs(work[1] - work256[1]) <= 0.02 * work[1], (work, work256)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [64, 128, 256])
def test_launch_order_follows_the_measured_work_of_the_last_run(P, ob, threads):
    """A resident batch larger than the device's resident slots starts its frames longest-first by what they cost in its LAST launch
    (align_kernels.hip::align_reorder_kernel; the threshold is lowered to 4 frames for this test): the order is a permutation, sorted by
    the patch-iterations the device counted, and -- scheduling only -- every frame's result equals the single-frame call's, run after run."""
    import os
    os.environ["PLSVO_ALIGN_REORDER_MIN"] = "4"
    try:
        ctx = P.capi.Context(0)
    finally:
        del os.environ["PLSVO_ALIGN_REORDER_MIN"]
    try:
        W, H, B = 320, 240, 9
        streams = [P.synth.make_align_stream(300 + i, W, H, 60 + 15 * (i % 4), 20 - 4 * (i % 3), max_level=3, motion_scale=0.3 + 0.2 * (i % 5)) for i in range(B)]
        imgs = P.synth.render_streams(streams).numpy()
        ctx.config_pyramids(2 * B, W, H, 4)
        for i in range(B):
            ctx.build_pyramid(2 * i, imgs[i, 0], 0)
            ctx.build_pyramid(2 * i + 1, imgs[i, 1], 0)
        jobs = [P.align_job_from_stream(s, 3, 1, ref_slot=2 * i, cur_slot=2 * i + 1) for i, s in enumerate(streams)]
        ctx.set_launch_shapes(align_threads=threads)                # (every shape a batch larger than its resident workgroups can run in)
        single = []
        for j in jobs:
            single.append(ctx.sparse_align(j))
        ctx.align_stage(jobs)
        order0 = ctx.align_launch_order(B)
        assert sorted(order0.tolist()) == list(range(B))            # the stage call's order: a permutation (most patches first)
        for rep in range(3):
            ctx.align_run()
            res = ctx.align_fetch()
            order = ctx.align_launch_order(B)
            assert sorted(order.tolist()) == list(range(B)), order
            iters = np.array([sum(r.iters_per_level) for r in res])
            assert iters[order[0]] >= iters[order[-1]]                # longest first (bins of 128 patch-iterations: not a strict sort)
            for k in range(B):
                assert np.array_equal(res[k].T, single[k].T) and res[k].n_meas == single[k].n_meas and res[k].iters_per_level == single[k].iters_per_level, (rep, k)
        # PLSVO_OPT_ALIGN_REORDER = 0 (round 6): back to the stage call's order at once, every later launch keeps it, results unchanged;
        # = 1: the next launch sorts again (what bench.py switches between: its timed steps run in the staged order)
        ctx.set_launch_order_refresh(align=False)
        assert np.array_equal(ctx.align_launch_order(B), order0)
        for rep in range(2):
            ctx.align_run()
            res = ctx.align_fetch()
            assert np.array_equal(ctx.align_launch_order(B), order0)
            for k in range(B):
                assert np.array_equal(res[k].T, single[k].T) and res[k].iters_per_level == single[k].iters_per_level, (rep, k)
        ctx.set_launch_order_refresh(align=True)
        ctx.align_run()
        ctx.align_run()
        order = ctx.align_launch_order(B)
        iters = np.array([sum(r.iters_per_level) for r in ctx.align_fetch()])
        assert sorted(order.tolist()) == list(range(B)) and iters[order[0]] >= iters[order[-1]]
        ctx.align_stage(jobs)                                        # staging again resets the order to the host's
        assert np.array_equal(ctx.align_launch_order(B), order0)
    finally:
        ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [16, 64])
def test_poseopt_launch_order_of_a_rerun_batch_changes_no_result(P, ob, threads):
    """A staged pose-optimiser batch that is run again takes its frames sorted by the feature-iterations each evaluated in the last launch
    (the four frames of a `pose_opt_rows_kernel` wave run until the last of them stops; threshold lowered to 4 frames here): scheduling
    only -- every run's results equal the single-frame calls', which equal the oracle's."""
    import os
    os.environ["PLSVO_POSEOPT_REORDER_MIN"] = "4"
    try:
        ctx = P.capi.Context(0)
    finally:
        del os.environ["PLSVO_POSEOPT_REORDER_MIN"]
    try:
        ctx.set_launch_shapes(poseopt_threads=threads)
        B = 11                                                       # (not a multiple of four: the last wave has idle rows)
        frames = [P.synth.make_poseopt_frame(700 + i, 40 + 13 * (i % 5), 10 + 7 * (i % 3), 640, 480) for i in range(B)]
        jobs = [P.poseopt_job_from_frame(f, n_iter_ref=(3 if i % 4 == 1 else -1)) for i, f in enumerate(frames)]
        single = [ctx.pose_optimize(j) for j in jobs]
        assert len({r.iters for r in single}) > 1                    # the batch does mix iteration counts
        for k, j in enumerate(jobs):
            o, _ = ob.pose_optimize(j)
            assert single[k].iters == o.iters and np.array_equal(single[k].pt_keep, o.pt_keep) and np.allclose(single[k].T, o.T, rtol=0, atol=1e-9), k
        ctx.poseopt_stage(jobs)
        for rep in range(3):
            ctx.poseopt_run()
            res = ctx.poseopt_fetch()
            for k in range(B):
                a, b = res[k], single[k]
                assert np.array_equal(a.T, b.T) and np.array_equal(a.cov, b.cov) and a.iters == b.iters and a.iters_ref == b.iters_ref, (rep, k)
                assert a.error_init == b.error_init and a.error_final == b.error_final and a.num_obs_pt == b.num_obs_pt and a.num_obs_ls == b.num_obs_ls, (rep, k)
                assert np.array_equal(a.pt_keep, b.pt_keep) and np.array_equal(a.seg_keep, b.seg_keep), (rep, k)
    finally:
        ctx.close()


def _device_bytes(host):
    """`host` (uint8 array) in memory the library's device pointers can address: HBM through torch on a GPU box; the array itself when
    the library is the host emulation build (tests/test_emu_parity.py), whose device memory is host memory.  -> (keep-alive, pointer, read-back)"""
    import torch
    if torch.cuda.is_available():
        t = torch.from_numpy(host).cuda()
        torch.cuda.synchronize()
        return t, t.data_ptr(), lambda: t.cpu().numpy()
    return host, host.ctypes.data, lambda: host


@pytest.mark.gpu
def test_abi_utility_entry_points(P, ob, gpu_ctx):
    """The entry points around the hot path that the benchmark and a device-resident caller use: pyramids built from level-0 images that
    are already in device memory (strided rows), slot-to-slot copies, the one-call form, pose copies on the device, work counters,
    kernel timers, device info -- each against the oracle or against the per-call path."""
    import ctypes as C
    W, H, nlev = 162, 122, 3                     # odd quarter widths: the 16-byte fast path and the byte tail of copy_level0 both run
    rng = np.random.default_rng(77)
    stride = W + 14
    imgs = rng.integers(0, 256, (3, H, stride), dtype=np.uint8)
    keep, d_ptr, _ = _device_bytes(imgs.reshape(-1).copy())
    gpu_ctx.config_pyramids(6, W, H, nlev)
    gpu_ctx.build_pyramids_dev(0, 3, d_ptr, stride, H * stride, 0)
    gpu_ctx.synchronize()
    for k in range(3):
        for d, o in zip(gpu_ctx.download_pyramid(k), ob.build_pyramid(np.ascontiguousarray(imgs[k, :, :W]), nlev, 0)):
            assert np.array_equal(d, o)
    gpu_ctx.copy_slots(3, 0, 3)
    for k in range(3):
        for d, o in zip(gpu_ctx.download_pyramid(3 + k), gpu_ctx.download_pyramid(k)):
            assert np.array_equal(d, o)
    with pytest.raises(P.capi.PlsvoError):
        gpu_ctx.copy_slots(1, 0, 3)              # overlapping ranges
    # one-call form == stage / run / fetch; the copied slots (tiled mirror included) align like the originals
    st, ref, cur, job = Hh.make_case(ob, 31, 320, 240, 60, 20, 4, 3, 1)
    gpu_ctx.config_pyramids(4, 320, 240, 4)
    gpu_ctx.upload_pyramid(0, ref)
    gpu_ctx.upload_pyramid(1, cur)
    gpu_ctx.copy_slots(2, 0, 2)
    gpu_ctx.align_set_trace(0)
    gpu_ctx.set_profiling(True)
    gpu_ctx.reset_profiling()
    r0 = gpu_ctx.sparse_align(job)
    out = P.abi.AlignOut()
    alive = np.ones(max(job.n_seg, 1), dtype=np.uint8)
    out.seg_alive_out = alive.ctypes.data_as(P.abi.c_u8_p)
    assert gpu_ctx.L.plsvo_sparse_align(gpu_ctx.h, C.byref(job.c), C.byref(out)) == 0
    assert np.array_equal(np.array(list(out.T_cur_from_ref)), r0.T) and np.array_equal(alive[:job.n_seg], r0.seg_alive)
    st2, ref2, cur2, job2 = Hh.make_case(ob, 31, 320, 240, 60, 20, 4, 3, 1)
    job2.c.ref_slot, job2.c.cur_slot = 2, 3
    for T in (0, 64):                            # 64: the shape that reads the tiled mirror and keeps the chi2 terms in HBM planes
        gpu_ctx.set_launch_shapes(align_threads=T)
        job.c.ref_slot, job.c.cur_slot = 0, 1
        a = gpu_ctx.sparse_align(job)
        b = gpu_ctx.sparse_align(job2)
        assert np.array_equal(a.T, b.T) and a.n_meas == b.n_meas
    levels, iters = gpu_ctx.align_work()
    assert 0 < levels <= iters and 0 < gpu_ctx.align_work_points() <= iters      # (point patch-iterations whose terms went to HBM)
    gpu_ctx.set_launch_shapes(align_threads=0)
    ms, launches = gpu_ctx.kernel_time(1)        # PLSVO_K_ALIGN_LEVEL: the fused alignment launch
    assert launches >= 5 and ms > 0.0
    gpu_ctx.set_profiling(False)
    # the result poses where a device-side consumer reads them
    keep2, d_dst, back = _device_bytes(np.zeros(7 * 8, np.uint8))
    gpu_ctx.align_copy_poses(d_dst)
    gpu_ctx.synchronize()
    assert np.array_equal(back().view(np.float64), b.T)
    name, cus, mem = gpu_ctx.device_info()
    assert cus > 0 and mem > 0 and name


@pytest.mark.gpu
def test_gather_poses_over_a_single_rank_rccl_communicator(P, gpu_ctx):
    """plsvo_gather_poses (C ABI, ncclAllGather on the ctx stream) with a world of one rank: the only configuration a
    1-GPU box can run.  The communicator comes straight from librccl through ctypes, as a C host would create it."""
    import ctypes as C
    import torch
    # by SONAME: the instance the dynamic loader already bound libplsvo_hip.so to (torch ships its own librccl.so.1 and
    # loads it first; a communicator made by a second copy of the library would be rejected as an invalid argument)
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        local = (torch.arange(5 * 96, dtype=torch.int32, device="cuda:0") % 251).to(torch.uint8).reshape(5, 96)   # five 96-byte records
        out = torch.zeros_like(local)
        torch.cuda.synchronize()
        gpu_ctx.gather_poses(comm.value, local.data_ptr(), 5, out.data_ptr())
        gpu_ctx.synchronize()
        assert torch.equal(out, local)
        with pytest.raises(P.capi.PlsvoError):
            gpu_ctx.gather_poses(None, local.data_ptr(), 5, out.data_ptr())
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


@pytest.mark.gpu
def test_config4_sharded_streams_and_pose_gather(P, ob):
    """BASELINE configs[3] on the hardware at hand: 64 seeded 640x480 streams (200 points + 80 segments, levels 3..1) as eight shards of
    8 streams -- eight plsvo_ctx on one device standing in for the eight ranks, seeds by pl-svo_amd/dist.py::rank_seeds exactly as
    bench.py shards them -- each shard running plsvo_align_run + plsvo_poseopt_run on its own streams and publishing its pose block
    through plsvo_gather_poses (a one-rank RCCL communicator per shard: the transport over xGMI is what a 1-GPU box cannot show).
    Every stream is checked against the oracle, the assembled table against the concatenation of the shards' results."""
    import torch
    D = P.dist
    world, B, W, H = 8, 8, 640, 480
    dev = torch.device("cuda", 0)
    table = torch.zeros((world * B, 96), dtype=torch.uint8, device=dev)      # world * B plsvo_pose_record
    local = torch.zeros((B, 96), dtype=torch.uint8, device=dev)
    align_T = np.zeros((world * B, 7))
    stream = torch.cuda.Stream(dev)
    ctxs, comms = [], []
    try:
        with torch.cuda.stream(stream):
            for r in range(world):
                c = P.capi.Context(0, stream=stream.cuda_stream)
                ctxs.append(c)
                comm = P.rccl.comm_init(1, 0, P.rccl.unique_id())
                comms.append(comm)
                seeds = D.rank_seeds(r, world, B)
                assert seeds == [1234 + r * B + i for i in range(B)]
                streams = [P.synth.make_align_stream(s_, W, H, 200, 80, max_level=3) for s_ in seeds]
                imgs = P.synth.render_streams(streams, device=dev)
                c.config_pyramids(2 * B, W, H, 4)
                c.build_pyramids_dev(0, 2 * B, imgs.data_ptr(), W, W * H, 0)
                jobs = [P.align_job_from_stream(s_, 3, 1, ref_slot=2 * i, cur_slot=2 * i + 1) for i, s_ in enumerate(streams)]
                frames = [P.synth.make_poseopt_frame(s_, 200, 80, W, H) for s_ in seeds]
                pjobs = [P.poseopt_job_from_frame(f) for f in frames]
                c.align_stage(jobs)
                c.poseopt_stage(pjobs)
                c.align_run()
                c.poseopt_run()
                # the shard's block of the table: its streams' 96-byte records packed on the device, then all-gathered
                assert c.pack_pose_records(local.data_ptr()) == B
                c.gather_poses(comm, local.data_ptr(), B, table.data_ptr() + r * B * 96)
                c.synchronize()
                ares, pres = c.align_fetch(), c.poseopt_fetch()
                pyrs = [(c.download_pyramid(2 * i), c.download_pyramid(2 * i + 1)) for i in range(B)]
                for i in range(B):
                    g = r * B + i
                    ro, _ = ob.sparse_align(jobs[i], pyrs[i][0], pyrs[i][1])
                    ang, tr, ok = Hh.pose_close(Hh.frame_pose(ares[i].T, streams[i]), Hh.frame_pose(ro.T, streams[i]))
                    ang2, tr2, ok2 = Hh.pose_close(ares[i].T, ro.T)
                    assert ok and ok2, f"stream {g}: align rot {ang:.2e}/{ang2:.2e} trans {tr:.2e}/{tr2:.2e}"
                    assert np.array_equal(ares[i].seg_alive, ro.seg_alive), g
                    po, _ = ob.pose_optimize(pjobs[i])
                    ang3, tr3, ok3 = Hh.pose_close(pres[i].T, po.T)
                    assert ok3 and ang3 < 1e-9, f"stream {g}: pose-opt rot {ang3:.2e} trans {tr3:.2e}"
                    assert np.array_equal(pres[i].pt_keep, po.pt_keep) and np.array_equal(pres[i].seg_keep, po.seg_keep), g
                    align_T[g] = ares[i].T
                    # the gathered record IS what the shard fetched: pose, SparseImgAlign::run's return value, the surviving observations
                    rec = D.tensor_to_records(table[g:g + 1])[0]
                    assert np.array_equal(rec["T_f_w"], np.asarray(pres[i].T)), g
                    assert int(rec["n_tracked"]) == int(ares[i].n_tracked) == int(ro.n_tracked) and int(rec["stream"]) == i, g
                    assert int(rec["num_obs_pt"]) == int(pres[i].num_obs_pt) == int(po.num_obs_pt), g
                    assert int(rec["num_obs_ls"]) == int(pres[i].num_obs_ls) == int(po.num_obs_ls) and rec["error_final"] == pres[i].error_final, g
                    assert int(rec["status"]) == (P.abi.REC_ALIGN | P.abi.REC_POSEOPT), g
        torch.cuda.synchronize()
        # rank-major concatenation: block r of the table holds shard r's streams, in seed order, nothing else touched
        recs = D.tensor_to_records(table)
        t = recs["T_f_w"]
        assert np.all(np.isfinite(t)) and np.all(np.abs(np.linalg.norm(t[:, :4], axis=1) - 1.0) < 1e-12)
        assert D.lost_streams(recs) == [] and list(recs["stream"]) == list(range(B)) * world
        assert D.lost_streams(recs, min_tracked=10 ** 6) == list(range(world * B))   # the thresholds the host applies (frame_handler_mono.cpp:272-274)
    finally:
        for comm in comms:
            P.rccl.comm_destroy(comm)
        for c in ctxs:
            c.close()


@pytest.mark.gpu
def test_bench_distributed_branch_with_one_rank():
    """bench.py's N>1 code path -- process group, RCCL communicator from the rendezvous, pose copy + plsvo_gather_poses every step,
    the stale-block check -- with a single rank (--dist-selftest), on BASELINE configs[3]'s shards."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "4", "--steps", "3", "--warmup", "1", "--dist-selftest",
                        "--no-cpu-baseline", "--no-latency"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["config"]["shards"] == 8 and d["config"]["global_batch"] == 64 and d["value"] > 0
    assert "dist-selftest" in d["config"]["parallelism"] and len(d["per_shard"]["us_per_step"]) == 8


@pytest.mark.gpu
def test_ldlt_flavour_option(P, ob, gpu_ctx):
    """plsvo_hip_set_option(PLSVO_OPT_LDLT_FLAVOUR): the Eigen 3.1...3.2.1 rule (320, default) and the Eigen >= 3.2.2 rule (330) are the
    same arithmetic on full-rank systems -- bit-identical device results -- and each follows the oracle's restatement of the same rule
    where they differ in a reproducible way: an all-zero system (no visible feature) and a one-observation system under 320."""
    st, ref, cur, job = Hh.make_case(ob, 1235, 640, 480, 200, 80, 4, 3, 1)
    gpu_ctx.config_pyramids(2, 640, 480, 4)
    gpu_ctx.upload_pyramid(0, ref)
    gpu_ctx.upload_pyramid(1, cur)
    gpu_ctx.align_set_trace(0)
    fr = P.synth.make_poseopt_frame(77, 120, 40)
    pj = P.poseopt_job_from_frame(fr)
    out = {}
    try:
        for flavour in (320, 330):
            gpu_ctx.set_ldlt_flavour(flavour)
            out[flavour] = (gpu_ctx.sparse_align(job), gpu_ctx.pose_optimize(pj))
        assert np.array_equal(out[320][0].T, out[330][0].T) and out[320][0].iters_per_level == out[330][0].iters_per_level
        assert np.array_equal(out[320][1].T, out[330][1].T) and np.array_equal(out[320][1].cov, out[330][1].cov)
        with pytest.raises(P.capi.PlsvoError):
            gpu_ctx.set_ldlt_flavour(321)
        # one point observation: rank 2.  Under 320 the unobservable directions come back as exact zeros, like the oracle's
        fr1 = P.synth.make_poseopt_frame(78, 1, 0)
        pj1 = P.poseopt_job_from_frame(fr1)
        gpu_ctx.set_ldlt_flavour(320)
        ob.set_ldlt_flavour(320)
        d1, o1 = gpu_ctx.pose_optimize(pj1), ob.pose_optimize(pj1)[0]
        assert Hh.pose_close(d1.T, o1.T)[2]
        gpu_ctx.set_ldlt_flavour(330)
        d2 = gpu_ctx.pose_optimize(pj1)          # 330: rounding residue divided by rounding residue -- only required not to fault
        assert d2.iters >= 1
    finally:
        gpu_ctx.set_ldlt_flavour(320)
        ob.set_ldlt_flavour(320)


@pytest.mark.gpu
def test_robust_weight_equals_the_reference_quotient_for_every_float():
    """The photometric robust weight of a point pixel, (float)(1.0 / (1.0 + (double)|res|)) (src/sparse_img_align.cpp:479), as the
    alignment kernel computes it (pl-svo_amd/csrc/robust_weight.hpp: the IEEE double quotient's own fma sequence without the scaling
    wrapper) against the device's IEEE division for EVERY float in [0, 256] -- 1 132 462 081 bit patterns -- and, on a stride, against
    the host's division: no mismatch.  (tools/robust_weight_exhaustive.hip, built by __graft_entry__.build().)"""
    import json, os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tools", "robust_weight_exhaustive")
    if not os.path.exists(exe):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", exe + ".hip", "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert out.returncode == 0 and d["inputs"] == 1132462081, out.stdout + out.stderr
    assert d["robust_weight_f64_mismatches"] == 0 and d["device_division_vs_host_division_mismatches"] == 0, d
