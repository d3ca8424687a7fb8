"""Direct feature matching (hot-path contract row (f) "next" #2): plsvo::Matcher::findMatchDirect for points and line
segments (src/matcher.cpp:159-275) with warp::* (:40-128) and feature_alignment::align1D/align2D
(src/feature_alignment.cpp:41-290).
CPU: the C oracle against an independent NumPy (np.float32 / np.float64 scalar) restatement and known answers.
GPU: the HIP kernel is compiled without fma contraction and must reproduce the oracle BIT FOR BIT."""
import math

import numpy as np
import pytest

import np_restatement as npr

f32, f64 = np.float32, np.float64


def _frames(P, ob, st, n_levels=4):
    imgs = P.synth.render_streams([st]).numpy()[0]
    return [ob.build_pyramid(imgs[0], n_levels), ob.build_pyramid(imgs[1], n_levels)]


# ---- NumPy restatement (written from the reference source, not from oracle/plsvo_oracle.c) ----------------------

def _np_warp_matrix(cam, px_ref, f_ref, depth, T_cur_ref, level):
    fx, fy, cx, cy = cam[:4]

    def cam2world(u, v):
        r = np.array([(u - cx) / fx, (v - cy) / fy, 1.0])
        return r / np.linalg.norm(r)

    def world2cam(p):
        return np.array([fx * p[0] / p[2] + cx, fy * p[1] / p[2] + cy])

    xyz = np.asarray(f_ref) * depth
    du = cam2world(px_ref[0] + 5.0 * (1 << level), px_ref[1])
    dv = cam2world(px_ref[0], px_ref[1] + 5.0 * (1 << level))
    du = du * (xyz[2] / du[2])
    dv = dv * (xyz[2] / dv[2])
    pc, pdu, pdv = (world2cam(npr.se3_act(T_cur_ref, p)) for p in (xyz, du, dv))
    A = np.zeros((2, 2))
    A[:, 0] = (pdu - pc) / 5
    A[:, 1] = (pdv - pc) / 5
    return A


def _np_interp(img, u, v):
    x, y = int(math.floor(u)), int(math.floor(v))
    sx, sy = f32(u - f32(x)), f32(v - f32(y))
    one = f32(1.0)
    w00 = f32((one - sx) * (one - sy))
    w01 = f32((one - sx) * sy)
    w10 = f32(sx * (one - sy))
    w11 = f32(f32(f32(one - w00) - w01) - w10)
    return f32(f32(f32(w00 * f32(img[y, x]) + w01 * f32(img[y + 1, x])) + w10 * f32(img[y, x + 1])) + w11 * f32(img[y + 1, x + 1]))


def _np_warp_affine(A, img, px_ref, level, search_level):
    det = A[0, 0] * A[1, 1] - A[1, 0] * A[0, 1]
    invdet = 1.0 / det
    a = np.array([[f32(A[1, 1] * invdet), f32(-A[0, 1] * invdet)], [f32(-A[1, 0] * invdet), f32(A[0, 0] * invdet)]], dtype=f32)
    rows, cols = img.shape
    r = np.array([f32(px_ref[0]) / f32(1 << level), f32(px_ref[1]) / f32(1 << level)], dtype=f32)
    patch = np.zeros((10, 10), np.uint8)
    for y in range(10):
        for x in range(10):
            p0, p1 = f32(f32(x - 5) * f32(1 << search_level)), f32(f32(y - 5) * f32(1 << search_level))
            q0 = f32(f32(a[0, 0] * p0 + a[0, 1] * p1) + r[0])
            q1 = f32(f32(a[1, 0] * p0 + a[1, 1] * p1) + r[1])
            if q0 < 0 or q1 < 0 or q0 >= cols - 1 or q1 >= rows - 1:
                patch[y, x] = 0
            else:
                patch[y, x] = int(_np_interp(img, q0, q1))      # (uint8_t) truncation
    return patch


def _np_weights(u, v, ui, vi):
    su, sv = f32(u - f32(ui)), f32(v - f32(vi))
    wTL = f32((1.0 - f64(su)) * (1.0 - f64(sv)))
    wTR = f32(f64(su) * (1.0 - f64(sv)))
    wBL = f32((1.0 - f64(su)) * f64(sv))
    wBR = f32(su * sv)
    return wTL, wTR, wBL, wBR


def _np_align2d(cur, pb, n_iter, est):
    rows, cols = cur.shape
    pbi = pb.astype(np.int64)
    dx = np.zeros((8, 8), f32)
    dy = np.zeros((8, 8), f32)
    H = np.zeros((3, 3), f32)
    for y in range(8):
        for x in range(8):
            J = np.array([0.5 * (pbi[y + 1, x + 2] - pbi[y + 1, x]), 0.5 * (pbi[y + 2, x + 1] - pbi[y, x + 1]), 1.0], dtype=f32)
            dx[y, x], dy[y, x] = J[0], J[1]
            H += np.outer(J, J).astype(f32)
    cof = lambda i, j: f32(f32(H[(i + 1) % 3, (j + 1) % 3] * H[(i + 2) % 3, (j + 2) % 3]) - f32(H[(i + 1) % 3, (j + 2) % 3] * H[(i + 2) % 3, (j + 1) % 3]))
    c00, c10, c20 = cof(0, 0), cof(1, 0), cof(2, 0)
    with np.errstate(all="ignore"):
        det = f32(f32(c00 * H[0, 0]) + f32(f32(c10 * H[1, 0]) + f32(c20 * H[2, 0])))
        invdet = f32(f32(1.0) / det)
        Hinv = np.array([[c00 * invdet, c10 * invdet, c20 * invdet],
                         [cof(0, 1) * invdet, cof(1, 1) * invdet, cof(2, 1) * invdet],
                         [cof(0, 2) * invdet, cof(1, 2) * invdet, cof(2, 2) * invdet]], dtype=f32)
    mean_diff = f32(0)
    u, v = f32(est[0]), f32(est[1])
    e = [f64(est[0]), f64(est[1])]
    min_up = f32(0.03 * 0.03)
    converged, it = False, 0
    with np.errstate(all="ignore"):
        while it < n_iter:
            ur, vr = f32(e[0]), f32(e[1])
            if np.isnan(ur) or np.isnan(vr):
                break
            ui, vi = int(math.floor(ur)), int(math.floor(vr))
            if ui < 4 or vi < 4 or ui >= cols - 4 or vi >= rows - 4:
                break
            wTL, wTR, wBL, wBR = _np_weights(ur, vr, ui, vi)
            Jr = [f32(0), f32(0), f32(0)]
            for y in range(8):
                for x in range(8):
                    yy, xx = vi - 4 + y, ui - 4 + x
                    sp = f32(f32(f32(wTL * f32(cur[yy, xx]) + wTR * f32(cur[yy, xx + 1])) + wBL * f32(cur[yy + 1, xx])) + wBR * f32(cur[yy + 1, xx + 1]))
                    res = f32(f32(sp - f32(pb[y + 1, x + 1])) + mean_diff)
                    Jr[0] = f32(Jr[0] - f32(res * dx[y, x]))
                    Jr[1] = f32(Jr[1] - f32(res * dy[y, x]))
                    Jr[2] = f32(Jr[2] - res)
            up = [f32(f32(f32(Hinv[i, 0] * Jr[0]) + f32(Hinv[i, 1] * Jr[1])) + f32(Hinv[i, 2] * Jr[2])) for i in range(3)]
            u, v = f32(u + up[0]), f32(v + up[1])
            e = [f64(u), f64(v)]
            mean_diff = f32(mean_diff + up[2])
            it += 1
            if f32(f32(up[0] * up[0]) + f32(up[1] * up[1])) < min_up:
                converged = True
                break
    return converged, it, (f64(u), f64(v))


def _np_align1d(cur, pb, dirv, n_iter, est):
    rows, cols = cur.shape
    pbi = pb.astype(np.int64)
    d0, d1 = f32(dirv[0]), f32(dirv[1])
    dv = np.zeros((8, 8), f32)
    H = np.zeros((2, 2), f32)
    for y in range(8):
        for x in range(8):
            g = f32(f32(d0 * f32(pbi[y + 1, x + 2] - pbi[y + 1, x])) + f32(d1 * f32(pbi[y + 2, x + 1] - pbi[y, x + 1])))
            J = np.array([f32(0.5 * f64(g)), 1.0], dtype=f32)
            dv[y, x] = J[0]
            H += np.outer(J, J).astype(f32)
    with np.errstate(all="ignore"):
        det = f32(f32(H[0, 0] * H[1, 1]) - f32(H[1, 0] * H[0, 1]))
        invdet = f32(f32(1.0) / det)
        Hinv = np.array([[H[1, 1] * invdet, -H[0, 1] * invdet], [-H[1, 0] * invdet, H[0, 0] * invdet]], dtype=f32)
    mean_diff, chi2 = f32(0), f32(0)
    u, v = f32(est[0]), f32(est[1])
    up = [f32(0), f32(0)]
    min_up = f32(0.03 * 0.03)
    converged, it = False, 0
    with np.errstate(all="ignore"):
        while it < n_iter:
            if np.isnan(u) or np.isnan(v):
                break
            ui, vi = int(math.floor(u)), int(math.floor(v))
            if ui < 4 or vi < 4 or ui >= cols - 4 or vi >= rows - 4:
                break
            wTL, wTR, wBL, wBR = _np_weights(u, v, ui, vi)
            Jr = [f32(0), f32(0)]
            new_chi2 = f32(0)
            for y in range(8):
                for x in range(8):
                    yy, xx = vi - 4 + y, ui - 4 + x
                    sp = f32(f32(f32(wTL * f32(cur[yy, xx]) + wTR * f32(cur[yy, xx + 1])) + wBL * f32(cur[yy + 1, xx])) + wBR * f32(cur[yy + 1, xx + 1]))
                    res = f32(f32(sp - f32(pb[y + 1, x + 1])) + mean_diff)
                    Jr[0] = f32(Jr[0] - f32(res * dv[y, x]))
                    Jr[1] = f32(Jr[1] - res)
                    new_chi2 = f32(new_chi2 + f32(res * res))
            it += 1
            if it > 1 and new_chi2 > chi2:
                u, v = f32(u - up[0]), f32(v - up[1])
                break
            chi2 = new_chi2
            up = [f32(f32(Hinv[0, 0] * Jr[0]) + f32(Hinv[0, 1] * Jr[1])), f32(f32(Hinv[1, 0] * Jr[0]) + f32(Hinv[1, 1] * Jr[1]))]
            u, v = f32(u + f32(up[0] * d0)), f32(v + f32(up[0] * d1))
            mean_diff = f32(mean_diff + up[1])
            if f32(f32(up[0] * up[0]) + f32(up[1] * up[1])) < min_up:
                converged = True
                break
    return converged, it, (f64(u), f64(v))


def _np_find_match_direct(d, frames, i, n_pyr_levels=3, max_iter=10):
    cam = d["cam"]
    level = int(d["ref_level"][i])
    px = d["ref_px"][i]
    W, Hh = int(cam[4]), int(cam[5])
    ox, oy = int(px[0]) // (1 << level), int(px[1]) // (1 << level)
    if not (6 <= ox < W // (1 << level) - 6 and 6 <= oy < Hh // (1 << level) - 6):
        return False, -1, 0, tuple(d["px_cur"][i])
    T_ref, T_cur = d["frame_T"][d["ref_frame"][i]], d["frame_T"][d["cur_frame"][i]]
    q = np.array([-T_ref[0], -T_ref[1], -T_ref[2], T_ref[3]])
    T_ref_inv = np.concatenate([q, npr.q_rot(q, -T_ref[4:])])
    T_cur_ref = npr.se3_mul(T_cur, T_ref_inv)
    depth = np.linalg.norm(T_ref_inv[4:] - d["pos"][i])
    A = _np_warp_matrix(cam, px, d["ref_f"][i], depth, T_cur_ref, level)
    D = A[0, 0] * A[1, 1] - A[1, 0] * A[0, 1]
    sl = 0
    while D > 3.0 and sl < n_pyr_levels - 1:
        sl += 1
        D *= 0.25
    pb = _np_warp_affine(A, frames[d["ref_frame"][i]][level], px, level, sl)
    cur = frames[d["cur_frame"][i]][sl]
    est = (d["px_cur"][i][0] / (1 << sl), d["px_cur"][i][1] / (1 << sl))
    if d["ref_type"][i] == 1:
        dc = A @ d["ref_grad"][i]
        dc = dc / math.sqrt(dc[0] * dc[0] + dc[1] * dc[1])
        ok, it, e = _np_align1d(cur, pb, dc, max_iter, est)
    else:
        ok, it, e = _np_align2d(cur, pb, max_iter, est)
    return ok, sl, it, (e[0] * (1 << sl), e[1] * (1 << sl))


# ---- CPU tests ----------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("seed,zoom", [(1, 0.0), (2, 0.45)])
def test_oracle_matches_numpy_restatement(P, ob, seed, zoom):
    st, d = P.synth.make_match_batch(seed, 320, 240, n_pts=36, n_seg=6, zoom=zoom, edgelet_frac=0.4)
    frames = _frames(P, ob, st)
    res = ob.match_direct(P.match_job_from_batch(d), frames)
    n_exact = 0
    for i in range(len(d["ref_px"])):
        ok, sl, it, e = _np_find_match_direct(d, frames, i)
        assert sl == res["search_level"][i], i
        assert it == res["n_iter"][i] and bool(ok) == bool(res["found"][i]), (i, it, res["n_iter"][i])
        got = res["px_cur"][i]
        if np.isnan(got).any():
            assert np.isnan(e).any()
            continue
        # the affine warp's double arithmetic is written with numpy's own (differently ordered) products, so a pixel
        # of the warped patch may round differently once in a while: the float path is compared exactly when it does not
        assert np.allclose(got, e, rtol=0, atol=2e-3), (i, got, e)
        n_exact += int(got[0] == e[0] and got[1] == e[1])
    assert n_exact >= 0.9 * len(d["ref_px"]), n_exact


def test_matching_reduces_the_reprojection_error(P, ob):
    st, d = P.synth.make_match_batch(3, 640, 480, n_pts=150, n_seg=40, edgelet_frac=0.0)
    frames = _frames(P, ob, st)
    res = ob.match_direct(P.match_job_from_batch(d), frames)
    f = res["found"].astype(bool)
    e0 = np.linalg.norm(d["px_cur"] - d["px_true"], axis=1)
    e1 = np.linalg.norm(res["px_cur"] - d["px_true"], axis=1)
    assert f.mean() > 0.8
    assert np.median(e1[f]) < 0.5 * np.median(e0[f])
    assert set(np.unique(res["search_level"])) <= {-1, 0, 1, 2}


def test_known_answer_pure_image_shift(P, ob):
    """identical poses, current image = keyframe image shifted by (+2, -1) pixels: a corner must move there"""
    st, d = P.synth.make_match_batch(4, 320, 240, n_pts=40, n_seg=0, edgelet_frac=0.0, px_noise=0.0, levels=(0,), level_p=(1.0,))
    imgs = P.synth.render_streams([st]).numpy()[0]
    ref = imgs[0]
    cur = np.roll(np.roll(ref, 2, axis=1), -1, axis=0)
    frames = [ob.build_pyramid(ref, 4), ob.build_pyramid(cur, 4)]
    d["frame_T"] = np.stack([st.T_ref_w, st.T_ref_w])
    d["px_cur"] = d["ref_px"].copy()
    res = ob.match_direct(P.match_job_from_batch(d), frames)
    f = res["found"].astype(bool)
    assert f.mean() > 0.8 and np.all(res["search_level"] == 0)
    shift = res["px_cur"][f] - d["ref_px"][f]
    assert np.allclose(np.median(shift, axis=0), [2.0, -1.0], atol=0.05)


def test_edge_cases(P, ob):
    st, d = P.synth.make_match_batch(5, 320, 240, n_pts=8, n_seg=2)
    frames = _frames(P, ob, st)
    # reference observation too close to the border for its level: rejected, px_cur untouched (matcher.cpp:168-170)
    d["ref_px"][0] = [5.0, 100.0]
    d["ref_level"][1] = 3
    d["ref_px"][1] = [40.0, 100.0]          # 40/8 = 5 < 6
    # initial estimate outside the current image: align2D leaves at once, not found, position unchanged
    d["px_cur"][2] = [-50.0, 20.0]
    d["ref_type"][2] = 0
    res = ob.match_direct(P.match_job_from_batch(d), frames)
    assert res["found"][0] == 0 and res["search_level"][0] == -1 and np.array_equal(res["px_cur"][0], d["px_cur"][0])
    assert res["found"][1] == 0 and res["search_level"][1] == -1
    assert res["found"][2] == 0 and res["n_iter"][2] == 0
    assert np.allclose(res["px_cur"][2], np.float32(d["px_cur"][2] / (1 << res["search_level"][2])).astype(float) * (1 << res["search_level"][2]))
    # align_max_iter = 0: nothing is found, estimates only pass through float
    r0 = ob.match_direct(P.match_job_from_batch(d, 3, 0), frames)
    assert not r0["found"].any() and not r0["n_iter"].any()
    sc = (1 << r0["search_level"][3:8].astype(int))[:, None].astype(float)     # `float u = cur_px_estimate.x(); ... cur_px_estimate << u, v`
    assert np.array_equal(r0["px_cur"][3:8], (d["px_cur"][3:8] / sc).astype(np.float32).astype(float) * sc)
    # empty batch
    for k in ("cur_frame", "ref_frame", "ref_px", "ref_f", "ref_level", "ref_type", "ref_grad", "pos", "px_cur"):
        d[k] = d[k][:0]
    assert ob.match_direct(P.match_job_from_batch(d), frames)["found"].size == 0


# ---- GPU tests: bit-exact against the oracle -------------------------------------------------------------------------

def _gpu_frames(ctx, W, H, frames, slots=(0, 1)):
    ctx.config_pyramids(max(slots) + 1, W, H, 4)
    for s, fr in zip(slots, frames):
        ctx.build_pyramid(s, fr[0], 0)


def _assert_same(rd, ro):
    assert np.array_equal(rd["search_level"], ro["search_level"])
    assert np.array_equal(rd["n_iter"], ro["n_iter"])
    assert np.array_equal(rd["found"], ro["found"])
    nan_d, nan_o = np.isnan(rd["px_cur"]), np.isnan(ro["px_cur"])
    assert np.array_equal(nan_d, nan_o)
    assert np.array_equal(rd["px_cur"][~nan_d], ro["px_cur"][~nan_o]), np.nanmax(np.abs(rd["px_cur"] - ro["px_cur"]))


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(31, 640, 480, 150, 50, 0.0), (32, 640, 480, 300, 100, 0.45), (33, 320, 240, 64, 0, 0.2),
                                  (34, 752, 480, 0, 90, 0.0), (35, 640, 480, 2000, 600, 0.3)])
def test_hip_match_direct_is_bit_exact(P, ob, gpu_ctx, case):
    seed, W, H, npts, nseg, zoom = case
    st, d = P.synth.make_match_batch(seed, W, H, npts, nseg, zoom=zoom)
    frames = _frames(P, ob, st)
    _gpu_frames(gpu_ctx, W, H, frames)
    job = P.match_job_from_batch(d)
    ro = ob.match_direct(job, frames)
    rd = gpu_ctx.match_direct(job)
    _assert_same(rd, ro)
    assert ro["found"].mean() > 0.2


@pytest.mark.gpu
def test_hip_match_direct_edge_cases(P, ob, gpu_ctx):
    st, d = P.synth.make_match_batch(41, 320, 240, n_pts=70, n_seg=10)
    frames = _frames(P, ob, st)
    _gpu_frames(gpu_ctx, 320, 240, frames)
    d["ref_px"][0] = [5.0, 100.0]
    d["ref_level"][1] = 3
    d["ref_px"][1] = [40.0, 100.0]
    d["px_cur"][2] = [-50.0, 20.0]
    d["px_cur"][3] = [318.5, 238.5]
    d["px_cur"][4] = [4.0, 4.0]
    for iters in (10, 1, 0):
        job = P.match_job_from_batch(d, 3, iters)
        _assert_same(gpu_ctx.match_direct(job), ob.match_direct(job, frames))
    # flat images: singular Hessians, NaN updates
    flat = [ob.build_pyramid(np.full((240, 320), 77, np.uint8), 4)] * 2
    _gpu_frames(gpu_ctx, 320, 240, flat)
    job = P.match_job_from_batch(d)
    ro, rd = ob.match_direct(job, flat), gpu_ctx.match_direct(job)
    _assert_same(rd, ro)
    assert np.isnan(ro["px_cur"]).any()      # patches that lie fully inside the flat image have H = 0: NaN updates on both sides
    # empty batch
    for k in ("cur_frame", "ref_frame", "ref_px", "ref_f", "ref_level", "ref_type", "ref_grad", "pos", "px_cur"):
        d[k] = d[k][:0]
    assert gpu_ctx.match_direct(P.match_job_from_batch(d))["found"].size == 0


@pytest.mark.gpu
def test_hip_match_direct_rejects_bad_input(P, gpu_ctx):
    st, d = P.synth.make_match_batch(42, 320, 240, n_pts=4, n_seg=0)
    gpu_ctx.config_pyramids(2, 320, 240, 4)
    d["ref_frame"][0] = 7
    with pytest.raises(P.capi.PlsvoError):
        gpu_ctx.match_direct(P.match_job_from_batch(d))
    d["ref_frame"][0] = 0
    d["ref_level"][0] = 9
    with pytest.raises(P.capi.PlsvoError):
        gpu_ctx.match_direct(P.match_job_from_batch(d))
