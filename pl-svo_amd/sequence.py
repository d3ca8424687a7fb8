"""A minimal frame-to-frame harness over the widened hot path, in the order FrameHandlerMono::processFrame runs it
(src/frame_handler_mono.cpp:266-340):

    sparse image alignment (previous frame -> new frame)        plsvo_sparse_align
    reprojection of the map into the new frame                  plsvo_reproject
    direct feature matching against the keyframe observations   plsvo_match_direct
    motion-only pose optimisation on the matches                plsvo_pose_optimize
    trajectory record (TUM line of T_f_w^-1)                    plsvo_trajectory_record

on a synthetic sequence with a known map (a textured plane carrying point and line-segment landmarks, observed in
keyframe 0).  It exists to exercise the entry points chained the way the reference chains them and to emit trajectory
files in the reference harness's format; it is not a VO system: no feature detection, depth filter,
keyframe selection or map maintenance.

`backend` is duck-typed: load_frames(list of level-0 images), sparse_align(job), reproject(job), match_direct(job),
pose_optimize(job).  The product backend is HipBackend (C ABI on the GPU, no fallback); tests pass an oracle-backed one
to check the whole chain end to end."""
import copy
import math

import numpy as np

from . import abi, synth


class HipBackend:
    def __init__(self, ctx, n_levels=4):
        self.ctx, self.n_levels = ctx, n_levels

    def load_frames(self, images):
        h, w = images[0].shape
        self.ctx.config_pyramids(len(images), w, h, self.n_levels)
        for k, img in enumerate(images):
            self.ctx.build_pyramid(k, img, 0)

    def sparse_align(self, job):
        return self.ctx.sparse_align(job)

    def reproject(self, job):
        return self.ctx.reproject(job)

    def match_direct(self, job):
        return self.ctx.match_direct(job)

    def pose_optimize(self, job):
        return self.ctx.pose_optimize(job)


def make_sequence(seed, n_frames=6, W=320, H=240, n_pts=120, n_seg=30, step_scale=0.5):
    """Camera moving smoothly over the textured plane of synth.make_align_stream.  Returns a dict with the level-0
    images, the camera, the true poses T_f_w of every frame and the map: landmark positions with their keyframe-0
    observations."""
    st = synth.make_align_stream(seed, W, H, n_pts, n_seg, 3, motion_scale=step_scale)
    rng = np.random.default_rng(seed + 300000)
    d0 = st.plane_d / st.plane_n[2]
    xi = np.concatenate([rng.uniform(-0.03, 0.03, 3) * d0, rng.uniform(-0.01, 0.01, 3)]) * step_scale
    subs, T_rel = [], []
    for k in range(1, n_frames):
        s = copy.copy(st)
        s.T_true = synth.se3_exp(k * xi)        # frame k from frame 0
        subs.append(s)
        T_rel.append(s.T_true)
    imgs = synth.render_streams([st] + subs).numpy()
    images = [imgs[0, 0]] + [imgs[k, 1] for k in range(1, n_frames)]
    poses = [st.T_ref_w] + [synth.se3_mul(T, st.T_ref_w) for T in T_rel]
    return dict(images=images, cam=st.cam, poses_true=np.stack(poses), stream=st,
                pt_pos=st.pt_pos_w, pt_px0=st.pt_px, pt_f0=st.pt_f,
                seg_spos=st.seg_spos_w, seg_epos=st.seg_epos_w, seg_spx0=st.seg_spx, seg_epx0=st.seg_epx, seg_sf0=st.seg_sf, seg_ef0=st.seg_ef)


def _bearing(cam, px):
    fx, fy, cx, cy = cam[:4]
    r = np.stack([(px[:, 0] - cx) / fx, (px[:, 1] - cy) / fy, np.ones(len(px))], axis=1)
    return r / np.linalg.norm(r, axis=1, keepdims=True)


def run_sequence(backend, seq, max_level=3, min_level=1, n_pyr_levels=3, reproj_thresh=2.0):
    """-> list of per-frame dicts (pose T_f_w, cov, counts).  Frame 0 is the keyframe with the true pose."""
    cam = seq["cam"]
    backend.load_frames(seq["images"])
    n_pts, n_seg = len(seq["pt_pos"]), len(seq["seg_spos"])
    T_prev = seq["poses_true"][0].copy()
    # features of the previous frame that still carry a landmark: index into the map + pixel position
    prev = dict(pt_idx=np.arange(n_pts), pt_px=seq["pt_px0"].copy(), seg_idx=np.arange(n_seg), seg_spx=seq["seg_spx0"].copy(),
                seg_epx=seq["seg_epx0"].copy())
    out = [dict(T=T_prev.copy(), cov=np.full((6, 6), 1e-9), n_align=0, n_matched_pt=n_pts, n_matched_seg=n_seg)]
    kf_T = seq["poses_true"][0]
    for k in range(1, len(seq["images"])):
        # ---- 1. sparse image alignment, previous frame -> frame k (processFrame :266-274) ----
        ref_pos = synth.se3_inv(T_prev)[4:]

        def scaled(px, pos):
            return _bearing(cam, px) * np.linalg.norm(pos - ref_pos, axis=1)[:, None]        # f * |pos - ref_pos| (:229-230)
        pi, si = prev["pt_idx"], prev["seg_idx"]
        job = abi.AlignJob(cam, max_level, min_level, 30, 1e-6, synth.se3_mul(T_prev, synth.se3_inv(T_prev)), prev["pt_px"],
                           scaled(prev["pt_px"], seq["pt_pos"][pi]), prev["seg_spx"], prev["seg_epx"],
                           np.linalg.norm(prev["seg_epx"] - prev["seg_spx"], axis=1), scaled(prev["seg_spx"], seq["seg_spos"][si]),
                           scaled(prev["seg_epx"], seq["seg_epos"][si]), ref_slot=k - 1, cur_slot=k)
        ar = backend.sparse_align(job)
        T_k = synth.se3_mul(ar.T, T_prev)                                                    # :92
        # ---- 2. reprojection of the whole map (Reprojector::reprojectMap) ----
        pos_all = np.concatenate([seq["pt_pos"], seq["seg_spos"], seq["seg_epos"]])
        rp = backend.reproject(abi.ReprojectJob(cam, np.stack([kf_T, T_k]), np.ones(len(pos_all), np.int32), pos_all, cell_size=30))
        vis = rp["cell"] >= 0
        seg_vis = vis[n_pts:n_pts + n_seg] & vis[n_pts + n_seg:]
        vis[n_pts:n_pts + n_seg] = seg_vis
        vis[n_pts + n_seg:] = seg_vis
        idx = np.nonzero(vis)[0]
        # ---- 3. direct matching against the keyframe-0 observations (Matcher::findMatchDirect) ----
        ref_px = np.concatenate([seq["pt_px0"], seq["seg_spx0"], seq["seg_epx0"]])[idx]
        ref_f = np.concatenate([seq["pt_f0"], seq["seg_sf0"], seq["seg_ef0"]])[idx]
        m = len(idx)
        mj = abi.MatchJob(cam, np.stack([kf_T, T_k]), np.array([0, k], np.int32), np.ones(m, np.int32), np.zeros(m, np.int32), ref_px, ref_f,
                          np.zeros(m, np.int32), np.zeros(m, np.uint8), np.zeros((m, 2)), pos_all[idx], rp["px"][idx], n_pyr_levels, 10)
        mr = backend.match_direct(mj)
        found = np.zeros(len(pos_all), bool)
        found[idx] = mr["found"].astype(bool)
        px_new = rp["px"].copy()
        px_new[idx] = mr["px_cur"]
        level = np.zeros(len(pos_all), np.int32)
        level[idx] = np.maximum(mr["search_level"], 0)
        pt_ok = found[:n_pts]
        seg_ok = found[n_pts:n_pts + n_seg] & found[n_pts + n_seg:]
        # ---- 4. motion-only pose optimisation on the matches (processFrame :327-329) ----
        pt_i, seg_i = np.nonzero(pt_ok)[0], np.nonzero(seg_ok)[0]
        sf, ef = _bearing(cam, px_new[n_pts + seg_i]), _bearing(cam, px_new[n_pts + n_seg + seg_i])
        line = np.cross(sf, ef)
        line = line / np.sqrt(line[:, 0:1] ** 2 + line[:, 1:2] ** 2) if len(seg_i) else np.zeros((0, 3))    # feature.cpp:103-104
        pj = abi.PoseOptJob(T_k, abs(cam[0]), reproj_thresh, 10, _bearing(cam, px_new[pt_i]), seq["pt_pos"][pt_i], level[pt_i], line,
                            seq["seg_spos"][seg_i], seq["seg_epos"][seg_i], level[n_pts + seg_i])
        pr = backend.pose_optimize(pj)
        T_k = pr.T.copy()
        pt_keep, seg_keep = pr.pt_keep.astype(bool), pr.seg_keep.astype(bool)
        prev = dict(pt_idx=pt_i[pt_keep], pt_px=px_new[pt_i[pt_keep]], seg_idx=seg_i[seg_keep],
                    seg_spx=px_new[n_pts + seg_i[seg_keep]], seg_epx=px_new[n_pts + n_seg + seg_i[seg_keep]])
        T_prev = T_k
        out.append(dict(T=T_k.copy(), cov=pr.cov.copy(), n_align=ar.n_tracked, n_matched_pt=int(pt_ok.sum()), n_matched_seg=int(seg_ok.sum()),
                        n_kept_pt=int(pt_keep.sum()), n_kept_seg=int(seg_keep.sum())))
    return out


def pose_errors(result, seq):
    """(rotation error [rad], translation error [m]) per frame against the true poses"""
    return [synth.se3_log_angle_dist(r["T"], T) for r, T in zip(result, seq["poses_true"])]
