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

With mapping=True two more steps run per frame, again as processFrame / the depth-filter thread order them:

    structure optimisation of the 20 least recently refined landmarks   plsvo_structure_optimize   (:340)
    depth-filter update of the seeds with the new frame                  plsvo_update_seeds         (depth_filter.cpp:262)

starting from a map that knows only part of the landmarks (with noisy positions) and holds the rest as seeds that turn
into landmarks when they converge.

`backend` is duck-typed: load_frames(list of level-0 images), sparse_align(job), reproject(job), match_direct(job),
pose_optimize(job), and for mapping structure_optimize(job), update_seeds(job).  The product backend is HipBackend (C ABI on the GPU, no fallback); tests pass an oracle-backed one
to check the whole chain end to end."""
import copy

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

    def align_ties(self):
        """Gauss-Newton iterations of the LAST alignment whose accept / roll-back decision was taken on the exact float chi2 sums (a frame
        that met none cannot have left its path on a last-bit tie)"""
        return int(self.ctx.align_chi2_ties()[1])

    def reproject(self, job):
        return self.ctx.reproject(job)

    def match_direct(self, job):
        return self.ctx.match_direct(job)

    def pose_optimize(self, job):
        return self.ctx.pose_optimize(job)

    def structure_optimize(self, job):
        return self.ctx.structure_optimize(job)

    def update_seeds(self, job):
        return self.ctx.update_seeds(job)


class HipChainBackend(HipBackend):
    """The same product backend with steps 1-4 of a frame -- alignment, reprojection, matching, pose optimisation -- as ONE resident
    call (plsvo_frame_step_batch): poses, candidates, matches and keep masks stay in HBM between the kernels."""

    def frame_step(self, job, T_prev, kf_T, kf_slot, n_pt, n_seg, pos_all, ref_px, ref_f, active, cam, n_pyr_levels, reproj_thresh):
        cj = abi.ChainJob(job, T_prev, kf_T, kf_slot, n_pt, n_seg, pos_all, ref_px, ref_f, active=active)
        return self.ctx.frame_step_batch([cj], cam, n_pyr_levels=n_pyr_levels, reproj_thresh=reproj_thresh)[0]


def make_sequence(seed, n_frames=6, W=320, H=240, n_pts=120, n_seg=30, step_scale=0.5, total=None):
    """Camera moving smoothly over the textured plane of synth.make_align_stream.  Returns a dict with the level-0
    images, the camera, the true poses T_f_w of every frame and the map: landmark positions with their keyframe-0
    observations.  Per-frame motion is step_scale x (3 % of the scene depth, 0.01 rad) in a random direction, or -- with
    `total` -- whatever divides a whole-sequence motion of total x (scene depth, 0.25 rad) into n_frames - 1 equal steps
    (long sequences must keep the keyframe's landmarks in view)."""
    st = synth.make_align_stream(seed, W, H, n_pts, n_seg, 3, motion_scale=step_scale)
    rng = np.random.default_rng(seed + 300000)
    d0 = st.plane_d / st.plane_n[2]
    xi = np.concatenate([rng.uniform(-0.03, 0.03, 3) * d0, rng.uniform(-0.01, 0.01, 3)]) * step_scale
    if total is not None:
        dirs = rng.uniform(-1.0, 1.0, 6)
        xi = np.concatenate([dirs[:3] / np.linalg.norm(dirs[:3]) * total * d0, dirs[3:] / np.linalg.norm(dirs[3:]) * total * 0.25]) / max(n_frames - 1, 1)
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


def run_sequence(backend, seq, max_level=3, min_level=1, n_pyr_levels=3, reproj_thresh=2.0, mapping=False, known_frac=0.6,
                 pos_noise=0.005, map_seed=0, kf_every=5):
    """-> list of per-frame dicts (pose T_f_w, cov, counts).  Frame 0 is the keyframe with the true pose.
    mapping=True: only `known_frac` of the point landmarks start in the map (positions off by `pos_noise` x depth along
    their viewing ray), the others are depth-filter seeds; the seed update runs every frame; every `kf_every`-th frame
    plays the keyframe: its matches become observations of their landmarks (Feature3D::obs_ grows at keyframes only, as
    in the reference) and the 20 least recently refined landmarks with >= 2 observations are structure-optimised."""
    cam = seq["cam"]
    backend.load_frames(seq["images"])
    n_pts, n_seg = len(seq["pt_pos"]), len(seq["seg_spos"])
    T_prev = seq["poses_true"][0].copy()
    kf_T = seq["poses_true"][0]
    P3 = seq["pt_pos"].copy()                      # the map's point positions (seq["pt_pos"] stays the truth)
    known = np.ones(n_pts, bool)
    if mapping:
        rng = np.random.default_rng(map_seed + 77)
        known = rng.uniform(size=n_pts) < known_frac
        kf_pos = synth.se3_inv(kf_T)[4:]
        ray = P3 - kf_pos
        P3[known] = (kf_pos + ray * (1.0 + rng.uniform(-pos_noise, pos_noise, n_pts))[:, None])[known]
        depth0 = np.linalg.norm(ray, axis=1)
        dmean, dmin = float(depth0[known].mean()), 0.7 * float(depth0[known].min())
        nsd = int((~known).sum())
        seeds = dict(idx=np.nonzero(~known)[0], a=np.full(nsd, 10.0, np.float32), b=np.full(nsd, 10.0, np.float32), mu=np.full(nsd, 1.0 / dmean, np.float32),
                     z_range=np.full(nsd, 1.0 / dmin, np.float32), sigma2=np.full(nsd, (1.0 / dmin) ** 2 / 36.0, np.float32))
        obs = {int(i): [(0, seq["pt_f0"][i])] for i in range(n_pts)}     # Feature3D::obs_: (frame, unit bearing)
        last_optim = np.zeros(n_pts, np.int64)
        poses_est = [kf_T.copy()]
    # features of the previous frame that still carry a landmark: index into the map + pixel position
    k0 = np.nonzero(known)[0]
    prev = dict(pt_idx=k0, pt_px=seq["pt_px0"][k0].copy(), seg_idx=np.arange(n_seg), seg_spx=seq["seg_spx0"].copy(),
                seg_epx=seq["seg_epx0"].copy())
    out = [dict(T=T_prev.copy(), cov=np.full((6, 6), 1e-9), n_align=0, n_matched_pt=int(known.sum()), n_matched_seg=n_seg)]
    for k in range(1, len(seq["images"])):
        # ---- 1. sparse image alignment, previous frame -> frame k (processFrame :266-274) ----
        ref_pos = synth.se3_inv(T_prev)[4:]

        def scaled(px, pos):
            return _bearing(cam, px) * np.linalg.norm(pos - ref_pos, axis=1)[:, None]        # f * |pos - ref_pos| (:229-230)
        pi, si = prev["pt_idx"], prev["seg_idx"]
        job = abi.AlignJob(cam, max_level, min_level, 30, 1e-6, synth.se3_mul(T_prev, synth.se3_inv(T_prev)), prev["pt_px"],
                           scaled(prev["pt_px"], P3[pi]), prev["seg_spx"], prev["seg_epx"],
                           np.linalg.norm(prev["seg_epx"] - prev["seg_spx"], axis=1), scaled(prev["seg_spx"], seq["seg_spos"][si]),
                           scaled(prev["seg_epx"], seq["seg_epos"][si]), ref_slot=k - 1, cur_slot=k)
        pos_all = np.concatenate([P3, seq["seg_spos"], seq["seg_epos"]])
        if hasattr(backend, "frame_step"):
            # ---- 1-4 as one resident call: nothing but the final results comes back ----
            act = np.concatenate([known, np.ones(2 * n_seg, bool)]).astype(np.uint8)
            ref_px_all = np.concatenate([seq["pt_px0"], seq["seg_spx0"], seq["seg_epx0"]])
            ref_f_all = np.concatenate([seq["pt_f0"], seq["seg_sf0"], seq["seg_ef0"]])
            cr = backend.frame_step(job, T_prev, kf_T, 0, n_pts, n_seg, pos_all, ref_px_all, ref_f_all, act, cam, n_pyr_levels, reproj_thresh)
            ar, pr = cr.align, cr.pose
            px_new = cr.px
            pt_i, seg_i = cr.sel_pt.astype(np.int64), cr.sel_seg.astype(np.int64)
            pt_ok = np.zeros(n_pts, bool); pt_ok[pt_i] = True
            seg_ok = np.zeros(n_seg, bool); seg_ok[seg_i] = True
        else:
            ar = backend.sparse_align(job)
            T_k = synth.se3_mul(ar.T, T_prev)                                                    # :92
            # ---- 2. reprojection of the whole map (Reprojector::reprojectMap) ----
            rp = backend.reproject(abi.ReprojectJob(cam, np.stack([kf_T, T_k]), np.ones(len(pos_all), np.int32), pos_all, cell_size=30))
            vis = rp["cell"] >= 0
            vis[:n_pts] &= known                     # seeds are not in the map yet
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
            pj = abi.PoseOptJob(T_k, abs(cam[0]), reproj_thresh, 10, _bearing(cam, px_new[pt_i]), P3[pt_i], level[pt_i], line,
                                seq["seg_spos"][seg_i], seq["seg_epos"][seg_i], level[n_pts + seg_i])
            pr = backend.pose_optimize(pj)
        n_ties = backend.align_ties() if hasattr(backend, "align_ties") else None
        T_k = pr.T.copy()
        pt_keep, seg_keep = pr.pt_keep.astype(bool), pr.seg_keep.astype(bool)
        prev = dict(pt_idx=pt_i[pt_keep], pt_px=px_new[pt_i[pt_keep]], seg_idx=seg_i[seg_keep],
                    seg_spx=px_new[n_pts + seg_i[seg_keep]], seg_epx=px_new[n_pts + n_seg + seg_i[seg_keep]])
        T_prev = T_k
        rec = dict(T=T_k.copy(), cov=pr.cov.copy(), n_align=ar.n_tracked, n_matched_pt=int(pt_ok.sum()), n_matched_seg=int(seg_ok.sum()),
                   n_kept_pt=int(pt_keep.sum()), n_kept_seg=int(seg_keep.sum()))
        if n_ties is not None:
            rec["align_ties"] = n_ties
        if mapping:
            poses_est.append(T_k.copy())
            kept = pt_i[pt_keep]
            is_kf = (k % kf_every) == 0
            if is_kf:
                for i, brg in zip(kept, _bearing(cam, px_new[kept])):
                    obs[int(i)].append((k, brg))
            # ---- 5. structure optimisation (FrameHandlerBase::optimizeStructure: the 20 least recently refined, :202-237) ----
            cand = np.array([i for i in kept if len(obs[int(i)]) >= 2], np.int64)
            if len(cand) and is_kf:
                sel = cand[np.argsort(last_optim[cand], kind="stable")[:20]]
                off, ofr, of_ = [0], [], []
                for i in sel:
                    for fr_, brg in obs[int(i)]:
                        ofr.append(fr_)
                        of_.append(brg)
                    off.append(len(ofr))
                z3, zi = np.zeros((0, 3)), np.zeros(0, np.int32)
                so = backend.structure_optimize(abi.StructOptJob(np.stack(poses_est), P3[sel], off, ofr, np.array(of_), z3, z3, np.zeros(1, np.int32), zi, z3, z3, 5, 5))
                P3[sel] = so["pt_pos"]
                last_optim[sel] = k
            # ---- 6. depth-filter update of the seeds with this frame (DepthFilter::updateSeeds) ----
            ns = len(seeds["idx"])
            if ns:
                si_ = seeds["idx"]
                ptd = dict(ref_frame=np.zeros(ns, np.int32), cur_frame=np.ones(ns, np.int32), px=seq["pt_px0"][si_], f=seq["pt_f0"][si_], level=np.zeros(ns, np.int32),
                           a=seeds["a"], b=seeds["b"], mu=seeds["mu"], z_range=seeds["z_range"], sigma2=seeds["sigma2"])
                sr = backend.update_seeds(abi.SeedsJob(cam, np.stack([kf_T, T_k]), np.array([0, k], np.int32), ptd, None, n_pyr_levels=n_pyr_levels))
                stt = sr["pt_status"]
                conv = stt == abi.SEED_CONVERGED
                P3[si_[conv]] = sr["pt_xyz_world"][conv]
                known[si_[conv]] = True
                keep_s = ~(conv | (stt == abi.SEED_NAN))
                seeds = dict(idx=si_[keep_s], a=sr["pt_a"][keep_s], b=sr["pt_b"][keep_s], mu=sr["pt_mu"][keep_s], z_range=seeds["z_range"][keep_s],
                             sigma2=sr["pt_sigma2"][keep_s])
                rec["n_seed_converged"] = int(conv.sum())
            rec.update(n_known=int(known.sum()), n_seeds=len(seeds["idx"]),
                       landmark_err=float(np.median(np.linalg.norm(P3[known] - seq["pt_pos"][known], axis=1))))
        out.append(rec)
    return out


def pose_errors(result, seq):
    """(rotation error [rad], translation error [m]) per frame against the true poses"""
    return [synth.se3_log_angle_dist(r["T"], T) for r, T in zip(result, seq["poses_true"])]
