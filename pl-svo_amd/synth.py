"""Deterministic synthetic inputs for the hot path: textured-plane frame pairs with
point and line-segment features for sparse image alignment, and noisy/outlier-contaminated
point/line observations for pose optimisation.

Per-stream parameters come from numpy's default_rng(seed) on the host (seed = 1234 + stream index by
convention); the images are rendered analytically (ray-plane intersection + sum-of-sinusoids texture)
with torch on whatever device is asked for, so a whole batch of streams can be rendered in HBM.
This is input generation only -- no part of the hot path lives here.
"""
import math
from dataclasses import dataclass, field

import numpy as np
import torch

# ------------------------------------------------------------------------------------------------
# tiny SE3 helpers (numpy, float64).  Pose = [qx qy qz qw tx ty tz], tangent = (upsilon, omega)
# ------------------------------------------------------------------------------------------------


def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz])


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def se3_exp(u):
    u = np.asarray(u, dtype=np.float64)
    ups, om = u[:3], u[3:]
    th = np.linalg.norm(om)
    if th < 1e-10:
        q = np.array([0.5 * om[0], 0.5 * om[1], 0.5 * om[2], 1.0])
        V = np.eye(3)
    else:
        q = np.concatenate([math.sin(0.5 * th) / th * om, [math.cos(0.5 * th)]])
        O = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
        V = np.eye(3) + (1 - math.cos(th)) / th ** 2 * O + (th - math.sin(th)) / th ** 3 * (O @ O)
    q = q / np.linalg.norm(q)
    return np.concatenate([q, V @ ups])


def se3_mul(A, B):
    q = quat_mul(A[:4], B[:4])
    q = q / np.linalg.norm(q)
    return np.concatenate([q, A[4:] + quat_to_R(A[:4]) @ B[4:]])


def se3_inv(A):
    q = np.array([-A[0], -A[1], -A[2], A[3]])
    return np.concatenate([q, -(quat_to_R(q) @ A[4:])])


def se3_act(T, p):
    return (quat_to_R(T[:4]) @ np.asarray(p).T).T + T[4:]


def se3_log_angle_dist(A, B):
    """(rotation angle [rad], translation distance) between two poses."""
    D = se3_mul(se3_inv(A), B)
    ang = 2.0 * math.atan2(np.linalg.norm(D[:3]), abs(D[3]))
    return ang, float(np.linalg.norm(D[4:]))


# ------------------------------------------------------------------------------------------------
# alignment streams
# ------------------------------------------------------------------------------------------------

N_SIN_MAIN = 24
N_SIN_FINE = 24


@dataclass
class AlignStream:
    seed: int
    W: int
    H: int
    cam: tuple                 # (fx, fy, cx, cy, W, H)
    plane_n: np.ndarray        # plane normal in the ref camera frame
    plane_d: float             # n . X = d
    e1: np.ndarray
    e2: np.ndarray
    tex: np.ndarray            # [K,4] = kx, ky, phase, amplitude (plane metric units)
    tex_norm: float
    T_true: np.ndarray         # cur_from_ref
    T_ref_w: np.ndarray        # ref frame's T_f_w
    T_cur_w_init: np.ndarray   # cur frame's initial T_f_w (= ref's, frame_handler_mono.cpp:266)
    T_init: np.ndarray         # cur_init * ref^-1 (identity up to rounding)
    # reference-frame features, as the reference's Frame/Feature/Point/LineSeg objects would hold them
    pt_px: np.ndarray = field(default=None)
    pt_f: np.ndarray = field(default=None)
    pt_pos_w: np.ndarray = field(default=None)
    seg_spx: np.ndarray = field(default=None)
    seg_epx: np.ndarray = field(default=None)
    seg_sf: np.ndarray = field(default=None)
    seg_ef: np.ndarray = field(default=None)
    seg_spos_w: np.ndarray = field(default=None)
    seg_epos_w: np.ndarray = field(default=None)
    # flattened for the C ABI (include/plsvo_hip.h, plsvo_align_in)
    ref_pos: np.ndarray = field(default=None)
    pt_xyz_ref: np.ndarray = field(default=None)
    seg_len: np.ndarray = field(default=None)
    seg_p_ref: np.ndarray = field(default=None)
    seg_q_ref: np.ndarray = field(default=None)


def _bearing(cam, px):
    fx, fy, cx, cy = cam[:4]
    r = np.stack([(px[:, 0] - cx) / fx, (px[:, 1] - cy) / fy, np.ones(len(px))], axis=1)
    return r / np.linalg.norm(r, axis=1, keepdims=True), r


def _on_plane(n, d, rays):
    """intersection of rays (through the origin) with n.X = d"""
    s = d / (rays @ n)
    return rays * s[:, None]


def make_align_stream(seed, W=640, H=480, n_pts=200, n_seg=80, max_level=3, motion_scale=0.5,
                      tex_lam=(16.0, 320.0), seg_len_range=None):
    rng = np.random.default_rng(seed)
    fx = fy = 0.65 * W
    cam = (fx, fy, W / 2.0, H / 2.0, W, H)
    d0 = rng.uniform(2.0, 6.0)
    tilt = math.radians(rng.uniform(0.0, 15.0))
    az = rng.uniform(0.0, 2 * math.pi)
    n = np.array([math.sin(tilt) * math.cos(az), math.sin(tilt) * math.sin(az), math.cos(tilt)])
    d = n[2] * d0
    e1 = np.array([1.0, 0.0, 0.0]) - n * n[0]
    e1 /= np.linalg.norm(e1)
    e2 = np.cross(n, e1)
    # texture: sinusoids with wavelength given in level-0 pixels at depth d0
    m_per_px = d0 / fx
    # 1/f-like amplitude spectrum (amplitude ~ wavelength), like natural images: smooth at the coarse
    # pyramid levels, detailed at the fine ones
    lam_px = np.concatenate([np.exp(rng.uniform(math.log(tex_lam[0]), math.log(tex_lam[1]), N_SIN_MAIN)),
                             np.exp(rng.uniform(math.log(4.0), math.log(tex_lam[0]), N_SIN_FINE))])
    lam = lam_px * m_per_px
    ang = rng.uniform(0.0, 2 * math.pi, lam.size)
    amp = rng.uniform(0.5, 1.0, lam.size) * lam_px / tex_lam[1]
    tex = np.stack([2 * math.pi / lam * np.cos(ang), 2 * math.pi / lam * np.sin(ang),
                    rng.uniform(0.0, 2 * math.pi, lam.size), amp], axis=1)
    tex_norm = math.sqrt(0.5 * float(np.sum(amp ** 2)))
    xi = np.concatenate([rng.uniform(-0.03, 0.03, 3) * d0, rng.uniform(-0.01, 0.01, 3)]) * motion_scale
    T_true = se3_exp(xi)
    T_ref_w = se3_exp(np.concatenate([rng.uniform(-1.0, 1.0, 3), rng.uniform(-0.3, 0.3, 3)]))
    T_w_ref = se3_inv(T_ref_w)
    T_cur_w_init = T_ref_w.copy()
    T_init = se3_mul(T_cur_w_init, T_w_ref)

    margin = 4.0 * (1 << max_level)
    pt_px = np.stack([rng.uniform(margin, W - margin, n_pts), rng.uniform(margin, H - margin, n_pts)], axis=1)
    lmin, lmax = (0.15 * W * H / (W + H), 0.35 * W) if seg_len_range is None else seg_len_range   # SURVEY.md 8(d) unless overridden
    spx = np.zeros((n_seg, 2))
    epx = np.zeros((n_seg, 2))
    k = 0
    while k < n_seg:
        s = np.array([rng.uniform(margin, W - margin), rng.uniform(margin, H - margin)])
        L = rng.uniform(lmin, lmax)
        a = rng.uniform(0.0, math.pi)
        e = s + L * np.array([math.cos(a), math.sin(a)])
        if margin <= e[0] < W - margin and margin <= e[1] < H - margin:
            spx[k], epx[k] = s, e
            k += 1

    st = AlignStream(seed=seed, W=W, H=H, cam=cam, plane_n=n, plane_d=d, e1=e1, e2=e2, tex=tex, tex_norm=tex_norm,
                     T_true=T_true, T_ref_w=T_ref_w, T_cur_w_init=T_cur_w_init, T_init=T_init)
    ref_pos = T_w_ref[4:].copy()  # Frame::pos() = T_f_w^-1 translation (frame.h:131)
    st.ref_pos = ref_pos

    def lift(px):
        f, rays = _bearing(cam, px)
        X = _on_plane(n, d, rays)
        pos_w = se3_act(T_w_ref, X)
        depth = np.linalg.norm(pos_w - ref_pos, axis=1)
        return f, pos_w, f * depth[:, None]

    st.pt_px = pt_px
    st.pt_f, st.pt_pos_w, st.pt_xyz_ref = lift(pt_px) if n_pts else (np.zeros((0, 3)),) * 3
    st.seg_spx, st.seg_epx = spx, epx
    if n_seg:
        st.seg_sf, st.seg_spos_w, st.seg_p_ref = lift(spx)
        st.seg_ef, st.seg_epos_w, st.seg_q_ref = lift(epx)
    else:
        z = np.zeros((0, 3))
        st.seg_sf = st.seg_spos_w = st.seg_p_ref = st.seg_ef = st.seg_epos_w = st.seg_q_ref = z
    st.seg_len = np.linalg.norm(epx - spx, axis=1)
    return st


def render_streams(streams, device="cpu", noise_sigma=2.0, chunk=64, which=(0, 1)):
    """Render the (ref, cur) level-0 images of a list of AlignStream -> uint8 tensor [B, 2, H, W] on `device`
    (`which` = (1,): only the cur images are rendered, [:, 0] is left unwritten)."""
    B = len(streams)
    W, H = streams[0].W, streams[0].H
    dev = torch.device(device)
    out = torch.empty((B, 2, H, W), dtype=torch.uint8, device=dev)
    fx, fy, cx, cy = streams[0].cam[:4]
    u = torch.arange(W, dtype=torch.float64, device=dev)
    v = torch.arange(H, dtype=torch.float64, device=dev)
    rays = torch.stack([((u - cx) / fx)[None, :].expand(H, W), ((v - cy) / fy)[:, None].expand(H, W),
                        torch.ones((H, W), dtype=torch.float64, device=dev)], dim=-1)  # [H,W,3]
    which_list = tuple(which)
    for c0 in range(0, B, chunk):
        sub = streams[c0:c0 + chunk]
        b = len(sub)
        t64 = lambda a: torch.as_tensor(np.stack(a), dtype=torch.float64, device=dev)
        n = t64([s.plane_n for s in sub])
        d = t64([np.array(s.plane_d) for s in sub])
        e1 = t64([s.e1 for s in sub])
        e2 = t64([s.e2 for s in sub])
        tex = t64([s.tex for s in sub])          # [b,K,4]
        tnorm = t64([np.array(s.tex_norm) for s in sub])
        gen = torch.Generator(device=dev)
        for which in which_list:
            if which == 0:
                o = torch.zeros((b, 3), dtype=torch.float64, device=dev)
                dirs = rays[None].expand(b, H, W, 3)
            else:
                R = t64([quat_to_R(s.T_true[:4]) for s in sub])      # cur_from_ref
                t = t64([s.T_true[4:] for s in sub])
                o = -torch.einsum("bji,bj->bi", R, t)                  # -R^T t
                dirs = torch.einsum("bji,hwj->bhwi", R, rays)          # R^T r
            nd = torch.einsum("bhwi,bi->bhw", dirs, n)
            s_ = (d - torch.einsum("bi,bi->b", n, o))[:, None, None] / nd
            X = o[:, None, None, :] + s_[..., None] * dirs
            a = torch.einsum("bhwi,bi->bhw", X, e1)
            bb = torch.einsum("bhwi,bi->bhw", X, e2)
            acc = torch.zeros_like(a)
            for k in range(tex.shape[1]):
                acc += tex[:, k, 3][:, None, None] * torch.sin(tex[:, k, 0][:, None, None] * a +
                                                               tex[:, k, 1][:, None, None] * bb +
                                                               tex[:, k, 2][:, None, None])
            img = 128.0 + 45.0 * acc / tnorm[:, None, None]
            img = img.clamp(16.0, 240.0)
            if noise_sigma > 0:
                gen.manual_seed(int(sub[0].seed) * 2 + which + 977)
                img = img + noise_sigma * torch.randn(img.shape, dtype=torch.float64, device=dev, generator=gen)
            out[c0:c0 + b, which] = img.round().clamp(0, 255).to(torch.uint8)
    return out


def stream_motion(st, k, motion_scale=0.5, model="independent"):
    """cur_from_ref of the k-th image of a MOVING sequence over stream `st`'s scene (bench.py's moving-inputs leg).
    "independent": drawn like make_align_stream draws T_true (the same range, the same scale) but independently for every k --
                   successive images of a stream share the scene, the features and the initial pose, nothing else: the harshest case
                   for anything learned from the previous launch;
    "smooth":      a camera on a smooth path -- one motion per stream (drawn as above), every image's motion within ~15 % of it
                   (per-image factor 1 + 0.15 N(0,1) on the whole twist, plus 5 % of its size in a random direction): what
                   successive inter-frame motions of a hand-held or vehicle camera look like."""
    d0 = st.plane_d / st.plane_n[2]
    draw = lambda rng: np.concatenate([rng.uniform(-0.03, 0.03, 3) * d0, rng.uniform(-0.01, 0.01, 3)]) * motion_scale
    if model == "independent":
        return se3_exp(draw(np.random.default_rng((int(st.seed) * 1000003 + 7919 * (int(k) + 1)) & 0x7fffffff)))
    base = draw(np.random.default_rng((int(st.seed) * 1000003 + 15485863) & 0x7fffffff))
    rng = np.random.default_rng((int(st.seed) * 1000003 + 32452843 * (int(k) + 1)) & 0x7fffffff)
    jitter = rng.normal(0.0, 1.0, 6)
    size = np.concatenate([np.full(3, np.linalg.norm(base[:3])), np.full(3, np.linalg.norm(base[3:]))])
    return se3_exp(base * (1.0 + 0.15 * rng.normal()) + 0.05 * size * jitter / np.sqrt(3.0))


def render_views(streams, poses, device="cpu", noise_sigma=2.0, chunk=64, noise_tag=0):
    """One level-0 image per stream, seen from `poses[i]` (cur_from_ref; None = the reference view) -> uint8 tensor [B, H, W] on
    `device`.  Same scene model and arithmetic as render_streams (which renders the pair ref / T_true)."""
    import copy
    out = None
    for c0 in range(0, len(streams), chunk):
        sub = []
        for s_, T in zip(streams[c0:c0 + chunk], poses[c0:c0 + chunk]):
            v = copy.copy(s_)
            v.T_true = np.array([0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0]) if T is None else np.asarray(T, dtype=np.float64)
            v.seed = int(s_.seed) + 104729 * (int(noise_tag) + 1)      # its own noise realisation
            sub.append(v)
        img = render_streams(sub, device=device, noise_sigma=noise_sigma, chunk=chunk, which=(1,))[:, 1]
        if out is None:
            out = torch.empty((len(streams),) + tuple(img.shape[1:]), dtype=torch.uint8, device=img.device)
        out[c0:c0 + len(sub)] = img
    return out


# ------------------------------------------------------------------------------------------------
# pose-optimisation frames
# ------------------------------------------------------------------------------------------------

@dataclass
class PoseOptFrame:
    seed: int
    fx: float
    T_true: np.ndarray
    T_init: np.ndarray
    pt_f: np.ndarray
    pt_pos: np.ndarray
    pt_level: np.ndarray
    pt_outlier: np.ndarray
    seg_line: np.ndarray
    seg_spos: np.ndarray
    seg_epos: np.ndarray
    seg_level: np.ndarray
    seg_outlier: np.ndarray


def make_poseopt_frame(seed, n_pts=500, n_seg=200, W=640, H=480, noise_px=1.0, outlier_frac=0.10,
                       outlier_px=20.0, pert_t=0.02, pert_r=0.01):
    rng = np.random.default_rng(seed + 500000)
    fx = fy = 0.65 * W
    cx, cy = W / 2.0, H / 2.0
    T_true = se3_exp(np.concatenate([rng.uniform(-1.0, 1.0, 3), rng.uniform(-0.3, 0.3, 3)]))   # T_f_w
    T_w_f = se3_inv(T_true)

    def sample_obs(k):
        px = np.stack([rng.uniform(8, W - 8, k), rng.uniform(8, H - 8, k)], axis=1)
        depth = rng.uniform(2.0, 10.0, k)
        rays = np.stack([(px[:, 0] - cx) / fx, (px[:, 1] - cy) / fy, np.ones(k)], axis=1)
        X_f = rays * depth[:, None]
        pos_w = se3_act(T_w_f, X_f)
        out = rng.uniform(size=k) < outlier_frac
        noise = rng.normal(0.0, noise_px, (k, 2))
        dirs = rng.uniform(0, 2 * math.pi, k)
        noise[out] += outlier_px * np.stack([np.cos(dirs[out]), np.sin(dirs[out])], axis=1)
        px_obs = px + noise
        r = np.stack([(px_obs[:, 0] - cx) / fx, (px_obs[:, 1] - cy) / fy, np.ones(k)], axis=1)
        f = r / np.linalg.norm(r, axis=1, keepdims=True)
        return f, pos_w, out

    pt_f, pt_pos, pt_out = sample_obs(n_pts)
    sf, spos, so = sample_obs(n_seg)
    ef, epos, eo = sample_obs(n_seg)
    line = np.cross(sf, ef)
    line = line / np.sqrt(line[:, 0:1] ** 2 + line[:, 1:2] ** 2) if n_seg else np.zeros((0, 3))
    pert = np.concatenate([rng.normal(0, 1, 3), rng.normal(0, 1, 3)])
    pert[:3] *= pert_t / max(np.linalg.norm(pert[:3]), 1e-12)
    pert[3:] *= pert_r / max(np.linalg.norm(pert[3:]), 1e-12)
    T_init = se3_mul(se3_exp(pert), T_true)
    return PoseOptFrame(seed=seed, fx=fx, T_true=T_true, T_init=T_init, pt_f=pt_f, pt_pos=pt_pos,
                        pt_level=rng.integers(0, 3, n_pts).astype(np.int32), pt_outlier=pt_out,
                        seg_line=line, seg_spos=spos, seg_epos=epos,
                        seg_level=rng.integers(0, 3, n_seg).astype(np.int32), seg_outlier=(so | eo))


# ------------------------------------------------------------------------------------------------
# structure optimisation: landmarks observed from several keyframes
# ------------------------------------------------------------------------------------------------

def make_structure_batch(seed, n_pts=20, n_seg=20, n_frames=6, W=640, H=480, noise_px=0.5, pert=0.05, obs_range=(2, 6)):
    """Landmarks in front of `n_frames` nearby cameras; every landmark is observed (noisy unit bearings) from a random
    subset of frames and starts `pert` metres off its true position.  Returns the flat arrays of plsvo_structopt_in
    plus the ground truth."""
    rng = np.random.default_rng(seed + 900000)
    fx = 0.65 * W
    frames = [se3_exp(np.concatenate([rng.uniform(-0.4, 0.4, 3), rng.uniform(-0.08, 0.08, 3)])) for _ in range(n_frames)]

    def bearing(T, X):
        p = se3_act(T, X)
        uv = p[:2] / p[2] + rng.normal(0.0, noise_px / fx, 2)
        r = np.array([uv[0], uv[1], 1.0])
        return r / np.linalg.norm(r)

    def landmark():
        return np.array([rng.uniform(-1.5, 1.5), rng.uniform(-1.0, 1.0), rng.uniform(3.0, 8.0)])

    def observe(points_of_landmark):
        k = int(rng.integers(obs_range[0], min(obs_range[1], n_frames) + 1))
        fr = rng.choice(n_frames, size=k, replace=False)
        return fr, [[bearing(frames[f], X) for f in fr] for X in points_of_landmark]

    pt_true = np.array([landmark() for _ in range(n_pts)]).reshape(-1, 3)
    pt_off, pt_fr, pt_f = [0], [], []
    for X in pt_true:
        fr, fs = observe([X])
        pt_fr += list(fr)
        pt_f += fs[0]
        pt_off.append(len(pt_fr))
    s_true = np.array([landmark() for _ in range(n_seg)]).reshape(-1, 3)
    e_true = s_true + rng.uniform(-0.6, 0.6, (n_seg, 3)) if n_seg else np.zeros((0, 3))
    sg_off, sg_fr, sg_sf, sg_ef = [0], [], [], []
    for Xs, Xe in zip(s_true, e_true):
        fr, fs = observe([Xs, Xe])
        sg_fr += list(fr)
        sg_sf += fs[0]
        sg_ef += fs[1]
        sg_off.append(len(sg_fr))
    z3 = np.zeros((0, 3))
    return dict(frame_T=np.array(frames), pt_true=pt_true, pt_pos=pt_true + rng.normal(0, pert, pt_true.shape),
                pt_obs_off=np.array(pt_off, np.int32), pt_obs_frame=np.array(pt_fr, np.int32), pt_obs_f=np.array(pt_f).reshape(-1, 3) if pt_f else z3,
                seg_s_true=s_true, seg_e_true=e_true, seg_spos=s_true + rng.normal(0, pert, s_true.shape),
                seg_epos=e_true + rng.normal(0, pert, e_true.shape), seg_obs_off=np.array(sg_off, np.int32),
                seg_obs_frame=np.array(sg_fr, np.int32), seg_obs_sf=np.array(sg_sf).reshape(-1, 3) if sg_sf else z3,
                seg_obs_ef=np.array(sg_ef).reshape(-1, 3) if sg_ef else z3)


# ------------------------------------------------------------------------------------------------
# direct feature matching: candidates of one (keyframe, current frame) pair
# ------------------------------------------------------------------------------------------------

def make_match_batch(seed, W=640, H=480, n_pts=120, n_seg=40, zoom=0.0, motion_scale=2.0, edgelet_frac=0.25, px_noise=1.5,
                     levels=(0, 1, 2, 3), level_p=(0.5, 0.3, 0.15, 0.05)):
    """A keyframe (frame 0) and a current frame (frame 1) looking at the textured plane of make_align_stream, plus the
    match candidates Reprojector would hand to Matcher::findMatchDirect: every landmark's closest-view observation is
    the keyframe's, its initial px_cur is the true projection displaced by up to `px_noise` pixels.  `zoom` moves the
    current camera that fraction of the scene depth towards the plane (det(A_cur_ref) > 3 selects search level 1 from
    zoom ~ 0.43).  Returns (stream, dict of plsvo_match_in arrays); render the two images with render_streams([stream]).
    Candidate order: points, segment start points, segment end points."""
    st = make_align_stream(seed, W, H, n_pts, n_seg, 3, motion_scale)
    rng = np.random.default_rng(seed + 700000)
    if zoom != 0.0:
        d0 = st.plane_d / st.plane_n[2]
        xi = np.concatenate([rng.uniform(-0.02, 0.02, 2) * d0, [-zoom * d0], rng.uniform(-0.01, 0.01, 3)])
        # cur_from_ref: a camera moved forward by z sees the scene shifted by -z ... sign chosen so that depth shrinks
        st.T_true = se3_exp(xi)
    T_cur_w = se3_mul(st.T_true, st.T_ref_w)
    fx, fy, cx, cy = st.cam[:4]
    ref_px = np.concatenate([st.pt_px, st.seg_spx, st.seg_epx])
    ref_f = np.concatenate([st.pt_f, st.seg_sf, st.seg_ef])
    pos = np.concatenate([st.pt_pos_w, st.seg_spos_w, st.seg_epos_w])
    n = ref_px.shape[0]
    p_cur = se3_act(T_cur_w, pos)
    px_true = np.stack([fx * p_cur[:, 0] / p_cur[:, 2] + cx, fy * p_cur[:, 1] / p_cur[:, 2] + cy], axis=1)
    px_cur = px_true + rng.uniform(-px_noise, px_noise, (n, 2))
    ref_level = rng.choice(np.asarray(levels), size=n, p=np.asarray(level_p)).astype(np.int32)
    ref_type = np.zeros(n, np.uint8)
    ref_type[:n_pts] = (rng.uniform(size=n_pts) < edgelet_frac).astype(np.uint8)
    ga = rng.uniform(0.0, 2 * math.pi, n)
    ref_grad = np.stack([np.cos(ga), np.sin(ga)], axis=1)
    d = dict(cam=st.cam, frame_T=np.stack([st.T_ref_w, T_cur_w]), frame_slot=np.array([0, 1], np.int32),
             cur_frame=np.ones(n, np.int32), ref_frame=np.zeros(n, np.int32), ref_px=ref_px, ref_f=ref_f, ref_level=ref_level,
             ref_type=ref_type, ref_grad=ref_grad, pos=pos, px_cur=px_cur, px_true=px_true, n_pts=n_pts, n_seg=n_seg)
    return st, d


# ------------------------------------------------------------------------------------------------
# depth-filter seeds: uninformed depth priors on the keyframe-0 features of a sequence
# ------------------------------------------------------------------------------------------------

def make_seeds(seq, cur_frame=1, edgelet_frac=0.2, depth_spread=0.8, seed=0):
    """Point and line seeds as DepthFilter::initializeSeeds creates them (src/depth_filter.cpp:49-71): mu = 1/depth_mean,
    z_range = 1/depth_min, sigma2 = z_range^2/36, a = b = 10 -- for every keyframe-0 feature of a sequence made by
    sequence.make_sequence (a dict).  Returns (pt, seg, truth): the field dicts of abi.SeedsJob and the true depths."""
    rng = np.random.default_rng(seed + 4100)
    cam, T = seq["cam"], seq["poses_true"]
    ref_pos = se3_inv(T[0])[4:]
    depth = np.linalg.norm(seq["pt_pos"] - ref_pos, axis=1)
    sdepth = np.linalg.norm(seq["seg_spos"] - ref_pos, axis=1)
    edepth = np.linalg.norm(seq["seg_epos"] - ref_pos, axis=1)
    all_d = np.concatenate([depth, sdepth, edepth])
    dmean, dmin = float(np.mean(all_d)), float(np.min(all_d)) * depth_spread
    n, ns = len(depth), len(sdepth)
    ga = rng.uniform(0, 2 * math.pi, n)
    pt = dict(ref_frame=np.zeros(n, np.int32), cur_frame=np.full(n, cur_frame, np.int32), px=seq["pt_px0"], f=seq["pt_f0"], level=np.zeros(n, np.int32),
              type=(rng.uniform(size=n) < edgelet_frac).astype(np.uint8), grad=np.stack([np.cos(ga), np.sin(ga)], axis=1),
              a=np.full(n, 10.0), b=np.full(n, 10.0), mu=np.full(n, 1.0 / dmean), z_range=np.full(n, 1.0 / dmin), sigma2=np.full(n, (1.0 / dmin) ** 2 / 36.0))
    mid = 0.5 * (seq["seg_spx0"] + seq["seg_epx0"])            # LineFeat's Feature::px is the segment centre
    fx, fy, cx, cy = cam[:4]
    fm = np.stack([(mid[:, 0] - cx) / fx, (mid[:, 1] - cy) / fy, np.ones(ns)], axis=1)
    fm /= np.linalg.norm(fm, axis=1, keepdims=True)
    z0 = np.full(ns, 1.0 / dmean)
    zr = np.full(ns, 1.0 / dmin)
    seg = dict(ref_frame=np.zeros(ns, np.int32), cur_frame=np.full(ns, cur_frame, np.int32), px=mid, f=fm, sf=seq["seg_sf0"], ef=seq["seg_ef0"],
               level=np.zeros(ns, np.int32), a=np.full(ns, 10.0), b=np.full(ns, 10.0), mu_s=z0.copy(), mu_e=z0.copy(), z_range_s=zr.copy(),
               z_range_e=zr.copy(), sigma2_s=zr ** 2 / 36.0, sigma2_e=zr ** 2 / 36.0)
    return pt, seg, dict(pt_depth=depth, seg_sdepth=sdepth, seg_edepth=edepth)


def apply_seed_update(pt, seg, res):
    """copy the posterior of a plsvo_update_seeds result back into the field dicts (what the host does per frame)"""
    for k in ("a", "b", "mu", "sigma2"):
        pt[k] = res["pt_" + k].copy()
    for k in ("a", "b", "mu_s", "mu_e", "sigma2_s", "sigma2_e"):
        seg[k] = res["seg_" + k].copy()
