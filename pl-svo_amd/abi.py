"""ctypes mirror of include/plsvo_hip.h (struct layouts + helpers to fill them from numpy arrays).

Used by the product binding (capi.py, loads libplsvo_hip.so); the test-only CPU checker shares these
struct layouts and imports this module from its own directory.  Nothing here computes anything.
"""
import ctypes as C

import numpy as np

MAX_LEVELS = 8
PATCH_AREA = 16

c_double_p = C.POINTER(C.c_double)
c_u8_p = C.POINTER(C.c_uint8)
c_i32_p = C.POINTER(C.c_int32)

# error codes (include/plsvo_hip.h)
OK, E_INVALID, E_NODEVICE, E_HIP, E_CAPACITY, E_STATE, E_RCCL = 0, -1, -2, -3, -4, -5, -6

# kernel families for plsvo_hip_kernel_time
K_ALIGN_INIT, K_ALIGN_LEVEL, K_POSEOPT, K_HALFSAMPLE, K_STRUCTOPT, K_MATCH, K_SEEDS, K_COUNT = 0, 1, 2, 3, 4, 5, 6, 7
SEED_NOT_VISIBLE, SEED_NO_MATCH, SEED_UPDATED, SEED_CONVERGED, SEED_NAN = 0, 1, 2, 3, 4
FTR_CORNER, FTR_EDGELET = 0, 1


class Pinhole(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32)]


class AlignIn(C.Structure):
    _fields_ = [("ref_slot", C.c_int32), ("cur_slot", C.c_int32), ("cam", Pinhole),
                ("max_level", C.c_int32), ("min_level", C.c_int32), ("n_iter", C.c_int32),
                ("reserved0", C.c_int32), ("eps", C.c_double), ("T_cur_from_ref", C.c_double * 7),
                ("n_pts", C.c_int32), ("n_seg", C.c_int32),
                ("pt_px", c_double_p), ("pt_xyz_ref", c_double_p),
                ("seg_spx", c_double_p), ("seg_epx", c_double_p), ("seg_len", c_double_p),
                ("seg_p_ref", c_double_p), ("seg_q_ref", c_double_p), ("seg_alive_in", c_u8_p)]


class AlignOut(C.Structure):
    _fields_ = [("T_cur_from_ref", C.c_double * 7), ("n_meas", C.c_uint64), ("n_tracked", C.c_uint64),
                ("H", C.c_double * 36), ("chi2", C.c_double), ("seg_alive_out", c_u8_p),
                ("iters_per_level", C.c_int32 * MAX_LEVELS), ("status", C.c_int32), ("reserved0", C.c_int32)]


class AlignIterLog(C.Structure):
    _fields_ = [("level", C.c_int32), ("iter", C.c_int32), ("accepted", C.c_int32), ("stop", C.c_int32),
                ("n_meas", C.c_uint64), ("new_chi2", C.c_double), ("H", C.c_double * 36),
                ("Jres", C.c_double * 6), ("x", C.c_double * 6), ("T_after", C.c_double * 7)]


class PoseOptIn(C.Structure):
    _fields_ = [("T_f_w", C.c_double * 7), ("fx", C.c_double), ("reproj_thresh", C.c_double),
                ("n_iter", C.c_int32), ("n_iter_ref", C.c_int32), ("n_pts", C.c_int32), ("n_seg", C.c_int32),
                ("pt_f", c_double_p), ("pt_pos", c_double_p), ("pt_level", c_i32_p),
                ("seg_line", c_double_p), ("seg_spos", c_double_p), ("seg_epos", c_double_p),
                ("seg_level", c_i32_p)]


class PoseOptOut(C.Structure):
    _fields_ = [("T_f_w", C.c_double * 7), ("cov", C.c_double * 36), ("estimated_scale", C.c_double),
                ("error_init", C.c_double), ("error_final", C.c_double), ("num_obs_pt", C.c_uint64),
                ("num_obs_ls", C.c_uint64), ("pt_keep", c_u8_p), ("seg_keep", c_u8_p), ("iters", C.c_int32),
                ("iters_ref", C.c_int32), ("status", C.c_int32), ("reserved0", C.c_int32)]


class PoseOptIterLog(C.Structure):
    _fields_ = [("phase", C.c_int32), ("iter", C.c_int32), ("accepted", C.c_int32), ("reserved0", C.c_int32),
                ("new_chi2", C.c_double), ("A", C.c_double * 36), ("b", C.c_double * 6), ("dT", C.c_double * 6),
                ("T_after", C.c_double * 7)]


class StructOptIn(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_iter_pts", C.c_int32), ("n_iter_segs", C.c_int32), ("n_pts", C.c_int32),
                ("n_seg", C.c_int32), ("reserved0", C.c_int32), ("frame_T", c_double_p), ("pt_pos", c_double_p),
                ("pt_obs_off", c_i32_p), ("pt_obs_frame", c_i32_p), ("pt_obs_f", c_double_p), ("seg_spos", c_double_p),
                ("seg_epos", c_double_p), ("seg_obs_off", c_i32_p), ("seg_obs_frame", c_i32_p), ("seg_obs_sf", c_double_p),
                ("seg_obs_ef", c_double_p)]


class StructOptOut(C.Structure):
    _fields_ = [("pt_pos", c_double_p), ("seg_spos", c_double_p), ("seg_epos", c_double_p), ("pt_iters", c_i32_p),
                ("seg_iters", c_i32_p)]


class MatchIn(C.Structure):
    _fields_ = [("cam", Pinhole), ("n_pyr_levels", C.c_int32), ("align_max_iter", C.c_int32), ("n_frames", C.c_int32),
                ("n", C.c_int32), ("frame_T", c_double_p), ("frame_slot", c_i32_p), ("cur_frame", c_i32_p),
                ("ref_frame", c_i32_p), ("ref_px", c_double_p), ("ref_f", c_double_p), ("ref_level", c_i32_p),
                ("ref_type", c_u8_p), ("ref_grad", c_double_p), ("pos", c_double_p), ("px_cur", c_double_p)]


class MatchOut(C.Structure):
    _fields_ = [("px_cur", c_double_p), ("found", c_u8_p), ("search_level", c_i32_p), ("n_iter", c_i32_p)]


class ReprojectIn(C.Structure):
    _fields_ = [("cam", Pinhole), ("n_frames", C.c_int32), ("n", C.c_int32), ("cell_size", C.c_int32), ("grid_n_cols", C.c_int32),
                ("boundary", C.c_int32), ("reserved0", C.c_int32), ("frame_T", c_double_p), ("frame", c_i32_p), ("pos", c_double_p)]


class ReprojectOut(C.Structure):
    _fields_ = [("px", c_double_p), ("cell", c_i32_p)]


class ChainIn(C.Structure):
    _fields_ = [("align", AlignIn), ("T_prev_w", C.c_double * 7), ("T_kf_w", C.c_double * 7), ("kf_slot", C.c_int32),
                ("n_cand_pt", C.c_int32), ("n_cand_seg", C.c_int32), ("reserved0", C.c_int32),
                ("pos", c_double_p), ("ref_px", c_double_p), ("ref_f", c_double_p), ("ref_level", c_i32_p), ("ref_type", c_u8_p),
                ("ref_grad", c_double_p), ("active", c_u8_p)]


class ChainParams(C.Structure):
    _fields_ = [("cam", Pinhole), ("n_pyr_levels", C.c_int32), ("align_max_iter", C.c_int32), ("cell_size", C.c_int32),
                ("cell_rule", C.c_int32), ("max_fts", C.c_int32), ("poseopt_n_iter", C.c_int32), ("cell_order", c_i32_p),
                ("reproj_thresh", C.c_double), ("seg_cell_size", C.c_int32), ("max_fts_segs", C.c_int32), ("seg_cell_order", c_i32_p)]


class ChainOut(C.Structure):
    _fields_ = [("align", AlignOut), ("pose", PoseOptOut), ("n_sel_pt", C.c_int32), ("n_sel_seg", C.c_int32),
                ("found", c_u8_p), ("px", c_double_p), ("search_level", c_i32_p), ("sel_pt", c_i32_p), ("sel_seg", c_i32_p)]


class PoseRecord(C.Structure):
    """plsvo_pose_record: the 96-byte per-stream record a rank publishes (plsvo_pack_pose_records / plsvo_gather_poses)"""
    _fields_ = [("T_f_w", C.c_double * 7), ("n_tracked", C.c_uint64), ("num_obs_pt", C.c_uint64), ("num_obs_ls", C.c_uint64),
                ("error_final", C.c_double), ("status", C.c_int32), ("stream", C.c_int32)]


POSE_RECORD_BYTES = 96
POSE_RECORD_DTYPE = np.dtype([("T_f_w", np.float64, 7), ("n_tracked", np.uint64), ("num_obs_pt", np.uint64), ("num_obs_ls", np.uint64),
                              ("error_final", np.float64), ("status", np.int32), ("stream", np.int32)])
REC_ALIGN, REC_ALIGN_STOP, REC_ALIGN_ERROR, REC_POSEOPT, REC_POSEOPT_EMPTY = 0x01, 0x02, 0x04, 0x08, 0x10

c_float_p = C.POINTER(C.c_float)


class SeedsIn(C.Structure):
    _fields_ = [("cam", Pinhole), ("n_pyr_levels", C.c_int32), ("align_max_iter", C.c_int32), ("max_epi_search_steps", C.c_int32),
                ("edgelet_filtering", C.c_int32), ("edgelet_max_angle", C.c_double), ("px_noise", C.c_double),
                ("convergence_sigma2_thresh", C.c_double), ("n_frames", C.c_int32), ("n_pt", C.c_int32), ("n_seg", C.c_int32),
                ("reserved0", C.c_int32), ("frame_T", c_double_p), ("frame_slot", c_i32_p),
                ("pt_ref_frame", c_i32_p), ("pt_cur_frame", c_i32_p), ("pt_px", c_double_p), ("pt_f", c_double_p), ("pt_level", c_i32_p),
                ("pt_type", c_u8_p), ("pt_grad", c_double_p), ("pt_a", c_float_p), ("pt_b", c_float_p), ("pt_mu", c_float_p),
                ("pt_z_range", c_float_p), ("pt_sigma2", c_float_p),
                ("seg_ref_frame", c_i32_p), ("seg_cur_frame", c_i32_p), ("seg_px", c_double_p), ("seg_f", c_double_p), ("seg_sf", c_double_p),
                ("seg_ef", c_double_p), ("seg_level", c_i32_p), ("seg_a", c_float_p), ("seg_b", c_float_p), ("seg_mu_s", c_float_p),
                ("seg_mu_e", c_float_p), ("seg_z_range_s", c_float_p), ("seg_z_range_e", c_float_p), ("seg_sigma2_s", c_float_p),
                ("seg_sigma2_e", c_float_p)]


class SeedsOut(C.Structure):
    _fields_ = [("pt_status", c_i32_p), ("pt_a", c_float_p), ("pt_b", c_float_p), ("pt_mu", c_float_p), ("pt_sigma2", c_float_p),
                ("pt_xyz_world", c_double_p), ("pt_px_cur", c_double_p), ("pt_depth", c_double_p),
                ("seg_status", c_i32_p), ("seg_a", c_float_p), ("seg_b", c_float_p), ("seg_mu_s", c_float_p), ("seg_mu_e", c_float_p),
                ("seg_sigma2_s", c_float_p), ("seg_sigma2_e", c_float_p), ("seg_xyz_world_s", c_double_p), ("seg_xyz_world_e", c_double_p),
                ("seg_depth_s", c_double_p), ("seg_depth_e", c_double_p)]


def _f64(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} doubles, got {a.size}")
    return a


def _ptr(a, typ):
    return a.ctypes.data_as(typ) if a.size else C.cast(None, typ)


class AlignJob:
    """Owns the numpy buffers behind one plsvo_align_in (keeps them alive) and exposes .c (the struct)."""

    def __init__(self, cam, max_level, min_level, n_iter, eps, T_cur_from_ref, pt_px, pt_xyz_ref,
                 seg_spx, seg_epx, seg_len, seg_p_ref, seg_q_ref, seg_alive_in=None, ref_slot=0, cur_slot=1):
        self.pt_px = _f64(pt_px).reshape(-1, 2)
        n_pts = self.pt_px.shape[0]
        self.pt_xyz_ref = _f64(pt_xyz_ref, 3 * n_pts).reshape(-1, 3)
        self.seg_spx = _f64(seg_spx).reshape(-1, 2)
        n_seg = self.seg_spx.shape[0]
        self.seg_epx = _f64(seg_epx, 2 * n_seg).reshape(-1, 2)
        self.seg_len = _f64(seg_len, n_seg).reshape(-1)
        self.seg_p_ref = _f64(seg_p_ref, 3 * n_seg).reshape(-1, 3)
        self.seg_q_ref = _f64(seg_q_ref, 3 * n_seg).reshape(-1, 3)
        self.seg_alive_in = None if seg_alive_in is None else np.ascontiguousarray(seg_alive_in, dtype=np.uint8)
        c = AlignIn()
        c.ref_slot, c.cur_slot = ref_slot, cur_slot
        c.cam = cam if isinstance(cam, Pinhole) else Pinhole(*cam)
        c.max_level, c.min_level, c.n_iter, c.eps = max_level, min_level, n_iter, eps
        c.T_cur_from_ref[:] = list(_f64(T_cur_from_ref, 7))
        c.n_pts, c.n_seg = n_pts, n_seg
        c.pt_px = _ptr(self.pt_px, c_double_p)
        c.pt_xyz_ref = _ptr(self.pt_xyz_ref, c_double_p)
        c.seg_spx = _ptr(self.seg_spx, c_double_p)
        c.seg_epx = _ptr(self.seg_epx, c_double_p)
        c.seg_len = _ptr(self.seg_len, c_double_p)
        c.seg_p_ref = _ptr(self.seg_p_ref, c_double_p)
        c.seg_q_ref = _ptr(self.seg_q_ref, c_double_p)
        c.seg_alive_in = (C.cast(None, c_u8_p) if self.seg_alive_in is None
                          else self.seg_alive_in.ctypes.data_as(c_u8_p))
        self.c = c
        self.n_pts, self.n_seg = n_pts, n_seg


class PoseOptJob:
    def __init__(self, T_f_w, fx, reproj_thresh, n_iter, pt_f, pt_pos, pt_level, seg_line, seg_spos, seg_epos,
                 seg_level, n_iter_ref=-1):
        self.pt_f = _f64(pt_f).reshape(-1, 3)
        n_pts = self.pt_f.shape[0]
        self.pt_pos = _f64(pt_pos, 3 * n_pts).reshape(-1, 3)
        self.pt_level = np.ascontiguousarray(pt_level, dtype=np.int32).reshape(-1)
        self.seg_line = _f64(seg_line).reshape(-1, 3)
        n_seg = self.seg_line.shape[0]
        self.seg_spos = _f64(seg_spos, 3 * n_seg).reshape(-1, 3)
        self.seg_epos = _f64(seg_epos, 3 * n_seg).reshape(-1, 3)
        self.seg_level = np.ascontiguousarray(seg_level, dtype=np.int32).reshape(-1)
        c = PoseOptIn()
        c.T_f_w[:] = list(_f64(T_f_w, 7))
        c.fx, c.reproj_thresh, c.n_iter, c.n_iter_ref = fx, reproj_thresh, n_iter, n_iter_ref
        c.n_pts, c.n_seg = n_pts, n_seg
        c.pt_f = _ptr(self.pt_f, c_double_p)
        c.pt_pos = _ptr(self.pt_pos, c_double_p)
        c.pt_level = _ptr(self.pt_level, c_i32_p)
        c.seg_line = _ptr(self.seg_line, c_double_p)
        c.seg_spos = _ptr(self.seg_spos, c_double_p)
        c.seg_epos = _ptr(self.seg_epos, c_double_p)
        c.seg_level = _ptr(self.seg_level, c_i32_p)
        self.c = c
        self.n_pts, self.n_seg = n_pts, n_seg


class AlignResult:
    """Plain-python view of a plsvo_align_out."""

    def __init__(self, out, seg_alive):
        self.T = np.array(out.T_cur_from_ref[:], dtype=np.float64)
        self.n_meas = int(out.n_meas)
        self.n_tracked = int(out.n_tracked)
        self.H = np.array(out.H[:], dtype=np.float64).reshape(6, 6)
        self.chi2 = float(out.chi2)
        self.iters_per_level = list(out.iters_per_level[:])
        self.status = int(out.status)
        self.seg_alive = seg_alive


class PoseOptResult:
    def __init__(self, out, pt_keep, seg_keep):
        self.T = np.array(out.T_f_w[:], dtype=np.float64)
        self.cov = np.array(out.cov[:], dtype=np.float64).reshape(6, 6)
        self.estimated_scale = float(out.estimated_scale)
        self.error_init = float(out.error_init)
        self.error_final = float(out.error_final)
        self.num_obs_pt = int(out.num_obs_pt)
        self.num_obs_ls = int(out.num_obs_ls)
        self.iters = int(out.iters)
        self.iters_ref = int(out.iters_ref)
        self.status = int(out.status)
        self.pt_keep = pt_keep
        self.seg_keep = seg_keep


def align_log_to_dicts(arr, n):
    return [dict(level=r.level, iter=r.iter, accepted=r.accepted, stop=r.stop, n_meas=int(r.n_meas),
                 new_chi2=float(r.new_chi2), H=np.array(r.H[:]).reshape(6, 6), Jres=np.array(r.Jres[:]),
                 x=np.array(r.x[:]), T_after=np.array(r.T_after[:])) for r in arr[:n]]


def poseopt_log_to_dicts(arr, n):
    return [dict(phase=r.phase, iter=r.iter, accepted=r.accepted, new_chi2=float(r.new_chi2),
                 A=np.array(r.A[:]).reshape(6, 6), b=np.array(r.b[:]), dT=np.array(r.dT[:]),
                 T_after=np.array(r.T_after[:])) for r in arr[:n]]


class StructOptJob:
    """A batch of landmarks for plsvo_structure_optimize; owns the numpy buffers and the output arrays."""

    def __init__(self, frame_T, pt_pos, pt_obs_off, pt_obs_frame, pt_obs_f, seg_spos, seg_epos, seg_obs_off, seg_obs_frame,
                 seg_obs_sf, seg_obs_ef, n_iter_pts=5, n_iter_segs=5):
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32).reshape(-1)
        self.frame_T = _f64(frame_T).reshape(-1, 7)
        self.pt_pos = _f64(pt_pos).reshape(-1, 3)
        self.pt_obs_off, self.pt_obs_frame, self.pt_obs_f = i32(pt_obs_off), i32(pt_obs_frame), _f64(pt_obs_f).reshape(-1, 3)
        self.seg_spos, self.seg_epos = _f64(seg_spos).reshape(-1, 3), _f64(seg_epos).reshape(-1, 3)
        self.seg_obs_off, self.seg_obs_frame = i32(seg_obs_off), i32(seg_obs_frame)
        self.seg_obs_sf, self.seg_obs_ef = _f64(seg_obs_sf).reshape(-1, 3), _f64(seg_obs_ef).reshape(-1, 3)
        self.n_pts, self.n_seg = self.pt_pos.shape[0], self.seg_spos.shape[0]
        c = StructOptIn()
        c.n_frames, c.n_iter_pts, c.n_iter_segs, c.n_pts, c.n_seg = self.frame_T.shape[0], n_iter_pts, n_iter_segs, self.n_pts, self.n_seg
        c.frame_T = _ptr(self.frame_T, c_double_p)
        c.pt_pos, c.pt_obs_off = _ptr(self.pt_pos, c_double_p), _ptr(self.pt_obs_off, c_i32_p)
        c.pt_obs_frame, c.pt_obs_f = _ptr(self.pt_obs_frame, c_i32_p), _ptr(self.pt_obs_f, c_double_p)
        c.seg_spos, c.seg_epos = _ptr(self.seg_spos, c_double_p), _ptr(self.seg_epos, c_double_p)
        c.seg_obs_off, c.seg_obs_frame = _ptr(self.seg_obs_off, c_i32_p), _ptr(self.seg_obs_frame, c_i32_p)
        c.seg_obs_sf, c.seg_obs_ef = _ptr(self.seg_obs_sf, c_double_p), _ptr(self.seg_obs_ef, c_double_p)
        self.c = c

    def make_out(self):
        o = StructOptOut()
        bufs = dict(pt_pos=np.zeros((max(self.n_pts, 1), 3)), seg_spos=np.zeros((max(self.n_seg, 1), 3)),
                    seg_epos=np.zeros((max(self.n_seg, 1), 3)), pt_iters=np.zeros(max(self.n_pts, 1), np.int32),
                    seg_iters=np.zeros(max(self.n_seg, 1), np.int32))
        o.pt_pos, o.seg_spos, o.seg_epos = (bufs[k].ctypes.data_as(c_double_p) for k in ("pt_pos", "seg_spos", "seg_epos"))
        o.pt_iters, o.seg_iters = bufs["pt_iters"].ctypes.data_as(c_i32_p), bufs["seg_iters"].ctypes.data_as(c_i32_p)
        return o, bufs

    def trim(self, bufs):
        return dict(pt_pos=bufs["pt_pos"][:self.n_pts].copy(), seg_spos=bufs["seg_spos"][:self.n_seg].copy(),
                    seg_epos=bufs["seg_epos"][:self.n_seg].copy(), pt_iters=bufs["pt_iters"][:self.n_pts].copy(),
                    seg_iters=bufs["seg_iters"][:self.n_seg].copy())


class MatchJob:
    """A batch of candidates for plsvo_match_direct; owns the numpy buffers and the output arrays."""

    def __init__(self, cam, frame_T, frame_slot, cur_frame, ref_frame, ref_px, ref_f, ref_level, ref_type, ref_grad, pos,
                 px_cur, n_pyr_levels=3, align_max_iter=10):
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32).reshape(-1)
        self.frame_T = _f64(frame_T).reshape(-1, 7)
        self.frame_slot = i32(frame_slot)
        self.ref_px = _f64(ref_px).reshape(-1, 2)
        n = self.n = self.ref_px.shape[0]
        self.cur_frame, self.ref_frame, self.ref_level = i32(cur_frame), i32(ref_frame), i32(ref_level)
        self.ref_f = _f64(ref_f, 3 * n).reshape(-1, 3)
        self.ref_type = np.ascontiguousarray(ref_type, dtype=np.uint8).reshape(-1)
        self.ref_grad = _f64(ref_grad, 2 * n).reshape(-1, 2)
        self.pos = _f64(pos, 3 * n).reshape(-1, 3)
        self.px_cur = _f64(px_cur, 2 * n).reshape(-1, 2)
        c = MatchIn()
        c.cam = cam if isinstance(cam, Pinhole) else Pinhole(*cam)
        c.n_pyr_levels, c.align_max_iter, c.n_frames, c.n = n_pyr_levels, align_max_iter, self.frame_T.shape[0], n
        c.frame_T, c.frame_slot = _ptr(self.frame_T, c_double_p), _ptr(self.frame_slot, c_i32_p)
        c.cur_frame, c.ref_frame = _ptr(self.cur_frame, c_i32_p), _ptr(self.ref_frame, c_i32_p)
        c.ref_px, c.ref_f = _ptr(self.ref_px, c_double_p), _ptr(self.ref_f, c_double_p)
        c.ref_level, c.ref_type = _ptr(self.ref_level, c_i32_p), _ptr(self.ref_type, c_u8_p)
        c.ref_grad, c.pos, c.px_cur = _ptr(self.ref_grad, c_double_p), _ptr(self.pos, c_double_p), _ptr(self.px_cur, c_double_p)
        self.c = c

    def make_out(self):
        o = MatchOut()
        m = max(self.n, 1)
        bufs = dict(px_cur=np.zeros((m, 2)), found=np.zeros(m, np.uint8), search_level=np.zeros(m, np.int32),
                    n_iter=np.zeros(m, np.int32))
        o.px_cur, o.found = bufs["px_cur"].ctypes.data_as(c_double_p), bufs["found"].ctypes.data_as(c_u8_p)
        o.search_level, o.n_iter = bufs["search_level"].ctypes.data_as(c_i32_p), bufs["n_iter"].ctypes.data_as(c_i32_p)
        return o, bufs

    def trim(self, bufs):
        return {k: v[:self.n].copy() for k, v in bufs.items()}


class ReprojectJob:
    """Landmark positions to project into frames (plsvo_reproject); owns the buffers."""

    def __init__(self, cam, frame_T, frame, pos, cell_size=30, boundary=8):
        self.frame_T = _f64(frame_T).reshape(-1, 7)
        self.pos = _f64(pos).reshape(-1, 3)
        self.n = self.pos.shape[0]
        self.frame = np.ascontiguousarray(frame, dtype=np.int32).reshape(-1)
        c = ReprojectIn()
        c.cam = cam if isinstance(cam, Pinhole) else Pinhole(*cam)
        c.n_frames, c.n, c.cell_size, c.boundary = self.frame_T.shape[0], self.n, cell_size, boundary
        c.grid_n_cols = -(-int(c.cam.width) // cell_size)          # ceil(width / cell_size), reprojector.cpp:59
        c.frame_T, c.frame, c.pos = _ptr(self.frame_T, c_double_p), _ptr(self.frame, c_i32_p), _ptr(self.pos, c_double_p)
        self.c = c

    def make_out(self):
        o = ReprojectOut()
        m = max(self.n, 1)
        bufs = dict(px=np.zeros((m, 2)), cell=np.zeros(m, np.int32))
        o.px, o.cell = bufs["px"].ctypes.data_as(c_double_p), bufs["cell"].ctypes.data_as(c_i32_p)
        return o, bufs

    def trim(self, bufs):
        return {k: v[:self.n].copy() for k, v in bufs.items()}


class SeedsJob:
    """Depth-filter seeds to update against one frame each (plsvo_update_seeds); owns the buffers.
    `pt` / `seg` are dicts of arrays named like the plsvo_seeds_in fields without the prefix."""
    PT_F32 = ("a", "b", "mu", "z_range", "sigma2")
    SEG_F32 = ("a", "b", "mu_s", "mu_e", "z_range_s", "z_range_e", "sigma2_s", "sigma2_e")

    def __init__(self, cam, frame_T, frame_slot, pt, seg, n_pyr_levels=3, align_max_iter=10, max_epi_search_steps=1000,
                 edgelet_filtering=True, edgelet_max_angle=0.7, px_noise=1.0, convergence_sigma2_thresh=200.0):
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32).reshape(-1)
        f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
        self.frame_T = _f64(frame_T).reshape(-1, 7)
        self.frame_slot = i32(frame_slot)
        c = SeedsIn()
        c.cam = cam if isinstance(cam, Pinhole) else Pinhole(*cam)
        c.n_pyr_levels, c.align_max_iter, c.max_epi_search_steps = n_pyr_levels, align_max_iter, max_epi_search_steps
        c.edgelet_filtering, c.edgelet_max_angle, c.px_noise = int(edgelet_filtering), edgelet_max_angle, px_noise
        c.convergence_sigma2_thresh = convergence_sigma2_thresh
        c.n_frames = self.frame_T.shape[0]
        c.frame_T, c.frame_slot = _ptr(self.frame_T, c_double_p), _ptr(self.frame_slot, c_i32_p)
        self.pt, self.seg = {}, {}
        npt = self.n_pt = len(pt["px"]) if pt else 0
        nsg = self.n_seg = len(seg["px"]) if seg else 0
        c.n_pt, c.n_seg = npt, nsg
        if npt:
            P_ = self.pt
            P_["ref_frame"], P_["cur_frame"], P_["level"] = i32(pt["ref_frame"]), i32(pt["cur_frame"]), i32(pt["level"])
            P_["px"], P_["f"] = _f64(pt["px"], 2 * npt).reshape(-1, 2), _f64(pt["f"], 3 * npt).reshape(-1, 3)
            P_["type"] = np.ascontiguousarray(pt.get("type", np.zeros(npt)), dtype=np.uint8).reshape(-1)
            P_["grad"] = _f64(pt.get("grad", np.zeros((npt, 2))), 2 * npt).reshape(-1, 2)
            for k in self.PT_F32:
                P_[k] = f32(pt[k])
            c.pt_ref_frame, c.pt_cur_frame, c.pt_level = (_ptr(P_[k], c_i32_p) for k in ("ref_frame", "cur_frame", "level"))
            c.pt_px, c.pt_f, c.pt_grad, c.pt_type = _ptr(P_["px"], c_double_p), _ptr(P_["f"], c_double_p), _ptr(P_["grad"], c_double_p), _ptr(P_["type"], c_u8_p)
            c.pt_a, c.pt_b, c.pt_mu, c.pt_z_range, c.pt_sigma2 = (_ptr(P_[k], c_float_p) for k in self.PT_F32)
        if nsg:
            S_ = self.seg
            S_["ref_frame"], S_["cur_frame"], S_["level"] = i32(seg["ref_frame"]), i32(seg["cur_frame"]), i32(seg["level"])
            for k, w in (("px", 2), ("f", 3), ("sf", 3), ("ef", 3)):
                S_[k] = _f64(seg[k], w * nsg).reshape(-1, w)
            for k in self.SEG_F32:
                S_[k] = f32(seg[k])
            c.seg_ref_frame, c.seg_cur_frame, c.seg_level = (_ptr(S_[k], c_i32_p) for k in ("ref_frame", "cur_frame", "level"))
            c.seg_px, c.seg_f, c.seg_sf, c.seg_ef = (_ptr(S_[k], c_double_p) for k in ("px", "f", "sf", "ef"))
            (c.seg_a, c.seg_b, c.seg_mu_s, c.seg_mu_e, c.seg_z_range_s, c.seg_z_range_e, c.seg_sigma2_s,
             c.seg_sigma2_e) = (_ptr(S_[k], c_float_p) for k in self.SEG_F32)
        self.c = c

    OUT_PT = (("pt_status", np.int32, 1), ("pt_a", np.float32, 1), ("pt_b", np.float32, 1), ("pt_mu", np.float32, 1), ("pt_sigma2", np.float32, 1),
              ("pt_xyz_world", np.float64, 3), ("pt_px_cur", np.float64, 2), ("pt_depth", np.float64, 1))
    OUT_SEG = (("seg_status", np.int32, 1), ("seg_a", np.float32, 1), ("seg_b", np.float32, 1), ("seg_mu_s", np.float32, 1), ("seg_mu_e", np.float32, 1),
               ("seg_sigma2_s", np.float32, 1), ("seg_sigma2_e", np.float32, 1), ("seg_xyz_world_s", np.float64, 3), ("seg_xyz_world_e", np.float64, 3),
               ("seg_depth_s", np.float64, 1), ("seg_depth_e", np.float64, 1))

    def make_out(self):
        o = SeedsOut()
        bufs = {}
        ptr_t = {np.int32: c_i32_p, np.float32: c_float_p, np.float64: c_double_p}
        for spec, n in ((self.OUT_PT, self.n_pt), (self.OUT_SEG, self.n_seg)):
            for name, dt, w in spec:
                bufs[name] = np.zeros((max(n, 1), w) if w > 1 else max(n, 1), dtype=dt)
                setattr(o, name, bufs[name].ctypes.data_as(ptr_t[dt]))
        return o, bufs

    def trim(self, bufs):
        out = {}
        for spec, n in ((self.OUT_PT, self.n_pt), (self.OUT_SEG, self.n_seg)):
            for name, _, _ in spec:
                out[name] = bufs[name][:n].copy()
        return out


class ChainJob:
    """One stream of a resident frame step (plsvo_chain_in): an AlignJob plus the map candidates of one keyframe, ordered
    [points | segment start points | segment end points]."""

    def __init__(self, align_job, T_prev_w, T_kf_w, kf_slot, n_cand_pt, n_cand_seg, pos, ref_px, ref_f, ref_level=None, ref_type=None,
                 ref_grad=None, active=None):
        nc = int(n_cand_pt) + 2 * int(n_cand_seg)
        self.align_job = align_job
        self.pos = _f64(pos, 3 * nc).reshape(-1, 3)
        self.ref_px = _f64(ref_px, 2 * nc).reshape(-1, 2)
        self.ref_f = _f64(ref_f, 3 * nc).reshape(-1, 3)
        self.ref_level = np.ascontiguousarray(np.zeros(nc, np.int32) if ref_level is None else ref_level, dtype=np.int32)
        self.ref_type = None if ref_type is None else np.ascontiguousarray(ref_type, dtype=np.uint8)
        self.ref_grad = None if ref_grad is None else _f64(ref_grad, 2 * nc)
        self.active = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        self.n_cand_pt, self.n_cand_seg, self.n_cand = int(n_cand_pt), int(n_cand_seg), nc
        c = ChainIn()
        c.align = align_job.c
        c.T_prev_w[:] = list(_f64(T_prev_w, 7))
        c.T_kf_w[:] = list(_f64(T_kf_w, 7))
        c.kf_slot, c.n_cand_pt, c.n_cand_seg = int(kf_slot), int(n_cand_pt), int(n_cand_seg)
        c.pos, c.ref_px, c.ref_f = _ptr(self.pos, c_double_p), _ptr(self.ref_px, c_double_p), _ptr(self.ref_f, c_double_p)
        c.ref_level = _ptr(self.ref_level, c_i32_p)
        c.ref_type = C.cast(None, c_u8_p) if self.ref_type is None else self.ref_type.ctypes.data_as(c_u8_p)
        c.ref_grad = C.cast(None, c_double_p) if self.ref_grad is None else self.ref_grad.ctypes.data_as(c_double_p)
        c.active = C.cast(None, c_u8_p) if self.active is None else self.active.ctypes.data_as(c_u8_p)
        self.c = c


class ChainResult:
    def __init__(self, out, job, seg_alive, pt_keep, seg_keep, found, px, level, sel_pt, sel_seg):
        self.align = AlignResult(out.align, seg_alive)
        self.pose = PoseOptResult(out.pose, pt_keep[:out.n_sel_pt].copy(), seg_keep[:out.n_sel_seg].copy())
        self.found, self.px, self.search_level = found[:job.n_cand].astype(bool), px[:job.n_cand].copy(), level[:job.n_cand].copy()
        self.sel_pt, self.sel_seg = sel_pt[:out.n_sel_pt].copy(), sel_seg[:out.n_sel_seg].copy()
