"""pl-svo_amd -- MI355X-native (gfx950) hot path of PL-SVO: sparse image alignment + pose optimisation.

Layout
  csrc/   hand-written HIP kernels + the C ABI (include/plsvo_hip.h)  -> libplsvo_hip.so
  host/   C++ adapter re-creating plsvo::SparseImgAlign / plsvo::pose_optimizer::optimizeGaussNewton
  abi.py  ctypes mirror of the ABI structs;  capi.py  ctypes binding of libplsvo_hip.so
  synth.py deterministic synthetic inputs

The directory name contains a hyphen, so import it with importlib.import_module("pl-svo_amd").
There is no CPU fallback: capi raises if libplsvo_hip.so is missing or no gfx950 device is usable.
"""
import importlib as _il

abi = _il.import_module(__name__ + ".abi")
synth = _il.import_module(__name__ + ".synth")


def __getattr__(name):
    if name in ("capi", "trajectory", "dist", "adapter_io", "sequence", "rccl"):
        return _il.import_module(__name__ + "." + name)
    raise AttributeError(name)


def align_job_from_stream(st, max_level, min_level, n_iter=30, eps=1e-6, ref_slot=0, cur_slot=1):
    """Flatten a synth.AlignStream into an abi.AlignJob (what the C++ adapter does for real frames)."""
    return abi.AlignJob(cam=st.cam, max_level=max_level, min_level=min_level, n_iter=n_iter, eps=eps,
                        T_cur_from_ref=st.T_init, pt_px=st.pt_px, pt_xyz_ref=st.pt_xyz_ref,
                        seg_spx=st.seg_spx, seg_epx=st.seg_epx, seg_len=st.seg_len,
                        seg_p_ref=st.seg_p_ref, seg_q_ref=st.seg_q_ref, ref_slot=ref_slot, cur_slot=cur_slot)


def poseopt_job_from_frame(fr, reproj_thresh=2.0, n_iter=10, n_iter_ref=-1):
    return abi.PoseOptJob(T_f_w=fr.T_init, fx=fr.fx, reproj_thresh=reproj_thresh, n_iter=n_iter,
                          pt_f=fr.pt_f, pt_pos=fr.pt_pos, pt_level=fr.pt_level, seg_line=fr.seg_line,
                          seg_spos=fr.seg_spos, seg_epos=fr.seg_epos, seg_level=fr.seg_level,
                          n_iter_ref=n_iter_ref)


def structopt_job_from_batch(d, n_iter_pts=5, n_iter_segs=5):
    return abi.StructOptJob(d["frame_T"], d["pt_pos"], d["pt_obs_off"], d["pt_obs_frame"], d["pt_obs_f"], d["seg_spos"], d["seg_epos"],
                            d["seg_obs_off"], d["seg_obs_frame"], d["seg_obs_sf"], d["seg_obs_ef"], n_iter_pts, n_iter_segs)


def match_job_from_batch(d, n_pyr_levels=3, align_max_iter=10):
    return abi.MatchJob(d["cam"], d["frame_T"], d["frame_slot"], d["cur_frame"], d["ref_frame"], d["ref_px"], d["ref_f"],
                        d["ref_level"], d["ref_type"], d["ref_grad"], d["pos"], d["px_cur"], n_pyr_levels, align_max_iter)
