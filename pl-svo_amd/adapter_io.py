"""Binary input of pl-svo_amd/host/adapter_driver (the C++ drop-in self-test / latency driver): frames with feature
lists as FrameHandlerMono::processFrame sees them (src/frame_handler_mono.cpp:266-274, 327-329).  Used by
tests/test_gpu_adapter.py and by bench.py's per-call latency leg."""
import numpy as np


def write_adapter_input(path, st, ref_pyr, cur_pyr, fr, n_levels, max_level, min_level, n_dead_seg=0):
    """st: synth alignment stream, ref_pyr/cur_pyr: lists of u8 level images, fr: synth pose-opt frame."""
    W, H = int(ref_pyr[0].shape[1]), int(ref_pyr[0].shape[0])
    npts, nseg = len(st.pt_px), len(st.seg_len)
    blob = [np.array([W, H, n_levels, max_level, min_level, npts, nseg, len(fr.pt_f), len(fr.seg_line), n_dead_seg, 0, 0], float),
            np.array(st.cam[:4], float), st.T_ref_w, st.T_cur_w_init]
    with open(path, "wb") as f:
        for a in blob:
            np.asarray(a, np.float64).tofile(f)
        for pyr in (ref_pyr, cur_pyr):
            for l in pyr[:n_levels]:
                np.ascontiguousarray(l, np.uint8).tofile(f)
        np.hstack([st.pt_px, st.pt_f, st.pt_pos_w]).astype(np.float64).tofile(f)
        if nseg:
            np.hstack([st.seg_spx, st.seg_epx, st.seg_sf, st.seg_ef, st.seg_spos_w, st.seg_epos_w, st.seg_len[:, None]]).astype(np.float64).tofile(f)
        np.asarray(fr.T_init, np.float64).tofile(f)
        np.hstack([fr.pt_f, fr.pt_pos, fr.pt_level[:, None].astype(float)]).astype(np.float64).tofile(f)
        if len(fr.seg_line):
            np.hstack([fr.seg_line, fr.seg_spos, fr.seg_epos, fr.seg_level[:, None].astype(float)]).astype(np.float64).tofile(f)
