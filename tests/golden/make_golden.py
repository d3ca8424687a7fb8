"""Generates the committed golden fixtures tests/golden/*.npz.

The reference ships no golden vectors and cannot be built here (DESIGN.md "oracle"), so these are produced
by this repo's CPU oracle (oracle/plsvo_oracle.c) on small synthetic inputs; they pin the oracle against
regressions and give the GPU box inputs+expected outputs that do not depend on regenerating anything.
Each file holds the exact inputs (level-0 images, flattened features, parameters) and the oracle's
outputs, including the per-iteration trace.   Run:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
P = importlib.import_module("pl-svo_amd")
from oracle import binding as ob  # noqa: E402
import helpers as Hh  # noqa: E402

ALIGN = [("align_small", 9001, 160, 120, 20, 8, 3, 2, 0), ("align_cfg1_crop", 9002, 320, 240, 60, 0, 3, 2, 0),
         ("align_lines", 9003, 320, 240, 30, 16, 4, 3, 1)]
POSE = [("pose_small", 9101, 40, 16, -1), ("pose_tenarg", 9102, 60, 20, 3), ("pose_cfg5", 9103, 500, 200, -1)]


def main():
    ob.build()
    for tag, seed, W, H, npts, nseg, nlev, maxl, minl in ALIGN:
        st, ref, cur, job = Hh.make_case(ob, seed, W, H, npts, nseg, nlev, maxl, minl)
        res, log = ob.sparse_align(job, ref, cur, max_log=200)
        np.savez_compressed(
            os.path.join(HERE, tag + ".npz"), ref0=ref[0], cur0=cur[0], n_levels=nlev, max_level=maxl, min_level=minl, n_iter=30,
            eps=1e-6, cam=np.array(st.cam, float), T_init=st.T_init, T_true=st.T_true, T_ref_w=st.T_ref_w, pt_px=st.pt_px,
            pt_xyz_ref=st.pt_xyz_ref, seg_spx=st.seg_spx, seg_epx=st.seg_epx, seg_len=st.seg_len, seg_p_ref=st.seg_p_ref,
            seg_q_ref=st.seg_q_ref, out_T=res.T, out_n_meas=res.n_meas, out_iters=np.array(res.iters_per_level),
            out_alive=res.seg_alive, out_chi2=res.chi2, out_H=res.H, out_status=res.status,
            log_level=np.array([r["level"] for r in log]), log_iter=np.array([r["iter"] for r in log]),
            log_accepted=np.array([r["accepted"] for r in log]), log_n_meas=np.array([r["n_meas"] for r in log]),
            log_chi2=np.array([r["new_chi2"] for r in log]), log_H=np.array([r["H"] for r in log]),
            log_Jres=np.array([r["Jres"] for r in log]), log_x=np.array([r["x"] for r in log]),
            log_T=np.array([r["T_after"] for r in log]))
        print(tag, "iters", res.iters_per_level[:nlev], "n_meas", res.n_meas)
    for tag, seed, npts, nseg, nref in POSE:
        fr = P.synth.make_poseopt_frame(seed, npts, nseg)
        res, log = ob.pose_optimize(P.poseopt_job_from_frame(fr, n_iter_ref=nref), max_log=40)
        np.savez_compressed(
            os.path.join(HERE, tag + ".npz"), T_init=fr.T_init, fx=fr.fx, reproj_thresh=2.0, n_iter=10, n_iter_ref=nref,
            pt_f=fr.pt_f, pt_pos=fr.pt_pos, pt_level=fr.pt_level, seg_line=fr.seg_line, seg_spos=fr.seg_spos, seg_epos=fr.seg_epos,
            seg_level=fr.seg_level, out_T=res.T, out_cov=res.cov, out_scale=res.estimated_scale, out_error_init=res.error_init,
            out_error_final=res.error_final, out_num_obs=np.array([res.num_obs_pt, res.num_obs_ls]), out_pt_keep=res.pt_keep,
            out_seg_keep=res.seg_keep, out_iters=np.array([res.iters, res.iters_ref]),
            log_chi2=np.array([r["new_chi2"] for r in log]), log_A=np.array([r["A"] for r in log]),
            log_dT=np.array([r["dT"] for r in log]), log_T=np.array([r["T_after"] for r in log]))
        print(tag, "iters", res.iters, res.iters_ref)


def widened_rows():
    """fixtures of the widened rows: structure optimisation, reprojection + direct matching, depth-filter seeds"""
    seqm = importlib.import_module("pl-svo_amd.sequence")
    # structure optimisation
    sb = P.synth.make_structure_batch(9201, 16, 12, 5)
    so = ob.structure_optimize(P.structopt_job_from_batch(sb))
    np.savez_compressed(os.path.join(HERE, "structopt_small.npz"), **{k: sb[k] for k in ("frame_T", "pt_pos", "pt_obs_off", "pt_obs_frame", "pt_obs_f",
                        "seg_spos", "seg_epos", "seg_obs_off", "seg_obs_frame", "seg_obs_sf", "seg_obs_ef")}, **{"out_" + k: v for k, v in so.items()})
    # reprojection + direct matching on a 160x120 keyframe / current-frame pair
    st, d = P.synth.make_match_batch(9301, 160, 120, 24, 6, zoom=0.2, edgelet_frac=0.3, levels=(0, 1), level_p=(0.7, 0.3))
    imgs = P.synth.render_streams([st]).numpy()[0]
    frames = [ob.build_pyramid(imgs[0], 4), ob.build_pyramid(imgs[1], 4)]
    rp = ob.reproject(P.abi.ReprojectJob(d["cam"], d["frame_T"], d["cur_frame"], d["pos"], cell_size=30))
    mr = ob.match_direct(P.match_job_from_batch(d), frames)
    keys = ("frame_T", "frame_slot", "cur_frame", "ref_frame", "ref_px", "ref_f", "ref_level", "ref_type", "ref_grad", "pos", "px_cur")
    np.savez_compressed(os.path.join(HERE, "match_small.npz"), img0=imgs[0], img1=imgs[1], cam=np.array(d["cam"], float), **{k: d[k] for k in keys},
                        out_reproj_px=rp["px"], out_reproj_cell=rp["cell"], **{"out_" + k: v for k, v in mr.items()})
    print("match_small found", int(mr["found"].sum()), "of", len(mr["found"]))
    # depth-filter seeds: keyframe 0, updated with frames 1 and 2
    seq = seqm.make_sequence(9401, n_frames=3, W=160, H=120, n_pts=24, n_seg=6, step_scale=1.0)
    fr = [ob.build_pyramid(im, 4) for im in seq["images"]]
    pt, seg, _ = P.synth.make_seeds(seq, cur_frame=1)
    outs = {}
    for k in (1, 2):
        pt["cur_frame"], seg["cur_frame"] = np.full(len(pt["px"]), k, np.int32), np.full(len(seg["px"]), k, np.int32)
        if k == 1:
            inputs = {"pt_" + n: np.array(v) for n, v in pt.items()}
            inputs.update({"seg_" + n: np.array(v) for n, v in seg.items()})
        r = ob.update_seeds(P.abi.SeedsJob(seq["cam"], seq["poses_true"], np.arange(3), pt, seg), fr)
        outs.update({f"out{k}_" + n: v for n, v in r.items()})
        P.synth.apply_seed_update(pt, seg, r)
    np.savez_compressed(os.path.join(HERE, "seeds_small.npz"), img0=seq["images"][0], img1=seq["images"][1], img2=seq["images"][2],
                        cam=np.array(seq["cam"], float), frame_T=seq["poses_true"], **inputs, **outs)
    print("seeds_small statuses", np.bincount(outs["out1_pt_status"], minlength=5), np.bincount(outs["out2_pt_status"], minlength=5))


def frame_chain():
    """fixture of the frame step chained as FrameHandlerMono::processFrame chains it (pl-svo_amd/sequence.py on the oracle): a 4-frame
    sequence, the per-frame poses / covariances / match counts of the oracle's chain"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_sequence import OracleBackend
    seqm = importlib.import_module("pl-svo_amd.sequence")
    seq = seqm.make_sequence(9501, n_frames=4, W=320, H=240, n_pts=60, n_seg=12)
    res = seqm.run_sequence(OracleBackend(ob), seq)
    keys = ("poses_true", "pt_pos", "pt_px0", "pt_f0", "seg_spos", "seg_epos", "seg_spx0", "seg_epx0", "seg_sf0", "seg_ef0")
    np.savez_compressed(os.path.join(HERE, "chain_small.npz"), images=np.stack(seq["images"]), cam=np.array(seq["cam"], float), **{k: seq[k] for k in keys},
                        out_T=np.stack([r["T"] for r in res]), out_cov=np.stack([r["cov"] for r in res]),
                        out_n_matched_pt=np.array([r["n_matched_pt"] for r in res]), out_n_matched_seg=np.array([r["n_matched_seg"] for r in res]),
                        out_n_kept_pt=np.array([r.get("n_kept_pt", -1) for r in res]), out_n_kept_seg=np.array([r.get("n_kept_seg", -1) for r in res]))
    print("chain_small matched", [r["n_matched_pt"] for r in res], [r["n_matched_seg"] for r in res])


if __name__ == "__main__":
    main()
    widened_rows()
    frame_chain()
