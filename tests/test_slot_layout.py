"""Host logic of the alignment path on the CPU: the static patch-slot layout plsvo_align_stage hands the kernel
(plsvo_align_slot_layout, host-only).  The kernel's line re-weighting relies on its guarantees: all samples of a segment with
N <= 64 sit inside ONE aligned group of 64 slots (one wave-round), nothing overlaps, points keep their index -- and its speed on the
packing: the segments are fitted into the rounds first-fit in decreasing N, so the rounds a pass walks are as few as the patches allow."""
import importlib

import numpy as np
import pytest

P = importlib.import_module("pl-svo_amd")


def _job(seed, W, H, n_pts, n_seg, max_level, seg_len_range=None, alive=None):
    st = P.synth.make_align_stream(seed, W, H, n_pts, n_seg, max_level=max_level, seg_len_range=seg_len_range)
    return st, P.abi.AlignJob(st.cam, max_level, 0, 30, 1e-6, st.T_init, st.pt_px, st.pt_xyz_ref, st.seg_spx, st.seg_epx, st.seg_len,
                              st.seg_p_ref, st.seg_q_ref, seg_alive_in=alive)


def _expected_samples(ob, st, level):
    n0 = np.array([ob.setup_sampling(s, e, L)[0] for s, e, L in zip(st.seg_spx, st.seg_epx, st.seg_len)], dtype=np.int64)
    return 1 + (n0 - 1) // (1 << level)             # src/sparse_img_align.cpp:320


@pytest.mark.parametrize("case", [(1, 640, 480, 200, 80, 3, None), (2, 1280, 720, 400, 150, 4, None), (3, 640, 480, 0, 60, 2, None),
                                  (4, 320, 240, 37, 11, 2, None), (5, 1920, 1080, 60, 14, 1, (1150.0, 1800.0)), (6, 640, 480, 64, 0, 3, None)])
def test_slot_layout_properties(ob, case):
    seed, W, H, n_pts, n_seg, max_level, seg_len_range = case
    st, job = _job(seed, W, H, n_pts, n_seg, max_level, seg_len_range)
    for level in range(max_level, -1, -1):
        first, n, n_slots, long_lines, n_patches = P.capi.align_slot_layout(job, level)
        placed = first >= 0
        # sample counts are the reference's (LineFeat::setupSampling + the per-level reduction)
        assert np.array_equal(n[placed], _expected_samples(ob, st, level)[placed])
        # points keep their index; segments follow behind them and never overlap
        if n_seg:
            assert (first[placed] >= n_pts).all()
            order = np.argsort(first[placed], kind="stable")
            fs, ns_ = first[placed][order], n[placed][order]
            assert (fs[1:] >= (fs + ns_)[:-1]).all(), "slot ranges overlap"
            ends = first[placed] + n[placed]
            assert n_slots == max(int(ends.max()) if placed.any() else 0, n_pts)
            # a segment with N <= 64 samples sits inside one aligned group of 64 slots (a wave-round)
            short = placed & (n <= 64)
            assert ((first[short] // 64) == ((first[short] + n[short] - 1) // 64)).all()
            assert long_lines == bool((n[placed] > 64).any())
            # the packing (no segment longer than a round): first fit leaves at most ONE round half empty or emptier -- a segment that
            # opens a new round did not fit into any earlier one -- and with the short segments of the benchmark workloads it reaches the
            # fewest rounds the patches allow
            if not long_lines and placed.any():
                rounds = -(-n_slots // 64)
                fill = np.bincount(np.concatenate([np.arange(f, f + k) for f, k in zip(first[placed], n[placed])] + [np.arange(n_pts)]) // 64, minlength=rounds)
                assert np.sum(fill <= 32) <= 1 + (1 if n_pts % 64 and n_pts % 64 <= 32 and fill[n_pts // 64] <= 32 else 0), fill
                if n[placed].max() <= 20:
                    assert rounds == -(-(n_pts + int(n[placed].sum())) // 64), (rounds, n_pts, int(n[placed].sum()))
        else:
            assert n_slots == n_pts and not long_lines
        assert n_patches == n_pts + int(n[placed].sum())
    if seg_len_range:
        assert P.capi.align_slot_layout(job, 0)[3], "the long-segment case must need two-pass levels at level 0"


def test_slot_layout_drops_segments_without_landmark_or_in_the_border(ob):
    alive = np.ones(80, np.uint8)
    alive[::5] = 0
    st, job = _job(7, 640, 480, 100, 80, 3, alive=alive)
    first, n, n_slots, _, _ = P.capi.align_slot_layout(job, 1)
    assert (first[::5] == -1).all() and (first[alive.astype(bool)] >= 0).all()
    # end point inside the 3-pixel border of level 3 (24 px at level 0) but outside it at level 1
    job.seg_spx[1] = [10.0, 200.0]
    f3 = P.capi.align_slot_layout(job, 3)[0]
    f1 = P.capi.align_slot_layout(job, 1)[0]
    assert f3[1] == -1 and f1[1] >= 0
    assert P.capi.align_slot_layout(job, 2)[0][1] == -1      # 10 px -> 2 at level 2: still inside the border


def test_slot_layout_rejects_bad_arguments():
    import ctypes as C
    L = P.capi.lib()
    assert L.plsvo_align_slot_layout(None, 0, None, None, None, None) == -1
