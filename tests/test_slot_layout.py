"""Host logic of the alignment path on the CPU: the static patch-slot layout plsvo_align_stage hands the kernel
(plsvo_align_slot_layout, host-only).  The kernel's line re-weighting relies on its guarantees: all samples of a segment with
N <= 32 sit inside ONE aligned group of 32 slots (one wave-round), nothing overlaps, points keep their index."""
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
                                  (4, 320, 240, 37, 11, 2, None), (5, 1280, 720, 60, 14, 1, (600.0, 1100.0)), (6, 640, 480, 64, 0, 3, None)])
def test_slot_layout_properties(ob, case):
    seed, W, H, n_pts, n_seg, max_level, seg_len_range = case
    st, job = _job(seed, W, H, n_pts, n_seg, max_level, seg_len_range)
    for level in range(max_level, -1, -1):
        first, n, n_slots, long_lines, n_patches = P.capi.align_slot_layout(job, level)
        placed = first >= 0
        # sample counts are the reference's (LineFeat::setupSampling + the per-level reduction)
        assert np.array_equal(n[placed], _expected_samples(ob, st, level)[placed])
        # points keep their index; segments start at the next multiple of 32 and never overlap, in feature order
        if n_seg:
            base = (n_pts + 31) & ~31
            assert (first[placed] >= base).all()
            order = np.argsort(first[placed], kind="stable")
            assert np.array_equal(order, np.arange(placed.sum())), "segments keep their feature order"
            ends = first[placed] + n[placed]
            assert (first[placed][1:] >= ends[:-1]).all(), "slot ranges overlap"
            assert n_slots == max(int(ends.max()) if placed.any() else 0, n_pts)
            # a segment with N <= 32 samples sits inside one aligned group of 32 slots
            short = placed & (n <= 32)
            assert ((first[short] // 32) == ((first[short] + n[short] - 1) // 32)).all()
            assert long_lines == bool((n[placed] > 32).any())
            # padding is bounded: fewer than 31 wasted slots per 32-group boundary crossed
            assert n_slots - base <= int(n[placed].sum()) + 31 * (int(n[placed].sum()) // 32 + 1)
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
