"""Model (NumPy, CPU) of the lead recorded in DESIGN.md 8: the reference's SEQUENTIAL float sum of non-negative terms
(src/sparse_img_align.cpp:484) evaluated slot-parallel and still bit-exact.

The device today re-adds the ~3300 terms of a near tie one after the other on one lane (align_kernels.hip::exact_chi2_pair).  The
property that would let 64 lanes share the work: while the running sum stays inside one binade [2^e, 2^(e+1)), every addition rounds
ITS TERM to the binade's quantum, s' = s + RN_ulp(t) -- the rounded terms add exactly and in any order -- except when the term sits
exactly half a quantum from a multiple (round-to-even then looks at the running mantissa's parity) or the addition crosses into the
next binade.  So: a slot (16 terms) whose exact double prefix says "this binade throughout, comfortably" is reduced independently
under that binade; a pass over the slots then checks each prediction against the true running sum and redoes the few slots that
straddle a crossing, hold a half-quantum term, or were mispredicted, term by term.

This file is the executable statement of that algorithm and the evidence that it is exact; nothing in the product depends on it."""
import numpy as np
import pytest

F = np.float32


def seq_sum(t):
    s = F(0.0)
    for x in t:
        s = F(s + x)
    return s


def slot_parallel_sum(t, slot=16, margin=2.0 ** -10):
    """-> (sum bit-identical to seq_sum(t), number of slots that took the independent path, number of slots)"""
    t = np.asarray(t, F)
    n = (len(t) + slot - 1) // slot * slot
    t = np.concatenate([t, np.zeros(n - len(t), F)])
    slots = t.reshape(-1, slot)
    prefix = np.concatenate([[0.0], np.cumsum(slots.astype(np.float64).sum(axis=1))])     # exact (to double rounding) sum before each slot
    # ---- independent part: what a lane can decide from its own slot and the predicted prefix ----
    delta = np.zeros(len(slots))
    e_pred = np.zeros(len(slots), int)
    safe = np.zeros(len(slots), bool)
    for p, ts in enumerate(slots):
        lo, hi = prefix[p], prefix[p + 1]
        if not (lo > 0.0 and np.isfinite(hi)):
            continue
        e = int(np.floor(np.log2(lo)))
        if not (lo * (1.0 - margin) >= 2.0 ** e and hi * (1.0 + margin) < 2.0 ** (e + 1)):
            continue                                                       # near a binade boundary: decided by the true running sum
        if e - 23 < -126 or e > 126:
            continue
        C = F(2.0 ** e)
        ulp = 2.0 ** (e - 23)
        r = (C + ts).astype(F) - C                                         # each term rounded to the binade's quantum (exact subtraction)
        half = np.abs(ts.astype(np.float64) - r.astype(np.float64)) == ulp / 2
        if half.any():
            continue                                                       # round-to-even depends on the running parity
        delta[p], e_pred[p], safe[p] = r.astype(np.float64).sum(), e, True
    # ---- verification pass, in slot order ----
    s = F(0.0)
    fast = 0
    for p, ts in enumerate(slots):
        ok = False
        if safe[p] and s > 0:
            e = e_pred[p]
            if 2.0 ** e <= float(s) and float(s) + delta[p] < 2.0 ** (e + 1):
                s_new = F(float(s) + delta[p])
                assert float(s_new) == float(s) + delta[p]                 # the rounded terms add exactly inside the binade
                s, ok = s_new, True
                fast += 1
        if not ok:
            for x in ts:
                s = F(s + x)
    return s, fast, len(slots)


def chi2_like_terms(rng, n, scale=6.0):
    res = rng.laplace(0.0, scale, n).astype(F)
    w = (1.0 / (1.0 + np.abs(res).astype(np.float64))).astype(F)
    return ((res * res).astype(F) * w).astype(F)


@pytest.mark.parametrize("n", [16, 100, 1600, 3200, 6400])
def test_slot_parallel_sum_is_bit_exact_on_chi2_like_terms(n):
    rng = np.random.default_rng(n)
    fast_total = slots_total = 0
    for _ in range(12 if n <= 1600 else 5):
        t = chi2_like_terms(rng, n)
        got, fast, ns = slot_parallel_sum(t)
        assert got.tobytes() == seq_sum(t).tobytes()
        fast_total += fast; slots_total += ns
    if n >= 1600:
        assert fast_total > 0.85 * slots_total       # measured: 94 % of the slots at n = 3200, 96 % at 6400


def test_slot_parallel_sum_on_adversarial_terms():
    rng = np.random.default_rng(7)
    cases = [
        np.zeros(64, F),                                                    # a static camera: every residual is exactly 0
        np.full(3200, F(0.5)),                                              # exact half-quanta all the way (every add above 2^23 * 0.5 is a tie)
        np.full(5000, F(1.0)),
        (rng.integers(0, 4, 3200) * 0.25).astype(F),                        # multiples of 1/4: ties as soon as the sum passes 2^22
        np.concatenate([np.full(100, F(1e-30)), chi2_like_terms(rng, 1000)]),   # tiny terms first
        np.concatenate([chi2_like_terms(rng, 1000), np.full(16, F(1e6))]),      # a binade jump of many binades inside one slot
        (chi2_like_terms(rng, 3200) * F(2.0 ** 60)).astype(F),
        (chi2_like_terms(rng, 3200) * F(2.0 ** -60)).astype(F),
        np.array([3.0e38, 3.0e38, 1.0], F),                                 # overflow to +inf, like the sequential sum
    ]
    for t in cases:
        with np.errstate(over="ignore"):
            got, _, _ = slot_parallel_sum(t)
            want = seq_sum(t)
        assert got.tobytes() == want.tobytes()


def test_additive_model_statement():
    """the property itself, on one binade: s + t rounds to s + RN_quantum(t) when no half-quantum term and no crossing are involved"""
    rng = np.random.default_rng(3)
    e = 13
    C, ulp = F(2.0 ** e), 2.0 ** (e - 23)
    for _ in range(20000):
        s = F(2.0 ** e + rng.integers(0, 2 ** 22) * ulp)
        t = F(rng.uniform(0.0, 40.0))
        r = F(F(C + t) - C)
        if abs(float(t) - float(r)) == ulp / 2 or float(s) + float(r) >= 2.0 ** (e + 1):
            continue
        assert float(F(s + t)) == float(s) + float(r)


# ------------------------------------------------------------------------------------------------------------------------------
# The same algorithm laid out the way one half-wave (32 lanes, one slot per lane, rounds of 32 slots) runs it on the device
# (align_kernels.hip::exact_sum_slots32): lane-local work, two exclusive scans, a first-failing-lane loop.  Mirrors the kernel
# statement by statement so that the kernel can be checked against it.
# ------------------------------------------------------------------------------------------------------------------------------
def half_wave_sum(t, margin=2.0 ** -10):
    t = np.asarray(t, F)
    n_slots = (len(t) + 15) // 16
    rounds = (n_slots + 31) // 32
    pad = rounds * 32 * 16
    t = np.concatenate([t, np.zeros(pad - len(t), F)]).reshape(rounds, 32, 16)
    s = F(0.0)                 # the true running float sum (wave-uniform inside the half)
    base = 0.0                 # predicted prefix: plain double sum of everything before the round
    redone = 0
    for k in range(rounds):
        tt = t[k]                                            # lane l holds tt[l, :]
        d = tt.astype(np.float64).sum(axis=1)               # lane-local double sum
        excl = np.concatenate([[0.0], np.cumsum(d)[:-1]])   # exclusive scan over the 32 lanes
        lo, hi = base + excl, base + excl + d
        e = np.zeros(32, int); safe = np.zeros(32, bool); delta = np.zeros(32)
        top = base + d.sum()
        e_top = int(np.floor(np.log2(top))) if top > 0.0 and np.isfinite(top) else -1023
        for l in range(32):
            if not (lo[l] > 0.0 and np.isfinite(hi[l])):
                continue
            el = int(np.floor(np.log2(lo[l])))
            if el < e_top - 26:                               # the scan of delta must stay exact in double: 26 binades of spread at most
                continue
            if not (lo[l] * (1.0 - margin) >= 2.0 ** el and hi[l] * (1.0 + margin) < 2.0 ** (el + 1)) or el - 24 < -126 or el > 126:
                continue
            C = F(2.0 ** el)
            r = (C + tt[l]).astype(F) - C
            dev = tt[l] - r                                   # exact in float
            if (np.abs(dev) == F(2.0 ** (el - 24))).any():
                continue
            e[l], safe[l], delta[l] = el, True, r.astype(np.float64).sum()
        dex = np.concatenate([[0.0], np.cumsum(delta)[:-1]])  # exclusive scan of the rounded slot sums
        start = 0                                            # first lane not yet settled
        s0 = float(s)                                        # true sum in front of lane `start`
        off = 0.0                                            # dex value of lane `start`
        while start < 32:
            sl = s0 + (dex - off)                            # each lane's candidate start value, valid if every lane in [start, l) passes
            ok = safe & (sl > 0) & (2.0 ** e.astype(float) <= sl) & (sl + delta < 2.0 ** (e + 1.0))
            ok[:start] = True
            bad = np.nonzero(~ok)[0]
            f = int(bad[0]) if len(bad) else 32
            if f == 32:                                       # everyone from `start` on passes
                s0 = s0 + (dex[31] + delta[31] - off)
                break
            sf = F(s0 + (dex[f] - off))                       # exact: every lane in [start, f) passed
            for x in tt[f]:
                sf = F(sf + x)                                # the failing lane re-adds its own sixteen terms in order
            redone += 1
            s0, start = float(sf), f + 1
            off = dex[f] + delta[f] if f < 31 else 0.0        # = dex[f + 1]: the scan value in front of the next lane
            if start == 32:
                break
        s = F(s0)
        assert float(s) == s0
        base += d.sum()
    return s, redone, n_slots


@pytest.mark.parametrize("n", [7, 16, 500, 3200, 6400])
def test_half_wave_layout_is_bit_exact(n):
    rng = np.random.default_rng(100 + n)
    for _ in range(8 if n <= 500 else 4):
        t = chi2_like_terms(rng, n)
        got, redone, ns = half_wave_sum(t)
        assert got.tobytes() == seq_sum(t).tobytes()
        if n >= 3200:
            assert redone < 0.15 * ns
    for t in (np.zeros(64, F), np.full(3200, F(0.5)), (rng.integers(0, 4, 3200) * 0.25).astype(F),
              np.concatenate([np.full(100, F(1e-30)), chi2_like_terms(rng, 1000)]), (chi2_like_terms(rng, 3200) * F(2.0 ** 60)).astype(F),
              (chi2_like_terms(rng, 3200) * F(2.0 ** -60)).astype(F),
              # 40 binades between the first slots' running sum and the end of the same 32-slot round
              np.concatenate([np.full(16, F(2.0 ** -36)), chi2_like_terms(rng, 16 * 31) * F(64.0)]).astype(F)):
        got, _, _ = half_wave_sum(t)
        assert got.tobytes() == seq_sum(t).tobytes()
