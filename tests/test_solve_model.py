"""CPU model of the device's 6x6 solve (pl-svo_amd/csrc/plsvo_wave.hpp::wave_solve6_core): Gauss-Jordan elimination of [H | b] in
the pivot order of Eigen's LDLT with the 3.2 zero-pivot rule.  The HIP code itself is checked on the GPU (tests/test_gpu_parity.py);
this file checks the ALGORITHM the kernel implements -- that a right-looking Gauss-Jordan in Eigen's fixed order returns what
ldlt().solve() returns, including the zero components of rank-deficient systems, the tie order and the infinite-diagonal case --
over thousands of systems, against the oracle's restatement of Eigen."""
import numpy as np

EPS = np.finfo(np.float64).eps
DBL_MAX = np.finfo(np.float64).max


def wave_solve6_model(H, b):
    """the kernel's algorithm, scalar: same decisions, same order of eliminations (reciprocal by division: 1-ulp detail left out)"""
    n = 6
    M = np.zeros((n, n + 1))
    M[:, :n] = H
    M[:, n] = b
    key0 = np.abs(np.diag(H)).copy()
    active = [True] * n
    pos = list(range(n))
    zero_piv = [False] * n
    cutoff, last_p = 0.0, 0
    with np.errstate(all="ignore"):
        for step in range(n):
            keys = [key0[i] if active[i] else -1.0 for i in range(n)]
            finite = [k for k in keys if k == k]
            kmax = max(finite) if finite else -1.0                    # v_max_f64: a NaN operand loses
            cand = [i for i in range(n) if active[i] and keys[i] == kmax]
            if not cand:
                cand = [i for i in range(n) if active[i]][:1]         # every remaining diagonal is NaN: the first
            p = min(cand, key=lambda i: pos[i])                       # exact ties: smallest current position
            piv = M[p, p]
            for i in range(n):                                        # Eigen swaps positions `step` and pos[p]
                if active[i] and pos[i] == step and i != p:
                    pos[i] = pos[p]
            pos[p] = step
            if step == 0:
                cutoff = abs(EPS * kmax)
            if not (kmax < cutoff) and abs(piv) > cutoff:
                row = M[p].copy()
                for i in range(n):
                    if i != p:
                        M[i] = M[i] - (M[i, p] * row) / piv
            else:
                zero_piv[p] = True
            active[p] = False
            last_p = p
        x = np.zeros(n)
        tiny = 1.0 / DBL_MAX
        for i in range(n):
            d, r = M[i, i], M[i, n]
            if zero_piv[i] or not (abs(d) > tiny):
                x[i] = (r + d) if (d != d or r != r) else 0.0
            else:
                x[i] = r / d
            if cutoff > DBL_MAX:                                      # an infinite diagonal (division by a zero line residual)
                x[i] = 0.0 if (i == last_p or abs(d) <= DBL_MAX) else np.nan
    return x


def test_full_rank_systems_agree_with_ldlt(ob):
    rng = np.random.default_rng(41)
    worst = 0.0
    for k in range(2000):
        cond_pow = rng.integers(0, 10)
        A = rng.normal(0, 1, (30, 6)) * np.logspace(0, -cond_pow / 2.0, 6)[rng.permutation(6)]
        H = A.T @ A
        b = H @ rng.normal(0, 1, 6)
        x, xo = wave_solve6_model(H, b), ob.ldlt_solve6(H, b)
        c = np.linalg.cond(H)
        err = np.linalg.norm(x - xo) / np.linalg.norm(xo)
        assert err <= 100 * c * EPS, (k, c, err)
        worst = max(worst, err / (c * EPS))
    assert worst < 100


def test_rank_deficient_systems_return_eigens_zero_components(ob):
    """J^T J of 1 or 2 point observations (structural zeros, H00 == H11 ties), random low-rank systems: the same components are
    exactly zero, the rest solve the reduced system"""
    rng = np.random.default_rng(42)
    for k in range(600):
        kind = k % 3
        if kind < 2:
            n_obs = kind + 1
            Js = [ob.jacobian_xyz2uv(p) for p in rng.uniform([-1, -1, 2], [1, 1, 6], (n_obs, 3))]
            w = rng.uniform(0.2, 1.0, n_obs)
            H = sum(wi * (J.T @ J) for wi, J in zip(w, Js))
            b = sum(wi * (J.T @ rng.normal(0, 1e-2, 2)) for wi, J in zip(w, Js))
            rank = 2 * n_obs
        else:
            rank = int(rng.integers(1, 6))
            J = rng.normal(0, 1, (rank, 6))
            H, b = J.T @ J, J.T @ rng.normal(0, 1, rank)
        x, xo = wave_solve6_model(H, b), ob.ldlt_solve6(H, b)
        # a residue pivot that lands within a factor ~2 of eps*max can be classified differently by the two roundings: skip those
        if np.count_nonzero(xo) != rank or np.count_nonzero(x) != rank:
            continue
        assert np.array_equal(x == 0.0, xo == 0.0), (k, x, xo)
        v = np.flatnonzero(xo)
        c = np.linalg.cond(H[np.ix_(v, v)])
        assert np.linalg.norm(x - xo) <= 1000 * c * EPS * np.linalg.norm(xo), (k, x, xo)


def test_rank_deficient_classification_is_usually_the_same(ob):
    """how often the skip above triggers.  One observation (rank 2): the residue pivots are exact zeros or ~1e-18 * max, far below the
    cutoff 2.2e-16 * max -- always the same classification.  Two observations (rank 4): the Schur residue of a 6x6 sum of four
    rank-1 terms reaches ~1e-16 * max, i.e. the cutoff itself, and in several percent of the systems ONE of the two formulations
    (left-looking LDLT, right-looking Gauss-Jordan) keeps a residue pivot the other drops.  Eigen says as much ("LDLT is not rank
    revealing"): with two point observations the reference's own step is decided by rounding in those cases."""
    rng = np.random.default_rng(43)
    differ = {1: 0, 2: 0}
    for k in range(600):
        n_obs = 1 + k % 2
        Js = [ob.jacobian_xyz2uv(p) for p in rng.uniform([-1, -1, 2], [1, 1, 6], (n_obs, 3))]
        H = sum(J.T @ J for J in Js)
        b = sum(J.T @ rng.normal(0, 1e-2, 2) for J in Js)
        differ[n_obs] += int(not np.array_equal(wave_solve6_model(H, b) == 0.0, ob.ldlt_solve6(H, b) == 0.0))
    assert differ[1] == 0, differ
    assert differ[2] <= 60, differ           # measured: ~13 % of 300


def test_special_systems(ob):
    assert np.array_equal(wave_solve6_model(np.zeros((6, 6)), np.ones(6)), np.zeros(6))
    # a NaN weight or residual poisons every entry of H = sum w J J^T and of Jres: x[0] is NaN, the reference's stop_ test fires
    # (a NaN in ONE off-diagonal entry with a finite diagonal cannot come out of that sum; there the two eliminations differ)
    H = np.full((6, 6), np.nan)
    assert np.isnan(wave_solve6_model(H, np.full(6, np.nan))[0]) and np.isnan(ob.ldlt_solve6(H, np.full(6, np.nan))[0])
    H = np.eye(6) * np.array([5.0, 4.0, 3.0, 2.0, 1.0, 0.5]) + 0.01
    assert np.isnan(wave_solve6_model(H, np.full(6, np.nan))[0]) and np.isnan(ob.ldlt_solve6(H, np.full(6, np.nan))[0])
    # infinite diagonal everywhere (static camera): NaN except the component pivoted last -- what the reference's stop_ test reads
    H = np.full((6, 6), np.inf)
    H[0, 1] = H[1, 0] = np.nan
    x, xo = wave_solve6_model(H, np.zeros(6)), ob.ldlt_solve6(H, np.zeros(6))
    assert np.isnan(x[0]) and np.isnan(xo[0])
    assert np.array_equal(np.isnan(x), np.isnan(xo)), (x, xo)
    # exact ties on the diagonal: same solution whichever goes first
    D = np.diag([3.0, 3.0, 3.0, 1.0, 1.0, 1.0]) + 0.1
    b = np.arange(6.0)
    assert np.allclose(wave_solve6_model(D, b), ob.ldlt_solve6(D, b), rtol=1e-13)


def test_five_scalars_per_patch_reproduce_the_per_pixel_sums(ob):
    """the identity the alignment kernel is built on: a patch's pixel Jacobians are J = fs (dx r0 + dy r1) with r0, r1
    the rows of the patch's 2x6 projection Jacobian, so sum w J J^T = fs^2 (A r0 r0^T + B (r0 r1^T + r1 r0^T) + C r1 r1^T) and
    sum w r J = fs (D r0 + E r1) with five scalars per patch -- checked against the reference's per-pixel accumulation
    (src/sparse_img_align.cpp:262-264, 485-492)"""
    rng = np.random.default_rng(44)
    for _ in range(200):
        xyz = rng.uniform([-2, -2, 1], [2, 2, 8])
        r = ob.jacobian_xyz2uv(xyz)
        fs = 416.0 / (1 << int(rng.integers(0, 4)))
        dx, dy = rng.normal(0, 8, 16).astype(np.float32), rng.normal(0, 8, 16).astype(np.float32)
        res = rng.normal(0, 5, 16).astype(np.float32)
        w = (1.0 / (1.0 + np.abs(res.astype(np.float64)))).astype(np.float32)
        H, g = np.zeros((6, 6)), np.zeros(6)
        for k in range(16):                                        # the reference: one 6-vector per pixel
            J = (float(dx[k]) * r[0] + float(dy[k]) * r[1]) * fs
            H += np.outer(J, J) * float(w[k])
            g -= J * float(res[k]) * float(w[k])
        wd, dxd, dyd, rd = (a.astype(np.float64) for a in (w, dx, dy, res))
        A, B, C = np.sum(wd * dxd * dxd), np.sum(wd * dxd * dyd), np.sum(wd * dyd * dyd)
        D, E = np.sum(wd * rd * dxd), np.sum(wd * rd * dyd)
        H5 = fs * fs * (A * np.outer(r[0], r[0]) + B * (np.outer(r[0], r[1]) + np.outer(r[1], r[0])) + C * np.outer(r[1], r[1]))
        g5 = -fs * (D * r[0] + E * r[1])
        assert np.allclose(H5, H, rtol=1e-12, atol=1e-12 * np.abs(H).max())
        assert np.allclose(g5, g, rtol=1e-12, atol=1e-12 * np.abs(g).max())
