"""The wave-level device code itself on the CPU: pl-svo_amd/csrc/plsvo_wave.hpp -- the header the kernels include, unchanged -- compiled
with g++ against a lock-step wave64 emulator (tests/host/emu/wave_emu.hpp: one fibre per lane, every DPP / readlane / bpermute / ballot a
rendezvous) and run by tests/host/wave_host_test.cpp.  The DPP reductions and the series exp are checked against plain loops there
("selfcheck"); the cooperative 6x6 solve is checked here against the oracle's restatement of Eigen's ldlt().solve() and against the
scalar model of tests/test_solve_model.py, over the same families of systems.  (The emulator's DPP semantics are the ISA manual's as
this repo reads them; that the same functions pass on an MI355X -- tests/test_gpu_parity.py -- closes the loop.)"""
import os
import subprocess

import numpy as np
import pytest

from test_solve_model import wave_solve6_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EPS = np.finfo(np.float64).eps
IU = [(i, j) for i in range(6) for j in range(i, 6)]


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("wave_host") / "wave_host_test")
    subprocess.run(["g++", "-O1", "-std=c++17", "-ffp-contract=off", "-Wno-unknown-pragmas", "-I", os.path.join(ROOT, "tests", "host", "emu"),
                    "-I", os.path.join(ROOT, "pl-svo_amd", "csrc"), os.path.join(ROOT, "tests", "host", "wave_host_test.cpp"), "-o", exe], check=True)
    return exe


def device_solve(exe, systems, flavour=320, mode="solve"):
    buf = np.array([[H[i, j] for i, j in IU] + list(b) for H, b in systems], dtype=np.float64).tobytes()
    out = subprocess.run([exe, mode, str(flavour)], input=buf, capture_output=True)
    assert out.returncode == 0, out.stderr.decode()[:500]
    return np.frombuffer(out.stdout, dtype=np.float64).reshape(len(systems), 6)


def test_reductions_scans_and_series_exp(harness):
    out = subprocess.run([harness, "selfcheck"], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_full_rank_solve_agrees_with_ldlt(harness, ob):
    rng = np.random.default_rng(41)
    systems = []
    for k in range(300):
        cond_pow = rng.integers(0, 10)
        A = rng.normal(0, 1, (30, 6)) * np.logspace(0, -cond_pow / 2.0, 6)[rng.permutation(6)]
        H = A.T @ A
        systems.append((H, H @ rng.normal(0, 1, 6)))
    X = device_solve(harness, systems)
    Xr = device_solve(harness, systems, mode="solve_reg")
    assert np.array_equal(X, Xr)                       # inputs handed over in registers: the same solve
    assert np.array_equal(X, device_solve(harness, systems, mode="solve_search"))   # static pivot order == per-step search, bit for bit
    for (H, b), x in zip(systems, X):
        xo = ob.ldlt_solve6(H, b)
        c = np.linalg.cond(H)
        assert np.linalg.norm(x - xo) <= 100 * c * EPS * np.linalg.norm(xo), (c, x, xo)


def test_rank_deficient_solve_returns_eigens_zero_components(harness, ob):
    rng = np.random.default_rng(42)
    systems, ranks = [], []
    for k in range(240):
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
        systems.append((H, b)); ranks.append(rank)
    X = device_solve(harness, systems)
    assert np.array_equal(X, device_solve(harness, systems, mode="solve_search"), equal_nan=True)   # zero-pivot rules included
    checked = 0
    for (H, b), rank, x in zip(systems, ranks, X):
        xm, xo = wave_solve6_model(H, b), ob.ldlt_solve6(H, b)
        # the device source and its scalar model take the same decisions (the model divides where the device multiplies by a Newton reciprocal:
        # a residue pivot within an ulp of the cutoff may still fall on the other side)
        if np.count_nonzero(x) == np.count_nonzero(xm):
            assert np.array_equal(x == 0.0, xm == 0.0), (x, xm)
        if np.count_nonzero(xo) != rank or np.count_nonzero(x) != rank:
            continue
        checked += 1
        assert np.array_equal(x == 0.0, xo == 0.0), (x, xo)
        v = np.flatnonzero(xo)
        c = np.linalg.cond(H[np.ix_(v, v)])
        assert np.linalg.norm(x - xo) <= 1000 * c * EPS * np.linalg.norm(xo), (x, xo)
    assert checked > 150


def test_special_systems_and_the_330_rule(harness, ob):
    nan6 = np.full(6, np.nan)
    Hinf = np.full((6, 6), np.inf)
    Hinf[0, 1] = Hinf[1, 0] = np.nan
    D = np.diag([3.0, 3.0, 3.0, 1.0, 1.0, 1.0]) + 0.1
    systems = [(np.zeros((6, 6)), np.ones(6)), (np.full((6, 6), np.nan), nan6),
               (np.eye(6) * np.array([5.0, 4.0, 3.0, 2.0, 1.0, 0.5]) + 0.01, nan6), (Hinf, np.zeros(6)), (D, np.arange(6.0))]
    X = device_solve(harness, systems)
    assert np.array_equal(X, device_solve(harness, systems, mode="solve_search"), equal_nan=True)
    assert np.array_equal(X[0], np.zeros(6))
    assert np.isnan(X[1][0]) and np.isnan(X[2][0])
    xo = ob.ldlt_solve6(Hinf, np.zeros(6))
    assert np.isnan(X[3][0]) and np.array_equal(np.isnan(X[3]), np.isnan(xo)), (X[3], xo)
    assert np.allclose(X[4], ob.ldlt_solve6(D, np.arange(6.0)), rtol=1e-13)
    # Eigen >= 3.2.2 (flavour 330): identical on full-rank systems; a one-observation system keeps its residue pivots
    rng = np.random.default_rng(7)
    A = rng.normal(0, 1, (20, 6))
    H = A.T @ A
    b = H @ rng.normal(0, 1, 6)
    assert np.array_equal(device_solve(harness, [(H, b)], 320), device_solve(harness, [(H, b)], 330))
    prev = ob.set_ldlt_flavour(330)
    try:
        J = ob.jacobian_xyz2uv(np.array([0.3, -0.2, 4.0]))
        H1, b1 = J.T @ J, J.T @ np.array([1e-2, -2e-2])
        x330, xo330 = device_solve(harness, [(H1, b1)], 330)[0], ob.ldlt_solve6(H1, b1)
        assert np.all(np.isfinite(x330) | np.isnan(x330))
        # exact zero pivots (structural zeros of one observation) are dropped by both
        assert np.array_equal(x330 == 0.0, xo330 == 0.0) or np.count_nonzero(x330) != np.count_nonzero(xo330)
    finally:
        ob.set_ldlt_flavour(prev)
