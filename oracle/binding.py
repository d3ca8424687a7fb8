"""ctypes binding of the CPU oracle (oracle/libplsvo_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (pl-svo_amd/) never imports this module.
"""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
abi = importlib.import_module("pl-svo_amd.abi")

_LIB_PATH = os.environ.get("PLSVO_ORACLE_LIB") or os.path.join(_HERE, "libplsvo_oracle.so")   # override: the sanitizer build (make -C oracle sanitize)


class OraclePyr(C.Structure):
    _fields_ = [("n_levels", C.c_int32), ("reserved0", C.c_int32), ("img", abi.c_u8_p * abi.MAX_LEVELS),
                ("width", C.c_int32 * abi.MAX_LEVELS), ("height", C.c_int32 * abi.MAX_LEVELS),
                ("stride", C.c_int32 * abi.MAX_LEVELS)]


def build(force=False):
    """Compile the oracle with its Makefile (gcc only; no reference sources involved)."""
    src = os.path.join(_HERE, "plsvo_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"] + (["-B"] if force else []), check=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.plsvo_oracle_sparse_align.restype = C.c_int
        L.plsvo_oracle_sparse_align.argtypes = [C.POINTER(abi.AlignIn), C.POINTER(OraclePyr), C.POINTER(OraclePyr),
                                                C.POINTER(abi.AlignOut), C.POINTER(abi.AlignIterLog), C.c_int,
                                                C.POINTER(C.c_int)]
        L.plsvo_oracle_pose_optimize.restype = C.c_int
        L.plsvo_oracle_pose_optimize.argtypes = [C.POINTER(abi.PoseOptIn), C.POINTER(abi.PoseOptOut),
                                                 C.POINTER(abi.PoseOptIterLog), C.c_int, C.POINTER(C.c_int)]
        L.plsvo_oracle_structure_optimize.restype = C.c_int
        L.plsvo_oracle_structure_optimize.argtypes = [C.POINTER(abi.StructOptIn), C.POINTER(abi.StructOptOut)]
        L.plsvo_oracle_match_direct.restype = C.c_int
        L.plsvo_oracle_match_direct.argtypes = [C.POINTER(abi.MatchIn), C.POINTER(OraclePyr), C.POINTER(abi.MatchOut)]
        L.plsvo_oracle_reproject.restype = C.c_int
        L.plsvo_oracle_reproject.argtypes = [C.POINTER(abi.ReprojectIn), C.POINTER(abi.ReprojectOut)]
        L.plsvo_oracle_trajectory_record.restype = C.c_int
        L.plsvo_oracle_trajectory_record.argtypes = [abi.c_double_p, abi.c_double_p, abi.c_double_p]
        L.plsvo_oracle_update_seeds.restype = C.c_int
        L.plsvo_oracle_update_seeds.argtypes = [C.POINTER(abi.SeedsIn), C.POINTER(OraclePyr), C.POINTER(abi.SeedsOut)]
        L.plsvo_oracle_bench.restype = C.c_longlong
        L.plsvo_oracle_bench.argtypes = [C.c_int, C.POINTER(abi.AlignIn), C.POINTER(OraclePyr), C.POINTER(OraclePyr), C.POINTER(abi.PoseOptIn),
                                         C.c_int, C.c_double, C.POINTER(C.c_double)]
        L.plsvo_oracle_halfsample.restype = None
        L.plsvo_oracle_halfsample.argtypes = [abi.c_u8_p, C.c_int, C.c_int, C.c_int, abi.c_u8_p, C.c_int, C.c_int]
        d = abi.c_double_p
        for name, args, res in [
            ("plsvo_oracle_se3_exp", [d, d], None), ("plsvo_oracle_se3_mul", [d, d, d], None),
            ("plsvo_oracle_se3_inv", [d, d], None), ("plsvo_oracle_se3_act", [d, d, d], None),
            ("plsvo_oracle_se3_matrix", [d, d, d], None), ("plsvo_oracle_ldlt_solve6", [d, d, d], C.c_int),
            ("plsvo_oracle_set_ldlt_flavour", [C.c_int], None), ("plsvo_oracle_get_ldlt_flavour", [], C.c_int),
            ("plsvo_oracle_inv6", [d, d], None), ("plsvo_oracle_jacobian_xyz2uv", [d, d], None),
            ("plsvo_oracle_setup_sampling", [d, d, C.c_double, C.c_uint64, d], C.c_uint64),
            ("plsvo_oracle_line_normal", [d, d, d], None), ("plsvo_oracle_scaled_bearing", [d, d, d, d], None),
            ("plsvo_oracle_cam2world", [C.POINTER(abi.Pinhole), d, d], None),
            ("plsvo_oracle_world2cam", [C.POINTER(abi.Pinhole), d, d], None),
            ("plsvo_oracle_tukey", [C.c_float], C.c_float),
            ("plsvo_oracle_mad_scale", [C.POINTER(C.c_float), C.c_uint64], C.c_float),
            ("plsvo_oracle_median_f64", [d, C.c_uint64], C.c_double),
            ("plsvo_oracle_zmssd", [abi.c_u8_p, abi.c_u8_p, C.c_int], C.c_int),
            ("plsvo_oracle_depth_from_triangulation", [d, d, d, d], C.c_int),
            ("plsvo_oracle_compute_tau", [d, d, C.c_double, C.c_double], C.c_double),
            ("plsvo_oracle_update_point_seed", [C.c_float, C.c_float, C.POINTER(C.c_float)], None),
        ]:
            f = getattr(L, name)
            f.argtypes = args
            f.restype = res
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(abi.c_double_p)


def make_pyr(levels):
    """levels: list of 2-D uint8 numpy arrays (level 0 first).  Returns (OraclePyr, keepalive)."""
    keep = [np.ascontiguousarray(l, dtype=np.uint8) for l in levels]
    p = OraclePyr()
    p.n_levels = len(keep)
    for i, l in enumerate(keep):
        p.img[i] = l.ctypes.data_as(abi.c_u8_p)
        p.height[i], p.width[i] = l.shape
        p.stride[i] = l.strides[0]
    return p, keep


def halfsample(img, rounding=0):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    out = np.empty((h // 2, w // 2), dtype=np.uint8)
    lib().plsvo_oracle_halfsample(img.ctypes.data_as(abi.c_u8_p), w, h, img.strides[0],
                                  out.ctypes.data_as(abi.c_u8_p), out.strides[0], rounding)
    return out


def build_pyramid(img0, n_levels, rounding=0):
    pyr = [np.ascontiguousarray(img0, dtype=np.uint8)]
    for _ in range(1, n_levels):
        pyr.append(halfsample(pyr[-1], rounding))
    return pyr


def sparse_align(job, ref_levels, cur_levels, max_log=0):
    """job: abi.AlignJob.  Returns (abi.AlignResult, [iteration log dicts])."""
    L = lib()
    rp, k1 = make_pyr(ref_levels)
    cp, k2 = make_pyr(cur_levels)
    out = abi.AlignOut()
    alive = np.ones(max(job.n_seg, 1), dtype=np.uint8)
    out.seg_alive_out = alive.ctypes.data_as(abi.c_u8_p)
    n_log = C.c_int(0)
    log = (abi.AlignIterLog * max(max_log, 1))()
    rc = L.plsvo_oracle_sparse_align(C.byref(job.c), C.byref(rp), C.byref(cp), C.byref(out),
                                     log if max_log > 0 else None, max_log, C.byref(n_log))
    if rc != 0:
        raise RuntimeError(f"oracle sparse_align failed rc={rc}")
    res = abi.AlignResult(out, alive[:job.n_seg].copy())
    return res, abi.align_log_to_dicts(log, min(n_log.value, max_log))


def pose_optimize(job, max_log=0):
    L = lib()
    out = abi.PoseOptOut()
    pk = np.ones(max(job.n_pts, 1), dtype=np.uint8)
    sk = np.ones(max(job.n_seg, 1), dtype=np.uint8)
    out.pt_keep = pk.ctypes.data_as(abi.c_u8_p)
    out.seg_keep = sk.ctypes.data_as(abi.c_u8_p)
    n_log = C.c_int(0)
    log = (abi.PoseOptIterLog * max(max_log, 1))()
    rc = L.plsvo_oracle_pose_optimize(C.byref(job.c), C.byref(out), log if max_log > 0 else None, max_log,
                                      C.byref(n_log))
    if rc != 0:
        raise RuntimeError(f"oracle pose_optimize failed rc={rc}")
    res = abi.PoseOptResult(out, pk[:job.n_pts].copy(), sk[:job.n_seg].copy())
    return res, abi.poseopt_log_to_dicts(log, min(n_log.value, max_log))


def structure_optimize(job):
    out, bufs = job.make_out()
    rc = lib().plsvo_oracle_structure_optimize(C.byref(job.c), C.byref(out))
    if rc != 0:
        raise RuntimeError(f"oracle structure_optimize failed rc={rc}")
    return job.trim(bufs)


def match_direct(job, frame_levels):
    """job: abi.MatchJob; frame_levels[k] = list of pyramid levels (uint8 arrays) of frame index k."""
    pyrs = (OraclePyr * len(frame_levels))()
    keep = []
    for k, levels in enumerate(frame_levels):
        p, kk = make_pyr(levels)
        pyrs[k] = p
        keep.append(kk)
    out, bufs = job.make_out()
    rc = lib().plsvo_oracle_match_direct(C.byref(job.c), pyrs, C.byref(out))
    if rc != 0:
        raise RuntimeError(f"oracle match_direct failed rc={rc}")
    return job.trim(bufs)


def reproject(job):
    out, bufs = job.make_out()
    rc = lib().plsvo_oracle_reproject(C.byref(job.c), C.byref(out))
    if rc != 0:
        raise RuntimeError(f"oracle reproject failed rc={rc}")
    return job.trim(bufs)


def trajectory_record(T_f_w, cov):
    T = np.ascontiguousarray(T_f_w, dtype=np.float64)
    Cv = np.ascontiguousarray(cov, dtype=np.float64).reshape(36)
    out = np.empty(7)
    ok = lib().plsvo_oracle_trajectory_record(_dp(T), _dp(Cv), _dp(out))
    return bool(ok), out


def update_seeds(job, frame_levels):
    pyrs = (OraclePyr * len(frame_levels))()
    keep = []
    for k, levels in enumerate(frame_levels):
        p, kk = make_pyr(levels)
        pyrs[k] = p
        keep.append(kk)
    out, bufs = job.make_out()
    rc = lib().plsvo_oracle_update_seeds(C.byref(job.c), pyrs, C.byref(out))
    if rc != 0:
        raise RuntimeError(f"oracle update_seeds failed rc={rc}")
    return job.trim(bufs)


def bench(align_jobs, ref_pyrs, cur_pyrs, pose_jobs, n_threads, seconds, what=3, latencies=False):
    """SparseImgAlign::run (what & 1) and/or optimizeGaussNewton (what & 2) over the given streams on n_threads POSIX threads
    (no Python in the loop).  Returns (frames completed, wall seconds) and, with latencies=True, also the per-frame wall
    times in microseconds of thread 0 (first 4096 frames)."""
    n = len(pose_jobs) if not (what & 1) else len(align_jobs)
    L = lib()
    L.plsvo_oracle_bench_mode.restype = C.c_longlong
    L.plsvo_oracle_bench_mode.argtypes = [C.c_int, C.POINTER(abi.AlignIn), C.POINTER(OraclePyr), C.POINTER(OraclePyr), C.POINTER(abi.PoseOptIn),
                                          C.c_int, C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    aj = (abi.AlignIn * n)(*[j.c for j in align_jobs[:n]]) if (what & 1) else (abi.AlignIn * n)()
    pj = (abi.PoseOptIn * n)(*[j.c for j in pose_jobs[:n]]) if (what & 2) else (abi.PoseOptIn * n)()
    rp, cp, keep = (OraclePyr * n)(), (OraclePyr * n)(), []
    if what & 1:
        for i in range(n):
            a, k1 = make_pyr(ref_pyrs[i])
            b, k2 = make_pyr(cur_pyrs[i])
            rp[i], cp[i] = a, b
            keep += [k1, k2]
    el = C.c_double(0.0)
    cap = 4096
    lat = (C.c_double * cap)()
    n_lat = C.c_int(0)
    done = L.plsvo_oracle_bench_mode(n, aj, rp, cp, pj, int(n_threads), float(seconds), int(what), C.byref(el), lat, cap, C.byref(n_lat))
    if latencies:
        return int(done), float(el.value), np.array(lat[:n_lat.value], dtype=np.float64)
    return int(done), float(el.value)


def bench_threads_pinned():
    """threads of the last bench() call that were pinned to a CPU of the process's affinity mask"""
    L = lib()
    L.plsvo_oracle_bench_threads_pinned.restype = C.c_int
    return int(L.plsvo_oracle_bench_threads_pinned())


# --- small helpers for unit tests -------------------------------------------------------------

def se3_exp(u):
    u = np.ascontiguousarray(u, dtype=np.float64)
    T = np.empty(7)
    lib().plsvo_oracle_se3_exp(_dp(u), _dp(T))
    return T


def se3_mul(A, B):
    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    Cc = np.empty(7)
    lib().plsvo_oracle_se3_mul(_dp(A), _dp(B), _dp(Cc))
    return Cc


def se3_inv(A):
    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.empty(7)
    lib().plsvo_oracle_se3_inv(_dp(A), _dp(B))
    return B


def se3_act(T, p):
    T = np.ascontiguousarray(T, dtype=np.float64)
    p = np.ascontiguousarray(p, dtype=np.float64)
    o = np.empty(3)
    lib().plsvo_oracle_se3_act(_dp(T), _dp(p), _dp(o))
    return o


def se3_matrix(T):
    T = np.ascontiguousarray(T, dtype=np.float64)
    R = np.empty(9)
    t = np.empty(3)
    lib().plsvo_oracle_se3_matrix(_dp(T), _dp(R), _dp(t))
    return R.reshape(3, 3), t


def set_ldlt_flavour(flavour):
    """320 (default; Eigen 3.1...3.2.1) or 330 (Eigen 3.3) zero-pivot rule of ldlt().solve(); returns the previous one"""
    prev = lib().plsvo_oracle_get_ldlt_flavour()
    lib().plsvo_oracle_set_ldlt_flavour(int(flavour))
    return prev


def ldlt_solve6(H, b):
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(36)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.empty(6)
    lib().plsvo_oracle_ldlt_solve6(_dp(H), _dp(b), _dp(x))
    return x


def inv6(A):
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(36)
    o = np.empty(36)
    lib().plsvo_oracle_inv6(_dp(A), _dp(o))
    return o.reshape(6, 6)


def jacobian_xyz2uv(xyz):
    xyz = np.ascontiguousarray(xyz, dtype=np.float64)
    J = np.empty(12)
    lib().plsvo_oracle_jacobian_xyz2uv(_dp(xyz), _dp(J))
    return J.reshape(2, 6)


def setup_sampling(spx, epx, length, patch_size=4):
    spx = np.ascontiguousarray(spx, dtype=np.float64)
    epx = np.ascontiguousarray(epx, dtype=np.float64)
    dif = np.empty(2)
    n = lib().plsvo_oracle_setup_sampling(_dp(spx), _dp(epx), float(length), patch_size, _dp(dif))
    return int(n), dif


def line_normal(sf, ef):
    sf = np.ascontiguousarray(sf, dtype=np.float64)
    ef = np.ascontiguousarray(ef, dtype=np.float64)
    o = np.empty(3)
    lib().plsvo_oracle_line_normal(_dp(sf), _dp(ef), _dp(o))
    return o


def tukey(x):
    return float(lib().plsvo_oracle_tukey(C.c_float(x)))


def mad_scale(errors):
    e = np.ascontiguousarray(errors, dtype=np.float32).copy()
    return float(lib().plsvo_oracle_mad_scale(e.ctypes.data_as(C.POINTER(C.c_float)), e.size))


def median_f64(v):
    v = np.ascontiguousarray(v, dtype=np.float64).copy()
    return float(lib().plsvo_oracle_median_f64(_dp(v), v.size))


def zmssd(ref_patch, cur_img, x0, y0):
    """ZMSSD<4> score of the 8x8 reference patch against the 8x8 window of cur_img whose top-left pixel is (x0, y0)"""
    ref = np.ascontiguousarray(ref_patch, dtype=np.uint8).reshape(64)
    img = np.ascontiguousarray(cur_img, dtype=np.uint8)
    ptr = C.cast(img.ctypes.data + y0 * img.strides[0] + x0, abi.c_u8_p)
    return int(lib().plsvo_oracle_zmssd(ref.ctypes.data_as(abi.c_u8_p), ptr, img.strides[0]))


def depth_from_triangulation(T_search_ref, f_ref, f_cur):
    T = np.ascontiguousarray(T_search_ref, dtype=np.float64)
    a = np.ascontiguousarray(f_ref, dtype=np.float64)
    b = np.ascontiguousarray(f_cur, dtype=np.float64)
    out = np.zeros(1)
    ok = lib().plsvo_oracle_depth_from_triangulation(_dp(T), _dp(a), _dp(b), _dp(out))
    return bool(ok), float(out[0])


def compute_tau(T_ref_cur, f, z, px_error_angle):
    T = np.ascontiguousarray(T_ref_cur, dtype=np.float64)
    a = np.ascontiguousarray(f, dtype=np.float64)
    return float(lib().plsvo_oracle_compute_tau(_dp(T), _dp(a), float(z), float(px_error_angle)))


def update_point_seed(x, tau2, a, b, mu, z_range, sigma2):
    st = (C.c_float * 5)(a, b, mu, z_range, sigma2)
    lib().plsvo_oracle_update_point_seed(C.c_float(x), C.c_float(tau2), st)
    return tuple(float(v) for v in st)
