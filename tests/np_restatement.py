"""Independent NumPy restatement of the hot path (test infrastructure only).

Written separately from oracle/plsvo_oracle.c (different language, different data structures) to pin
the C oracle: both must agree per Gauss-Newton iteration ("oracle cross-check").
Follows /root/reference src/sparse_img_align.cpp:54-710 and src/pose_optimizer.cpp:38-582, typed
float32/float64 like the reference.  `order="device"` evaluates the per-patch sums the way the HIP
kernel does (five scalars per patch + 6x6 expansion) so the kernel's algebra can be checked on a CPU.
"""
import math

import numpy as np

f32 = np.float32
PATCH = 4
HALF = 2


# ---------------------------------------------------------------------------------------------
# [ext] Sophus SE3 (quaternion + translation), tangent (upsilon, omega)
# ---------------------------------------------------------------------------------------------
def q_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def q_rot(q, v):
    qv = q[:3]
    uv = 2.0 * np.cross(qv, v)
    return v + q[3] * uv + np.cross(qv, uv)


def se3_exp(u):
    ups, om = np.asarray(u[:3], float), np.asarray(u[3:], float)
    th = math.sqrt(float(om @ om))
    if th < 1e-10:
        imag = 0.5 - 0.0208333 * th * th + 0.000260417 * th ** 4
    else:
        imag = math.sin(0.5 * th) / th
    q = np.array([imag * om[0], imag * om[1], imag * om[2], math.cos(0.5 * th)])
    q = q / math.sqrt(float(q @ q))
    if th < 1e-10:
        x, y, z, w = q
        V = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    else:
        O = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
        V = np.eye(3) + (1 - math.cos(th)) / (th * th) * O + (th - math.sin(th)) / (th * th * th) * (O @ O)
    return np.concatenate([q, V @ ups])


def se3_mul(A, B):
    q = q_mul(A[:4], B[:4])
    return np.concatenate([q / math.sqrt(float(q @ q)), A[4:] + q_rot(A[:4], B[4:])])


def se3_act(T, p):
    return q_rot(T[:4], np.asarray(p, float)) + T[4:]


def jacobian_xyz2uv(xyz):
    x, y, z = xyz
    zi = 1.0 / z
    zi2 = zi * zi
    J = np.zeros((2, 6))
    J[0, 0] = -zi
    J[0, 2] = x * zi2
    J[0, 3] = y * J[0, 2]
    J[0, 4] = -(1.0 + x * J[0, 2])
    J[0, 5] = y * zi
    J[1, 1] = -zi
    J[1, 2] = y * zi2
    J[1, 3] = 1.0 + y * J[1, 2]
    J[1, 4] = -J[0, 3]
    J[1, 5] = -x * zi
    return J


def ldlt_solve(H, b):
    """Any stable symmetric solve agrees with Eigen's pivoted LDLT to rounding * cond; a zero matrix gives 0."""
    if not np.all(np.isfinite(H)) or not np.all(np.isfinite(b)):
        return np.full(6, np.nan)
    if not np.any(H):
        return np.zeros(6)
    return np.linalg.solve(H, b)


def eigen32_ldlt_solve(H, b):
    """A.ldlt().solve(b) as Eigen 3.1 ... 3.2.1 compute it, written from the mathematics rather than from the in-place code:
    Eigen's unblocked LDLT is left-looking, so the diagonal it pivots on is never updated -- the pivot order is a selection sort of
    the ORIGINAL |diagonal| (first maximum in the order the swaps leave behind).  Permute first, then factor P A P^T = L D L^T
    without pivoting (column k = stored column minus the sum over j < k of L_ij * (D_j L_kj), summed first, subtracted once),
    scale a column only if |D_k| > cutoff = eps * largest diagonal, stop when the largest remaining stored diagonal is below the
    cutoff, and solve with the pseudo-inverse of D (|D_i| <= max(max|D| eps, DBL_MIN) -> 0).
    Independent of oracle/plsvo_oracle.c::ldlt_solve_n; tests require bitwise agreement."""
    A = np.array(H, dtype=np.float64)
    n = A.shape[0]
    eps = np.finfo(np.float64).eps
    order = list(range(n))
    cutoff = 0.0
    stop_at = n
    for k in range(n):                       # selection sort of the original diagonal, with Eigen's stopping rule
        best, bestv = k, abs(A[order[k], order[k]])
        for j in range(k + 1, n):
            v = abs(A[order[j], order[j]])
            if v > bestv:
                best, bestv = j, v
        if k == 0:
            cutoff = abs(eps * bestv)
        if bestv < cutoff:
            stop_at = k
            break
        order[k], order[best] = order[best], order[k]
    a = A[np.ix_(order, order)]
    L = np.tril(a, -1).copy()                # unprocessed columns keep the stored (raw) entries, like the in-place algorithm
    D = np.diag(a).copy()
    for k in range(stop_at):
        temp = [D[j] * L[k, j] for j in range(k)]
        if k > 0:
            acc = 0.0
            for j in range(k):
                acc += L[k, j] * temp[j]
            D[k] = a[k, k] - acc
            for i in range(k + 1, n):
                acc = 0.0
                for j in range(k):
                    acc += L[i, j] * temp[j]
                L[i, k] = a[i, k] - acc
        if k < n - 1 and abs(D[k]) > cutoff:
            with np.errstate(all="ignore"):
                L[k + 1:, k] = L[k + 1:, k] / D[k]
    y = np.array(b, dtype=np.float64)[order]
    for i in range(n):
        for j in range(i):
            y[i] -= L[i, j] * y[j]
    maxd = abs(D[0])
    for i in range(1, n):
        if abs(D[i]) > maxd:
            maxd = abs(D[i])
    tiny = 1.0 / np.finfo(np.float64).max
    tol = maxd * eps if maxd * eps > tiny else tiny
    with np.errstate(all="ignore"):
        for i in range(n):
            y[i] = y[i] / D[i] if abs(D[i]) > tol else 0.0
        for i in range(n - 1, -1, -1):
            for j in range(i + 1, n):
                y[i] -= L[j, i] * y[j]
    x = np.empty(n)
    x[order] = y
    return x


def setup_sampling(spx, epx, length):
    dif = np.asarray(epx, float) - np.asarray(spx, float)
    a0, a1 = abs(dif[0]), abs(dif[1])
    with np.errstate(all="ignore"):
        tan_dir = np.float64(min(a0, a1)) / np.float64(max(a0, a1))
    sin_dir = tan_dir / math.sqrt(1.0 + tan_dir * tan_dir) if np.isfinite(tan_dir) else float("nan")
    corr = 2.0 * math.sqrt(1.0 + sin_dir * sin_dir) if np.isfinite(sin_dir) else float("nan")
    v = length / (2.0 * PATCH * corr)
    n = 1.0 if not (v > 1.0) else v
    return int(n), dif


# ---------------------------------------------------------------------------------------------
# Patch (include/plsvo/feature.h:107-147, src/feature.cpp:175-218)
# ---------------------------------------------------------------------------------------------
class Patch:
    def __init__(self, img):
        self.img = img
        self.rows, self.cols = img.shape

    def set_position(self, px, py):
        self.u = f32(px)
        self.v = f32(py)
        self.ui = int(math.floor(float(self.u)))
        self.vi = int(math.floor(float(self.v)))

    def in_frame(self, b):
        return not (self.ui < b or self.vi < b or self.ui >= self.cols - b or self.vi >= self.rows - b)

    def weights(self):
        su = f32(self.u - f32(self.ui))
        sv = f32(self.v - f32(self.vi))
        su, sv = float(su), float(sv)
        self.wTL = f32((1.0 - su) * (1.0 - sv))
        self.wTR = f32(su * (1.0 - sv))
        self.wBL = f32((1.0 - su) * sv)
        self.wBR = f32(su * sv)

    def window(self, dr, dc):
        """4x4 float32 block of pixels at ROI offset (dr, dc)"""
        r0, c0 = self.vi - HALF + dr, self.ui - HALF + dc
        return self.img[r0:r0 + PATCH, c0:c0 + PATCH].astype(np.float32)

    def interp(self, dr=0, dc=0):
        w = self
        return w.wTL * w.window(dr, dc) + w.wTR * w.window(dr, dc + 1) + w.wBL * w.window(dr + 1, dc) + w.wBR * w.window(dr + 1, dc + 1)


def _seq_sum_f32(values):
    s = f32(0.0)
    for v in values:
        s = f32(s + f32(v))
    return s


# ---------------------------------------------------------------------------------------------
# SparseImgAlign
# ---------------------------------------------------------------------------------------------
class SparseAlignNP:
    def __init__(self, cam, max_level, min_level, n_iter, eps, order="reference"):
        self.fx, self.fy, self.cx, self.cy, self.W, self.H = cam
        self.max_level, self.min_level, self.n_iter, self.eps = max_level, min_level, n_iter, eps
        self.order = order
        self.log = []

    def world2cam(self, xyz):
        return np.array([self.fx * (xyz[0] / xyz[2]) + self.cx, self.fy * (xyz[1] / xyz[2]) + self.cy])

    def cam_in_frame(self, ox, oy, b, level):
        return b <= ox < self.W // (1 << level) - b and b <= oy < self.H // (1 << level) - b

    # -- reference patches (:195-378) --
    def _ref_patch(self, patch, xyz_ref, level):
        ref = patch.interp(0, 0)
        dx = f32(0.5) * (patch.interp(0, 1) - patch.interp(0, -1))
        dy = f32(0.5) * (patch.interp(1, 0) - patch.interp(-1, 0))
        fj = jacobian_xyz2uv(xyz_ref)
        s = abs(self.fx) / (1 << level)
        J = (dx.reshape(16, 1).astype(np.float64) * fj[0][None, :] + dy.reshape(16, 1).astype(np.float64) * fj[1][None, :]) * s
        return dict(ref=ref.reshape(16), dx=dx.reshape(16), dy=dy.reshape(16), J=J, fj=fj, xyz=np.array(xyz_ref, float))

    def precompute(self, level):
        scale = 1.0 / (1 << level)
        patch = Patch(self.ref_pyr[level])
        self.pt_cache = {}
        for i in range(len(self.pt_px)):
            patch.set_position(self.pt_px[i, 0] * scale, self.pt_px[i, 1] * scale)
            if not patch.in_frame(HALF + 1):
                continue
            patch.weights()
            self.pt_visible[i] = True
            self.pt_cache[i] = self._ref_patch(patch, self.pt_xyz[i], level)
        self.seg_cache = {}
        for s in range(len(self.seg_spx)):
            if not self.seg_alive[s]:
                continue
            spx, epx = self.seg_spx[s], self.seg_epx[s]
            if not (self.cam_in_frame(int(spx[0] * scale), int(spx[1] * scale), HALF + 1, level) and
                    self.cam_in_frame(int(epx[0] * scale), int(epx[1] * scale), HALF + 1, level)):
                continue
            self.seg_visible[s] = True
            N, inc2d = setup_sampling(spx, epx, self.seg_len[s])
            N = 1 + (N - 1) // (1 << level)
            with np.errstate(all="ignore"):
                inc2d = inc2d * scale / np.float64(N - 1)
                inc3d = (self.seg_q[s] - self.seg_p[s]) / np.float64(N - 1)
            px = spx * scale
            xyz = self.seg_p[s].copy()
            samples = []
            for _ in range(N):
                patch.set_position(px[0], px[1])
                patch.weights()
                samples.append(self._ref_patch(patch, xyz, level))
                px = px + inc2d
                xyz = xyz + inc3d
            self.seg_cache[s] = samples

    # -- residuals (:380-695) --
    def _cur_residual(self, patch, T, cache, scale):
        xyz_cur = se3_act(T, cache["xyz"])
        uv = self.world2cam(xyz_cur) * scale
        patch.set_position(uv[0], uv[1])
        if not patch.in_frame(HALF):
            return None
        patch.weights()
        cur = patch.interp(0, 0).reshape(16)
        return cur - cache["ref"]

    def _accumulate(self, cache, res, w, H, Jres):
        """H += sum_pix w J J^T ; Jres -= sum_pix w res J  (w: per-pixel float32 array)"""
        wd = w.astype(np.float64)
        rd = res.astype(np.float64)
        if self.order == "reference":
            J = cache["J"]
            H += np.einsum("pi,pj,p->ij", J, J, wd)
            Jres -= np.einsum("pi,p->i", J, rd * wd)
        else:  # the HIP kernel's formulation
            dx = cache["dx"].astype(np.float64)
            dy = cache["dy"].astype(np.float64)
            A, B, C = np.sum(wd * dx * dx), np.sum(wd * dx * dy), np.sum(wd * dy * dy)
            D, E = np.sum(wd * dx * rd), np.sum(wd * dy * rd)
            r0, r1 = cache["fj"][0], cache["fj"][1]
            fs = abs(self.fx) / (1 << self.level)
            H += fs * fs * (A * np.outer(r0, r0) + B * (np.outer(r0, r1) + np.outer(r1, r0)) + C * np.outer(r1, r1))
            Jres -= fs * (D * r0 + E * r1)

    def compute_residuals(self, T):
        level = self.level
        if not self.have_cache:
            self.precompute(level)
            self.have_cache = True
        scale = 1.0 / (1 << level)
        patch = Patch(self.cur_pyr[level])
        H = np.zeros((6, 6))
        Jres = np.zeros(6)
        chi2 = f32(0.0)
        for i in sorted(self.pt_cache.keys()):
            if not self.pt_visible[i]:
                continue
            res = self._cur_residual(patch, T, self.pt_cache[i], scale)
            if res is None:
                continue
            w = np.array([f32(1.0 / (1.0 + float(abs(r)))) for r in res], dtype=np.float32)
            for r, ww in zip(res, w):
                chi2 = f32(chi2 + f32(f32(r * r) * ww))
            self.n_meas += 16
            self._accumulate(self.pt_cache[i], res, w, H, Jres)
        seg_chi2 = f32(0.0)
        for s in range(len(self.seg_spx)):
            if not self.seg_alive[s] or not self.seg_visible[s] or s not in self.seg_cache:
                continue
            samples = self.seg_cache[s]
            N = len(samples)
            Hl = np.zeros((6, 6))
            Jl = np.zeros(6)
            allres = []
            good = True
            for c in samples:
                res = self._cur_residual(patch, T, c, scale)
                if res is None:
                    good = False
                    break
                allres.extend(list(res))
                self._accumulate(c, res, np.ones(16, dtype=np.float32), Hl, Jl)
            res_ = _seq_sum_f32([abs(r) for r in allres])
            res_ = f32(float(res_) / float(N))
            if good and float(res_) < 200.0:
                w = f32(1.0 / (1.0 + float(res_)))
                with np.errstate(all="ignore"):
                    H += Hl * float(w) / float(res_)
                Jres += Jl * float(w)
                seg_chi2 = f32(seg_chi2 + f32(f32(res_ * res_) * w))
                self.n_meas += 1
            else:
                self.seg_alive[s] = False
        self.Hm, self.Jres = H, Jres
        total = f32(chi2 + seg_chi2)
        with np.errstate(all="ignore"):
            return float(f32(total) / f32(self.n_meas))

    def run(self, T0, ref_pyr, cur_pyr, pt_px, pt_xyz, seg_spx, seg_epx, seg_len, seg_p, seg_q, seg_alive=None):
        self.ref_pyr, self.cur_pyr = ref_pyr, cur_pyr
        self.pt_px, self.pt_xyz = np.asarray(pt_px, float).reshape(-1, 2), np.asarray(pt_xyz, float).reshape(-1, 3)
        self.seg_spx, self.seg_epx = np.asarray(seg_spx, float).reshape(-1, 2), np.asarray(seg_epx, float).reshape(-1, 2)
        self.seg_len = np.asarray(seg_len, float).reshape(-1)
        self.seg_p, self.seg_q = np.asarray(seg_p, float).reshape(-1, 3), np.asarray(seg_q, float).reshape(-1, 3)
        ns = len(self.seg_spx)
        self.seg_alive = np.ones(ns, bool) if seg_alive is None else np.asarray(seg_alive, bool).copy()
        self.pt_visible = np.zeros(len(self.pt_px), bool)
        self.seg_visible = np.zeros(ns, bool)
        T = np.asarray(T0, float).copy()
        chi2_, stop, self.n_meas = 1e10, False, 0
        iters = {}
        self.Hm = np.zeros((6, 6))
        if len(self.pt_px) == 0 and ns == 0:
            return dict(T=T, n_meas=0, iters=iters, alive=self.seg_alive, H=self.Hm, chi2=chi2_, stop=stop)
        use_weights = False
        for level in range(self.max_level, self.min_level - 1, -1):
            self.level = level
            self.have_cache = False
            if use_weights:
                self.compute_residuals(T)  # the solver's weight-scale pass (idempotent here)
            use_weights = True
            old = T.copy()
            n_it = 0
            for it in range(self.n_iter):
                self.n_meas = 0
                new_chi2 = self.compute_residuals(T)
                n_it += 1
                x = ldlt_solve(self.Hm, self.Jres)
                if np.isnan(x[0]):
                    stop = True
                if (it > 0 and new_chi2 > chi2_) or stop:
                    T = old.copy()
                    self.log.append(dict(level=level, iter=it, accepted=0, n_meas=self.n_meas, new_chi2=new_chi2,
                                         H=self.Hm.copy(), Jres=self.Jres.copy(), x=x.copy(), T_after=T.copy()))
                    break
                Tn = se3_mul(T, se3_exp(-x))
                old = T.copy()
                T = Tn
                chi2_ = new_chi2
                self.log.append(dict(level=level, iter=it, accepted=1, n_meas=self.n_meas, new_chi2=new_chi2,
                                     H=self.Hm.copy(), Jres=self.Jres.copy(), x=x.copy(), T_after=T.copy()))
                if np.max(np.abs(x)) <= self.eps:
                    break
            iters[level] = n_it
        return dict(T=T, n_meas=self.n_meas, iters=iters, alive=self.seg_alive, H=self.Hm, chi2=chi2_, stop=stop)


# ---------------------------------------------------------------------------------------------
# pose_optimizer::optimizeGaussNewton (src/pose_optimizer.cpp:38-582)
# ---------------------------------------------------------------------------------------------
def tukey(x):
    x = f32(x)
    b2 = f32(f32(4.6851) * f32(4.6851))
    x2 = f32(x * x)
    if x2 <= b2:
        t = f32(f32(1.0) - f32(x2 / b2))
        return f32(t * t)
    return f32(0.0)


def median_upper(v):
    v = np.sort(np.asarray(v))
    return v[len(v) // 2]


def pose_optimize_np(T0, fx, reproj_thresh, n_iter, pt_f, pt_pos, pt_level, seg_line, seg_spos, seg_epos, seg_level,
                     n_iter_ref=-1):
    T = np.asarray(T0, float).copy()
    pt_f, pt_pos = np.asarray(pt_f, float).reshape(-1, 3), np.asarray(pt_pos, float).reshape(-1, 3)
    seg_line = np.asarray(seg_line, float).reshape(-1, 3)
    seg_spos, seg_epos = np.asarray(seg_spos, float).reshape(-1, 3), np.asarray(seg_epos, float).reshape(-1, 3)
    np_, ns = len(pt_f), len(seg_line)
    pt_keep, seg_keep = np.ones(np_, bool), np.ones(ns, bool)
    log = []

    def pt_err(i, Tm):
        x = se3_act(Tm, pt_pos[i])
        e = pt_f[i, :2] / pt_f[i, 2] - x[:2] / x[2]
        return e * (1.0 / (1 << int(pt_level[i]))), x

    def seg_d(s, Tm):
        xs, xe = se3_act(Tm, seg_spos[s]), se3_act(Tm, seg_epos[s])
        l = seg_line[s]
        return l[0] * (xs[0] / xs[2]) + l[1] * (xs[1] / xs[2]) + l[2] * 1.0, l[0] * (xe[0] / xe[2]) + l[1] * (xe[1] / xe[2]) + l[2] * 1.0, xs, xe

    errors = [f32(np.linalg.norm(pt_err(i, T)[0])) for i in range(np_)]
    scale_pt = float(f32(f32(1.48) * median_upper(np.array(errors, dtype=np.float32)))) if np_ else 1.0
    errs_ls = []
    for s in range(ns):
        es, ee, _, _ = seg_d(s, T)
        es, ee = f32(es), f32(ee)
        errs_ls.append(np.sqrt(f32(f32(es * es) + f32(ee * ee))))
    if np_ + ns == 0:
        return dict(T=T, status=1)
    scale_ls = float(f32(f32(1.48) * median_upper(np.array(errs_ls, dtype=np.float32)))) if ns else 1.0
    chi2_init, state = [], dict(T=T, T_old=T.copy(), chi2=0.0, A=np.zeros((6, 6)))

    def gn(n_it, phase):
        iters = 0
        for it in range(n_it):
            A, b, new_chi2 = np.zeros((6, 6)), np.zeros(6), 0.0
            iters += 1
            Tm = state["T"]
            for i in range(np_):
                if not pt_keep[i]:
                    continue
                e, x = pt_err(i, Tm)
                sic = 1.0 / (1 << int(pt_level[i]))
                J = jacobian_xyz2uv(x) * sic
                if it == 0:
                    chi2_init.append(float(e @ e))
                w = float(tukey(np.linalg.norm(e) / scale_pt))
                A += J.T @ J * w
                b -= J.T @ e * w
                new_chi2 += float(e @ e) * w
            for s in range(ns):
                if not seg_keep[s]:
                    continue
                ds, de, xs, xe = seg_d(s, Tm)
                ds, de = f32(ds), f32(de)
                sic = 1.0 / (1 << int(seg_level[s]))
                e = np.array([float(ds), float(de)]) * sic
                if it == 0:
                    chi2_init.append(float(e @ e))
                with np.errstate(all="ignore"):
                    k = sic * float(ds) / np.linalg.norm(e)
                Js, Je = jacobian_xyz2uv(xs) * k, jacobian_xyz2uv(xe) * k
                l = seg_line[s, :2]
                J = np.stack([l @ Js, l @ Je])
                w = float(tukey(np.linalg.norm(e) / scale_ls))
                A += J.T @ J * w
                b -= J.T @ e * w
                new_chi2 += float(e @ e) * w
            state["A"] = A
            dT = ldlt_solve(A, b)
            acc = 1
            if (it > 0 and new_chi2 > state["chi2"]) or np.isnan(dT[0]):
                state["T"] = state["T_old"].copy()
                acc = 0
            else:
                Tn = se3_mul(se3_exp(dT), state["T"])
                state["T_old"] = state["T"].copy()
                state["T"] = Tn
                state["chi2"] = new_chi2
            log.append(dict(phase=phase, iter=it, accepted=acc, new_chi2=new_chi2, A=A.copy(), b=b.copy(), dT=dT.copy(), T_after=state["T"].copy()))
            if not acc or np.max(np.abs(dT)) <= 1e-10:
                break
        return iters

    iters = gn(n_iter, 0)
    with np.errstate(all="ignore"):
        cov = np.linalg.inv(state["A"] * fx * fx) if np.linalg.det(state["A"]) != 0 else np.full((6, 6), np.nan)
    thr_pt = reproj_thresh / fx
    thr_ls = thr_pt * scale_ls / scale_pt
    chi2_final, ndp, ndl = [], 0, 0
    Tm = state["T"]
    for i in range(np_):
        e, _ = pt_err(i, Tm)
        chi2_final.append(float(e @ e))
        if np.linalg.norm(e) > thr_pt:
            pt_keep[i] = False
            ndp += 1
    for s in range(ns):
        es, ee, _, _ = seg_d(s, Tm)
        e = np.array([es, ee]) * (1.0 / (1 << int(seg_level[s])))
        chi2_final.append(float(e @ e))
        if np.linalg.norm(e) > thr_ls:
            seg_keep[s] = False
            ndl += 1
    iters_ref = gn(n_iter_ref, 1) if n_iter_ref >= 0 else 0
    return dict(T=state["T"], cov=cov, estimated_scale=scale_pt * fx,
                error_init=math.sqrt(median_upper(chi2_init)) * fx if chi2_init else 0.0,
                error_final=math.sqrt(median_upper(chi2_final)) * fx if chi2_final else 0.0,
                num_obs_pt=np_ - ndp, num_obs_ls=ns - ndl, pt_keep=pt_keep, seg_keep=seg_keep, iters=iters,
                iters_ref=iters_ref, status=0, log=log)
