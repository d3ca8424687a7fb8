// structopt_kernels.hip -- structure optimisation on gfx950: one LANE per 3-D landmark.
//
// Replaces (reference file:line):
//   plsvo::Point::optimize      src/feature3D_impl.cpp:36-95
//   plsvo::LineSeg::optimize    src/feature3D_impl.cpp:97-174
//   Point::jacobian_xyz2uv      include/plsvo/feature3D.h:126-140
//   [ext] Eigen::LDLT<Matrix3d>::solve
// called per frame from FrameHandlerBase::optimizeStructure (src/frame_handler_base.cpp:202-237) on <= 20 points
// and <= 20 segments, 5 iterations each.  Every landmark is an independent 3x3 Gauss-Newton over a handful of
// observations, so there is nothing to reduce across lanes: a batch of landmarks (many frames' worth) is one
// launch, one lane each, observations read through flat offset tables.
//
// This file is compiled with -ffp-contract=off and evaluates every expression in the reference's order, so the
// results are BIT-IDENTICAL to the CPU oracle (the reference's own arithmetic here is unambiguous double math).
#include <hip/hip_runtime.h>

#include "plsvo_dev.hpp"
#include "plsvo_math.hpp"

namespace plsvo_hip {

// A.ldlt().solve(b) for a symmetric 3x3 (src/feature3D_impl.cpp:83,146-147; [ext] Eigen::LDLT, release 3.2's zero-pivot rule
// as in plsvo_wave.hpp::wave_solve6_core), written out on scalars: no arrays, no run-time indexing, nothing in scratch.
// Eigen's unblocked LDLT pivots on the stored diagonal of the tail, which a left-looking factorisation has not updated yet: the
// order is a sort of |A00|, |A11|, |A22| (first maximum wins, in the order its own swaps leave behind), applied as two
// conditional symmetric swaps; the factorisation and the two triangular solves then follow Eigen's operation order term by term
// (this translation unit is compiled without fma contraction), so the result is the CPU oracle's bit for bit.
__device__ __forceinline__ void ldlt_solve3(const double* A, const double* b, double* x) {
  double a00 = A[0], a10 = A[3], a11 = A[4], a20 = A[6], a21 = A[7], a22 = A[8];   // lower triangle
  double y0 = b[0], y1 = b[1], y2 = b[2];
  auto swp = [](double& p, double& q, bool c) { const double t = p; p = c ? q : p; q = c ? t : q; };
  // step 0: largest of the three diagonals to the front
  int big0 = 0; double bigv = fabs(a00);
  if (fabs(a11) > bigv) { bigv = fabs(a11); big0 = 1; }
  if (fabs(a22) > bigv) { bigv = fabs(a22); big0 = 2; }
  const double cutoff = fabs(2.220446049250313e-16 * bigv);
  swp(a00, a11, big0 == 1); swp(a20, a21, big0 == 1);       // rows/columns 0 <-> 1 of the lower triangle: a10 stays
  swp(a00, a22, big0 == 2); swp(a10, a21, big0 == 2);       // rows/columns 0 <-> 2: a20 stays
  swp(y0, y1, big0 == 1); swp(y0, y2, big0 == 2);
  double d0 = a00, l10 = a10, l20 = a20;
  if (fabs(d0) > cutoff) { l10 = a10 / d0; l20 = a20 / d0; }
  // step 1 (skipped, like everything after it, once the largest remaining diagonal is below the cutoff)
  double d1 = a11, l21 = a21, d2 = a22;
  bool big1 = false;
  const bool go1 = !(fmax(fabs(a11), fabs(a22)) < cutoff) || fabs(a11) != fabs(a11);
  if (go1) {
    big1 = fabs(a22) > fabs(a11);
    swp(a11, a22, big1); swp(l10, l20, big1);               // rows/columns 1 <-> 2: a21 stays
    const double t0 = d0 * l10;
    d1 = a11 - l10 * t0;
    l21 = a21 - l20 * t0;
    if (fabs(d1) > cutoff) l21 = l21 / d1;
    // step 2
    d2 = a22;
    if (!(fabs(a22) < cutoff)) {
      const double u0 = d0 * l20, u1 = d1 * l21;
      d2 = a22 - (l20 * u0 + l21 * u1);
    }
  }
  swp(y1, y2, big1);
  // P b -> L^-1 -> D^+ -> L^-T -> P^T
  y1 -= l10 * y0;
  y2 -= l20 * y0;
  y2 -= l21 * y1;
  double maxd = fabs(d0);
  if (fabs(d1) > maxd) maxd = fabs(d1);
  if (fabs(d2) > maxd) maxd = fabs(d2);
  double tolerance = 1.0 / 1.7976931348623157e308;
  if (maxd * 2.220446049250313e-16 > tolerance) tolerance = maxd * 2.220446049250313e-16;
  y0 = (fabs(d0) > tolerance) ? y0 / d0 : 0.0;
  y1 = (fabs(d1) > tolerance) ? y1 / d1 : 0.0;
  y2 = (fabs(d2) > tolerance) ? y2 / d2 : 0.0;
  y1 -= l21 * y2;
  y0 -= l10 * y1;
  y0 -= l20 * y2;
  swp(y1, y2, big1);
  swp(y0, y2, big0 == 2); swp(y0, y1, big0 == 1);
  x[0] = y0; x[1] = y1; x[2] = y2;
}

// one observation: Point::jacobian_xyz2uv, e = project2d(f) - project2d(p_in_f), A += J^T J, b -= J^T e, chi2 += |e|^2
__device__ __forceinline__ void accumulate_obs(const double* Tf, const double* f, const double* pos, double* A, double* b, double* chi2) {
  const SE3d T = se3_load(Tf);
  double R[9], p[3];
  quat_to_matrix(T.q, R);
  se3_act(T, pos, p);
  const double z_inv = 1.0 / p[2];
  const double z_inv_sq = z_inv * z_inv;
  const double P[6] = { z_inv, 0.0, -p[0] * z_inv_sq, 0.0, z_inv, -p[1] * z_inv_sq };
  double J[6];
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 3; ++c)
      J[r * 3 + c] = (-P[r * 3 + 0]) * R[0 * 3 + c] + (-P[r * 3 + 1]) * R[1 * 3 + c] + (-P[r * 3 + 2]) * R[2 * 3 + c];
  const double e0 = f[0] / f[2] - p[0] / p[2], e1 = f[1] / f[2] - p[1] / p[2];
  *chi2 += e0 * e0 + e1 * e1;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) A[i * 3 + j] += J[i] * J[j] + J[3 + i] * J[3 + j];
    b[i] -= J[i] * e0 + J[3 + i] * e1;
  }
}

__device__ __forceinline__ double norm_max3(const double* v) {
  double m = 0; for (int i = 0; i < 3; ++i) { const double a = fabs(v[i]); if (a > m) m = a; } return m;
}

__global__ __launch_bounds__(64) void structopt_kernel(StructBatchDev s) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < s.n_pts) {                                     // Point::optimize :36-95
    const int i = idx;
    double pos[3] = { s.pt_pos[3 * i], s.pt_pos[3 * i + 1], s.pt_pos[3 * i + 2] };
    double old_point[3] = { pos[0], pos[1], pos[2] };
    double chi2 = 0.0; int iters = 0;
    const int o0 = s.pt_obs_off[i], o1 = s.pt_obs_off[i + 1];
    for (int it = 0; it < s.n_iter_pts; ++it) {
      double A[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, b[3] = { 0, 0, 0 }, new_chi2 = 0.0;
      ++iters;
      for (int o = o0; o < o1; ++o) accumulate_obs(s.frame_T + 7 * s.pt_obs_frame[o], s.pt_obs_f + 3 * o, pos, A, b, &new_chi2);
      double dp[3];
      ldlt_solve3(A, b, dp);
      if ((it > 0 && new_chi2 > chi2) || isnan(dp[0])) { pos[0] = old_point[0]; pos[1] = old_point[1]; pos[2] = old_point[2]; break; }
      for (int k = 0; k < 3; ++k) { old_point[k] = pos[k]; pos[k] = pos[k] + dp[k]; }
      chi2 = new_chi2;
      if (norm_max3(dp) <= 0.0000000001) break;            // EPS, global.h:99
    }
    s.pt_pos_out[3 * i] = pos[0]; s.pt_pos_out[3 * i + 1] = pos[1]; s.pt_pos_out[3 * i + 2] = pos[2];
    s.pt_iters[i] = iters;
  } else if (idx < s.n_pts + s.n_seg) {                    // LineSeg::optimize :97-174
    const int i = idx - s.n_pts;
    double sp[3], ep[3], old_s[3], old_e[3];
    for (int k = 0; k < 3; ++k) { sp[k] = old_s[k] = s.seg_spos[3 * i + k]; ep[k] = old_e[k] = s.seg_epos[3 * i + k]; }
    double chi2s = 0.0, chi2e = 0.0; int iters = 0;
    const int o0 = s.seg_obs_off[i], o1 = s.seg_obs_off[i + 1];
    for (int it = 0; it < s.n_iter_segs; ++it) {
      double As[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, Ae[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, bs[3] = { 0, 0, 0 }, be[3] = { 0, 0, 0 };
      double ncs = 0.0, nce = 0.0;
      ++iters;
      for (int o = o0; o < o1; ++o) {
        const double* Tf = s.frame_T + 7 * s.seg_obs_frame[o];
        accumulate_obs(Tf, s.seg_obs_sf + 3 * o, sp, As, bs, &ncs);
        accumulate_obs(Tf, s.seg_obs_ef + 3 * o, ep, Ae, be, &nce);
      }
      double dps[3], dpe[3];
      ldlt_solve3(As, bs, dps);
      ldlt_solve3(Ae, be, dpe);
      if ((it > 0 && ncs > chi2s) || isnan(dps[0]) || (it > 0 && nce > chi2e) || isnan(dpe[0])) {
        for (int k = 0; k < 3; ++k) { sp[k] = old_s[k]; ep[k] = old_e[k]; }
        break;
      }
      for (int k = 0; k < 3; ++k) { old_s[k] = sp[k]; sp[k] = sp[k] + dps[k]; old_e[k] = ep[k]; ep[k] = ep[k] + dpe[k]; }
      chi2s = ncs; chi2e = nce;
      if (norm_max3(dps) <= 0.0000000001 || norm_max3(dpe) <= 0.0000000001) break;
    }
    for (int k = 0; k < 3; ++k) { s.seg_spos_out[3 * i + k] = sp[k]; s.seg_epos_out[3 * i + k] = ep[k]; }
    s.seg_iters[i] = iters;
  }
}

hipError_t launch_structopt(const StructBatchDev& s, hipStream_t stream) {
  const int n = s.n_pts + s.n_seg;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(structopt_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, s);
  return hipGetLastError();
}

}  // namespace plsvo_hip
