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

// Eigen::LDLT<Matrix3d> compute + solve (diagonal pivoting), the 3x3 instance of plsvo_math.hpp::ldlt_solve6
__device__ __forceinline__ void ldlt_solve3(const double* A, const double* b, double* x) {
  double m[3][3]; int tr[3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = A[i * 3 + j];
  for (int k = 0; k < 3; ++k) {
    int big = k; double bigv = fabs(m[k][k]);
    for (int i = k + 1; i < 3; ++i) { const double v = fabs(m[i][i]); if (v > bigv) { bigv = v; big = i; } }
    tr[k] = big;
    if (k != big) {
      for (int j = 0; j < k; ++j) { const double t = m[k][j]; m[k][j] = m[big][j]; m[big][j] = t; }
      for (int i = big + 1; i < 3; ++i) { const double t = m[i][k]; m[i][k] = m[i][big]; m[i][big] = t; }
      { const double t = m[k][k]; m[k][k] = m[big][big]; m[big][big] = t; }
      for (int i = k + 1; i < big; ++i) { const double t = m[i][k]; m[i][k] = m[big][i]; m[big][i] = t; }
    }
    if (k > 0) {
      double temp[3];
      for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
      double acc = 0.0;
      for (int j = 0; j < k; ++j) acc += m[k][j] * temp[j];
      m[k][k] -= acc;
      for (int i = k + 1; i < 3; ++i) {
        double a2 = 0.0;
        for (int j = 0; j < k; ++j) a2 += m[i][j] * temp[j];
        m[i][k] -= a2;
      }
    }
    const double akk = m[k][k];
    const bool pivot_is_valid = fabs(akk) > 0.0;
    if (k == 0 && !pivot_is_valid) { for (int j = 0; j < 3; ++j) tr[j] = j; break; }
    if (k < 2 && pivot_is_valid) for (int i = k + 1; i < 3; ++i) m[i][k] /= akk;
  }
  double d[3];
  for (int i = 0; i < 3; ++i) d[i] = b[i];
  for (int k = 0; k < 3; ++k) { const double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  for (int i = 0; i < 3; ++i) for (int j = 0; j < i; ++j) d[i] -= m[i][j] * d[j];
  const double tolerance = 1.0 / 1.7976931348623157e308;
  for (int i = 0; i < 3; ++i) { if (fabs(m[i][i]) > tolerance) d[i] /= m[i][i]; else d[i] = 0.0; }
  for (int i = 2; i >= 0; --i) for (int j = i + 1; j < 3; ++j) d[i] -= m[j][i] * d[j];
  for (int k = 2; k >= 0; --k) { const double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  for (int i = 0; i < 3; ++i) x[i] = d[i];
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
