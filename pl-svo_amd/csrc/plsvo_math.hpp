// plsvo_math.hpp -- small fixed-size SE(3) / 6x6 algebra shared by the HIP kernels and the host adapter.
//
// Everything the hot path needs from Sophus / Eigen / vikit is a handful of 3- and 6-dimensional
// operations; they are written out here (double precision) so the device code has no dependency.
// Conventions follow the reference's libraries so results agree to rounding:
//   pose        = unit quaternion (x,y,z,w) + translation          (Sophus::SE3 storage)
//   tangent     = (upsilon[0:3], omega[3:6])                       (Sophus::SE3::exp)
//   compose     = t_A + R_A t_B, normalize(q_A q_B)                (Sophus::SE3::operator*)
//   6x6 solve   = LDL^T with diagonal pivoting                     (Eigen::LDLT, src/sparse_img_align.cpp:699 and
//                 src/pose_optimizer.cpp:170): on the device, plsvo_wave.hpp::wave_solve6_core
//   6x6 inverse = LU with partial pivoting                         (Eigen inverse(), src/pose_optimizer.cpp:199)
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PLSVO_HD __host__ __device__ __forceinline__
#else
#define PLSVO_HD inline
#endif

namespace plsvo_hip {

struct Quat { double x, y, z, w; };
struct SE3d { Quat q; double t[3]; };

PLSVO_HD SE3d se3_load(const double* T) {
  SE3d r; r.q.x = T[0]; r.q.y = T[1]; r.q.z = T[2]; r.q.w = T[3]; r.t[0] = T[4]; r.t[1] = T[5]; r.t[2] = T[6];
  return r;
}
PLSVO_HD void se3_store(const SE3d& s, double* T) {
  T[0] = s.q.x; T[1] = s.q.y; T[2] = s.q.z; T[3] = s.q.w; T[4] = s.t[0]; T[5] = s.t[1]; T[6] = s.t[2];
}
PLSVO_HD Quat quat_mul(const Quat& a, const Quat& b) {
  Quat r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
PLSVO_HD Quat quat_normalized(const Quat& a) {
  const double n = sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
  Quat r = { a.x / n, a.y / n, a.z / n, a.w / n };
  return r;
}
// row-major 3x3 (Eigen::Quaternion::toRotationMatrix)
PLSVO_HD void quat_to_matrix(const Quat& q, double* R) {
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}
// Eigen::Quaternion::_transformVector
PLSVO_HD void quat_rotate(const Quat& q, const double* v, double* out) {
  double ux = q.y * v[2] - q.z * v[1], uy = q.z * v[0] - q.x * v[2], uz = q.x * v[1] - q.y * v[0];
  ux += ux; uy += uy; uz += uz;
  out[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
  out[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
  out[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}
PLSVO_HD SE3d se3_mul(const SE3d& A, const SE3d& B) {
  SE3d r; double rt[3];
  quat_rotate(A.q, B.t, rt);
  r.t[0] = A.t[0] + rt[0]; r.t[1] = A.t[1] + rt[1]; r.t[2] = A.t[2] + rt[2];
  r.q = quat_normalized(quat_mul(A.q, B.q));
  return r;
}
PLSVO_HD SE3d se3_inv(const SE3d& A) {
  SE3d r; const double nt[3] = { -A.t[0], -A.t[1], -A.t[2] };
  r.q.x = -A.q.x; r.q.y = -A.q.y; r.q.z = -A.q.z; r.q.w = A.q.w;
  quat_rotate(r.q, nt, r.t);
  return r;
}
PLSVO_HD void se3_act(const SE3d& T, const double* p, double* out) {
  quat_rotate(T.q, p, out);
  out[0] += T.t[0]; out[1] += T.t[1]; out[2] += T.t[2];
}
// Sophus::SE3::exp (SO3::expAndTheta + the V matrix); SMALL_EPS = 1e-10
PLSVO_HD SE3d se3_exp(const double* u) {
  SE3d r;
  const double ox = u[3], oy = u[4], oz = u[5];
  const double theta = sqrt(ox * ox + oy * oy + oz * oz);
  const double half_theta = 0.5 * theta;
  double imag_factor;
  const double real_factor = cos(half_theta);
  if (theta < 1e-10) {
    const double theta_sq = theta * theta;
    const double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
  } else {
    imag_factor = sin(half_theta) / theta;
  }
  Quat q = { imag_factor * ox, imag_factor * oy, imag_factor * oz, real_factor };
  r.q = quat_normalized(q);
  double V[9];
  if (theta < 1e-10) {
    quat_to_matrix(r.q, V);
  } else {
    const double theta_sq = theta * theta;
    const double a = (1 - cos(theta)) / theta_sq;
    const double b = (theta - sin(theta)) / (theta_sq * theta);
    // Omega = hat(omega), Omega^2 = omega omega^T - theta^2 I (written out entry by entry)
    const double O2_00 = -(oy * oy + oz * oz), O2_11 = -(ox * ox + oz * oz), O2_22 = -(ox * ox + oy * oy);
    const double O2_01 = ox * oy, O2_02 = ox * oz, O2_12 = oy * oz;
    V[0] = 1.0 + b * O2_00;      V[1] = a * -oz + b * O2_01;  V[2] = a * oy + b * O2_02;
    V[3] = a * oz + b * O2_01;   V[4] = 1.0 + b * O2_11;      V[5] = a * -ox + b * O2_12;
    V[6] = a * -oy + b * O2_02;  V[7] = a * ox + b * O2_12;   V[8] = 1.0 + b * O2_22;
  }
  for (int i = 0; i < 3; ++i) r.t[i] = V[i * 3 + 0] * u[0] + V[i * 3 + 1] * u[1] + V[i * 3 + 2] * u[2];
  return r;
}

// Frame::jacobian_xyz2uv (include/plsvo/frame.h:138-160), row-major 2x6
PLSVO_HD void jacobian_xyz2uv(const double* xyz, double* J) {
  const double x = xyz[0], y = xyz[1];
  const double z_inv = 1. / xyz[2];
  const double z_inv_2 = z_inv * z_inv;
  J[0] = -z_inv; J[1] = 0.0; J[2] = x * z_inv_2; J[3] = y * J[2]; J[4] = -(1.0 + x * J[2]); J[5] = y * z_inv;
  J[6] = 0.0; J[7] = -z_inv; J[8] = y * z_inv_2; J[9] = 1.0 + y * J[8]; J[10] = -J[3]; J[11] = -x * z_inv;
}

// 6x6 inverse through LU with partial pivoting (Eigen PartialPivLU::inverse)
PLSVO_HD void inv6(const double* A, double* Ainv) {
  double lu[6][6]; int perm[6];
  for (int i = 0; i < 6; ++i) { perm[i] = i; for (int j = 0; j < 6; ++j) lu[i][j] = A[i * 6 + j]; }
  for (int k = 0; k < 6; ++k) {
    int piv = k; double pv = fabs(lu[k][k]);
    for (int i = k + 1; i < 6; ++i) { const double v = fabs(lu[i][k]); if (v > pv) { pv = v; piv = i; } }
    if (piv != k) {
      for (int j = 0; j < 6; ++j) { const double t = lu[k][j]; lu[k][j] = lu[piv][j]; lu[piv][j] = t; }
      const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    for (int i = k + 1; i < 6; ++i) {
      lu[i][k] /= lu[k][k];
      for (int j = k + 1; j < 6; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
    }
  }
  for (int c = 0; c < 6; ++c) {
    double y[6];
    for (int i = 0; i < 6; ++i) y[i] = (perm[i] == c) ? 1.0 : 0.0;
    for (int i = 0; i < 6; ++i) for (int j = 0; j < i; ++j) y[i] -= lu[i][j] * y[j];
    for (int i = 5; i >= 0; --i) { for (int j = i + 1; j < 6; ++j) y[i] -= lu[i][j] * y[j]; y[i] /= lu[i][i]; }
    for (int i = 0; i < 6; ++i) Ainv[i * 6 + c] = y[i];
  }
}

#if defined(__HIPCC__)
// inv6 on the device with the LU factors in LDS (the pivot rows are indexed dynamically, which in private memory means
// scratch): one lane factorises in place (lu6_lds), then six lanes solve for one column of the inverse each
// (inv6_column_lds).  Same arithmetic and pivot order as inv6.
__device__ __forceinline__ void lu6_lds(double* lu /*36, LDS, in: A, out: LU*/, int* perm /*6, LDS*/) {
  for (int i = 0; i < 6; ++i) perm[i] = i;
  for (int k = 0; k < 6; ++k) {
    int piv = k; double pv = fabs(lu[k * 6 + k]);
    for (int i = k + 1; i < 6; ++i) { const double v = fabs(lu[i * 6 + k]); if (v > pv) { pv = v; piv = i; } }
    if (piv != k) {
      for (int j = 0; j < 6; ++j) { const double t = lu[k * 6 + j]; lu[k * 6 + j] = lu[piv * 6 + j]; lu[piv * 6 + j] = t; }
      const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    for (int i = k + 1; i < 6; ++i) {
      lu[i * 6 + k] /= lu[k * 6 + k];
      for (int j = k + 1; j < 6; ++j) lu[i * 6 + j] -= lu[i * 6 + k] * lu[k * 6 + j];
    }
  }
}
__device__ __forceinline__ void inv6_column_lds(const double* lu, const int* perm, int c, double* Ainv) {
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) y[i] = (perm[i] == c) ? 1.0 : 0.0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j < i; ++j) y[i] -= lu[i * 6 + j] * y[j];
#pragma unroll
  for (int i = 5; i >= 0; --i) {
#pragma unroll
    for (int j = i + 1; j < 6; ++j) y[i] -= lu[i * 6 + j] * y[j];
    y[i] /= lu[i * 6 + i];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) Ainv[i * 6 + c] = y[i];
}
#endif

PLSVO_HD double norm_max6(const double* v) {
  double m = 0; for (int i = 0; i < 6; ++i) { const double a = fabs(v[i]); if (a > m) m = a; } return m;
}

// vk::robust_cost::TukeyWeightFunction::value (b = 4.6851f), float in / float out
PLSVO_HD float tukey_weight(float x) {
  const float b = 4.6851f; const float b_square = b * b;
  const float x_square = x * x;
  if (x_square <= b_square) { const float tmp = 1.0f - x_square / b_square; return tmp * tmp; }
  return 0.0f;
}

// LineFeat::setupSampling (src/feature.cpp:160-173) followed by the per-level reduction
// N = 1 + (N-1) / 2^level (src/sparse_img_align.cpp:320, 565)
// (no fma contraction: host staging, device kernel and the CPU oracle must agree on N bit for bit)
PLSVO_HD int seg_num_samples(double sx, double sy, double ex, double ey, double length, int level) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const double d0 = fabs(ex - sx), d1 = fabs(ey - sy);
  const double tan_dir = (d0 < d1 ? d0 : d1) / (d0 > d1 ? d0 : d1);
  const double sin_dir = tan_dir / sqrt(1.0 + tan_dir * tan_dir);
  const double correction = 2.0 * sqrt(1.0 + sin_dir * sin_dir);
  const double v = length / (2.0 * 4 * correction);
  const long long n0 = (long long)((1.0 < v) ? v : 1.0);  // std::max(1.0, v); NaN -> 1
  return (int)(1 + (n0 - 1) / (1 << level));
}

}  // namespace plsvo_hip
