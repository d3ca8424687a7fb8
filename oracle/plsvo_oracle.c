#ifndef _GNU_SOURCE
#define _GNU_SOURCE   /* sched_getaffinity / pthread_setaffinity_np for the CPU-baseline harness at the end of this file */
#endif
/*
 * plsvo_oracle.c -- CPU restatement of PL-SVO's sparse image alignment + pose optimisation.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see plsvo_oracle.h for what that means and why).
 *
 * Typing follows the reference exactly: image interpolation, residuals and robust weights in
 * float; geometry, Jacobians and all accumulators in double.  Build with -ffp-contract=off so the
 * float expressions round like written C (the reference's build flags, CMakeLists.txt:25-36, do not
 * pin contraction either way; see DESIGN.md "numerics").
 *
 * Every function cites the reference file:line it follows; "[ext]" marks semantics of a third-party
 * dependency that is NOT under /root/reference and is restated from its published source.
 */
#include "plsvo_oracle.h"

#include <float.h>
#include <stdatomic.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ============================================================================================ */
/* [ext] Sophus (non-templated) SO3 / SE3, storage: unit quaternion (x,y,z,w) + translation      */
/* ============================================================================================ */

#define SMALL_EPS 1e-10 /* [ext] sophus/so3.h */

typedef struct { double x, y, z, w; } quat_t;
typedef struct { quat_t q; double t[3]; } se3_t;

static se3_t se3_load(const double T[7]) {
  se3_t r; r.q.x = T[0]; r.q.y = T[1]; r.q.z = T[2]; r.q.w = T[3];
  r.t[0] = T[4]; r.t[1] = T[5]; r.t[2] = T[6]; return r;
}
static void se3_store(const se3_t* s, double T[7]) {
  T[0] = s->q.x; T[1] = s->q.y; T[2] = s->q.z; T[3] = s->q.w;
  T[4] = s->t[0]; T[5] = s->t[1]; T[6] = s->t[2];
}

/* [ext] Eigen::Quaternion product */
static quat_t quat_mul(quat_t a, quat_t b) {
  quat_t r;
  r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
  r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
  return r;
}
/* [ext] Eigen::Quaternion::normalize: coeffs /= sqrt(squaredNorm) */
static quat_t quat_normalized(quat_t a) {
  const double n = sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w);
  quat_t r = { a.x / n, a.y / n, a.z / n, a.w / n };
  return r;
}
/* [ext] Eigen::Quaternion::_transformVector:  uv = 2 * vec x v;  v + w*uv + vec x uv
 * (SO3::operator*(Vector3d) in Sophus calls this) */
static void quat_rotate(quat_t q, const double v[3], double out[3]) {
  double uv[3];
  uv[0] = q.y * v[2] - q.z * v[1];
  uv[1] = q.z * v[0] - q.x * v[2];
  uv[2] = q.x * v[1] - q.y * v[0];
  uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
  out[0] = v[0] + q.w * uv[0] + (q.y * uv[2] - q.z * uv[1]);
  out[1] = v[1] + q.w * uv[1] + (q.z * uv[0] - q.x * uv[2]);
  out[2] = v[2] + q.w * uv[2] + (q.x * uv[1] - q.y * uv[0]);
}
/* [ext] Eigen::Quaternion::toRotationMatrix, row-major */
static void quat_to_matrix(quat_t q, double R[9]) {
  const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
  R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

/* [ext] Sophus::SE3::operator*: t = t_A + R_A t_B; q = normalize(q_A q_B) */
static se3_t se3_mul(const se3_t* A, const se3_t* B) {
  se3_t r; double rt[3];
  quat_rotate(A->q, B->t, rt);
  r.t[0] = A->t[0] + rt[0]; r.t[1] = A->t[1] + rt[1]; r.t[2] = A->t[2] + rt[2];
  r.q = quat_normalized(quat_mul(A->q, B->q));
  return r;
}
/* [ext] Sophus::SE3::inverse: (q^-1, q^-1 * (-t)) */
static se3_t se3_inv(const se3_t* A) {
  se3_t r; double nt[3] = { A->t[0] * -1., A->t[1] * -1., A->t[2] * -1. };
  r.q.x = -A->q.x; r.q.y = -A->q.y; r.q.z = -A->q.z; r.q.w = A->q.w;
  quat_rotate(r.q, nt, r.t);
  return r;
}
static void se3_act(const se3_t* T, const double p[3], double out[3]) {
  quat_rotate(T->q, p, out);
  out[0] += T->t[0]; out[1] += T->t[1]; out[2] += T->t[2];
}
/* [ext] Sophus::SE3::exp(Vector6d): tangent = (upsilon[0:3], omega[3:6]); SO3::expAndTheta */
static se3_t se3_exp(const double u[6]) {
  se3_t r;
  const double ox = u[3], oy = u[4], oz = u[5];
  const double theta = sqrt(ox * ox + oy * oy + oz * oz);
  const double half_theta = 0.5 * theta;
  double imag_factor;
  const double real_factor = cos(half_theta);
  if (theta < SMALL_EPS) {
    const double theta_sq = theta * theta;
    const double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
  } else {
    const double sin_half_theta = sin(half_theta);
    imag_factor = sin_half_theta / theta;
  }
  quat_t q = { imag_factor * ox, imag_factor * oy, imag_factor * oz, real_factor };
  r.q = quat_normalized(q); /* SO3(Quaterniond) ctor normalises */
  /* V = I + (1-cos t)/t^2 Omega + (t - sin t)/t^3 Omega^2, or the rotation matrix when t is tiny */
  double V[9];
  if (theta < SMALL_EPS) {
    quat_to_matrix(r.q, V);
  } else {
    const double O[9] = { 0, -oz, oy, oz, 0, -ox, -oy, ox, 0 };
    double O2[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        O2[i * 3 + j] = O[i * 3 + 0] * O[0 * 3 + j] + O[i * 3 + 1] * O[1 * 3 + j] + O[i * 3 + 2] * O[2 * 3 + j];
    const double theta_sq = theta * theta;
    const double a = (1 - cos(theta)) / (theta_sq);
    const double b = (theta - sin(theta)) / (theta_sq * theta);
    for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * O[i] + b * O2[i];
  }
  for (int i = 0; i < 3; ++i) r.t[i] = V[i * 3 + 0] * u[0] + V[i * 3 + 1] * u[1] + V[i * 3 + 2] * u[2];
  return r;
}

void plsvo_oracle_se3_exp(const double u[6], double T[7]) { se3_t r = se3_exp(u); se3_store(&r, T); }
void plsvo_oracle_se3_mul(const double A[7], const double B[7], double C[7]) {
  se3_t a = se3_load(A), b = se3_load(B), c = se3_mul(&a, &b); se3_store(&c, C);
}
void plsvo_oracle_se3_inv(const double A[7], double B[7]) { se3_t a = se3_load(A), b = se3_inv(&a); se3_store(&b, B); }
void plsvo_oracle_se3_act(const double T[7], const double p[3], double out[3]) { se3_t t = se3_load(T); se3_act(&t, p, out); }
void plsvo_oracle_se3_matrix(const double T[7], double R[9], double t[3]) {
  se3_t s = se3_load(T); quat_to_matrix(s.q, R); t[0] = s.t[0]; t[1] = s.t[1]; t[2] = s.t[2];
}

/* ============================================================================================ */
/* [ext] Eigen 3 dense pieces: LDLT with diagonal pivoting (ldlt().solve) and PartialPivLU inverse */
/* ============================================================================================ */

/* Eigen::LDLT<MatrixNd>::compute (unblocked, lower, N <= 6) followed by solve():
 *   dst = P b;  L^-1;  D^+ (pseudo-inverse of D);  L^-T;  P^T.
 * The pivot of step k is the largest |diagonal| of the tail -- of the matrix AS STORED: the algorithm is left-looking, only
 * column k is updated in step k, so the tail diagonal still holds the (permuted) ORIGINAL entries.  First maximum wins.
 *
 * What counts as a zero pivot depends on the Eigen release, and PL-SVO names three platforms (README.md:27: Ubuntu 12.04,
 * 14.04, 16.04 = libeigen3-dev 3.0.5, 3.2.0, 3.2.92) without pinning one (CMakeLists.txt:40).  Two flavours are restated:
 *   320 (default; Eigen 3.1 ... 3.2.1, Ubuntu 14.04): cutoff = |eps * largest diagonal| fixed at step 0; the factorisation
 *       stops when the largest remaining (stored) diagonal is below it; column k is divided by its pivot only if
 *       |pivot| > cutoff; solve() zeroes the components whose |D| <= max(max|D| * eps, 1/highest()).
 *   330 (Eigen 3.3): a pivot is invalid only if it is exactly 0; solve() zeroes components with |D| <= 1/highest().
 * They are the same arithmetic, bit for bit, whenever every pivot exceeds eps * the largest diagonal -- every full-rank system
 * this path produces (tests/test_oracle_unit.py checks that, and the committed fixtures do not depend on the flavour).  They
 * differ on rank-deficient normal equations (fewer than three point observations), where 330 divides rounding residue by
 * rounding residue and 320 returns zero components for the unobservable directions.
 * NaN/Inf propagate into x. */
static _Atomic int g_ldlt_flavour = 320;   /* read by the threads of plsvo_oracle_bench while a test may set it: atomic */
void plsvo_oracle_set_ldlt_flavour(int flavour) { atomic_store(&g_ldlt_flavour, (flavour == 330) ? 330 : 320); }
int plsvo_oracle_get_ldlt_flavour(void) { return atomic_load(&g_ldlt_flavour); }

static void ldlt_solve_n(const int N, const double* A, const double* b, double* x) {
  double m[6][6]; int tr[6];
  const int flavour = atomic_load(&g_ldlt_flavour);
  double cutoff = 0.0;
  for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) m[i][j] = A[i * N + j];
  for (int k = 0; k < N; ++k) {
    /* largest |diagonal| in the remaining corner (first maximum wins) */
    int big = k; double bigv = fabs(m[k][k]);
    for (int i = k + 1; i < N; ++i) { const double v = fabs(m[i][i]); if (v > bigv) { bigv = v; big = i; } }
    if (flavour == 320) {
      if (k == 0) cutoff = fabs(DBL_EPSILON * bigv);
      if (bigv < cutoff) { for (int i = k; i < N; ++i) tr[i] = i; break; }   /* "finish early if the matrix is not full rank" */
    }
    tr[k] = big;
    if (k != big) {
      const int s = N - big - 1;
      for (int j = 0; j < k; ++j) { const double t = m[k][j]; m[k][j] = m[big][j]; m[big][j] = t; }
      for (int i = 0; i < s; ++i) { const double t = m[big + 1 + i][k]; m[big + 1 + i][k] = m[big + 1 + i][big]; m[big + 1 + i][big] = t; }
      { const double t = m[k][k]; m[k][k] = m[big][big]; m[big][big] = t; }
      for (int i = k + 1; i < big; ++i) { const double t = m[i][k]; m[i][k] = m[big][i]; m[big][i] = t; }
    }
    const int rs = N - k - 1;
    if (k > 0) {
      double temp[6];
      for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
      double acc = 0.0;
      for (int j = 0; j < k; ++j) acc += m[k][j] * temp[j];
      m[k][k] -= acc;
      for (int i = k + 1; i < N; ++i) {
        double a2 = 0.0;
        for (int j = 0; j < k; ++j) a2 += m[i][j] * temp[j];
        m[i][k] -= a2;
      }
    }
    const double akk = m[k][k];
    if (flavour == 320) {
      if (rs > 0 && fabs(akk) > cutoff) for (int i = k + 1; i < N; ++i) m[i][k] /= akk;
    } else {
      const int pivot_is_valid = fabs(akk) > 0.0;
      if (k == 0 && !pivot_is_valid) { /* the whole diagonal is zero: nothing more to do */
        for (int j = 0; j < N; ++j) tr[j] = j;
        break;
      }
      if (rs > 0 && pivot_is_valid) for (int i = k + 1; i < N; ++i) m[i][k] /= akk;
    }
  }
  double d[6];
  for (int i = 0; i < N; ++i) d[i] = b[i];
  for (int k = 0; k < N; ++k) { const double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  for (int i = 0; i < N; ++i) for (int j = 0; j < i; ++j) d[i] -= m[i][j] * d[j];
  double tolerance = 1.0 / 1.7976931348623157e308;
  if (flavour == 320) {
    double maxd = fabs(m[0][0]);
    for (int i = 1; i < N; ++i) { const double v = fabs(m[i][i]); if (v > maxd) maxd = v; }
    const double rel = maxd * DBL_EPSILON;
    if (rel > tolerance) tolerance = rel;   /* (max)(maxAbsD * eps, 1/highest) */
  }
  for (int i = 0; i < N; ++i) { if (fabs(m[i][i]) > tolerance) d[i] /= m[i][i]; else d[i] = 0.0; }
  for (int i = N - 1; i >= 0; --i) for (int j = i + 1; j < N; ++j) d[i] -= m[j][i] * d[j];
  for (int k = N - 1; k >= 0; --k) { const double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  for (int i = 0; i < N; ++i) x[i] = d[i];
}

/* returns 0 always (Eigen reports nothing here) */
int plsvo_oracle_ldlt_solve6(const double H[36], const double b[6], double x[6]) {
  ldlt_solve_n(6, H, b, x);
  return 0;
}

/* Eigen's inverse() of a fixed 6x6 goes through PartialPivLU: LU with row pivoting, solve for I */
void plsvo_oracle_inv6(const double A[36], double Ainv[36]) {
  enum { N = 6 };
  double lu[N][N]; int perm[N];
  for (int i = 0; i < N; ++i) { perm[i] = i; for (int j = 0; j < N; ++j) lu[i][j] = A[i * N + j]; }
  for (int k = 0; k < N; ++k) {
    int piv = k; double pv = fabs(lu[k][k]);
    for (int i = k + 1; i < N; ++i) { const double v = fabs(lu[i][k]); if (v > pv) { pv = v; piv = i; } }
    if (piv != k) {
      for (int j = 0; j < N; ++j) { const double t = lu[k][j]; lu[k][j] = lu[piv][j]; lu[piv][j] = t; }
      const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    for (int i = k + 1; i < N; ++i) {
      lu[i][k] /= lu[k][k];
      for (int j = k + 1; j < N; ++j) lu[i][j] -= lu[i][k] * lu[k][j];
    }
  }
  for (int c = 0; c < N; ++c) {
    double y[N];
    for (int i = 0; i < N; ++i) y[i] = (perm[i] == c) ? 1.0 : 0.0;
    for (int i = 0; i < N; ++i) for (int j = 0; j < i; ++j) y[i] -= lu[i][j] * y[j];
    for (int i = N - 1; i >= 0; --i) { for (int j = i + 1; j < N; ++j) y[i] -= lu[i][j] * y[j]; y[i] /= lu[i][i]; }
    for (int i = 0; i < N; ++i) Ainv[i * N + c] = y[i];
  }
}

/* ============================================================================================ */
/* [ext] vikit_common: pinhole camera, math_utils, robust_cost, halfSample                        */
/* ============================================================================================ */

/* vk::PinholeCamera::world2cam (no distortion): project2d, then fx*u+cx */
void plsvo_oracle_world2cam(const plsvo_pinhole* cam, const double xyz[3], double px[2]) {
  const double u = xyz[0] / xyz[2], v = xyz[1] / xyz[2];
  px[0] = cam->fx * u + cam->cx;
  px[1] = cam->fy * v + cam->cy;
}
/* vk::PinholeCamera::cam2world (no distortion): normalised ((u-cx)/fx, (v-cy)/fy, 1) */
void plsvo_oracle_cam2world(const plsvo_pinhole* cam, const double px[2], double f[3]) {
  double x = (px[0] - cam->cx) / cam->fx, y = (px[1] - cam->cy) / cam->fy, z = 1.0;
  const double n = sqrt(x * x + y * y + z * z);
  f[0] = x / n; f[1] = y / n; f[2] = z / n;
}
/* vk::AbstractCamera::isInFrame(Vector2i obs, int boundary, int level) */
static int cam_is_in_frame(const plsvo_pinhole* cam, int ox, int oy, int boundary, int level) {
  return ox >= boundary && ox < cam->width / (1 << level) - boundary &&
         oy >= boundary && oy < cam->height / (1 << level) - boundary;
}

static int cmp_f32(const void* a, const void* b) { const float x = *(const float*)a, y = *(const float*)b; return (x > y) - (x < y); }
static int cmp_f64(const void* a, const void* b) { const double x = *(const double*)a, y = *(const double*)b; return (x > y) - (x < y); }
/* vk::getMedian: nth_element at floor(n/2) -- value-equivalent to sorting and indexing */
static float median_f32(float* v, uint64_t n) { qsort(v, n, sizeof(float), cmp_f32); return v[n / 2]; }
double plsvo_oracle_median_f64(double* v, uint64_t n) { qsort(v, n, sizeof(double), cmp_f64); return v[n / 2]; }
/* vk::robust_cost::MADScaleEstimator::compute = 1.48f * median */
float plsvo_oracle_mad_scale(float* errors, uint64_t n) { return 1.48f * median_f32(errors, n); }
/* vk::robust_cost::TukeyWeightFunction::value, b = 4.6851f */
float plsvo_oracle_tukey(float x) {
  const float b = 4.6851f; const float b_square = b * b;
  const float x_square = x * x;
  if (x_square <= b_square) { const float tmp = 1.0f - x_square / b_square; return tmp * tmp; }
  return 0.0f;
}
static double norm_max6(const double v[6]) { double m = 0; for (int i = 0; i < 6; ++i) { const double a = fabs(v[i]); if (a > m) m = a; } return m; }

void plsvo_oracle_halfsample(const uint8_t* in, int w, int h, int stride, uint8_t* out, int out_stride, int rounding) {
  const int ow = w / 2, oh = h / 2;
  for (int y = 0; y < oh; ++y) {
    const uint8_t* r0 = in + (size_t)(2 * y) * stride; const uint8_t* r1 = r0 + stride;
    for (int x = 0; x < ow; ++x) {
      const int a = r0[2 * x], b = r0[2 * x + 1], c = r1[2 * x], d = r1[2 * x + 1];
      if (rounding == 0) { /* SSE2 path: _mm_avg_epu8 of the rows, then _mm_avg_epu16 of neighbours */
        const int ac = (a + c + 1) >> 1, bd = (b + d + 1) >> 1;
        out[(size_t)y * out_stride + x] = (uint8_t)((ac + bd + 1) >> 1);
      } else {
        out[(size_t)y * out_stride + x] = (uint8_t)((a + b + c + d) / 4);
      }
    }
  }
}

/* ============================================================================================ */
/* reference-owned helpers                                                                       */
/* ============================================================================================ */

/* Frame::jacobian_xyz2uv  include/plsvo/frame.h:138-160 (row-major 2x6) */
void plsvo_oracle_jacobian_xyz2uv(const double xyz[3], double J[12]) {
  const double x = xyz[0], y = xyz[1];
  const double z_inv = 1. / xyz[2];
  const double z_inv_2 = z_inv * z_inv;
  J[0] = -z_inv; J[1] = 0.0; J[2] = x * z_inv_2; J[3] = y * J[2]; J[4] = -(1.0 + x * J[2]); J[5] = y * z_inv;
  J[6] = 0.0; J[7] = -z_inv; J[8] = y * z_inv_2; J[9] = 1.0 + y * J[8]; J[10] = -J[3]; J[11] = -x * z_inv;
}

/* LineFeat::setupSampling  src/feature.cpp:160-173 */
uint64_t plsvo_oracle_setup_sampling(const double spx[2], const double epx[2], double length, uint64_t patch_size, double dif[2]) {
  dif[0] = epx[0] - spx[0]; dif[1] = epx[1] - spx[1];
  const double a0 = fabs(dif[0]), a1 = fabs(dif[1]);
  const double tan_dir = (a0 < a1 ? a0 : a1) / (a0 > a1 ? a0 : a1);
  const double sin_dir = tan_dir / sqrt(1.0 + tan_dir * tan_dir);
  const double correction = 2.0 * sqrt(1.0 + sin_dir * sin_dir);
  const double v = length / (2.0 * patch_size * correction);
  return (uint64_t)((1.0 < v) ? v : 1.0); /* std::max(1.0, v) = (1.0 < v) ? v : 1.0; NaN (zero-length segment) gives 1.0 */
}

/* LineFeat ctor  src/feature.cpp:103-104: line = sf x ef, scaled so (l0,l1) is unit */
void plsvo_oracle_line_normal(const double sf[3], const double ef[3], double line[3]) {
  double l[3] = { sf[1] * ef[2] - sf[2] * ef[1], sf[2] * ef[0] - sf[0] * ef[2], sf[0] * ef[1] - sf[1] * ef[0] };
  const double n = sqrt(l[0] * l[0] + l[1] * l[1]);
  line[0] = l[0] / n; line[1] = l[1] / n; line[2] = l[2] / n;
}

/* src/sparse_img_align.cpp:229-230 (also :327-330, 422-423, 568-571): f * ||pos - ref_pos|| */
void plsvo_oracle_scaled_bearing(const double f[3], const double pos[3], const double ref_pos[3], double out[3]) {
  const double d0 = pos[0] - ref_pos[0], d1 = pos[1] - ref_pos[1], d2 = pos[2] - ref_pos[2];
  const double depth = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
  out[0] = f[0] * depth; out[1] = f[1] * depth; out[2] = f[2] * depth;
}

/* Patch  include/plsvo/feature.h:107-147, src/feature.cpp:175-218 */
typedef struct {
  float u_ref, v_ref; int u_ref_i, v_ref_i;
  float wTL, wTR, wBL, wBR;
  const uint8_t* img; int cols, rows, stride;
} patch_t;
enum { P_SIZE = 4, P_HALF = 2, P_AREA = 16 };

static void patch_init(patch_t* p, const plsvo_oracle_pyr* pyr, int level) {
  p->img = pyr->img[level]; p->cols = pyr->width[level]; p->rows = pyr->height[level]; p->stride = pyr->stride[level];
}
static void patch_set_position(patch_t* p, double px, double py) { /* feature.cpp:189-197 */
  p->u_ref = (float)px; p->v_ref = (float)py;
  p->u_ref_i = (int)floorf(p->u_ref); p->v_ref_i = (int)floorf(p->v_ref);
}
static int patch_is_in_frame(const patch_t* p, int boundary) { /* feature.h:139-144 */
  return !(p->u_ref_i < boundary || p->v_ref_i < boundary || p->u_ref_i >= p->cols - boundary || p->v_ref_i >= p->rows - boundary);
}
static void patch_interp_weights(patch_t* p) { /* feature.cpp:199-208: double arithmetic, float storage */
  const float subpix_u_ref = p->u_ref - p->u_ref_i;
  const float subpix_v_ref = p->v_ref - p->v_ref_i;
  p->wTL = (1.0 - subpix_u_ref) * (1.0 - subpix_v_ref);
  p->wTR = subpix_u_ref * (1.0 - subpix_v_ref);
  p->wBL = (1.0 - subpix_u_ref) * subpix_v_ref;
  p->wBR = subpix_u_ref * subpix_v_ref;
}
/* pointer to ROI row y: cv::Rect(u_i-2, v_i-2, 4, 4)  feature.cpp:210-218 */
static const uint8_t* patch_row(const patch_t* p, int y) {
  return p->img + (ptrdiff_t)(p->v_ref_i - P_HALF + y) * p->stride + (p->u_ref_i - P_HALF);
}

/* ============================================================================================ */
/* SparseImgAlign  src/sparse_img_align.cpp                                                      */
/* ============================================================================================ */

typedef struct { /* SparseImgAlign::Cache  sparse_img_align.h:83-96 */
  double* jacobian;   /* 6 x (n*16), column-major */
  float* ref_patch;   /* n x 16 */
  uint8_t* visible;   /* n */
  size_t n;
} cache_t;

typedef struct {
  const plsvo_align_in* in; const plsvo_oracle_pyr* ref; const plsvo_oracle_pyr* cur;
  int level;
  cache_t pt_cache, seg_cache;
  size_t* patch_offset;
  int have_ref_patch_cache;
  uint8_t* seg_alive; /* LineFeat::feat3D != NULL */
  /* [ext] vk::NLLSSolver<6,SE3> state */
  double H[36], Jres[6], x[6];
  double chi2; size_t n_meas; int iter, n_iter, stop, use_weights;
  double scale_, scale_ls, scale_pt;
} align_t;

static void cache_alloc(cache_t* c, size_t n) {
  c->n = n;
  c->jacobian = (double*)calloc(6 * (n ? n : 1) * P_AREA, sizeof(double));
  c->ref_patch = (float*)calloc((n ? n : 1) * P_AREA, sizeof(float));
  c->visible = (uint8_t*)calloc(n ? n : 1, 1);
}
static void cache_free(cache_t* c) { free(c->jacobian); free(c->ref_patch); free(c->visible); }

/* the 16-pixel inner loop shared by :243-264 and :354-375 */
static void precompute_patch(const patch_t* patch, const double frame_jac[12], double focal_over_scale,
                             float* cache_ptr, double* jac_cols) {
  const int stride = patch->stride;
  const float wTL = patch->wTL, wTR = patch->wTR, wBL = patch->wBL, wBR = patch->wBR;
  int pixel_counter = 0;
  for (int y = 0; y < P_SIZE; ++y) {
    const uint8_t* img_ptr = patch_row(patch, y);
    for (int x = 0; x < P_SIZE; ++x, ++img_ptr, ++cache_ptr, ++pixel_counter) {
      *cache_ptr = wTL * img_ptr[0] + wTR * img_ptr[1] + wBL * img_ptr[stride] + wBR * img_ptr[stride + 1];
      float dx = 0.5f * ((wTL * img_ptr[1] + wTR * img_ptr[2] + wBL * img_ptr[stride + 1] + wBR * img_ptr[stride + 2])
                       - (wTL * img_ptr[-1] + wTR * img_ptr[0] + wBL * img_ptr[stride - 1] + wBR * img_ptr[stride]));
      float dy = 0.5f * ((wTL * img_ptr[stride] + wTR * img_ptr[1 + stride] + wBL * img_ptr[stride * 2] + wBR * img_ptr[stride * 2 + 1])
                       - (wTL * img_ptr[-stride] + wTR * img_ptr[1 - stride] + wBL * img_ptr[0] + wBR * img_ptr[1]));
      double* col = jac_cols + 6 * pixel_counter;
      for (int k = 0; k < 6; ++k) col[k] = (dx * frame_jac[k] + dy * frame_jac[6 + k]) * focal_over_scale;
    }
  }
}

/* precomputeGaussNewtonParamsPoints  :195-268 */
static void precompute_points(align_t* a) {
  const plsvo_align_in* in = a->in;
  patch_t patch; patch_init(&patch, a->ref, a->level);
  const float scale = 1.0f / (1 << a->level);
  const double focal_length = fabs(in->cam.fx); /* errorMultiplier2() */
  for (int i = 0; i < in->n_pts; ++i) {
    /* feat3D == NULL entries are not part of the flattened input */
    patch_set_position(&patch, in->pt_px[2 * i] * scale, in->pt_px[2 * i + 1] * scale);
    if (!patch_is_in_frame(&patch, P_HALF + 1)) continue;
    patch_interp_weights(&patch);
    a->pt_cache.visible[i] = 1;
    double frame_jac[12];
    plsvo_oracle_jacobian_xyz2uv(&in->pt_xyz_ref[3 * i], frame_jac);
    precompute_patch(&patch, frame_jac, focal_length / (1 << a->level),
                     a->pt_cache.ref_patch + (size_t)P_AREA * i, a->pt_cache.jacobian + (size_t)6 * P_AREA * i);
  }
}

/* precomputeGaussNewtonParamsSegments  :270-378 */
static void precompute_segments(align_t* a) {
  const plsvo_align_in* in = a->in;
  patch_t patch; patch_init(&patch, a->ref, a->level);
  const float scale = 1.0f / (1 << a->level);
  const double focal_length = fabs(in->cam.fx);
  size_t cache_idx = 0;
  for (int s = 0; s < in->n_seg; ++s) {
    a->patch_offset[s] = cache_idx;
    if (!a->seg_alive[s]) continue;
    const double* spx = &in->seg_spx[2 * s]; const double* epx = &in->seg_epx[2 * s];
    /* (spx*scale).cast<int>() truncates toward zero */
    if (!cam_is_in_frame(&in->cam, (int)(spx[0] * scale), (int)(spx[1] * scale), P_HALF + 1, a->level) ||
        !cam_is_in_frame(&in->cam, (int)(epx[0] * scale), (int)(epx[1] * scale), P_HALF + 1, a->level))
      continue;
    a->seg_cache.visible[s] = 1;
    double inc2d[2];
    size_t N_samples = plsvo_oracle_setup_sampling(spx, epx, in->seg_len[s], P_SIZE, inc2d);
    N_samples = 1 + (N_samples - 1) / (1 << a->level);
    inc2d[0] = inc2d[0] * scale / (double)(N_samples - 1);
    inc2d[1] = inc2d[1] * scale / (double)(N_samples - 1);
    double px_ref[2] = { spx[0] * scale, spx[1] * scale };
    const double* p_ref = &in->seg_p_ref[3 * s]; const double* q_ref = &in->seg_q_ref[3 * s];
    double inc3d[3], xyz_ref[3];
    for (int k = 0; k < 3; ++k) { inc3d[k] = (q_ref[k] - p_ref[k]) / (double)(N_samples - 1); xyz_ref[k] = p_ref[k]; }
    for (size_t sample = 0; sample < N_samples; ++sample) {
      patch_set_position(&patch, px_ref[0], px_ref[1]);
      patch_interp_weights(&patch);
      double frame_jac[12];
      plsvo_oracle_jacobian_xyz2uv(xyz_ref, frame_jac);
      precompute_patch(&patch, frame_jac, focal_length / (1 << a->level),
                       a->seg_cache.ref_patch + cache_idx, a->seg_cache.jacobian + 6 * cache_idx);
      cache_idx += P_AREA;
      px_ref[0] += inc2d[0]; px_ref[1] += inc2d[1];
      for (int k = 0; k < 3; ++k) xyz_ref[k] += inc3d[k];
    }
  }
}

/* computeGaussNewtonParamsPoints  :380-502  (linearize_system=true, compute_weight_scale=false, use_weights_=true) */
static void compute_points(align_t* a, const se3_t* T, double H[36], double Jres[6], float* chi2) {
  const plsvo_align_in* in = a->in;
  patch_t patch; patch_init(&patch, a->cur, a->level);
  const float scale = 1.0f / (1 << a->level);
  *chi2 = 0.0;
  memset(H, 0, 36 * sizeof(double)); memset(Jres, 0, 6 * sizeof(double));
  for (int i = 0; i < in->n_pts; ++i) {
    if (!a->pt_cache.visible[i]) continue;
    double xyz_cur[3], uv[2];
    se3_act(T, &in->pt_xyz_ref[3 * i], xyz_cur);
    plsvo_oracle_world2cam(&in->cam, xyz_cur, uv);
    patch_set_position(&patch, uv[0] * scale, uv[1] * scale);
    if (!patch_is_in_frame(&patch, P_HALF)) continue;
    patch_interp_weights(&patch);
    size_t pixel_counter = 0;
    const float* cache_ptr = a->pt_cache.ref_patch + (size_t)P_AREA * i;
    const int stride = patch.stride;
    for (int y = 0; y < P_SIZE; ++y) {
      const uint8_t* img_ptr = patch_row(&patch, y);
      for (int x = 0; x < P_SIZE; ++x, ++img_ptr, ++cache_ptr, ++pixel_counter) {
        const float intensity_cur = patch.wTL * img_ptr[0] + patch.wTR * img_ptr[1] + patch.wBL * img_ptr[stride] + patch.wBR * img_ptr[stride + 1];
        const float res = intensity_cur - (*cache_ptr);
        float weight = 1.0;
        weight = 1.0 / (1.0 + fabsf(res)); /* :479 (the scale_pt branch :470-475 is unreachable) */
        *chi2 += res * res * weight;
        a->n_meas++;
        const double* J = a->pt_cache.jacobian + 6 * ((size_t)i * P_AREA + pixel_counter);
        for (int r = 0; r < 6; ++r) {
          for (int c = 0; c < 6; ++c) H[r * 6 + c] += J[r] * J[c] * weight;
          Jres[r] -= J[r] * res * weight;
        }
      }
    }
  }
}

/* computeGaussNewtonParamsSegments  :504-695 */
static void compute_segments(align_t* a, const se3_t* T, double H[36], double Jres[6], float* chi2) {
  const plsvo_align_in* in = a->in;
  patch_t patch; patch_init(&patch, a->cur, a->level);
  const float scale = 1.0f / (1 << a->level);
  *chi2 = 0.0;
  memset(H, 0, 36 * sizeof(double)); memset(Jres, 0, 6 * sizeof(double));
  float* ls_res = (float*)malloc(sizeof(float) * (a->seg_cache.n * P_AREA + P_AREA));
  for (int s = 0; s < in->n_seg; ++s) {
    if (!a->seg_alive[s]) continue;
    if (!a->seg_cache.visible[s]) continue;
    size_t cache_idx = a->patch_offset[s];
    double inc2d[2];
    size_t N_samples = plsvo_oracle_setup_sampling(&in->seg_spx[2 * s], &in->seg_epx[2 * s], in->seg_len[s], P_SIZE, inc2d);
    N_samples = 1 + (N_samples - 1) / (1 << a->level);
    const double* p_ref = &in->seg_p_ref[3 * s]; const double* q_ref = &in->seg_q_ref[3 * s];
    double inc3d[3], xyz_ref[3];
    for (int k = 0; k < 3; ++k) { inc3d[k] = (q_ref[k] - p_ref[k]) / (double)(N_samples - 1); xyz_ref[k] = p_ref[k]; }
    double H_[36], Jres_[6];
    memset(H_, 0, sizeof(H_)); memset(Jres_, 0, sizeof(Jres_));
    size_t n_res = 0;
    int good_line = 1;
    for (size_t sample = 0; sample < N_samples; ++sample) {
      double xyz_cur[3], uv[2];
      se3_act(T, xyz_ref, xyz_cur);
      plsvo_oracle_world2cam(&in->cam, xyz_cur, uv);
      patch_set_position(&patch, uv[0] * scale, uv[1] * scale);
      if (!patch_is_in_frame(&patch, P_HALF)) { good_line = 0; break; } /* :588-594 */
      patch_interp_weights(&patch);
      const float* cache_ptr = a->seg_cache.ref_patch + cache_idx;
      const int stride = patch.stride;
      for (int y = 0; y < P_SIZE; ++y) {
        const uint8_t* img_ptr = patch_row(&patch, y);
        for (int x = 0; x < P_SIZE; ++x, ++img_ptr, ++cache_ptr, ++cache_idx) {
          const float intensity_cur = patch.wTL * img_ptr[0] + patch.wTR * img_ptr[1] + patch.wBL * img_ptr[stride] + patch.wBR * img_ptr[stride + 1];
          const float res = intensity_cur - (*cache_ptr);
          ls_res[n_res++] = res;
          const double* J = a->seg_cache.jacobian + 6 * cache_idx;
          for (int r = 0; r < 6; ++r) {
            for (int c = 0; c < 6; ++c) H_[r * 6 + c] += J[r] * J[c];
            Jres_[r] -= J[r] * res;
          }
        }
      }
      for (int k = 0; k < 3; ++k) xyz_ref[k] += inc3d[k];
    }
    float res_ = 0.0;
    for (size_t k = 0; k < n_res; ++k) res_ += fabsf(ls_res[k]);
    res_ = res_ / (double)N_samples;
    if (good_line && res_ < 200.0) {
      float weight = 1.0;
      weight = 1.0 / (1.0 + res_); /* :675 */
      for (int k = 0; k < 36; ++k) H[k] += H_[k] * weight / res_;
      for (int k = 0; k < 6; ++k) Jres[k] += Jres_[k] * weight;
      *chi2 += res_ * res_ * weight;
      a->n_meas++;
    } else {
      a->seg_alive[s] = 0; /* it->feat3D = NULL  :687-688 */
    }
  }
  free(ls_res);
}

/* computeResiduals  :112-193 */
static double compute_residuals(align_t* a, const se3_t* T) {
  if (!a->have_ref_patch_cache) { /* precomputeReferencePatches :104-110 */
    precompute_points(a);
    precompute_segments(a);
    a->have_ref_patch_cache = 1;
  }
  a->use_weights = 1; /* :130-132 */
  double pt_H[36], pt_Jres[6], seg_H[36], seg_Jres[6];
  float pt_chi2 = 0.0, seg_chi2 = 0.0;
  compute_points(a, T, pt_H, pt_Jres, &pt_chi2);
  compute_segments(a, T, seg_H, seg_Jres, &seg_chi2);
  for (int k = 0; k < 36; ++k) a->H[k] = pt_H[k] + seg_H[k];
  for (int k = 0; k < 6; ++k) a->Jres[k] = pt_Jres[k] + seg_Jres[k];
  float chi2 = pt_chi2 + seg_chi2;
  if (a->iter == 0) { a->scale_ = 1.0; a->scale_ls = 1.0; a->scale_pt = 1.0; } /* :186-190, never read */
  return chi2 / a->n_meas; /* float / size_t -> float division */
}

static void log_iter(plsvo_align_iterlog* log, int max_log, int* n_log, const align_t* a, int accepted,
                     double new_chi2, const se3_t* model) {
  if (!log || *n_log >= max_log) { if (n_log) ++*n_log; return; }
  plsvo_align_iterlog* r = &log[*n_log];
  r->level = a->level; r->iter = a->iter; r->accepted = accepted; r->stop = a->stop;
  r->n_meas = a->n_meas; r->new_chi2 = new_chi2;
  memcpy(r->H, a->H, sizeof(r->H)); memcpy(r->Jres, a->Jres, sizeof(r->Jres)); memcpy(r->x, a->x, sizeof(r->x));
  se3_store(model, r->T_after);
  ++*n_log;
}

/* [ext] vk::NLLSSolver<6,SE3>::optimize -> optimizeGaussNewton, with SparseImgAlign::solve :697-703
 * and ::update :705-710 */
static void optimize_gauss_newton(align_t* a, se3_t* model, int* iters_out, plsvo_align_iterlog* log, int max_log, int* n_log) {
  if (a->use_weights) compute_residuals(a, model); /* weight-scale pass; result-idempotent here */
  se3_t old_model = *model;
  int iters = 0;
  for (a->iter = 0; a->iter < a->n_iter; ++a->iter) {
    memset(a->H, 0, sizeof(a->H)); memset(a->Jres, 0, sizeof(a->Jres));
    a->n_meas = 0;
    const double new_chi2 = compute_residuals(a, model);
    ++iters;
    plsvo_oracle_ldlt_solve6(a->H, a->Jres, a->x); /* solve() */
    if (isnan(a->x[0])) a->stop = 1;
    if ((a->iter > 0 && new_chi2 > a->chi2) || a->stop) {
      *model = old_model; /* rollback */
      log_iter(log, max_log, n_log, a, 0, new_chi2, model);
      break;
    }
    double mx[6]; for (int k = 0; k < 6; ++k) mx[k] = -a->x[k];
    const se3_t ex = se3_exp(mx);
    const se3_t new_model = se3_mul(model, &ex); /* update(): T_old * exp(-x) */
    old_model = *model;
    *model = new_model;
    a->chi2 = new_chi2;
    log_iter(log, max_log, n_log, a, 1, new_chi2, model);
    if (norm_max6(a->x) <= a->in->eps) break;
  }
  *iters_out = iters;
}

int plsvo_oracle_sparse_align(const plsvo_align_in* in, const plsvo_oracle_pyr* ref, const plsvo_oracle_pyr* cur,
                              plsvo_align_out* out, plsvo_align_iterlog* log, int max_log, int* n_log) {
  if (!in || !ref || !cur || !out) return PLSVO_E_INVALID;
  if (in->max_level >= ref->n_levels || in->max_level >= cur->n_levels || in->min_level < 0 || in->max_level >= PLSVO_MAX_LEVELS)
    return PLSVO_E_INVALID;
  int dummy_n = 0; if (!n_log) n_log = &dummy_n; *n_log = 0;
  align_t a; memset(&a, 0, sizeof(a));
  a.in = in; a.ref = ref; a.cur = cur;
  /* reset()  [ext] */
  a.chi2 = 1e10; a.n_meas = 0; a.n_iter = in->n_iter; a.iter = 0; a.stop = 0; a.use_weights = 0;
  uint8_t* alive_out = out->seg_alive_out;
  memset(out, 0, sizeof(*out)); out->seg_alive_out = alive_out;
  se3_t T = se3_load(in->T_cur_from_ref);
  if (in->n_pts == 0 && in->n_seg == 0) { /* :58-62 */
    se3_store(&T, out->T_cur_from_ref); out->chi2 = a.chi2; return 0;
  }
  /* :69-78 cache sizing */
  float total_length = 0;
  for (int s = 0; s < in->n_seg; ++s) total_length += in->seg_len[s];
  const int max_num_seg_samples = (int)ceilf(total_length / P_SIZE);
  cache_alloc(&a.pt_cache, (size_t)in->n_pts);
  cache_alloc(&a.seg_cache, (size_t)(max_num_seg_samples > 0 ? max_num_seg_samples : 0));
  free(a.seg_cache.visible); a.seg_cache.visible = (uint8_t*)calloc(in->n_seg ? in->n_seg : 1, 1);
  a.patch_offset = (size_t*)calloc(in->n_seg ? in->n_seg : 1, sizeof(size_t));
  a.seg_alive = (uint8_t*)malloc(in->n_seg ? in->n_seg : 1);
  for (int s = 0; s < in->n_seg; ++s) a.seg_alive[s] = in->seg_alive_in ? (in->seg_alive_in[s] != 0) : 1;

  for (a.level = in->max_level; a.level >= in->min_level; --a.level) { /* :82-91 */
    memset(a.pt_cache.jacobian, 0, sizeof(double) * 6 * (a.pt_cache.n ? a.pt_cache.n : 1) * P_AREA);
    memset(a.seg_cache.jacobian, 0, sizeof(double) * 6 * (a.seg_cache.n ? a.seg_cache.n : 1) * P_AREA);
    a.have_ref_patch_cache = 0;
    int iters = 0;
    optimize_gauss_newton(&a, &T, &iters, log, max_log, n_log);
    out->iters_per_level[a.level] = iters;
  }
  se3_store(&T, out->T_cur_from_ref);
  out->n_meas = a.n_meas; out->n_tracked = a.n_meas / P_AREA;
  memcpy(out->H, a.H, sizeof(out->H));
  out->chi2 = a.chi2; out->status = a.stop ? 1 : 0;
  if (out->seg_alive_out) memcpy(out->seg_alive_out, a.seg_alive, (size_t)in->n_seg);
  cache_free(&a.pt_cache); cache_free(&a.seg_cache); free(a.patch_offset); free(a.seg_alive);
  return 0;
}

/* ============================================================================================ */
/* pose_optimizer::optimizeGaussNewton  src/pose_optimizer.cpp:38-260 (9-arg), :262-582 (10-arg)  */
/* ============================================================================================ */

static void project2d(const double v[3], double out[2]) { out[0] = v[0] / v[2]; out[1] = v[1] / v[2]; } /* [ext] vk::project2d */
static double norm2(const double e[2]) { return sqrt(e[0] * e[0] + e[1] * e[1]); }
static double sqnorm2(const double e[2]) { return e[0] * e[0] + e[1] * e[1]; }

typedef struct {
  const plsvo_poseopt_in* in;
  uint8_t* pt_keep; uint8_t* seg_keep;
  se3_t T, T_old; double chi2;
  double A[36], b[6];
  double* chi2_vec_init; size_t n_init;
} popt_t;

/* one pass of the GN loop body :103-195 (identical text at :339-431 and :469-563); returns 1 to continue */
static int popt_gn_loop(popt_t* p, size_t n_iter, double scale_pt, double scale_ls, int phase, int* iters_out,
                        plsvo_poseopt_iterlog* log, int max_log, int* n_log) {
  const plsvo_poseopt_in* in = p->in;
  int iters = 0;
  for (size_t iter = 0; iter < n_iter; iter++) {
    memset(p->b, 0, sizeof(p->b)); memset(p->A, 0, sizeof(p->A));
    double new_chi2 = 0.0;
    ++iters;
    for (int i = 0; i < in->n_pts; ++i) {
      if (!p->pt_keep[i]) continue;
      double J[12], xyz_f[3], pf[2], pp[2], e[2];
      se3_act(&p->T, &in->pt_pos[3 * i], xyz_f);
      plsvo_oracle_jacobian_xyz2uv(xyz_f, J);
      project2d(&in->pt_f[3 * i], pf); project2d(xyz_f, pp);
      e[0] = pf[0] - pp[0]; e[1] = pf[1] - pp[1];
      const double sqrt_inv_cov = 1.0 / (1 << in->pt_level[i]);
      e[0] *= sqrt_inv_cov; e[1] *= sqrt_inv_cov;
      if (iter == 0) p->chi2_vec_init[p->n_init++] = sqnorm2(e);
      for (int k = 0; k < 12; ++k) J[k] *= sqrt_inv_cov;
      const double weight = plsvo_oracle_tukey((float)(norm2(e) / scale_pt));
      for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) p->A[r * 6 + c] += (J[r] * J[c] + J[6 + r] * J[6 + c]) * weight;
        p->b[r] -= (J[r] * e[0] + J[6 + r] * e[1]) * weight;
      }
      new_chi2 += sqnorm2(e) * weight;
    }
    for (int s = 0; s < in->n_seg; ++s) {
      if (!p->seg_keep[s]) continue;
      double J_s[12], J_e[12], J[12], xs[3], xe[3], e[2];
      se3_act(&p->T, &in->seg_spos[3 * s], xs);
      se3_act(&p->T, &in->seg_epos[3 * s], xe);
      plsvo_oracle_jacobian_xyz2uv(xs, J_s);
      plsvo_oracle_jacobian_xyz2uv(xe, J_e);
      double sp[2], ep[2];
      project2d(xs, sp); project2d(xe, ep);
      const double* line = &in->seg_line[3 * s];
      float ds = line[0] * sp[0] + line[1] * sp[1] + line[2] * 1.0;
      float de = line[0] * ep[0] + line[1] * ep[1] + line[2] * 1.0;
      e[0] = ds; e[1] = de;
      const double sqrt_inv_cov = 1.0 / (1 << in->seg_level[s]);
      e[0] *= sqrt_inv_cov; e[1] *= sqrt_inv_cov;
      if (iter == 0) p->chi2_vec_init[p->n_init++] = sqnorm2(e);
      const double k_s = sqrt_inv_cov * ds / norm2(e); /* the same factor for both rows; `de` unused (:156-157) */
      for (int k = 0; k < 12; ++k) { J_s[k] *= k_s; J_e[k] *= k_s; }
      for (int c = 0; c < 6; ++c) {
        J[c] = line[0] * J_s[c] + line[1] * J_s[6 + c];
        J[6 + c] = line[0] * J_e[c] + line[1] * J_e[6 + c];
      }
      const double weight = plsvo_oracle_tukey((float)(norm2(e) / scale_ls));
      for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) p->A[r * 6 + c] += (J[r] * J[c] + J[6 + r] * J[6 + c]) * weight;
        p->b[r] -= (J[r] * e[0] + J[6 + r] * e[1]) * weight;
      }
      new_chi2 += sqnorm2(e) * weight;
    }
    double dT[6];
    plsvo_oracle_ldlt_solve6(p->A, p->b, dT);
    int accepted = 1;
    if ((iter > 0 && new_chi2 > p->chi2) || isnan(dT[0])) {
      p->T = p->T_old; /* roll-back */
      accepted = 0;
    } else {
      const se3_t ex = se3_exp(dT);
      const se3_t T_new = se3_mul(&ex, &p->T); /* SE3::exp(dT) * T */
      p->T_old = p->T; p->T = T_new; p->chi2 = new_chi2;
    }
    if (log && *n_log < max_log) {
      plsvo_poseopt_iterlog* r = &log[*n_log];
      r->phase = phase; r->iter = (int)iter; r->accepted = accepted; r->reserved0 = 0; r->new_chi2 = new_chi2;
      memcpy(r->A, p->A, sizeof(r->A)); memcpy(r->b, p->b, sizeof(r->b)); memcpy(r->dT, dT, sizeof(r->dT));
      se3_store(&p->T, r->T_after);
    }
    ++*n_log;
    if (!accepted) break;
    if (norm_max6(dT) <= 0.0000000001 /* EPS, global.h:99 */) break;
  }
  *iters_out = iters;
  return 0;
}

/* residual used by the scale pass (:61-87) and the cull pass (:205-240) for one segment */
static void seg_endpoint_dists(const se3_t* T, const double* line, const double* spos, const double* epos, double* es, double* ee) {
  double xs[3], xe[3], sp[2], ep[2];
  se3_act(T, spos, xs); se3_act(T, epos, xe);
  project2d(xs, sp); project2d(xe, ep);
  *es = line[0] * sp[0] + line[1] * sp[1] + line[2] * 1.0;
  *ee = line[0] * ep[0] + line[1] * ep[1] + line[2] * 1.0;
}

int plsvo_oracle_pose_optimize(const plsvo_poseopt_in* in, plsvo_poseopt_out* out, plsvo_poseopt_iterlog* log, int max_log, int* n_log) {
  if (!in || !out) return PLSVO_E_INVALID;
  int dummy_n = 0; if (!n_log) n_log = &dummy_n; *n_log = 0;
  uint8_t* pk = out->pt_keep; uint8_t* sk = out->seg_keep;
  memset(out, 0, sizeof(*out)); out->pt_keep = pk; out->seg_keep = sk;
  const int np = in->n_pts, ns = in->n_seg;
  popt_t p; memset(&p, 0, sizeof(p));
  p.in = in;
  p.pt_keep = (uint8_t*)malloc(np ? np : 1); p.seg_keep = (uint8_t*)malloc(ns ? ns : 1);
  memset(p.pt_keep, 1, np ? np : 1); memset(p.seg_keep, 1, ns ? ns : 1);
  p.T = se3_load(in->T_f_w); p.T_old = p.T; p.chi2 = 0.0;
  p.chi2_vec_init = (double*)malloc(sizeof(double) * 2 * (size_t)(np + ns + 1));
  double* chi2_vec_final = (double*)malloc(sizeof(double) * (size_t)(np + ns + 1));
  float* errors = (float*)malloc(sizeof(float) * (size_t)(np + ns + 1));
  float* errors_ls = (float*)malloc(sizeof(float) * (size_t)(ns + 1));
  size_t n_err = 0, n_err_ls = 0, n_final = 0;
  int rc = 0;

  /* scale pass :57-95 */
  for (int i = 0; i < np; ++i) {
    double pf[2], pp[2], xyz[3], e[2];
    project2d(&in->pt_f[3 * i], pf);
    se3_act(&p.T, &in->pt_pos[3 * i], xyz); project2d(xyz, pp);
    e[0] = pf[0] - pp[0]; e[1] = pf[1] - pp[1];
    const double s = 1.0 / (1 << in->pt_level[i]);
    e[0] *= s; e[1] *= s;
    errors[n_err++] = (float)norm2(e);
  }
  /* :70 runs the estimator on the point errors even when there are none (undefined behaviour in the
   * reference: getMedian asserts non-empty).  Documented deviation: zero points -> scale_pt = 1.0. */
  double estimated_scale_pt = (n_err > 0) ? (double)plsvo_oracle_mad_scale(errors, n_err) : 1.0;
  out->num_obs_pt = n_err;
  for (int s = 0; s < ns; ++s) {
    double es_d, ee_d;
    seg_endpoint_dists(&p.T, &in->seg_line[3 * s], &in->seg_spos[3 * s], &in->seg_epos[3 * s], &es_d, &ee_d);
    float es = es_d, ee = ee_d;
    const float v = sqrtf(es * es + ee * ee);
    errors[n_err++] = v; errors_ls[n_err_ls++] = v;
  }
  if (n_err == 0) { /* :88-89 early return, nothing else written */
    out->status = 1; se3_store(&p.T, out->T_f_w);
    if (out->pt_keep && np > 0) memset(out->pt_keep, 1, (size_t)np);
    if (out->seg_keep && ns > 0) memset(out->seg_keep, 1, (size_t)ns);
    goto done;
  }
  out->num_obs_ls = n_err_ls;
  double estimated_scale_ls = 1.f;
  if (n_err_ls > 0) estimated_scale_ls = plsvo_oracle_mad_scale(errors_ls, n_err_ls);
  double estimated_scale = estimated_scale_pt;
  const double scale_pt = estimated_scale_pt, scale_ls = estimated_scale_ls;

  popt_gn_loop(&p, (size_t)in->n_iter, scale_pt, scale_ls, 0, &out->iters, log, max_log, n_log);

  /* covariance :197-199 */
  {
    double Af[36];
    const double f2 = pow(in->fx, 2);
    for (int k = 0; k < 36; ++k) Af[k] = p.A[k] * f2;
    plsvo_oracle_inv6(Af, out->cov);
  }
  /* cull :201-242 */
  const double reproj_thresh_scaled_pt = in->reproj_thresh / in->fx;
  const double reproj_thresh_scaled_ls = reproj_thresh_scaled_pt * estimated_scale_ls / estimated_scale_pt;
  size_t n_deleted_refs_pt = 0, n_deleted_refs_ls = 0;
  for (int i = 0; i < np; ++i) {
    double pf[2], pp[2], xyz[3], e[2];
    project2d(&in->pt_f[3 * i], pf);
    se3_act(&p.T, &in->pt_pos[3 * i], xyz); project2d(xyz, pp);
    e[0] = pf[0] - pp[0]; e[1] = pf[1] - pp[1];
    const double s = 1.0 / (1 << in->pt_level[i]);
    e[0] *= s; e[1] *= s;
    chi2_vec_final[n_final++] = sqnorm2(e);
    if (norm2(e) > reproj_thresh_scaled_pt) { p.pt_keep[i] = 0; ++n_deleted_refs_pt; }
  }
  for (int s = 0; s < ns; ++s) {
    double e[2];
    seg_endpoint_dists(&p.T, &in->seg_line[3 * s], &in->seg_spos[3 * s], &in->seg_epos[3 * s], &e[0], &e[1]);
    const double c = 1.0 / (1 << in->seg_level[s]);
    e[0] *= c; e[1] *= c;
    chi2_vec_final[n_final++] = sqnorm2(e);
    if (norm2(e) > reproj_thresh_scaled_ls) { p.seg_keep[s] = 0; ++n_deleted_refs_ls; }
  }
  /* refinement with inliers :469-563 (10-argument overload only); chi2 and T_old carry over */
  if (in->n_iter_ref >= 0)
    popt_gn_loop(&p, (size_t)in->n_iter_ref, scale_pt, scale_ls, 1, &out->iters_ref, log, max_log, n_log);

  out->error_init = 0.0; out->error_final = 0.0;
  if (p.n_init > 0) out->error_init = sqrt(plsvo_oracle_median_f64(p.chi2_vec_init, p.n_init)) * in->fx;
  if (n_final > 0) out->error_final = sqrt(plsvo_oracle_median_f64(chi2_vec_final, n_final)) * in->fx;
  estimated_scale *= in->fx;
  out->estimated_scale = estimated_scale;
  out->num_obs_pt -= n_deleted_refs_pt;
  out->num_obs_ls -= n_deleted_refs_ls;
  se3_store(&p.T, out->T_f_w);
  if (out->pt_keep) memcpy(out->pt_keep, p.pt_keep, np);
  if (out->seg_keep) memcpy(out->seg_keep, p.seg_keep, ns);
done:
  free(p.pt_keep); free(p.seg_keep); free(p.chi2_vec_init); free(chi2_vec_final); free(errors); free(errors_ls);
  return rc;
}

/* ============================================================================================ */
/* Point::optimize / LineSeg::optimize  src/feature3D_impl.cpp:36-174                            */
/* ============================================================================================ */

/* [ext] Eigen::LDLT<Matrix3d>::compute + solve, the 3x3 instance of the algorithm restated above */
static void ldlt_solve3(const double A[9], const double b[3], double x[3]) { ldlt_solve_n(3, A, b, x); }

/* one observation's contribution: Point::jacobian_xyz2uv (include/plsvo/feature3D.h:126-140),
 * e = project2d(f) - project2d(p_in_f), A += J^T J, b -= J^T e, chi2 += |e|^2  (feature3D_impl.cpp:49-58) */
static void structopt_accumulate(const se3_t* T, const double R[9], const double f[3], const double pos[3],
                                 double A[9], double b[3], double* chi2) {
  double p[3];
  se3_act(T, pos, p);
  const double z_inv = 1.0 / p[2];
  const double z_inv_sq = z_inv * z_inv;
  const double P[6] = { z_inv, 0.0, -p[0] * z_inv_sq, 0.0, z_inv, -p[1] * z_inv_sq };
  double J[6];
  for (int r = 0; r < 2; ++r)
    for (int c = 0; c < 3; ++c)   /* point_jac = - point_jac * R_f_w  parses as (-point_jac) * R_f_w */
      J[r * 3 + c] = (-P[r * 3 + 0]) * R[0 * 3 + c] + (-P[r * 3 + 1]) * R[1 * 3 + c] + (-P[r * 3 + 2]) * R[2 * 3 + c];
  const double e0 = f[0] / f[2] - p[0] / p[2], e1 = f[1] / f[2] - p[1] / p[2];
  *chi2 += e0 * e0 + e1 * e1;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) A[i * 3 + j] += J[i] * J[j] + J[3 + i] * J[3 + j];
    b[i] -= J[i] * e0 + J[3 + i] * e1;
  }
}

static double norm_max3(const double v[3]) { double m = 0; for (int i = 0; i < 3; ++i) { const double a = fabs(v[i]); if (a > m) m = a; } return m; }

int plsvo_oracle_structure_optimize(const plsvo_structopt_in* in, plsvo_structopt_out* out) {
  if (!in || !out) return PLSVO_E_INVALID;
  se3_t* T = (se3_t*)malloc(sizeof(se3_t) * (size_t)(in->n_frames > 0 ? in->n_frames : 1));
  double* Rm = (double*)malloc(sizeof(double) * 9 * (size_t)(in->n_frames > 0 ? in->n_frames : 1));
  for (int k = 0; k < in->n_frames; ++k) { T[k] = se3_load(in->frame_T + 7 * k); quat_to_matrix(T[k].q, Rm + 9 * k); }
  /* Point::optimize :36-95 */
  for (int i = 0; i < in->n_pts; ++i) {
    double pos[3] = { in->pt_pos[3 * i], in->pt_pos[3 * i + 1], in->pt_pos[3 * i + 2] };
    double old_point[3] = { pos[0], pos[1], pos[2] };
    double chi2 = 0.0; int iters = 0;
    for (int it = 0; it < in->n_iter_pts; ++it) {
      double A[9] = { 0 }, b[3] = { 0 }, new_chi2 = 0.0;
      ++iters;
      for (int o = in->pt_obs_off[i]; o < in->pt_obs_off[i + 1]; ++o) {
        const int fr = in->pt_obs_frame[o];
        structopt_accumulate(&T[fr], Rm + 9 * fr, in->pt_obs_f + 3 * o, pos, A, b, &new_chi2);
      }
      double dp[3];
      ldlt_solve3(A, b, dp);
      if ((it > 0 && new_chi2 > chi2) || isnan(dp[0])) { pos[0] = old_point[0]; pos[1] = old_point[1]; pos[2] = old_point[2]; break; }
      for (int k = 0; k < 3; ++k) { old_point[k] = pos[k]; pos[k] = pos[k] + dp[k]; }
      chi2 = new_chi2;
      if (norm_max3(dp) <= 0.0000000001) break;
    }
    if (out->pt_pos) { out->pt_pos[3 * i] = pos[0]; out->pt_pos[3 * i + 1] = pos[1]; out->pt_pos[3 * i + 2] = pos[2]; }
    if (out->pt_iters) out->pt_iters[i] = iters;
  }
  /* LineSeg::optimize :97-174: both end points advance together and roll back together */
  for (int i = 0; i < in->n_seg; ++i) {
    double sp[3], ep[3], old_s[3], old_e[3];
    for (int k = 0; k < 3; ++k) { sp[k] = old_s[k] = in->seg_spos[3 * i + k]; ep[k] = old_e[k] = in->seg_epos[3 * i + k]; }
    double chi2s = 0.0, chi2e = 0.0; int iters = 0;
    for (int it = 0; it < in->n_iter_segs; ++it) {
      double As[9] = { 0 }, Ae[9] = { 0 }, bs[3] = { 0 }, be[3] = { 0 }, ncs = 0.0, nce = 0.0;
      ++iters;
      for (int o = in->seg_obs_off[i]; o < in->seg_obs_off[i + 1]; ++o) {
        const int fr = in->seg_obs_frame[o];
        structopt_accumulate(&T[fr], Rm + 9 * fr, in->seg_obs_sf + 3 * o, sp, As, bs, &ncs);
        structopt_accumulate(&T[fr], Rm + 9 * fr, in->seg_obs_ef + 3 * o, ep, Ae, be, &nce);
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
    if (out->seg_spos) for (int k = 0; k < 3; ++k) out->seg_spos[3 * i + k] = sp[k];
    if (out->seg_epos) for (int k = 0; k < 3; ++k) out->seg_epos[3 * i + k] = ep[k];
    if (out->seg_iters) out->seg_iters[i] = iters;
  }
  free(T); free(Rm);
  return 0;
}

/* ============================================================================================ */
/* direct feature matching: Matcher::findMatchDirect (src/matcher.cpp:159-275) and callees      */
/* ============================================================================================ */
/* Summation orders of the Eigen fixed-size float expressions below are restated from Eigen 3.2 (the
 * release contemporary with the reference; no version is pinned, CMakeLists.txt:39): coefficient-based
 * products accumulate left to right, `.sum()` of 3 terms is a0 + (a1 + a2) (redux_novec_unroller).   */

/* float -> int conversion as x86 cvttss2si does it for NaN (what `int u_r = floor(u)` compiles to there);
 * written out so that the restatement has no undefined behaviour and the device can do the same */
static int f2i_trunc(float x) { return isnan(x) ? INT32_MIN : (int)x; }

/* [ext] vk::interpolateMat_8u (vikit/vision.h): bilinear lookup in float; the caller guarantees
 * 0 <= u < cols-1, 0 <= v < rows-1 (matcher.cpp:124) */
static float interpolate_mat_8u(const uint8_t* img, int stride, float u, float v) {
  const int x = (int)floorf(u), y = (int)floorf(v);
  const float subpix_x = u - x, subpix_y = v - y;
  const float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
  const float w01 = (1.0f - subpix_x) * subpix_y;
  const float w10 = subpix_x * (1.0f - subpix_y);
  const float w11 = 1.0f - w00 - w01 - w10;
  const uint8_t* ptr = img + (ptrdiff_t)y * stride + x;
  return w00 * ptr[0] + w01 * ptr[stride] + w10 * ptr[1] + w11 * ptr[stride + 1];
}

/* warp::getWarpMatrixAffine (matcher.cpp:44-71).  A is row-major {a00, a01, a10, a11}. */
static void warp_matrix_affine(const plsvo_pinhole* cam, const double px_ref[2], const double f_ref[3],
                               double depth_ref, const se3_t* T_cur_ref, int level_ref, double A[4]) {
  const int halfpatch_size = 5;
  const double xyz_ref[3] = { f_ref[0] * depth_ref, f_ref[1] * depth_ref, f_ref[2] * depth_ref };
  const double pdu[2] = { px_ref[0] + (double)halfpatch_size * (1 << level_ref), px_ref[1] + 0.0 * (1 << level_ref) };
  const double pdv[2] = { px_ref[0] + 0.0 * (1 << level_ref), px_ref[1] + (double)halfpatch_size * (1 << level_ref) };
  double xyz_du_ref[3], xyz_dv_ref[3];
  plsvo_oracle_cam2world(cam, pdu, xyz_du_ref);
  plsvo_oracle_cam2world(cam, pdv, xyz_dv_ref);
  const double su = xyz_ref[2] / xyz_du_ref[2], sv = xyz_ref[2] / xyz_dv_ref[2];
  for (int i = 0; i < 3; ++i) { xyz_du_ref[i] *= su; xyz_dv_ref[i] *= sv; }
  double c[3], px_cur[2], px_du[2], px_dv[2];
  se3_act(T_cur_ref, xyz_ref, c);    plsvo_oracle_world2cam(cam, c, px_cur);
  se3_act(T_cur_ref, xyz_du_ref, c); plsvo_oracle_world2cam(cam, c, px_du);
  se3_act(T_cur_ref, xyz_dv_ref, c); plsvo_oracle_world2cam(cam, c, px_dv);
  A[0] = (px_du[0] - px_cur[0]) / halfpatch_size; A[2] = (px_du[1] - px_cur[1]) / halfpatch_size;  /* col(0) */
  A[1] = (px_dv[0] - px_cur[0]) / halfpatch_size; A[3] = (px_dv[1] - px_cur[1]) / halfpatch_size;  /* col(1) */
}

/* warp::getBestSearchLevel (matcher.cpp:73-86) */
static int best_search_level(const double A[4], int max_level) {
  int search_level = 0;
  double D = A[0] * A[3] - A[2] * A[1];   /* [ext] Eigen 2x2 determinant */
  while (D > 3.0 && search_level < max_level) { search_level += 1; D *= 0.25; }
  return search_level;
}

/* warp::warpAffine (matcher.cpp:88-129); returns 0 when the inverse warp is NaN (the reference then leaves
 * the Matcher's previous patch in place, :96-100 -- state this restatement does not carry: such a candidate is
 * reported as not found) */
static int warp_affine(const double A_cur_ref[4], const uint8_t* img_ref, int cols, int rows, int stride,
                       const double px_ref[2], int level_ref, int search_level, int halfpatch_size, uint8_t* patch) {
  const int patch_size = halfpatch_size * 2;
  const double det = A_cur_ref[0] * A_cur_ref[3] - A_cur_ref[2] * A_cur_ref[1];
  const double invdet = 1.0 / det;                      /* [ext] Eigen compute_inverse_size2_helper */
  const float a00 = (float)(A_cur_ref[3] * invdet), a01 = (float)(-A_cur_ref[1] * invdet);
  const float a10 = (float)(-A_cur_ref[2] * invdet), a11 = (float)(A_cur_ref[0] * invdet);
  if (isnan(a00)) return 0;
  const float rx = (float)px_ref[0] / (float)(1 << level_ref), ry = (float)px_ref[1] / (float)(1 << level_ref);
  uint8_t* patch_ptr = patch;
  for (int y = 0; y < patch_size; ++y) {
    for (int x = 0; x < patch_size; ++x, ++patch_ptr) {
      float ppx = (float)(x - halfpatch_size), ppy = (float)(y - halfpatch_size);
      ppx *= (float)(1 << search_level); ppy *= (float)(1 << search_level);
      const float px0 = (a00 * ppx + a01 * ppy) + rx;
      const float px1 = (a10 * ppx + a11 * ppy) + ry;
      if (px0 < 0 || px1 < 0 || px0 >= cols - 1 || px1 >= rows - 1) *patch_ptr = 0;
      else *patch_ptr = (uint8_t)interpolate_mat_8u(img_ref, stride, px0, px1);
    }
  }
  return 1;
}

enum { M_PATCH = 8, M_HALF = 4, M_AREA = 64, M_STEP = 10 };

/* feature_alignment::align2D (feature_alignment.cpp:160-290), scalar path */
static int align_2d(const uint8_t* cur_img, int cols, int rows, int cur_step, const uint8_t* ref_patch_with_border,
                    const uint8_t* ref_patch, int n_iter, double cur_px_estimate[2], int* iters) {
  int converged = 0;
  float ref_patch_dx[M_AREA], ref_patch_dy[M_AREA];
  float H[3][3] = { { 0 } };
  const int ref_step = M_STEP;
  float *it_dx = ref_patch_dx, *it_dy = ref_patch_dy;
  for (int y = 0; y < M_PATCH; ++y) {
    const uint8_t* it = ref_patch_with_border + (y + 1) * ref_step + 1;
    for (int x = 0; x < M_PATCH; ++x, ++it, ++it_dx, ++it_dy) {
      float J[3];
      J[0] = (float)(0.5 * (it[1] - it[-1]));
      J[1] = (float)(0.5 * (it[ref_step] - it[-ref_step]));
      J[2] = 1;
      *it_dx = J[0]; *it_dy = J[1];
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) H[i][j] += J[i] * J[j];
    }
  }
  /* [ext] Eigen::Matrix3f::inverse(): compute_inverse_size3_helper (cofactors, det = col0 . cofactors_col0) */
#define COF(i, j) (H[((i) + 1) % 3][((j) + 1) % 3] * H[((i) + 2) % 3][((j) + 2) % 3] - H[((i) + 1) % 3][((j) + 2) % 3] * H[((i) + 2) % 3][((j) + 1) % 3])
  float Hinv[3][3];
  {
    const float c00 = COF(0, 0), c10 = COF(1, 0), c20 = COF(2, 0);
    const float det = c00 * H[0][0] + (c10 * H[1][0] + c20 * H[2][0]);
    const float invdet = 1.0f / det;
    Hinv[0][0] = c00 * invdet; Hinv[0][1] = c10 * invdet; Hinv[0][2] = c20 * invdet;
    Hinv[1][0] = COF(0, 1) * invdet; Hinv[1][1] = COF(1, 1) * invdet; Hinv[1][2] = COF(2, 1) * invdet;
    Hinv[2][0] = COF(0, 2) * invdet; Hinv[2][1] = COF(1, 2) * invdet; Hinv[2][2] = COF(2, 2) * invdet;
  }
#undef COF
  float mean_diff = 0;
  float u = (float)cur_px_estimate[0], v = (float)cur_px_estimate[1];
  const float min_update_squared = (float)(0.03 * 0.03);
  float update[3] = { 0, 0, 0 };
  int iter = 0;
  for (; iter < n_iter; ++iter) {
    /* Patch::setPosition / isInFrame(halfsize) / computeInterpWeights / setRoi (feature.cpp:189-218) */
    const float u_ref = (float)cur_px_estimate[0], v_ref = (float)cur_px_estimate[1];
    const int u_ref_i = f2i_trunc(floorf(u_ref)), v_ref_i = f2i_trunc(floorf(v_ref));
    if (u_ref_i < M_HALF || v_ref_i < M_HALF || u_ref_i >= cols - M_HALF || v_ref_i >= rows - M_HALF) break;
    const float subpix_u_ref = u_ref - u_ref_i, subpix_v_ref = v_ref - v_ref_i;
    const float wTL = (float)((1.0 - subpix_u_ref) * (1.0 - subpix_v_ref));
    const float wTR = (float)(subpix_u_ref * (1.0 - subpix_v_ref));
    const float wBL = (float)((1.0 - subpix_u_ref) * subpix_v_ref);
    const float wBR = subpix_u_ref * subpix_v_ref;
    const uint8_t* it_ref = ref_patch;
    const float *it_ref_dx = ref_patch_dx, *it_ref_dy = ref_patch_dy;
    float Jres[3] = { 0, 0, 0 };
    for (int y = 0; y < M_PATCH; ++y) {
      const uint8_t* ptr = cur_img + (ptrdiff_t)(v_ref_i - M_HALF + y) * cur_step + (u_ref_i - M_HALF);
      for (int x = 0; x < M_PATCH; ++x, ++ptr, ++it_ref, ++it_ref_dx, ++it_ref_dy) {
        const float search_pixel = wTL * ptr[0] + wTR * ptr[1] + wBL * ptr[cur_step] + wBR * ptr[cur_step + 1];
        const float res = search_pixel - *it_ref + mean_diff;
        Jres[0] -= res * (*it_ref_dx);
        Jres[1] -= res * (*it_ref_dy);
        Jres[2] -= res;
      }
    }
    for (int i = 0; i < 3; ++i) update[i] = Hinv[i][0] * Jres[0] + Hinv[i][1] * Jres[1] + Hinv[i][2] * Jres[2];
    u += update[0]; v += update[1];
    cur_px_estimate[0] = u; cur_px_estimate[1] = v;
    mean_diff += update[2];
    if (update[0] * update[0] + update[1] * update[1] < min_update_squared) { converged = 1; ++iter; break; }
  }
  *iters = iter;
  cur_px_estimate[0] = u; cur_px_estimate[1] = v;
  return converged;
}

/* feature_alignment::align1D (feature_alignment.cpp:41-158) */
static int align_1d(const uint8_t* cur_img, int cols, int rows, int cur_step, const float dir[2],
                    const uint8_t* ref_patch_with_border, const uint8_t* ref_patch, int n_iter,
                    double cur_px_estimate[2], int* iters) {
  int converged = 0;
  float ref_patch_dv[M_AREA];
  float H[2][2] = { { 0 } };
  const int ref_step = M_STEP;
  float* it_dv = ref_patch_dv;
  for (int y = 0; y < M_PATCH; ++y) {
    const uint8_t* it = ref_patch_with_border + (y + 1) * ref_step + 1;
    for (int x = 0; x < M_PATCH; ++x, ++it, ++it_dv) {
      float J[2];
      J[0] = (float)(0.5 * (dir[0] * (it[1] - it[-1]) + dir[1] * (it[ref_step] - it[-ref_step])));
      J[1] = 1;
      *it_dv = J[0];
      for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) H[i][j] += J[i] * J[j];
    }
  }
  float Hinv[2][2];
  {
    const float det = H[0][0] * H[1][1] - H[1][0] * H[0][1];
    const float invdet = 1.0f / det;                  /* [ext] Eigen compute_inverse_size2_helper */
    Hinv[0][0] = H[1][1] * invdet; Hinv[1][0] = -H[1][0] * invdet;
    Hinv[0][1] = -H[0][1] * invdet; Hinv[1][1] = H[0][0] * invdet;
  }
  float mean_diff = 0;
  float u = (float)cur_px_estimate[0], v = (float)cur_px_estimate[1];
  const float min_update_squared = (float)(0.03 * 0.03);
  float chi2 = 0;
  float update[2] = { 0, 0 };
  int iter = 0;
  for (; iter < n_iter; ++iter) {
    const int u_r = f2i_trunc(floorf(u)), v_r = f2i_trunc(floorf(v));
    if (u_r < M_HALF || v_r < M_HALF || u_r >= cols - M_HALF || v_r >= rows - M_HALF) break;
    if (isnan(u) || isnan(v)) { *iters = iter; return 0; }   /* :94-95 returns without writing cur_px_estimate */
    const float subpix_x = u - u_r, subpix_y = v - v_r;
    const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
    const float wTR = (float)(subpix_x * (1.0 - subpix_y));
    const float wBL = (float)((1.0 - subpix_x) * subpix_y);
    const float wBR = subpix_x * subpix_y;
    const uint8_t* it_ref = ref_patch;
    const float* it_ref_dv = ref_patch_dv;
    float new_chi2 = 0.0f;
    float Jres[2] = { 0, 0 };
    for (int y = 0; y < M_PATCH; ++y) {
      const uint8_t* it = cur_img + (ptrdiff_t)(v_r + y - M_HALF) * cur_step + u_r - M_HALF;
      for (int x = 0; x < M_PATCH; ++x, ++it, ++it_ref, ++it_ref_dv) {
        const float search_pixel = wTL * it[0] + wTR * it[1] + wBL * it[cur_step] + wBR * it[cur_step + 1];
        const float res = search_pixel - *it_ref + mean_diff;
        Jres[0] -= res * (*it_ref_dv);
        Jres[1] -= res;
        new_chi2 += res * res;
      }
    }
    if (iter > 0 && new_chi2 > chi2) { u -= update[0]; v -= update[1]; ++iter; break; }   /* :123-131, as written */
    chi2 = new_chi2;
    update[0] = Hinv[0][0] * Jres[0] + Hinv[0][1] * Jres[1];
    update[1] = Hinv[1][0] * Jres[0] + Hinv[1][1] * Jres[1];
    u += update[0] * dir[0];
    v += update[0] * dir[1];
    mean_diff += update[1];
    if (update[0] * update[0] + update[1] * update[1] < min_update_squared) { converged = 1; ++iter; break; }
  }
  *iters = iter;
  cur_px_estimate[0] = u; cur_px_estimate[1] = v;
  return converged;
}

/* Matcher::findMatchDirect for one candidate: a point (matcher.cpp:159-207) or one end point of a line
 * segment (:209-230 + :253-274).  The closest-view observation has been chosen by the caller (:165, :239). */
int plsvo_oracle_match_direct(const plsvo_match_in* in, const plsvo_oracle_pyr* frames, plsvo_match_out* out) {
  if (!in || !out || in->n < 0 || in->n_frames <= 0 || !frames) return PLSVO_E_INVALID;
  const int halfpatch_size_ = 4;
  for (int i = 0; i < in->n; ++i) {
    const int rf = in->ref_frame[i], cf = in->cur_frame[i], level = in->ref_level[i];
    double px_cur[2] = { in->px_cur[2 * i], in->px_cur[2 * i + 1] };
    int found = 0, search_level = -1, iters = 0;
    const double* rpx = in->ref_px + 2 * i;
    /* :168-170  px.cast<int>() / (1<<level): truncation, then integer division */
    if (cam_is_in_frame(&in->cam, (int)rpx[0] / (1 << level), (int)rpx[1] / (1 << level), halfpatch_size_ + 2, level)) {
      const se3_t T_ref = se3_load(in->frame_T + 7 * rf), T_cur = se3_load(in->frame_T + 7 * cf);
      const se3_t T_ref_inv = se3_inv(&T_ref);
      const se3_t T_cur_ref = se3_mul(&T_cur, &T_ref_inv);
      /* Frame::pos() = T_f_w_.inverse().translation() (frame.h:106); depth = |pos_ref - pos| */
      const double d[3] = { T_ref_inv.t[0] - in->pos[3 * i], T_ref_inv.t[1] - in->pos[3 * i + 1], T_ref_inv.t[2] - in->pos[3 * i + 2] };
      const double depth_ref = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      double A[4];
      warp_matrix_affine(&in->cam, rpx, in->ref_f + 3 * i, depth_ref, &T_cur_ref, level, A);
      search_level = best_search_level(A, in->n_pyr_levels - 1);
      uint8_t patch_with_border[M_STEP * M_STEP], patch[M_AREA];
      const plsvo_oracle_pyr* rp = &frames[rf];
      const plsvo_oracle_pyr* cp = &frames[cf];
      if (warp_affine(A, rp->img[level], rp->width[level], rp->height[level], rp->stride[level], rpx, level, search_level,
                      halfpatch_size_ + 1, patch_with_border)) {
        for (int y = 1; y < M_PATCH + 1; ++y)          /* createPatchFromPatchWithBorder :148-157 */
          for (int x = 0; x < M_PATCH; ++x) patch[(y - 1) * M_PATCH + x] = patch_with_border[y * M_STEP + 1 + x];
        double px_scaled[2] = { px_cur[0] / (1 << search_level), px_cur[1] / (1 << search_level) };
        if (in->ref_type[i] == PLSVO_FTR_EDGELET) {
          double dc[2] = { A[0] * in->ref_grad[2 * i] + A[1] * in->ref_grad[2 * i + 1], A[2] * in->ref_grad[2 * i] + A[3] * in->ref_grad[2 * i + 1] };
          const double nrm = sqrt(dc[0] * dc[0] + dc[1] * dc[1]);
          dc[0] /= nrm; dc[1] /= nrm;
          const float dir[2] = { (float)dc[0], (float)dc[1] };
          found = align_1d(cp->img[search_level], cp->width[search_level], cp->height[search_level], cp->stride[search_level], dir,
                           patch_with_border, patch, in->align_max_iter, px_scaled, &iters);
        } else {
          found = align_2d(cp->img[search_level], cp->width[search_level], cp->height[search_level], cp->stride[search_level],
                           patch_with_border, patch, in->align_max_iter, px_scaled, &iters);
        }
        px_cur[0] = px_scaled[0] * (1 << search_level); px_cur[1] = px_scaled[1] * (1 << search_level);
      }
    }
    if (out->px_cur) { out->px_cur[2 * i] = px_cur[0]; out->px_cur[2 * i + 1] = px_cur[1]; }
    if (out->found) out->found[i] = (uint8_t)found;
    if (out->search_level) out->search_level[i] = search_level;
    if (out->n_iter) out->n_iter[i] = iters;
  }
  return PLSVO_OK;
}

/* Reprojector::reproject for points and segment end points (src/reprojector.cpp:389-423) */
int plsvo_oracle_reproject(const plsvo_reproject_in* in, plsvo_reproject_out* out) {
  if (!in || !out || in->n < 0 || in->cell_size <= 0) return PLSVO_E_INVALID;
  for (int i = 0; i < in->n; ++i) {
    const se3_t T = se3_load(in->frame_T + 7 * in->frame[i]);
    double c[3], px[2];
    se3_act(&T, in->pos + 3 * i, c);                  /* Frame::w2c: cam_->world2cam(T_f_w_ * xyz_w), frame.h:113 */
    plsvo_oracle_world2cam(&in->cam, c, px);
    int cell = -1;
    /* isInFrame(cur_px.cast<int>(), 8): level-0 overload, [ext] vk::AbstractCamera.  A NaN / out-of-range cast is
     * undefined in C; the device and this restatement both treat it as "not in frame". */
    if (px[0] == px[0] && px[1] == px[1] && fabs(px[0]) < 1e9 && fabs(px[1]) < 1e9) {
      const int ox = (int)px[0], oy = (int)px[1];
      if (ox >= in->boundary && ox < in->cam.width - in->boundary && oy >= in->boundary && oy < in->cam.height - in->boundary)
        cell = (int)(px[1] / in->cell_size) * in->grid_n_cols + (int)(px[0] / in->cell_size);   /* :397-398 */
    }
    if (out->px) { out->px[2 * i] = px[0]; out->px[2 * i + 1] = px[1]; }
    if (out->cell) out->cell[i] = cell;
  }
  return PLSVO_OK;
}

/* trajectory record of app/run_pipeline.cpp:425-451 */
int plsvo_oracle_trajectory_record(const double T_f_w[7], const double cov[36], double out7[7]) {
  int skip_frame = 0;
  for (int i = 0; i < 36; ++i) if (!((1.e-16 < fabs(cov[i])) && (fabs(cov[i]) < 1.e+16))) skip_frame = 1;
  const se3_t T = se3_load(T_f_w);
  const se3_t W = se3_inv(&T);
  if ((W.t[0] == 0. && W.t[1] == 0. && W.t[2] == 0.) && (W.q.x == -0. && W.q.y == -0. && W.q.z == -0. && W.q.w == 1.)) skip_frame = 1;
  out7[0] = W.t[0]; out7[1] = W.t[1]; out7[2] = W.t[2]; out7[3] = W.q.x; out7[4] = W.q.y; out7[5] = W.q.z; out7[6] = W.q.w;
  return skip_frame ? 0 : 1;
}

/* ============================================================================================ */
/* depth-filter seed update: DepthFilter::updatePointSeeds / updateLineSeeds (src/depth_filter.cpp:270-471) */
/* ============================================================================================ */

/* [ext] vk::patch_score::ZMSSD<4> (vikit/patch_score.h): zero-mean SSD of two 8x8 u8 patches, all integer */
typedef struct { const uint8_t* ref_patch; int sumA, sumAA; } zmssd_t;
static void zmssd_init(zmssd_t* z, const uint8_t* ref_patch) {
  uint32_t sumA_uint = 0, sumAA_uint = 0;
  for (int r = 0; r < M_AREA; ++r) { const uint8_t n = ref_patch[r]; sumA_uint += n; sumAA_uint += n * n; }
  z->ref_patch = ref_patch; z->sumA = (int)sumA_uint; z->sumAA = (int)sumAA_uint;
}
static int zmssd_threshold(void) { return 2000 * M_AREA; }
static int zmssd_score(const zmssd_t* z, const uint8_t* cur_patch, int stride) {
  uint32_t sumB_uint = 0, sumBB_uint = 0, sumAB_uint = 0;
  for (int y = 0, r = 0; y < M_PATCH; ++y) {
    const uint8_t* cur_patch_ptr = cur_patch + (ptrdiff_t)y * stride;
    for (int x = 0; x < M_PATCH; ++x, ++r) {
      const uint8_t cur_px = cur_patch_ptr[x];
      sumB_uint += cur_px; sumBB_uint += cur_px * cur_px; sumAB_uint += cur_px * z->ref_patch[r];
    }
  }
  const int sumB = (int)sumB_uint, sumBB = (int)sumBB_uint, sumAB = (int)sumAB_uint;
  return z->sumAA - 2 * sumAB + sumBB - (z->sumA * z->sumA - 2 * z->sumA * sumB + sumB * sumB) / M_AREA;
}

/* depthFromTriangulation (src/matcher.cpp:133-146) */
static int depth_from_triangulation(const se3_t* T_search_ref, const double f_ref[3], const double f_cur[3], double* depth) {
  double R[9], c0[3];
  quat_to_matrix(T_search_ref->q, R);                       /* rotation_matrix() * f_ref */
  for (int i = 0; i < 3; ++i) c0[i] = R[3 * i] * f_ref[0] + R[3 * i + 1] * f_ref[1] + R[3 * i + 2] * f_ref[2];
  /* AtA = A^T A with A = [c0, f_cur] */
  const double a00 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2];
  const double a01 = c0[0] * f_cur[0] + c0[1] * f_cur[1] + c0[2] * f_cur[2];
  const double a11 = f_cur[0] * f_cur[0] + f_cur[1] * f_cur[1] + f_cur[2] * f_cur[2];
  const double a10 = a01;
  const double det = a00 * a11 - a10 * a01;
  if (det < 0.000001) return 0;
  const double invdet = 1.0 / det;
  /* (-AtA.inverse()) * A^T * t : row 0 of the 2x3 product, then the dot product with the translation */
  const double i00 = -(a11 * invdet), i01 = -(-a01 * invdet);
  const double m0 = i00 * c0[0] + i01 * f_cur[0], m1 = i00 * c0[1] + i01 * f_cur[1], m2 = i00 * c0[2] + i01 * f_cur[2];
  const double d0 = m0 * T_search_ref->t[0] + m1 * T_search_ref->t[1] + m2 * T_search_ref->t[2];
  *depth = fabs(d0);
  return 1;
}

/* Matcher::findEpipolarMatchDirect (src/matcher.cpp:277-420; segment_endpoint == 0) and
 * findEpipolarMatchDirectSegmentEndpoint (:422-586; segment_endpoint == 1).  Returns 1 on success. */
typedef struct { double px_cur[2]; int search_level; int reject; int n_evals; } epi_out_t;
static int find_epipolar_match_direct(const plsvo_seeds_in* in, const plsvo_oracle_pyr* frames, int rf, int cf, const double px_ref[2],
                                      const double f_ref[3], int level, int type, const double grad[2], double d_estimate,
                                      double d_min, double d_max, int segment_endpoint, double* depth, epi_out_t* eo) {
  const int halfpatch_size_ = 4, patch_size_ = 8;
  eo->px_cur[0] = eo->px_cur[1] = 0.0; eo->search_level = -1; eo->reject = 0; eo->n_evals = 0;
  const se3_t T_ref = se3_load(in->frame_T + 7 * rf), T_cur = se3_load(in->frame_T + 7 * cf);
  const se3_t T_ref_inv = se3_inv(&T_ref);
  const se3_t T_cur_ref = se3_mul(&T_cur, &T_ref_inv);
  int zmssd_best = zmssd_threshold();
  double uv_best[2] = { 0.0, 0.0 };
  if (segment_endpoint && (isnan(d_min) || isnan(d_max))) { eo->reject = 1; return 0; }         /* :436-440 */
  /* epipolar segment on the unit plane: A = project2d(T * (f*d_min)), B = project2d(T * (f*d_max)) */
  double pa[3] = { f_ref[0] * d_min, f_ref[1] * d_min, f_ref[2] * d_min }, pb[3] = { f_ref[0] * d_max, f_ref[1] * d_max, f_ref[2] * d_max };
  double ca[3], cb[3], A[2], B[2];
  se3_act(&T_cur_ref, pa, ca); se3_act(&T_cur_ref, pb, cb);
  project2d(ca, A); project2d(cb, B);
  const double epi_dir[2] = { A[0] - B[0], A[1] - B[1] };
  double Aw[4];
  warp_matrix_affine(&in->cam, px_ref, f_ref, d_estimate, &T_cur_ref, level, Aw);
  if (!segment_endpoint && type == PLSVO_FTR_EDGELET && in->edgelet_filtering) {                 /* :303-311 */
    double g[2] = { Aw[0] * grad[0] + Aw[1] * grad[1], Aw[2] * grad[0] + Aw[3] * grad[1] };
    const double gn = sqrt(g[0] * g[0] + g[1] * g[1]);
    g[0] /= gn; g[1] /= gn;
    const double en = sqrt(epi_dir[0] * epi_dir[0] + epi_dir[1] * epi_dir[1]);
    const double cosangle = fabs(g[0] * (epi_dir[0] / en) + g[1] * (epi_dir[1] / en));
    if (cosangle < in->edgelet_max_angle) { eo->reject = 1; return 0; }
  }
  const int search_level = best_search_level(Aw, in->n_pyr_levels - 1);
  eo->search_level = search_level;
  double uvA[3] = { A[0], A[1], 1.0 }, uvB[3] = { B[0], B[1], 1.0 }, px_A[2], px_B[2];
  plsvo_oracle_world2cam(&in->cam, uvA, px_A);          /* world2cam(Vector2d uv) = (fx*u + cx, fy*v + cy) */
  plsvo_oracle_world2cam(&in->cam, uvB, px_B);
  const double dA[2] = { px_A[0] - px_B[0], px_A[1] - px_B[1] };
  const double epi_length = sqrt(dA[0] * dA[0] + dA[1] * dA[1]) / (1 << search_level);
  /* :480-484 (segments) reject NaN/inf before the warp; for points the reference runs into the step loop with an
   * undefined n_steps (:347) -- both are reported as "no match" here */
  if (isnan(epi_length) || isinf(epi_length)) { if (segment_endpoint) eo->reject = 1; return 0; }
  uint8_t patch_with_border[M_STEP * M_STEP], patch[M_AREA];
  const plsvo_oracle_pyr* rp = &frames[rf];
  const plsvo_oracle_pyr* cp = &frames[cf];
  if (!warp_affine(Aw, rp->img[level], rp->width[level], rp->height[level], rp->stride[level], px_ref, level, search_level,
                   halfpatch_size_ + 1, patch_with_border)) return 0;
  for (int y = 1; y < M_PATCH + 1; ++y) for (int x = 0; x < M_PATCH; ++x) patch[(y - 1) * M_PATCH + x] = patch_with_border[y * M_STEP + 1 + x];
  const uint8_t* cimg = cp->img[search_level];
  const int ccols = cp->width[search_level], crows = cp->height[search_level], cstride = cp->stride[search_level];
  const double sc = (double)(1 << search_level);
  int iters = 0;
  if (epi_length < 2.0) {                                                                        /* :325-344 */
    double px_scaled[2] = { ((px_A[0] + px_B[0]) / 2.0) / sc, ((px_A[1] + px_B[1]) / 2.0) / sc };
    eo->px_cur[0] = (px_A[0] + px_B[0]) / 2.0; eo->px_cur[1] = (px_A[1] + px_B[1]) / 2.0;
    if (align_2d(cimg, ccols, crows, cstride, patch_with_border, patch, in->align_max_iter, px_scaled, &iters)) {
      eo->px_cur[0] = px_scaled[0] * sc; eo->px_cur[1] = px_scaled[1] * sc;
      double f_cur[3];
      plsvo_oracle_cam2world(&in->cam, eo->px_cur, f_cur);
      if (depth_from_triangulation(&T_cur_ref, f_ref, f_cur, depth)) return 1;
    }
    return 0;
  }
  size_t n_steps = (size_t)(epi_length / 0.7);                                                    /* :347 */
  const double step[2] = { epi_dir[0] / (double)n_steps, epi_dir[1] / (double)n_steps };
  if (n_steps > (size_t)in->max_epi_search_steps) return 0;                                       /* :350-355 */
  zmssd_t score;
  zmssd_init(&score, patch);
  double uv[2] = { B[0] - step[0], B[1] - step[1] };
  int last_x = 0, last_y = 0;
  ++n_steps;
  for (size_t i = 0; i < n_steps; ++i, uv[0] += step[0], uv[1] += step[1]) {
    const double px0 = in->cam.fx * uv[0] + in->cam.cx, px1 = in->cam.fy * uv[1] + in->cam.cy;
    const double q0 = px0 / (1 << search_level) + 0.5, q1 = px1 / (1 << search_level) + 0.5;
    if (!(fabs(q0) < 1e9 && fabs(q1) < 1e9)) continue;       /* the int cast of NaN / huge values is undefined: treated as out of frame */
    const int pxi0 = (int)q0, pxi1 = (int)q1;
    if (pxi0 == last_x && pxi1 == last_y) continue;
    last_x = pxi0; last_y = pxi1;
    if (!cam_is_in_frame(&in->cam, pxi0, pxi1, patch_size_, search_level)) continue;
    /* the reference indexes with img.cols (:381-383); the pyramids here are tight (cols == step) */
    const uint8_t* cur_patch_ptr = cimg + (ptrdiff_t)(pxi1 - halfpatch_size_) * cstride + (pxi0 - halfpatch_size_);
    const int z = zmssd_score(&score, cur_patch_ptr, cstride);
    eo->n_evals += 1;
    if (z < zmssd_best) { zmssd_best = z; uv_best[0] = uv[0]; uv_best[1] = uv[1]; }
  }
  if (zmssd_best < zmssd_threshold()) {                                                           /* :392-412 subpix_refinement = true */
    eo->px_cur[0] = in->cam.fx * uv_best[0] + in->cam.cx; eo->px_cur[1] = in->cam.fy * uv_best[1] + in->cam.cy;
    double px_scaled[2] = { eo->px_cur[0] / sc, eo->px_cur[1] / sc };
    if (align_2d(cimg, ccols, crows, cstride, patch_with_border, patch, in->align_max_iter, px_scaled, &iters)) {
      eo->px_cur[0] = px_scaled[0] * sc; eo->px_cur[1] = px_scaled[1] * sc;
      double f_cur[3];
      plsvo_oracle_cam2world(&in->cam, eo->px_cur, f_cur);
      if (depth_from_triangulation(&T_cur_ref, f_ref, f_cur, depth)) return 1;
    }
    return 0;
  }
  return 0;
}

/* DepthFilter::computeTau (src/depth_filter.cpp:568-584) */
static double compute_tau(const se3_t* T_ref_cur, const double f[3], double z, double px_error_angle) {
  const double* t = T_ref_cur->t;
  const double a[3] = { f[0] * z - t[0], f[1] * z - t[1], f[2] * z - t[2] };
  const double t_norm = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
  const double a_norm = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  const double alpha = acos((f[0] * t[0] + f[1] * t[1] + f[2] * t[2]) / t_norm);
  const double beta = acos((a[0] * -t[0] + a[1] * -t[1] + a[2] * -t[2]) / (t_norm * a_norm));
  const double beta_plus = beta + px_error_angle;
  const double gamma_plus = 3.14159265 - alpha - beta_plus;    /* PI as defined in include/plsvo/global.h */
  const double z_plus = t_norm * sin(beta_plus) / sin(gamma_plus);
  return z_plus - z;
}

/* [ext] boost::math::pdf(normal_distribution<float>(mean, sd), x) */
static float normal_pdf_f(float mean, float sd, float x) {
  if (isinf(x)) return 0.0f;
  float exponent = x - mean;
  exponent *= -exponent;
  exponent /= 2 * sd * sd;
  float result = expf(exponent);
  result /= sd * sqrtf(2 * 3.14159265358979323846f);
  return result;
}

/* one end of the Vogiatzis-Hernandez update: the common text of updatePointSeed (:489-512) and of the two halves of
 * updateLineSeed (:514-552).  Returns f and e through pointers; mu/sigma2 are updated in place. */
static void seed_update_end(float x, float tau2, float a, float b, float z_range, float* mu, float* sigma2, float norm_scale, float* f_out, float* e_out) {
  const float pdf = normal_pdf_f(*mu, norm_scale, x);
  float s2 = 1. / (1. / *sigma2 + 1. / tau2);
  float m = s2 * (*mu / *sigma2 + x / tau2);
  float C1 = a / (a + b) * pdf;
  float C2 = b / (a + b) * 1. / z_range;
  float normalization_constant = C1 + C2;
  C1 /= normalization_constant;
  C2 /= normalization_constant;
  float f = C1 * (a + 1.) / (a + b + 1.) + C2 * a / (a + b + 1.);
  float e = C1 * (a + 1.) * (a + 2.) / ((a + b + 1.) * (a + b + 2.)) + C2 * a * (a + 1.0f) / ((a + b + 1.0f) * (a + b + 2.0f));
  float mu_new = C1 * m + C2 * *mu;
  *sigma2 = C1 * (s2 + m * m) + C2 * (*sigma2 + *mu * *mu) - mu_new * mu_new;
  *mu = mu_new;
  *f_out = f; *e_out = e;
}

int plsvo_oracle_update_seeds(const plsvo_seeds_in* in, const plsvo_oracle_pyr* frames, plsvo_seeds_out* out) {
  if (!in || !out || !frames || in->n_pt < 0 || in->n_seg < 0 || in->n_frames <= 0) return PLSVO_E_INVALID;
  const double focal_length = fabs(in->cam.fx);                                  /* errorMultiplier2() */
  const double px_error_angle = atan(in->px_noise / (2.0 * focal_length)) * 2.0; /* :279-280 */
  for (int i = 0; i < in->n_pt; ++i) {
    float a = in->pt_a[i], b = in->pt_b[i], mu = in->pt_mu[i], sigma2 = in->pt_sigma2[i];
    const float z_range = in->pt_z_range[i];
    int status = PLSVO_SEED_NOT_VISIBLE;
    double xyz_world[3] = { 0, 0, 0 }, z = 0.0;
    epi_out_t eo; eo.px_cur[0] = eo.px_cur[1] = 0.0;
    const int rf = in->pt_ref_frame[i], cf = in->pt_cur_frame[i];
    const double* f = in->pt_f + 3 * i;
    const se3_t T_ref = se3_load(in->frame_T + 7 * rf), T_cur = se3_load(in->frame_T + 7 * cf);
    const se3_t T_cur_inv = se3_inv(&T_cur);
    const se3_t T_ref_cur = se3_mul(&T_ref, &T_cur_inv);                         /* :295 */
    const se3_t T_cur_ref = se3_inv(&T_ref_cur);
    const double s = 1.0 / mu;
    const double p[3] = { s * f[0], s * f[1], s * f[2] };
    double xyz_f[3], px[2];
    se3_act(&T_cur_ref, p, xyz_f);                                               /* :296 */
    int visible = !(xyz_f[2] < 0.0);
    if (visible) {
      plsvo_oracle_world2cam(&in->cam, xyz_f, px);
      visible = px[0] == px[0] && px[1] == px[1] && fabs(px[0]) < 1e9 && fabs(px[1]) < 1e9 && cam_is_in_frame(&in->cam, (int)px[0], (int)px[1], 0, 0);
    }
    if (visible) {
      const float z_inv_min = mu + sqrtf(sigma2);
      const float t_ = mu - sqrtf(sigma2);
      const float z_inv_max = t_ > 0.00000001f ? t_ : 0.00000001f;               /* max(a, b) = (a < b) ? b : a */
      const double g0[2] = { 0, 0 };
      if (!find_epipolar_match_direct(in, frames, rf, cf, in->pt_px + 2 * i, f, in->pt_level[i], in->pt_type[i],
                                      in->pt_grad ? in->pt_grad + 2 * i : g0, 1.0 / mu, 1.0 / z_inv_min, 1.0 / z_inv_max, 0, &z, &eo)) {
        b += 1.0f;                                                               /* it->b++ :315 */
        status = PLSVO_SEED_NO_MATCH;
      } else {
        const double tau = compute_tau(&T_ref_cur, f, z, px_error_angle);
        const double zm = z - tau;
        const double tau_inverse = 0.5 * (1.0 / (0.0000001 < zm ? zm : 0.0000001) - 1.0 / (z + tau));
        /* updatePointSeed(1./z, tau_inverse*tau_inverse, &*it) :489-512 */
        const float x = (float)(1. / z), tau2 = (float)(tau_inverse * tau_inverse);
        const float norm_scale = sqrtf(sigma2 + tau2);
        if (!isnan(norm_scale)) {
          float fq, eq;
          seed_update_end(x, tau2, a, b, z_range, &mu, &sigma2, norm_scale, &fq, &eq);
          a = (eq - fq) / (fq - eq / fq);
          b = a * (1.0f - fq) / fq;
        }
        status = PLSVO_SEED_UPDATED;
        if (sqrtf(sigma2) < z_range / in->convergence_sigma2_thresh) {            /* :334 */
          const double sw = 1.0 / mu;
          const double pw[3] = { f[0] * sw, f[1] * sw, f[2] * sw };
          const se3_t T_ref_inv = se3_inv(&T_ref);                              /* :337 */
          se3_act(&T_ref_inv, pw, xyz_world);
          status = PLSVO_SEED_CONVERGED;
        } else if (isnan(z_inv_min)) {
          status = PLSVO_SEED_NAN;
        }
      }
    }
    if (out->pt_status) out->pt_status[i] = status;
    if (out->pt_a) out->pt_a[i] = a;
    if (out->pt_b) out->pt_b[i] = b;
    if (out->pt_mu) out->pt_mu[i] = mu;
    if (out->pt_sigma2) out->pt_sigma2[i] = sigma2;
    if (out->pt_xyz_world) for (int k = 0; k < 3; ++k) out->pt_xyz_world[3 * i + k] = xyz_world[k];
    if (out->pt_px_cur) { out->pt_px_cur[2 * i] = eo.px_cur[0]; out->pt_px_cur[2 * i + 1] = eo.px_cur[1]; }
    if (out->pt_depth) out->pt_depth[i] = z;
  }
  /* ---- line seeds: updateLineSeeds :367-471 ---- */
  for (int i = 0; i < in->n_seg; ++i) {
    float a = in->seg_a[i], b = in->seg_b[i], mu_s = in->seg_mu_s[i], mu_e = in->seg_mu_e[i];
    float sigma2_s = in->seg_sigma2_s[i], sigma2_e = in->seg_sigma2_e[i];
    const float z_range_s = in->seg_z_range_s[i], z_range_e = in->seg_z_range_e[i];
    int status = PLSVO_SEED_NOT_VISIBLE;
    double xw_s[3] = { 0, 0, 0 }, xw_e[3] = { 0, 0, 0 }, z_s = 0.0, z_e = 0.0;
    const int rf = in->seg_ref_frame[i], cf = in->seg_cur_frame[i];
    const double* sf = in->seg_sf + 3 * i;
    const double* ef = in->seg_ef + 3 * i;
    const se3_t T_ref = se3_load(in->frame_T + 7 * rf), T_cur = se3_load(in->frame_T + 7 * cf);
    const se3_t T_cur_inv = se3_inv(&T_cur);
    const se3_t T_ref_cur = se3_mul(&T_ref, &T_cur_inv);
    const se3_t T_cur_ref = se3_inv(&T_ref_cur);
    const double ss = 1.0 / mu_s, se = 1.0 / mu_e;
    const double ps[3] = { ss * sf[0], ss * sf[1], ss * sf[2] }, pe[3] = { se * ef[0], se * ef[1], se * ef[2] };
    double xs[3], xe[3], pxs[2], pxe[2];
    se3_act(&T_cur_ref, ps, xs); se3_act(&T_cur_ref, pe, xe);                    /* :393-394 */
    int visible = !(xs[2] < 0.0 || xe[2] < 0.0);
    if (visible) {
      plsvo_oracle_world2cam(&in->cam, xs, pxs); plsvo_oracle_world2cam(&in->cam, xe, pxe);
      visible = pxs[0] == pxs[0] && pxs[1] == pxs[1] && fabs(pxs[0]) < 1e9 && fabs(pxs[1]) < 1e9 && cam_is_in_frame(&in->cam, (int)pxs[0], (int)pxs[1], 0, 0) &&
                pxe[0] == pxe[0] && pxe[1] == pxe[1] && fabs(pxe[0]) < 1e9 && fabs(pxe[1]) < 1e9 && cam_is_in_frame(&in->cam, (int)pxe[0], (int)pxe[1], 0, 0);
    }
    if (visible) {
      const float z_inv_min_s = mu_s + sqrtf(sigma2_s);
      const float ts_ = mu_s - sqrtf(sigma2_s);
      const float z_inv_max_s = ts_ > 0.00000001f ? ts_ : 0.00000001f;
      const float z_inv_min_e = mu_e + sqrtf(sigma2_e);
      const float te_ = mu_e - sqrtf(sigma2_e);
      const float z_inv_max_e = te_ > 0.00000001f ? te_ : 0.00000001f;
      const double g0[2] = { 0, 0 };
      epi_out_t eo;
      /* both searches take Feature::px / Feature::f of the segment feature (*it->ftr), :411-414; the second is not
       * evaluated when the first fails */
      if (!find_epipolar_match_direct(in, frames, rf, cf, in->seg_px + 2 * i, in->seg_f + 3 * i, in->seg_level[i], PLSVO_FTR_CORNER, g0,
                                      1.0 / mu_s, 1.0 / z_inv_min_s, 1.0 / z_inv_max_s, 1, &z_s, &eo) ||
          !find_epipolar_match_direct(in, frames, rf, cf, in->seg_px + 2 * i, in->seg_f + 3 * i, in->seg_level[i], PLSVO_FTR_CORNER, g0,
                                      1.0 / mu_e, 1.0 / z_inv_min_e, 1.0 / z_inv_max_e, 1, &z_e, &eo)) {
        b += 1.0f;
        status = PLSVO_SEED_NO_MATCH;
      } else {
        const double tau_s = compute_tau(&T_ref_cur, sf, z_s, px_error_angle);
        const double zms = z_s - tau_s;
        const double tau_inverse_s = 0.5 * (1.0 / (0.0000001 < zms ? zms : 0.0000001) - 1.0 / (z_s + tau_s));
        const double tau_e = compute_tau(&T_ref_cur, ef, z_e, px_error_angle);
        const double zme = z_e - tau_e;
        const double tau_inverse_e = 0.5 * (1.0 / (0.0000001 < zme ? zme : 0.0000001) - 1.0 / (z_e + tau_e));
        /* updateLineSeed :514-566 */
        const float x_s = (float)(1. / z_s), tau2_s = (float)(tau_inverse_s * tau_inverse_s);
        const float x_e = (float)(1. / z_e), tau2_e = (float)(tau_inverse_e * tau_inverse_e);
        const float norm_scale_s = sqrtf(sigma2_s + tau2_s), norm_scale_e = sqrtf(sigma2_e + tau2_e);
        if (!(isnan(norm_scale_s) || isnan(norm_scale_e))) {
          float f_s, e_s, f_e, e_e;
          seed_update_end(x_s, tau2_s, a, b, z_range_s, &mu_s, &sigma2_s, norm_scale_s, &f_s, &e_s);
          seed_update_end(x_e, tau2_e, a, b, z_range_e, &mu_e, &sigma2_e, norm_scale_e, &f_e, &e_e);
          const float a_s = (e_s - f_s) / (f_s - e_s / f_s), a_e = (e_e - f_e) / (f_e - e_e / f_e);
          const float b_s = a_s * (1.f - f_s) / f_s, b_e = a_e * (1.f - f_e) / f_e;
          a = (a_s < a_e) ? a_e : a_s;                                           /* std::max(a_s, a_e) */
          b = (b_e < b_s) ? b_e : b_s;                                           /* std::min(b_s, b_e) */
        }
        status = PLSVO_SEED_UPDATED;
        if (sqrtf(sigma2_s) < z_range_s / in->convergence_sigma2_thresh && sqrtf(sigma2_e) < z_range_e / in->convergence_sigma2_thresh) {
          const se3_t T_ref_inv = se3_inv(&T_ref);
          const double ws = 1.0 / mu_s, we = 1.0 / mu_e;
          const double pws[3] = { sf[0] * ws, sf[1] * ws, sf[2] * ws }, pwe[3] = { ef[0] * we, ef[1] * we, ef[2] * we };
          se3_act(&T_ref_inv, pws, xw_s); se3_act(&T_ref_inv, pwe, xw_e);
          status = PLSVO_SEED_CONVERGED;
        } else if (isnan(z_inv_min_s) || isnan(z_inv_min_e)) {
          status = PLSVO_SEED_NAN;
        }
      }
    }
    if (out->seg_status) out->seg_status[i] = status;
    if (out->seg_a) out->seg_a[i] = a;
    if (out->seg_b) out->seg_b[i] = b;
    if (out->seg_mu_s) out->seg_mu_s[i] = mu_s;
    if (out->seg_mu_e) out->seg_mu_e[i] = mu_e;
    if (out->seg_sigma2_s) out->seg_sigma2_s[i] = sigma2_s;
    if (out->seg_sigma2_e) out->seg_sigma2_e[i] = sigma2_e;
    if (out->seg_xyz_world_s) for (int k = 0; k < 3; ++k) out->seg_xyz_world_s[3 * i + k] = xw_s[k];
    if (out->seg_xyz_world_e) for (int k = 0; k < 3; ++k) out->seg_xyz_world_e[3 * i + k] = xw_e[k];
    if (out->seg_depth_s) out->seg_depth_s[i] = z_s;
    if (out->seg_depth_e) out->seg_depth_e[i] = z_e;
  }
  return PLSVO_OK;
}

/* --- depth-filter pieces exposed for unit tests ------------------------------------------------ */
int plsvo_oracle_zmssd(const uint8_t* ref_patch64, const uint8_t* cur_patch, int stride) {
  zmssd_t z; zmssd_init(&z, ref_patch64); return zmssd_score(&z, cur_patch, stride);
}
int plsvo_oracle_depth_from_triangulation(const double T_search_ref[7], const double f_ref[3], const double f_cur[3], double* depth) {
  const se3_t T = se3_load(T_search_ref); return depth_from_triangulation(&T, f_ref, f_cur, depth);
}
double plsvo_oracle_compute_tau(const double T_ref_cur[7], const double f[3], double z, double px_error_angle) {
  const se3_t T = se3_load(T_ref_cur); return compute_tau(&T, f, z, px_error_angle);
}
/* updatePointSeed (src/depth_filter.cpp:489-512): state = {a, b, mu, z_range, sigma2} in/out */
void plsvo_oracle_update_point_seed(float x, float tau2, float state[5]) {
  float a = state[0], b = state[1], mu = state[2], sigma2 = state[4];
  const float norm_scale = sqrtf(sigma2 + tau2);
  if (isnan(norm_scale)) return;
  float fq, eq;
  seed_update_end(x, tau2, a, b, state[3], &mu, &sigma2, norm_scale, &fq, &eq);
  a = (eq - fq) / (fq - eq / fq);
  b = a * (1.0f - fq) / fq;
  state[0] = a; state[1] = b; state[2] = mu; state[4] = sigma2;
}

/* ============================================================================================ */
/* CPU-baseline harness: the two hot functions over independent streams on n_threads POSIX threads */
/* (bench.py's cpu_baseline leg; the reference itself is single-threaded on this path)           */
/* ============================================================================================ */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE   /* sched_getaffinity / pthread_setaffinity_np: the baseline's threads are pinned */
#endif
#include <pthread.h>
#include <sched.h>
#include <time.h>

typedef struct {
  int tid, n_threads, n_streams, max_seg, max_pts;
  const plsvo_align_in* aj; const plsvo_oracle_pyr* ref; const plsvo_oracle_pyr* cur; const plsvo_poseopt_in* pj;
  double seconds; long long done;
  int what;                 /* bit 0: SparseImgAlign::run, bit 1: optimizeGaussNewton */
  double* lat_us; int lat_cap, n_lat;   /* per-frame wall times of this thread (first lat_cap frames), or NULL */
  int pinned;
} bench_arg_t;

static int g_bench_threads_pinned = 0;   /* of the last plsvo_oracle_bench* call */
int plsvo_oracle_bench_threads_pinned(void) { return g_bench_threads_pinned; }

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

/* pin the calling thread to the (tid mod #allowed)-th CPU of the process's affinity mask; returns 1 when pinned */
static int pin_to_allowed_cpu(int tid) {
  cpu_set_t allowed;
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return 0;
  const int n = CPU_COUNT(&allowed);
  if (n <= 0) return 0;
  int want = tid % n, seen = 0;
  for (int c = 0; c < CPU_SETSIZE; ++c) {
    if (!CPU_ISSET(c, &allowed)) continue;
    if (seen++ == want) {
      cpu_set_t one; CPU_ZERO(&one); CPU_SET(c, &one);
      return pthread_setaffinity_np(pthread_self(), sizeof(one), &one) == 0;
    }
  }
  return 0;
}

static void* bench_worker(void* p) {
  bench_arg_t* a = (bench_arg_t*)p;
  a->pinned = pin_to_allowed_cpu(a->tid);
  uint8_t* alive = (uint8_t*)malloc((size_t)a->max_seg + 1);
  uint8_t* pk = (uint8_t*)malloc((size_t)a->max_pts + 1);
  uint8_t* sk = (uint8_t*)malloc((size_t)a->max_seg + 1);
  const double t_end = now_s() + a->seconds;
  long long done = 0;
  int n_log = 0;
  for (int i = a->tid; now_s() < t_end; i += a->n_threads) {
    const int s = i % a->n_streams;
    const double t_frame = now_s();
    if (a->what & 1) {
      plsvo_align_out ao; memset(&ao, 0, sizeof(ao)); ao.seg_alive_out = alive;
      plsvo_oracle_sparse_align(&a->aj[s], &a->ref[s], &a->cur[s], &ao, NULL, 0, &n_log);
    }
    if (a->what & 2) {
      plsvo_poseopt_out po; memset(&po, 0, sizeof(po)); po.pt_keep = pk; po.seg_keep = sk;
      plsvo_oracle_pose_optimize(&a->pj[s], &po, NULL, 0, &n_log);
    }
    if (a->lat_us && a->n_lat < a->lat_cap) a->lat_us[a->n_lat++] = 1e6 * (now_s() - t_frame);
    ++done;
  }
  free(alive); free(pk); free(sk);
  a->done = done;
  return NULL;
}

/* runs SparseImgAlign::run + optimizeGaussNewton on streams 0..n_streams-1 (round-robin over the threads) for `seconds`;
 * returns the number of frames completed by all threads, *elapsed gets the wall time */
long long plsvo_oracle_bench_mode(int n_streams, const plsvo_align_in* aj, const plsvo_oracle_pyr* ref, const plsvo_oracle_pyr* cur,
                                  const plsvo_poseopt_in* pj, int n_threads, double seconds, int what, double* elapsed,
                                  double* lat_us, int lat_cap, int* n_lat);
long long plsvo_oracle_bench(int n_streams, const plsvo_align_in* aj, const plsvo_oracle_pyr* ref, const plsvo_oracle_pyr* cur,
                             const plsvo_poseopt_in* pj, int n_threads, double seconds, double* elapsed) {
  return plsvo_oracle_bench_mode(n_streams, aj, ref, cur, pj, n_threads, seconds, 3, elapsed, NULL, 0, NULL);
}
/* what: bit 0 = SparseImgAlign::run, bit 1 = optimizeGaussNewton; lat_us/lat_cap/n_lat: per-frame wall times of thread 0 */
long long plsvo_oracle_bench_mode(int n_streams, const plsvo_align_in* aj, const plsvo_oracle_pyr* ref, const plsvo_oracle_pyr* cur,
                                  const plsvo_poseopt_in* pj, int n_threads, double seconds, int what, double* elapsed,
                                  double* lat_us, int lat_cap, int* n_lat) {
  if (n_streams <= 0 || n_threads <= 0 || !(what & 3)) return -1;
  int max_seg = 1, max_pts = 1;
  for (int s = 0; s < n_streams; ++s) {
    if ((what & 1) && aj[s].n_seg > max_seg) max_seg = aj[s].n_seg;
    if ((what & 2) && pj[s].n_seg > max_seg) max_seg = pj[s].n_seg;
    if ((what & 2) && pj[s].n_pts > max_pts) max_pts = pj[s].n_pts;
  }
  pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
  bench_arg_t* args = (bench_arg_t*)malloc(sizeof(bench_arg_t) * (size_t)n_threads);
  const double t0 = now_s();
  for (int t = 0; t < n_threads; ++t) {
    bench_arg_t a = { t, n_threads, n_streams, max_seg, max_pts, aj, ref, cur, pj, seconds, 0, what, t == 0 ? lat_us : NULL, lat_cap, 0, 0 };
    args[t] = a;
    pthread_create(&th[t], NULL, bench_worker, &args[t]);
  }
  long long total = 0;
  int pinned = 0;
  for (int t = 0; t < n_threads; ++t) { pthread_join(th[t], NULL); total += args[t].done; pinned += args[t].pinned; }
  g_bench_threads_pinned = pinned;
  if (elapsed) *elapsed = now_s() - t0;
  if (n_lat) *n_lat = args[0].n_lat;
  free(th); free(args);
  return total;
}
