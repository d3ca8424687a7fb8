// plsvo_wave.hpp -- wave64 building blocks for the gfx950 kernels: DPP reductions and a wave-cooperative
// 6x6 symmetric solve (one matrix entry per lane).
#pragma once
#include <hip/hip_runtime.h>

#include "plsvo_math.hpp"

namespace plsvo_hip {

template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_bcast_f64(double v) {  // lanes outside ROW_MASK receive 0
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
#define DPP_QUAD_XOR1 0xB1  // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E  // quad_perm [2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

// sum over the 4 lanes of a quad, result in all 4 lanes
__device__ __forceinline__ double quad_sum(double v) {
  v += dpp_mov_f64<DPP_QUAD_XOR1>(v);
  v += dpp_mov_f64<DPP_QUAD_XOR2>(v);
  return v;
}
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_mov_f32<DPP_QUAD_XOR1>(v);
  v += dpp_mov_f32<DPP_QUAD_XOR2>(v);
  return v;
}
// sum over the 64 lanes of a wave; the total is valid in lane 63
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v = quad_sum(v);
  v += dpp_mov_f64<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_mov_f64<DPP_ROW_MIRROR>(v);
  v += dpp_bcast_f64<DPP_ROW_BCAST15, 0xA>(v);
  v += dpp_bcast_f64<DPP_ROW_BCAST31, 0xC>(v);
  return v;
}

// wave-sum of N values at once (N multiple of 6): six independent DPP chains at a time; totals valid in lane 63
template <int N>
__device__ __forceinline__ void wave_sum_array(double* v) {
#pragma unroll
  for (int c0 = 0; c0 < N; c0 += 6) {
#pragma unroll
    for (int k = c0; k < c0 + 6; ++k) v[k] += dpp_mov_f64<DPP_QUAD_XOR1>(v[k]);
#pragma unroll
    for (int k = c0; k < c0 + 6; ++k) v[k] += dpp_mov_f64<DPP_QUAD_XOR2>(v[k]);
#pragma unroll
    for (int k = c0; k < c0 + 6; ++k) v[k] += dpp_mov_f64<DPP_ROW_HALF_MIRROR>(v[k]);
#pragma unroll
    for (int k = c0; k < c0 + 6; ++k) v[k] += dpp_mov_f64<DPP_ROW_MIRROR>(v[k]);
#pragma unroll
    for (int k = c0; k < c0 + 6; ++k) v[k] += dpp_bcast_f64<DPP_ROW_BCAST15, 0xA>(v[k]);
#pragma unroll
    for (int k = c0; k < c0 + 6; ++k) v[k] += dpp_bcast_f64<DPP_ROW_BCAST31, 0xC>(v[k]);
  }
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane /*wave-uniform*/) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// index of entry (r,c) in the row-major upper triangle of a symmetric 6x6 (21 entries)
__device__ __forceinline__ int sym6_index(int r, int c) {
  const int a = r < c ? r : c, b = r < c ? c : r;
  return a * 6 - (a * (a - 1)) / 2 + (b - a);
}

// Solve H x = rhs for a symmetric 6x6 H, cooperatively by ONE FULL WAVE (all 64 lanes must be active).
//   tot[0..20]  upper triangle of H (row-major), tot[21..26] rhs  -- readable by every lane (LDS)
//   x[0..5]     result, returned in registers of every lane (wave-uniform)
// Method: Gauss-Jordan elimination of the augmented 6x7 system with the pivot order of Eigen's LDLT
// (largest remaining |diagonal| of the Schur complement, src/sparse_img_align.cpp:699 and
// src/pose_optimizer.cpp:170 call H.ldlt().solve()).  Lane 8*i+j holds entry (i,j); column 6 is the rhs.
// The pivots are exactly LDLT's D entries; like Eigen's solve, a pivot that is zero (or below 1/DBL_MAX)
// yields a zero component, so an all-zero system returns x = 0.  NaN/Inf propagate into x.
__device__ __forceinline__ void wave_solve6_core(double m, double* x);

__device__ __forceinline__ void wave_solve6(const double* tot, double* x) {
  const int lane = threadIdx.x & 63;
  const int i = lane >> 3, j = lane & 7;
  double m = 0.0;
  if (i < 6 && j < 6) m = tot[sym6_index(i, j)];
  else if (i < 6 && j == 6) m = tot[21 + i];
  wave_solve6_core(m, x);
}

// the same, with the 27 inputs held in registers: lane k (k < 27) passes tot[k] in `tot_lane`
__device__ __forceinline__ void wave_solve6_reg(double tot_lane, double* x) {
  const int lane = threadIdx.x & 63;
  const int i = lane >> 3, j = lane & 7;
  const int src = (i < 6 && j < 6) ? sym6_index(i, j) : ((i < 6 && j == 6) ? 21 + i : 0);
  double m = __shfl(tot_lane, src, 64);
  if (!(i < 6 && j <= 6)) m = 0.0;
  wave_solve6_core(m, x);
}

__device__ __forceinline__ void wave_solve6_core(double m, double* x) {
  const int lane = threadIdx.x & 63;
  const int i = lane >> 3, j = lane & 7;
  unsigned done = 0u, zero_piv = 0u;
#pragma unroll 1
  for (int step = 0; step < 6; ++step) {
    int p = -1; double best = -1.0, piv = 0.0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double dg = readlane_f64(m, 9 * r);
      const double a = fabs(dg);
      if (!((done >> r) & 1u) && (p < 0 || a > best)) { best = a; p = r; piv = dg; }
    }
    done |= 1u << p;
    if (fabs(piv) > 0.0) {
      const double mp_j = __shfl(m, 8 * p + j, 64);  // M[p][j]
      const double mi_p = __shfl(m, 8 * i + p, 64);  // M[i][p]
      if (i != p && i < 6) m -= (mi_p / piv) * mp_j;
    } else {
      zero_piv |= 1u << p;
    }
  }
  const double rhs = __shfl(m, 8 * i + 6, 64);
  const double dgi = __shfl(m, 9 * (i < 6 ? i : 0), 64);
  const double tolerance = 1.0 / 1.7976931348623157e308;
  const double xi = (((zero_piv >> i) & 1u) || !(fabs(dgi) > tolerance)) ? ((dgi != dgi || rhs != rhs) ? (rhs + dgi) : 0.0) : rhs / dgi;
#pragma unroll
  for (int r = 0; r < 6; ++r) x[r] = readlane_f64(xi, 8 * r);
}

// sin and cos of a small angle (|x| <= pi/4: Taylor/Horner to x^17 / x^16, < 1 ulp); larger angles use ocml.
// Gauss-Newton updates are tiny rotations, so the fast path is the one that runs.
__device__ __forceinline__ void sincos_small(double x, double* s, double* c) {
  if (fabs(x) <= 0.7853981633974483) {
    const double z = x * x;
    double ps = -1.0 / 355687428096000.0;                      // -1/17!
    ps = ps * z + 1.0 / 1307674368000.0;                       //  1/15!
    ps = ps * z - 1.0 / 6227020800.0;                          // -1/13!
    ps = ps * z + 1.0 / 39916800.0;                            //  1/11!
    ps = ps * z - 1.0 / 362880.0;                              // -1/9!
    ps = ps * z + 1.0 / 5040.0;                                //  1/7!
    ps = ps * z - 1.0 / 120.0;                                 // -1/5!
    ps = ps * z + 1.0 / 6.0;                                   //  1/3!  (sign folded below)
    *s = x - x * z * ps;
    double pc = 1.0 / 20922789888000.0;                        //  1/16!
    pc = pc * z - 1.0 / 87178291200.0;                         // -1/14!
    pc = pc * z + 1.0 / 479001600.0;                           //  1/12!
    pc = pc * z - 1.0 / 3628800.0;                             // -1/10!
    pc = pc * z + 1.0 / 40320.0;                               //  1/8!
    pc = pc * z - 1.0 / 720.0;                                 // -1/6!
    pc = pc * z + 1.0 / 24.0;                                  //  1/4!
    *c = 1.0 - 0.5 * z + z * z * pc;
  } else {
    *s = sin(x); *c = cos(x);
  }
}

// Sophus::SE3::exp with the small-angle sincos above (same formulas as plsvo_math.hpp::se3_exp)
__device__ __forceinline__ SE3d se3_exp_dev(const double* u) {
  SE3d r;
  const double ox = u[3], oy = u[4], oz = u[5];
  const double theta = sqrt(ox * ox + oy * oy + oz * oz);
  double sh, ch, st, ct;
  sincos_small(0.5 * theta, &sh, &ch);
  sincos_small(theta, &st, &ct);
  double imag_factor;
  if (theta < 1e-10) {
    const double theta_sq = theta * theta;
    imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * (theta_sq * theta_sq);
  } else {
    imag_factor = sh / theta;
  }
  Quat q = { imag_factor * ox, imag_factor * oy, imag_factor * oz, ch };
  r.q = quat_normalized(q);
  double V[9];
  if (theta < 1e-10) {
    quat_to_matrix(r.q, V);
  } else {
    const double theta_sq = theta * theta;
    const double a = (1 - ct) / theta_sq;
    const double b = (theta - st) / (theta_sq * theta);
    const double O2_00 = -(oy * oy + oz * oz), O2_11 = -(ox * ox + oz * oz), O2_22 = -(ox * ox + oy * oy);
    const double O2_01 = ox * oy, O2_02 = ox * oz, O2_12 = oy * oz;
    V[0] = 1.0 + b * O2_00;      V[1] = a * -oz + b * O2_01;  V[2] = a * oy + b * O2_02;
    V[3] = a * oz + b * O2_01;   V[4] = 1.0 + b * O2_11;      V[5] = a * -ox + b * O2_12;
    V[6] = a * -oy + b * O2_02;  V[7] = a * ox + b * O2_12;   V[8] = 1.0 + b * O2_22;
  }
  for (int i = 0; i < 3; ++i) r.t[i] = V[i * 3 + 0] * u[0] + V[i * 3 + 1] * u[1] + V[i * 3 + 2] * u[2];
  return r;
}


}  // namespace plsvo_hip
