// poseopt_kernels.hip -- motion-only pose optimisation on gfx950: one workgroup per frame, the scale
// pass, all Gauss-Newton iterations, the covariance, the outlier cull and the medians run in ONE launch.
//
// Replaces (reference file:line):
//   plsvo::pose_optimizer::optimizeGaussNewton, 9-argument   src/pose_optimizer.cpp:38-260
//   plsvo::pose_optimizer::optimizeGaussNewton, 10-argument  src/pose_optimizer.cpp:262-582
//   [ext] vk::robust_cost::MADScaleEstimator / TukeyWeightFunction, vk::getMedian, Eigen ldlt()/inverse()
//
// Work per call is tiny (hundreds of features x ~100 flops x <=10 iterations): this kernel is
// latency-bound by construction; it exists so that a batch of frames is one launch and the pose never
// leaves the device between the alignment and the optimisation.  One lane per feature, lane-private
// 6x6 (upper) + 6x1 accumulators in double, fixed-shape reduction (butterfly reduce-scatter inside the DPP
// rows, rows and waves through LDS), wave-cooperative 6x6 solve.  The workgroup is ONE wave per frame for
// large batches (nothing to gain from cross-wave work on ~300 features: 8 frames per CU) and four waves for
// small ones (the Gauss-Newton loop of a single frame is then ~2x shorter).
// Medians (vk::getMedian = element floor(n/2) of the sorted vector) are exact: block-wide radix
// select over the IEEE bit patterns (all values are non-negative), 8 bits per pass.
#include <hip/hip_runtime.h>

#include "plsvo_dev.hpp"
#include "plsvo_math.hpp"
#include "plsvo_wave.hpp"

// Waves per SIMD.  The wave-per-frame shape (64 threads) is capped at 168 VGPRs = THREE waves per SIMD: its feature loops stay spill-free
// (the 58 spilled registers are touched ~20 times per Gauss-Newton iteration, around the solve) and the f64 division chains of a third
// wave fill latency the other two leave: 500 + 200 features x 32768 frames 3.26 -> 3.19 ms.  The row shape LOSES 9 % at three waves
// (1.40 -> 1.53 ms at 200 + 80: four frames' serial code per wave meets the spills) and both lose 2x at four; measured on one MI355X box,
// profiles/r05_poseopt_experiments.log.  -DPLSVO_POSEOPT_WAVES=n / -DPLSVO_POSEOPT_ROWS_WAVES=n rebuild the experiment.
#ifndef PLSVO_POSEOPT_WAVES
#define PLSVO_POSEOPT_WAVES 3
#endif
#ifdef PLSVO_WAVE_EMU   // (the host emulation build, tests/host/: kernels are plain functions there)
#define PLSVO_PO_OCC(T)
#else
#define PLSVO_PO_OCC(T) __attribute__((amdgpu_waves_per_eu((T) == 64 ? PLSVO_POSEOPT_WAVES : 1, (T) == 64 ? PLSVO_POSEOPT_WAVES : 8)))
#endif
#if defined(PLSVO_POSEOPT_ROWS_WAVES) && !defined(PLSVO_WAVE_EMU)
#define PLSVO_PO_ROWS_OCC __attribute__((amdgpu_waves_per_eu(PLSVO_POSEOPT_ROWS_WAVES, PLSVO_POSEOPT_ROWS_WAVES)))
#else
#define PLSVO_PO_ROWS_OCC
#endif
namespace plsvo_hip {

// optional per-phase timing (compile with -DPLSVO_TIMING): thread 0 accumulates s_memtime deltas
#ifdef PLSVO_TIMING
#define PTICK(slot) do { if (threadIdx.x == 0) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); s_time[slot] += t__ - s_tlast; s_tlast = t__; } } while (0)
#else
#define PTICK(slot) do { } while (0)
#endif

#define PO_RED 32
#ifndef PO_RADIX_BITS
#define PO_RADIX_BITS 8    // bits per radix-select pass (measured on MI355X: 8 -> 0.72 ms, 11 -> 0.81 ms, 6 -> 0.74 ms per 8192 frames)
#endif
#define PO_BINS (1 << PO_RADIX_BITS)

// k-th smallest (0-based) of n non-negative IEEE values given as unsigned bit patterns of BITS bits.
// get(i) returns the pattern of element i (invalid elements must return all-ones).
template <int PO_T, int BITS, typename U, typename GET>
__device__ U block_radix_select(GET get, int n, int k, int* s_hist, int* s_sel) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  U prefix = 0, mask = 0;
  int shift = BITS;
  while (shift > 0) {
    const int bits = shift >= PO_RADIX_BITS ? PO_RADIX_BITS : shift;
    shift -= bits;
    const int nb = 1 << bits;
    for (int i = tid; i < PO_BINS; i += PO_T) s_hist[i] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += PO_T) {
      const U v = get(i);
      if ((v & mask) == prefix) atomicAdd(&s_hist[(int)((v >> shift) & (U)(nb - 1))], 1);
    }
    __syncthreads();
    // block scan over the bins: each thread owns PO_BINS/PO_T consecutive bins
    constexpr int PER = PO_BINS >= PO_T ? PO_BINS / PO_T : 1;   // more threads than bins: the surplus threads own no bin
    const bool owner = tid * PER < PO_BINS;
    int loc[PER]; int local = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { loc[j] = owner ? s_hist[tid * PER + j] : 0; local += loc[j]; }
    int incl = local;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
    if (lane == 63) s_sel[4 + wave] = incl;
    __syncthreads();
    int wave_off = 0;
    for (int w = 0; w < wave; ++w) wave_off += s_sel[4 + w];
    int cum = wave_off + incl - local;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      if (owner && k >= cum && k < cum + loc[j]) { s_sel[0] = tid * PER + j; s_sel[1] = k - cum; s_sel[2] = loc[j]; }
      cum += loc[j];
    }
    __syncthreads();
    prefix |= ((U)s_sel[0]) << shift;
    mask |= ((U)(nb - 1)) << shift;
    k = s_sel[1];
    const int survivors = s_sel[2];
    __syncthreads();
    if (survivors == 1 && shift > 0) {
      // one candidate left: it is the answer, the remaining digits need no histogram (typical after 2-3 of the 4 / 8 passes)
      for (int i = tid; i < n; i += PO_T) {
        const U v = get(i);
        if ((v & mask) == prefix) { s_hist[0] = (int)(unsigned)(v & (U)0xffffffffu); s_hist[1] = (int)(unsigned)((unsigned long long)v >> 32); }
      }
      __syncthreads();
      const U r = (U)(((unsigned long long)(unsigned)s_hist[1] << 32) | (unsigned long long)(unsigned)s_hist[0]);
      __syncthreads();
      return r;
    }
  }
  return prefix;
}

// LDS roles: s_red 32*(PO_T/16) doubles (row partials); s_pose 0..8 R, 9..11 t, 12..18 model, 19..25 T_old, 26 chi2;
// s_ctl[0] break flag

// one GN loop (src/pose_optimizer.cpp:103-195 and the identical text at :469-563)
// sqrt(a*a + b*b) in float with every product and the sum rounded on its own (Eigen's Vector2f::norm() in the un-contracted build the
// oracle pins, src/pose_optimizer.cpp:84-87).  HIP's __fmul_rn / __fadd_rn are plain `*` / `+` and hipcc contracts by default: fused,
// the value is one ulp off for some inputs, and when that input is the median line error the MAD scale -- hence every Tukey weight --
// moves by a float ulp (found on the second config-5 seed set: chi2 of the first iteration 2.9e-8 off, tests/test_gpu_parity.py).
__device__ __forceinline__ float norm2_f32(float a, float b) {
#pragma clang fp contract(off)
  const float p = a * a, q = b * b;
  return sqrtf(p + q);
}

// ---- per-feature arithmetic, shared by the workgroup-per-frame kernel and the row-per-frame kernel ----------------------------------
// P[0..8] = R (row-major), P[9..11] = t of the current model.  obs[2f], obs[2f+1] = f.x / f.z, f.y / f.z of point f: the observation's
// normalised coordinates do not depend on the pose, so the scale pass divides once and the Gauss-Newton and cull passes read the two
// quotients back (same operands, same IEEE results: bit-identical, two f64 divisions less per point and pass).

// 1.0 / (1 << level) (the reference's scale of a pyramid level, :119-120): a power of two, written into the exponent -- exact
__device__ __forceinline__ double pow2_neg(int level) { return __hiloint2double((1023 - level) << 20, 0); }

// one feature of a Gauss-Newton iteration (src/pose_optimizer.cpp:107-166 / :473-533): residual, Jacobian, Tukey weight, accumulation
// into acc (0..20 A upper row-major, 21..26 b, 27 chi2, 28 #points, 29 #segments)
__device__ __forceinline__ void popt_accumulate_feature(const PoseBatchDev& b, const PoseJobDev& job, int f, const double (&P)[12], double scale_pt,
                                                        double scale_ls, bool first_iter, double* init_vec, const double* obs, double* acc) {
  const int np = job.n_pts;
  const double R0 = P[0], R1 = P[1], R2 = P[2], R3 = P[3], R4 = P[4], R5 = P[5], R6 = P[6], R7 = P[7], R8 = P[8], t0 = P[9], t1 = P[10], t2 = P[11];
  double J[12], e0, e1, weight, cnt_pt, cnt_ls;   // (the two counters are added at the common tail: `acc[28 + is_segment] += 1` would be a dynamically indexed private array, i.e. scratch)
  if (f < np) {
    const int i = job.pt_off + f;
    if (!b.pt_keep[i]) return;
    const double x = b.pt_pos[3 * i], y = b.pt_pos[3 * i + 1], z = b.pt_pos[3 * i + 2];
    const double xyz[3] = { R0 * x + R1 * y + R2 * z + t0, R3 * x + R4 * y + R5 * z + t1, R6 * x + R7 * y + R8 * z + t2 };
    jacobian_xyz2uv(xyz, J);
    e0 = obs[2 * f] - xyz[0] / xyz[2];
    e1 = obs[2 * f + 1] - xyz[1] / xyz[2];
    const double sic = pow2_neg(b.pt_level[i]);
    e0 *= sic; e1 *= sic;
    if (first_iter) init_vec[f] = e0 * e0 + e1 * e1;
#pragma unroll
    for (int k = 0; k < 12; ++k) J[k] *= sic;
    weight = (double)tukey_weight((float)(sqrt(e0 * e0 + e1 * e1) / scale_pt));
    cnt_pt = 1.0; cnt_ls = 0.0;
  } else {
    const int s = job.seg_off + (f - np);
    if (!b.seg_keep[s]) return;
    double Js[12], Je[12];
    const double sx = b.seg_spos[3 * s], sy = b.seg_spos[3 * s + 1], sz = b.seg_spos[3 * s + 2];
    const double ex = b.seg_epos[3 * s], ey = b.seg_epos[3 * s + 1], ez = b.seg_epos[3 * s + 2];
    const double xs[3] = { R0 * sx + R1 * sy + R2 * sz + t0, R3 * sx + R4 * sy + R5 * sz + t1, R6 * sx + R7 * sy + R8 * sz + t2 };
    const double xe[3] = { R0 * ex + R1 * ey + R2 * ez + t0, R3 * ex + R4 * ey + R5 * ez + t1, R6 * ex + R7 * ey + R8 * ez + t2 };
    jacobian_xyz2uv(xs, Js);
    jacobian_xyz2uv(xe, Je);
    const double l0 = b.seg_line[3 * s], l1 = b.seg_line[3 * s + 1], l2 = b.seg_line[3 * s + 2];
    const float ds = (float)(l0 * (xs[0] / xs[2]) + l1 * (xs[1] / xs[2]) + l2 * 1.0);
    const float de = (float)(l0 * (xe[0] / xe[2]) + l1 * (xe[1] / xe[2]) + l2 * 1.0);
    const double sic = pow2_neg(b.seg_level[s]);
    e0 = (double)ds * sic; e1 = (double)de * sic;
    if (first_iter) init_vec[f] = e0 * e0 + e1 * e1;
    const double en = sqrt(e0 * e0 + e1 * e1);
    const double ks = sic * (double)ds / en;  // same factor for both rows (:156-157)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      J[c] = l0 * (Js[c] * ks) + l1 * (Js[6 + c] * ks);
      J[6 + c] = l0 * (Je[c] * ks) + l1 * (Je[6 + c] * ks);
    }
    weight = (double)tukey_weight((float)(en / scale_ls));
    cnt_pt = 0.0; cnt_ls = 1.0;
  }
  // A += J^T J w, b -= J^T e w (:163-165): the two rows of J are weighted once (12 products) and every entry is two fused multiply-adds
  // -- 21 x 2 + 6 x 2 + 12 double instructions where (J_i J_j + J'_i J'_j) w, written out, issues 21 x 4 + 6 x 4; the sums differ from the
  // reference's in the last bits of each term only (A, b agree with the oracle to 1e-15).  Measured -2 % per launch; sharing a refined
  // reciprocal between the quotients of one denominator (bit-identical, exponent-guarded) measured +11 % and is not in the tree
  // (profiles/r05_poseopt_experiments.log).
  double wJ[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) wJ[k] = J[k] * weight;
  int k = 0;
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int jj = i; jj < 6; ++jj) { acc[k] = fma(wJ[i], J[jj], fma(wJ[6 + i], J[6 + jj], acc[k])); ++k; }
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[21 + i] = fma(-wJ[i], e0, fma(-wJ[6 + i], e1, acc[21 + i]));
  acc[27] += (e0 * e0 + e1 * e1) * weight;
  acc[28] += cnt_pt; acc[29] += cnt_ls;
}

// the float error of a feature in the scale pass (:57-87): points sqrt(e0^2 + e1^2) scaled by the level, lines the un-scaled float norm
__device__ __forceinline__ float popt_scale_error(const PoseBatchDev& b, const PoseJobDev& job, int f, const double (&P)[12], double* obs) {
  const int np = job.n_pts;
  const double R0 = P[0], R1 = P[1], R2 = P[2], R3 = P[3], R4 = P[4], R5 = P[5], R6 = P[6], R7 = P[7], R8 = P[8], t0 = P[9], t1 = P[10], t2 = P[11];
  if (f < np) {
    const int i = job.pt_off + f;
    const double x = b.pt_pos[3 * i], y = b.pt_pos[3 * i + 1], z = b.pt_pos[3 * i + 2];
    const double xc = R0 * x + R1 * y + R2 * z + t0, yc = R3 * x + R4 * y + R5 * z + t1, zc = R6 * x + R7 * y + R8 * z + t2;
    const double fz = b.pt_f[3 * i + 2];
    const double ox = b.pt_f[3 * i] / fz, oy = b.pt_f[3 * i + 1] / fz;
    obs[2 * f] = ox; obs[2 * f + 1] = oy;
    double e0 = ox - xc / zc, e1 = oy - yc / zc;
    const double sic = pow2_neg(b.pt_level[i]);
    e0 *= sic; e1 *= sic;
    return (float)sqrt(e0 * e0 + e1 * e1);
  }
  const int s = job.seg_off + (f - np);
  const double sx = b.seg_spos[3 * s], sy = b.seg_spos[3 * s + 1], sz = b.seg_spos[3 * s + 2];
  const double ex = b.seg_epos[3 * s], ey = b.seg_epos[3 * s + 1], ez = b.seg_epos[3 * s + 2];
  const double xs0 = R0 * sx + R1 * sy + R2 * sz + t0, xs1 = R3 * sx + R4 * sy + R5 * sz + t1, xs2 = R6 * sx + R7 * sy + R8 * sz + t2;
  const double xe0 = R0 * ex + R1 * ey + R2 * ez + t0, xe1 = R3 * ex + R4 * ey + R5 * ez + t1, xe2 = R6 * ex + R7 * ey + R8 * ez + t2;
  const double l0 = b.seg_line[3 * s], l1 = b.seg_line[3 * s + 1], l2 = b.seg_line[3 * s + 2];
  const float es = (float)(l0 * (xs0 / xs2) + l1 * (xs1 / xs2) + l2 * 1.0);   // not scaled by the level (:84-87)
  const float ee = (float)(l0 * (xe0 / xe2) + l1 * (xe1 / xe2) + l2 * 1.0);
  return norm2_f32(es, ee);
}

// the cull of one feature (:201-242): clears its keep flag when the error exceeds the threshold, returns the squared error;
// deleted = 1 (point) / 2 (segment) / 0
__device__ __forceinline__ double popt_cull_feature(const PoseBatchDev& b, const PoseJobDev& job, int f, const double (&P)[12], double thr_pt, double thr_ls,
                                                    const double* obs, int& deleted) {
  const int np = job.n_pts;
  const double R0 = P[0], R1 = P[1], R2 = P[2], R3 = P[3], R4 = P[4], R5 = P[5], R6 = P[6], R7 = P[7], R8 = P[8], t0 = P[9], t1 = P[10], t2 = P[11];
  double e0, e1;
  deleted = 0;
  if (f < np) {
    const int i = job.pt_off + f;
    const double x = b.pt_pos[3 * i], y = b.pt_pos[3 * i + 1], z = b.pt_pos[3 * i + 2];
    const double xc = R0 * x + R1 * y + R2 * z + t0, yc = R3 * x + R4 * y + R5 * z + t1, zc = R6 * x + R7 * y + R8 * z + t2;
    e0 = obs[2 * f] - xc / zc; e1 = obs[2 * f + 1] - yc / zc;
    const double sic = pow2_neg(b.pt_level[i]);
    e0 *= sic; e1 *= sic;
    if (sqrt(e0 * e0 + e1 * e1) > thr_pt) { b.pt_keep[i] = 0; deleted = 1; }
  } else {
    const int s = job.seg_off + (f - np);
    const double sx = b.seg_spos[3 * s], sy = b.seg_spos[3 * s + 1], sz = b.seg_spos[3 * s + 2];
    const double ex = b.seg_epos[3 * s], ey = b.seg_epos[3 * s + 1], ez = b.seg_epos[3 * s + 2];
    const double xs0 = R0 * sx + R1 * sy + R2 * sz + t0, xs1 = R3 * sx + R4 * sy + R5 * sz + t1, xs2 = R6 * sx + R7 * sy + R8 * sz + t2;
    const double xe0 = R0 * ex + R1 * ey + R2 * ez + t0, xe1 = R3 * ex + R4 * ey + R5 * ez + t1, xe2 = R6 * ex + R7 * ey + R8 * ez + t2;
    const double l0 = b.seg_line[3 * s], l1 = b.seg_line[3 * s + 1], l2 = b.seg_line[3 * s + 2];
    const double sic = pow2_neg(b.seg_level[s]);
    e0 = (l0 * (xs0 / xs2) + l1 * (xs1 / xs2) + l2 * 1.0) * sic;     // doubles here, no float truncation (:229)
    e1 = (l0 * (xe0 / xe2) + l1 * (xe1 / xe2) + l2 * 1.0) * sic;
    if (sqrt(e0 * e0 + e1 * e1) > thr_ls) { b.seg_keep[s] = 0; deleted = 2; }
  }
  return e0 * e0 + e1 * e1;
}

template <int PO_T>
__device__ void popt_gn_loop(const PoseBatchDev& b, const PoseJobDev& job, PoseStateDev* st, int job_id, double* s_red,
                             double* s_pose, int* s_ctl, int n_iter, int phase, double scale_pt, double scale_ls,
                             double* init_vec, const double* obs, double* s_tot) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int np = job.n_pts, ns = job.n_seg, nf = np + ns;
  for (int iter = 0; iter < n_iter; ++iter) {
    double acc[32];   // 0..20 A (upper, row-major), 21..26 b, 27 chi2, 28 #points, 29 #segments, 30..31 unused
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
    if constexpr (PO_T == 64) {
      // (wave per frame, three waves per SIMD: the pose is re-read from LDS for every feature round -- twelve broadcast reads against ~250
      //  f64 instructions -- instead of living in 24 registers across the loop; the laundered offset keeps the compiler from hoisting them)
      for (int f = tid; f < nf; f += PO_T) {
        int off = 0;
#ifndef PLSVO_WAVE_EMU
        asm volatile("" : "+v"(off));
#endif
        double P[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) P[k] = s_pose[off + k];
        popt_accumulate_feature(b, job, f, P, scale_pt, scale_ls, iter == 0, init_vec, obs, acc);
      }
    } else {
    double P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = s_pose[k];
    for (int f = tid; f < nf; f += PO_T) popt_accumulate_feature(b, job, f, P, scale_pt, scale_ls, iter == 0, init_vec, obs, acc);
    }
    {
      double out2[2];
      row_reduce_scatter32(acc, out2);
      *reinterpret_cast<double2*>(s_red + (tid >> 4) * 32 + row_reduce_scatter32_index(lane)) = make_double2(out2[0], out2[1]);
    }
    __syncthreads();
    if (wave == 0) {
      const double tot = reduce_rows_finish<PO_T / 16>(s_red);
      if (lane < 30) s_tot[lane] = tot;
      double dT[6];
      wave_solve6_reg(tot, dT, job.ldlt_flavour);                            // A.ldlt().solve(b) :170
      const double new_chi2 = readlane_f64(tot, 27), npt = readlane_f64(tot, 28), nls = readlane_f64(tot, 29);
      if (lane == 0) {
        s_pose[27] += npt; s_pose[28] += nls;
        s_ctl[1 + phase] += 1;
        SE3d model = se3_load(s_pose + 12);
        int accepted = 1, brk = 0;
        if ((iter > 0 && new_chi2 > s_pose[26]) || isnan(dT[0])) {          // :173-180
          model = se3_load(s_pose + 19); accepted = 0; brk = 1;
        } else {
          const SE3d Tn = se3_mul_dev(se3_exp_dev(dT), model);               // :183 left update
          se3_store(model, s_pose + 19);
          model = Tn; s_pose[26] = new_chi2;
          if (norm_max6(dT) <= 0.0000000001) brk = 1;                        // EPS, global.h:99
        }
        se3_store(model, s_pose + 12);
        quat_to_matrix(model.q, s_pose); s_pose[9] = model.t[0]; s_pose[10] = model.t[1]; s_pose[11] = model.t[2];
        s_ctl[0] = brk;
        if (b.log) {
          const int lc = st->log_count;
          if (lc < b.log_cap) {
            plsvo_poseopt_iterlog* r = b.log + (size_t)job_id * b.log_cap + lc;
            r->phase = phase; r->iter = iter; r->accepted = accepted; r->reserved0 = 0; r->new_chi2 = new_chi2;
            for (int k = 0; k < 6; ++k) r->dT[k] = dT[k];
            se3_store(model, r->T_after);
          }
          st->log_count = lc + 1;
        }
      }
    }
    __syncthreads();
    if (b.log && tid == 0) {   // A and b of the trace come from s_tot (written by lanes 0..26 of wave 0 above)
      const int lc = st->log_count - 1;
      if (lc >= 0 && lc < b.log_cap) {
        plsvo_poseopt_iterlog* r = b.log + (size_t)job_id * b.log_cap + lc;
        for (int i = 0; i < 6; ++i) for (int jj = 0; jj < 6; ++jj) r->A[i * 6 + jj] = s_tot[sym6_index(i, jj)];
        for (int k = 0; k < 6; ++k) r->b[k] = s_tot[21 + k];
      }
    }
    if (s_ctl[0]) break;
  }
}

template <int PO_T>
__global__ __launch_bounds__(PO_T) PLSVO_PO_OCC(PO_T) void pose_opt_kernel(PoseBatchDev b, double* poses) {
  const int job_id = b.order ? b.order[blockIdx.x] : (int)blockIdx.x;
  const PoseJobDev job = b.jobs[job_id];
  PoseStateDev* st = b.state + job_id;
  const int tid = threadIdx.x;
  if (tid == 0 && b.work_key) b.work_key[job_id] = 0;
  const int np = job.n_pts, ns = job.n_seg, nf = np + ns;

  __shared__ __align__(16) double s_red[32 * (PO_T / 16)];
  __shared__ double s_lu[36];     // covariance: the LU factors live in LDS (dynamic pivot indexing would put them in scratch)
  __shared__ int s_perm[8];
  __shared__ double s_pose[32];   // 27: point-iterations, 28: line-iterations
  __shared__ double s_tot[32];
  __shared__ int s_ctl[32];
  __shared__ int s_hist[PO_BINS];                           // 8 KB: radix histogram, or staging for the rank select
  __shared__ int s_sel[16];

  // scratch: floats [0,np) point errors, [np, np+ns) line errors; doubles [0,nf) init (first loop),
  // [nf,2nf) init (refinement), [2nf,3nf) final, [3nf, 3nf + 2 np) the points' normalised observations
  const size_t fbase = (size_t)job.pt_off + (size_t)job.seg_off;
  float* errs = b.scratch_f32 + fbase;
  double* vec = b.scratch_f64 + 5 * fbase;

#ifdef PLSVO_TIMING
  __shared__ unsigned long long s_time[8];
  __shared__ unsigned long long s_tlast;
  if (tid == 0) { for (int k = 0; k < 8; ++k) s_time[k] = 0; s_tlast = __builtin_amdgcn_s_memtime(); }
#endif
  if (tid == 0) {
    SE3d m = se3_load(job.T0);
    se3_store(m, s_pose + 12); se3_store(m, s_pose + 19); s_pose[26] = 0.0; s_pose[27] = 0.0; s_pose[28] = 0.0;
    for (int k = 0; k < 32; ++k) s_tot[k] = 0.0;
    quat_to_matrix(m.q, s_pose); s_pose[9] = m.t[0]; s_pose[10] = m.t[1]; s_pose[11] = m.t[2];
    s_ctl[0] = 0; s_ctl[1] = 0; s_ctl[2] = 0;
    st->log_count = 0; st->status = 0; st->iters = 0; st->iters_ref = 0; st->pt_iters = 0; st->seg_iters = 0;
    st->num_obs_pt = 0; st->num_obs_ls = 0; st->estimated_scale = 0; st->error_init = 0; st->error_final = 0;
    for (int k = 0; k < 36; ++k) st->cov[k] = 0.0;
    for (int k = 0; k < 7; ++k) st->T[k] = job.T0[k];
  }
  for (int i = tid; i < np; i += PO_T) b.pt_keep[job.pt_off + i] = 1;
  for (int s = tid; s < ns; s += PO_T) b.seg_keep[job.seg_off + s] = 1;
  __syncthreads();
  if (nf == 0) {                                                              // errors.empty() :88-89
    if (tid == 0) { st->status = 1; if (poses) for (int k = 0; k < 7; ++k) poses[7 * job_id + k] = job.T0[k]; }
    return;
  }

  // ---- scale pass :57-95 ----
  if constexpr (PO_T == 64) {   // (the pose from LDS per feature round: see popt_gn_loop)
    for (int f = tid; f < nf; f += PO_T) {
      int off = 0;
#ifndef PLSVO_WAVE_EMU
      asm volatile("" : "+v"(off));
#endif
      double P[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) P[k] = s_pose[off + k];
      errs[f] = popt_scale_error(b, job, f, P, vec + 3 * nf);
    }
  } else {
    double P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = s_pose[k];
    for (int f = tid; f < nf; f += PO_T) errs[f] = popt_scale_error(b, job, f, P, vec + 3 * nf);
  }
  __syncthreads();
  PTICK(0);
  // MAD scale = 1.48f * median (float).  Zero points: the reference is undefined (:70); we define 1.0.
  double scale_pt = 1.0, scale_ls = 1.0;
  auto median_f32 = [&](const float* v, int n) -> float {
    return __uint_as_float(block_radix_select<PO_T, 32, uint32_t>([&](int i) { return (uint32_t)__float_as_uint(v[i]); }, n, n / 2, s_hist, s_sel));
  };
  if (np > 0) scale_pt = (double)__fmul_rn(1.48f, median_f32(errs, np));
  if (ns > 0) scale_ls = (double)__fmul_rn(1.48f, median_f32(errs + np, ns));

  PTICK(1);
  // ---- first GN loop ----
  if (job.n_iter <= 0) for (int f = tid; f < nf; f += PO_T) vec[f] = __longlong_as_double(0x7ff0000000000000LL);
  popt_gn_loop<PO_T>(b, job, st, job_id, s_red, s_pose, s_ctl, job.n_iter, 0, scale_pt, scale_ls, vec, vec + 3 * nf, s_tot);

  PTICK(2);
  // ---- covariance :197-199 (from the last assembled A, even if that iteration was rolled back) ----
  if (tid < 64) {   // wave 0: lane 0 factorises, lanes 0..5 each solve for one column of the inverse
    const double f2 = job.fx * job.fx;
    if (tid < 36) s_lu[tid] = s_tot[sym6_index(tid / 6, tid % 6)] * f2;
    wave_lds_fence();
    if (tid == 0) lu6_lds(s_lu, s_perm);
    wave_lds_fence();
    if (tid < 6) inv6_column_lds(s_lu, s_perm, tid, st->cov);
  }

  // ---- cull :201-242 ----
  const double thr_pt = job.reproj_thresh / job.fx;
  const double thr_ls = thr_pt * scale_ls / scale_pt;
  int del_pt = 0, del_ls = 0;
  {
    double P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = s_pose[k];
    for (int f = tid; f < nf; f += PO_T) {
      if constexpr (PO_T == 64) {   // (the pose from LDS per feature round: see popt_gn_loop)
        int off = 0;
#ifndef PLSVO_WAVE_EMU
        asm volatile("" : "+v"(off));
#endif
#pragma unroll
        for (int k = 0; k < 12; ++k) P[k] = s_pose[off + k];
      }
      int deleted;
      vec[2 * nf + f] = popt_cull_feature(b, job, f, P, thr_pt, thr_ls, vec + 3 * nf, deleted);
      del_pt += deleted == 1; del_ls += deleted == 2;
      vec[nf + f] = __longlong_as_double(0x7ff0000000000000LL);  // +inf sentinel for the refinement's init entries
    }
  }
  // deleted counts -> thread 0
  {
    const int lane = tid & 63, wave = tid >> 6;
    const double a = wave_sum_to_lane63((double)del_pt), c = wave_sum_to_lane63((double)del_ls);
    __syncthreads();
    if (lane == 63) { s_red[PO_RED * wave] = a; s_red[PO_RED * wave + 1] = c; }
    __syncthreads();
  }
  int n_del_pt = 0, n_del_ls = 0;
  for (int w = 0; w < PO_T / 64; ++w) { n_del_pt += (int)(s_red[PO_RED * w] + 0.5); n_del_ls += (int)(s_red[PO_RED * w + 1] + 0.5); }
  __syncthreads();

  // ---- refinement with inliers :469-563 (10-argument overload) ----
  int n_init = job.n_iter > 0 ? nf : 0;
  if (job.n_iter_ref >= 0) {
    popt_gn_loop<PO_T>(b, job, st, job_id, s_red, s_pose, s_ctl, job.n_iter_ref, 1, scale_pt, scale_ls, vec + nf, vec + 3 * nf, s_tot);
    if (job.n_iter_ref > 0) n_init += nf - n_del_pt - n_del_ls;
  }
  __syncthreads();

  PTICK(3);
  // ---- medians :244-249 ----
  auto kth_f64 = [&](const double* v, int n, int k) -> unsigned long long {
    return block_radix_select<PO_T, 64, unsigned long long>([&](int i) { return (unsigned long long)__double_as_longlong(v[i]); }, n, k, s_hist, s_sel);
  };
  unsigned long long mi = 0;
  if (n_init > 0) mi = kth_f64(vec, (job.n_iter_ref > 0) ? 2 * nf : nf, n_init / 2);   // unwritten refinement entries hold +inf and sort last
  const unsigned long long mf = kth_f64(vec + 2 * nf, nf, nf / 2);
  if (tid == 0) {
    for (int k = 0; k < 7; ++k) st->T[k] = s_pose[12 + k];
    if (poses) for (int k = 0; k < 7; ++k) poses[7 * job_id + k] = s_pose[12 + k];
    st->error_init = sqrt(__longlong_as_double((long long)mi)) * job.fx;
    st->error_final = sqrt(__longlong_as_double((long long)mf)) * job.fx;
    st->estimated_scale = scale_pt * job.fx;
    st->num_obs_pt = (unsigned long long)(np - n_del_pt);
    st->num_obs_ls = (unsigned long long)(ns - n_del_ls);
    st->iters = s_ctl[1]; st->iters_ref = s_ctl[2];
    st->pt_iters = (unsigned long long)(s_pose[27] + 0.5); st->seg_iters = (unsigned long long)(s_pose[28] + 0.5);
    if (b.work_key) b.work_key[job_id] = (int)fmin(s_pose[27] + s_pose[28] + 0.5, 2147483647.0);   // what this frame cost: the next launch's sort key
#ifdef PLSVO_TIMING
    { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); s_time[4] += t__ - s_tlast; }
    for (int k = 0; k < 8; ++k) st->phase_ticks[k] = s_time[k];
#endif
  }
}

// ================================================================================================================================
// Large batches: ONE 16-LANE DPP ROW PER FRAME, four frames per wave (pose_opt_rows_kernel).
//
// A frame is a few hundred features and a chain of serial steps (reduce -> 6x6 solve -> SE3 update -> medians): with a whole wave per
// frame the wave-cooperative solve and the update on lane 0 issue ~1400 wave-instructions per Gauss-Newton iteration for ONE frame --
// more than the feature pass itself (five 64-lane rounds) -- and the launch is bound by VALU issue, not by memory or occupancy
// (measured: 2.0 ms per 32768 frames at 198 VGPRs, two waves per SIMD).  Here a wave carries four frames:
//   * features: lane rl of a row takes features rl, rl + 16, ... of ITS frame (18 rounds for 280 features, 97 % of the lanes busy);
//   * reduction: row_reduce_scatter32 is a butterfly INSIDE the 16-lane DPP rows -- its result is already the per-frame total, no
//     cross-row combine, no workgroup barrier anywhere in the kernel;
//   * solve + update: lane 0 of every row solves its frame's 6x6 system serially in registers (lane_solve6: Eigen's pivot order is the
//     original diagonal sorted once, so the matrix is gathered from LDS already permuted and eliminated with static indices -- nothing
//     indexed at run time, no scratch) and applies the update: four frames per instruction stream instead of one;
//   * medians: radix select per row over a 256-bin LDS histogram per frame, bins scanned with row-level DPP shifts.
// Rows of a wave run in lock step: a loop runs to the largest trip count of its four frames (same feature counts in practice; the
// Gauss-Newton loops differ by an iteration or two).
// ================================================================================================================================
#define DPP_ROW_SHR_(n) (0x110 + (n))
template <int N>
__device__ __forceinline__ int row_shr_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR_(N), 0xf, 0xf, true); }   // lanes without a source receive 0
// inclusive prefix sum over the 16 lanes of a row
__device__ __forceinline__ int row_scan_incl_i32(int v) {
  v += row_shr_i32<1>(v); v += row_shr_i32<2>(v); v += row_shr_i32<4>(v); v += row_shr_i32<8>(v);
  return v;
}
// sum over the 16 lanes of a row, result in every lane of the row
__device__ __forceinline__ int row_sum_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, DPP_QUAD_XOR1, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_QUAD_XOR2, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_HALF_MIRROR, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_MIRROR, 0xf, 0xf, true);
  return v;
}

struct PoseRowLds {        // per frame (row) of the workgroup
  double pose[32];         // 0..8 R, 9..11 t, 12..18 model, 19..25 T_old, 26 chi2, 27 point-iterations, 28 line-iterations
  double tot[32];          // totals of the last iteration: 21 A, 6 b, chi2, #points, #segments
  double lu[36];           // covariance: LU factors
  double x[8];             // lane_solve6: un-permuted result
  int perm[8];
  int ctl[8];              // 0 break, 1 / 2 iterations of the two loops
  int sel[8];              // radix select: chosen bin, rank inside it, its count, (4,5) the surviving value
  int hist[PO_BINS];
};

// k-th smallest (0-based) of n non-negative IEEE values given as unsigned bit patterns of BITS bits, by the 16 lanes of ONE ROW; every
// row of the wave calls it together (`active`: this row takes part).  get(i) returns the pattern of element i.
template <int BITS, typename U, typename GET>
__device__ __forceinline__ U row_radix_select(GET get, int n, int k, bool active, int* hist, int* sel) {
  static_assert(PO_RADIX_BITS == 8 && BITS % 8 == 0, "16 lanes x 16 bins");
  const int rl = threadIdx.x & 15;
  U prefix = 0, mask = 0, result = 0;
  bool done = !active;
  for (int shift = BITS - 8; shift >= 0; shift -= 8) {
    if (!__any(!done)) break;
#pragma unroll
    for (int j = 0; j < 16; j += 4) *reinterpret_cast<uint4*>(hist + rl * 16 + j) = make_uint4(0u, 0u, 0u, 0u);
    wave_lds_fence();
    if (!done) for (int i = rl; i < n; i += 16) {
      const U v = get(i);
      if ((v & mask) == prefix) atomicAdd(&hist[(int)((v >> shift) & (U)255)], 1);
    }
    wave_lds_fence();
    int loc[16], local = 0;
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
      const uint4 h4 = *reinterpret_cast<const uint4*>(hist + rl * 16 + j);
      loc[j] = (int)h4.x; loc[j + 1] = (int)h4.y; loc[j + 2] = (int)h4.z; loc[j + 3] = (int)h4.w;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) local += loc[j];
    int cum = row_scan_incl_i32(local) - local;
    if (!done) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (k >= cum && k < cum + loc[j]) { sel[0] = rl * 16 + j; sel[1] = k - cum; sel[2] = loc[j]; }
        cum += loc[j];
      }
    }
    wave_lds_fence();
    bool single = false;
    if (!done) {
      prefix |= ((U)(unsigned)sel[0]) << shift;
      mask |= ((U)255) << shift;
      k = sel[1];
      single = sel[2] == 1 && shift > 0;   // one candidate left: it is the answer, the remaining digits need no histogram
      if (single) for (int i = rl; i < n; i += 16) {
        const U v = get(i);
        if ((v & mask) == prefix) { sel[4] = (int)(unsigned)(v & (U)0xffffffffu); sel[5] = (int)(unsigned)((unsigned long long)v >> 32); }
      }
    }
    wave_lds_fence();
    if (single) { result = (U)(((unsigned long long)(unsigned)sel[5] << 32) | (unsigned long long)(unsigned)sel[4]); done = true; }
    wave_lds_fence();   // sel is rewritten by the next pass
  }
  return done ? result : prefix;
}

// H x = rhs for a symmetric 6x6 H by ONE LANE: the arithmetic of wave_solve6_core (Gauss-Jordan in the pivot order of Eigen's LDLT, the
// zero-pivot rule selected by `flavour`, NaN / Inf propagation -- see there) with the system held in this lane's registers.  Eigen's
// order is the original |diagonal| sorted once (selection sort with Eigen's own transpositions: first maximum wins, numbers beat NaN),
// so the 27 inputs are gathered from LDS already permuted and every index below is a compile-time constant.  Columns already
// eliminated are dead (they never reach x) and are not updated.  tot: 21 A (upper, row-major) + 6 b in LDS; xs: 6 doubles of LDS.
__device__ __forceinline__ void lane_solve6(const double* tot, double* xs, double* x, int flavour) {
  double key[6]; int idx[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) { key[i] = fabs(tot[sym6_index(i, i)]); idx[i] = i; }
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    double bk = key[k]; int bi = idx[k], bp = k;
#pragma unroll
    for (int j = k + 1; j < 6; ++j) {
      const bool better = key[j] > bk || (bk != bk && key[j] == key[j]);
      if (better) { bk = key[j]; bi = idx[j]; bp = j; }
    }
    const double ok_ = key[k]; const int oi = idx[k];
#pragma unroll
    for (int j = k + 1; j < 6; ++j) if (bp == j) { key[j] = ok_; idx[j] = oi; }
    key[k] = bk; idx[k] = bi;
  }
  double M[6][6], r[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
#pragma unroll
    for (int c = 0; c < 6; ++c) M[a][c] = tot[sym6_index(idx[a], idx[c])];
    r[a] = tot[21 + idx[a]];
  }
  unsigned zero_piv = 0u;
  const double cutoff = flavour == 330 ? 0.0 : fabs(2.220446049250313e-16 * key[0]);
#pragma unroll
  for (int p = 0; p < 6; ++p) {
    const double piv = M[p][p];
    // Eigen 3.2: stop at "biggest_in_corner < cutoff"; no scaling unless |pivot| > cutoff; D^+ drops |d| <= max|D| eps
    if (!(key[p] < cutoff) && fabs(piv) > cutoff) {
      const double rinv = fast_rcp(piv);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        if (i == p) continue;
        const double f = M[i][p];
#pragma unroll
        for (int j = p + 1; j < 6; ++j) M[i][j] = fma(-(f * M[p][j]), rinv, M[i][j]);
        r[i] = fma(-(f * r[p]), rinv, r[i]);
      }
    } else {
      zero_piv |= 1u << p;
    }
  }
  const double tolerance = 1.0 / 1.7976931348623157e308;
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    const double dgi = M[a][a], m = r[a];
    double xi = (((zero_piv >> a) & 1u) || !(fabs(dgi) > tolerance)) ? ((dgi != dgi || m != m) ? (m + dgi) : 0.0) : m / dgi;
    if (cutoff > 1.7976931348623157e308) xi = (a == 5 || fabs(dgi) <= 1.7976931348623157e308) ? 0.0 : __builtin_nan("");   // infinite diagonal: see wave_solve6_core
    xs[idx[a]] = xi;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // same lane: LDS keeps program order
#pragma unroll
  for (int c = 0; c < 6; ++c) x[c] = xs[c];
}

// one GN loop (:103-195 / :469-563) for the four frames of the wave
__device__ __forceinline__ void rows_gn_loop(const PoseBatchDev& b, const PoseJobDev& job, PoseStateDev* st, int job_id, PoseRowLds& L, bool row_on,
                                             int n_iter, int phase, double scale_pt, double scale_ls, double* init_vec, const double* obs) {
  const int lane = threadIdx.x & 63, rl = lane & 15;
  const int nf = job.n_pts + job.n_seg;
  bool running = row_on && n_iter > 0;
  for (int iter = 0; __any(running); ++iter) {
    double acc[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
    if (running) {
      double P[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) P[k] = L.pose[k];
      for (int f = rl; f < nf; f += 16) popt_accumulate_feature(b, job, f, P, scale_pt, scale_ls, iter == 0, init_vec, obs, acc);
    }
    double out2[2];
    row_reduce_scatter32(acc, out2);     // every lane of the wave takes part (DPP); the row's totals are the frame's
    if (running) *reinterpret_cast<double2*>(L.tot + row_reduce_scatter32_index(lane)) = make_double2(out2[0], out2[1]);
    wave_lds_fence();
    if (running && rl == 0) {
      double dT[6];
      lane_solve6(L.tot, L.x, dT, job.ldlt_flavour);                          // A.ldlt().solve(b) :170
      const double new_chi2 = L.tot[27];
      L.pose[27] += L.tot[28]; L.pose[28] += L.tot[29];
      L.ctl[1 + phase] += 1;
      SE3d model = se3_load(L.pose + 12);
      int accepted = 1, brk = 0;
      if ((iter > 0 && new_chi2 > L.pose[26]) || isnan(dT[0])) {              // :173-180
        model = se3_load(L.pose + 19); accepted = 0; brk = 1;
      } else {
        const SE3d Tn = se3_mul_dev(se3_exp_dev(dT), model);                   // :183 left update
        se3_store(model, L.pose + 19);
        model = Tn; L.pose[26] = new_chi2;
        if (norm_max6(dT) <= 0.0000000001) brk = 1;                            // EPS, global.h:99
      }
      if (iter + 1 >= n_iter) brk = 1;
      se3_store(model, L.pose + 12);
      quat_to_matrix(model.q, L.pose); L.pose[9] = model.t[0]; L.pose[10] = model.t[1]; L.pose[11] = model.t[2];
      L.ctl[0] = brk;
      if (b.log) {
        const int lc = st->log_count;
        if (lc < b.log_cap) {
          plsvo_poseopt_iterlog* r = b.log + (size_t)job_id * b.log_cap + lc;
          r->phase = phase; r->iter = iter; r->accepted = accepted; r->reserved0 = 0; r->new_chi2 = new_chi2;
          for (int k = 0; k < 6; ++k) r->dT[k] = dT[k];
          se3_store(model, r->T_after);
          for (int i = 0; i < 6; ++i) for (int jj = 0; jj < 6; ++jj) r->A[i * 6 + jj] = L.tot[sym6_index(i, jj)];
          for (int k = 0; k < 6; ++k) r->b[k] = L.tot[21 + k];
        }
        st->log_count = lc + 1;
      }
    }
    wave_lds_fence();
    if (running && L.ctl[0]) running = false;
  }
}

__global__ __launch_bounds__(64) PLSVO_PO_ROWS_OCC void pose_opt_rows_kernel(PoseBatchDev b, double* poses) {
  __shared__ __align__(16) PoseRowLds s_rows[4];
  const int lane = threadIdx.x & 63, row = lane >> 4, rl = lane & 15;
  const int job_raw = blockIdx.x * 4 + row;
  const bool row_valid = job_raw < b.n_jobs;
  // a re-run staged batch comes sorted by the feature-iterations of its last launch (plsvo_capi.hip::plsvo_poseopt_run): the four frames of a
  // wave run in lock step until the LAST of them stops, and the iteration counts are bimodal (4-6, or all 10) -- in an arbitrary order three
  // waves of four hold a ten-iteration frame and run ten; sorted, the rows of a wave stop together
  const int job_id = row_valid ? (b.order ? b.order[job_raw] : job_raw) : b.n_jobs - 1;
  const PoseJobDev job = b.jobs[job_id];
  PoseStateDev* st = b.state + job_id;
  PoseRowLds& L = s_rows[row];
  const int np = job.n_pts, ns = job.n_seg, nf = np + ns;

  // scratch: floats [0,np) point errors, [np, np+ns) line errors; doubles [0,nf) init (first loop), [nf,2nf) init (refinement), [2nf,3nf) final, [3nf, 3nf + 2 np) the points' normalised observations
  const size_t fbase = (size_t)job.pt_off + (size_t)job.seg_off;
  float* errs = b.scratch_f32 + fbase;
  double* vec = b.scratch_f64 + 5 * fbase;

  if (row_valid && rl == 0) {
    SE3d m = se3_load(job.T0);
    se3_store(m, L.pose + 12); se3_store(m, L.pose + 19); L.pose[26] = 0.0; L.pose[27] = 0.0; L.pose[28] = 0.0;
    for (int k = 0; k < 32; ++k) L.tot[k] = 0.0;
    quat_to_matrix(m.q, L.pose); L.pose[9] = m.t[0]; L.pose[10] = m.t[1]; L.pose[11] = m.t[2];
    L.ctl[0] = 0; L.ctl[1] = 0; L.ctl[2] = 0;
    st->log_count = 0; st->status = 0; st->iters = 0; st->iters_ref = 0; st->pt_iters = 0; st->seg_iters = 0;
    st->num_obs_pt = 0; st->num_obs_ls = 0; st->estimated_scale = 0; st->error_init = 0; st->error_final = 0;
    for (int k = 0; k < 36; ++k) st->cov[k] = 0.0;
    for (int k = 0; k < 7; ++k) st->T[k] = job.T0[k];
    for (int k = 0; k < 8; ++k) st->phase_ticks[k] = 0;
    if (nf == 0) { st->status = 1; if (poses) for (int k = 0; k < 7; ++k) poses[7 * job_id + k] = job.T0[k]; }   // errors.empty() :88-89
    if (b.work_key) b.work_key[job_id] = 0;
  }
  const bool row_on = row_valid && nf > 0;
  if (row_on) {
    for (int i = rl; i < np; i += 16) b.pt_keep[job.pt_off + i] = 1;
    for (int s = rl; s < ns; s += 16) b.seg_keep[job.seg_off + s] = 1;
  }
  wave_lds_fence();

  // ---- scale pass :57-95 ----
  if (row_on) {
    double P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = L.pose[k];
    for (int f = rl; f < nf; f += 16) errs[f] = popt_scale_error(b, job, f, P, vec + 3 * nf);
  }
  wave_lds_fence();   // (global scratch written and read by lanes of this wave only)
  // MAD scale = 1.48f * median (float).  Zero points: the reference is undefined (:70); we define 1.0.
  double scale_pt = 1.0, scale_ls = 1.0;
  {
    const uint32_t m_pt = row_radix_select<32, uint32_t>([&](int i) { return (uint32_t)__float_as_uint(errs[i]); }, np, np / 2, row_on && np > 0, L.hist, L.sel);
    const uint32_t m_ls = row_radix_select<32, uint32_t>([&](int i) { return (uint32_t)__float_as_uint(errs[np + i]); }, ns, ns / 2, row_on && ns > 0, L.hist, L.sel);
    if (np > 0) scale_pt = (double)__fmul_rn(1.48f, __uint_as_float(m_pt));
    if (ns > 0) scale_ls = (double)__fmul_rn(1.48f, __uint_as_float(m_ls));
  }

  // ---- first GN loop ----
  if (row_on && job.n_iter <= 0) for (int f = rl; f < nf; f += 16) vec[f] = __longlong_as_double(0x7ff0000000000000LL);
  rows_gn_loop(b, job, st, job_id, L, row_on, job.n_iter, 0, scale_pt, scale_ls, vec, vec + 3 * nf);

  // ---- covariance :197-199 (from the last assembled A, even if that iteration was rolled back) ----
  if (row_on) {
    const double f2 = job.fx * job.fx;
    for (int t = rl; t < 36; t += 16) L.lu[t] = L.tot[sym6_index(t / 6, t % 6)] * f2;
  }
  wave_lds_fence();
  if (row_on && rl == 0) lu6_lds(L.lu, L.perm);
  wave_lds_fence();
  if (row_on && rl < 6) inv6_column_lds(L.lu, L.perm, rl, st->cov);

  // ---- cull :201-242 ----
  const double thr_pt = job.reproj_thresh / job.fx;
  const double thr_ls = thr_pt * scale_ls / scale_pt;
  int del_pt = 0, del_ls = 0;
  if (row_on) {
    double P[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) P[k] = L.pose[k];
    for (int f = rl; f < nf; f += 16) {
      int deleted;
      vec[2 * nf + f] = popt_cull_feature(b, job, f, P, thr_pt, thr_ls, vec + 3 * nf, deleted);
      del_pt += deleted == 1; del_ls += deleted == 2;
      vec[nf + f] = __longlong_as_double(0x7ff0000000000000LL);  // +inf sentinel for the refinement's init entries
    }
  }
  const int n_del_pt = row_sum_i32(del_pt), n_del_ls = row_sum_i32(del_ls);
  wave_lds_fence();   // keep flags and vec visible to the row

  // ---- refinement with inliers :469-563 (10-argument overload) ----
  int n_init = job.n_iter > 0 ? nf : 0;
  rows_gn_loop(b, job, st, job_id, L, row_on && job.n_iter_ref >= 0, job.n_iter_ref, 1, scale_pt, scale_ls, vec + nf, vec + 3 * nf);
  if (job.n_iter_ref > 0) n_init += nf - n_del_pt - n_del_ls;
  wave_lds_fence();

  // ---- medians :244-249 ----
  const unsigned long long mi = row_radix_select<64, unsigned long long>([&](int i) { return (unsigned long long)__double_as_longlong(vec[i]); },
                                                                         (job.n_iter_ref > 0) ? 2 * nf : nf, n_init / 2, row_on && n_init > 0, L.hist, L.sel);   // unwritten refinement entries hold +inf and sort last
  const unsigned long long mf = row_radix_select<64, unsigned long long>([&](int i) { return (unsigned long long)__double_as_longlong(vec[2 * nf + i]); },
                                                                         nf, nf / 2, row_on, L.hist, L.sel);
  if (row_on && rl == 0) {
    for (int k = 0; k < 7; ++k) st->T[k] = L.pose[12 + k];
    if (poses) for (int k = 0; k < 7; ++k) poses[7 * job_id + k] = L.pose[12 + k];
    st->error_init = n_init > 0 ? sqrt(__longlong_as_double((long long)mi)) * job.fx : 0.0;
    st->error_final = sqrt(__longlong_as_double((long long)mf)) * job.fx;
    st->estimated_scale = scale_pt * job.fx;
    st->num_obs_pt = (unsigned long long)(np - n_del_pt);
    st->num_obs_ls = (unsigned long long)(ns - n_del_ls);
    st->iters = L.ctl[1]; st->iters_ref = L.ctl[2];
    st->pt_iters = (unsigned long long)(L.pose[27] + 0.5); st->seg_iters = (unsigned long long)(L.pose[28] + 0.5);
    if (b.work_key) b.work_key[job_id] = (int)fmin(L.pose[27] + L.pose[28] + 0.5, 2147483647.0);
  }
}

hipError_t launch_pose_opt(const PoseBatchDev& b, double* d_poses, int threads, hipStream_t stream) {
  switch (threads) {
    case 16: hipLaunchKernelGGL(pose_opt_rows_kernel, dim3((b.n_jobs + 3) / 4), dim3(64), 0, stream, b, d_poses); break;   // a 16-lane row per frame
    case 64: hipLaunchKernelGGL((pose_opt_kernel<64>), dim3(b.n_jobs), dim3(64), 0, stream, b, d_poses); break;
    case 256: hipLaunchKernelGGL((pose_opt_kernel<256>), dim3(b.n_jobs), dim3(256), 0, stream, b, d_poses); break;
    case 512: hipLaunchKernelGGL((pose_opt_kernel<512>), dim3(b.n_jobs), dim3(512), 0, stream, b, d_poses); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace plsvo_hip
