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

template <int PO_T>
__device__ void popt_gn_loop(const PoseBatchDev& b, const PoseJobDev& job, PoseStateDev* st, int job_id, double* s_red,
                             double* s_pose, int* s_ctl, int n_iter, int phase, double scale_pt, double scale_ls,
                             double* init_vec, double* s_tot) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int np = job.n_pts, ns = job.n_seg, nf = np + ns;
  for (int iter = 0; iter < n_iter; ++iter) {
    double acc[32];   // 0..20 A (upper, row-major), 21..26 b, 27 chi2, 28 #points, 29 #segments, 30..31 unused
#pragma unroll
    for (int k = 0; k < 32; ++k) acc[k] = 0.0;
    const double R0 = s_pose[0], R1 = s_pose[1], R2 = s_pose[2], R3 = s_pose[3], R4 = s_pose[4], R5 = s_pose[5],
                 R6 = s_pose[6], R7 = s_pose[7], R8 = s_pose[8], t0 = s_pose[9], t1 = s_pose[10], t2 = s_pose[11];
    for (int f = tid; f < nf; f += PO_T) {
      double J[12], e0, e1, weight;
      if (f < np) {
        const int i = job.pt_off + f;
        if (!b.pt_keep[i]) continue;
        const double x = b.pt_pos[3 * i], y = b.pt_pos[3 * i + 1], z = b.pt_pos[3 * i + 2];
        const double xyz[3] = { R0 * x + R1 * y + R2 * z + t0, R3 * x + R4 * y + R5 * z + t1, R6 * x + R7 * y + R8 * z + t2 };
        jacobian_xyz2uv(xyz, J);
        const double fz = b.pt_f[3 * i + 2];
        e0 = b.pt_f[3 * i] / fz - xyz[0] / xyz[2];
        e1 = b.pt_f[3 * i + 1] / fz - xyz[1] / xyz[2];
        const double sic = 1.0 / (double)(1 << b.pt_level[i]);
        e0 *= sic; e1 *= sic;
        if (iter == 0) init_vec[f] = e0 * e0 + e1 * e1;
#pragma unroll
        for (int k = 0; k < 12; ++k) J[k] *= sic;
        weight = (double)tukey_weight((float)(sqrt(e0 * e0 + e1 * e1) / scale_pt));
        acc[28] += 1.0;
      } else {
        const int s = job.seg_off + (f - np);
        if (!b.seg_keep[s]) continue;
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
        const double sic = 1.0 / (double)(1 << b.seg_level[s]);
        e0 = (double)ds * sic; e1 = (double)de * sic;
        if (iter == 0) init_vec[f] = e0 * e0 + e1 * e1;
        const double en = sqrt(e0 * e0 + e1 * e1);
        const double ks = sic * (double)ds / en;  // same factor for both rows (:156-157)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
          J[c] = l0 * (Js[c] * ks) + l1 * (Js[6 + c] * ks);
          J[6 + c] = l0 * (Je[c] * ks) + l1 * (Je[6 + c] * ks);
        }
        weight = (double)tukey_weight((float)(en / scale_ls));
        acc[29] += 1.0;
      }
      int k = 0;
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int jj = i; jj < 6; ++jj) { acc[k] += (J[i] * J[jj] + J[6 + i] * J[6 + jj]) * weight; ++k; }
#pragma unroll
      for (int i = 0; i < 6; ++i) acc[21 + i] -= (J[i] * e0 + J[6 + i] * e1) * weight;
      acc[27] += (e0 * e0 + e1 * e1) * weight;
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
__global__ __launch_bounds__(PO_T) void pose_opt_kernel(PoseBatchDev b, double* poses) {
  const int job_id = blockIdx.x;
  const PoseJobDev job = b.jobs[job_id];
  PoseStateDev* st = b.state + job_id;
  const int tid = threadIdx.x;
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
  // [nf,2nf) init (refinement), [2nf,3nf) final
  const size_t fbase = (size_t)job.pt_off + (size_t)job.seg_off;
  float* errs = b.scratch_f32 + fbase;
  double* vec = b.scratch_f64 + 3 * fbase;

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
  {
    const double R0 = s_pose[0], R1 = s_pose[1], R2 = s_pose[2], R3 = s_pose[3], R4 = s_pose[4], R5 = s_pose[5],
                 R6 = s_pose[6], R7 = s_pose[7], R8 = s_pose[8], t0 = s_pose[9], t1 = s_pose[10], t2 = s_pose[11];
    for (int f = tid; f < nf; f += PO_T) {
      if (f < np) {
        const int i = job.pt_off + f;
        const double x = b.pt_pos[3 * i], y = b.pt_pos[3 * i + 1], z = b.pt_pos[3 * i + 2];
        const double xc = R0 * x + R1 * y + R2 * z + t0, yc = R3 * x + R4 * y + R5 * z + t1, zc = R6 * x + R7 * y + R8 * z + t2;
        const double fz = b.pt_f[3 * i + 2];
        double e0 = b.pt_f[3 * i] / fz - xc / zc, e1 = b.pt_f[3 * i + 1] / fz - yc / zc;
        const double sic = 1.0 / (double)(1 << b.pt_level[i]);
        e0 *= sic; e1 *= sic;
        errs[f] = (float)sqrt(e0 * e0 + e1 * e1);
      } else {
        const int s = job.seg_off + (f - np);
        const double sx = b.seg_spos[3 * s], sy = b.seg_spos[3 * s + 1], sz = b.seg_spos[3 * s + 2];
        const double ex = b.seg_epos[3 * s], ey = b.seg_epos[3 * s + 1], ez = b.seg_epos[3 * s + 2];
        const double xs0 = R0 * sx + R1 * sy + R2 * sz + t0, xs1 = R3 * sx + R4 * sy + R5 * sz + t1, xs2 = R6 * sx + R7 * sy + R8 * sz + t2;
        const double xe0 = R0 * ex + R1 * ey + R2 * ez + t0, xe1 = R3 * ex + R4 * ey + R5 * ez + t1, xe2 = R6 * ex + R7 * ey + R8 * ez + t2;
        const double l0 = b.seg_line[3 * s], l1 = b.seg_line[3 * s + 1], l2 = b.seg_line[3 * s + 2];
        const float es = (float)(l0 * (xs0 / xs2) + l1 * (xs1 / xs2) + l2 * 1.0);   // not scaled by the level (:84-87)
        const float ee = (float)(l0 * (xe0 / xe2) + l1 * (xe1 / xe2) + l2 * 1.0);
        errs[f] = norm2_f32(es, ee);
      }
    }
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
  popt_gn_loop<PO_T>(b, job, st, job_id, s_red, s_pose, s_ctl, job.n_iter, 0, scale_pt, scale_ls, vec, s_tot);

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
    const double R0 = s_pose[0], R1 = s_pose[1], R2 = s_pose[2], R3 = s_pose[3], R4 = s_pose[4], R5 = s_pose[5],
                 R6 = s_pose[6], R7 = s_pose[7], R8 = s_pose[8], t0 = s_pose[9], t1 = s_pose[10], t2 = s_pose[11];
    for (int f = tid; f < nf; f += PO_T) {
      double e0, e1;
      if (f < np) {
        const int i = job.pt_off + f;
        const double x = b.pt_pos[3 * i], y = b.pt_pos[3 * i + 1], z = b.pt_pos[3 * i + 2];
        const double xc = R0 * x + R1 * y + R2 * z + t0, yc = R3 * x + R4 * y + R5 * z + t1, zc = R6 * x + R7 * y + R8 * z + t2;
        const double fz = b.pt_f[3 * i + 2];
        e0 = b.pt_f[3 * i] / fz - xc / zc; e1 = b.pt_f[3 * i + 1] / fz - yc / zc;
        const double sic = 1.0 / (double)(1 << b.pt_level[i]);
        e0 *= sic; e1 *= sic;
        if (sqrt(e0 * e0 + e1 * e1) > thr_pt) { b.pt_keep[i] = 0; ++del_pt; }
      } else {
        const int s = job.seg_off + (f - np);
        const double sx = b.seg_spos[3 * s], sy = b.seg_spos[3 * s + 1], sz = b.seg_spos[3 * s + 2];
        const double ex = b.seg_epos[3 * s], ey = b.seg_epos[3 * s + 1], ez = b.seg_epos[3 * s + 2];
        const double xs0 = R0 * sx + R1 * sy + R2 * sz + t0, xs1 = R3 * sx + R4 * sy + R5 * sz + t1, xs2 = R6 * sx + R7 * sy + R8 * sz + t2;
        const double xe0 = R0 * ex + R1 * ey + R2 * ez + t0, xe1 = R3 * ex + R4 * ey + R5 * ez + t1, xe2 = R6 * ex + R7 * ey + R8 * ez + t2;
        const double l0 = b.seg_line[3 * s], l1 = b.seg_line[3 * s + 1], l2 = b.seg_line[3 * s + 2];
        const double sic = 1.0 / (double)(1 << b.seg_level[s]);
        e0 = (l0 * (xs0 / xs2) + l1 * (xs1 / xs2) + l2 * 1.0) * sic;     // doubles here, no float truncation (:229)
        e1 = (l0 * (xe0 / xe2) + l1 * (xe1 / xe2) + l2 * 1.0) * sic;
        if (sqrt(e0 * e0 + e1 * e1) > thr_ls) { b.seg_keep[s] = 0; ++del_ls; }
      }
      vec[2 * nf + f] = e0 * e0 + e1 * e1;
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
    popt_gn_loop<PO_T>(b, job, st, job_id, s_red, s_pose, s_ctl, job.n_iter_ref, 1, scale_pt, scale_ls, vec + nf, s_tot);
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
#ifdef PLSVO_TIMING
    { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); s_time[4] += t__ - s_tlast; }
    for (int k = 0; k < 8; ++k) st->phase_ticks[k] = s_time[k];
#endif
  }
}

hipError_t launch_pose_opt(const PoseBatchDev& b, double* d_poses, int threads, hipStream_t stream) {
  switch (threads) {
    case 64: hipLaunchKernelGGL((pose_opt_kernel<64>), dim3(b.n_jobs), dim3(64), 0, stream, b, d_poses); break;
    case 256: hipLaunchKernelGGL((pose_opt_kernel<256>), dim3(b.n_jobs), dim3(256), 0, stream, b, d_poses); break;
    case 512: hipLaunchKernelGGL((pose_opt_kernel<512>), dim3(b.n_jobs), dim3(512), 0, stream, b, d_poses); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace plsvo_hip
