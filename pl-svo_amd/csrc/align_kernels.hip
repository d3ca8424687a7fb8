// align_kernels.hip -- sparse image alignment on gfx950 (CDNA4): one workgroup per (ref,cur) frame pair,
// ONE launch for all pyramid levels; the coarse-to-fine loop and every Gauss-Newton iteration run on the device.
//
// Replaces (reference file:line):
//   SparseImgAlign::precomputeGaussNewtonParamsPoints/Segments   src/sparse_img_align.cpp:195-268, 270-378
//   SparseImgAlign::computeGaussNewtonParamsPoints/Segments      src/sparse_img_align.cpp:380-502, 504-695
//   SparseImgAlign::computeResiduals / solve / update            src/sparse_img_align.cpp:112-193, 697-710
//   [ext] vk::NLLSSolver<6,SE3>::optimizeGaussNewton             (GN loop, chi2 rollback, eps stop)
//
// Design (not a translation of the CPU loops):
//   * features of a level are flattened into a PATCH TABLE (points: 1 patch, segments: N samples),
//     built on the device by a block-wide scan; a patch is 4x4 pixels;
//   * the level image is gathered through L2 (a 5x5 window costs two aligned dword reads + v_alignbyte per image
//     row); LDS holds only the per-patch tables (~15 KB per 128-thread workgroup), so four workgroups share a CU and
//     hide each other's serial solve/update tails.  Staging the level image in LDS (77 KB at level 1 -> one
//     workgroup per CU) and software-pipelining the phase-1 loads were both measured and lost;
//   * phase 1: 2 lanes per patch, two adjacent patch rows (8 pixels) per lane.  The 6-vector Jacobian of a pixel is
//     J = fs * (dx * r0 + dy * r1) with r0, r1 the two rows of the 2x6 projection Jacobian of the PATCH,
//     so sum_pix w J J^T = fs^2 (A r0 r0^T + B (r0 r1^T + r1 r0^T) + C r1 r1^T) with
//     A = sum w dx^2, B = sum w dx dy, C = sum w dy^2, and sum_pix w res J = fs (D r0 + E r1).
//     Only the five scalars A..E are accumulated per pixel (double); the 6x6 expansion is done once
//     per patch, one lane per patch.  The cached Jacobian therefore shrinks from 16x6 doubles to
//     16x2 floats per patch (dx, dy) and the per-pixel work from 27 to 5 double FMAs.
//   * per-line re-weighting (H += H_line * w / r, Jres += Jres_line * w, cull if r >= 200 or a sample
//     leaves the image, src/sparse_img_align.cpp:640-688) needs the line's mean |residual| first: each
//     sample's sum|res| goes to LDS, then every sample lane recomputes its line's total in fixed order;
//   * reductions are fixed-shape (DPP inside a wave, LDS across waves): results are deterministic.
//
// Numerics: image interpolation and residuals in float with the reference's operation order and NO
// fma contraction (__fmul_rn/__fadd_rn); geometry and all accumulators in double.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "plsvo_dev.hpp"
#include "plsvo_math.hpp"
#include "plsvo_wave.hpp"

namespace plsvo_hip {

// ------------------------------------------------------------------------------------------------
// image gather: bytes [off, off+NB) of a u8 image as floats, via aligned dword reads + v_alignbyte.
// Works for LDS and global pointers; the image allocation is padded so the over-read stays inside it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_row5(const uint8_t* img, int off, float* o5) {
  const int a = off & ~3, sh = off & 3;
  const uint32_t d0 = *reinterpret_cast<const uint32_t*>(img + a);
  const uint32_t d1 = *reinterpret_cast<const uint32_t*>(img + a + 4);
  const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, sh);  // bytes off..off+3
  const uint32_t b4 = (d1 >> (8 * sh)) & 0xffu;                 // byte off+4
  o5[0] = (float)(w0 & 0xffu); o5[1] = (float)((w0 >> 8) & 0xffu);
  o5[2] = (float)((w0 >> 16) & 0xffu); o5[3] = (float)(w0 >> 24);
  o5[4] = (float)b4;
}
__device__ __forceinline__ void load_row7(const uint8_t* img, int off, float* o7) {
  const int a = off & ~3, sh = off & 3;
  const uint32_t d0 = *reinterpret_cast<const uint32_t*>(img + a);
  const uint32_t d1 = *reinterpret_cast<const uint32_t*>(img + a + 4);
  const uint32_t d2 = *reinterpret_cast<const uint32_t*>(img + a + 8);
  const uint32_t w0 = __builtin_amdgcn_alignbyte(d1, d0, sh);  // bytes off..off+3
  const uint32_t w1 = __builtin_amdgcn_alignbyte(d2, d1, sh);  // bytes off+4..off+7
  o7[0] = (float)(w0 & 0xffu); o7[1] = (float)((w0 >> 8) & 0xffu);
  o7[2] = (float)((w0 >> 16) & 0xffu); o7[3] = (float)(w0 >> 24);
  o7[4] = (float)(w1 & 0xffu); o7[5] = (float)((w1 >> 8) & 0xffu);
  o7[6] = (float)((w1 >> 16) & 0xffu);
}

// wTL*a + wTR*b + wBL*c + wBR*d, evaluated left to right in float without contraction
// (src/sparse_img_align.cpp:251, 458, 620)
__device__ __forceinline__ float bilinear(float wTL, float wTR, float wBL, float wBR, float a, float b, float c, float d) {
  return __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(wTL, a), __fmul_rn(wTR, b)), __fmul_rn(wBL, c)), __fmul_rn(wBR, d));
}

// (float)(1.0 / (1.0 + (double)a)) for a float a >= 0 -- the reference's robust weight
// (src/sparse_img_align.cpp:479) -- without a double division: 1 + a is split exactly into s_hi + s_lo
// (two-sum), y0 = rcp(s_hi), the exact residual e = 1 - s_hi*y0 comes from one fma, and
// y0 + y0*(e - s_lo*y0) is the quotient to ~1e-14 relative before the final rounding, i.e. the correctly
// rounded float except when the quotient sits within ~1e-7 ulp of a rounding boundary.
__device__ __forceinline__ float robust_weight(float a) {
  const float s_hi = __fadd_rn(1.0f, a);
  const float bv = __fsub_rn(s_hi, 1.0f);
  const float s_lo = __fadd_rn(__fsub_rn(1.0f, __fsub_rn(s_hi, bv)), __fsub_rn(a, bv));  // exact: (1 + a) - s_hi
  const float y0 = __builtin_amdgcn_rcpf(s_hi);
  const float e = __fmaf_rn(-s_hi, y0, 1.0f);
  const float c = __fmaf_rn(-s_lo, y0, e);
  return __fmaf_rn(y0, c, y0);
}

// Patch::setPosition + computeInterpWeights (src/feature.cpp:189-208): position as float, weights
// computed in double and stored as float
struct PatchW { int ui, vi; float wTL, wTR, wBL, wBR; };
__device__ __forceinline__ PatchW patch_weights(float u, float v) {
  PatchW p;
  const float fu = floorf(u), fv = floorf(v);
  p.ui = (int)fu; p.vi = (int)fv;
  const float su = u - fu, sv = v - fv;
  p.wTL = (float)((1.0 - (double)su) * (1.0 - (double)sv));
  p.wTR = (float)((double)su * (1.0 - (double)sv));
  p.wBL = (float)((1.0 - (double)su) * (double)sv);
  p.wBR = (float)((double)su * (double)sv);
  return p;
}

// block-wide exclusive scan of n ints in LDS (in place); returns the total.  s_tmp: T/64 + 1 ints.
template <int T>
__device__ int block_exclusive_scan(int* a, int n, int* s_tmp) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = (n + T - 1) / T;
  const int beg = min(tid * chunk, n), end = min(beg + chunk, n);
  int local = 0;
  for (int i = beg; i < end; ++i) local += a[i];
  int incl = local;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  if (lane == 63) s_tmp[wave] = incl;
  __syncthreads();
  int wave_off = 0, total = 0;
  for (int w = 0; w < T / 64; ++w) { const int v = s_tmp[w]; if (w < wave) wave_off += v; total += v; }
  int running = wave_off + incl - local;
  for (int i = beg; i < end; ++i) { const int v = a[i]; a[i] = running; running += v; }
  __syncthreads();
  return total;
}

// ------------------------------------------------------------------------------------------------
// init: model <- T0, alive <- alive_in, solver reset() ([ext] vk::NLLSSolver::reset)
// ------------------------------------------------------------------------------------------------
__global__ void align_init_kernel(AlignBatchDev b) {
  const int j = blockIdx.x;
  const AlignJobDev job = b.jobs[j];
  AlignStateDev* st = b.state + j;
  for (int s = threadIdx.x; s < job.n_seg; s += blockDim.x)
    b.seg_alive[job.seg_off + s] = b.seg_alive_in ? (b.seg_alive_in[job.seg_off + s] != 0) : 1;
  if (threadIdx.x == 0) {
    for (int k = 0; k < 7; ++k) st->T[k] = b.T0[7 * j + k];
    st->chi2 = 1e10; st->n_meas = 0; st->stop = 0; st->log_count = 0; st->error = 0;
    for (int k = 0; k < 36; ++k) st->H[k] = 0.0;
    for (int k = 0; k < PLSVO_MAX_LEVELS; ++k) st->iters[k] = 0;
    st->patch_levels = 0; st->patch_iters = 0;
    for (int k = 0; k < 8; ++k) st->phase_ticks[k] = 0;
  }
}

// ------------------------------------------------------------------------------------------------
// one pyramid level of SparseImgAlign::run for every job of the batch
// ------------------------------------------------------------------------------------------------
// optional per-phase timing (compile with -DPLSVO_TIMING): thread 0 accumulates s_memtime deltas
#ifdef PLSVO_TIMING
#define TICK(slot) do { if (tid == 0) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); s_time[slot] += t__ - s_tlast; s_tlast = t__; } } while (0)
#else
#define TICK(slot) do { } while (0)
#endif

#define RED_N 32  // doubles per wave in the block reduction (21 H + 6 Jres + chi2 + 2 counters + pad)

// what one lane of phase 1 loads for its half patch (2 lanes per patch, two adjacent patch rows = 8 pixels each: the
// middle image row is shared, one cross-lane step instead of two, address and weight work amortised over 8 pixels;
// measured 11 % faster than 4 lanes per patch with one row each)
struct P1Fetch2 {
  float2 uv;
  uint32_t r0a, r0b, r1a, r1b, r2a, r2b;   // aligned dword pairs covering 5 bytes of image rows 2h, 2h+1, 2h+2 of the 5x5 window
  int off;                                 // byte offset of the first of them (the shifts follow from it)
  float4 vr0, vx0, vy0, vr1, vx1, vy1;     // cached reference intensity and gradient of patch rows 2h, 2h+1
  int flags;                               // bit 0: live (in frame, line alive), bit 1: point patch
};

// ------------------------------------------------------------------------------------------------
// SparseImgAlign::run for every job of the batch: levels [level_hi .. level_lo] of each job's range
// ------------------------------------------------------------------------------------------------
#ifndef PLSVO_MIN_WAVES
#define PLSVO_MIN_WAVES 2   // measured: capping VGPRs at 128 (4 waves/SIMD) spills and loses to 2 unspilled waves/SIMD
#endif
// The per-iteration patch sums (6 doubles per patch) round-trip through L2 between phase 1 and phase 2.  Keeping them in
// LDS was measured three ways on MI355X: all of them (one workgroup per CU fewer: -8 %), as many as fit beside the tables
// at four workgroups per CU (no difference), none (this code).
template <int T>
__global__ __launch_bounds__(T, PLSVO_MIN_WAVES) void align_fused_kernel(AlignBatchDev b, int cap, int fcap, int level_hi, int level_lo) {
  // longest-processing-time-first: the hardware hands out workgroups in blockIdx order, so the jobs with the most patches
  // start first and the launch tail is made of the cheapest frames
  const int job_id = b.order ? b.order[blockIdx.x] : (int)blockIdx.x;
  const AlignJobDev job = b.jobs[job_id];
  if (job.skip) return;
  const int lv_first = min(job.max_level, level_hi), lv_last = max(job.min_level, level_lo);
  if (lv_first < lv_last) return;
  AlignStateDev* st = b.state + job_id;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int G = T / 4;  // 4-lane patch groups per workgroup (reference-patch precompute: one patch row per lane)
  const int grp = tid >> 2, row = tid & 3;

  extern __shared__ __align__(16) unsigned char smem[];
  double* s_red = reinterpret_cast<double*>(smem);                       // RED_N * (T/64)
  double* s_pose = s_red + RED_N * (T / 64);                             // 0..8 R, 9..11 t, 12..18 model, 19..25 old model, 26 chi2_, 27 #evals
  double* s_tot = s_pose + 32;                                           // block totals of the last iteration: 21 H, 6 Jres, chi2, n_meas, evals
  int* s_ctl = reinterpret_cast<int*>(s_tot + 32);                       // 0 break, 1 stop, 2 iterations done, 3 error, 4.. scan tmp
  float2* s_uv = reinterpret_cast<float2*>(s_ctl + 32);                  // cap
  int2* s_meta = reinterpret_cast<int2*>(s_uv + cap);                    // cap
  float* s_abs = reinterpret_cast<float*>(s_meta + cap);                 // cap
  int* s_dead = reinterpret_cast<int*>(s_abs + cap);                     // fcap: per segment, "culled at this level"
  int* s_cnt = s_dead + fcap;                                            // fcap + 4: per-feature patch count / offset

#ifdef PLSVO_TIMING
  __shared__ unsigned long long s_time[8];
  __shared__ unsigned long long s_tlast;
  if (tid == 0) { for (int k = 0; k < 8; ++k) s_time[k] = 0; s_tlast = __builtin_amdgcn_s_memtime(); }
#endif
  if (tid == 0) {
    for (int k = 0; k < 7; ++k) { s_pose[12 + k] = st->T[k]; s_pose[19 + k] = st->T[k]; }
    s_pose[26] = st->chi2;
    for (int k = 0; k < 32; ++k) s_tot[k] = 0.0;
    s_ctl[1] = st->stop; s_ctl[3] = 0;
  }
  const size_t pbase = (size_t)job.patch_off;
  const int nfeat = job.n_pts + job.n_seg;
  double* const part = b.partial + 6 * pbase;        // per-iteration patch sums
  double* const pxyz = b.patch_xyz + 3 * pbase;      // 3-D point of every patch (ref frame)

  for (int level = lv_first; level >= lv_last; --level) {
    // (level geometry is recomputed instead of indexing the kernel-argument arrays with a run-time level,
    //  which would force the whole argument struct into scratch memory)
    const int W = job.width >> level, Hh = job.height >> level;
    const unsigned int lvl_off = pyr_level_offset(job.width, job.height, level);
    const uint8_t* ref_img = b.pyr.base + (size_t)job.ref_slot * b.pyr.slot_bytes + lvl_off;
    const uint8_t* cur_img = b.pyr.base + (size_t)job.cur_slot * b.pyr.slot_bytes + lvl_off;
    __syncthreads();  // previous level done with every LDS table

    if (tid == 0) { s_ctl[0] = 0; s_ctl[2] = 0; s_pose[27] = 0.0; }

    // ---- patch table: count, scan, emit ----
    const double scale = 1.0 / (double)(1 << level);  // the reference's float scale is a power of two: exact
    for (int f = tid; f < nfeat; f += T) {
      int cnt = 0;
      if (f < job.n_pts) {
        // precomputeGaussNewtonParamsPoints :216-219: floor of the float position, 3 px border
        const float u = (float)(b.pt_px[2 * (job.pt_off + f)] * scale), v = (float)(b.pt_px[2 * (job.pt_off + f) + 1] * scale);
        cnt = (u >= 3.0f && v >= 3.0f && u < (float)(W - 3) && v < (float)(Hh - 3)) ? 1 : 0;
      } else {
        const int s = job.seg_off + (f - job.n_pts);
        if (b.seg_alive[s]) {
          // precomputeGaussNewtonParamsSegments :299-301: (px*scale).cast<int>() against cam->isInFrame(.,3,level)
          const double sx = b.seg_spx[2 * s], sy = b.seg_spx[2 * s + 1], ex = b.seg_epx[2 * s], ey = b.seg_epx[2 * s + 1];
          const int cw = job.width / (1 << level), ch = job.height / (1 << level);
          const int isx = (int)(sx * scale), isy = (int)(sy * scale), iex = (int)(ex * scale), iey = (int)(ey * scale);
          const bool vis = isx >= 3 && isx < cw - 3 && isy >= 3 && isy < ch - 3 && iex >= 3 && iex < cw - 3 && iey >= 3 && iey < ch - 3;
          if (vis) cnt = seg_num_samples(sx, sy, ex, ey, b.seg_len[s], level);
        }
      }
      s_cnt[f] = cnt;
    }
    __syncthreads();
    const int n_patch = block_exclusive_scan<T>(s_cnt, nfeat, s_ctl + 4);
    if (n_patch > cap || n_patch > job.patch_cap) {  // host capacity bound violated: flag and bail out (uniform)
      if (tid == 0) st->error = 1;
      return;
    }
    for (int f = tid; f < nfeat; f += T) {
      const int p0 = s_cnt[f];
      const int my_cnt_f = (f + 1 < nfeat ? s_cnt[f + 1] : n_patch) - p0;
      if (my_cnt_f == 0) continue;
      if (f < job.n_pts) {
        const int i = job.pt_off + f;
        s_meta[p0] = make_int2(f, p0 | (1 << 20));                 // x >= 0: point index; y: first | N<<20
        b.patch_uvref[2 * (pbase + p0)] = (float)(b.pt_px[2 * i] * scale);
        b.patch_uvref[2 * (pbase + p0) + 1] = (float)(b.pt_px[2 * i + 1] * scale);
        pxyz[3 * p0] = b.pt_xyz[3 * i];
        pxyz[3 * p0 + 1] = b.pt_xyz[3 * i + 1];
        pxyz[3 * p0 + 2] = b.pt_xyz[3 * i + 2];
      } else {
        const int sl = f - job.n_pts, s = job.seg_off + sl;
        const int N = my_cnt_f;
        // :316-332: 2-D step on the level image, 3-D step between the end points, both accumulated
        const double sx = b.seg_spx[2 * s], sy = b.seg_spx[2 * s + 1];
        double inc2x = (b.seg_epx[2 * s] - sx) * scale / (double)(N - 1);
        double inc2y = (b.seg_epx[2 * s + 1] - sy) * scale / (double)(N - 1);
        double px = sx * scale, py = sy * scale;
        double xr[3], inc3[3];
        for (int c = 0; c < 3; ++c) {
          const double pr = b.seg_p[3 * s + c];
          inc3[c] = (b.seg_q[3 * s + c] - pr) / (double)(N - 1);
          xr[c] = pr;
        }
        s_dead[sl] = 0;
        for (int n = 0; n < N; ++n) {
          const int p = p0 + n;
          s_meta[p] = make_int2(-1 - sl, p0 | (N << 20));           // x < 0: segment index = -1 - x
          b.patch_uvref[2 * (pbase + p)] = (float)px;
          b.patch_uvref[2 * (pbase + p) + 1] = (float)py;
          pxyz[3 * p] = xr[0];
          pxyz[3 * p + 1] = xr[1];
          pxyz[3 * p + 2] = xr[2];
          px += inc2x; py += inc2y;
          xr[0] += inc3[0]; xr[1] += inc3[1]; xr[2] += inc3[2];
        }
      }
    }
    __syncthreads();  // patch_uvref / patch_xyz (global) and s_meta (LDS) visible to the workgroup

    // ---- reference patches: interpolated intensity + central-difference gradient (:236-264, :348-375) ----
    for (int pb = 0; pb < n_patch; pb += G) {
      const int p = pb + grp;
      if (p < n_patch) {
        const float u = b.patch_uvref[2 * (pbase + p)], v = b.patch_uvref[2 * (pbase + p) + 1];
        const PatchW pw = patch_weights(u, v);
        // patch row `row` sits on image row vi-2+row; the stencil needs image rows -1..+2 around it
        const int r0 = pw.vi - 2 + row - 1;
        const int c0 = pw.ui - 2 - 1;
        float I[4][7];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) load_row7(ref_img, (r0 + rr) * W + c0, I[rr]);
        float4 vr, vx, vy;
        float* pr = reinterpret_cast<float*>(&vr); float* pxp = reinterpret_cast<float*>(&vx); float* pyp = reinterpret_cast<float*>(&vy);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int c = x + 1;  // column of pixel x inside I[.][0..6]
          // B(r,c) = wTL*I[r][c] + wTR*I[r][c+1] + wBL*I[r+1][c] + wBR*I[r+1][c+1]
          const float ref = bilinear(pw.wTL, pw.wTR, pw.wBL, pw.wBR, I[1][c], I[1][c + 1], I[2][c], I[2][c + 1]);
          const float xp = bilinear(pw.wTL, pw.wTR, pw.wBL, pw.wBR, I[1][c + 1], I[1][c + 2], I[2][c + 1], I[2][c + 2]);
          const float xm = bilinear(pw.wTL, pw.wTR, pw.wBL, pw.wBR, I[1][c - 1], I[1][c], I[2][c - 1], I[2][c]);
          const float yp = bilinear(pw.wTL, pw.wTR, pw.wBL, pw.wBR, I[2][c], I[2][c + 1], I[3][c], I[3][c + 1]);
          const float ym = bilinear(pw.wTL, pw.wTR, pw.wBL, pw.wBR, I[0][c], I[0][c + 1], I[1][c], I[1][c + 1]);
          pr[x] = ref;
          pxp[x] = __fmul_rn(0.5f, __fsub_rn(xp, xm));
          pyp[x] = __fmul_rn(0.5f, __fsub_rn(yp, ym));
        }
        const size_t q = (pbase + p) * 4 + row;  // float4 index: patch-major, row-minor -> coalesced
        reinterpret_cast<float4*>(b.cache_ref)[q] = vr;
        reinterpret_cast<float4*>(b.cache_dx)[q] = vx;
        reinterpret_cast<float4*>(b.cache_dy)[q] = vy;
      }
    }
    if (tid == 0) { SE3d m = se3_load(s_pose + 12); quat_to_matrix(m.q, s_pose); s_pose[9] = m.t[0]; s_pose[10] = m.t[1]; s_pose[11] = m.t[2]; }
    __syncthreads();  // patch tables, pose state and cache complete
    TICK(0);

    // ---- Gauss-Newton iterations ([ext] NLLSSolver::optimizeGaussNewton) ----
    const double fs = fabs(job.fx) / (double)(1 << level);  // focal_length / (1<<level)  :262
    const float colmax = (float)(W - 2), rowmax = (float)(Hh - 2);

    for (int iter = 0; iter < job.n_iter; ++iter) {
      // -- phase 0: one lane per patch: warp the 3-D point, project, in-frame test (:422-431, :583-594)
      {
        const double R0 = s_pose[0], R1 = s_pose[1], R2 = s_pose[2], R3 = s_pose[3], R4 = s_pose[4], R5 = s_pose[5],
                     R6 = s_pose[6], R7 = s_pose[7], R8 = s_pose[8], t0 = s_pose[9], t1 = s_pose[10], t2 = s_pose[11];
        for (int p = tid; p < n_patch; p += T) {
          const int2 meta = s_meta[p];
          const int first = meta.y & 0xfffff;
          float2 uv;
          if (meta.x < 0 && s_dead[-1 - meta.x]) {
            uv = make_float2(-2.0f, -2.0f);  // line already culled
          } else {
            const double x = pxyz[3 * p], y = pxyz[3 * p + 1], z = pxyz[3 * p + 2];
            const double xc = R0 * x + R1 * y + R2 * z + t0;
            const double yc = R3 * x + R4 * y + R5 * z + t1;
            const double zc = R6 * x + R7 * y + R8 * z + t2;
            const float u = (float)((job.fx * (xc / zc) + job.cx) * scale);
            const float v = (float)((job.fy * (yc / zc) + job.cy) * scale);
            // Patch::isInFrame(halfsize=2) on floorf(u), floorf(v); NaN -> out of frame
            const bool in = (u >= 2.0f) && (v >= 2.0f) && (u < colmax) && (v < rowmax);
            uv = in ? make_float2(u, v) : make_float2(-1.0f, -1.0f);
          }
          s_uv[p] = uv;
        }
      }
      __syncthreads();
      TICK(1);

      // -- phase 1: 2 lanes per patch, two patch rows (8 pixels) each: residuals and the five patch sums.
      int evals = 0;
      {
        constexpr int G2 = T / 2;
        const int grp2 = tid >> 1, half = tid & 1;
        auto fetch2 = [&](int pb) -> P1Fetch2 {
          P1Fetch2 f;
          f.flags = 0; f.uv = make_float2(-1.0f, -1.0f); f.r0a = f.r0b = f.r1a = f.r1b = f.r2a = f.r2b = 0u; f.off = 0;
          f.vr0 = f.vx0 = f.vy0 = f.vr1 = f.vx1 = f.vy1 = make_float4(0.f, 0.f, 0.f, 0.f);
          const int p = pb + grp2;
          if (p < n_patch) {
            f.uv = s_uv[p];
            if (f.uv.x >= 0.0f) {
              f.flags = 1 | ((s_meta[p].x >= 0) ? 2 : 0);
              const int ui = (int)floorf(f.uv.x), vi = (int)floorf(f.uv.y);
              const int off = (vi - 2 + 2 * half) * W + (ui - 2);
              f.off = off;
              const int a0 = off & ~3, a1 = (off + W) & ~3, a2 = (off + 2 * W) & ~3;
              f.r0a = *reinterpret_cast<const uint32_t*>(cur_img + a0); f.r0b = *reinterpret_cast<const uint32_t*>(cur_img + a0 + 4);
              f.r1a = *reinterpret_cast<const uint32_t*>(cur_img + a1); f.r1b = *reinterpret_cast<const uint32_t*>(cur_img + a1 + 4);
              f.r2a = *reinterpret_cast<const uint32_t*>(cur_img + a2); f.r2b = *reinterpret_cast<const uint32_t*>(cur_img + a2 + 4);
              const size_t q = (pbase + p) * 4 + 2 * half;
              f.vr0 = reinterpret_cast<const float4*>(b.cache_ref)[q]; f.vr1 = reinterpret_cast<const float4*>(b.cache_ref)[q + 1];
              f.vx0 = reinterpret_cast<const float4*>(b.cache_dx)[q];  f.vx1 = reinterpret_cast<const float4*>(b.cache_dx)[q + 1];
              f.vy0 = reinterpret_cast<const float4*>(b.cache_dy)[q];  f.vy1 = reinterpret_cast<const float4*>(b.cache_dy)[q + 1];
            }
          }
          return f;
        };
        auto unpack5 = [](uint32_t lo, uint32_t hi, int sh, float* o) {
          const uint32_t w0 = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)sh);
          o[0] = (float)(w0 & 0xffu); o[1] = (float)((w0 >> 8) & 0xffu); o[2] = (float)((w0 >> 16) & 0xffu); o[3] = (float)(w0 >> 24);
          o[4] = (float)((hi >> (8 * sh)) & 0xffu);
        };
#ifndef PLSVO_PREFETCH
        P1Fetch2 cur;
        for (int pb = 0; pb < n_patch; pb += G2) {
          cur = fetch2(pb);
#else
        P1Fetch2 cur = fetch2(0);
        for (int pb = 0; pb < n_patch; pb += G2) {
          const P1Fetch2 nxt = fetch2(pb + G2);
#endif
          const int p = pb + grp2;
          double sA = 0, sB = 0, sC = 0, sD = 0, sE = 0, sChi = 0;
          float sAbs = 0.0f;
          const bool live = (cur.flags & 1) != 0;
          const bool any_point = __any((cur.flags & 3) == 3) != 0;   // wave-uniform
          if (live) {
            const bool is_point = (cur.flags & 2) != 0;
            const PatchW pw = patch_weights(cur.uv.x, cur.uv.y);
            float r0[5], r1[5], r2[5];
            unpack5(cur.r0a, cur.r0b, cur.off & 3, r0);
            unpack5(cur.r1a, cur.r1b, (cur.off + W) & 3, r1);
            unpack5(cur.r2a, cur.r2b, (cur.off + 2 * W) & 3, r2);
            // (a packed-FP32 form of this loop, two pixels per v_pk_* instruction, was measured 12 % slower: the operand
            //  pairs need extra moves and the scalar form already issues at full rate)
            // WEIGHTED is decided per wave: the patch table lists points first, then line samples, so most rounds are
            // homogeneous and the line-only ones skip the robust weight (11 instructions per pixel) and the chi2 term
            auto row4 = [&](auto WEIGHTED, const float* top, const float* bot, const float4& vr, const float4& vx, const float4& vy) {
              constexpr bool weighted = decltype(WEIGHTED)::value;
              const float* pr = reinterpret_cast<const float*>(&vr);
              const float* pxp = reinterpret_cast<const float*>(&vx);
              const float* pyp = reinterpret_cast<const float*>(&vy);
#pragma unroll
              for (int x = 0; x < 4; ++x) {
                const float c = bilinear(pw.wTL, pw.wTR, pw.wBL, pw.wBR, top[x], top[x + 1], bot[x], bot[x + 1]);
                const float res = __fsub_rn(c, pr[x]);
                const float ares = fabsf(res);
                const double rd = (double)res, dx = (double)pxp[x], dy = (double)pyp[x];
                if (weighted) {
                  // points: w = 1/(1+|r|) (:479); line pixels are accumulated unweighted (:627-629)
                  const float w = is_point ? robust_weight(ares) : 1.0f;
                  const double wd = (double)w;
                  const double wdx = wd * dx, wdy = wd * dy;
                  sA += wdx * dx; sB += wdx * dy; sC += wdy * dy;
                  sD += wdx * rd; sE += wdy * rd;
                  sChi += (double)__fmul_rn(__fmul_rn(res, res), w);
                } else {
                  sA += dx * dx; sB += dx * dy; sC += dy * dy;
                  sD += dx * rd; sE += dy * rd;
                }
                sAbs += ares;
              }
            };
            if (any_point) {
              row4(std::true_type{}, r0, r1, cur.vr0, cur.vx0, cur.vy0);
              row4(std::true_type{}, r1, r2, cur.vr1, cur.vx1, cur.vy1);
            } else {
              row4(std::false_type{}, r0, r1, cur.vr0, cur.vx0, cur.vy0);
              row4(std::false_type{}, r1, r2, cur.vr1, cur.vx1, cur.vy1);
            }
          }
          sA += dpp_mov_f64<DPP_QUAD_XOR1>(sA); sB += dpp_mov_f64<DPP_QUAD_XOR1>(sB); sC += dpp_mov_f64<DPP_QUAD_XOR1>(sC);
          sD += dpp_mov_f64<DPP_QUAD_XOR1>(sD); sE += dpp_mov_f64<DPP_QUAD_XOR1>(sE); sChi += dpp_mov_f64<DPP_QUAD_XOR1>(sChi);
          sAbs += dpp_mov_f32<DPP_QUAD_XOR1>(sAbs);
          if (p < n_patch && half == 0) {
            double* dst = part + 6 * p;
            dst[0] = sA; dst[1] = sB; dst[2] = sC; dst[3] = sD; dst[4] = sE; dst[5] = sChi;
            s_abs[p] = live ? sAbs : -1.0f;
            evals += live ? 1 : 0;
          }
#ifdef PLSVO_PREFETCH
          cur = nxt;   // software pipelining one round ahead: measured no faster than letting the two waves per SIMD overlap
#endif
        }
      }
      __syncthreads();
      TICK(2);

      // -- phase 2: one lane per patch: per-line weight, 6x6 expansion, lane-private accumulation
      double acc[30];   // 0..20 H (upper, row-major), 21..26 Jres, 27 chi2, 28 n_meas, 29 evals
#pragma unroll
      for (int k = 0; k < 30; ++k) acc[k] = 0.0;
      int n_meas = 0;
      for (int p = tid; p < n_patch; p += T) {
        const int2 meta = s_meta[p];
        const float2 uv = s_uv[p];
        double wh = 0.0, wj = 0.0;
        const double* src = part + 6 * p;
        if (meta.x >= 0) {
          if (uv.x >= 0.0f) { wh = 1.0; wj = 1.0; n_meas += PLSVO_PATCH_AREA; acc[27] += src[5]; }
        } else if (uv.x > -1.5f) {  // live line (not culled earlier); uv.x == -1 marks an out-of-frame sample
          const int first = meta.y & 0xfffff, N = meta.y >> 20;
          bool good = true; float sum = 0.0f;
          for (int n = 0; n < N; ++n) { const float a = s_abs[first + n]; good = good && (a >= 0.0f); sum += a; }
          const float res_ = (float)((double)sum / (double)N);                 // :647 (divides by #samples)
          if (good && (double)res_ < 200.0) {                                  // :648
            const float w = (float)(1.0 / (1.0 + (double)res_));               // :675
            wh = (double)w / (double)res_;                                     // :681  H += H_ * w / res_
            wj = (double)w;                                                    // :682  Jres += Jres_ * w
            if (p == first) { acc[27] += (double)__fmul_rn(__fmul_rn(res_, res_), w); n_meas += 1; }  // :683-684
          } else if (p == first) {
            s_dead[-1 - meta.x] = 1;                                           // :687-688 it->feat3D = NULL
            b.seg_alive[job.seg_off + (-1 - meta.x)] = 0;
          }
        }
        if (wh != 0.0 || wj != 0.0) {
          double xyz[3] = { pxyz[3 * p], pxyz[3 * p + 1], pxyz[3 * p + 2] };
          double J[12];
          jacobian_xyz2uv(xyz, J);
          const double hs = wh * fs * fs, js = wj * fs;
          const double A = src[0] * hs, B = src[1] * hs, C = src[2] * hs, D = src[3] * js, E = src[4] * js;
          // H += r0 (A r0 + B r1)^T + r1 (B r0 + C r1)^T, in two passes to keep the live register set small
          {
            double Pv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) Pv[k] = A * J[k] + B * J[6 + k];
            int k = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
              for (int jj = i; jj < 6; ++jj) { acc[k] += J[i] * Pv[jj]; ++k; }
          }
          {
            double Qv[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) Qv[k] = B * J[k] + C * J[6 + k];
            int k = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
              for (int jj = i; jj < 6; ++jj) { acc[k] += J[6 + i] * Qv[jj]; ++k; }
          }
#pragma unroll
          for (int i = 0; i < 6; ++i) acc[21 + i] -= D * J[i] + E * J[6 + i];
        }
      }
      acc[28] = (double)n_meas; acc[29] = (double)evals;
      TICK(3);
      // -- block reduction (fixed shape).  Step-major over all 30 values so the DPP chains interleave.
#pragma unroll
      for (int c0 = 0; c0 < 30; c0 += 6) {   // six independent chains at a time: enough ILP, bounded live registers
#pragma unroll
        for (int k = c0; k < c0 + 6; ++k) acc[k] += dpp_mov_f64<DPP_QUAD_XOR1>(acc[k]);
#pragma unroll
        for (int k = c0; k < c0 + 6; ++k) acc[k] += dpp_mov_f64<DPP_QUAD_XOR2>(acc[k]);
#pragma unroll
        for (int k = c0; k < c0 + 6; ++k) acc[k] += dpp_mov_f64<DPP_ROW_HALF_MIRROR>(acc[k]);
#pragma unroll
        for (int k = c0; k < c0 + 6; ++k) acc[k] += dpp_mov_f64<DPP_ROW_MIRROR>(acc[k]);
#pragma unroll
        for (int k = c0; k < c0 + 6; ++k) acc[k] += dpp_bcast_f64<DPP_ROW_BCAST15, 0xA>(acc[k]);
#pragma unroll
        for (int k = c0; k < c0 + 6; ++k) acc[k] += dpp_bcast_f64<DPP_ROW_BCAST31, 0xC>(acc[k]);
      }
      if (lane == 63) {
        double* dst = s_red + RED_N * wave;
#pragma unroll
        for (int k = 0; k < 30; ++k) dst[k] = acc[k];
      }
      __syncthreads();
      TICK(4);

      // -- wave 0: cross-wave totals (fixed order) in lanes 0..29, cooperative 6x6 solve, then lane 0 takes
      //    the accept / roll back / update decision
      if (wave == 0) {
        double tot = 0.0;
        if (lane < 30) { for (int w = 0; w < T / 64; ++w) tot += s_red[RED_N * w + lane]; s_tot[lane] = tot; }
        double x[6];
        wave_solve6_reg(tot, x);                                               // solve() :699
        const double chi_sum = readlane_f64(tot, 27), nm_d = readlane_f64(tot, 28), ev_d = readlane_f64(tot, 29);
        if (lane == 0) {
          const unsigned long long nm = (unsigned long long)(nm_d + 0.5);
          s_pose[27] += ev_d;
          s_ctl[2] += 1;
          // computeResiduals returns float chi2 / n_meas_ (:171,192)
          const double new_chi2 = (double)((float)chi_sum / (float)nm);
          int stop = s_ctl[1];
          if (isnan(x[0])) stop = 1;                                           // :700
          SE3d model = se3_load(s_pose + 12);
          int accepted, brk = 0;
          if ((iter > 0 && new_chi2 > s_pose[26]) || stop) {
            model = se3_load(s_pose + 19);                                     // rollback to old_model
            accepted = 0; brk = 1;
          } else {
            double mx[6];
            for (int k = 0; k < 6; ++k) mx[k] = -x[k];
            const SE3d nm_ = se3_mul(model, se3_exp_dev(mx));                  // update() :709
            se3_store(model, s_pose + 19);                                     // old_model = model
            model = nm_;
            s_pose[26] = new_chi2;
            accepted = 1;
            if (norm_max6(x) <= job.eps) brk = 1;
          }
          se3_store(model, s_pose + 12);
          quat_to_matrix(model.q, s_pose); s_pose[9] = model.t[0]; s_pose[10] = model.t[1]; s_pose[11] = model.t[2];
          s_ctl[1] = stop; s_ctl[0] = brk;
          if (b.log) {
            const int lc = st->log_count;
            if (lc < b.log_cap) {
              plsvo_align_iterlog* r = b.log + (size_t)job_id * b.log_cap + lc;
              r->level = level; r->iter = iter; r->accepted = accepted; r->stop = stop; r->n_meas = nm; r->new_chi2 = new_chi2;
              for (int k = 0; k < 6; ++k) r->x[k] = x[k];
              se3_store(model, r->T_after);
            }
            st->log_count = lc + 1;
          }
        }
      }
      __syncthreads();
      if (b.log && tid == 0) {  // H and Jres of the trace come from s_tot (written by lanes 0..26 above)
        const int lc = st->log_count - 1;
        if (lc >= 0 && lc < b.log_cap) {
          plsvo_align_iterlog* r = b.log + (size_t)job_id * b.log_cap + lc;
          for (int i = 0; i < 6; ++i) for (int jj = 0; jj < 6; ++jj) r->H[i * 6 + jj] = s_tot[sym6_index(i, jj)];
          for (int k = 0; k < 6; ++k) r->Jres[k] = s_tot[21 + k];
        }
      }
      TICK(5);
      if (s_ctl[0]) break;
    }

    if (tid == 0) {
      st->iters[level] = s_ctl[2];
      st->patch_levels += (unsigned long long)n_patch;
      st->patch_iters += (unsigned long long)(s_pose[27] + 0.5);
    }
  }  // levels

  __syncthreads();
  if (tid == 0) {
    for (int k = 0; k < 7; ++k) st->T[k] = s_pose[12 + k];
    st->chi2 = s_pose[26];
    st->stop = s_ctl[1];
    st->n_meas = (unsigned long long)(s_tot[28] + 0.5);
    for (int i = 0; i < 6; ++i) for (int jj = 0; jj < 6; ++jj) st->H[i * 6 + jj] = s_tot[sym6_index(i, jj)];
#ifdef PLSVO_TIMING
    for (int k = 0; k < 8; ++k) st->phase_ticks[k] += s_time[k];
#endif
  }
}

// LDS bytes the kernel needs for patch capacity `cap` and feature capacity `fcap` (host side helper)
size_t align_level_lds_bytes(int threads, int cap, int fcap) {
  size_t o = sizeof(double) * RED_N * (threads / 64) + sizeof(double) * 64 + sizeof(int) * 32;
  o += (size_t)cap * (sizeof(float2) + sizeof(int2) + sizeof(float)) + (size_t)(2 * fcap + 4) * sizeof(int) + 16;
  return o;
}

template <int T>
static hipError_t launch_fused_T(const AlignBatchDev& b, int cap, int fcap, int level_hi, int level_lo, size_t lds, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(align_fused_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((align_fused_kernel<T>), dim3(b.n_jobs), dim3(T), lds, stream, b, cap, fcap, level_hi, level_lo);
  return hipGetLastError();
}

hipError_t launch_align_levels(const AlignBatchDev& b, int cap, int fcap, int level_hi, int level_lo, int threads, size_t lds, hipStream_t stream) {
  switch (threads) {
    case 64: return launch_fused_T<64>(b, cap, fcap, level_hi, level_lo, lds, stream);
    case 128: return launch_fused_T<128>(b, cap, fcap, level_hi, level_lo, lds, stream);
    case 256: return launch_fused_T<256>(b, cap, fcap, level_hi, level_lo, lds, stream);
    case 512: return launch_fused_T<512>(b, cap, fcap, level_hi, level_lo, lds, stream);
    case 1024: return launch_fused_T<1024>(b, cap, fcap, level_hi, level_lo, lds, stream);
    default: return hipErrorInvalidValue;
  }
}

// gather the result poses into one contiguous n x 7 array (what a device-side consumer / RCCL gather reads)
__global__ void align_finish_kernel(AlignBatchDev b, double* poses) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= b.n_jobs) return;
  for (int k = 0; k < 7; ++k) poses[7 * j + k] = b.state[j].T[k];
}

hipError_t launch_align_finish(const AlignBatchDev& b, double* d_poses, hipStream_t stream) {
  hipLaunchKernelGGL(align_finish_kernel, dim3((b.n_jobs + 63) / 64), dim3(64), 0, stream, b, d_poses);
  return hipGetLastError();
}

hipError_t launch_align_init(const AlignBatchDev& b, hipStream_t stream) {
  hipLaunchKernelGGL(align_init_kernel, dim3(b.n_jobs), dim3(64), 0, stream, b);
  return hipGetLastError();
}

}  // namespace plsvo_hip
