// seeds_kernels.hip -- depth-filter seed update on gfx950: one LANE per seed.
//
// Replaces (reference file:line) the per-seed bodies of
//   DepthFilter::updatePointSeeds / updateLineSeeds          src/depth_filter.cpp:270-365, :367-471
// with everything they call:
//   Matcher::findEpipolarMatchDirect                         src/matcher.cpp:277-420
//   Matcher::findEpipolarMatchDirectSegmentEndpoint          src/matcher.cpp:422-586
//   depthFromTriangulation                                   src/matcher.cpp:133-146
//   DepthFilter::computeTau / updatePointSeed / updateLineSeed   src/depth_filter.cpp:568-584, :489-512, :514-566
//   [ext] vk::patch_score::ZMSSD<4>, boost::math::pdf(normal_distribution<float>), warp::*, align2D (match_device.hpp)
// The seed lists, their age test, the converged-seed callbacks and the detector's grid occupancy stay on the host.
//
// A seed is an independent problem: project its inverse-depth interval into the new frame, walk the epipolar segment
// in 0.7-pixel steps scoring an 8x8 warped reference patch by zero-mean SSD (integer), refine the best position with
// the 8x8 Lucas-Kanade of the direct matcher, triangulate, and fold the measurement into the seed's
// Gaussian x Beta posterior (Vogiatzis & Hernandez).  The walk is sequential by construction (strict `<` keeps the FIRST
// best score; the sample position is accumulated, uv += step), so a lane owns a seed; the ZMSSD of one position is
// 6 v_dot4_u32_u8 per patch row on bytes fetched with aligned dwords + v_alignbyte.  Integer scores, the warp and the
// alignment are bit-identical to the CPU oracle (-ffp-contract=off); tau and the posterior use device acos/sin/exp, which
// differ from glibc in the last bit at most.
#include <hip/hip_runtime.h>

#include "match_device.hpp"

namespace plsvo_hip {

#pragma clang fp contract(off)

// depthFromTriangulation (src/matcher.cpp:133-146)
__device__ __forceinline__ bool depth_from_triangulation(const SE3d& T_search_ref, const double* f_ref, const double* f_cur, double* depth) {
  double R[9], c0[3];
  quat_to_matrix(T_search_ref.q, R);
  for (int i = 0; i < 3; ++i) c0[i] = R[3 * i] * f_ref[0] + R[3 * i + 1] * f_ref[1] + R[3 * i + 2] * f_ref[2];
  const double a00 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2];
  const double a01 = c0[0] * f_cur[0] + c0[1] * f_cur[1] + c0[2] * f_cur[2];
  const double a11 = f_cur[0] * f_cur[0] + f_cur[1] * f_cur[1] + f_cur[2] * f_cur[2];
  const double a10 = a01;
  const double det = a00 * a11 - a10 * a01;
  if (det < 0.000001) return false;
  const double invdet = 1.0 / det;
  const double i00 = -(a11 * invdet), i01 = -(-a01 * invdet);
  const double m0 = i00 * c0[0] + i01 * f_cur[0], m1 = i00 * c0[1] + i01 * f_cur[1], m2 = i00 * c0[2] + i01 * f_cur[2];
  const double d0 = m0 * T_search_ref.t[0] + m1 * T_search_ref.t[1] + m2 * T_search_ref.t[2];
  *depth = fabs(d0);
  return true;
}

// 8 bytes at byte offset `off` of a u8 image as two dwords
__device__ __forceinline__ void load_row8(const uint8_t* img, long off, uint32_t& lo, uint32_t& hi) {
  const uint32_t* p = reinterpret_cast<const uint32_t*>(img + (off & ~3l));
  const uint32_t sh = (uint32_t)(off & 3);
  const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
  lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
  hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
}
__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }

// Matcher::findEpipolarMatchDirect (seg == 0) / findEpipolarMatchDirectSegmentEndpoint (seg == 1).
// Kept out of line: a line seed calls it twice and a point seed once from the same kernel.
// Everything travels by value -- the handful of batch fields it reads (EpiCtx), the bearing, the result (EpiOut): handing the
// kernel's 500-byte argument struct and three stack arrays over by address cost the kernel a 576-byte scratch frame per lane.
struct EpiCtx {
  const uint8_t* pyr_base; unsigned long long slot_bytes; const double* frame_T; const int* frame_slot;
  double fx, fy, cx, cy, edgelet_max_angle;
  int width, height, cam_width, cam_height, n_pyr_levels, align_max_iter, max_epi_search_steps, edgelet_filtering;
};
struct EpiOut { int ok; double depth, px0, px1; };
__device__ __forceinline__ EpiCtx epi_ctx(const SeedsBatchDev& s) {
  EpiCtx e;
  e.pyr_base = s.pyr_base; e.slot_bytes = s.slot_bytes; e.frame_T = s.frame_T; e.frame_slot = s.frame_slot;
  e.fx = s.fx; e.fy = s.fy; e.cx = s.cx; e.cy = s.cy; e.edgelet_max_angle = s.edgelet_max_angle;
  e.width = s.width; e.height = s.height; e.cam_width = s.cam_width; e.cam_height = s.cam_height; e.n_pyr_levels = s.n_pyr_levels;
  e.align_max_iter = s.align_max_iter; e.max_epi_search_steps = s.max_epi_search_steps; e.edgelet_filtering = s.edgelet_filtering;
  return e;
}
__device__ __forceinline__ bool epipolar_search_body(const EpiCtx& b, uint32_t* my, int rf, int cf, double rpx0, double rpx1, const double* f_ref,
                                                     int level, int type, double g0, double g1, double d_estimate, double d_min, double d_max, int seg,
                                                     double* depth, double* px_cur);
// (an aggregate passed by value travels through the stack as well: the context sits in LDS, written once per workgroup, and the
//  out-of-line function receives its address)
typedef const __attribute__((address_space(3))) EpiCtx* EpiCtxLds;
__device__ __noinline__ EpiOut epipolar_search(EpiCtxLds c, uint32_t* my, int rf, int cf, double rpx0, double rpx1, double f0, double f1, double f2,
                                               int level, int type, double g0, double g1, double d_estimate, double d_min, double d_max, int seg) {
  EpiCtx b;
  b.pyr_base = c->pyr_base; b.slot_bytes = c->slot_bytes; b.frame_T = c->frame_T; b.frame_slot = c->frame_slot;
  b.fx = c->fx; b.fy = c->fy; b.cx = c->cx; b.cy = c->cy; b.edgelet_max_angle = c->edgelet_max_angle;
  b.width = c->width; b.height = c->height; b.cam_width = c->cam_width; b.cam_height = c->cam_height; b.n_pyr_levels = c->n_pyr_levels;
  b.align_max_iter = c->align_max_iter; b.max_epi_search_steps = c->max_epi_search_steps; b.edgelet_filtering = c->edgelet_filtering;
  const double f_ref[3] = { f0, f1, f2 };
  double depth = 0.0, px_cur[2] = { 0.0, 0.0 };
  EpiOut o;
  o.ok = epipolar_search_body(b, my, rf, cf, rpx0, rpx1, f_ref, level, type, g0, g1, d_estimate, d_min, d_max, seg, &depth, px_cur) ? 1 : 0;
  o.depth = depth; o.px0 = px_cur[0]; o.px1 = px_cur[1];
  return o;
}
__device__ __forceinline__ bool epipolar_search_body(const EpiCtx& b, uint32_t* my, int rf, int cf, double rpx0, double rpx1, const double* f_ref,
                                                     int level, int type, double g0, double g1, double d_estimate, double d_min, double d_max, int seg,
                                                     double* depth, double* px_cur) {
  const CamDev cam{b.fx, b.fy, b.cx, b.cy, b.cam_width, b.cam_height};
  px_cur[0] = 0.0; px_cur[1] = 0.0;
  const SE3d T_ref = se3_load(b.frame_T + 7 * rf), T_cur = se3_load(b.frame_T + 7 * cf);
  const SE3d T_ref_inv = se3_inv(T_ref);
  const SE3d T_cur_ref = se3_mul(T_cur, T_ref_inv);
  if (seg && (d_min != d_min || d_max != d_max)) return false;                                   // :436-440
  const double pa[3] = { f_ref[0] * d_min, f_ref[1] * d_min, f_ref[2] * d_min }, pb[3] = { f_ref[0] * d_max, f_ref[1] * d_max, f_ref[2] * d_max };
  double ca[3], cb[3];
  se3_act(T_cur_ref, pa, ca); se3_act(T_cur_ref, pb, cb);
  const double A0 = ca[0] / ca[2], A1 = ca[1] / ca[2], B0 = cb[0] / cb[2], B1 = cb[1] / cb[2];   // vk::project2d
  const double e0 = A0 - B0, e1 = A1 - B1;                                                       // epi_dir_
  double Aw[4];
  warp_matrix_affine(cam, rpx0, rpx1, f_ref, d_estimate, T_cur_ref, level, Aw);
  if (!seg && type == PLSVO_FTR_EDGELET && b.edgelet_filtering) {                                // :303-311
    double h0 = Aw[0] * g0 + Aw[1] * g1, h1 = Aw[2] * g0 + Aw[3] * g1;
    const double gn = sqrt(h0 * h0 + h1 * h1);
    h0 /= gn; h1 /= gn;
    const double en = sqrt(e0 * e0 + e1 * e1);
    const double cosangle = fabs(h0 * (e0 / en) + h1 * (e1 / en));
    if (cosangle < b.edgelet_max_angle) return false;
  }
  const int search_level = best_search_level(Aw, b.n_pyr_levels - 1);
  const double pxA0 = b.fx * A0 + b.cx, pxA1 = b.fy * A1 + b.cy, pxB0 = b.fx * B0 + b.cx, pxB1 = b.fy * B1 + b.cy;
  const double dA0 = pxA0 - pxB0, dA1 = pxA1 - pxB1;
  const double epi_length = sqrt(dA0 * dA0 + dA1 * dA1) / (1 << search_level);
  if (epi_length != epi_length || fabs(epi_length) > 1.7976931348623157e308) return false;
  const uint8_t* img_ref = b.pyr_base + (unsigned long long)b.frame_slot[rf] * b.slot_bytes + pyr_level_offset(b.width, b.height, level);
  if (!warp_affine_lds(Aw, img_ref, b.width >> level, b.height >> level, rpx0, rpx1, level, search_level, my)) return false;
  const int cols = b.width >> search_level, rows = b.height >> search_level;
  const uint8_t* cur_img = b.pyr_base + (unsigned long long)b.frame_slot[cf] * b.slot_bytes + pyr_level_offset(b.width, b.height, search_level);
  const double sc = (double)(1 << search_level);
  int iters = 0;
  if (epi_length < 2.0) {                                                                        // :325-344
    px_cur[0] = (pxA0 + pxB0) / 2.0; px_cur[1] = (pxA1 + pxB1) / 2.0;
    double est0 = px_cur[0] / sc, est1 = px_cur[1] / sc;
    if (align2d_lds(cur_img, cols, rows, my, b.align_max_iter, est0, est1, iters)) {
      px_cur[0] = est0 * sc; px_cur[1] = est1 * sc;
      double f_cur[3];
      cam2world(cam, px_cur[0], px_cur[1], f_cur);
      if (depth_from_triangulation(T_cur_ref, f_ref, f_cur, depth)) return true;
    }
    return false;
  }
  unsigned long long n_steps = (unsigned long long)(epi_length / 0.7);                            // :347
  const double step0 = e0 / (double)n_steps, step1 = e1 / (double)n_steps;
  if (n_steps > (unsigned long long)b.max_epi_search_steps) return false;                         // :350-355
  // ZMSSD<4>: the warped reference patch = interior of the LDS patch, as 2 dwords per row
  uint32_t ra[8], rb[8], sumA = 0, sumAA = 0;
#pragma unroll
  for (int y = 0; y < 8; ++y) {
    const uint32_t w0 = PBW(my, y + 1, 0), w1 = PBW(my, y + 1, 1), w2 = PBW(my, y + 1, 2);
    ra[y] = __builtin_amdgcn_alignbyte(w1, w0, 1u);
    rb[y] = __builtin_amdgcn_alignbyte(w2, w1, 1u);
    sumA = dot4(ra[y], 0x01010101u, sumA); sumA = dot4(rb[y], 0x01010101u, sumA);
    sumAA = dot4(ra[y], ra[y], sumAA); sumAA = dot4(rb[y], rb[y], sumAA);
  }
  const int threshold = 2000 * 64;
  int zmssd_best = threshold;
  double uvb0 = 0.0, uvb1 = 0.0;
  double uv0 = B0 - step0, uv1 = B1 - step1;
  int last_x = 0, last_y = 0;
  ++n_steps;
  for (unsigned long long i = 0; i < n_steps; ++i, uv0 += step0, uv1 += step1) {
    const double q0 = (b.fx * uv0 + b.cx) / (1 << search_level) + 0.5, q1 = (b.fy * uv1 + b.cy) / (1 << search_level) + 0.5;
    if (!(fabs(q0) < 1e9 && fabs(q1) < 1e9)) continue;
    const int pxi0 = (int)q0, pxi1 = (int)q1;
    if (pxi0 == last_x && pxi1 == last_y) continue;
    last_x = pxi0; last_y = pxi1;
    if (!cam_is_in_frame(cam, pxi0, pxi1, 8, search_level)) continue;
    long off = (long)(pxi1 - 4) * cols + (pxi0 - 4);
    uint32_t sumB = 0, sumBB = 0, sumAB = 0;
#pragma unroll
    for (int y = 0; y < 8; ++y, off += cols) {
      uint32_t lo, hi;
      load_row8(cur_img, off, lo, hi);
      sumB = dot4(lo, 0x01010101u, sumB); sumB = dot4(hi, 0x01010101u, sumB);
      sumBB = dot4(lo, lo, sumBB); sumBB = dot4(hi, hi, sumBB);
      sumAB = dot4(lo, ra[y], sumAB); sumAB = dot4(hi, rb[y], sumAB);
    }
    const int sA = (int)sumA, sB = (int)sumB;
    const int z = (int)sumAA - 2 * (int)sumAB + (int)sumBB - (sA * sA - 2 * sA * sB + sB * sB) / 64;
    if (z < zmssd_best) { zmssd_best = z; uvb0 = uv0; uvb1 = uv1; }
  }
  if (zmssd_best < threshold) {                                                                  // :392-412
    px_cur[0] = b.fx * uvb0 + b.cx; px_cur[1] = b.fy * uvb1 + b.cy;
    double est0 = px_cur[0] / sc, est1 = px_cur[1] / sc;
    if (align2d_lds(cur_img, cols, rows, my, b.align_max_iter, est0, est1, iters)) {
      px_cur[0] = est0 * sc; px_cur[1] = est1 * sc;
      double f_cur[3];
      cam2world(cam, px_cur[0], px_cur[1], f_cur);
      if (depth_from_triangulation(T_cur_ref, f_ref, f_cur, depth)) return true;
    }
  }
  return false;
}

// DepthFilter::computeTau (src/depth_filter.cpp:568-584)
__device__ __forceinline__ double compute_tau(const SE3d& T_ref_cur, const double* f, double z, double px_error_angle) {
  const double* t = T_ref_cur.t;
  const double a[3] = { f[0] * z - t[0], f[1] * z - t[1], f[2] * z - t[2] };
  const double t_norm = sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
  const double a_norm = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  const double alpha = acos((f[0] * t[0] + f[1] * t[1] + f[2] * t[2]) / t_norm);
  const double beta = acos((a[0] * -t[0] + a[1] * -t[1] + a[2] * -t[2]) / (t_norm * a_norm));
  const double beta_plus = beta + px_error_angle;
  const double gamma_plus = 3.14159265 - alpha - beta_plus;    // PI of include/plsvo/global.h:100
  const double z_plus = t_norm * sin(beta_plus) / sin(gamma_plus);
  return z_plus - z;
}

// one end of the Vogiatzis-Hernandez update (updatePointSeed :489-512 and each half of updateLineSeed :514-552), in the
// reference's mix of float and double; [ext] boost::math::pdf(normal_distribution<float>) written out
__device__ __forceinline__ void seed_update_end(float x, float tau2, float a, float b, float z_range, float& mu, float& sigma2, float norm_scale,
                                                float& f_out, float& e_out) {
  float pdf = 0.0f;
  if (!(fabsf(x) > 3.402823466e+38f)) {
    float exponent = x - mu;
    exponent *= -exponent;
    exponent /= 2 * norm_scale * norm_scale;
    pdf = (float)exp((double)exponent);                 // the correctly rounded float exp (glibc's expf is within 0.502 ulp of it)
    pdf /= norm_scale * sqrtf(2 * 3.14159265358979323846f);
  }
  float s2 = 1. / (1. / sigma2 + 1. / tau2);
  float m = s2 * (mu / sigma2 + x / tau2);
  float C1 = a / (a + b) * pdf;
  float C2 = b / (a + b) * 1. / z_range;
  float normalization_constant = C1 + C2;
  C1 /= normalization_constant;
  C2 /= normalization_constant;
  float f = C1 * (a + 1.) / (a + b + 1.) + C2 * a / (a + b + 1.);
  float e = C1 * (a + 1.) * (a + 2.) / ((a + b + 1.) * (a + b + 2.)) + C2 * a * (a + 1.0f) / ((a + b + 1.0f) * (a + b + 2.0f));
  float mu_new = C1 * m + C2 * mu;
  sigma2 = C1 * (s2 + m * m) + C2 * (sigma2 + mu * mu) - mu_new * mu_new;
  mu = mu_new;
  f_out = f; e_out = e;
}

__device__ __forceinline__ bool px_in_image(const CamDev& cam, const double* px) {
  return px[0] == px[0] && px[1] == px[1] && fabs(px[0]) < 1e9 && fabs(px[1]) < 1e9 && cam_is_in_frame(cam, (int)px[0], (int)px[1], 0, 0);
}

__global__ void __launch_bounds__(MT) update_seeds_kernel(const SeedsBatchDev b) {
  __shared__ uint32_t s_patch[PB_ROWS * PB_WORDS * MT];
  __shared__ EpiCtx s_epi;
  const int lane = threadIdx.x;
  if (lane == 0) s_epi = epi_ctx(b);
  __syncthreads();
  const EpiCtxLds epi = (EpiCtxLds)&s_epi;
  const int gi = blockIdx.x * MT + lane;
  if (gi >= b.n_pt + b.n_seg) return;
  uint32_t* my = s_patch + lane;
  const CamDev cam{b.fx, b.fy, b.cx, b.cy, b.cam_width, b.cam_height};
  if (gi < b.n_pt) {
    // ---- DepthFilter::updatePointSeeds, one seed (:295-360) ----
    const int i = gi;
    float a = b.pt_a[i], bb = b.pt_b[i], mu = b.pt_mu[i], sigma2 = b.pt_sigma2[i];
    const float z_range = b.pt_z_range[i];
    int status = PLSVO_SEED_NOT_VISIBLE;
    double xyz_world[3] = { 0, 0, 0 }, z = 0.0, px_cur[2] = { 0, 0 };
    const int rf = b.pt_ref_frame[i], cf = b.pt_cur_frame[i];
    const double f[3] = { b.pt_f[3 * i], b.pt_f[3 * i + 1], b.pt_f[3 * i + 2] };
    const SE3d T_ref = se3_load(b.frame_T + 7 * rf), T_cur = se3_load(b.frame_T + 7 * cf);
    const SE3d T_ref_cur = se3_mul(T_ref, se3_inv(T_cur));
    const SE3d T_cur_ref = se3_inv(T_ref_cur);
    const double s = 1.0 / mu;
    const double p[3] = { s * f[0], s * f[1], s * f[2] };
    double xyz_f[3], px[2];
    se3_act(T_cur_ref, p, xyz_f);
    bool visible = !(xyz_f[2] < 0.0);
    if (visible) { world2cam(cam, xyz_f, px); visible = px_in_image(cam, px); }
    if (visible) {
      const float z_inv_min = mu + sqrtf(sigma2);
      const float t_ = mu - sqrtf(sigma2);
      const float z_inv_max = t_ > 0.00000001f ? t_ : 0.00000001f;
      const double g0 = b.pt_grad ? b.pt_grad[2 * i] : 0.0, g1 = b.pt_grad ? b.pt_grad[2 * i + 1] : 0.0;
      const EpiOut eo = epipolar_search(epi, my, rf, cf, b.pt_px[2 * i], b.pt_px[2 * i + 1], f[0], f[1], f[2], b.pt_level[i], b.pt_type[i], g0, g1,
                                        1.0 / mu, 1.0 / z_inv_min, 1.0 / z_inv_max, 0);
      z = eo.depth; px_cur[0] = eo.px0; px_cur[1] = eo.px1;
      if (!eo.ok) {
        bb += 1.0f;
        status = PLSVO_SEED_NO_MATCH;
      } else {
        const double tau = compute_tau(T_ref_cur, f, z, b.px_error_angle);
        const double zm = z - tau;
        const double tau_inverse = 0.5 * (1.0 / (0.0000001 < zm ? zm : 0.0000001) - 1.0 / (z + tau));
        const float x = (float)(1. / z), tau2 = (float)(tau_inverse * tau_inverse);
        const float norm_scale = sqrtf(sigma2 + tau2);
        if (!(norm_scale != norm_scale)) {
          float fq, eq;
          seed_update_end(x, tau2, a, bb, z_range, mu, sigma2, norm_scale, fq, eq);
          a = (eq - fq) / (fq - eq / fq);
          bb = a * (1.0f - fq) / fq;
        }
        status = PLSVO_SEED_UPDATED;
        if (sqrtf(sigma2) < z_range / b.convergence_sigma2_thresh) {
          const double sw = 1.0 / mu;
          const double pw[3] = { f[0] * sw, f[1] * sw, f[2] * sw };
          se3_act(se3_inv(T_ref), pw, xyz_world);
          status = PLSVO_SEED_CONVERGED;
        } else if (z_inv_min != z_inv_min) {
          status = PLSVO_SEED_NAN;
        }
      }
    }
    b.o_pt_status[i] = status; b.o_pt_a[i] = a; b.o_pt_b[i] = bb; b.o_pt_mu[i] = mu; b.o_pt_sigma2[i] = sigma2;
    for (int k = 0; k < 3; ++k) b.o_pt_xyz[3 * i + k] = xyz_world[k];
    b.o_pt_px[2 * i] = px_cur[0]; b.o_pt_px[2 * i + 1] = px_cur[1];
    b.o_pt_depth[i] = z;
  } else {
    // ---- DepthFilter::updateLineSeeds, one seed (:391-467) ----
    const int i = gi - b.n_pt;
    float a = b.seg_a[i], bb = b.seg_b[i], mu_s = b.seg_mu_s[i], mu_e = b.seg_mu_e[i], sigma2_s = b.seg_sigma2_s[i], sigma2_e = b.seg_sigma2_e[i];
    const float z_range_s = b.seg_z_range_s[i], z_range_e = b.seg_z_range_e[i];
    int status = PLSVO_SEED_NOT_VISIBLE;
    double xw_s[3] = { 0, 0, 0 }, xw_e[3] = { 0, 0, 0 }, z_s = 0.0, z_e = 0.0, px_cur[2];
    const int rf = b.seg_ref_frame[i], cf = b.seg_cur_frame[i];
    const double sf[3] = { b.seg_sf[3 * i], b.seg_sf[3 * i + 1], b.seg_sf[3 * i + 2] }, ef[3] = { b.seg_ef[3 * i], b.seg_ef[3 * i + 1], b.seg_ef[3 * i + 2] };
    const double fc[3] = { b.seg_f[3 * i], b.seg_f[3 * i + 1], b.seg_f[3 * i + 2] };
    const SE3d T_ref = se3_load(b.frame_T + 7 * rf), T_cur = se3_load(b.frame_T + 7 * cf);
    const SE3d T_ref_cur = se3_mul(T_ref, se3_inv(T_cur));
    const SE3d T_cur_ref = se3_inv(T_ref_cur);
    const double ss = 1.0 / mu_s, se = 1.0 / mu_e;
    const double ps[3] = { ss * sf[0], ss * sf[1], ss * sf[2] }, pe[3] = { se * ef[0], se * ef[1], se * ef[2] };
    double xs[3], xe[3], pxs[2], pxe[2];
    se3_act(T_cur_ref, ps, xs); se3_act(T_cur_ref, pe, xe);
    bool visible = !(xs[2] < 0.0 || xe[2] < 0.0);
    if (visible) { world2cam(cam, xs, pxs); world2cam(cam, xe, pxe); visible = px_in_image(cam, pxs) && px_in_image(cam, pxe); }
    if (visible) {
      const float z_inv_min_s = mu_s + sqrtf(sigma2_s);
      const float ts_ = mu_s - sqrtf(sigma2_s);
      const float z_inv_max_s = ts_ > 0.00000001f ? ts_ : 0.00000001f;
      const float z_inv_min_e = mu_e + sqrtf(sigma2_e);
      const float te_ = mu_e - sqrtf(sigma2_e);
      const float z_inv_max_e = te_ > 0.00000001f ? te_ : 0.00000001f;
      const double px0 = b.seg_px[2 * i], px1 = b.seg_px[2 * i + 1];
      const int level = b.seg_level[i];
      // (the second search runs only when the first one succeeds, like the reference's `||`)
      const EpiOut es = epipolar_search(epi, my, rf, cf, px0, px1, fc[0], fc[1], fc[2], level, PLSVO_FTR_CORNER, 0.0, 0.0, 1.0 / mu_s, 1.0 / z_inv_min_s, 1.0 / z_inv_max_s, 1);
      z_s = es.depth;
      EpiOut ee; ee.ok = 0; ee.depth = 0.0; ee.px0 = 0.0; ee.px1 = 0.0;
      if (es.ok) { ee = epipolar_search(epi, my, rf, cf, px0, px1, fc[0], fc[1], fc[2], level, PLSVO_FTR_CORNER, 0.0, 0.0, 1.0 / mu_e, 1.0 / z_inv_min_e, 1.0 / z_inv_max_e, 1); z_e = ee.depth; }
      if (!es.ok || !ee.ok) {
        bb += 1.0f;
        status = PLSVO_SEED_NO_MATCH;
      } else {
        const double tau_s = compute_tau(T_ref_cur, sf, z_s, b.px_error_angle);
        const double zms = z_s - tau_s;
        const double tau_inverse_s = 0.5 * (1.0 / (0.0000001 < zms ? zms : 0.0000001) - 1.0 / (z_s + tau_s));
        const double tau_e = compute_tau(T_ref_cur, ef, z_e, b.px_error_angle);
        const double zme = z_e - tau_e;
        const double tau_inverse_e = 0.5 * (1.0 / (0.0000001 < zme ? zme : 0.0000001) - 1.0 / (z_e + tau_e));
        const float x_s = (float)(1. / z_s), tau2_s = (float)(tau_inverse_s * tau_inverse_s);
        const float x_e = (float)(1. / z_e), tau2_e = (float)(tau_inverse_e * tau_inverse_e);
        const float norm_scale_s = sqrtf(sigma2_s + tau2_s), norm_scale_e = sqrtf(sigma2_e + tau2_e);
        if (!(norm_scale_s != norm_scale_s || norm_scale_e != norm_scale_e)) {
          float f_s, e_s, f_e, e_e;
          seed_update_end(x_s, tau2_s, a, bb, z_range_s, mu_s, sigma2_s, norm_scale_s, f_s, e_s);
          seed_update_end(x_e, tau2_e, a, bb, z_range_e, mu_e, sigma2_e, norm_scale_e, f_e, e_e);
          const float a_s = (e_s - f_s) / (f_s - e_s / f_s), a_e = (e_e - f_e) / (f_e - e_e / f_e);
          const float b_s = a_s * (1.f - f_s) / f_s, b_e = a_e * (1.f - f_e) / f_e;
          a = (a_s < a_e) ? a_e : a_s;
          bb = (b_e < b_s) ? b_e : b_s;
        }
        status = PLSVO_SEED_UPDATED;
        if (sqrtf(sigma2_s) < z_range_s / b.convergence_sigma2_thresh && sqrtf(sigma2_e) < z_range_e / b.convergence_sigma2_thresh) {
          const SE3d T_ref_inv = se3_inv(T_ref);
          const double ws = 1.0 / mu_s, we = 1.0 / mu_e;
          const double pws[3] = { sf[0] * ws, sf[1] * ws, sf[2] * ws }, pwe[3] = { ef[0] * we, ef[1] * we, ef[2] * we };
          se3_act(T_ref_inv, pws, xw_s); se3_act(T_ref_inv, pwe, xw_e);
          status = PLSVO_SEED_CONVERGED;
        } else if (z_inv_min_s != z_inv_min_s || z_inv_min_e != z_inv_min_e) {
          status = PLSVO_SEED_NAN;
        }
      }
    }
    b.o_seg_status[i] = status; b.o_seg_a[i] = a; b.o_seg_b[i] = bb; b.o_seg_mu_s[i] = mu_s; b.o_seg_mu_e[i] = mu_e;
    b.o_seg_sigma2_s[i] = sigma2_s; b.o_seg_sigma2_e[i] = sigma2_e;
    for (int k = 0; k < 3; ++k) { b.o_seg_xyz_s[3 * i + k] = xw_s[k]; b.o_seg_xyz_e[3 * i + k] = xw_e[k]; }
    b.o_seg_depth_s[i] = z_s; b.o_seg_depth_e[i] = z_e;
  }
}

hipError_t launch_update_seeds(const SeedsBatchDev& b, hipStream_t stream) {
  const int n = b.n_pt + b.n_seg;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(update_seeds_kernel, dim3((n + MT - 1) / MT), dim3(MT), 0, stream, b);
  return hipGetLastError();
}

}  // namespace plsvo_hip
