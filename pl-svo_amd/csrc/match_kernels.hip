// match_kernels.hip -- direct feature matching on gfx950: one LANE per match candidate.
//
// Replaces (reference file:line):
//   plsvo::Matcher::findMatchDirect (Point)     src/matcher.cpp:157-208
//   plsvo::Matcher::findMatchDirect (LineSeg)   src/matcher.cpp:233-280   (two candidates, ANDed by the caller)
//   Matcher::precomputeRefPatch                 src/matcher.cpp:210-231
//   Matcher::createPatchFromPatchWithBorder     src/matcher.cpp:146-155
//   warp::getWarpMatrixAffine / getBestSearchLevel / warpAffine   src/matcher.cpp:40-128
//   feature_alignment::align1D / align2D        src/feature_alignment.cpp:41-157, :159-283
//   Patch::setPosition / computeInterpWeights / setRoi / isInFrame  src/feature.cpp:189-218, include/plsvo/feature.h:139-144
//   [ext] vk::interpolateMat_8u, vk::AbstractCamera::isInFrame, Eigen 2x2 / 3x3 inverse()
// called once per selected map point / segment per frame from Reprojector::refineBestCandidate
// (src/reprojector.cpp:288, :348).
//
// Every candidate is an independent 8x8 inverse-compositional Lucas-Kanade problem (<= 10 iterations, 3x3 or 2x2
// float Hessian), about 10k float operations and < 1 KB of image bytes: there is nothing to reduce across lanes, so a
// batch of candidates (all frames' worth) is one launch, one lane each.  The float accumulations of the reference
// (Jres, H) are sequential sums over the 64 patch pixels; a lane-per-candidate layout keeps exactly that order, and
// this file is compiled with -ffp-contract=off, so results are BIT-IDENTICAL to the CPU oracle.
//
// Data movement per candidate: the 10x10 warped reference patch is interpolated once from the keyframe's pyramid
// level (400 scattered byte reads, L2-resident neighbourhoods) and parked in LDS, lane-interleaved by dword so the
// 64 lanes of a wave never collide on a bank; each iteration then reads a 9x9 window of the current image as 27
// aligned dwords (v_alignbyte for the sub-dword offset) and the reference rows from LDS.  The gradient images
// ref_patch_dx/dy of the reference (2 x 64 floats per candidate) are never materialised: 0.5*(it[1]-it[-1]) is exact
// in float, so it is recomputed from the LDS bytes.
#include <hip/hip_runtime.h>

#include "plsvo_dev.hpp"
#include "plsvo_math.hpp"

namespace plsvo_hip {

#pragma clang fp contract(off)

constexpr int MT = 64;          // lanes (candidates) per workgroup
constexpr int PB_STEP = 10;     // patch_with_border row length (patch_size_+2)
constexpr int PB_WORDS = 3;     // dwords per LDS row (10 bytes + 2 pad)
constexpr int PB_ROWS = 10;

__device__ __forceinline__ int f2i_trunc(float x) { return (x != x) ? INT32_MIN : (int)x; }

// [ext] vk::PinholeCamera::world2cam / cam2world without distortion
__device__ __forceinline__ void world2cam(const MatchBatchDev& b, const double* xyz, double* px) {
  const double u = xyz[0] / xyz[2], v = xyz[1] / xyz[2];
  px[0] = b.fx * u + b.cx;
  px[1] = b.fy * v + b.cy;
}
__device__ __forceinline__ void cam2world(const MatchBatchDev& b, double pu, double pv, double* f) {
  const double x = (pu - b.cx) / b.fx, y = (pv - b.cy) / b.fy, z = 1.0;
  const double n = sqrt(x * x + y * y + z * z);
  f[0] = x / n; f[1] = y / n; f[2] = z / n;
}

// [ext] vk::interpolateMat_8u
__device__ __forceinline__ float interpolate_mat_8u(const uint8_t* img, int stride, float u, float v) {
  const int x = (int)floorf(u), y = (int)floorf(v);
  const float subpix_x = u - x, subpix_y = v - y;
  const float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
  const float w01 = (1.0f - subpix_x) * subpix_y;
  const float w10 = subpix_x * (1.0f - subpix_y);
  const float w11 = 1.0f - w00 - w01 - w10;
  const uint8_t* ptr = img + (long)y * stride + x;
  return w00 * ptr[0] + w01 * ptr[stride] + w10 * ptr[1] + w11 * ptr[stride + 1];
}

// bytes [off, off+9) of a u8 image as three dwords holding bytes 0-3, 4-7, 8-11 (aligned dword reads + v_alignbyte;
// every pyramid level is followed by >= 64 bytes of slack, plsvo_dev.hpp)
struct Row9 { uint32_t a, b, c; };
__device__ __forceinline__ Row9 load_row9(const uint8_t* img, long off) {
  const uint32_t* p = reinterpret_cast<const uint32_t*>(img + (off & ~3l));
  const uint32_t sh = (uint32_t)(off & 3);
  const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
  Row9 r;
  r.a = __builtin_amdgcn_alignbyte(d1, d0, sh);
  r.b = __builtin_amdgcn_alignbyte(d2, d1, sh);
  r.c = d2 >> (8 * sh);
  return r;
}
__device__ __forceinline__ int row_byte(uint32_t a, uint32_t b, uint32_t c, int k) {   // k is a compile-time constant after unrolling
  const uint32_t w = k < 4 ? a : (k < 8 ? b : c);
  return (int)((w >> (8 * (k & 3))) & 0xffu);
}

__global__ void __launch_bounds__(MT) match_direct_kernel(const MatchBatchDev b) {
  __shared__ uint32_t s_patch[PB_ROWS * PB_WORDS * MT];   // patch_with_border_, word-major / lane-minor
  const int lane = threadIdx.x;
  const int i = blockIdx.x * MT + lane;
  if (i >= b.n) return;
  uint32_t* my = s_patch + lane;
#define PBW(row, w) my[((row) * PB_WORDS + (w)) * MT]

  const int rf = b.ref_frame[i], cf = b.cur_frame[i], level = b.ref_level[i];
  double px_cur[2] = { b.px_cur[2 * i], b.px_cur[2 * i + 1] };
  const double rpx0 = b.ref_px[2 * i], rpx1 = b.ref_px[2 * i + 1];
  int found = 0, search_level = -1, iters = 0;

  // matcher.cpp:166-168  isInFrame(px.cast<int>()/(1<<level), halfpatch_size_+2, level)
  const int ox = (int)rpx0 / (1 << level), oy = (int)rpx1 / (1 << level);
  const int boundary = 6;
  const bool in_frame = ox >= boundary && ox < b.cam_width / (1 << level) - boundary && oy >= boundary && oy < b.cam_height / (1 << level) - boundary;
  if (in_frame) {
    // ---- warp::getWarpMatrixAffine (matcher.cpp:40-68) ----
    const SE3d T_ref = se3_load(b.frame_T + 7 * rf), T_cur = se3_load(b.frame_T + 7 * cf);
    const SE3d T_ref_inv = se3_inv(T_ref);
    const SE3d T_cur_ref = se3_mul(T_cur, T_ref_inv);
    const double d0 = T_ref_inv.t[0] - b.pos[3 * i], d1 = T_ref_inv.t[1] - b.pos[3 * i + 1], d2 = T_ref_inv.t[2] - b.pos[3 * i + 2];
    const double depth_ref = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    double A[4];
    {
      const int halfpatch_size = 5;
      const double xyz_ref[3] = { b.ref_f[3 * i] * depth_ref, b.ref_f[3 * i + 1] * depth_ref, b.ref_f[3 * i + 2] * depth_ref };
      double xyz_du_ref[3], xyz_dv_ref[3];
      cam2world(b, rpx0 + (double)halfpatch_size * (1 << level), rpx1 + 0.0 * (1 << level), xyz_du_ref);
      cam2world(b, rpx0 + 0.0 * (1 << level), rpx1 + (double)halfpatch_size * (1 << level), xyz_dv_ref);
      const double su = xyz_ref[2] / xyz_du_ref[2], sv = xyz_ref[2] / xyz_dv_ref[2];
      for (int k = 0; k < 3; ++k) { xyz_du_ref[k] *= su; xyz_dv_ref[k] *= sv; }
      double c3[3], pc[2], pdu[2], pdv[2];
      se3_act(T_cur_ref, xyz_ref, c3);    world2cam(b, c3, pc);
      se3_act(T_cur_ref, xyz_du_ref, c3); world2cam(b, c3, pdu);
      se3_act(T_cur_ref, xyz_dv_ref, c3); world2cam(b, c3, pdv);
      A[0] = (pdu[0] - pc[0]) / halfpatch_size; A[2] = (pdu[1] - pc[1]) / halfpatch_size;
      A[1] = (pdv[0] - pc[0]) / halfpatch_size; A[3] = (pdv[1] - pc[1]) / halfpatch_size;
    }
    // ---- warp::getBestSearchLevel (matcher.cpp:70-84) ----
    {
      search_level = 0;
      double D = A[0] * A[3] - A[2] * A[1];
      const int max_level = b.n_pyr_levels - 1;
      while (D > 3.0 && search_level < max_level) { search_level += 1; D *= 0.25; }
    }
    // ---- warp::warpAffine into patch_with_border_ (matcher.cpp:86-128), halfpatch 5 ----
    const double det = A[0] * A[3] - A[2] * A[1];
    const double invdet = 1.0 / det;
    const float a00 = (float)(A[3] * invdet), a01 = (float)(-A[1] * invdet);
    const float a10 = (float)(-A[2] * invdet), a11 = (float)(A[0] * invdet);
    if (!(a00 != a00)) {
      {
        const int rcols = b.width >> level, rrows = b.height >> level;
        const uint8_t* img_ref = b.pyr_base + (unsigned long long)b.frame_slot[rf] * b.slot_bytes + pyr_level_offset(b.width, b.height, level);
        const float rx = (float)rpx0 / (float)(1 << level), ry = (float)rpx1 / (float)(1 << level);
        const float fscale = (float)(1 << search_level);
        for (int y = 0; y < PB_ROWS; ++y) {
          uint32_t w[PB_WORDS] = { 0u, 0u, 0u };
#pragma unroll
          for (int x = 0; x < PB_STEP; ++x) {
            float ppx = (float)(x - 5), ppy = (float)(y - 5);
            ppx *= fscale; ppy *= fscale;
            const float px0 = (a00 * ppx + a01 * ppy) + rx;
            const float px1 = (a10 * ppx + a11 * ppy) + ry;
            uint32_t val = 0u;
            if (!(px0 < 0 || px1 < 0 || px0 >= rcols - 1 || px1 >= rrows - 1))
              val = (uint32_t)(uint8_t)interpolate_mat_8u(img_ref, rcols, px0, px1);
            w[x >> 2] |= val << (8 * (x & 3));
          }
          PBW(y, 0) = w[0]; PBW(y, 1) = w[1]; PBW(y, 2) = w[2];
        }
      }
      // every lane reads back only its own words: no barrier needed
      const int cols = b.width >> search_level, rows = b.height >> search_level;
      const uint8_t* cur_img = b.pyr_base + (unsigned long long)b.frame_slot[cf] * b.slot_bytes + pyr_level_offset(b.width, b.height, search_level);
      const double scale = (double)(1 << search_level);
      double est0 = px_cur[0] / scale, est1 = px_cur[1] / scale;     // px_scaled (matcher.cpp:183)
      const float min_update_squared = (float)(0.03 * 0.03);
      const int n_iter = b.align_max_iter;

      if (b.ref_type[i] == PLSVO_FTR_EDGELET) {
        // ---- feature_alignment::align1D (feature_alignment.cpp:41-157) ----
        double dc0 = A[0] * b.ref_grad[2 * i] + A[1] * b.ref_grad[2 * i + 1], dc1 = A[2] * b.ref_grad[2 * i] + A[3] * b.ref_grad[2 * i + 1];
        const double nrm = sqrt(dc0 * dc0 + dc1 * dc1);
        dc0 /= nrm; dc1 /= nrm;
        const float dir0 = (float)dc0, dir1 = (float)dc1;
        float H00 = 0, H01 = 0, H10 = 0, H11 = 0;
        for (int y = 0; y < 8; ++y) {
          const uint32_t t0 = PBW(y, 0), t1 = PBW(y, 1), t2 = PBW(y, 2);
          const uint32_t m0 = PBW(y + 1, 0), m1 = PBW(y + 1, 1), m2 = PBW(y + 1, 2);
          const uint32_t u0 = PBW(y + 2, 0), u1 = PBW(y + 2, 1), u2 = PBW(y + 2, 2);
#pragma unroll
          for (int x = 0; x < 8; ++x) {
            const float J0 = (float)(0.5 * (double)(dir0 * (float)(row_byte(m0, m1, m2, x + 2) - row_byte(m0, m1, m2, x)) +
                                                    dir1 * (float)(row_byte(u0, u1, u2, x + 1) - row_byte(t0, t1, t2, x + 1))));
            const float J1 = 1;
            H00 += J0 * J0; H01 += J0 * J1; H10 += J1 * J0; H11 += J1 * J1;
          }
        }
        const float detf = H00 * H11 - H10 * H01;
        const float invdetf = 1.0f / detf;
        const float Hi00 = H11 * invdetf, Hi10 = -H10 * invdetf, Hi01 = -H01 * invdetf, Hi11 = H00 * invdetf;
        float mean_diff = 0;
        float u = (float)est0, v = (float)est1;
        float chi2 = 0, up0 = 0, up1 = 0;
        bool early_return = false;
        int iter = 0;
        for (; iter < n_iter; ++iter) {
          const int u_r = f2i_trunc(floorf(u)), v_r = f2i_trunc(floorf(v));
          if (u_r < 4 || v_r < 4 || u_r >= cols - 4 || v_r >= rows - 4) break;
          if (u != u || v != v) { early_return = true; break; }
          const float subpix_x = u - u_r, subpix_y = v - v_r;
          const float wTL = (float)((1.0 - subpix_x) * (1.0 - subpix_y));
          const float wTR = (float)(subpix_x * (1.0 - subpix_y));
          const float wBL = (float)((1.0 - subpix_x) * subpix_y);
          const float wBR = subpix_x * subpix_y;
          float new_chi2 = 0.0f, Jr0 = 0, Jr1 = 0;
          long off = (long)(v_r - 4) * cols + (u_r - 4);
          Row9 top = load_row9(cur_img, off);
          for (int y = 0; y < 8; ++y) {
            off += cols;
            const Row9 bot = load_row9(cur_img, off);
            const uint32_t t0 = PBW(y, 0), t1 = PBW(y, 1), t2 = PBW(y, 2);
            const uint32_t m0 = PBW(y + 1, 0), m1 = PBW(y + 1, 1), m2 = PBW(y + 1, 2);
            const uint32_t u0 = PBW(y + 2, 0), u1 = PBW(y + 2, 1), u2 = PBW(y + 2, 2);
#pragma unroll
            for (int x = 0; x < 8; ++x) {
              const float dv = (float)(0.5 * (double)(dir0 * (float)(row_byte(m0, m1, m2, x + 2) - row_byte(m0, m1, m2, x)) +
                                                      dir1 * (float)(row_byte(u0, u1, u2, x + 1) - row_byte(t0, t1, t2, x + 1))));
              const float search_pixel = wTL * (float)row_byte(top.a, top.b, top.c, x) + wTR * (float)row_byte(top.a, top.b, top.c, x + 1) +
                                         wBL * (float)row_byte(bot.a, bot.b, bot.c, x) + wBR * (float)row_byte(bot.a, bot.b, bot.c, x + 1);
              const float res = search_pixel - (float)row_byte(m0, m1, m2, x + 1) + mean_diff;
              Jr0 -= res * dv;
              Jr1 -= res;
              new_chi2 += res * res;
            }
            top = bot;
          }
          if (iter > 0 && new_chi2 > chi2) { u -= up0; v -= up1; ++iter; break; }
          chi2 = new_chi2;
          up0 = Hi00 * Jr0 + Hi01 * Jr1;
          up1 = Hi10 * Jr0 + Hi11 * Jr1;
          u += up0 * dir0;
          v += up0 * dir1;
          mean_diff += up1;
          if (up0 * up0 + up1 * up1 < min_update_squared) { found = 1; ++iter; break; }
        }
        iters = iter;
        if (!early_return) { est0 = u; est1 = v; }
      } else {
        // ---- feature_alignment::align2D (feature_alignment.cpp:159-283) ----
        float H[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
        for (int y = 0; y < 8; ++y) {
          const uint32_t t0 = PBW(y, 0), t1 = PBW(y, 1), t2 = PBW(y, 2);
          const uint32_t m0 = PBW(y + 1, 0), m1 = PBW(y + 1, 1), m2 = PBW(y + 1, 2);
          const uint32_t u0 = PBW(y + 2, 0), u1 = PBW(y + 2, 1), u2 = PBW(y + 2, 2);
#pragma unroll
          for (int x = 0; x < 8; ++x) {
            float J[3];
            J[0] = 0.5f * (float)(row_byte(m0, m1, m2, x + 2) - row_byte(m0, m1, m2, x));      // exact: |int| <= 255
            J[1] = 0.5f * (float)(row_byte(u0, u1, u2, x + 1) - row_byte(t0, t1, t2, x + 1));
            J[2] = 1;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
              for (int c = 0; c < 3; ++c) H[r][c] += J[r] * J[c];
          }
        }
#define COF(i, j) (H[((i) + 1) % 3][((j) + 1) % 3] * H[((i) + 2) % 3][((j) + 2) % 3] - H[((i) + 1) % 3][((j) + 2) % 3] * H[((i) + 2) % 3][((j) + 1) % 3])
        float Hinv[3][3];
        {
          const float c00 = COF(0, 0), c10 = COF(1, 0), c20 = COF(2, 0);
          const float detf = c00 * H[0][0] + (c10 * H[1][0] + c20 * H[2][0]);
          const float invdetf = 1.0f / detf;
          Hinv[0][0] = c00 * invdetf; Hinv[0][1] = c10 * invdetf; Hinv[0][2] = c20 * invdetf;
          Hinv[1][0] = COF(0, 1) * invdetf; Hinv[1][1] = COF(1, 1) * invdetf; Hinv[1][2] = COF(2, 1) * invdetf;
          Hinv[2][0] = COF(0, 2) * invdetf; Hinv[2][1] = COF(1, 2) * invdetf; Hinv[2][2] = COF(2, 2) * invdetf;
        }
#undef COF
        float mean_diff = 0;
        float u = (float)est0, v = (float)est1;
        int iter = 0;
        for (; iter < n_iter; ++iter) {
          const float u_ref = (float)est0, v_ref = (float)est1;      // Patch::setPosition(cur_px_estimate)
          const int u_ref_i = f2i_trunc(floorf(u_ref)), v_ref_i = f2i_trunc(floorf(v_ref));
          if (u_ref_i < 4 || v_ref_i < 4 || u_ref_i >= cols - 4 || v_ref_i >= rows - 4) break;
          const float subpix_u = u_ref - u_ref_i, subpix_v = v_ref - v_ref_i;
          const float wTL = (float)((1.0 - subpix_u) * (1.0 - subpix_v));
          const float wTR = (float)(subpix_u * (1.0 - subpix_v));
          const float wBL = (float)((1.0 - subpix_u) * subpix_v);
          const float wBR = subpix_u * subpix_v;
          float Jr0 = 0, Jr1 = 0, Jr2 = 0;
          long off = (long)(v_ref_i - 4) * cols + (u_ref_i - 4);
          Row9 top = load_row9(cur_img, off);
          for (int y = 0; y < 8; ++y) {
            off += cols;
            const Row9 bot = load_row9(cur_img, off);
            const uint32_t t0 = PBW(y, 0), t1 = PBW(y, 1), t2 = PBW(y, 2);
            const uint32_t m0 = PBW(y + 1, 0), m1 = PBW(y + 1, 1), m2 = PBW(y + 1, 2);
            const uint32_t u0 = PBW(y + 2, 0), u1 = PBW(y + 2, 1), u2 = PBW(y + 2, 2);
#pragma unroll
            for (int x = 0; x < 8; ++x) {
              const float dx = 0.5f * (float)(row_byte(m0, m1, m2, x + 2) - row_byte(m0, m1, m2, x));
              const float dy = 0.5f * (float)(row_byte(u0, u1, u2, x + 1) - row_byte(t0, t1, t2, x + 1));
              const float search_pixel = wTL * (float)row_byte(top.a, top.b, top.c, x) + wTR * (float)row_byte(top.a, top.b, top.c, x + 1) +
                                         wBL * (float)row_byte(bot.a, bot.b, bot.c, x) + wBR * (float)row_byte(bot.a, bot.b, bot.c, x + 1);
              const float res = search_pixel - (float)row_byte(m0, m1, m2, x + 1) + mean_diff;
              Jr0 -= res * dx;
              Jr1 -= res * dy;
              Jr2 -= res;
            }
            top = bot;
          }
          const float up0 = Hinv[0][0] * Jr0 + Hinv[0][1] * Jr1 + Hinv[0][2] * Jr2;
          const float up1 = Hinv[1][0] * Jr0 + Hinv[1][1] * Jr1 + Hinv[1][2] * Jr2;
          const float up2 = Hinv[2][0] * Jr0 + Hinv[2][1] * Jr1 + Hinv[2][2] * Jr2;
          u += up0; v += up1;
          est0 = u; est1 = v;
          mean_diff += up2;
          if (up0 * up0 + up1 * up1 < min_update_squared) { found = 1; ++iter; break; }
        }
        iters = iter;
        est0 = u; est1 = v;
      }
      px_cur[0] = est0 * scale; px_cur[1] = est1 * scale;
    }
  }
#undef PBW
  b.px_out[2 * i] = px_cur[0]; b.px_out[2 * i + 1] = px_cur[1];
  b.found[i] = (uint8_t)found;
  b.search_level[i] = search_level;
  b.n_iter[i] = iters;
}

// Reprojector::reproject (src/reprojector.cpp:387-423): one lane per landmark position
__global__ void __launch_bounds__(256) reproject_kernel(const ReprojBatchDev b) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= b.n) return;
  const SE3d T = se3_load(b.frame_T + 7 * b.frame[i]);
  double c[3];
  se3_act(T, b.pos + 3 * i, c);
  const double u = c[0] / c[2], v = c[1] / c[2];
  const double px0 = b.fx * u + b.cx, px1 = b.fy * v + b.cy;
  int cell = -1;
  if (px0 == px0 && px1 == px1 && fabs(px0) < 1e9 && fabs(px1) < 1e9) {
    const int ox = (int)px0, oy = (int)px1;
    if (ox >= b.boundary && ox < b.cam_width - b.boundary && oy >= b.boundary && oy < b.cam_height - b.boundary)
      cell = (int)(px1 / b.cell_size) * b.grid_n_cols + (int)(px0 / b.cell_size);
  }
  b.px[2 * i] = px0; b.px[2 * i + 1] = px1;
  b.cell[i] = cell;
}

hipError_t launch_reproject(const ReprojBatchDev& b, hipStream_t stream) {
  if (b.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(reproject_kernel, dim3((b.n + 255) / 256), dim3(256), 0, stream, b);
  return hipGetLastError();
}

hipError_t launch_match_direct(const MatchBatchDev& b, hipStream_t stream) {
  if (b.n <= 0) return hipSuccess;
  const int blocks = (b.n + MT - 1) / MT;
  hipLaunchKernelGGL(match_direct_kernel, dim3(blocks), dim3(MT), 0, stream, b);
  return hipGetLastError();
}

}  // namespace plsvo_hip
