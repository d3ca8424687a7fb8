// match_kernels.hip -- direct feature matching on gfx950: one LANE per match candidate.
//
// Replaces (reference file:line):
//   plsvo::Matcher::findMatchDirect (Point)     src/matcher.cpp:159-207
//   plsvo::Matcher::findMatchDirect (LineSeg)   src/matcher.cpp:232-275   (two candidates, ANDed by the caller)
//   Matcher::precomputeRefPatch                 src/matcher.cpp:209-230
//   Matcher::createPatchFromPatchWithBorder     src/matcher.cpp:148-157
//   warp::getWarpMatrixAffine / getBestSearchLevel / warpAffine   src/matcher.cpp:44-129
//   feature_alignment::align1D / align2D        src/feature_alignment.cpp:41-158, :159-283
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

#include "match_device.hpp"

namespace plsvo_hip {

#pragma clang fp contract(off)

__global__ void __launch_bounds__(MT) match_direct_kernel(const MatchBatchDev b) {
  __shared__ uint32_t s_patch[PB_ROWS * PB_WORDS * MT];   // patch_with_border_, word-major / lane-minor
  __shared__ uint32_t s_src[SRC_ROWS * SRC_WORDS * MT];   // the keyframe-image window a warp reads, staged once (match_device.hpp::warp_affine_lds)
  const int lane = threadIdx.x;
  const int i = blockIdx.x * MT + lane;
  if (i >= b.n) return;
  uint32_t* my = s_patch + lane;
  const CamDev cam{b.fx, b.fy, b.cx, b.cy, b.cam_width, b.cam_height};

  const int rf = b.ref_frame[i], cf = b.cur_frame[i], level = b.ref_level[i];
  double px_cur[2] = { b.px_cur[2 * i], b.px_cur[2 * i + 1] };
  const double rpx0 = b.ref_px[2 * i], rpx1 = b.ref_px[2 * i + 1];
  int found = 0, search_level = -1, iters = 0;
  const bool wanted = !b.active || b.active[i];   // resident frame step: the candidate was not visible / not selected for matching

  // matcher.cpp:168-170  isInFrame(px.cast<int>()/(1<<level), halfpatch_size_+2, level)
  if (wanted && cam_is_in_frame(cam, (int)rpx0 / (1 << level), (int)rpx1 / (1 << level), 6, level)) {
    const SE3d T_ref = se3_load(b.frame_T + 7 * rf), T_cur = se3_load(b.frame_T + 7 * cf);
    const SE3d T_ref_inv = se3_inv(T_ref);
    const SE3d T_cur_ref = se3_mul(T_cur, T_ref_inv);
    const double d0 = T_ref_inv.t[0] - b.pos[3 * i], d1 = T_ref_inv.t[1] - b.pos[3 * i + 1], d2 = T_ref_inv.t[2] - b.pos[3 * i + 2];
    const double depth_ref = sqrt(d0 * d0 + d1 * d1 + d2 * d2);     // (ref_frame.pos() - pt.pos_).norm()
    double A[4];
    warp_matrix_affine(cam, rpx0, rpx1, b.ref_f + 3 * i, depth_ref, T_cur_ref, level, A);
    search_level = best_search_level(A, b.n_pyr_levels - 1);
    const uint8_t* img_ref = b.pyr_base + (unsigned long long)b.frame_slot[rf] * b.slot_bytes + pyr_level_offset(b.width, b.height, level);
    if (warp_affine_lds(A, img_ref, b.width >> level, b.height >> level, rpx0, rpx1, level, search_level, my, s_src + lane)) {
      const int cols = b.width >> search_level, rows = b.height >> search_level;
      const uint8_t* cur_img = b.pyr_base + (unsigned long long)b.frame_slot[cf] * b.slot_bytes + pyr_level_offset(b.width, b.height, search_level);
      const double scale = (double)(1 << search_level);
      double est0 = px_cur[0] / scale, est1 = px_cur[1] / scale;     // px_scaled (matcher.cpp:187)
      const float min_update_squared = (float)(0.03 * 0.03);
      const int n_iter = b.align_max_iter;
      if (b.ref_type[i] == PLSVO_FTR_EDGELET) {
        // ---- feature_alignment::align1D (feature_alignment.cpp:41-158) ----
        double dc0 = A[0] * b.ref_grad[2 * i] + A[1] * b.ref_grad[2 * i + 1], dc1 = A[2] * b.ref_grad[2 * i] + A[3] * b.ref_grad[2 * i + 1];
        const double nrm = sqrt(dc0 * dc0 + dc1 * dc1);
        dc0 /= nrm; dc1 /= nrm;
        const float dir0 = (float)dc0, dir1 = (float)dc1;
        float H00 = 0, H01 = 0, H10 = 0, H11 = 0;
        for (int y = 0; y < 8; ++y) {
          const uint32_t t0 = PBW(my, y, 0), t1 = PBW(my, y, 1), t2 = PBW(my, y, 2);
          const uint32_t m0 = PBW(my, y + 1, 0), m1 = PBW(my, y + 1, 1), m2 = PBW(my, y + 1, 2);
          const uint32_t u0 = PBW(my, y + 2, 0), u1 = PBW(my, y + 2, 1), u2 = PBW(my, y + 2, 2);
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
            const uint32_t t0 = PBW(my, y, 0), t1 = PBW(my, y, 1), t2 = PBW(my, y, 2);
            const uint32_t m0 = PBW(my, y + 1, 0), m1 = PBW(my, y + 1, 1), m2 = PBW(my, y + 1, 2);
            const uint32_t u0 = PBW(my, y + 2, 0), u1 = PBW(my, y + 2, 1), u2 = PBW(my, y + 2, 2);
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
        found = align2d_lds(cur_img, cols, rows, my, n_iter, est0, est1, iters) ? 1 : 0;
      }
      px_cur[0] = est0 * scale; px_cur[1] = est1 * scale;
    }
  }
  b.px_out[2 * i] = px_cur[0]; b.px_out[2 * i + 1] = px_cur[1];
  b.found[i] = (uint8_t)found;
  b.search_level[i] = search_level;
  b.n_iter[i] = iters;
}

// Reprojector::reproject (src/reprojector.cpp:389-423): one lane per landmark position
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
