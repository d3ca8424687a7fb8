// match_device.hpp -- device building blocks shared by the direct matcher (match_kernels.hip) and the depth-filter
// seed update (seeds_kernels.hip): pinhole camera, the affine patch warp into a per-lane LDS patch, and the 8x8
// inverse-compositional alignment on that patch.  Everything here follows the reference's expression order and the
// including translation units are compiled with -ffp-contract=off, so results are bit-identical to the CPU oracle.
//
//   warp::getWarpMatrixAffine / getBestSearchLevel / warpAffine        src/matcher.cpp:44-129
//   Matcher::createPatchFromPatchWithBorder                            src/matcher.cpp:148-157 (implicit: the 8x8 patch is
//                                                                      the interior of the LDS-resident 10x10 one)
//   feature_alignment::align2D                                         src/feature_alignment.cpp:160-290
//   [ext] vk::PinholeCamera::world2cam / cam2world, vk::interpolateMat_8u, Eigen 2x2 / 3x3 inverse()
#pragma once
#include <hip/hip_runtime.h>

#include "plsvo_dev.hpp"
#include "plsvo_math.hpp"

namespace plsvo_hip {

#pragma clang fp contract(off)

constexpr int MT = 64;          // lanes (candidates / seeds) per workgroup
constexpr int PB_STEP = 10;     // patch_with_border row length (patch_size_+2)
constexpr int PB_WORDS = 3;     // dwords per LDS row (10 bytes + 2 pad)
constexpr int PB_ROWS = 10;
// the warped 10x10 reference patch of lane `lane`, word-major / lane-minor in LDS (no bank conflicts)
#define PBW(my, row, w) (my)[((row) * PB_WORDS + (w)) * MT]

struct CamDev { double fx, fy, cx, cy; int width, height; };

__device__ __forceinline__ int f2i_trunc(float x) { return (x != x) ? INT32_MIN : (int)x; }

// [ext] vk::PinholeCamera::world2cam / cam2world without distortion
__device__ __forceinline__ void world2cam(const CamDev& c, const double* xyz, double* px) {
  const double u = xyz[0] / xyz[2], v = xyz[1] / xyz[2];
  px[0] = c.fx * u + c.cx;
  px[1] = c.fy * v + c.cy;
}
__device__ __forceinline__ void cam2world(const CamDev& c, double pu, double pv, double* f) {
  const double x = (pu - c.cx) / c.fx, y = (pv - c.cy) / c.fy, z = 1.0;
  const double n = sqrt(x * x + y * y + z * z);
  f[0] = x / n; f[1] = y / n; f[2] = z / n;
}
// [ext] vk::AbstractCamera::isInFrame(Vector2i obs, int boundary, int level)
__device__ __forceinline__ bool cam_is_in_frame(const CamDev& c, int ox, int oy, int boundary, int level) {
  return ox >= boundary && ox < c.width / (1 << level) - boundary && oy >= boundary && oy < c.height / (1 << level) - boundary;
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

// warp::getWarpMatrixAffine (matcher.cpp:44-71); A is row-major {a00, a01, a10, a11}
__device__ __forceinline__ void warp_matrix_affine(const CamDev& cam, double rpx0, double rpx1, const double* f_ref, double depth_ref,
                                                   const SE3d& T_cur_ref, int level, double* A) {
  const int halfpatch_size = 5;
  const double xyz_ref[3] = { f_ref[0] * depth_ref, f_ref[1] * depth_ref, f_ref[2] * depth_ref };
  double xyz_du_ref[3], xyz_dv_ref[3];
  cam2world(cam, rpx0 + (double)halfpatch_size * (1 << level), rpx1 + 0.0 * (1 << level), xyz_du_ref);
  cam2world(cam, rpx0 + 0.0 * (1 << level), rpx1 + (double)halfpatch_size * (1 << level), xyz_dv_ref);
  const double su = xyz_ref[2] / xyz_du_ref[2], sv = xyz_ref[2] / xyz_dv_ref[2];
  for (int k = 0; k < 3; ++k) { xyz_du_ref[k] *= su; xyz_dv_ref[k] *= sv; }
  double c3[3], pc[2], pdu[2], pdv[2];
  se3_act(T_cur_ref, xyz_ref, c3);    world2cam(cam, c3, pc);
  se3_act(T_cur_ref, xyz_du_ref, c3); world2cam(cam, c3, pdu);
  se3_act(T_cur_ref, xyz_dv_ref, c3); world2cam(cam, c3, pdv);
  A[0] = (pdu[0] - pc[0]) / halfpatch_size; A[2] = (pdu[1] - pc[1]) / halfpatch_size;
  A[1] = (pdv[0] - pc[0]) / halfpatch_size; A[3] = (pdv[1] - pc[1]) / halfpatch_size;
}

// warp::getBestSearchLevel (matcher.cpp:73-86)
__device__ __forceinline__ int best_search_level(const double* A, int max_level) {
  int search_level = 0;
  double D = A[0] * A[3] - A[2] * A[1];
  while (D > 3.0 && search_level < max_level) { search_level += 1; D *= 0.25; }
  return search_level;
}

// the source window of a warp staged in LDS (match_direct_kernel): SRC_ROWS rows of SRC_WORDS dwords per lane, word-major / lane-minor, filled
// once per GROUP of SRC_GROUP rows of the 10x10 patch
constexpr int SRC_GROUP = 2;
constexpr int SRC_ROWS = 6;
constexpr int SRC_WORDS = 5;
#define SRCW(src, row, w) (src)[((row) * SRC_WORDS + (w)) * MT]

// warp::warpAffine with halfpatch 5 into the lane's LDS patch (matcher.cpp:88-129).  False when the inverse warp is
// NaN (the reference then keeps the Matcher's previous patch, :99-103; such a candidate is reported as not found).
//
// `src` (optional): SRC_ROWS x SRC_WORDS dwords of LDS per lane.  The reference reads four single bytes of the keyframe image per warped
// pixel (vk::interpolateMat_8u: 400 scattered byte reads per candidate, every one a line the CU's 32 KB L1 has long evicted when 64 lanes
// warp 64 different patches -- the kernel waits on them, it does not compute).  The warp is affine, so the pixels a block of patch rows
// touches lie inside the box of the block's four corners (each corner expression is monotone in both grid coordinates, rounding
// included): the patch is warped in groups of SRC_GROUP rows, and when a group's box is at most SRC_ROWS x 4 SRC_WORDS - 1 bytes (every
// warp that neither zooms by more than ~1.6 nor turns by more than ~20 degrees: the rule, not the exception) its rows are fetched ONCE as
// aligned dwords, parked in LDS lane-interleaved, and the same float expressions read their four taps from there; a group whose box is
// larger reads its taps from the image.  Same values, same order: bit-identical results.
// (Round 5 staged the box of the WHOLE patch, 14 rows: 280 B of LDS per lane capped the kernel at six one-wave workgroups per CU, and a
//  candidate is a chain of dependent float additions -- the reference's sequential sums -- that wants many waves per SIMD, not few.  Five
//  groups of <= 6 rows re-read ~40 % more image rows from L1 / L2 and take 120 B per lane: ten workgroups per CU.)
__device__ __forceinline__ bool warp_affine_lds(const double* A, const uint8_t* img_ref, int rcols, int rrows, double rpx0, double rpx1,
                                                int level, int search_level, uint32_t* my, uint32_t* src = nullptr) {
  const double det = A[0] * A[3] - A[2] * A[1];
  const double invdet = 1.0 / det;
  const float a00 = (float)(A[3] * invdet), a01 = (float)(-A[1] * invdet);
  const float a10 = (float)(-A[2] * invdet), a11 = (float)(A[0] * invdet);
  if (a00 != a00) return false;
  const float rx = (float)rpx0 / (float)(1 << level), ry = (float)rpx1 / (float)(1 << level);
  const float fscale = (float)(1 << search_level);
  const uint8_t* const src_b = reinterpret_cast<const uint8_t*>(src);
  static_assert(PB_ROWS % SRC_GROUP == 0, "the patch rows are warped in whole groups");
  for (int y0 = 0; y0 < PB_ROWS; y0 += SRC_GROUP) {
    bool staged = false;
    int cbase = 0, rmin = 0;
    if (src) {
      // the four corners of the group's 10 x SRC_GROUP grid, with the very expressions of the pixel loop
      float lo0 = 3.0e38f, hi0 = -3.0e38f, lo1 = 3.0e38f, hi1 = -3.0e38f;
      bool inside = true;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float ppx = (float)((k & 1) ? 4 : -5), ppy = (float)(((k & 2) ? y0 + SRC_GROUP - 1 : y0) - 5);
        ppx *= fscale; ppy *= fscale;
        const float px0 = (a00 * ppx + a01 * ppy) + rx;
        const float px1 = (a10 * ppx + a11 * ppy) + ry;
        inside = inside && !(px0 < 0 || px1 < 0 || px0 >= rcols - 1 || px1 >= rrows - 1) && px0 == px0 && px1 == px1;
        lo0 = fminf(lo0, px0); hi0 = fmaxf(hi0, px0); lo1 = fminf(lo1, px1); hi1 = fmaxf(hi1, px1);
      }
      if (inside) {
        const int cmin = (int)floorf(lo0), cmax = (int)floorf(hi0) + 1, rmax = (int)floorf(hi1) + 1;
        rmin = (int)floorf(lo1);
        cbase = cmin & ~3;
        staged = (cmax - cbase) <= 4 * SRC_WORDS - 1 && (rmax - rmin) <= SRC_ROWS - 1;
        if (staged) {
          const int n_rows = rmax - rmin + 1;
#pragma unroll 2
          for (int r = 0; r < SRC_ROWS; ++r) {
            if (r < n_rows) {   // (rows rmin .. rmax <= rrows - 1: inside the level; the dword reads run <= 27 bytes past a row's end: the level's slack)
              const long off = (long)(rmin + r) * rcols + cbase;
              const uint32_t* p = reinterpret_cast<const uint32_t*>(img_ref + (off & ~3l));
              const uint32_t sh = (uint32_t)(off & 3);
              const uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4], d5 = p[5];
              SRCW(src, r, 0) = __builtin_amdgcn_alignbyte(d1, d0, sh); SRCW(src, r, 1) = __builtin_amdgcn_alignbyte(d2, d1, sh);
              SRCW(src, r, 2) = __builtin_amdgcn_alignbyte(d3, d2, sh); SRCW(src, r, 3) = __builtin_amdgcn_alignbyte(d4, d3, sh);
              SRCW(src, r, 4) = __builtin_amdgcn_alignbyte(d5, d4, sh);
            }
          }
        }
      }
    }
    auto tap = [&](int lr, int lc) -> float {   // byte (row rmin + lr, column cbase + lc) of the staged window
      return (float)src_b[((lr * SRC_WORDS + (lc >> 2)) * MT) * 4 + (lc & 3)];
    };
    for (int y = y0; y < y0 + SRC_GROUP; ++y) {
      uint32_t w[PB_WORDS] = { 0u, 0u, 0u };
#pragma unroll
      for (int x = 0; x < PB_STEP; ++x) {
        float ppx = (float)(x - 5), ppy = (float)(y - 5);
        ppx *= fscale; ppy *= fscale;
        const float px0 = (a00 * ppx + a01 * ppy) + rx;
        const float px1 = (a10 * ppx + a11 * ppy) + ry;
        uint32_t val = 0u;
        if (staged) {
          // [ext] vk::interpolateMat_8u, its four taps read from the staged window (every pixel is inside the image: the corners are)
          const int xi = (int)floorf(px0), yi = (int)floorf(px1);
          const float subpix_x = px0 - xi, subpix_y = px1 - yi;
          const float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
          const float w01 = (1.0f - subpix_x) * subpix_y;
          const float w10 = subpix_x * (1.0f - subpix_y);
          const float w11 = 1.0f - w00 - w01 - w10;
          const int lr = yi - rmin, lc = xi - cbase;
          val = (uint32_t)(uint8_t)(w00 * tap(lr, lc) + w01 * tap(lr + 1, lc) + w10 * tap(lr, lc + 1) + w11 * tap(lr + 1, lc + 1));
        } else if (!(px0 < 0 || px1 < 0 || px0 >= rcols - 1 || px1 >= rrows - 1)) {
          val = (uint32_t)(uint8_t)interpolate_mat_8u(img_ref, rcols, px0, px1);
        }
        w[x >> 2] |= val << (8 * (x & 3));
      }
      PBW(my, y, 0) = w[0]; PBW(my, y, 1) = w[1]; PBW(my, y, 2) = w[2];
    }
  }
  return true;   // every lane reads back only its own words: no barrier needed
}

// feature_alignment::align2D (feature_alignment.cpp:160-290) on the lane's LDS patch; est = cur_px_estimate in/out,
// returns `converged`, iters = residual passes executed.  ref_patch_dx/dy are never materialised: 0.5*(it[1]-it[-1])
// is exact in float and is recomputed from the LDS bytes.
__device__ __forceinline__ bool align2d_lds(const uint8_t* cur_img, int cols, int rows, const uint32_t* my, int n_iter,
                                            double& est0, double& est1, int& iters) {
  const float min_update_squared = (float)(0.03 * 0.03);
  bool converged = false;
  float H[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
  for (int y = 0; y < 8; ++y) {
    const uint32_t t0 = PBW(my, y, 0), t1 = PBW(my, y, 1), t2 = PBW(my, y, 2);
    const uint32_t m0 = PBW(my, y + 1, 0), m1 = PBW(my, y + 1, 1), m2 = PBW(my, y + 1, 2);
    const uint32_t u0 = PBW(my, y + 2, 0), u1 = PBW(my, y + 2, 1), u2 = PBW(my, y + 2, 2);
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
#define PLSVO_COF(i, j) (H[((i) + 1) % 3][((j) + 1) % 3] * H[((i) + 2) % 3][((j) + 2) % 3] - H[((i) + 1) % 3][((j) + 2) % 3] * H[((i) + 2) % 3][((j) + 1) % 3])
  float Hinv[3][3];
  {
    const float c00 = PLSVO_COF(0, 0), c10 = PLSVO_COF(1, 0), c20 = PLSVO_COF(2, 0);
    const float detf = c00 * H[0][0] + (c10 * H[1][0] + c20 * H[2][0]);
    const float invdetf = 1.0f / detf;
    Hinv[0][0] = c00 * invdetf; Hinv[0][1] = c10 * invdetf; Hinv[0][2] = c20 * invdetf;
    Hinv[1][0] = PLSVO_COF(0, 1) * invdetf; Hinv[1][1] = PLSVO_COF(1, 1) * invdetf; Hinv[1][2] = PLSVO_COF(2, 1) * invdetf;
    Hinv[2][0] = PLSVO_COF(0, 2) * invdetf; Hinv[2][1] = PLSVO_COF(1, 2) * invdetf; Hinv[2][2] = PLSVO_COF(2, 2) * invdetf;
  }
#undef PLSVO_COF
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
      const uint32_t t0 = PBW(my, y, 0), t1 = PBW(my, y, 1), t2 = PBW(my, y, 2);
      const uint32_t m0 = PBW(my, y + 1, 0), m1 = PBW(my, y + 1, 1), m2 = PBW(my, y + 1, 2);
      const uint32_t u0 = PBW(my, y + 2, 0), u1 = PBW(my, y + 2, 1), u2 = PBW(my, y + 2, 2);
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
    if (up0 * up0 + up1 * up1 < min_update_squared) { converged = true; ++iter; break; }
  }
  iters = iter;
  est0 = u; est1 = v;
  return converged;
}

}  // namespace plsvo_hip
