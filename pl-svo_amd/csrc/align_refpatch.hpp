// align_refpatch.hpp -- the float arithmetic of a reference patch (interpolated intensity + central-difference gradient) of
// align_kernels.hip, every product and sum rounded on its own: the kernel's record form (RecordRows) and, as the statement of the
// reference's precompute it must equal bit for bit, the direct form (ref_row_direct); tests/test_refpatch_host.py compiles both for the host.
// Reference: SparseImgAlign::precomputeGaussNewtonParamsPoints/Segments src/sparse_img_align.cpp:236-264, :348-375;
//            Patch::setPosition + computeInterpWeights src/feature.cpp:189-208.
#pragma once
#ifdef __HIPCC__
#define PLSVO_HD __host__ __device__ __forceinline__
#else
#define PLSVO_HD inline
#endif

namespace plsvo_hip {

// wTL*a + wTR*b + wBL*c + wBR*d, evaluated left to right in float, every product and every sum rounded on its own
// (src/sparse_img_align.cpp:251, 458, 620).  HIP's __fmul_rn / __fadd_rn are plain `*` / `+` and hipcc contracts by default, so
// the guarantee comes from the pragma: without it the compiler fused different products at the two call sites, and a static
// camera (cur == ref, T = I) saw one-ulp residuals where the reference sees exact zeros (tests: static-camera cases).
PLSVO_HD float bilinear(float wTL, float wTR, float wBL, float wBR, float a, float b, float c, float d) {
#pragma clang fp contract(off)
  const float p0 = wTL * a, p1 = wTR * b, p2 = wBL * c, p3 = wBR * d;
  return ((p0 + p1) + p2) + p3;
}

// Patch::setPosition + computeInterpWeights (src/feature.cpp:189-208): position as float, weights
// computed in double and stored as float
struct PatchW { int ui, vi; float wTL, wTR, wBL, wBR; };
PLSVO_HD PatchW patch_weights(float u, float v) {
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

// one patch row from the four image rows around it (I[rr][0..6] = image row y-1+rr, columns ui-3 .. ui+3), as the precompute wrote it
PLSVO_HD void ref_row_direct(const float (*I)[7], float wTL, float wTR, float wBL, float wBR, float4& vr, float4& vx, float4& vy) {
#pragma clang fp contract(off)
  float* pr = reinterpret_cast<float*>(&vr); float* pxp = reinterpret_cast<float*>(&vx); float* pyp = reinterpret_cast<float*>(&vy);
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    const int c = x + 1;  // column of pixel x inside I[.][0..6]
    // B(r,c) = wTL*I[r][c] + wTR*I[r][c+1] + wBL*I[r+1][c] + wBR*I[r+1][c+1]
    const float ref = bilinear(wTL, wTR, wBL, wBR, I[1][c], I[1][c + 1], I[2][c], I[2][c + 1]);
    const float xp = bilinear(wTL, wTR, wBL, wBR, I[1][c + 1], I[1][c + 2], I[2][c + 1], I[2][c + 2]);
    const float xm = bilinear(wTL, wTR, wBL, wBR, I[1][c - 1], I[1][c], I[2][c - 1], I[2][c]);
    const float yp = bilinear(wTL, wTR, wBL, wBR, I[2][c], I[2][c + 1], I[3][c], I[3][c + 1]);
    const float ym = bilinear(wTL, wTR, wBL, wBR, I[0][c], I[0][c + 1], I[1][c], I[1][c + 1]);
    pr[x] = ref;
    pxp[x] = 0.5f * (xp - xm);
    pyp[x] = 0.5f * (yp - ym);
  }
}

// ---- record form of the reference-patch cache (what align_kernels.hip keeps per slot): the 7x7 window of image BYTES around the patch
// (record row k = image row vi-3+k, columns ui-3 .. ui+3, 8 bytes per row) and the two sub-pixel fractions, 64 B per slot instead of
// 3 x 16 floats.  Every iteration rebuilds ref / dx / dy from it with the very operations ref_row_direct uses -- B[k][c] = bilinear over
// record rows k, k+1 at columns c, c+1; ref = B[y+1][x+1], dx = 0.5 (B[y+1][x+2] - B[y+1][x]), dy = 0.5 (B[y+2][x+1] - B[y][x+1]) -- so the
// values, and every result, are bit-identical (tests/test_refpatch_host.py compiles both forms for the host and compares them).
PLSVO_HD void interp_weights(float su, float sv, float& wTL, float& wTR, float& wBL, float& wBR) {   // the arithmetic of patch_weights
  wTL = (float)((1.0 - (double)su) * (1.0 - (double)sv));
  wTR = (float)((double)su * (1.0 - (double)sv));
  wBL = (float)((1.0 - (double)su) * (double)sv);
  wBR = (float)((double)su * (double)sv);
}
PLSVO_HD void unpack_row7(uint32_t lo, uint32_t hi, float* o7) {
  o7[0] = (float)(lo & 0xffu); o7[1] = (float)((lo >> 8) & 0xffu); o7[2] = (float)((lo >> 16) & 0xffu); o7[3] = (float)(lo >> 24);
  o7[4] = (float)(hi & 0xffu); o7[5] = (float)((hi >> 8) & 0xffu); o7[6] = (float)((hi >> 16) & 0xffu);
}
// one B row: columns c0 .. c0+n-1 from two unpacked record rows
template <int C0, int N>
PLSVO_HD void record_b_row(const float* top, const float* bot, float wTL, float wTR, float wBL, float wBR, float* B6) {
#pragma unroll
  for (int c = C0; c < C0 + N; ++c) B6[c] = bilinear(wTL, wTR, wBL, wBR, top[c], top[c + 1], bot[c], bot[c + 1]);
}
// The patch rows of a slot from its 64-byte record (q[k] = bytes 16k .. 16k+15: record rows 2k, 2k+1; q[3].zw = the fractions), one at a
// time: start() prepares B[0] and B[1], row(y) returns patch row y (y = 0, 1, 2, 3 in this order) -- a rolling window over the record
// rows, so that only one patch row of ref / dx / dy is alive beside the pixel arithmetic that consumes it.
struct RecordRows {
  float wTL, wTR, wBL, wBR;
  float ra[7], rb[7], Bm[6], Bc[6];
  uint32_t lo[7], hi[7];
  PLSVO_HD void start(const uint4* q) {
#pragma clang fp contract(off)
    float su, sv;
    { const uint32_t a = q[3].z, b_ = q[3].w; su = *reinterpret_cast<const float*>(&a); sv = *reinterpret_cast<const float*>(&b_); }
    interp_weights(su, sv, wTL, wTR, wBL, wBR);
    lo[0] = q[0].x; lo[1] = q[0].z; lo[2] = q[1].x; lo[3] = q[1].z; lo[4] = q[2].x; lo[5] = q[2].z; lo[6] = q[3].x;
    hi[0] = q[0].y; hi[1] = q[0].w; hi[2] = q[1].y; hi[3] = q[1].w; hi[4] = q[2].y; hi[5] = q[2].w; hi[6] = q[3].y;
    unpack_row7(lo[0], hi[0], ra); unpack_row7(lo[1], hi[1], rb);
    record_b_row<1, 4>(ra, rb, wTL, wTR, wBL, wBR, Bm);            // B[0]: columns 1..4 (dy of patch row 0)
    unpack_row7(lo[2], hi[2], ra);
    record_b_row<0, 6>(rb, ra, wTL, wTR, wBL, wBR, Bc);            // B[1]
  }
  template <int Y>
  PLSVO_HD void row(float4& vr, float4& vx, float4& vy) {
#pragma clang fp contract(off)
    float* const top = (Y & 1) ? rb : ra;                          // record row Y+2
    float* const bot = (Y & 1) ? ra : rb;                          // record row Y+3 (unpacked now)
    unpack_row7(lo[Y + 3], hi[Y + 3], bot);
    float Bp[6];
    if (Y < 3) record_b_row<0, 6>(top, bot, wTL, wTR, wBL, wBR, Bp); else record_b_row<1, 4>(top, bot, wTL, wTR, wBL, wBR, Bp);   // B[Y+2]
    float* pr = reinterpret_cast<float*>(&vr); float* px = reinterpret_cast<float*>(&vx); float* py = reinterpret_cast<float*>(&vy);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      pr[x] = Bc[x + 1];
      px[x] = 0.5f * (Bc[x + 2] - Bc[x]);
      py[x] = 0.5f * (Bp[x + 1] - Bm[x + 1]);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) { Bm[c] = Bc[c]; Bc[c] = (Y < 3 || (c >= 1 && c <= 4)) ? Bp[c] : 0.0f; }
  }
};
// all four rows at once (host test)
PLSVO_HD void ref_patch_from_record(const uint4* q, float4* vr, float4* vx, float4* vy) {
  RecordRows rr;
  rr.start(q);
  rr.row<0>(vr[0], vx[0], vy[0]); rr.row<1>(vr[1], vx[1], vy[1]); rr.row<2>(vr[2], vx[2], vy[2]); rr.row<3>(vr[3], vx[3], vy[3]);
}

}  // namespace plsvo_hip
