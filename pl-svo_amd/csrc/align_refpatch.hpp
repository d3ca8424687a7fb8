// align_refpatch.hpp -- the float arithmetic of a reference patch (interpolated intensity + central-difference gradient), shared by
// align_kernels.hip (device) and tests/test_refpatch_host.py (the same source compiled for the host with g++ -ffp-contract=off): the
// record form (PLSVO_BYTE_CACHE) must rebuild bit for bit what the direct form computes.
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

// computeInterpWeights from the stored fractions su = u - floor(u), sv = v - floor(v): the arithmetic of patch_weights
PLSVO_HD void interp_weights(float su, float sv, float& wTL, float& wTR, float& wBL, float& wBR) {
  wTL = (float)((1.0 - (double)su) * (1.0 - (double)sv));
  wTR = (float)((double)su * (1.0 - (double)sv));
  wBL = (float)((1.0 - (double)su) * (double)sv);
  wBR = (float)((double)su * (double)sv);
}
PLSVO_HD void unpack_row7(uint32_t lo, uint32_t hi, float* o7) {
  o7[0] = (float)(lo & 0xffu); o7[1] = (float)((lo >> 8) & 0xffu); o7[2] = (float)((lo >> 16) & 0xffu); o7[3] = (float)(lo >> 24);
  o7[4] = (float)(hi & 0xffu); o7[5] = (float)((hi >> 8) & 0xffu); o7[6] = (float)((hi >> 16) & 0xffu);
}
// This lane's two patch rows (2h, 2h+1) of interpolated reference intensity and central-difference gradient from record rows
// 2h .. 2h+4 (q01 = rows 2h, 2h+1; q23 = rows 2h+2, 2h+3; q4 = row 2h+4).  B_kl[c] = bilinear over record rows k, l at columns c, c+1;
// the precompute's expressions (:236-264, :348-375): ref = B(y+1, x+1), dx = 0.5 (B(y+1, x+2) - B(y+1, x)), dy = 0.5 (B(y+2, x+1) - B(y, x+1)).
PLSVO_HD void ref_rows_from_record(const uint4& q01, const uint4& q23, const uint2& q4, float su, float sv,
                                   float4& vr0, float4& vx0, float4& vy0, float4& vr1, float4& vx1, float4& vy1) {
#pragma clang fp contract(off)
  float wTL, wTR, wBL, wBR;
  interp_weights(su, sv, wTL, wTR, wBL, wBR);
  float R0[7], R1[7], R2[7], R3[7], R4[7];
  unpack_row7(q01.x, q01.y, R0); unpack_row7(q01.z, q01.w, R1); unpack_row7(q23.x, q23.y, R2); unpack_row7(q23.z, q23.w, R3); unpack_row7(q4.x, q4.y, R4);
  float B01[4], B12[6], B23[6], B34[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    B01[c] = bilinear(wTL, wTR, wBL, wBR, R0[c + 1], R0[c + 2], R1[c + 1], R1[c + 2]);
    B34[c] = bilinear(wTL, wTR, wBL, wBR, R3[c + 1], R3[c + 2], R4[c + 1], R4[c + 2]);
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    B12[c] = bilinear(wTL, wTR, wBL, wBR, R1[c], R1[c + 1], R2[c], R2[c + 1]);
    B23[c] = bilinear(wTL, wTR, wBL, wBR, R2[c], R2[c + 1], R3[c], R3[c + 1]);
  }
  float* r0 = reinterpret_cast<float*>(&vr0); float* x0 = reinterpret_cast<float*>(&vx0); float* y0 = reinterpret_cast<float*>(&vy0);
  float* r1 = reinterpret_cast<float*>(&vr1); float* x1 = reinterpret_cast<float*>(&vx1); float* y1 = reinterpret_cast<float*>(&vy1);
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    r0[x] = B12[x + 1];
    x0[x] = 0.5f * (B12[x + 2] - B12[x]);
    y0[x] = 0.5f * (B23[x + 1] - B01[x]);
    r1[x] = B23[x + 1];
    x1[x] = 0.5f * (B23[x + 2] - B23[x]);
    y1[x] = 0.5f * (B34[x] - B12[x + 1]);
  }
}

}  // namespace plsvo_hip
