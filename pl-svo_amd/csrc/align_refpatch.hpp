// align_refpatch.hpp -- the float arithmetic of a reference patch (interpolated intensity + central-difference gradient) of
// align_kernels.hip, every product and sum rounded on its own.
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

}  // namespace plsvo_hip
