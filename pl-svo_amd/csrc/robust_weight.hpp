// robust_weight.hpp -- the photometric robust weight of a point pixel, (float)(1.0 / (1.0 + (double)|res|))
// (src/sparse_img_align.cpp:479: `weight = 1.0 / (1.0 + fabs(res))` with float res, double arithmetic, float result).
// Shared by align_kernels.hip and tools/robust_weight_exhaustive.hip (every float |res| in [0, 256] against the compiler's IEEE division).
#pragma once
#include <hip/hip_runtime.h>

namespace plsvo_hip {

// The reference's value BIT FOR BIT, double rounding included: d = 1 + a is exact in double (a >= 2^-29; below that both sides round the
// same sum), and the quotient is formed by the very fma sequence the compiler emits for an IEEE double division -- reciprocal seed, two
// Newton steps, quotient, remainder, one correction -- without its v_div_scale / v_div_fmas / v_div_fixup wrapping, which only rescales
// operands whose exponents could overflow an intermediate (d is in [1, 257], the numerator is 1: the scale is 1 and the fix-up the
// identity).  10 vector instructions, as many as the float-only form below.
__device__ __forceinline__ float robust_weight_f64(float a) {
  const double d = 1.0 + (double)a;
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  const double q = fma(fma(-d, r, 1.0), r, r);   // quotient 1 * r, remainder 1 - d r, correction
  return (float)q;
}

// Float-only form (rounds 1-3 of this build; kept for tools/robust_weight_exhaustive.hip, NOT used by the kernel): 1 + a split exactly
// into s_hi + s_lo (two-sum), y0 = rcp(s_hi), exact residual by fma, one correction: the quotient to ~1e-14 relative before the final
// rounding.  NOT bit-exact: whenever the quotient lies within ~1e-14 of a float rounding boundary the result can be the neighbouring
// float -- 13 of the 1 132 462 081 floats in [0, 256] on the MI355X (13 ... 285 in a host enumeration with reciprocal seeds within
// +-1 ulp), and the reference's own double rounding (double quotient exactly on a float midpoint: 11 inputs, e.g. a = 0.0645160973 ->
// 31/33) cannot be reproduced by any single rounding.
__device__ __forceinline__ float robust_weight_f32(float a) {
  const float s_hi = __fadd_rn(1.0f, a);
  const float bv = __fsub_rn(s_hi, 1.0f);
  const float s_lo = __fadd_rn(__fsub_rn(1.0f, __fsub_rn(s_hi, bv)), __fsub_rn(a, bv));  // exact: (1 + a) - s_hi
  const float y0 = __builtin_amdgcn_rcpf(s_hi);
  const float e = __fmaf_rn(-s_hi, y0, 1.0f);
  const float c = __fmaf_rn(-s_lo, y0, e);
  return __fmaf_rn(y0, c, y0);
}

}  // namespace plsvo_hip
