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
//   * PATCH SLOTS.  The features of a level are flattened into patch slots (points: 1 patch, segments: N samples; a
//     patch is 4x4 pixels).  The slot layout is STATIC: the host computes it once per job and level (points own slots
//     [0, n_pts), the segments are packed behind them into the wave-rounds of 64 slots -- first-fit in decreasing N, a segment
//     with N <= 64 samples never straddles a multiple of 64), so the device needs no scan, the rounds of a pass are as few as the
//     patches allow (a round costs the same whether its lanes hold patches or holes), and all samples of a line sit in ONE wave-round;
//   * ONE PASS PER ITERATION, ONE LANE PER SLOT.  A lane owns a slot for the whole iteration: it warps and projects the 3-D
//     point, gathers the 5x5 window of the current image through L2 (five rows of aligned dword pairs + v_alignbyte; LDS holds
//     only the small slot tables, so several workgroups share a CU), rebuilds the reference patch's interpolated intensity and
//     gradient row by row from the slot's 64-byte record of image bytes (align_refpatch.hpp), evaluates the 16 pixels, exchanges
//     the line residuals through LDS with a WAVE-level fence (no workgroup barrier), and expands the 6x6 contribution of the
//     slot.  The only workgroup barriers of an iteration are the two around the 6x6 solve;
//   * FIVE SCALARS PER PATCH.  The 6-vector Jacobian of a pixel is J = fs * (dx * r0 + dy * r1) with r0, r1 the rows of
//     the 2x6 projection Jacobian of the PATCH, so sum_pix w J J^T = fs^2 (A r0 r0^T + B (r0 r1^T + r1 r0^T) + C r1 r1^T)
//     with A = sum w dx^2, B = sum w dx dy, C = sum w dy^2, and sum_pix w res J = fs (D r0 + E r1).  Only A..E are
//     accumulated per pixel;
//   * per-line re-weighting (H += H_line * w / r, Jres += Jres_line * w, cull if r >= 200 or a sample leaves the image,
//     src/sparse_img_align.cpp:640-688) needs the line's mean |residual| first: every sample lane sums its line's
//     sample residuals from LDS in fixed order.  A job with a line of more than 64 samples at some level runs that level
//     in two passes (residuals, workgroup barrier, expansion) -- same code, selected by a workgroup-uniform flag;
//   * REDUCTION: 27 + 3 lane-private doubles -> butterfly reduce-scatter inside each 16-lane DPP row (30 exchanges
//     instead of 4 x 30), rows and waves combined through LDS in fixed order: deterministic.
//   * TWO FAMILIES OF LAUNCH SHAPES (template parameter T = threads per frame).  THROUGHPUT (64 / 128: thousands of frames in flight, eight
//     workgroups per CU): the reference patch is a 64-byte record of image bytes rebuilt every iteration, pyramids read through their tiled
//     mirror (64), 255-256 registers, two waves per SIMD.  LATENCY (256 / 512: a frame owns most of a CU; round 5): the reference patch as
//     float rows, slot tables in LDS, a shorter solve, a trimmed level set-up, and -- with at most cu_count / 2 frames -- TWO WORKGROUPS PER
//     FRAME: rank 0 the point slots, rank 1 the segments' samples, 32 partial sums exchanged per iteration as tagged granules through L2
//     (plsvo_wave.hpp::pair_allgather32), the identical solver on both.
//
// Numerics: image interpolation and residuals in float with the reference's operation order and NO
// fma contraction (__fmul_rn/__fadd_rn); geometry and all accumulators in double.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "plsvo_dev.hpp"
#include "plsvo_math.hpp"
#include "plsvo_wave.hpp"
#include "align_refpatch.hpp"
#include "robust_weight.hpp"

namespace plsvo_hip {

// ------------------------------------------------------------------------------------------------
// image gather: pixels [x, x+7) of row y of a u8 pyramid level as floats, via aligned dword reads + v_alignbyte.
//   TILED   the tiled mirror (plsvo_dev.hpp: 16 x 8 pixel tiles of 128 B, `pitch` = tiles per row): what the one-wave-per-frame launch
//           shape reads -- it runs within a few per cent of the achievable HBM rate and a window touches 1.9 lines instead of 5 (-7 %
//           launch time at 32768 frames).  An aligned dword never straddles a tile row, so the over-read stays inside the level.
//   !TILED  the row-major slab (`pitch` = level width; the allocation is padded for the over-read): what a frame that has a CU to
//           itself reads -- it is bound by latency and issue, and there the tile arithmetic and the three separate dword requests
//           per row (instead of one 12-byte request) cost 12 % of a pass and 25-45 % of a level's set-up.
// ------------------------------------------------------------------------------------------------
// robust weight of a point pixel: robust_weight.hpp (the reference's (float)(1.0 / (1.0 + (double)|res|)) bit for bit: 0 mismatches over
// every float in [0, 256] on the MI355X; +0.9 % launch time against the float-only form of rounds 1-3, which missed 13 inputs)
__device__ __forceinline__ float robust_weight(float a) { return robust_weight_f64(a); }

// The per-slot cache of the reference patch (the reference's ref_patch_cache_ + jacobian_cache_, include/plsvo/sparse_img_align.h:83-96) is
// the 64-byte RECORD of align_refpatch.hpp: the 7x7 window of image bytes around the patch + the two sub-pixel fractions.  Every
// iteration the slot's lane rebuilds interpolated intensity and central-difference gradient from it, one patch row at a time, with the
// very operations of the reference's precompute (:236-264, :348-375) -- bit-identical values (tests/test_refpatch_host.py) -- for 64 B
// streamed per patch-iteration instead of the 192 B of three float arrays (rounds 1-3) and ~340 more float operations per slot.
// Measured on MI355X (round 4, one lane per slot): 15.14 -> 13.91 ms per 32768-frame launch.
// bytes [x, x+8) of row y of a u8 pyramid level: aligned dword reads + v_alignbyte
template <bool TILED>
__device__ __forceinline__ uint2 load_row8_raw(const uint8_t* img, int pitch, int x, int y) {
  uint32_t d0, d1, d2, sh;
  if constexpr (TILED) {
    const int a = x & ~3;
    sh = (uint32_t)(x & 3);
    const int row = tiled_row_offset(pitch, y);
    d0 = *reinterpret_cast<const uint32_t*>(img + (row + tiled_col_offset(a)));
    d1 = *reinterpret_cast<const uint32_t*>(img + (row + tiled_col_offset(a + 4)));
    d2 = *reinterpret_cast<const uint32_t*>(img + (row + tiled_col_offset(a + 8)));
  } else {
    const int off = y * pitch + x, a = off & ~3;
    sh = (uint32_t)(off & 3);
    d0 = *reinterpret_cast<const uint32_t*>(img + a);
    d1 = *reinterpret_cast<const uint32_t*>(img + a + 4);
    d2 = *reinterpret_cast<const uint32_t*>(img + a + 8);
  }
  return make_uint2(__builtin_amdgcn_alignbyte(d1, d0, sh), __builtin_amdgcn_alignbyte(d2, d1, sh));
}
// optional per-phase timing (compile with -DPLSVO_TIMING): thread 0 accumulates s_memtime deltas
#ifdef PLSVO_TIMING
#define TICK_RAW(slot) do { if (tid == 0) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); s_time[slot] += t__ - s_tlast; s_tlast = t__; } } while (0)
#if PLSVO_TIMING == 3
// the per-level set-up split into its steps (slots 1..5; the iterations all land in slot 6): -DPLSVO_TIMING=3, tools/gpu_phase_timing.py SETUP=1
#define TICK(slot) do { if ((slot) == 0) TICK_RAW(5); } while (0)
#define TICKS(slot) TICK_RAW(slot)
#else
#define TICK(slot) TICK_RAW(slot)
#define TICKS(slot) do { } while (0)
#endif
#else
#define TICK(slot) do { } while (0)
#define TICKS(slot) do { } while (0)
#endif

#define SLOT_HOLE ((int)0x80000000)   // s_meta[p].x of a slot no live feature owns at this level

// workgroup barrier; a one-wave workgroup needs only the wave-level form
template <int T>
__device__ __forceinline__ void block_sync() {
  if constexpr (T == 64) wave_lds_fence(); else __syncthreads();
}

// launch shapes up to this many threads per frame read the tiled pyramid mirror, larger ones the row-major slab.
// Measured: 64 threads (32768 frames, the chip saturated) -7 % launch time tiled; 128 threads (8192 frames) +4 % tiled; 512 +12 % per pass.
constexpr int kTiledMaxThreads = 64;
// half-width of the near-tie band, in units of sqrt(n_meas) * 2^-24 (one sigma of the reference's float sum is ~0.25 of that)
constexpr float kChiBand = 1.5f;

// LDS layout shared by the kernel and the host-side size helper: the tables end at this offset, the 4 KB window follows
__host__ __device__ inline size_t align_chi_window_offset(int threads, int cap, int scap) {
  size_t o = sizeof(double) * 32 * (threads / 16) + sizeof(double) * 64 + sizeof(int) * 32;
  o += (size_t)cap * (sizeof(int2) + sizeof(float)) + (size_t)scap * sizeof(int) + ((size_t)2 * scap + 2) * sizeof(float);
  return (o + 15) & ~(size_t)15;
}
// the latency shapes' slot tables follow the window / the planes
__host__ __device__ inline size_t align_quad_offset(int threads, int cap, int scap, int chi_lds_pts) {
  const size_t window = 1024 * sizeof(float), planes = (size_t)2 * chi_lds_pts * 16 * sizeof(float);
  return (align_chi_window_offset(threads, cap, scap) + (planes > window ? planes : window) + 15) & ~(size_t)15;
}
// everything the kernel's own tables take
__host__ __device__ inline size_t align_lds_used(int threads, int cap, int scap, int chi_lds_pts) {
  // latency shapes: every slot's 3-D point (24 B) and reference pixel position (8 B) -- global arrays in the throughput shapes
  const size_t quad = threads >= kQuadMinThreads ? (size_t)cap * 32 : 0;
  return align_quad_offset(threads, cap, scap, chi_lds_pts) + quad + 16;
}
// ------------------------------------------------------------------------------------------------
// The chi2 the solver compares (`new_chi2 > chi2_`, [ext] vk::NLLSSolver::optimizeGaussNewton) is
//     (float)(pt_chi2 + seg_chi2) / (float)n_meas_                                   src/sparse_img_align.cpp:171, 192
// with pt_chi2 the SEQUENTIAL float sum of res*res*weight over every pixel of every visible point in feature order (:484)
// and seg_chi2 the sequential float sum of res_*res_*weight over the surviving lines (:683).  The passes add the same float
// terms exactly (in double) and round once: ~1e-6 away from the reference's value -- its own summation noise -- and near
// convergence two successive chi2 values are closer than that in one iteration out of seven, so the accept / roll-back
// decision would be a coin the device and the reference toss separately (15 % of frames ended on a different path).
// When |new_chi2 - chi2_| is inside the band that noise can span, ONE wave re-adds the stored per-pixel terms of this and of
// the previous iteration in the reference's order, and the decision is taken on those two floats.
//
//   bufA / bufB   this job's slice of the chi_terms plane written at iteration `iter` / `iter - 1` (16 floats per point slot;
//                 invisible points and patches outside the current image hold +0)
//   s_lterm       the lines' terms of the two iterations as the passes computed them (plane = iteration parity); a line
//                 culled at iteration k (s_dead = k + 1) counts for the iterations before k only
//   s_out[0..1]   chi2 (before the division by n_meas_) of iteration `iter`, `iter - 1`
// 32 slots (2 KB per plane) at a time are staged through LDS with coalesced 16-byte loads, the next window's loads in flight
// under the current window's chains; lane 0 / lane 1 run the chain of plane A / B, the next slot's LDS reads in flight under
// the sixteen dependent additions of the current one.  The line terms are NOT re-derived: a line's float term differs from the
// reference's only through the summation order of its <= 512 |res| values (~1e-7 of a term, and the lines carry a few per cent
// of chi2) -- storing 64 B per line sample and iteration to remove that would cost more HBM traffic than everything else here.
// Must be called by one full wave.
// (explicit address spaces: the function is not inlined -- its registers must not add to the main pass's -- and without them every
//  LDS access would be a flat_load that also waits for the global prefetch in flight)
// ------------------------------------------------------------------------------------------------
#define PLSVO_LDS __attribute__((address_space(3)))
#define PLSVO_GLOBAL __attribute__((address_space(1)))
typedef float plsvo_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float chain4(float s, plsvo_v4f v) {   // four sequential float additions, never re-associated
  s = __fadd_rn(s, v.x); s = __fadd_rn(s, v.y); s = __fadd_rn(s, v.z); return __fadd_rn(s, v.w);
}
// One round of the SLOT-PARALLEL form of that sequential sum (tests/test_exact_sum_model.py::half_wave_sum is its CPU statement and the
// evidence that it is bit-exact): the 32 lanes of a half-wave hold one slot each (t[0..15], in summation order; lanes without a slot
// hold +0), s is the true running float sum in front of the round (uniform in the half), base the plain double sum of everything before.
// While the running sum stays inside one binade every addition rounds ITS TERM to the binade's quantum (s' = s + RN_ulp(t): the rounded
// terms add exactly, in any order) unless the term sits exactly half a quantum from a multiple (round-to-even looks at the running
// parity) or the sum crosses into the next binade.  So each lane, from the binade its double prefix predicts, rounds its sixteen terms
// and adds them (delta); an exclusive scan of delta gives every lane its candidate start value; the first lane whose prediction does
// not hold (wrong binade, crossing, half quantum: ~6 % of the slots) re-adds its own sixteen terms one after the other from the exact
// value in front of it, and the lanes behind it are checked again from there.  ~200 dependent additions are left of 3300.
// (Rounds 3 walked the terms on ONE lane: 8.8 us per near tie for a lone frame; measured on MI355X: -2 ... -3 % per small-batch step.)
// 32-lane inclusive prefix sum inside each half-wave without touching LDS: four row_shr steps inside the 16-lane DPP rows, then the
// last lane of rows 0 / 2 is added to rows 1 / 3 (row_bcast15, row mask 0xA).  Lanes a shift has no source for receive 0.
__device__ __forceinline__ double half_scan_incl(double v) {
  v += dpp_bcast_f64<DPP_ROW_SHR(1), 0xF>(v);
  v += dpp_bcast_f64<DPP_ROW_SHR(2), 0xF>(v);
  v += dpp_bcast_f64<DPP_ROW_SHR(4), 0xF>(v);
  v += dpp_bcast_f64<DPP_ROW_SHR(8), 0xF>(v);
  v += dpp_bcast_f64<DPP_ROW_BCAST15, 0xA>(v);
  return v;
}
// value of lane `la` (half 0) / `32 + lb` (half 1), each half-wave its own; la, lb wave-uniform
__device__ __forceinline__ double half_readlane(double v, int la, int lb) {
  const double a = readlane_f64(v, la), b = readlane_f64(v, 32 + lb);
  return (threadIdx.x & 32) ? b : a;
}
__device__ __forceinline__ float half_readlane(float v, int la, int lb) {
  const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), la)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32 + lb));
  return (threadIdx.x & 32) ? b : a;
}

__device__ __forceinline__ void exact_round32(const float (&t)[16], float& s, double& base) {
  const int lane = threadIdx.x & 63, l32 = lane & 31;
  double d = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) d += (double)t[i];
  const double inc = half_scan_incl(d);
  const double excl = inc - d;                               // a predictor only: its last bit does not matter
  const double tot = half_readlane(inc, 31, 31);
  const double lo = base + excl, hi = lo + d;
  int e = (int)((__double2hiint(lo) >> 20) & 0x7ff) - 1023;
  bool safe = lo > 0.0 && hi < 1.7976931348623157e308 && e - 24 >= -126 && e <= 126;
  e = min(max(e, -100), 126);
  const double two_e = __hiloint2double((e + 1023) << 20, 0), two_e1 = __hiloint2double((e + 1024) << 20, 0);
  safe = safe && lo * (1.0 - 0.0009765625) >= two_e && hi * (1.0 + 0.0009765625) < two_e1;
  // the scan of `delta` below must be exact in double: every lane's rounded terms are multiples of ITS quantum 2^(e-23), the round's sum
  // stays below 2^(e_top+1) with e_top the binade the round ends in -- so a lane more than 26 binades below that walks its terms itself
  const int e_top = (int)((__double2hiint(base + tot) >> 20) & 0x7ff) - 1023;
  safe = safe && e >= e_top - 26;
  const float C = __int_as_float((e + 127) << 23), half_q = __int_as_float((e - 24 + 127) << 23);
  double delta = 0.0;
  bool half_hit = false;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float r = __fsub_rn(__fadd_rn(C, t[i]), C);
    half_hit = half_hit || fabsf(__fsub_rn(t[i], r)) == half_q;
    delta += (double)r;
  }
  safe = safe && !half_hit;
  if (!safe) delta = 0.0;
  const double dinc = half_scan_incl(delta);
  const double dex = dinc - delta;                           // exact: every value is a multiple of the smallest quantum of the round, < 2^53 of them
  const double dlast = half_readlane(dinc, 31, 31);
  int start = 0;
  double s0 = (double)s, off = 0.0;
  bool done = false;
  for (;;) {
    const double sl = s0 + (dex - off);
    bool ok = safe && sl > 0.0 && two_e <= sl && sl + delta < two_e1;
    if (l32 < start || done) ok = true;
    const unsigned long long m = __ballot(!ok);              // scalar: the first failing lane of each half is a scalar too
    const unsigned ba = (unsigned)m, bb = (unsigned)(m >> 32);
    const int fa = ba ? __builtin_ctz(ba) : 32, fb = bb ? __builtin_ctz(bb) : 32;
    const int f = (lane & 32) ? fb : fa;
    const double dex_f = half_readlane(dex, fa & 31, fb & 31), delta_f = half_readlane(delta, fa & 31, fb & 31);
    float sf = (float)(s0 + (dex_f - off));
    if (!done && l32 == f) {
#pragma unroll
      for (int i = 0; i < 16; ++i) sf = __fadd_rn(sf, t[i]);
    }
    sf = half_readlane(sf, fa & 31, fb & 31);
    if (!done) {
      if (f == 32) { s0 = s0 + (dlast - off); done = true; }
      else { s0 = (double)sf; start = f + 1; off = dex_f + delta_f; if (start == 32) done = true; }
    }
    if (!__any(!done)) break;
  }
  s = (float)s0;
  base += tot;
}

__device__ __noinline__ void exact_chi2_pair(const PLSVO_GLOBAL float* bufA, const PLSVO_GLOBAL float* bufB, int n_pts, int n_seg, int iter,
                                             const PLSVO_LDS int* s_dead, PLSVO_LDS float* s_win, const PLSVO_LDS float* s_lterm, int scap,
                                             PLSVO_LDS float* s_out) {
  const int lane = threadIdx.x & 63;
  const plsvo_v4f z4 = { 0.f, 0.f, 0.f, 0.f };
  auto load_win = [&](int r, plsvo_v4f* v) {   // float4 `lane` and `lane + 64` of the round's 128-float4 window, both planes
    const size_t f0 = (size_t)r * 512;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = lane + 64 * h;
      const bool ok = 32 * r + (q >> 2) < n_pts;
      v[h] = ok ? reinterpret_cast<const PLSVO_GLOBAL plsvo_v4f*>(bufA + f0)[q] : z4;
      v[2 + h] = ok ? reinterpret_cast<const PLSVO_GLOBAL plsvo_v4f*>(bufB + f0)[q] : z4;
    }
  };
  float sum = 0.0f;   // lanes 0..31: plane A, lanes 32..63: plane B
  double base = 0.0;
  const int n_rounds = (n_pts + 31) >> 5;
  plsvo_v4f nxt[4] = { z4, z4, z4, z4 };
  if (n_rounds > 0) load_win(0, nxt);
  for (int r = 0; r < n_rounds; ++r) {
    const plsvo_v4f c0 = nxt[0], c1 = nxt[1], c2 = nxt[2], c3 = nxt[3];
    if (r + 1 < n_rounds) load_win(r + 1, nxt);
    wave_lds_fence();   // the previous window has been consumed
    PLSVO_LDS plsvo_v4f* const w4 = reinterpret_cast<PLSVO_LDS plsvo_v4f*>(s_win);
    w4[lane] = c0; w4[lane + 64] = c1; w4[128 + lane] = c2; w4[128 + lane + 64] = c3;
    wave_lds_fence();
    {   // lanes 0..31: the window's slots of plane A, lanes 32..63: of plane B (the window holds +0 beyond the last point)
      const PLSVO_LDS plsvo_v4f* w = w4 + 128 * (lane >> 5) + 4 * (lane & 31);
      const plsvo_v4f q0 = w[0], q1 = w[1], q2 = w[2], q3 = w[3];
      const float t[16] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w };
      exact_round32(t, sum, base);
    }
  }
  if ((lane & 31) == 0) {
    const int h = lane >> 5;                                             // 0: this iteration (plane A), 1: the previous one (plane B)
    float seg_sum = 0.0f;                                                // :683, segments in feature order
    for (int sg = 0; sg < n_seg; ++sg) {
      const int dead = s_dead[sg];
      const float t = s_lterm[((iter - h) & 1) * scap + sg];
      // the line took part in iteration j (= iter - h) unless it was culled at an iteration <= j
      seg_sum = __fadd_rn(seg_sum, (dead == 0 || dead > iter - h + 1) ? t : 0.0f);
    }
    s_out[h] = __fadd_rn(sum, seg_sum);                                  // :171  chi2 = pt_chi2 + seg_chi2
  }
  wave_lds_fence();
}

// the same with the two planes in LDS (small batches): no staging, lane 0 / lane 1 walk plane A / B directly
__device__ __noinline__ void exact_chi2_pair_lds(const PLSVO_LDS float* planeA, const PLSVO_LDS float* planeB, int n_pts, int n_seg, int iter,
                                                 const PLSVO_LDS int* s_dead, const PLSVO_LDS float* s_lterm, int scap, PLSVO_LDS float* s_out) {
  const int lane = threadIdx.x & 63, h = lane >> 5, l32 = lane & 31;
  const PLSVO_LDS plsvo_v4f* plane = reinterpret_cast<const PLSVO_LDS plsvo_v4f*>(h ? planeB : planeA);
  float sum = 0.0f;
  double base = 0.0;
  for (int p0 = 0; p0 < n_pts; p0 += 32) {
    const int p = p0 + l32;
    const plsvo_v4f z4 = { 0.f, 0.f, 0.f, 0.f };
    plsvo_v4f q0 = z4, q1 = z4, q2 = z4, q3 = z4;
    if (p < n_pts) { q0 = plane[4 * p]; q1 = plane[4 * p + 1]; q2 = plane[4 * p + 2]; q3 = plane[4 * p + 3]; }
    const float t[16] = { q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w };
    exact_round32(t, sum, base);
  }
  if (l32 == 0) {
    float seg_sum = 0.0f;
    for (int sg = 0; sg < n_seg; ++sg) {
      const int dead = s_dead[sg];
      const float t = s_lterm[((iter - h) & 1) * scap + sg];
      seg_sum = __fadd_rn(seg_sum, (dead == 0 || dead > iter - h + 1) ? t : 0.0f);
    }
    s_out[h] = __fadd_rn(sum, seg_sum);
  }
  wave_lds_fence();
}

// ------------------------------------------------------------------------------------------------
// SparseImgAlign::run for every job of the batch: levels [level_hi .. level_lo] of each job's range
// ------------------------------------------------------------------------------------------------
constexpr int kMinWavesPerSimd = 2;   // VGPR budget of 256: capping at 168 / 128 (3 / 4 waves per SIMD) spills
template <int T>
__global__ __launch_bounds__(T, kMinWavesPerSimd) void align_fused_kernel(AlignBatchDev b, int cap, int scap, int level_hi, int level_lo, int do_init) {
  // longest-processing-time-first: the hardware hands out workgroups in blockIdx order, so the jobs with the most patches
  // start first and the launch tail is made of the cheapest frames
  // Two workgroups per frame (b.pair, latency shapes only): blocks q and q + 8 of every group of 16 share frame (group, q) -- the hardware
  // deals consecutive blocks to the eight XCDs in turn, so the pair shares an L2 (a speed bonus, never a correctness condition).
  const bool pair = T >= kQuadMinThreads && b.pair != 0;
  const int rank = pair ? (int)((blockIdx.x >> 3) & 1u) : 0;
  const int wg_index = pair ? (int)((blockIdx.x >> 4) * 8u + (blockIdx.x & 7u)) : (int)blockIdx.x;
  if (wg_index >= b.n_jobs) return;
  const bool lead = rank == 0;                    // the workgroup that publishes the frame's state, pose and trace
  const bool own_pts = !pair || rank == 0, own_segs = !pair || rank == 1;
  const int job_id = b.order ? b.order[wg_index] : wg_index;
  const AlignJobDev job = b.jobs[job_id];
  AlignStateDev* st = b.state + job_id;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = tid >> 2, row = tid & 3;     // reference-patch precompute: four lanes per slot, one patch row each
  constexpr int ROWS = T / 16;                 // DPP rows of the workgroup (row partials of the reduction)

  extern __shared__ __align__(16) unsigned char smem[];
  double* s_red = reinterpret_cast<double*>(smem);                       // ROWS * 32: row partials
  double* s_pose = s_red + ROWS * 32;                                    // 0..8 R, 9..11 t, 12..18 model, 19..25 old model, 26 chi2_, 27 #evals, 28/29/31 work counters, 30 n_meas_ of the previous iteration
  double* s_tot = s_pose + 32;                                           // block totals of the last iteration: 21 H, 6 Jres, chi2, n_meas, evals
  int* s_ctl = reinterpret_cast<int*>(s_tot + 32);                       // 0 break, 1 stop, 2 iterations done, 3 error, 5 #patches of the level
  int2* s_meta = reinterpret_cast<int2*>(s_ctl + 32);                    // cap: x = feature (>= 0 point, < 0 segment -1-x, SLOT_HOLE), y = first slot | N << 20
  float* s_abs = reinterpret_cast<float*>(s_meta + cap);                 // cap: sum |res| of the slot's 16 pixels, -1 = sample not in the image
  int* s_dead = reinterpret_cast<int*>(s_abs + cap);                     // scap: per segment, 0 = alive, k + 1 = culled at iteration k of this level
  float* s_lterm = reinterpret_cast<float*>(s_dead + scap);              // 2 * scap + 2: exact chi2 term of every line, two iterations; the two sums
  float* s_win = reinterpret_cast<float*>(smem + align_chi_window_offset(T, cap, scap));   // 1024: two 32-slot windows of chi_terms, or the two planes themselves (chi_lds_pts)
  constexpr bool kQuad = T >= kQuadMinThreads;
  double* const s_xyz = reinterpret_cast<double*>(kQuad ? smem + align_quad_offset(T, cap, scap, b.chi_lds_pts) : smem);   // latency shapes: cap x 3, every slot's 3-D point (ref frame)
  float* const s_uvr = reinterpret_cast<float*>(s_xyz + 3 * (kQuad ? cap : 0));                         //   cap x 2: every slot's reference pixel position at the level

#ifdef PLSVO_TIMING
  __shared__ unsigned long long s_time[8];
  __shared__ unsigned long long s_tlast;
  if (tid == 0) { for (int k = 0; k < 8; ++k) s_time[k] = 0; s_tlast = __builtin_amdgcn_s_memtime(); }
#endif
  const int lv_first = min(job.max_level, level_hi), lv_last = max(job.min_level, level_lo);
  const bool nothing = job.skip || lv_first < lv_last;
  if (do_init && own_segs) {
    // solver reset() ([ext] vk::NLLSSolver::reset) and the working copy of the segment flags
    for (int s = tid; s < job.n_seg; s += T)
      b.seg_alive[job.seg_off + s] = b.seg_alive_in ? (b.seg_alive_in[job.seg_off + s] != 0) : 1;
  }
  if (tid == 0) {
    // (the seven pose values are requested together and go to LDS from registers: written to the state and read back one by one, each
    //  of the fourteen accesses waited for the one before -- a lone frame pays a microsecond per dependent first touch)
    double T_in[7], chi_in = 1e10; int stop_in = 0;
    if (do_init) {
#pragma unroll
      for (int k = 0; k < 7; ++k) T_in[k] = b.T0[7 * job_id + k];
      if (lead) {
#pragma unroll
      for (int k = 0; k < 7; ++k) st->T[k] = T_in[k];
      st->chi2 = 1e10; st->n_meas = 0; st->stop = 0; st->log_count = 0; st->error = 0;
      for (int k = 0; k < 36; ++k) st->H[k] = 0.0;
      for (int k = 0; k < PLSVO_MAX_LEVELS; ++k) st->iters[k] = 0;
      st->patch_levels = 0; st->patch_iters = 0; st->patch_iters_pt = 0; st->chi2_ties = 0; st->chi2_unarmed = 0;
      for (int k = 0; k < 8; ++k) st->phase_ticks[k] = 0;
      if (nothing && b.poses) for (int k = 0; k < 7; ++k) b.poses[7 * job_id + k] = T_in[k];
      if (nothing && b.work_key) b.work_key[job_id] = 0;   // a job without work sorts last in the next launch's order
      }
    } else if (!nothing) {   // per-level debug launches: the state crosses launches in HBM
#pragma unroll
      for (int k = 0; k < 7; ++k) T_in[k] = st->T[k];
      chi_in = st->chi2; stop_in = st->stop;
    }
    if (!nothing) {
#pragma unroll
      for (int k = 0; k < 7; ++k) { s_pose[12 + k] = T_in[k]; s_pose[19 + k] = T_in[k]; }
      s_pose[26] = chi_in; s_pose[28] = 0.0; s_pose[29] = 0.0; s_pose[31] = 0.0;
      for (int k = 0; k < 32; ++k) s_tot[k] = 0.0;
      s_ctl[1] = stop_in; s_ctl[3] = 0; s_ctl[6] = 0; s_ctl[9] = 0;
      s_ctl[10] = 0;
    }
  }
  if (nothing) return;
  block_sync<T>();   // seg_alive / state of this job initialised (same workgroup: visible after the barrier)
  if (b.chi_lds_pts > 0) {   // LDS planes: the slots between the last point and the next multiple of 4 are read by the exact sums: +0
    const int tail0 = job.n_pts * 16, tail1 = ((job.n_pts + 3) & ~3) * 16;
    for (int k = tail0 + tid; k < tail1; k += T) { s_win[k] = 0.0f; s_win[b.chi_lds_pts * 16 + k] = 0.0f; }
  }
  const size_t pbase = (size_t)job.patch_off;
  const int nfeat = job.n_pts + job.n_seg;
  double* const pxyz = b.patch_xyz + 3 * pbase;      // 3-D point of every slot (ref frame)
  // two workgroups per frame: rank 0 works on the slots [0, line0) -- the points --, rank 1 on [line0, n_slots) -- the segments' samples
  const int line0 = pair ? ((job.n_pts + 63) & ~63) : 0;
  unsigned long long* const xb = pair ? b.xbuf + (size_t)job_id * 256 : nullptr;
  unsigned xseq = b.xseq0;                            // (advanced identically by both workgroups: one per exchange)
  bool x_lost = false;                                // wave 0: an exchange gave up waiting for the partner (pair_allgather32's bounded poll): the frame ends with error 2

  for (int level = lv_first; level >= lv_last; --level) {
    // (level geometry is recomputed instead of indexing the kernel-argument arrays with a run-time level,
    //  which would force the whole argument struct into scratch memory)
    const int W = job.width >> level, Hh = job.height >> level;
    // throughput shapes read the tiled mirror of the pyramids, latency shapes the row-major slab
    constexpr bool kTiled = T <= kTiledMaxThreads;
    const unsigned int lvl_off = kTiled ? pyr_tiled_level_offset(job.width, job.height, level) : pyr_level_offset(job.width, job.height, level);
    const uint8_t* const pyr_base = kTiled ? b.pyr.tbase : b.pyr.base;
    const unsigned long long pyr_slot = kTiled ? b.pyr.tslot_bytes : b.pyr.slot_bytes;
    const uint8_t* ref_img = pyr_base + (size_t)job.ref_slot * pyr_slot + lvl_off;
    const uint8_t* cur_img = pyr_base + (size_t)job.cur_slot * pyr_slot + lvl_off;
    const int pitch = kTiled ? (W + 15) >> 4 : W;   // tiles per row / bytes per row
    int n_slots = 0; bool long_lines = false;
#pragma unroll
    for (int l = 0; l < PLSVO_MAX_LEVELS; ++l) if (l == level) { n_slots = job.n_slots[l]; long_lines = ((job.long_mask >> l) & 1) != 0; }
    if (n_slots > cap || n_slots > job.patch_cap) {  // host layout inconsistent with the launch: flag and bail out (uniform)
      if (tid == 0 && lead) { st->error = 1; if (b.work_key) b.work_key[job_id] = 0; }
      return;
    }
    const int slot_lo = (pair && rank == 1) ? min(line0, n_slots) : 0, slot_hi = (pair && rank == 0) ? min(line0, n_slots) : n_slots;
    block_sync<T>();  // previous level done with every LDS table
    TICKS(0);

    if (tid == 0) { s_ctl[0] = 0; s_ctl[2] = 0; s_ctl[5] = 0; s_ctl[7] = 0; s_ctl[8] = 0; s_pose[27] = 0.0; }
    for (int p = tid; p < n_slots; p += T) s_meta[p] = make_int2(SLOT_HOLE, 0);
    block_sync<T>();
    TICKS(1);

    // ---- slot table: every feature fills the slots the host layout gives it ----
    const double scale = 1.0 / (double)(1 << level);  // the reference's float scale is a power of two: exact
    int my_patches = 0;
    for (int f = (own_pts ? 0 : job.n_pts) + tid; f < (own_segs ? nfeat : job.n_pts); f += T) {
      if (f < job.n_pts) {
        // precomputeGaussNewtonParamsPoints :216-219: floor of the float position, 3 px border
        const int i = job.pt_off + f;
        if constexpr (kQuad) {   // (latency shapes: position and 3-D point are ONE round trip -- the point's table entries are written whether or not its slot is used)
          s_xyz[3 * f] = b.pt_xyz[3 * i]; s_xyz[3 * f + 1] = b.pt_xyz[3 * i + 1]; s_xyz[3 * f + 2] = b.pt_xyz[3 * i + 2];
        }
        const float u = (float)(b.pt_px[2 * i] * scale), v = (float)(b.pt_px[2 * i + 1] * scale);
        if constexpr (kQuad) { s_uvr[2 * f] = u; s_uvr[2 * f + 1] = v; }
        if (u >= 3.0f && v >= 3.0f && u < (float)(W - 3) && v < (float)(Hh - 3)) {
          s_meta[f] = make_int2(f, f | (1 << 20));
          if constexpr (kQuad) {
          } else {
          b.patch_uvref[2 * (pbase + f)] = u;
          b.patch_uvref[2 * (pbase + f) + 1] = v;
          pxyz[3 * f] = b.pt_xyz[3 * i];
          pxyz[3 * f + 1] = b.pt_xyz[3 * i + 1];
          pxyz[3 * f + 2] = b.pt_xyz[3 * i + 2];
          }
          ++my_patches;
        }
      } else {
        const int sl = f - job.n_pts, s = job.seg_off + sl;
        s_dead[sl] = 0; s_lterm[sl] = 0.0f; s_lterm[scap + sl] = 0.0f;
        // host layout: first slot | N << 20, or -1 when the segment has no landmark on entry or fails
        // precomputeGaussNewtonParamsSegments :299-301 ((px*scale).cast<int>() against cam->isInFrame(.,3,level));
        // N = 1 + (N0-1)/2^level samples (:320, LineFeat::setupSampling src/feature.cpp:160-173)
        // (latency shapes: the segment's ten doubles are requested together with its slot code and its flag, not after them)
        double g_s[2] = { 0.0, 0.0 }, g_e[2] = { 0.0, 0.0 }, g_p[3] = { 0.0, 0.0, 0.0 }, g_q[3] = { 0.0, 0.0, 0.0 };
        if constexpr (kQuad) {
          g_s[0] = b.seg_spx[2 * s]; g_s[1] = b.seg_spx[2 * s + 1]; g_e[0] = b.seg_epx[2 * s]; g_e[1] = b.seg_epx[2 * s + 1];
#pragma unroll
          for (int c = 0; c < 3; ++c) { g_p[c] = b.seg_p[3 * s + c]; g_q[c] = b.seg_q[3 * s + c]; }
        }
        const int code = b.seg_slot[(size_t)(level - b.slot_level0) * b.slot_stride + s];
        if (code >= 0 && b.seg_alive[s]) {
          const int p0 = code & 0xfffff, N = code >> 20;
          // :316-332: 2-D step on the level image, 3-D step between the end points, both accumulated
          const double sx = kQuad ? g_s[0] : b.seg_spx[2 * s], sy = kQuad ? g_s[1] : b.seg_spx[2 * s + 1];
          double inc2x = ((kQuad ? g_e[0] : b.seg_epx[2 * s]) - sx) * scale / (double)(N - 1);
          double inc2y = ((kQuad ? g_e[1] : b.seg_epx[2 * s + 1]) - sy) * scale / (double)(N - 1);
          double px = sx * scale, py = sy * scale;
          double xr[3], inc3[3];
          for (int c = 0; c < 3; ++c) {
            const double pr = kQuad ? g_p[c] : b.seg_p[3 * s + c];
            inc3[c] = ((kQuad ? g_q[c] : b.seg_q[3 * s + c]) - pr) / (double)(N - 1);
            xr[c] = pr;
          }
          for (int n = 0; n < N; ++n) {
            const int p = p0 + n;
            s_meta[p] = make_int2(-1 - sl, p0 | (N << 20));
            if constexpr (kQuad) {
              s_uvr[2 * p] = (float)px; s_uvr[2 * p + 1] = (float)py;
              s_xyz[3 * p] = xr[0]; s_xyz[3 * p + 1] = xr[1]; s_xyz[3 * p + 2] = xr[2];
            } else {
            b.patch_uvref[2 * (pbase + p)] = (float)px;
            b.patch_uvref[2 * (pbase + p) + 1] = (float)py;
            pxyz[3 * p] = xr[0];
            pxyz[3 * p + 1] = xr[1];
            pxyz[3 * p + 2] = xr[2];
            }
            px += inc2x; py += inc2y;
            xr[0] += inc3[0]; xr[1] += inc3[1]; xr[2] += inc3[2];
          }
          my_patches += N;
        }
      }
    }
    // integer count: order-independent.  (One atomic per WAVE: sixty-four lanes adding to the same LDS word serialise -- measured 4.6 k
    //  cycles per level for a lone frame, a fifth of its set-up.)
    if constexpr (kQuad) {
      const int wave_patches = wave_sum_i32_to_lane63(my_patches);
      if (lane == 63 && wave_patches) atomicAdd(&s_ctl[5], wave_patches);
    } else {
    if (my_patches) atomicAdd(&s_ctl[5], my_patches);
    }
    TICKS(2);
    block_sync<T>();  // patch_uvref / patch_xyz (global) and s_meta (LDS) visible to the workgroup
    TICKS(3);

    // ---- reference patches (:236-264, :348-375) ----
    if constexpr (kQuad) {
      // latency shapes: interpolated intensity and central-difference gradient as FLOAT rows, {ref[4], dx[4], dy[4]} per patch row (192 B per
      // slot), which the slot's lane reads back every iteration.  ONE LANE PER SLOT: it gathers the 7x7 window of reference-image bytes
      // (seven rows of aligned dwords + v_alignbyte) into the 64-byte record form in registers and lets align_refpatch.hpp::RecordRows produce
      // the four rows -- the operations the throughput shapes repeat every iteration, bit-identical to the reference's precompute
      // (tests/test_refpatch_host.py) -- sharing every interpolated value between the rows that need it: 340 float instructions per slot
      // where four lanes per slot (a row each, the first form of this block) issued 4 x 230; measured 5-8 k -> ... cycles per level.
      for (int pb = slot_lo + wave * 64; pb < slot_hi; pb += T) {
        const int p = pb + lane;
        if (p < slot_hi && s_meta[p].x != SLOT_HOLE) {
          const float u = s_uvr[2 * p], v = s_uvr[2 * p + 1];
          const PatchW pw = patch_weights(u, v);
          uint32_t d[7][3]; int shf[7];
#pragma unroll
          for (int k = 0; k < 7; ++k) {   // image rows vi-3 .. vi+3, columns ui-3 .. ui+3 (+1 byte of padding per row)
            const int off = (pw.vi - 3 + k) * pitch + (pw.ui - 3), a = off & ~3;
            shf[k] = off & 3;
            d[k][0] = *reinterpret_cast<const uint32_t*>(ref_img + a);
            d[k][1] = *reinterpret_cast<const uint32_t*>(ref_img + a + 4);
            d[k][2] = *reinterpret_cast<const uint32_t*>(ref_img + a + 8);
          }
          uint32_t lo[7], hi[7];
#pragma unroll
          for (int k = 0; k < 7; ++k) {
            lo[k] = __builtin_amdgcn_alignbyte(d[k][1], d[k][0], (uint32_t)shf[k]);
            hi[k] = __builtin_amdgcn_alignbyte(d[k][2], d[k][1], (uint32_t)shf[k]);
          }
          // the record's layout: q[k] = record rows 2k, 2k+1 (8 bytes each); q[3].zw = the two sub-pixel fractions
          uint4 q[4];
          q[0] = make_uint4(lo[0], hi[0], lo[1], hi[1]); q[1] = make_uint4(lo[2], hi[2], lo[3], hi[3]); q[2] = make_uint4(lo[4], hi[4], lo[5], hi[5]);
          q[3] = make_uint4(lo[6], hi[6], __float_as_uint(u - floorf(u)), __float_as_uint(v - floorf(v)));
          float4* const dst = reinterpret_cast<float4*>(b.cache_ref) + (pbase + p) * 12;
          RecordRows rec;
          rec.start(q);
          float4 vr, vx, vy;
          rec.template row<0>(vr, vx, vy); dst[0] = vr; dst[1] = vx; dst[2] = vy;
          rec.template row<1>(vr, vx, vy); dst[3] = vr; dst[4] = vx; dst[5] = vy;
          rec.template row<2>(vr, vx, vy); dst[6] = vr; dst[7] = vx; dst[8] = vy;
          rec.template row<3>(vr, vx, vy); dst[9] = vr; dst[10] = vx; dst[11] = vy;
        }
      }
    } else {
    // throughput shapes: the byte record the iterations rebuild intensity and gradient from
    for (int pb = 0; pb < n_slots; pb += T / 4) {
      const int p = pb + grp;
      if (p < n_slots && s_meta[p].x != SLOT_HOLE) {
        const float u = b.patch_uvref[2 * (pbase + p)], v = b.patch_uvref[2 * (pbase + p) + 1];
        const PatchW pw = patch_weights(u, v);
        // lane `row` of the slot's four writes record rows `row` and `row + 4` (image rows vi-3+.., columns ui-3 .. ui+3); lane 3 the fractions
        unsigned char* const rec = reinterpret_cast<unsigned char*>(b.cache_ref) + ((pbase + p) << 6);
        const int cx = pw.ui - 3, ry = pw.vi - 3 + row;
        *reinterpret_cast<uint2*>(rec + 8 * row) = load_row8_raw<kTiled>(ref_img, pitch, cx, ry);
        if (row < 3) *reinterpret_cast<uint2*>(rec + 8 * (row + 4)) = load_row8_raw<kTiled>(ref_img, pitch, cx, ry + 4);
        else *reinterpret_cast<float2*>(rec + 56) = make_float2(u - floorf(u), v - floorf(v));
      }
    }
    }
    TICKS(4);
    if (tid == 0) { SE3d m = se3_load(s_pose + 12); quat_to_matrix(m.q, s_pose); s_pose[9] = m.t[0]; s_pose[10] = m.t[1]; s_pose[11] = m.t[2]; }
    block_sync<T>();  // slot tables, pose state and cache complete
    TICK(0);

    // ---- Gauss-Newton iterations ([ext] NLLSSolver::optimizeGaussNewton) ----
    const double fs = fabs(job.fx) / (double)(1 << level);  // focal_length / (1<<level)  :262
    const float colmax = (float)(W - 2), rowmax = (float)(Hh - 2);

    for (int iter = 0; iter < job.n_iter; ++iter) {
     // A near tie whose per-pixel terms were NOT kept (HBM planes are written only while the solver is armed: 471 of 69 631 near ties
     // of a 32768-frame step; over 1000 seeds 4 of 19 such ties went the other way, among them the sweep's worst error, 1.6e-3 of the
     // inter-frame translation) re-runs the pixel arithmetic of the point patches for the missing iteration(s) -- stage 1: this
     // iteration's pose, stage 2: the previous one's (old_model_) -- writes the planes, and only then decides.  Nothing changes on the
     // path every other iteration takes (measured on MI355X: +0.9 % launch time, both tie cases follow the oracle).
     int redo_mask = 0;   // bit 0 = this iteration's terms are missing, bit 1 = the previous iteration's (workgroup-uniform)
     for (int stage = 0; stage < 3; ++stage) {   // 0: the iteration proper; 1, 2: terms-only re-runs of a near tie (rare)
      if (stage > 0 && !(redo_mask & stage)) continue;
      const bool terms_only = stage > 0;
      const int last_stage = (redo_mask & 2) ? 2 : ((redo_mask & 1) ? 1 : 0);
      // the pose (stage 2: the previous iteration's, whose rotation matrix thread 0 left in s_red -- free between the reduction and
      // the next pass)
      const double* const pose_rt = (stage == 2) ? s_red : s_pose;

      // HBM planes are written only while the solver is ARMED: the step that led to this iteration was small (||x||_inf < 1e-3), which is
      // when two successive chi2 values can come within the rounding noise of the reference's sums (99 % of the near ties of 60
      // config-2 frames had both iterations armed; 64 % of all iterations are).  LDS planes (small batches) are always written.
      const bool store_chi = terms_only || b.chi_lds_pts > 0 || s_ctl[7] != 0;
      const int chi_par = (stage == 2) ? ((iter & 1) ^ 1) : (iter & 1);
      float* const chi_it = b.chi_terms + (size_t)chi_par * b.chi_plane + (size_t)job.pt_off * 16;   // this iteration's plane of the points' chi2 terms (stage 2: the previous iteration's)

      double acc[32];   // 0..20 H (upper, row-major), 21..26 Jres, 27 chi2, 28 n_meas, 29 evals, 30 evals of point patches whose chi2 terms went to HBM, 31 unused
#pragma unroll
      for (int k = 0; k < 32; ++k) acc[k] = 0.0;

      // pass 0: residual sums only (two-pass levels); pass 1: the (fused) pass that accumulates
      for (int pass = (long_lines && !terms_only) ? 0 : 1; pass < 2; ++pass) {
        const bool write_abs = (!long_lines || pass == 0) && !terms_only;
        const bool accumulate = pass == 1 && !terms_only;
        const int n_rounds_slots = terms_only ? min(n_slots, job.n_pts) : (kQuad ? slot_hi : n_slots);   // a terms-only re-run visits the point slots only

        if constexpr (kQuad) {
        // ---- the pieces of the latency shape's pass ----
        // the sums over a patch's pixels, in the reference's pixel order: A = sum w dx^2, B = sum w dx dy, C = sum w dy^2, D = sum w res dx,
        // E = sum w res dy (points: w = robust weight; line pixels: w = 1), the chi2 terms and sum |res|
        struct PixSums { double A, B, C, D, E, Chi; float Abs; };
        auto unpack5 = [](uint32_t lo, uint32_t hi, int sh, float* o) {
          const uint32_t w0 = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)sh);
          o[0] = (float)(w0 & 0xffu); o[1] = (float)((w0 >> 8) & 0xffu); o[2] = (float)((w0 >> 16) & 0xffu); o[3] = (float)(w0 >> 24);
          o[4] = (float)((hi >> (8 * sh)) & 0xffu);
        };
        // one patch row = four pixels.  WEIGHTED is decided per wave: the slot table lists points first, then line samples, so most rounds
        // are homogeneous and the line-only ones skip the robust weight and the chi2 term.
        // (a packed-FP32 form of this loop, two pixels per v_pk_* instruction, was measured 12 % slower)
        // (every POINT pixel's chi2 term res*res*w goes to this iteration's plane of chi_terms, one float4 per patch row: what
        //  exact_chi2_pair re-adds in the reference's order on a near tie)
        auto row4 = [&](auto WEIGHTED, bool is_point, const PatchW& pw, const float* top, const float* bot, const float4& r4, const float4& x4,
                        const float4& y4, float4& tv4, PixSums& ps) {
          constexpr bool weighted = decltype(WEIGHTED)::value;
          const float* pr = reinterpret_cast<const float*>(&r4);
          const float* pxp = reinterpret_cast<const float*>(&x4);
          const float* pyp = reinterpret_cast<const float*>(&y4);
          float* tv = reinterpret_cast<float*>(&tv4);
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
              ps.A += wdx * dx; ps.B += wdx * dy; ps.C += wdy * dy;
              ps.D += wdx * rd; ps.E += wdy * rd;
              const float term = __fmul_rn(__fmul_rn(res, res), w);      // :484  chi2 += res*res*weight (float)
              ps.Chi += (double)term;
              tv[x] = is_point ? term : ares;
            } else {
              ps.A += dx * dx; ps.B += dx * dy; ps.C += dy * dy;
              ps.D += dx * rd; ps.E += dy * rd;
              tv[x] = ares;
            }
            ps.Abs += ares;
          }
        };
        // The chi2 terms of a POINT slot (16 floats = 64 contiguous bytes) go to this iteration's plane of chi_terms, row by row (written
        // while the solver is armed or a near tie is being re-run; line pixels are not stored: every byte written costs its time).
        // (two explicit address spaces: one generic pointer would make these flat stores, which also wait for the LDS counter)
        auto chi_store = [&](int p_, int r, const float4& t) {
          const plsvo_v4f t4 = { t.x, t.y, t.z, t.w };
          if (b.chi_lds_pts > 0) ((PLSVO_LDS plsvo_v4f*)(s_win + (iter & 1) * b.chi_lds_pts * 16 + p_ * 16))[r] = t4;   // small batches: the planes are in LDS (kernel-uniform)
          else ((PLSVO_GLOBAL plsvo_v4f*)(chi_it + (unsigned)(p_ * 16)))[r] = t4;                                         // wave-uniform base + 32-bit lane offset
        };
        // what follows a slot's pixel sums: the weights -- points 1; a line's samples share w / mean|res| (H) and w (Jres), :640-688 -- and
        // the slot's 6x6 contribution.  Called by every lane of the wave (wave-uniform branches and a wave-level LDS fence inside).
        auto slot_finish = [&](int p, const int2& meta, bool is_line, bool cand, bool live, double X, double Y, double Z, double z_inv, const PixSums& ps) {
          double wh = 0.0, wj = 0.0;
          if (__any(is_line && cand)) {   // wave-uniform
            if (write_abs) {
              if (is_line && cand) s_abs[p] = live ? ps.Abs : -1.0f;
              wave_lds_fence();   // all samples of a line sit in this wave's round (host layout); two-pass levels: see the barrier below
            }
            if (accumulate && is_line && cand) {
              const int first = meta.y & 0xfffff, N = meta.y >> 20;
              bool good = true; float sum = 0.0f;
              for (int n = 0; n < N; ++n) { const float a = s_abs[first + n]; good = good && (a >= 0.0f); sum += a; }
              const float res_ = (float)((double)sum / (double)N);                 // :647 (divides by #samples)
              if (good && (double)res_ < 200.0) {                                  // :648
                const float w = (float)(1.0 / (1.0 + (double)res_));               // :675
                wh = (double)w / (double)res_;                                     // :681  H += H_ * w / res_
                wj = (double)w;                                                    // :682  Jres += Jres_ * w
                if (p == first) {                                                  // :683-684
                  const float term = __fmul_rn(__fmul_rn(res_, res_), w);
                  acc[27] += (double)term; acc[28] += 1.0;
                  s_lterm[(iter & 1) * scap + (-1 - meta.x)] = term;               // this iteration's plane (exact_chi2_pair)
                }
              } else if (p == first) {
                s_dead[-1 - meta.x] = iter + 1;                                    // :687-688 it->feat3D = NULL
                b.seg_alive[job.seg_off + (-1 - meta.x)] = 0;
              }
            }
          }
          if (accumulate) {
            if (!is_line && live) {
              wh = 1.0; wj = 1.0;
              acc[27] += ps.Chi; acc[28] += (double)PLSVO_PATCH_AREA;
            }
            if (live) { acc[29] += 1.0; if (!is_line && store_chi && b.chi_lds_pts == 0) acc[30] += 1.0; }
            // -- 6x6 expansion: sum_pix w J J^T = fs^2 (r0 (A r0 + B r1)^T + r1 (B r0 + C r1)^T), sum_pix w res J = fs (D r0 + E r1)
            if (wh != 0.0 || wj != 0.0) {
              // Frame::jacobian_xyz2uv (include/plsvo/frame.h:138-160) with its z_inv = 1. / z handed in (plsvo_math.hpp::jacobian_xyz2uv: the same operations)
              double J[12];
              {
                const double z_inv_2 = z_inv * z_inv;
                J[0] = -z_inv; J[1] = 0.0; J[2] = X * z_inv_2; J[3] = Y * J[2]; J[4] = -(1.0 + X * J[2]); J[5] = Y * z_inv;
                J[6] = 0.0; J[7] = -z_inv; J[8] = Y * z_inv_2; J[9] = 1.0 + Y * J[8]; J[10] = -J[3]; J[11] = -X * z_inv;
              }
              const double hs = wh * fs * fs, js = wj * fs;
              const double hA = ps.A * hs, hB = ps.B * hs, hC = ps.C * hs, jD = ps.D * js, jE = ps.E * js;
              double v0[6], v1[6];
#pragma unroll
              for (int k = 0; k < 6; ++k) { v0[k] = hA * J[k] + hB * J[6 + k]; v1[k] = hB * J[k] + hC * J[6 + k]; }
              int k = 0;
#pragma unroll
              for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int jj = i; jj < 6; ++jj) { acc[k] += J[i] * v0[jj] + J[6 + i] * v1[jj]; ++k; }
#pragma unroll
              for (int i = 0; i < 6; ++i) acc[21 + i] -= jD * J[i] + jE * J[6 + i];
            }
          }
        };
        // warp + project a slot's 3-D point (:422-431, :583-594); the pose is re-read from LDS (twelve doubles that would otherwise stay in
        // registers across the whole pass); Patch::isInFrame(halfsize=2) on floorf(u), floorf(v); NaN -> out of frame
        auto project = [&](bool cand_, double X, double Y, double Z, float& u, float& v) -> bool {
          const PLSVO_LDS double* const pq = (const PLSVO_LDS double*)pose_rt;
          const double x_cam = pq[0] * X + pq[1] * Y + pq[2] * Z + pq[9];
          const double y_cam = pq[3] * X + pq[4] * Y + pq[5] * Z + pq[10];
          const double z_cam = pq[6] * X + pq[7] * Y + pq[8] * Z + pq[11];
          u = (float)((job.fx * (x_cam / z_cam) + job.cx) * scale);
          v = (float)((job.fy * (y_cam / z_cam) + job.cy) * scale);
          return cand_ && (u >= 2.0f) && (v >= 2.0f) && (u < colmax) && (v < rowmax);
        };

          // LATENCY SHAPE (a frame owns most of a CU): a lane per slot like the throughput shape, but with what a lone frame can afford --
          // the reference patch as FLOAT rows (192 B per slot, written once per level; no rebuild from the byte record: -340 float
          // instructions per slot and iteration), the slot's 3-D point from LDS, the Jacobian's 1/z issued before the pixel arithmetic.
          // A wave takes the slots [64 u, 64 u + 64) of its units u = wave, wave + NW, ...
          // (Round 5 also built and measured the pass with FOUR LANES PER SLOT -- project / four quad-rounds of 16 slots / expand, through
          //  per-wave LDS scratch: 330 instructions per quad-round against ~700 per 64-slot round here, i.e. twice the issue slots per
          //  slot; a wave's unit took 12 k cycles, the two waves of a SIMD serialise on its issue port and the pass got SLOWER, 15.3 k ->
          //  16.8 k cycles per iteration incl. the wait for the partner wave.  profiles/r05_quad_pass_phase_ticks.log)
          const float4* const cache_f = reinterpret_cast<const float4*>(b.cache_ref);
          for (int pb = (terms_only ? 0 : slot_lo) + wave * 64; pb < n_rounds_slots; pb += T) {
            const int p = pb + lane;
            struct Rows3 { float4 r4, x4, y4; };
            auto load_row = [&](int r) -> Rows3 {          // patch row r of the slot's cached reference patch: does not depend on the pose
              Rows3 c;
              c.r4 = make_float4(0.f, 0.f, 0.f, 0.f); c.x4 = c.r4; c.y4 = c.r4;
              if (p < n_rounds_slots) {
                const float4* const cf = cache_f + ((pbase + p) * 4 + r) * 3;
                c.r4 = cf[0]; c.x4 = cf[1]; c.y4 = cf[2];
              }
              return c;
            };
            Rows3 row_cur = load_row(0);                   // (under the projection)
            int2 meta = make_int2(SLOT_HOLE, 0);
            if (p < n_slots) meta = s_meta[p];
            const bool hole = meta.x == SLOT_HOLE;
            const bool is_line = !hole && meta.x < 0;
            bool cand = !hole && p < n_rounds_slots;
            if (cand && is_line && s_dead[-1 - meta.x]) cand = false;   // line culled at an earlier iteration of this level
            double X = 0.0, Y = 0.0, Z = 1.0;
            if (cand) { X = s_xyz[3 * p]; Y = s_xyz[3 * p + 1]; Z = s_xyz[3 * p + 2]; }
            float u, v;
            const bool live = project(cand, X, Y, Z, u, v);
            // the 5x5 window of the current image: rows vi-2 .. vi+2, columns ui-2 .. ui+2, two aligned dwords per row
            uint32_t wlo[5], whi[5]; int wsh[5];
#pragma unroll
            for (int r = 0; r < 5; ++r) { wlo[r] = 0u; whi[r] = 0u; wsh[r] = 0; }
            PatchW pw = { 0, 0, 0.f, 0.f, 0.f, 0.f };
            if (live) {
              pw = patch_weights(u, v);
#pragma unroll
              for (int r = 0; r < 5; ++r) {
                const int off = (pw.vi - 2 + r) * pitch + (pw.ui - 2);
                wsh[r] = off & 3;
                wlo[r] = *reinterpret_cast<const uint32_t*>(cur_img + (off & ~3)); whi[r] = *reinterpret_cast<const uint32_t*>(cur_img + (off & ~3) + 4);
              }
            }
            const double z_inv = 1.0 / Z;          // (the Jacobian's division: its latency goes under the pixel arithmetic)
            PixSums ps = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0f };
            const bool any_point = __any(live && !is_line) != 0;   // wave-uniform
            const bool chi_out = (accumulate || terms_only) && store_chi && p < job.n_pts;
            if (chi_out && !live) {   // a patch outside the current image contributes nothing (:432-433): +0
#pragma unroll
              for (int r = 0; r < 4; ++r) chi_store(p, r, make_float4(0.f, 0.f, 0.f, 0.f));
            }
            if (__any(live)) {        // wave-uniform (the rows' loads sit outside the divergent part: one request stream per wave)
              float ra[5], rb[5];
              unpack5(wlo[0], whi[0], wsh[0], ra);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const Rows3 c = row_cur;
                if (r < 3) row_cur = load_row(r + 1);     // the next patch row is in flight under this row's arithmetic
                float* const top = (r & 1) ? rb : ra;
                float* const bot = (r & 1) ? ra : rb;
                unpack5(wlo[r + 1], whi[r + 1], wsh[r + 1], bot);
                if (live) {
                  float4 chi_t = make_float4(0.f, 0.f, 0.f, 0.f);
                  if (any_point) row4(std::true_type{}, !is_line, pw, top, bot, c.r4, c.x4, c.y4, chi_t, ps);
                  else row4(std::false_type{}, !is_line, pw, top, bot, c.r4, c.x4, c.y4, chi_t, ps);
                  if (chi_out) chi_store(p, r, chi_t);
                }
              }
            }
            if (!terms_only) slot_finish(p, meta, is_line, cand, live, X, Y, Z, z_inv, ps);
          }
        } else {
        // THROUGHPUT SHAPE (unchanged since round 4; its 255-256 registers leave no room for another formulation of the same code)
        // ONE LANE PER SLOT.  Per round a lane (a) reads its table entry and 3-D point, (b) warps and projects the point and requests the
        // 5x5 window of the current image (five rows, two aligned dwords each), (c) reads the four cached rows of reference intensity and
        // gradient, then evaluates the 16 pixels row by row -- the sums over a patch in the reference's own pixel order -- exchanges the
        // line residuals through LDS and expands the slot's 6x6 contribution.  Table entry and 3-D point of round r+1 are requested
        // before round r's arithmetic.  (Rounds 1-3 of this build gave a slot to a lane PAIR: per-slot work -- projection, Jacobian,
        // line weights, half of the expansion -- was issued twice per slot, a wave-round covered 32 slots and a frame kept 64 slots
        // in flight per SIMD at two waves of 249 VGPRs; a lane per slot issues that work once and keeps 128 slots in flight.)
        struct SlotA { int2 meta; bool cand; double X, Y, Z; };
        auto stage_a = [&](int pb_) -> SlotA {
          SlotA f;
          const int p_ = pb_ + tid;
          f.meta = make_int2(SLOT_HOLE, 0);
          if (p_ < n_slots) f.meta = s_meta[p_];
          f.cand = f.meta.x != SLOT_HOLE;
          if (f.cand && f.meta.x < 0 && s_dead[-1 - f.meta.x]) f.cand = false;   // line culled at an earlier iteration of this level
          f.X = 0.0; f.Y = 0.0; f.Z = 1.0;
          if (f.cand) { f.X = pxyz[3 * p_]; f.Y = pxyz[3 * p_ + 1]; f.Z = pxyz[3 * p_ + 2]; }
          return f;
        };
        struct SlotC { uint4 q[4]; };
        auto stage_c = [&](int pb_, bool cand_) -> SlotC {
          SlotC c;
#pragma unroll
          for (int r = 0; r < 4; ++r) c.q[r] = make_uint4(0u, 0u, 0u, 0u);
          if (cand_) {
            const uint4* const rec = reinterpret_cast<const uint4*>(b.cache_ref) + (pbase + pb_ + tid) * 4;   // one 64-byte record per lane
#pragma unroll
            for (int r = 0; r < 4; ++r) c.q[r] = rec[r];
          }
          return c;
        };
        // stage B: warp + project the slot's point, request the 5x5 window of the current image
        struct SlotB { bool live; float u, v; uint32_t wlo[5], whi[5]; int wsh[5]; };
        auto stage_b = [&](const SlotA& a_) -> SlotB {
          SlotB g;
          // warp the 3-D point, project (:422-431, :583-594); the pose is re-read from LDS every round (twelve doubles that would
          // otherwise stay in registers across the whole pass)
          const PLSVO_LDS double* const pq = (const PLSVO_LDS double*)pose_rt;
          const double x_cam = pq[0] * a_.X + pq[1] * a_.Y + pq[2] * a_.Z + pq[9];
          const double y_cam = pq[3] * a_.X + pq[4] * a_.Y + pq[5] * a_.Z + pq[10];
          const double z_cam = pq[6] * a_.X + pq[7] * a_.Y + pq[8] * a_.Z + pq[11];
          g.u = (float)((job.fx * (x_cam / z_cam) + job.cx) * scale);
          g.v = (float)((job.fy * (y_cam / z_cam) + job.cy) * scale);
          // Patch::isInFrame(halfsize=2) on floorf(u), floorf(v); NaN -> out of frame
          g.live = a_.cand && (g.u >= 2.0f) && (g.v >= 2.0f) && (g.u < colmax) && (g.v < rowmax);
          // the window: rows vi-2 .. vi+2, columns ui-2 .. ui+2, two aligned dwords per row (32-bit offsets from the wave-uniform level base)
#pragma unroll
          for (int r = 0; r < 5; ++r) { g.wlo[r] = 0u; g.whi[r] = 0u; g.wsh[r] = 0; }
          if (g.live) {
            const int ui = (int)floorf(g.u), vi = (int)floorf(g.v);
            const int x0 = ui - 2, y0 = vi - 2;
            if constexpr (kTiled) {
              const int ca = tiled_col_offset(x0 & ~3), cb = tiled_col_offset((x0 & ~3) + 4);
#pragma unroll
              for (int r = 0; r < 5; ++r) {
                const int q = tiled_row_offset(pitch, y0 + r);
                g.wsh[r] = x0 & 3;
                g.wlo[r] = *reinterpret_cast<const uint32_t*>(cur_img + (q + ca)); g.whi[r] = *reinterpret_cast<const uint32_t*>(cur_img + (q + cb));
              }
            } else {
#pragma unroll
              for (int r = 0; r < 5; ++r) {
                const int off = (y0 + r) * pitch + x0;
                g.wsh[r] = off & 3;
                g.wlo[r] = *reinterpret_cast<const uint32_t*>(cur_img + (off & ~3)); g.whi[r] = *reinterpret_cast<const uint32_t*>(cur_img + (off & ~3) + 4);
              }
            }
          }
          return g;
        };
        // (Projecting round r+1 and requesting its window BEFORE round r's arithmetic -- a second SlotA and a SlotB in flight -- was
        //  measured slower on MI355X, 15.1 -> 16.4 ms and 13.9 -> 15.5 ms with the byte records: the registers it takes cost more than
        //  the latency it hides.  profiles/r04d_byte_records_gather_ahead_ab.log.  Round 6 built the same schedule with NO register parked:
        //  everything a round prefetches staged in LDS by LDS-DMA (global_load_lds: the next round's windows as 10 dword pieces, its record
        //  as 4 x 16 B, the 3-D points of the round after) -- parity-green and 10.5 % SLOWER, 13.93 -> 15.41 ms: seventeen DMA pieces per
        //  round cost more to issue than the latency they hide.  tools/patches/r06_lds_dma_prefetch.patch, profiles/r06_lds_dma_prefetch_ab.log)
        SlotA a_nxt = stage_a(0);
        SlotC c_nxt = stage_c(0, a_nxt.cand);
        for (int pb = 0; pb < n_rounds_slots; pb += T) {
          const int p = pb + tid;
          const SlotA sa = a_nxt;
          const SlotC sc = c_nxt;
          const SlotB sb = stage_b(sa);
          a_nxt = stage_a(pb + T);           // (slots beyond the table come back as holes: no loads)
          const bool next_cand = a_nxt.cand;
          const int2 meta = sa.meta;
          const bool hole = meta.x == SLOT_HOLE;
          const bool is_line = !hole && meta.x < 0;
          const bool cand = sa.cand;
          const double X = sa.X, Y = sa.Y, Z = sa.Z;
          const float u = sb.u, v = sb.v;
          const bool live = sb.live;
          const uint32_t* const wlo = sb.wlo; const uint32_t* const whi = sb.whi; const int* const wsh = sb.wsh;

          // -- residuals and the five patch sums over the 16 pixels, row by row
          double sA = 0, sB = 0, sC = 0, sD = 0, sE = 0, sChi = 0;
          float sAbs = 0.0f;
          const bool any_point = __any(live && !is_line) != 0;   // wave-uniform
          // The chi2 terms of a POINT slot (16 floats = 64 contiguous bytes per lane) go to this iteration's plane of chi_terms, row by
          // row (written while the solver is armed or a near tie is being re-run; line pixels are not stored: every byte written costs
          // its time).  A patch outside the current image contributes nothing (:432-433): +0.
          const bool chi_out = (accumulate || terms_only) && store_chi && p < job.n_pts;
          // (two explicit address spaces: one generic pointer would make these flat stores, which also wait for the LDS counter)
          PLSVO_LDS plsvo_v4f* const chi_lds = (PLSVO_LDS plsvo_v4f*)(s_win + (iter & 1) * b.chi_lds_pts * 16 + p * 16);     // small batches: the planes are in LDS (kernel-uniform)
          PLSVO_GLOBAL plsvo_v4f* const chi_glb = (PLSVO_GLOBAL plsvo_v4f*)(chi_it + (unsigned)(p * 16));                    // wave-uniform base + 32-bit lane offset
          auto chi_store = [&](int r, const float4& t) {
            const plsvo_v4f t4 = { t.x, t.y, t.z, t.w };
            if (b.chi_lds_pts > 0) chi_lds[r] = t4; else chi_glb[r] = t4;
          };
          if (chi_out && !live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) chi_store(r, make_float4(0.f, 0.f, 0.f, 0.f));
          }
          if (live) {
            const PatchW pw = patch_weights(u, v);
            auto unpack5 = [](uint32_t lo, uint32_t hi, int sh, float* o) {
              const uint32_t w0 = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)sh);
              o[0] = (float)(w0 & 0xffu); o[1] = (float)((w0 >> 8) & 0xffu); o[2] = (float)((w0 >> 16) & 0xffu); o[3] = (float)(w0 >> 24);
              o[4] = (float)((hi >> (8 * sh)) & 0xffu);
            };
            const bool is_point = !is_line;
            // WEIGHTED is decided per wave: the slot table lists points first, then line samples, so most rounds are
            // homogeneous and the line-only ones skip the robust weight and the chi2 term.
            // (a packed-FP32 form of this loop, two pixels per v_pk_* instruction, was measured 12 % slower)
            // (every POINT pixel's chi2 term res*res*w goes to this iteration's plane of chi_terms, one float4 per patch row: what
            //  exact_chi2_pair re-adds in the reference's order on a near tie)
            auto row4 = [&](auto WEIGHTED, const float* top, const float* bot, const float4& r4, const float4& x4, const float4& y4, float4& tv4) {
              constexpr bool weighted = decltype(WEIGHTED)::value;
              const float* pr = reinterpret_cast<const float*>(&r4);
              const float* pxp = reinterpret_cast<const float*>(&x4);
              const float* pyp = reinterpret_cast<const float*>(&y4);
              float* tv = reinterpret_cast<float*>(&tv4);
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
                  const float term = __fmul_rn(__fmul_rn(res, res), w);      // :484  chi2 += res*res*weight (float)
                  sChi += (double)term;
                  tv[x] = is_point ? term : ares;
                } else {
                  sA += dx * dx; sB += dx * dy; sC += dy * dy;
                  sD += dx * rd; sE += dy * rd;
                  tv[x] = ares;
                }
                sAbs += ares;
              }
            };
            float ra[5], rb[5];
            unpack5(wlo[0], whi[0], wsh[0], ra);
            RecordRows rec;
            rec.start(sc.q);
            auto patch_row = [&](auto RI) {
              constexpr int r = decltype(RI)::value;
              float* const top = (r & 1) ? rb : ra;
              float* const bot = (r & 1) ? ra : rb;
              unpack5(wlo[r + 1], whi[r + 1], wsh[r + 1], bot);
              float4 chi_t = make_float4(0.f, 0.f, 0.f, 0.f);
              float4 r4, x4, y4;
              rec.template row<r>(r4, x4, y4);
              if (any_point) row4(std::true_type{}, top, bot, r4, x4, y4, chi_t);
              else row4(std::false_type{}, top, bot, r4, x4, y4, chi_t);
              if (chi_out) chi_store(r, chi_t);
            };
            patch_row(std::integral_constant<int, 0>{}); patch_row(std::integral_constant<int, 1>{});
            patch_row(std::integral_constant<int, 2>{}); patch_row(std::integral_constant<int, 3>{});
          }
          c_nxt = stage_c(pb + T, next_cand);   // next round's cache rows: this round's are dead now

          // -- weights: points 1; a line's samples share w / mean|res| (H) and w (Jres), :640-688
          double wh = 0.0, wj = 0.0;
          if (__any(is_line && cand)) {   // wave-uniform
            if (write_abs) {
              if (is_line && cand) s_abs[p] = live ? sAbs : -1.0f;
              wave_lds_fence();   // all samples of a line sit in this wave's round (host layout); two-pass levels: see the barrier below
            }
            if (accumulate && is_line && cand) {
              const int first = meta.y & 0xfffff, N = meta.y >> 20;
              bool good = true; float sum = 0.0f;
              for (int n = 0; n < N; ++n) { const float a = s_abs[first + n]; good = good && (a >= 0.0f); sum += a; }
              const float res_ = (float)((double)sum / (double)N);                 // :647 (divides by #samples)
              if (good && (double)res_ < 200.0) {                                  // :648
                const float w = (float)(1.0 / (1.0 + (double)res_));               // :675
                wh = (double)w / (double)res_;                                     // :681  H += H_ * w / res_
                wj = (double)w;                                                    // :682  Jres += Jres_ * w
                if (p == first) {                                                  // :683-684
                  const float term = __fmul_rn(__fmul_rn(res_, res_), w);
                  acc[27] += (double)term; acc[28] += 1.0;
                  s_lterm[(iter & 1) * scap + (-1 - meta.x)] = term;               // this iteration's plane (exact_chi2_pair)
                }
              } else if (p == first) {
                s_dead[-1 - meta.x] = iter + 1;                                    // :687-688 it->feat3D = NULL
                b.seg_alive[job.seg_off + (-1 - meta.x)] = 0;
              }
            }
          }
          if (accumulate) {
            if (!is_line && live) {
              wh = 1.0; wj = 1.0;
              acc[27] += sChi; acc[28] += (double)PLSVO_PATCH_AREA;
            }
            if (live) { acc[29] += 1.0; if (!is_line && store_chi && b.chi_lds_pts == 0) acc[30] += 1.0; }
            // -- 6x6 expansion: sum_pix w J J^T = fs^2 (r0 (A r0 + B r1)^T + r1 (B r0 + C r1)^T), sum_pix w res J = fs (D r0 + E r1)
            if (wh != 0.0 || wj != 0.0) {
              const double xyz[3] = { X, Y, Z };
              double J[12];
              jacobian_xyz2uv(xyz, J);
              const double hs = wh * fs * fs, js = wj * fs;
              const double hA = sA * hs, hB = sB * hs, hC = sC * hs, jD = sD * js, jE = sE * js;
              double v0[6], v1[6];
#pragma unroll
              for (int k = 0; k < 6; ++k) { v0[k] = hA * J[k] + hB * J[6 + k]; v1[k] = hB * J[k] + hC * J[6 + k]; }
              int k = 0;
#pragma unroll
              for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int jj = i; jj < 6; ++jj) { acc[k] += J[i] * v0[jj] + J[6 + i] * v1[jj]; ++k; }
#pragma unroll
              for (int i = 0; i < 6; ++i) acc[21 + i] -= jD * J[i] + jE * J[6 + i];
            }
          }
        }
        }
        if (long_lines && pass == 0) block_sync<T>();   // every sample's |res| sum is in LDS before any line is weighted
      }
      TICK(1);

      // -- reduction (fixed shape): butterfly inside the DPP rows, rows and waves through LDS
      if (!terms_only) {
        double out2[2];
        row_reduce_scatter32(acc, out2);
        const int k0 = row_reduce_scatter32_index(lane);
        *reinterpret_cast<double2*>(s_red + (tid >> 4) * 32 + k0) = make_double2(out2[0], out2[1]);
      }
      block_sync<T>();   // (after a terms-only re-run: the planes it wrote are visible to wave 0)
      TICK(2);

      // -- wave 0: totals (fixed order) in lanes 0..29, cooperative 6x6 solve, then lane 0 takes the accept / roll back /
      //    update decision
      if (wave == 0 && stage == last_stage) {
        // (after terms-only re-runs the totals of the iteration are read back from s_tot and the solve is simply done again)
        double tot = terms_only ? s_tot[lane & 31] : reduce_rows_finish<ROWS>(s_red);
        if constexpr (kQuad) {
          if (pair) {   // two workgroups per frame: the partner's partial sums (value 31 carries its patch count of the level); both add in the
            //             same order, so both hold the same totals bit for bit and take the same decisions from here on
            if ((lane & 31) == 31) tot = (double)s_ctl[5];
            const double theirs = pair_allgather32(xb, rank, xseq, tot, x_lost);
            ++xseq;   // (only wave 0 exchanges, and only its copy of the counter is ever read)
            tot = rank == 0 ? tot + theirs : theirs + tot;
          }
        }
        if (lane < 32 && !terms_only) s_tot[lane] = tot;   // (0..26 the system, 27..30 chi2 / n_meas / work counters: a terms-only re-run reads them back)
        TICK(3);
        double x[6];
        // (static pivot order in the latency shapes only: the throughput shapes sit at 255-256 registers without scratch, and the few
        //  values the shorter solve keeps live across the iteration tip them into spilling -- there the solve is 2 % of a launch)
        wave_solve6_reg(tot, x, job.ldlt_flavour, kQuad);                      // solve() :699
        TICK(4);
        const double chi_sum = readlane_f64(tot, 27), nm_d = readlane_f64(tot, 28), ev_d = readlane_f64(tot, 29), ev_pt = readlane_f64(tot, 30);
        const unsigned long long nm = (unsigned long long)(nm_d + 0.5);
        // computeResiduals returns float chi2 / n_meas_ (:171,192); chi_sum is the exact sum of the float terms, rounded once
        double new_chi2 = (double)((float)chi_sum / (float)nm);
        double old_chi2 = s_pose[26];                                          // solver chi2_
        // the reference's sequential float sums sit within ~0.25 sqrt(n) 2^-24 (one sigma) of these: inside 2 sqrt(n) 2^-24
        // (> 5 sigma of the difference of two such sums) the order of the two values is taken from the exact float sums
        const float band = kChiBand * __fsqrt_rn((float)nm_d) * 5.9604644775390625e-8f;
        const bool near = iter > 0 && !s_ctl[1] && !isnan(x[0]) && fabs(new_chi2 - old_chi2) <= (double)band * old_chi2;
        const bool have_terms = terms_only || b.chi_lds_pts > 0 || (s_ctl[7] != 0 && s_ctl[8] != 0);   // both iterations' terms were kept (or have just been rebuilt)
        const bool tie = near && have_terms;
        const bool defer = near && !have_terms;   // wave-uniform: rebuild the missing plane(s) first, decide afterwards
        if (defer && lane == 0) {
          s_ctl[10] = (s_ctl[7] != 0 ? 0 : 1) | (s_ctl[8] != 0 ? 0 : 2);
          if (s_ctl[8] == 0) {   // rotation matrix + translation of old_model_ for stage 2 (s_red is free until the next reduction)
            const SE3d om = se3_load(s_pose + 19);
            quat_to_matrix(om.q, s_red); s_red[9] = om.t[0]; s_red[10] = om.t[1]; s_red[11] = om.t[2];
          }
        }
        if (tie) {   // wave-uniform
          if (b.chi_lds_pts > 0)
            // (two workgroups per frame: rank 0 re-adds the points' terms, rank 1 the lines' -- each sum starts at +0, so each side's
            //  output IS its partial sum)
            exact_chi2_pair_lds((const PLSVO_LDS float*)(s_win + (iter & 1) * b.chi_lds_pts * 16), (const PLSVO_LDS float*)(s_win + ((iter & 1) ^ 1) * b.chi_lds_pts * 16),
                                own_pts ? job.n_pts : 0, own_segs ? job.n_seg : 0, iter, (const PLSVO_LDS int*)s_dead, (const PLSVO_LDS float*)s_lterm, scap, (PLSVO_LDS float*)s_lterm + 2 * scap);
          else
            exact_chi2_pair((const PLSVO_GLOBAL float*)(b.chi_terms + (size_t)(iter & 1) * b.chi_plane + (size_t)job.pt_off * 16),
                            (const PLSVO_GLOBAL float*)(b.chi_terms + (size_t)((iter & 1) ^ 1) * b.chi_plane + (size_t)job.pt_off * 16),
                            job.n_pts, job.n_seg, iter, (const PLSVO_LDS int*)s_dead, (PLSVO_LDS float*)s_win, (const PLSVO_LDS float*)s_lterm, scap,
                            (PLSVO_LDS float*)s_lterm + 2 * scap);
          float FA = s_lterm[2 * scap], FB = s_lterm[2 * scap + 1];   // chi2 of this / of the previous iteration, before the division
          if constexpr (kQuad) {
            if (pair) {   // chi2 = pt_chi2 + seg_chi2 (:171): the points' sums from rank 0, the lines' from rank 1
              const double mine = (lane & 31) == 0 ? (double)FA : ((lane & 31) == 1 ? (double)FB : 0.0);
              const double theirs = pair_allgather32(xb, rank, xseq, mine, x_lost);
              ++xseq;
              const float TA = (float)readlane_f64(theirs, 0), TB = (float)readlane_f64(theirs, 1);
              FA = rank == 0 ? __fadd_rn(FA, TA) : __fadd_rn(TA, FA);
              FB = rank == 0 ? __fadd_rn(FB, TB) : __fadd_rn(TB, FB);
            }
          }
          new_chi2 = (double)(FA / (float)nm);
          old_chi2 = (double)(FB / (float)(unsigned long long)(s_pose[30] + 0.5));
        }
        TICK(7);
        if (x_lost && lane == 0) s_ctl[3] = 2;   // the partner workgroup never answered: flag the frame, stop iterating (both sides time out alike)
        if (lane == 0 && !defer) {
          s_ctl[10] = 0;
          s_pose[27] += ev_d;
          s_pose[31] += ev_pt;
          s_pose[30] = nm_d;                                                   // n_meas_ of this iteration, for the next one's tie
          s_ctl[2] += 1;
          if (tie) s_ctl[6] += 1;
          if (near && !have_terms) s_ctl[9] += 1;   // decided on the rounded-once sums after all
          int stop = s_ctl[1];
          if (isnan(x[0]) || x_lost) stop = 1;                                 // :700
          SE3d model = se3_load(s_pose + 12);
          int accepted, brk = 0;
          if ((iter > 0 && new_chi2 > old_chi2) || stop) {
            model = se3_load(s_pose + 19);                                     // rollback to old_model
            accepted = 0; brk = 1;
          } else {
            double mx[6];
            for (int k = 0; k < 6; ++k) mx[k] = -x[k];
            const SE3d nm_ = se3_mul_dev(model, se3_exp_dev(mx));              // update() :709
            se3_store(model, s_pose + 19);                                     // old_model = model
            model = nm_;
            s_pose[26] = new_chi2;
            accepted = 1;
            if (norm_max6(x) <= job.eps) brk = 1;
          }
          s_ctl[8] = s_ctl[7];
          s_ctl[7] = (accepted && norm_max6(x) < 1e-3) ? 1 : 0;                // armed for the next iteration
          se3_store(model, s_pose + 12);
          quat_to_matrix(model.q, s_pose); s_pose[9] = model.t[0]; s_pose[10] = model.t[1]; s_pose[11] = model.t[2];
          s_ctl[1] = stop; s_ctl[0] = brk;
          if (b.log && lead) {
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
      TICK(5);
      block_sync<T>();
      if (stage == 0) redo_mask = s_ctl[10];   // (0 unless wave 0 deferred its decision)
     }   // stages of the iteration
      if (b.log && tid == 0 && lead) {  // H and Jres of the trace come from s_tot (written by lanes 0..26 above)
        const int lc = st->log_count - 1;
        if (lc >= 0 && lc < b.log_cap) {
          plsvo_align_iterlog* r = b.log + (size_t)job_id * b.log_cap + lc;
          for (int i = 0; i < 6; ++i) for (int jj = 0; jj < 6; ++jj) r->H[i * 6 + jj] = s_tot[sym6_index(i, jj)];
          for (int k = 0; k < 6; ++k) r->Jres[k] = s_tot[21 + k];
        }
      }
#if defined(PLSVO_TIMING) && PLSVO_TIMING == 3
      if (iter == 0) TICK_RAW(7); else TICK_RAW(6);   // the first iteration of the launch runs its code cold (instruction fetch): slot 7, the others slot 6
#endif
      TICK(6);
      if (s_ctl[0]) break;
    }

    if (tid == 0) {   // work counters stay in LDS until the end of the launch (a global read-modify-write here would stall every level)
      if (lead) st->iters[level] = s_ctl[2];
      s_pose[28] += pair ? s_tot[31] : (double)s_ctl[5];   // (two workgroups per frame: value 31 of the exchanged totals = both patch counts)
      s_pose[29] += s_pose[27];
    }
  }  // levels

  block_sync<T>();
  if (tid == 0 && lead) {
    for (int k = 0; k < 7; ++k) st->T[k] = s_pose[12 + k];
    if (b.poses) for (int k = 0; k < 7; ++k) b.poses[7 * job_id + k] = s_pose[12 + k];
    st->chi2 = s_pose[26];
    st->patch_levels += (unsigned long long)(s_pose[28] + 0.5);
    st->patch_iters += (unsigned long long)(s_pose[29] + 0.5);
    if (b.work_key) b.work_key[job_id] = (int)fmin(s_pose[29] + 0.5, 2147483647.0);   // what this frame cost: the next launch's sort key
    st->patch_iters_pt += (unsigned long long)(s_pose[31] + 0.5);
    st->stop = s_ctl[1];
    if (s_ctl[3]) st->error = s_ctl[3];
    st->chi2_ties += s_ctl[6];
    st->chi2_unarmed += s_ctl[9];
    st->n_meas = (unsigned long long)(s_tot[28] + 0.5);
    for (int i = 0; i < 6; ++i) for (int jj = 0; jj < 6; ++jj) st->H[i * 6 + jj] = s_tot[sym6_index(i, jj)];
#ifdef PLSVO_TIMING
    for (int k = 0; k < 8; ++k) st->phase_ticks[k] += s_time[k];
#endif
  }
}


// LDS bytes the kernel needs for slot capacity `cap` and segment capacity `scap` (host side helper)
size_t align_level_lds_bytes(int threads, int cap, int scap, int chi_lds_pts) { return align_lds_used(threads, cap, scap, chi_lds_pts); }

}  // namespace plsvo_hip
extern "C" const char* plsvo_hip_build_flags(void) { return ""; }   // no compile-time experiment switch is left in this build
namespace plsvo_hip {

template <int T>
static hipError_t launch_fused_T(const AlignBatchDev& b, int cap, int scap, int level_hi, int level_lo, int do_init, size_t lds, hipStream_t stream) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(align_fused_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  // one workgroup per frame, or (b.pair) two: blocks q and q + 8 of every group of 16
  const unsigned grid = (T >= kQuadMinThreads && b.pair) ? (unsigned)((b.n_jobs + 7) / 8) * 16u : (unsigned)b.n_jobs;
#ifdef PLSVO_WAVE_EMU
  wave_emu::pair_stride() = (T >= kQuadMinThreads && b.pair) ? 8 : 0;   // the emulator runs the two workgroups of a frame on two OS threads
#endif
  hipLaunchKernelGGL((align_fused_kernel<T>), dim3(grid), dim3(T), lds, stream, b, cap, scap, level_hi, level_lo, do_init);
#ifdef PLSVO_WAVE_EMU
  wave_emu::pair_stride() = 0;
#endif
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Launch order from MEASURED work.  The hardware hands out workgroups in blockIdx order as slots free up, so a launch of 32768 frames on
// 2048 resident slots is a list schedule: in an arbitrary order its tail -- the last frames to start may be the longest -- costs ~5 % of
// the launch (frames take 10 .. 30 Gauss-Newton iterations; simulated: makespan 292 against an ideal 277, 278 longest-first).  The host
// can only sort by patch count (the stage call); what a frame really costs is known after it ran: this kernel sorts the jobs of a resident
// batch by the patch-iterations of its LAST launch (written by every frame's workgroup as it ends; counting sort over 1024 bins, longest first) into the order
// the NEXT launch of the same batch uses.  A tracker's streams change slowly from frame to frame, a benchmark's not at all.  The results of a
// job do not depend on where it sits in the launch (tests: batch == single, bit for bit), so this is scheduling only.  One workgroup.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void align_reorder_kernel(const int* work_key, int n, int* order_out, int shift) {
  __shared__ int s_hist[1024];
  __shared__ int s_scan[1024];
  const int tid = threadIdx.x;
  auto bin_of = [&](int j) -> int {
    const int b_ = work_key[j] >> shift;                               // alignment: bins of 128 patch-iterations (a config-2 frame: ~6000)
    return 1023 - (b_ > 1023 ? 1023 : (b_ < 0 ? 0 : b_));              // descending: the most work first
  };
  s_hist[tid] = 0;
  __syncthreads();
  for (int j = tid; j < n; j += 1024) atomicAdd(&s_hist[bin_of(j)], 1);
  __syncthreads();
  // exclusive prefix sum over the 1024 bins (Hillis-Steele, ten steps)
  int v = s_hist[tid];
  s_scan[tid] = v;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int add = tid >= d ? s_scan[tid - d] : 0;
    __syncthreads();
    s_scan[tid] += add;
    __syncthreads();
  }
  s_hist[tid] = s_scan[tid] - v;      // first position of the bin
  __syncthreads();
  for (int j = tid; j < n; j += 1024) order_out[atomicAdd(&s_hist[bin_of(j)], 1)] = j;
}
hipError_t launch_align_reorder(const int* work_key, int n, int* order_out, int shift, hipStream_t stream) {
  hipLaunchKernelGGL(align_reorder_kernel, dim3(1), dim3(1024), 0, stream, work_key, n, order_out, shift);
  return hipGetLastError();
}

hipError_t launch_align_levels(const AlignBatchDev& b, int cap, int scap, int level_hi, int level_lo, int do_init, int threads, size_t lds,
                               hipStream_t stream) {
  switch (threads) {
    case 64: return launch_fused_T<64>(b, cap, scap, level_hi, level_lo, do_init, lds, stream);
    case 128: return launch_fused_T<128>(b, cap, scap, level_hi, level_lo, do_init, lds, stream);
    case 256: return launch_fused_T<256>(b, cap, scap, level_hi, level_lo, do_init, lds, stream);
    case 512: return launch_fused_T<512>(b, cap, scap, level_hi, level_lo, do_init, lds, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace plsvo_hip
