// plsvo_wave.hpp -- wave64 building blocks for the gfx950 kernels: DPP reductions and a wave-cooperative
// 6x6 symmetric solve (one matrix entry per lane).
#pragma once
#include <hip/hip_runtime.h>

#include "plsvo_math.hpp"

namespace plsvo_hip {

template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_bcast_f64(double v) {  // lanes outside ROW_MASK receive 0
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
#define DPP_QUAD_XOR1 0xB1  // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E  // quad_perm [2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143

// sum over the 4 lanes of a quad, result in all 4 lanes
__device__ __forceinline__ double quad_sum(double v) {
  v += dpp_mov_f64<DPP_QUAD_XOR1>(v);
  v += dpp_mov_f64<DPP_QUAD_XOR2>(v);
  return v;
}
// sum over the 64 lanes of a wave; the total is valid in lane 63
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v = quad_sum(v);
  v += dpp_mov_f64<DPP_ROW_HALF_MIRROR>(v);
  v += dpp_mov_f64<DPP_ROW_MIRROR>(v);
  v += dpp_bcast_f64<DPP_ROW_BCAST15, 0xA>(v);
  v += dpp_bcast_f64<DPP_ROW_BCAST31, 0xC>(v);
  return v;
}

// sum of an int over the 64 lanes of a wave; the total is valid in lane 63
__device__ __forceinline__ int wave_sum_i32_to_lane63(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, DPP_QUAD_XOR1, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_QUAD_XOR2, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_HALF_MIRROR, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_MIRROR, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST15, 0xA, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST31, 0xC, 0xf, false);
  return v;
}

// ------------------------------------------------------------------------------------------------
// Reduce-scatter of 32 lane-private doubles over the 16 lanes of every DPP row (butterfly, fixed shape):
// step s halves the number of values a lane carries and pairs it with the lane whose index differs in bit s,
// so after four steps lane l holds the ROW totals of the two values k0, k0+1 with
//   k0 = 16*bit0(l) + 8*bit1(l) + 4*bit2(l) + 2*bit3(l).
// 30 exchanges + adds instead of the 4 x 32 of a plain all-lanes tree; the four rows of a wave (and the waves
// of a workgroup) are then combined through LDS by reduce_rows_finish().  Deterministic: the pairing is fixed.
// lane^4 / lane^8 have no single DPP pattern: two bank-masked row shifts fill the two halves of the pairing.
// ------------------------------------------------------------------------------------------------
#define DPP_ROW_SHL(n) (0x100 + (n))   // lane l reads lane l+n of its row
#define DPP_ROW_SHR(n) (0x110 + (n))   // lane l reads lane l-n of its row
template <int SHIFT, int BANKS_LO, int BANKS_HI>
__device__ __forceinline__ double dpp_row_xor_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  int rlo = __builtin_amdgcn_update_dpp(0, lo, DPP_ROW_SHL(SHIFT), 0xf, BANKS_LO, false);
  rlo = __builtin_amdgcn_update_dpp(rlo, lo, DPP_ROW_SHR(SHIFT), 0xf, BANKS_HI, false);
  int rhi = __builtin_amdgcn_update_dpp(0, hi, DPP_ROW_SHL(SHIFT), 0xf, BANKS_LO, false);
  rhi = __builtin_amdgcn_update_dpp(rhi, hi, DPP_ROW_SHR(SHIFT), 0xf, BANKS_HI, false);
  return __hiloint2double(rhi, rlo);
}
__device__ __forceinline__ double dpp_xor4_f64(double v) { return dpp_row_xor_f64<4, 0x5, 0xA>(v); }   // banks = groups of 4 lanes
__device__ __forceinline__ double dpp_xor8_f64(double v) { return dpp_row_xor_f64<8, 0x3, 0xC>(v); }

__device__ __forceinline__ void row_reduce_scatter32(const double* v, double* out2) {
  const int lane = threadIdx.x & 63;
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0, b3 = (lane & 8) != 0;
  double w[16], u[8], t[4];
#pragma unroll
  for (int i = 0; i < 16; ++i) { const double keep = b0 ? v[i + 16] : v[i], send = b0 ? v[i] : v[i + 16]; w[i] = keep + dpp_mov_f64<DPP_QUAD_XOR1>(send); }
#pragma unroll
  for (int i = 0; i < 8; ++i) { const double keep = b1 ? w[i + 8] : w[i], send = b1 ? w[i] : w[i + 8]; u[i] = keep + dpp_mov_f64<DPP_QUAD_XOR2>(send); }
#pragma unroll
  for (int i = 0; i < 4; ++i) { const double keep = b2 ? u[i + 4] : u[i], send = b2 ? u[i] : u[i + 4]; t[i] = keep + dpp_xor4_f64(send); }
#pragma unroll
  for (int i = 0; i < 2; ++i) { const double keep = b3 ? t[i + 2] : t[i], send = b3 ? t[i] : t[i + 2]; out2[i] = keep + dpp_xor8_f64(send); }
}
// index of the first of the two values lane `lane` holds after row_reduce_scatter32
__device__ __forceinline__ int row_reduce_scatter32_index(int lane) {
  return ((lane & 1) << 4) | ((lane & 2) << 2) | (lane & 4) | ((lane & 8) >> 2);
}
// Combine `rows` row partials (32 doubles each, written to LDS as s_rows[row * 32 + k]) in a fixed order:
// must be called by one full wave after the partials are visible; lane l (and l ^ 32) returns the total of value l & 31.
template <int ROWS>
__device__ __forceinline__ double reduce_rows_finish(const double* s_rows) {
  const int lane = threadIdx.x & 63, k = lane & 31, h = lane >> 5;
  const double* src = s_rows + (h * (ROWS / 2)) * 32 + k;
  double v[ROWS / 2];
#pragma unroll
  for (int r = 0; r < ROWS / 2; ++r) v[r] = src[r * 32];     // independent LDS reads, issued back to back
  double mine = v[0];
#pragma unroll
  for (int r = 1; r < ROWS / 2; ++r) mine += v[r];
  return mine + __shfl_xor(mine, 32, 64);
}

// LDS (and global) exchange between the lanes of ONE wave: the memory pipelines keep a wave's accesses in order, the fences
// keep the compiler from moving them (no s_barrier is emitted)
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ double readlane_f64(double v, int src_lane /*wave-uniform*/) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// index of entry (r,c) in the row-major upper triangle of a symmetric 6x6 (21 entries)
__device__ __forceinline__ int sym6_index(int r, int c) {
  const int a = r < c ? r : c, b = r < c ? c : r;
  return a * 6 - (a * (a - 1)) / 2 + (b - a);
}

// 1/b and a/b without the IEEE division's scaling and fix-up steps (v_rcp_f64 / v_rsq_f64 + two Newton steps): results
// within 1 ulp for normal, finite operands.  Used only where the reference's own libraries leave the last bit open
// (Eigen's LDLT column scaling, Sophus' quaternion normalisation) and the value is O(1): the serial part of a Gauss-Newton
// iteration is latency-bound and an IEEE f64 division is a chain of ~14 dependent instructions.
__device__ __forceinline__ double fast_rcp(double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double fast_div(double a, double b) {
  const double r = fast_rcp(b);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ double fast_sqrt(double x) {   // x >= 0, finite; sqrt(0) = 0
  if (!(x > 0.0)) return x == 0.0 ? 0.0 : sqrt(x);
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  return fma(fma(-g, g, x), h, g);
}

// Solve H x = rhs for a symmetric 6x6 H, cooperatively by ONE FULL WAVE (all 64 lanes must be active).
//   tot[0..20]  upper triangle of H (row-major), tot[21..26] rhs
//   x[0..5]     result, returned in registers of every lane (wave-uniform)
// Method: Gauss-Jordan elimination of the augmented 6x7 system in the pivot order of Eigen's LDLT (src/sparse_img_align.cpp:699
// and src/pose_optimizer.cpp:170 call H.ldlt().solve()).  Eigen's unblocked LDLT is left-looking -- step k updates column k
// only -- so "the largest remaining |diagonal|" it pivots on is the largest remaining ORIGINAL diagonal entry: the order is
// fixed by the input (descending |H_ii|; on exact ties the first in Eigen's permuted order) and the D entries are the Schur pivots met in that order.
// Zero pivots follow Eigen 3.1...3.2.1 (libeigen3-dev 3.2.0 of Ubuntu 14.04, one of the three platforms PL-SVO names; the rule
// under which rank-deficient systems have a reproducible answer at all): with cutoff = eps * largest |H_ii|, the factorisation
// ends when the next original diagonal is below the cutoff, and a Schur pivot with |d| <= cutoff gives a zero component
// (solve()'s pseudo-inverse of D: |d| <= max|D| * eps; max|D| is the first pivot for the positive semi-definite H of this
// path).  Full-rank systems never meet either rule.  NaN/Inf propagate into x; an all-zero system returns x = 0.
// Lane 8*i+j holds entry (i,j); column 6 is the rhs.
// The serial part of a Gauss-Newton iteration is one wave issuing dependent instructions at ~4-5 cycles each, so a step
// is written for instruction count: the pivot is a DPP max over the lanes' own |diagonal| keys + one ballot (no per-row
// readlanes, no compare chain), its reciprocal comes from fast_rcp while the two ds_bpermute round trips (row p, column p)
// are in flight, and the update is one multiply + one fma per lane.
__device__ __forceinline__ void wave_solve6_core(double m, double* x, int flavour = 320, bool allow_static_order = true);

__device__ __forceinline__ void wave_solve6(const double* tot, double* x) {
  const int lane = threadIdx.x & 63;
  const int i = lane >> 3, j = lane & 7;
  double m = 0.0;
  if (i < 6 && j < 6) m = tot[sym6_index(i, j)];
  else if (i < 6 && j == 6) m = tot[21 + i];
  wave_solve6_core(m, x);
}

// the same, with the 27 inputs held in registers: lane k (k < 27) passes tot[k] in `tot_lane`
__device__ __forceinline__ void wave_solve6_reg(double tot_lane, double* x, int flavour = 320, bool allow_static_order = true) {
  const int lane = threadIdx.x & 63;
  const int i = lane >> 3, j = lane & 7;
  const int src = (i < 6 && j < 6) ? sym6_index(i, j) : ((i < 6 && j == 6) ? 21 + i : 0);
  double m = __shfl(tot_lane, src, 64);
  if (!(i < 6 && j <= 6)) m = 0.0;
  wave_solve6_core(m, x, flavour, allow_static_order);
}

__device__ __forceinline__ double bpermute_f64(int byte_addr, double v) {
  const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// max over the 64 lanes; the result is valid in lane 63 (v_max_f64: a NaN operand loses against a number)
__device__ __forceinline__ double wave_max_to_lane63(double v) {
  v = fmax(v, dpp_mov_f64<DPP_QUAD_XOR1>(v));
  v = fmax(v, dpp_mov_f64<DPP_QUAD_XOR2>(v));
  v = fmax(v, dpp_mov_f64<DPP_ROW_HALF_MIRROR>(v));
  v = fmax(v, dpp_mov_f64<DPP_ROW_MIRROR>(v));
  // lanes outside the row mask receive 0 from the broadcast: harmless, every key of interest is >= 0 and lane 63 is inside
  v = fmax(v, dpp_bcast_f64<DPP_ROW_BCAST15, 0xA>(v));
  v = fmax(v, dpp_bcast_f64<DPP_ROW_BCAST31, 0xC>(v));
  return v;
}

// flavour (wave-uniform): 320 = the rule above; 330 = Eigen 3.2.2 and later (what Ubuntu 16.04's libeigen3-dev 3.3-beta ships): no
// cutoff -- the factorisation never stops early, only an exactly zero pivot leaves its column unscaled, and solve() drops
// |d| <= 1/DBL_MAX only.  Identical arithmetic on every full-rank system; on rank-deficient ones 330 divides rounding residue by
// rounding residue (tests/test_solve_model.py), which no two implementations reproduce alike.
__device__ __forceinline__ void wave_solve6_core(double m, double* x, int flavour, bool allow_static_order) {
  const int lane = threadIdx.x & 63;
  const int i = lane >> 3, j = lane & 7;
  const int addr_row = 4 * j, addr_col = 4 * 8 * i;      // byte addresses of lanes (p,j) / (i,p) once 32p / 4p is added
  bool diag_active = (i == j) && (i < 6);                // this lane holds a diagonal entry not yet used as pivot
  const bool in_system = (i < 6) && (j <= 6);
  const double key0 = fabs(m);                           // |original diagonal|: Eigen's pivot order is fixed by the input
  unsigned zero_piv = 0u;
  double cutoff = 0.0;
  int last_p = 0;
  // STATIC ORDER (the usual case): six distinct, non-NaN |diagonals| -- the pivot order is their descending sort, known before the first
  // elimination step.  Lane (i,j) of the 6x6 block compares |H_jj| with |H_ii|; one ballot gives every diagonal its rank (the number of
  // larger ones), and the six steps run without a search: pivot = v_readlane, its reciprocal under the two ds_bpermute round trips, one
  // multiply + one fma per lane (~16 instructions per step; with the per-step DPP maximum, ballot and tie test it was ~50, and the
  // solve 4.5 k of the 25 k cycles of a lone frame's Gauss-Newton iteration -- profiles/r05a_phase_ticks_b1.log).  Same arithmetic in
  // the same order as the search below: bit-identical results (tests/test_wave_host.py compares the two on every system it solves).
  // Exact ties (Eigen: the first in its PERMUTED order) and NaN diagonals take the search.
  const bool in66 = (i < 6) && (j < 6);
  const double k_i = bpermute_f64(4 * 9 * (i < 6 ? i : 0), key0), k_j = bpermute_f64(4 * 9 * (j < 6 ? j : 0), key0);
  const unsigned long long larger = __ballot(in66 && k_j > k_i);
  const unsigned long long odd = __ballot(in66 && ((i != j && k_j == k_i) || k_i != k_i));
  if (allow_static_order && odd == 0ull) {               // wave-uniform
    int ord[6] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const int rank = __popcll(larger & (0x3full << (8 * a)));
#pragma unroll
      for (int s_ = 0; s_ < 6; ++s_) if (rank == s_) ord[s_] = a;
    }
    const double kmax0 = readlane_f64(key0, 9 * ord[0]);
    cutoff = flavour == 330 ? 0.0 : fabs(2.220446049250313e-16 * kmax0);
#pragma unroll
    for (int step = 0; step < 6; ++step) {
      const int p = ord[step];
      const double kmax = readlane_f64(key0, 9 * p);
      const double piv = readlane_f64(m, 9 * p);
      if (!(kmax < cutoff) && fabs(piv) > cutoff) {
        const double mp_j = bpermute_f64(addr_row + 32 * p, m);   // M[p][j]
        const double mi_p = bpermute_f64(addr_col + 4 * p, m);    // M[i][p]
        const double rinv = fast_rcp(piv);
        if (in_system && i != p) m = fma(-(mi_p * mp_j), rinv, m);
      } else {
        zero_piv |= 1u << p;
      }
      last_p = p;
    }
  } else {
  int pos = diag_active ? i : 64;                        // current position of this diagonal entry under Eigen's transpositions
#pragma unroll
  for (int step = 0; step < 6; ++step) {
    // pivot: largest original |diagonal| among the rows not yet eliminated
    const double key = diag_active ? key0 : -1.0;
    const double kmax = readlane_f64(wave_max_to_lane63(key), 63);
    unsigned long long cand = __ballot(diag_active && key == kmax);
    if (cand == 0ull) cand = __ballot(diag_active);      // every remaining diagonal is NaN: take the first (as a compare chain would)
    int plane = __builtin_ctzll(cand);                   // lane 9p
    if (cand & (cand - 1ull)) {
      // exactly equal diagonals (points-only pose optimisation: A00 == A11): Eigen's maxCoeff takes the first one in its PERMUTED
      // order, i.e. the smallest current position -- rare, wave-uniform, so a scalar loop over the tied lanes
      int best = 64;
      for (unsigned long long c = cand; c; c &= c - 1ull) {
        const int l = __builtin_ctzll(c);
        const int ps = __builtin_amdgcn_readlane(pos, l);
        if (ps < best) { best = ps; plane = l; }
      }
    }
    const int p = (plane * 57) >> 9;                     // plane / 9 for plane in {0, 9, .., 45}
    const double piv = readlane_f64(m, plane);
    // Eigen swaps positions `step` and pos[p]: the entry that sat at position `step` moves to where the pivot was
    const int pos_p = __builtin_amdgcn_readlane(pos, plane);
    if (pos == step) pos = pos_p;
    if (step == 0) cutoff = flavour == 330 ? 0.0 : fabs(2.220446049250313e-16 * kmax);
    // Eigen 3.2: stop at "biggest_in_corner < cutoff"; no scaling unless |pivot| > cutoff; D^+ drops |d| <= max|D| eps
    if (!(kmax < cutoff) && fabs(piv) > cutoff) {
      const double mp_j = bpermute_f64(addr_row + 32 * p, m);   // M[p][j]
      const double mi_p = bpermute_f64(addr_col + 4 * p, m);    // M[i][p]
      const double rinv = fast_rcp(piv);
      if (in_system && i != p) m = fma(-(mi_p * mp_j), rinv, m);
    } else {
      zero_piv |= 1u << p;
    }
    diag_active = diag_active && (i != p);
    last_p = p;
  }
  }
  const double dgi = bpermute_f64(addr_col + 4 * i, m);     // M[i][i]
  const double tolerance = 1.0 / 1.7976931348623157e308;
  double xi = (((zero_piv >> i) & 1u) || !(fabs(dgi) > tolerance)) ? ((dgi != dgi || m != m) ? (m + dgi) : 0.0) : m / dgi;   // lanes 8i+6: m = rhs
  // An infinite diagonal (a line whose mean |residual| is exactly 0 -- a static camera -- makes `H += H_line * w / res_`,
  // src/sparse_img_align.cpp:681, divide by zero): Eigen's cutoff and solve() tolerance are then inf as well, no column is scaled,
  // D^+ zeroes every component and the back-substitution multiplies those zeros with the infinite L entries: NaN everywhere
  // except in the component pivoted last (nothing left to subtract) and in components whose own row is finite.  x[0] = NaN is what
  // makes the reference set stop_ (:700).
  if (cutoff > 1.7976931348623157e308) xi = (i == last_p || fabs(dgi) <= 1.7976931348623157e308) ? 0.0 : __builtin_nan("");
#pragma unroll
  for (int r = 0; r < 6; ++r) x[r] = readlane_f64(xi, 8 * r + 6);
}

// ------------------------------------------------------------------------------------------------
// All-gather of 32 doubles between the TWO workgroups that share a frame (align_kernels.hip, latency shapes): called by one full wave of
// each; lane l hands in value l & 31 (lanes 32..63 repeat lanes 0..31) and gets the PEER's value l & 31 back.
// Transport: data-tagged 8-byte granules {tag << 32 | half of a double}, one per lane, each written by ONE agent-scope relaxed store
// (global_store_dwordx2 ... sc1: write-through, visible to the other CU whichever XCD it sits on) and polled with agent-scope relaxed loads
// (sc1: never served from this CU's L1) until the tag is the exchange's sequence number -- data and tag arrive together, so no fence,
// no flag, no ordering between granules is needed (MI355X_MICROARCH.md, "handoff-1to1": 0.8-1.0 us on an idle chip).  Two buffers used in
// turn (seq & 1): a workgroup can be at most one exchange ahead of its peer, so the granules of exchange n are intact until the peer has
// read them.  `seq` must be non-zero, the same on both sides, and never repeat within the buffer's life (host: launch number << 10).
// xb: the frame's 2 (parity) x 2 (rank) x 64 granules.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void granule_store(unsigned long long* p, unsigned long long v) {
#ifdef PLSVO_WAVE_EMU
  __atomic_store_n(p, v, __ATOMIC_RELEASE);
#else
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long* p) {
#ifdef PLSVO_WAVE_EMU
  return __atomic_load_n(p, __ATOMIC_ACQUIRE);
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// The poll is BOUNDED (kPairPollMax rounds of a load + s_sleep, ~1 s): a partner that never publishes -- tags out of step, a workgroup
// that was never scheduled -- sets `lost` instead of hanging the GPU; the caller flags the frame (AlignStateDev::error = 2) and stops.
constexpr unsigned kPairPollMax = 1u << 20;
__device__ __forceinline__ double pair_allgather32(unsigned long long* xb, int rank, unsigned seq, double mine, bool& lost) {
  const int lane = threadIdx.x & 63;
  const double v = __shfl(mine, lane >> 1, 64);                                   // lane l ships half (l & 1) of value l >> 1
  const unsigned half = (lane & 1) ? (unsigned)__double2hiint(v) : (unsigned)__double2loint(v);
  unsigned long long* const own = xb + (((seq & 1u) * 2u + (unsigned)rank) * 64u + (unsigned)lane);
  const unsigned long long* const peer = xb + (((seq & 1u) * 2u + (unsigned)(rank ^ 1)) * 64u + (unsigned)lane);
  granule_store(own, ((unsigned long long)seq << 32) | (unsigned long long)half);
  unsigned long long g;
  for (unsigned polls = 0;; ++polls) {
    g = granule_load(peer);
    if (!__any((unsigned)(g >> 32) != seq)) break;
    if (polls >= kPairPollMax || lost) { lost = true; break; }   // (once lost, later exchanges of the launch do not wait again)
#ifdef PLSVO_WAVE_EMU
    wave_emu_yield_thread();
#else
    __builtin_amdgcn_s_sleep(2);
#endif
  }
  const int k = lane & 31;
  const unsigned lo = (unsigned)__shfl((int)(unsigned)g, 2 * k, 64), hi = (unsigned)__shfl((int)(unsigned)g, 2 * k + 1, 64);
  return __hiloint2double((int)hi, (int)lo);
}

__device__ __forceinline__ Quat quat_normalized_fast(const Quat& a) {
  const double inv = fast_rcp(fast_sqrt(a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w));
  Quat r = { a.x * inv, a.y * inv, a.z * inv, a.w * inv };
  return r;
}

// Sophus::SE3::exp (same formulas as plsvo_math.hpp::se3_exp).  Gauss-Newton updates are tiny rotations: for
// theta^2 <= 0.01 the four functions of theta the formulas need -- cos(theta/2), sin(theta/2)/theta, (1 - cos theta)/theta^2
// and (theta - sin theta)/theta^3 -- are evaluated as series in z = theta^2 (truncation < 3e-19 relative), which costs 24 fma
// and needs neither theta itself, nor a division, nor the cancellation of the closed forms.  Larger angles take the closed forms.
__device__ __forceinline__ SE3d se3_exp_dev(const double* u) {
  SE3d r;
  const double ox = u[3], oy = u[4], oz = u[5];
  const double z = ox * ox + oy * oy + oz * oz;
  double real_factor, imag_factor, a, b;
  if (z <= 0.01) {
    // cos(t/2) = sum (-1)^k z^k / (4^k (2k)!)
    real_factor = 1.0 / 3715891200.0;                           // 1/(4^5 10!)
    real_factor = fma(real_factor, -z, 1.0 / 10321920.0);       // 1/(4^4 8!)
    real_factor = fma(real_factor, -z, 1.0 / 46080.0);          // 1/(4^3 6!)
    real_factor = fma(real_factor, -z, 1.0 / 384.0);            // 1/(4^2 4!)
    real_factor = fma(real_factor, -z, 1.0 / 8.0);              // 1/(4 2!)
    real_factor = fma(real_factor, -z, 1.0);
    // sin(t/2)/t = sum (-1)^k z^k / (2 4^k (2k+1)!)
    imag_factor = 1.0 / 81749606400.0;                          // 1/(2 4^5 11!)
    imag_factor = fma(imag_factor, -z, 1.0 / 185794560.0);      // 1/(2 4^4 9!)
    imag_factor = fma(imag_factor, -z, 1.0 / 645120.0);         // 1/(2 4^3 7!)
    imag_factor = fma(imag_factor, -z, 1.0 / 3840.0);           // 1/(2 4^2 5!)
    imag_factor = fma(imag_factor, -z, 1.0 / 48.0);             // 1/(2 4 3!)
    imag_factor = fma(imag_factor, -z, 0.5);
    // (1 - cos t)/t^2 = sum (-1)^k z^k / (2k+2)!
    a = 1.0 / 479001600.0;                                      // 1/12!
    a = fma(a, -z, 1.0 / 3628800.0);                            // 1/10!
    a = fma(a, -z, 1.0 / 40320.0);                              // 1/8!
    a = fma(a, -z, 1.0 / 720.0);                                // 1/6!
    a = fma(a, -z, 1.0 / 24.0);                                 // 1/4!
    a = fma(a, -z, 0.5);
    // (t - sin t)/t^3 = sum (-1)^k z^k / (2k+3)!
    b = 1.0 / 6227020800.0;                                     // 1/13!
    b = fma(b, -z, 1.0 / 39916800.0);                           // 1/11!
    b = fma(b, -z, 1.0 / 362880.0);                             // 1/9!
    b = fma(b, -z, 1.0 / 5040.0);                               // 1/7!
    b = fma(b, -z, 1.0 / 120.0);                                // 1/5!
    b = fma(b, -z, 1.0 / 6.0);
  } else {
    const double theta = sqrt(z);
    real_factor = cos(0.5 * theta);
    imag_factor = sin(0.5 * theta) / theta;
    a = (1 - cos(theta)) / z;
    b = (theta - sin(theta)) / (z * theta);
  }
  Quat q = { imag_factor * ox, imag_factor * oy, imag_factor * oz, real_factor };
  r.q = quat_normalized_fast(q);
  // V = I + a hat(omega) + b hat(omega)^2   (theta < 1e-10 in the reference uses the rotation matrix itself: equal to rounding)
  const double O2_00 = -(oy * oy + oz * oz), O2_11 = -(ox * ox + oz * oz), O2_22 = -(ox * ox + oy * oy);
  const double O2_01 = ox * oy, O2_02 = ox * oz, O2_12 = oy * oz;
  double V[9];
  V[0] = 1.0 + b * O2_00;      V[1] = a * -oz + b * O2_01;  V[2] = a * oy + b * O2_02;
  V[3] = a * oz + b * O2_01;   V[4] = 1.0 + b * O2_11;      V[5] = a * -ox + b * O2_12;
  V[6] = a * -oy + b * O2_02;  V[7] = a * ox + b * O2_12;   V[8] = 1.0 + b * O2_22;
  for (int i = 0; i < 3; ++i) r.t[i] = V[i * 3 + 0] * u[0] + V[i * 3 + 1] * u[1] + V[i * 3 + 2] * u[2];
  return r;
}

// Sophus::SE3::operator* on the device: the quaternion product is renormalised by multiplying with 1/|q| (fast_rcp of
// fast_sqrt) instead of four IEEE divisions -- last-bit differences only
__device__ __forceinline__ SE3d se3_mul_dev(const SE3d& A, const SE3d& B) {
  SE3d r; double rt[3];
  quat_rotate(A.q, B.t, rt);
  r.t[0] = A.t[0] + rt[0]; r.t[1] = A.t[1] + rt[1]; r.t[2] = A.t[2] + rt[2];
  r.q = quat_normalized_fast(quat_mul(A.q, B.q));
  return r;
}


}  // namespace plsvo_hip
