// wave_emu.hpp -- a lock-step wave64 emulator for the HOST, test infrastructure only (tests/host/): it lets g++ compile the device
// headers of pl-svo_amd/csrc/ unchanged (tests/host/emu/hip/hip_runtime.h stands in for <hip/hip_runtime.h>) and run their wave-level
// functions on 64 cooperative fibres (ucontext), one per lane.  Every cross-lane operation -- DPP, v_readlane, ds_bpermute, ballot,
// shuffles, the wave barrier -- is a rendezvous: each lane deposits its operand and yields; when all 64 have arrived each computes its
// own result from the others' deposits.  Only wave-uniform control flow around cross-lane operations is supported, and that is CHECKED:
// lanes meeting at different operations abort the run.
//
// DPP semantics implemented (CDNA3/4 ISA, `v_mov_b32_dpp`): quad_perm, row_shl/shr/ror, row_mirror, row_half_mirror, row_bcast15,
// row_bcast31, wave_shr1; a lane whose row or bank is masked out keeps `old`; a lane without a source lane receives 0 with bound_ctrl
// and keeps `old` without.  The functions of the tree that have run on an MI355X (reductions, solve) are the check of these semantics.
#pragma once
#include <ucontext.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

namespace wave_emu {
constexpr int W = 64;
struct State {
  ucontext_t main_ctx, ctx[W];
  std::vector<char> stack[W];
  bool done[W];
  int cur = -1;
  uint64_t slot[2][W];
  int tag[W];
  long seq[W];
  std::function<void()> body;
  unsigned tid_base = 0;
};
inline State& S() { static State s; return s; }
inline int lane() { return S().cur; }
inline void yield_lane() { State& s = S(); swapcontext(&s.ctx[s.cur], &s.main_ctx); }
inline void trampoline() {
  State& s = S();
  s.body();
  s.done[s.cur] = true;
  for (;;) swapcontext(&s.ctx[s.cur], &s.main_ctx);
}
// run `f` on the 64 lanes of one wave, in lock step at every cross-lane operation
template <class F>
void run_wave(F f, unsigned tid_base = 0) {
  State& s = S();
  s.body = f; s.tid_base = tid_base;
  for (int l = 0; l < W; ++l) {
    s.stack[l].assign(512 * 1024, 0);
    getcontext(&s.ctx[l]);
    s.ctx[l].uc_stack.ss_sp = s.stack[l].data();
    s.ctx[l].uc_stack.ss_size = s.stack[l].size();
    s.ctx[l].uc_link = &s.main_ctx;
    makecontext(&s.ctx[l], trampoline, 0);
    s.done[l] = false; s.seq[l] = 0; s.tag[l] = 0;
  }
  for (;;) {
    int alive = 0;
    for (int l = 0; l < W; ++l)
      if (!s.done[l]) { s.cur = l; swapcontext(&s.main_ctx, &s.ctx[l]); }
    long q = -1; int t = 0;
    for (int l = 0; l < W; ++l) {
      if (s.done[l]) continue;
      if (alive++ == 0) { q = s.seq[l]; t = s.tag[l]; }
      else if (s.seq[l] != q || s.tag[l] != t) {
        fprintf(stderr, "wave_emu: lanes diverged at a cross-lane operation (lane %d: op #%ld tag %d, expected #%ld tag %d)\n", l, s.seq[l], s.tag[l], q, t);
        abort();
      }
    }
    if (alive == 0) break;
    if (alive != W) { fprintf(stderr, "wave_emu: %d lanes left the function while others wait at a cross-lane operation\n", W - alive); abort(); }
  }
  s.cur = -1;
}
// rendezvous: deposit v, wait for the other lanes; peek(q, l) then reads lane l's deposit
inline long rendezvous(uint64_t v, int tag) {
  State& s = S();
  const int l = s.cur;
  const long q = ++s.seq[l];
  s.slot[q & 1][l] = v; s.tag[l] = tag;
  yield_lane();
  return q;
}
inline uint64_t peek(long q, int l) { return S().slot[q & 1][l & 63]; }

inline int dpp_source(int l, int ctrl) {   // source lane of lane l, or -1
  const int row = l & ~15, r = l & 15;
  if (ctrl >= 0x000 && ctrl <= 0x0FF) return (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);
  if (ctrl >= 0x101 && ctrl <= 0x10F) { const int n = ctrl - 0x100; return r + n < 16 ? l + n : -1; }
  if (ctrl >= 0x111 && ctrl <= 0x11F) { const int n = ctrl - 0x110; return r >= n ? l - n : -1; }
  if (ctrl >= 0x121 && ctrl <= 0x12F) { const int n = ctrl - 0x120; return row | ((r - n) & 15); }
  if (ctrl == 0x138) return l >= 1 ? l - 1 : -1;   // wave_shr:1
  if (ctrl == 0x140) return row | (15 - r);
  if (ctrl == 0x141) return (l & ~7) | (7 - (l & 7));
  if (ctrl == 0x142) return l >= 16 ? row - 1 : -1;            // lane 15 of the previous row
  if (ctrl == 0x143) return l >= 32 ? 31 : -1;                 // lane 31 (rows 2 and 3)
  fprintf(stderr, "wave_emu: DPP control 0x%x not implemented\n", ctrl); abort();
}
inline int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const long q = rendezvous((uint32_t)src, 1000 + ctrl);
  const int l = lane();
  if (!((row_mask >> (l >> 4)) & 1) || !((bank_mask >> ((l & 15) >> 2)) & 1)) return old;
  const int s = dpp_source(l, ctrl);
  if (s < 0) return bound_ctrl ? 0 : old;
  return (int)(uint32_t)peek(q, s);
}
inline int readlane(int v, int src_lane) {
  const long q = rendezvous((uint32_t)v, 2000);
  return (int)(uint32_t)peek(q, src_lane);
}
inline int ds_bpermute(int byte_addr, int v) {
  const long q = rendezvous((uint32_t)v, 3000);
  return (int)(uint32_t)peek(q, (byte_addr >> 2) & 63);
}
inline unsigned long long ballot(bool p) {
  const long q = rendezvous(p ? 1u : 0u, 4000);
  unsigned long long m = 0;
  for (int l = 0; l < W; ++l) m |= (unsigned long long)(peek(q, l) & 1u) << l;
  return m;
}
inline void wave_barrier() { rendezvous(0, 5000); }
template <class T>
inline T shfl_generic(T v, int src) {
  static_assert(sizeof(T) <= 8, "");
  uint64_t bits = 0; memcpy(&bits, &v, sizeof(T));
  const long q = rendezvous(bits, 6000 + (int)sizeof(T));
  const uint64_t r = peek(q, src);
  T out; memcpy(&out, &r, sizeof(T));
  return out;
}
}  // namespace wave_emu

// ---- what the device headers expect from <hip/hip_runtime.h> ------------------------------------------------------------------
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __noinline__
struct WaveEmuTid { operator unsigned() const { return wave_emu::S().tid_base + (unsigned)wave_emu::lane(); } };
struct WaveEmuDim { WaveEmuTid x; };
static const WaveEmuDim threadIdx = {};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct int2 { int x, y; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
inline double2 make_double2(double x, double y) { double2 r = { x, y }; return r; }
inline int2 make_int2(int x, int y) { int2 r = { x, y }; return r; }
inline float4 make_float4(float x, float y, float z, float w) { float4 r = { x, y, z, w }; return r; }

inline int __double2loint(double v) { uint64_t b; memcpy(&b, &v, 8); return (int)(uint32_t)b; }
inline int __double2hiint(double v) { uint64_t b; memcpy(&b, &v, 8); return (int)(uint32_t)(b >> 32); }
inline double __hiloint2double(int hi, int lo) { const uint64_t b = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double v; memcpy(&v, &b, 8); return v; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fsqrt_rn(float a) { return sqrtf(a); }
inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
#define __builtin_amdgcn_update_dpp wave_emu::update_dpp
#define __builtin_amdgcn_readlane wave_emu::readlane
#define __builtin_amdgcn_ds_bpermute wave_emu::ds_bpermute
#define __builtin_amdgcn_wave_barrier wave_emu::wave_barrier
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_rsq(x) (1.0 / sqrt(x))
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
inline unsigned long long __ballot(int p) { return wave_emu::ballot(p != 0); }
inline int __any(int p) { return wave_emu::ballot(p != 0) != 0ull; }
inline int __all(int p) { return wave_emu::ballot(p != 0) == ~0ull; }
// HIP's shuffles: `width` (a power of two) splits the wave into independent segments, source lanes are relative to the segment
template <class T> inline T __shfl(T v, int src, int width = 64) {
  const int l = wave_emu::lane();
  return wave_emu::shfl_generic(v, (l & ~(width - 1)) | (src & (width - 1)));
}
template <class T> inline T __shfl_xor(T v, int mask, int width = 64) {
  const int l = wave_emu::lane(), t = l ^ mask;
  return wave_emu::shfl_generic(v, (t & ~(width - 1)) == (l & ~(width - 1)) ? t : l);
}
template <class T> inline T __shfl_up(T v, unsigned d, int width = 64) {
  const int l = wave_emu::lane(), base = l & ~(width - 1);
  return wave_emu::shfl_generic(v, (l - (int)d) >= base ? l - (int)d : l);
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
  const int l = wave_emu::lane(), end = (l & ~(width - 1)) + width;
  return wave_emu::shfl_generic(v, (l + (int)d) < end ? l + (int)d : l);
}
