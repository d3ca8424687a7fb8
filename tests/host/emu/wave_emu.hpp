// wave_emu.hpp -- a lock-step wave64 / workgroup emulator for the HOST.  Test infrastructure only (tests/host/): it lets a host compiler
// build the device sources of pl-svo_amd/csrc/ unchanged (tests/host/emu/hip/hip_runtime.h stands in for <hip/hip_runtime.h>) and run
// them on cooperative fibres (ucontext), one per lane.  Every cross-lane operation -- DPP, v_readlane, ds_bpermute, ballot, shuffles,
// the wave barrier -- is a rendezvous of the 64 lanes of a wave: each lane deposits its operand and yields; when it is resumed all lanes
// have arrived and it computes its own result from their deposits.  __syncthreads is a rendezvous of the workgroup.  Only wave-uniform
// control flow around cross-lane operations is supported, and that is CHECKED: lanes of a wave meeting at different operations abort.
//
// DPP semantics implemented (CDNA3/4 ISA, `v_mov_b32_dpp`): quad_perm, row_shl/shr/ror, row_mirror, row_half_mirror, row_bcast15,
// row_bcast31, wave_shr1; a lane whose row or bank is masked out keeps `old`; a lane without a source lane (or whose source lane has left
// the kernel) receives 0 with bound_ctrl and keeps `old` without.  The functions of the tree that have run on an MI355X (reductions,
// solve, the alignment kernel) are the check of these semantics.
#pragma once
#if !defined(__x86_64__)
#include <ucontext.h>
#endif

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define WAVE_EMU_ASAN 1
#include <sanitizer/asan_interface.h>
#include <sanitizer/common_interface_defs.h>
#endif
#endif

namespace wave_emu {
// ---- fibre switch.  x86-64: a hand-written switch of the callee-saved registers and the stack pointer (swapcontext() makes two
// rt_sigprocmask system calls per switch, which was 80 % of the emulation's run time); elsewhere: ucontext.
#if defined(__x86_64__)
struct Context { void* sp = nullptr; };
__attribute__((naked, noinline)) inline void context_switch(void** /*save_sp: rdi*/, void* /*load_sp: rsi*/) {
  __asm__ volatile(
      "pushq %rbp\n pushq %rbx\n pushq %r12\n pushq %r13\n pushq %r14\n pushq %r15\n"
      "movq %rsp, (%rdi)\n"
      "movq %rsi, %rsp\n"
      "popq %r15\n popq %r14\n popq %r13\n popq %r12\n popq %rbx\n popq %rbp\n"
      "ret\n");
}
inline void context_make(Context& c, char* stack, size_t size, void (*entry)()) {
  uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
  void** p = reinterpret_cast<void**>(top);
  *--p = nullptr;                                  // the return address `entry` would return to (it never returns)
  *--p = reinterpret_cast<void*>(entry);           // popped by the switch's `ret`
  for (int k = 0; k < 6; ++k) *--p = nullptr;      // rbp, rbx, r12..r15
  c.sp = p;
}
inline void context_swap(Context& from, Context& to) { context_switch(&from.sp, to.sp); }
#else
struct Context { ucontext_t uc; };
inline void context_make(Context& c, char* stack, size_t size, void (*entry)()) {
  getcontext(&c.uc); c.uc.uc_stack.ss_sp = stack; c.uc.uc_stack.ss_size = size; c.uc.uc_link = nullptr; makecontext(&c.uc, entry, 0);
}
inline void context_swap(Context& from, Context& to) { swapcontext(&from.uc, &to.uc); }
#endif

constexpr int W = 64;
constexpr int MAXT = 1024;
constexpr int TAG_SYNC = 9000;
struct State {
  Context main_ctx;
  std::vector<Context> ctx;
  std::vector<std::vector<char>> stack;
  std::vector<char> done, at_sync;
  std::vector<uint64_t> slot[2];
  std::vector<long> slot_seq[2];   // the operation a deposit belongs to: a lane takes part in operation q iff slot_seq[q & 1][lane] == q
  std::vector<int> tag, site_line;
  std::vector<const char*> site_file;
  std::vector<long> seq;
  int T = 0, cur = -1, sync_arrived = 0;
  std::function<void()> body;
  unsigned tid_base = 0;
  const void* main_stack_bottom = nullptr;
  size_t main_stack_size = 0;
  State() : ctx(MAXT), stack(MAXT), done(MAXT), at_sync(MAXT), tag(MAXT), site_line(MAXT), site_file(MAXT, ""), seq(MAXT) { for (int k = 0; k < 2; ++k) { slot[k].resize(MAXT); slot_seq[k].resize(MAXT); } }
};
inline State& S() { static thread_local State s; return s; }   // (one per OS thread: the two workgroups of a frame run on two threads)
inline int fibre() { return S().cur; }
inline int lane() { return S().cur & 63; }
// AddressSanitizer builds (tests/host/build_emu.sh ... -fsanitize=address) tell the runtime about every stack switch
inline void switch_to_lane(int l) {
  State& s = S();
  s.cur = l;
#ifdef WAVE_EMU_ASAN
  void* fake = nullptr;
  __sanitizer_start_switch_fiber(&fake, s.stack[l].data(), s.stack[l].size());
  context_swap(s.main_ctx, s.ctx[l]);
  __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#else
  context_swap(s.main_ctx, s.ctx[l]);
#endif
}
inline void yield_lane() {
  State& s = S();
#ifdef WAVE_EMU_ASAN
  void* fake = nullptr;
  __sanitizer_start_switch_fiber(&fake, s.main_stack_bottom, s.main_stack_size);
  context_swap(s.ctx[s.cur], s.main_ctx);
  __sanitizer_finish_switch_fiber(fake, &s.main_stack_bottom, &s.main_stack_size);
#else
  context_swap(s.ctx[s.cur], s.main_ctx);
#endif
}
inline void trampoline() {
  State& s = S();
#ifdef WAVE_EMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &s.main_stack_bottom, &s.main_stack_size);
#endif
  s.body();
  s.done[s.cur] = 1;
  for (;;) yield_lane();
}
// run `f` on the T threads of one workgroup (T <= 1024): waves in lock step at every cross-lane operation, the workgroup at __syncthreads
template <class F>
void run_block(int T, F f, unsigned tid_base = 0) {
  State& s = S();
  if (T < 1 || T > MAXT) { fprintf(stderr, "wave_emu: workgroup of %d threads\n", T); abort(); }
  s.body = f; s.tid_base = tid_base; s.T = T; s.sync_arrived = 0;
  for (int l = 0; l < T; ++l) {
    if (s.stack[l].empty()) s.stack[l].assign(384 * 1024, 0);
#ifdef WAVE_EMU_ASAN
    __asan_unpoison_memory_region(s.stack[l].data(), s.stack[l].size());   // frames a finished fibre abandoned
#endif
    context_make(s.ctx[l], s.stack[l].data(), s.stack[l].size(), trampoline);
    s.done[l] = 0; s.at_sync[l] = 0; s.seq[l] = 0; s.tag[l] = 0; s.slot_seq[0][l] = -1; s.slot_seq[1][l] = -1;
  }
  for (;;) {
    // Between two rendezvous a lane runs alone.  On the hardware the lanes of a wave execute every instruction together, so "all lanes
    // read a shared value, then lane 0 overwrites it" needs no fence there; here the lanes are resumed in DESCENDING order so that the
    // usual single writer -- thread 0 / lane 0 -- runs last and its writes cannot reach reads that precede them in program order.
    // (A hazard this does not cover makes the emulated result wrong, never silently right: the tests compare against the oracle.)
    // WAVE_EMU_ORDER=asc0last resumes them 1, 2, .., T-1, then the lane 0 of every wave: a run that passes in BOTH orders does not depend on
    // the lock step for any exchange through memory between lanes other than "lane 0 writes last" (tests/host/run_emu_orders.sh).
    static const bool asc0last = getenv("WAVE_EMU_ORDER") && !strcmp(getenv("WAVE_EMU_ORDER"), "asc0last");
    if (asc0last) {
      for (int l = 0; l < T; ++l) if ((l & 63) != 0 && !s.done[l]) switch_to_lane(l);
      for (int l = 0; l < T; l += 64) if (!s.done[l]) switch_to_lane(l);
    } else {
      for (int l = T - 1; l >= 0; --l)
        if (!s.done[l]) switch_to_lane(l);
    }
    int alive = 0;
    for (int w0 = 0; w0 < T; w0 += W) {   // lock-step check, wave by wave
      long q = -1; int t = 0, first = -1;
      for (int l = w0; l < w0 + W && l < T; ++l) {
        if (s.done[l]) continue;
        ++alive;
        if (first < 0) { first = l; q = s.seq[l]; t = s.tag[l]; }
        else if (s.seq[l] != q || s.tag[l] != t) {
          fprintf(stderr, "wave_emu: lanes of a wave diverged at a cross-lane operation (thread %d: op #%ld tag %d at %s:%d; thread %d: op #%ld tag %d at %s:%d)\n",
                  l, s.seq[l], s.tag[l], s.site_file[l], s.site_line[l], first, q, t, s.site_file[first], s.site_line[first]);
          abort();
        }
      }
    }
    if (alive == 0) break;
    if (s.sync_arrived > 0 && s.sync_arrived == alive) {   // every thread still in the kernel waits at the barrier: release (between sweeps)
      for (int l = 0; l < T; ++l) s.at_sync[l] = 0;
      s.sync_arrived = 0;
    }
  }
  s.cur = -1;
}
template <class F>
void run_wave(F f, unsigned tid_base = 0) { run_block(W, f, tid_base); }

// rendezvous of a wave: deposit v, wait for the other lanes; peek(q, l) then reads the deposit of lane l of this wave
inline long rendezvous(uint64_t v, int tag) {
  State& s = S();
  const int l = s.cur;
  const long q = ++s.seq[l];
  s.slot[q & 1][l] = v; s.slot_seq[q & 1][l] = q; s.tag[l] = tag;
  yield_lane();
  return q;
}
inline uint64_t peek(long q, int l) { State& s = S(); return s.slot[q & 1][(s.cur & ~63) | (l & 63)]; }
inline bool lane_present(long q, int l) { State& s = S(); const int f = (s.cur & ~63) | (l & 63); return f < s.T && s.slot_seq[q & 1][f] == q; }   // lane l takes part in operation q
inline void set_site(const char* f, int line) { State& s = S(); s.site_file[s.cur] = f; s.site_line[s.cur] = line; }
inline void syncthreads() {
  State& s = S();
  const int l = s.cur;
  ++s.seq[l]; s.tag[l] = TAG_SYNC;
  s.at_sync[l] = 1; ++s.sync_arrived;
  while (s.at_sync[l]) yield_lane();
}

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
  if (s < 0 || !lane_present(q, s)) return bound_ctrl ? 0 : old;
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
  for (int l = 0; l < W; ++l) if (lane_present(q, l)) m |= (unsigned long long)(peek(q, l) & 1u) << l;
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
