// Host run of the near-tie resolver of align_kernels.hip -- exact_chi2_pair (planes in HBM, staged through an LDS window) and
// exact_chi2_pair_lds (planes in LDS) -- on the lock-step wave emulator, against the plain sequential float sums they must reproduce
// bit for bit (the reference's chi2 += res*res*weight loop, src/sparse_img_align.cpp:484, :683, :171).  The functions are taken from
// the source file itself: tests/test_exact_sum_host.py cuts the lines from `chain4` to the end of `exact_chi2_pair_lds` into SNIPPET --
// from the tree, and from the tree with tools/patches/slot_parallel_exact_sum_*.patch applied.
// prints "<cases> <mismatches>"
#include <hip/hip_runtime.h>

#include <vector>

#include "plsvo_wave.hpp"
#define PLSVO_LDS
#define PLSVO_GLOBAL
struct plsvo_v4f { float x, y, z, w; };
namespace plsvo_hip {
#include SNIPPET
}
using namespace plsvo_hip;

static uint64_t rng = 0x2545F4914F6CDD1Dull;
static uint32_t rnd() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 32); }
static float unif() { return (float)rnd() / 4294967296.0f; }

static float term(int mode) {
  switch (mode) {
    case 0: { const float r = -6.0f * logf(unif() + 1e-9f) * (rnd() & 1 ? 1.f : -1.f); const float a = fabsf(r); return (r * r) * (1.0f / (1.0f + a)); }   // chi2-like
    case 1: return 0.0f;
    case 2: return ldexpf(1.0f, (int)(rnd() % 12) - 14);                      // powers of two: exact half quanta of a larger running sum
    case 3: return unif() < 0.02f ? ldexpf(1.0f + unif(), 30) : unif();       // jumps over many binades
    case 4: return 0.75f;                                                     // all equal
    default: return ldexpf((float)(rnd() % 4096), -18);                       // short mantissas: many exact ties
  }
}

int main(int argc, char** argv) {
  const int cases = argc > 1 ? atoi(argv[1]) : 60;
  long bad = 0;
  for (int n = 0; n < cases; ++n) {
    const int mode = n % 6;
    const int n_pts = n < 6 ? n : (int)(rnd() % 420), n_seg = (int)(rnd() % 90), scap = 96, iter = 1 + (int)(rnd() % 20);
    const int slots = ((n_pts + 31) & ~31) + 32;
    std::vector<float> A(16 * slots, 0.0f), B(16 * slots, 0.0f), win(1024, 0.0f), lterm(2 * scap + 2, 0.0f);
    std::vector<int> dead(scap, 0);
    for (int i = 0; i < 16 * n_pts; ++i) { A[i] = term(mode); B[i] = mode == 0 && (rnd() & 3) ? A[i] * (1.0f + 1e-3f * (unif() - 0.5f)) : term(mode); }
    for (int s = 0; s < n_seg; ++s) { lterm[s] = term(0); lterm[scap + s] = term(0); dead[s] = (rnd() % 5 == 0) ? 1 + (int)(rnd() % (iter + 2)) : 0; }
    // ---- the reference's order, plain loops: plane A = iteration `iter`, plane B = `iter - 1`
    float want[2];
    for (int h = 0; h < 2; ++h) {
      const std::vector<float>& P = h ? B : A;
      float s = 0.0f;
      for (int i = 0; i < 16 * n_pts; ++i) s = s + P[i];
      float seg = 0.0f;
      for (int sg = 0; sg < n_seg; ++sg) seg = seg + ((dead[sg] == 0 || dead[sg] > iter - h + 1) ? lterm[((iter - h) & 1) * scap + sg] : 0.0f);
      want[h] = s + seg;
    }
    float out_lds[2] = { -1.f, -1.f }, out_hbm[2] = { -1.f, -1.f };
    wave_emu::run_wave([&]() { exact_chi2_pair_lds(A.data(), B.data(), n_pts, n_seg, iter, dead.data(), lterm.data(), scap, out_lds); });
    wave_emu::run_wave([&]() { exact_chi2_pair(A.data(), B.data(), n_pts, n_seg, iter, dead.data(), win.data(), lterm.data(), scap, out_hbm); });
    for (int h = 0; h < 2; ++h) {
      if (memcmp(&out_lds[h], &want[h], 4) != 0) { ++bad; if (bad < 5) fprintf(stderr, "case %d mode %d n_pts %d plane %d: lds %.9g want %.9g\n", n, mode, n_pts, h, out_lds[h], want[h]); }
      if (memcmp(&out_hbm[h], &want[h], 4) != 0) { ++bad; if (bad < 5) fprintf(stderr, "case %d mode %d n_pts %d plane %d: hbm %.9g want %.9g\n", n, mode, n_pts, h, out_hbm[h], want[h]); }
    }
  }
  printf("%d %ld\n", cases, bad);
  return bad != 0;
}
