// Host run of pl-svo_amd/csrc/plsvo_wave.hpp -- the very header the kernels include -- on the lock-step wave emulator
// (tests/host/emu/wave_emu.hpp).  Driven by tests/test_wave_host.py.
//   wave_host_test solve <flavour>   stdin: n x 27 doubles (upper triangle of H row-major, rhs)   stdout: n x 6 doubles (x)
//   wave_host_test solve_reg <flavour>   the same through wave_solve6_reg (inputs in lanes 0..26)
//   wave_host_test solve_search <flavour>  the same with the static pivot order switched off (every step searches its pivot)
//   wave_host_test selfcheck         reductions, scans and the series exp against plain loops; prints "ok" or the first failure
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "plsvo_wave.hpp"
using namespace plsvo_hip;

static std::vector<double> read_all() {
  std::vector<double> v; double buf[1024]; size_t n;
  while ((n = fread(buf, sizeof(double), 1024, stdin)) > 0) v.insert(v.end(), buf, buf + n);
  return v;
}

static int fail(const char* what, int a = 0, int b = 0) { printf("FAIL %s %d %d\n", what, a, b); return 1; }

static int selfcheck() {
  uint64_t rng = 88172645463325252ull;
  auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  // ---- wave_sum_to_lane63 / wave_max_to_lane63: integer-valued doubles, exact
  {
    double in[64], sum63 = 0, max63 = 0;
    double want_sum = 0, want_max = 0;
    for (int l = 0; l < 64; ++l) { in[l] = (double)(rnd() % 100000); want_sum += in[l]; want_max = in[l] > want_max ? in[l] : want_max; }
    wave_emu::run_wave([&]() {
      const int l = threadIdx.x & 63;
      const double s = wave_sum_to_lane63(in[l]), m = wave_max_to_lane63(in[l]);
      if (l == 63) { sum63 = s; max63 = m; }
    });
    if (sum63 != want_sum) return fail("wave_sum_to_lane63");
    if (max63 != want_max) { fprintf(stderr, "max63 %g want %g\n", max63, want_max); return fail("wave_max_to_lane63"); }
  }
  // ---- row_reduce_scatter32 + reduce_rows_finish<ROWS>: a 256-thread workgroup = 4 waves run one after the other, 16 row partials
  {
    constexpr int T = 256, ROWS = T / 16;
    static double v[T][32], s_red[ROWS * 32], want[32], got[64];
    for (int k = 0; k < 32; ++k) want[k] = 0;
    for (int t = 0; t < T; ++t) for (int k = 0; k < 32; ++k) { v[t][k] = (double)(int)(rnd() % 2000001) - 1000000.0; want[k] += v[t][k]; }
    for (int w = 0; w < T / 64; ++w)
      wave_emu::run_wave([&]() {
        const int tid = threadIdx.x, lane = tid & 63;
        double out2[2];
        row_reduce_scatter32(v[tid], out2);
        const int k0 = row_reduce_scatter32_index(lane);
        s_red[(tid >> 4) * 32 + k0] = out2[0]; s_red[(tid >> 4) * 32 + k0 + 1] = out2[1];
      }, 64 * w);
    wave_emu::run_wave([&]() { const int lane = threadIdx.x & 63; got[lane] = reduce_rows_finish<ROWS>(s_red); });
    for (int l = 0; l < 64; ++l) if (got[l] != want[l & 31]) return fail("reduce_rows_finish", l);
  }
  // ---- se3_exp_dev (series for small angles, closed forms above) against plsvo_math.hpp::se3_exp; se3_mul_dev against se3_mul
  for (int n = 0; n < 400; ++n) {
    double u[6];
    const double mag = n < 200 ? 0.05 : 2.0;
    for (int k = 0; k < 6; ++k) u[k] = mag * ((double)(rnd() % 2000001) / 1e6 - 1.0);
    if (n % 50 == 0) u[3] = u[4] = u[5] = 0.0;
    const SE3d a = se3_exp_dev(u), b = se3_exp(u);
    const double da[7] = { a.q.x, a.q.y, a.q.z, a.q.w, a.t[0], a.t[1], a.t[2] }, db[7] = { b.q.x, b.q.y, b.q.z, b.q.w, b.t[0], b.t[1], b.t[2] };
    const double tol = n < 200 ? 1e-15 : 2e-14;   // small angles: series against closed forms; large: both closed forms, written differently
    for (int k = 0; k < 7; ++k) if (!(fabs(da[k] - db[k]) <= tol * (1.0 + fabs(db[k])))) return fail("se3_exp_dev", n, k);
    const SE3d c = se3_mul_dev(a, b), d = se3_mul(a, b);
    const double dc[7] = { c.q.x, c.q.y, c.q.z, c.q.w, c.t[0], c.t[1], c.t[2] }, dd[7] = { d.q.x, d.q.y, d.q.z, d.q.w, d.t[0], d.t[1], d.t[2] };
    for (int k = 0; k < 7; ++k) if (!(fabs(dc[k] - dd[k]) <= 1e-15 * (1.0 + fabs(dd[k])))) return fail("se3_mul_dev", n, k);
  }
  // ---- fast_rcp / fast_div / fast_sqrt: within 1 ulp (here the seed is exact, so this checks the Newton algebra, not the hardware seed)
  for (int n = 0; n < 1000; ++n) {
    const double a = ldexp((double)(rnd() % 1000003) + 1.0, (int)(rnd() % 80) - 40), b = ldexp((double)(rnd() % 999983) + 1.0, (int)(rnd() % 80) - 40);
    if (!(fabs(fast_div(a, b) - a / b) <= 2.3e-16 * (a / b))) return fail("fast_div", n);
    if (!(fabs(fast_sqrt(a) - sqrt(a)) <= 2.3e-16 * sqrt(a))) return fail("fast_sqrt", n);
  }
  if (fast_sqrt(0.0) != 0.0) return fail("fast_sqrt(0)");
  printf("ok\n");
  return 0;
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "selfcheck";
  if (mode == "selfcheck") return selfcheck();
  const int flavour = argc > 2 ? atoi(argv[2]) : 320;
  const std::vector<double> in = read_all();
  const size_t n = in.size() / 27;
  std::vector<double> out(6 * n);
  for (size_t s = 0; s < n; ++s) {
    const double* tot = in.data() + 27 * s;
    double xs[64][6];
    wave_emu::run_wave([&]() {
      const int lane = threadIdx.x & 63;
      double x[6];
      if (mode == "solve_reg") wave_solve6_reg(lane < 27 ? tot[lane] : 0.0, x, flavour);
      else {
        const int i = lane >> 3, j = lane & 7;
        double m = 0.0;
        if (i < 6 && j < 6) m = tot[sym6_index(i, j)];
        else if (i < 6 && j == 6) m = tot[21 + i];
        wave_solve6_core(m, x, flavour, mode != "solve_search");   // solve_search: the per-step pivot search also where the static order applies
      }
      for (int k = 0; k < 6; ++k) xs[lane][k] = x[k];
    });
    for (int l = 1; l < 64; ++l) if (memcmp(xs[l], xs[0], sizeof(xs[0])) != 0) { fprintf(stderr, "x not wave-uniform (system %zu, lane %d)\n", s, l); return 2; }
    memcpy(out.data() + 6 * s, xs[0], sizeof(xs[0]));
  }
  fwrite(out.data(), sizeof(double), out.size(), stdout);
  return 0;
}
