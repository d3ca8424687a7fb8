// Micro-benchmark of the serial tail of a Gauss-Newton iteration (one wave, the rest of the CU idle): the 6x6 solve and the SE3 update as
// align_kernels.hip runs them, timed with s_memtime.  Build: hipcc -O3 --offload-arch=gfx950 -I pl-svo_amd/csrc tools/micro/update_bench.hip -o tools/micro/update_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "plsvo_wave.hpp"
using namespace plsvo_hip;

#ifndef VARIANT
#define VARIANT 0
#endif

__device__ __forceinline__ Quat quat_normalized_rsq(const Quat& a) {
  // 1/sqrt(n) from v_rsq_f64 + two Newton steps: y <- y (1.5 - 0.5 n y^2)
  const double n = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
  double y = __builtin_amdgcn_rsq(n);
  const double hn = 0.5 * n;
  y = y * fma(-hn * y, y, 1.5);
  y = y * fma(-hn * y, y, 1.5);
  Quat r = { a.x * y, a.y * y, a.z * y, a.w * y };
  return r;
}

__global__ void k_update(const double* in_tot, const double* in_pose, double* out, unsigned long long* ticks, int reps) {
  __shared__ double s_pose[32];
  __shared__ double s_tot[32];
  __shared__ int s_ctl[16];
  const int lane = threadIdx.x & 63;
  if (lane < 32) { s_pose[lane] = in_pose[lane]; s_tot[lane] = in_tot[lane]; }
  if (lane < 16) s_ctl[lane] = 0;
  __syncthreads();
  unsigned long long t_solve = 0, t_upd = 0;
  for (int rep = 0; rep < reps; ++rep) {
    const double tot = s_tot[lane & 31];
    wave_lds_fence();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    double x[6];
    wave_solve6_reg(tot, x, 320, true);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
      int stop = s_ctl[1];
      if (isnan(x[0])) stop = 1;
      SE3d model = se3_load(s_pose + 12);
      int accepted, brk = 0;
      const double new_chi2 = s_tot[27], old_chi2 = s_pose[26];
      if ((rep < 0 && new_chi2 > old_chi2) || stop) {
        model = se3_load(s_pose + 19);
        accepted = 0; brk = 1;
      } else {
        double mx[6];
        for (int k = 0; k < 6; ++k) mx[k] = -x[k] * 1e-3;
#if VARIANT == 0
        const SE3d nm_ = se3_mul_dev(model, se3_exp_dev(mx));
#else
        SE3d e = se3_exp_dev(mx);
        SE3d nm_; double rt[3];
        quat_rotate(model.q, e.t, rt);
        nm_.t[0] = model.t[0] + rt[0]; nm_.t[1] = model.t[1] + rt[1]; nm_.t[2] = model.t[2] + rt[2];
        nm_.q = quat_normalized_rsq(quat_mul(model.q, e.q));
#endif
        se3_store(model, s_pose + 19);
        model = nm_;
        s_pose[26] = new_chi2;
        accepted = 1;
        if (norm_max6(x) <= 1e-30) brk = 1;
      }
      s_ctl[8] = s_ctl[7];
      s_ctl[7] = (accepted && norm_max6(x) < 1e-3) ? 1 : 0;
      se3_store(model, s_pose + 12);
      quat_to_matrix(model.q, s_pose); s_pose[9] = model.t[0]; s_pose[10] = model.t[1]; s_pose[11] = model.t[2];
      s_ctl[1] = stop; s_ctl[0] = brk;
    }
    wave_lds_fence();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    t_solve += t1 - t0; t_upd += t2 - t1;
  }
  if (lane == 0) { ticks[0] = t_solve; ticks[1] = t_upd; for (int k = 0; k < 12; ++k) out[k] = s_pose[k]; for (int k = 0; k < 7; ++k) out[12 + k] = s_pose[12 + k]; }
}

int main() {
  std::vector<double> tot(32, 0.0), pose(32, 0.0);
  // a well-conditioned SPD system
  const double Hm[6][6] = {{9,1,0.5,0.2,0.1,0.3},{1,8,0.4,0.1,0.2,0.1},{0.5,0.4,7,0.3,0.2,0.1},{0.2,0.1,0.3,6,0.5,0.2},{0.1,0.2,0.2,0.5,5,0.4},{0.3,0.1,0.1,0.2,0.4,4}};
  int k = 0;
  for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) tot[k++] = Hm[i][j];
  for (int i = 0; i < 6; ++i) tot[21 + i] = 0.1 * (i + 1);
  tot[27] = 1.0;
  pose[12 + 3] = 1.0; pose[19 + 3] = 1.0; pose[0] = pose[4] = pose[8] = 1.0; pose[26] = 1e10;
  double *d_tot, *d_pose, *d_out; unsigned long long* d_t;
  hipMalloc(&d_tot, 32 * 8); hipMalloc(&d_pose, 32 * 8); hipMalloc(&d_out, 32 * 8); hipMalloc(&d_t, 16);
  hipMemcpy(d_tot, tot.data(), 32 * 8, hipMemcpyHostToDevice); hipMemcpy(d_pose, pose.data(), 32 * 8, hipMemcpyHostToDevice);
  const int reps = 200;
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k_update, dim3(1), dim3(64), 0, 0, d_tot, d_pose, d_out, d_t, reps);
  hipDeviceSynchronize();
  unsigned long long t[2]; double out[32];
  hipMemcpy(t, d_t, 16, hipMemcpyDeviceToHost); hipMemcpy(out, d_out, 19 * 8, hipMemcpyDeviceToHost);
  printf("variant %d: solve %.0f ticks, update %.0f ticks per iteration; q = %.17g %.17g %.17g %.17g\n", VARIANT, (double)t[0] / reps, (double)t[1] / reps, out[12], out[13], out[14], out[15]);
  return 0;
}
