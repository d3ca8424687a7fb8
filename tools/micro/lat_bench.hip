// first-touch latency of global loads right after a kernel boundary (one wave on an idle chip): what a lone frame's set-up pays
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void touch(double* p, int n) { for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = p[i] * 1.0000001 + 1.0; }
__global__ void k_lat(const double* a, const int* idx, unsigned long long* out) {
  const int lane = threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const double v0 = a[lane * 3];                       // first touch after the boundary
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const double v1 = a[4096 + lane * 3 + (int)(v0 * 0.0)];    // dependent, another line
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  const double v2 = a[lane * 3 + 1 + (int)(v1 * 0.0)];       // same lines as the first: L1/L2 warm
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t3 = __builtin_amdgcn_s_memtime();
  const int j = idx[lane];                                    // another array
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t4 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = (unsigned long long)(v2 + j); }
}
int main() {
  double* a; int* idx; unsigned long long* out;
  hipMalloc(&a, 1 << 20); hipMalloc(&idx, 4096); hipMalloc(&out, 64);
  hipMemset(a, 0, 1 << 20); hipMemset(idx, 0, 4096);
  for (int rep = 0; rep < 4; ++rep) {
    hipLaunchKernelGGL(touch, dim3(1), dim3(256), 0, 0, a, 1 << 14);   // a previous kernel wrote the data (like the stage / previous step)
    hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, a, idx, out);
    hipDeviceSynchronize();
    unsigned long long h[5]; hipMemcpy(h, out, 40, hipMemcpyDeviceToHost);
    printf("rep %d: first touch %llu ticks, dependent other line %llu, warm line %llu, other array %llu\n", rep, h[0], h[1], h[2], h[3]);
  }
  return 0;
}
