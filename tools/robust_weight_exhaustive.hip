// robust_weight_exhaustive.hip -- every float a in [0, 256] (1 132 462 081 bit patterns) through the alignment kernel's robust weight
// (pl-svo_amd/csrc/robust_weight.hpp) against the device's IEEE double division, (float)(1.0 / (1.0 + (double)a)): what the reference
// computes at src/sparse_img_align.cpp:479.  The host then re-checks a stride of the same inputs against its own IEEE division, so that
// the device division itself is pinned.  Prints one JSON line; exit code 1 if robust_weight_f64 (the form the kernel uses) differs anywhere.
// Build: hipcc -O3 --offload-arch=gfx950 tools/robust_weight_exhaustive.hip -o tools/robust_weight_exhaustive
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../pl-svo_amd/csrc/robust_weight.hpp"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 2; } } while (0)

__global__ void sweep(uint32_t first, uint32_t count, unsigned long long* bad64, unsigned long long* bad32, uint32_t* first_bad64, float* sample, uint32_t sample_stride) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const uint32_t u = first + i;
  const float a = __uint_as_float(u);
  const float ref = (float)(1.0 / (1.0 + (double)a));
  const float w64 = plsvo_hip::robust_weight_f64(a), w32 = plsvo_hip::robust_weight_f32(a);
  if (__float_as_uint(w64) != __float_as_uint(ref)) { atomicAdd(bad64, 1ull); atomicMin(first_bad64, u); }
  if (__float_as_uint(w32) != __float_as_uint(ref)) atomicAdd(bad32, 1ull);
  if (u % sample_stride == 0) sample[u / sample_stride] = ref;
}

int main() {
  float top = 256.0f; uint32_t hi; memcpy(&hi, &top, 4);
  const uint32_t stride = 4099;   // prime: the host re-checks every 4099th input against its own division
  unsigned long long *d_bad64, *d_bad32; uint32_t* d_first; float* d_sample;
  const size_t n_sample = hi / stride + 1;
  CHECK(hipMalloc(&d_bad64, 8)); CHECK(hipMalloc(&d_bad32, 8)); CHECK(hipMalloc(&d_first, 4)); CHECK(hipMalloc(&d_sample, n_sample * 4));
  CHECK(hipMemset(d_bad64, 0, 8)); CHECK(hipMemset(d_bad32, 0, 8)); CHECK(hipMemset(d_first, 0xff, 4));
  const uint32_t chunk = 1u << 26;
  for (uint64_t first = 0; first <= hi; first += chunk) {
    const uint32_t count = (uint32_t)((hi - first + 1 < chunk) ? hi - first + 1 : chunk);
    hipLaunchKernelGGL(sweep, dim3((count + 255) / 256), dim3(256), 0, 0, (uint32_t)first, count, d_bad64, d_bad32, d_first, d_sample, stride);
    CHECK(hipGetLastError());
  }
  CHECK(hipDeviceSynchronize());
  unsigned long long bad64 = 0, bad32 = 0; uint32_t first_bad = 0;
  std::vector<float> sample(n_sample);
  CHECK(hipMemcpy(&bad64, d_bad64, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&bad32, d_bad32, 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&first_bad, d_first, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(sample.data(), d_sample, n_sample * 4, hipMemcpyDeviceToHost));
  unsigned long long host_bad = 0;
  for (size_t k = 0; k < n_sample; ++k) {
    const uint32_t u = (uint32_t)(k * stride); float a; memcpy(&a, &u, 4);
    const volatile double d = 1.0 + (double)a;
    const float ref = (float)(1.0 / d);
    if (memcmp(&ref, &sample[k], 4) != 0) ++host_bad;
  }
  printf("{\"inputs\": %u, \"robust_weight_f64_mismatches\": %llu, \"first_f64_mismatch_bits\": \"0x%08x\", \"robust_weight_f32_mismatches\": %llu, "
         "\"device_division_vs_host_division_mismatches\": %llu, \"host_checked\": %zu}\n", hi + 1, bad64, first_bad, bad32, host_bad, n_sample);
  return (bad64 != 0 || host_bad != 0) ? 1 : 0;
}
