// Probe of gfx950 LDS-DMA semantics through __builtin_amdgcn_global_load_lds (sizes 16 / 12 / 4, lane-linear destination, the immediate offset,
// partially masked waves): what pl-svo_amd/csrc/plsvo_wave.hpp::lds_dma relies on.  hipcc -O3 --offload-arch=gfx950 glds_probe.hip -o glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
__global__ void probe(const unsigned* src, unsigned* out, int n) {
  __shared__ __align__(16) unsigned buf[64 * 4 * 2 + 64 * 3 + 64];
  const int lane = threadIdx.x;
  // size 16: lane l reads src[4*(l*3 % 64) ..] (a permuted source) -> LDS lane-linear
  const unsigned* g16 = src + 4 * ((lane * 3) % 64);
  __builtin_amdgcn_global_load_lds((glb_void*)g16, (lds_void*)&buf[0], 16, 0, 0);
  // size 16 with immediate offset 16 bytes: global address + 16, LDS address + 16?
  __builtin_amdgcn_global_load_lds((glb_void*)g16, (lds_void*)&buf[256], 16, 16, 0);
  // size 12
  const unsigned* g12 = src + 3 * lane;
  __builtin_amdgcn_global_load_lds((glb_void*)g12, (lds_void*)&buf[512], 12, 0, 0);
  // size 4
  const unsigned* g4 = src + (63 - lane);
  __builtin_amdgcn_global_load_lds((glb_void*)g4, (lds_void*)&buf[512 + 192], 4, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int k = lane; k < 512 + 192 + 64; k += 64) out[k] = buf[k];
  // masked: only odd lanes ask; even lanes' LDS words must keep the marker
  __shared__ unsigned mk[64];
  mk[lane] = 7u;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (lane & 1) __builtin_amdgcn_global_load_lds((glb_void*)(src + lane), (lds_void*)&mk[0], 4, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[800 + lane] = mk[lane];
}
int main() {
  std::vector<unsigned> h(1024); for (int i = 0; i < 1024; ++i) h[i] = 1000 + i;
  unsigned *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 4096); hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice); hipMemset(o, 0, 4096);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, 0);
  std::vector<unsigned> r(1024); hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int c = 0; c < 4; ++c) if (r[4 * l + c] != 1000u + 4 * ((l * 3) % 64) + c) ++bad;
  printf("size16 lane-linear: %d mismatches\n", bad);
  printf("size16 offset16: buf[256..263] = %u %u %u %u %u %u %u %u (lane0 src=%u)\n", r[256], r[257], r[258], r[259], r[260], r[261], r[262], r[263], 1000u);
  bad = 0; for (int l = 0; l < 64; ++l) for (int c = 0; c < 3; ++c) if (r[512 + 3 * l + c] != 1000u + 3 * l + c) ++bad;
  printf("size12 lane*12: %d mismatches; words 512.. = %u %u %u %u %u %u %u %u %u %u %u %u (source 1000, 1001, ...)\n", bad, r[512], r[513], r[514], r[515], r[516], r[517], r[518], r[519], r[520], r[521], r[522], r[523]);
  bad = 0; for (int l = 0; l < 64; ++l) if (r[512 + 192 + l] != 1000u + 63 - l) ++bad;
  printf("size4 lane*4: %d mismatches\n", bad);
  bad = 0; for (int l = 0; l < 64; ++l) if (r[800 + l] != ((l & 1) ? 1000u + l : 7u)) ++bad;
  printf("masked (odd lanes only) lane*4 by lane id: %d mismatches (r[800..803] = %u %u %u %u)\n", bad, r[800], r[801], r[802], r[803]);
  return 0;
}
