// pmc_calib.hip -- known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern").
//   calib_stream_read   wide coalesced streaming read, 16 B per lane (what the ref/dx/dy cache reads of align_fused_kernel look like)
//   calib_gather_dword  the current-image gather of align_fused_kernel: per lane 3 image rows x 2 aligned dwords at a pseudo-random
//                       position of a 320x240 level image.  As in the real kernel every WORKGROUP owns its images (one frame =
//                       one workgroup = one XCD's L2: no line is fetched by two L2s), 1024 lanes per image, 8 images per workgroup
//                       one after the other; the host replays the positions and counts the DISTINCT 32 / 64 / 128-byte blocks
//                       touched per image, so the counter can be compared with each granularity
//   calib_stream_write  wide coalesced streaming write, 16 B per lane
// All three sweep buffers far larger than the 256 MB Infinity Cache, so memory-side counters see the traffic.
// Build: hipcc -O3 --offload-arch=gfx950 tools/pmc_calib.hip -o tools/pmc_calib      Run under: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/pmc_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__host__ __device__ inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

constexpr int IMG_W = 320, IMG_H = 240;                 // pyramid level 1 of a 640x480 frame
constexpr size_t IMG_BYTES = (size_t)IMG_W * IMG_H;     // rows tight, as in the pyramid slab

constexpr uint32_t LANES_PER_IMAGE = 1024, IMAGES_PER_BLOCK = 8;   // ~500 patches x 2 lanes of one frame at level 1
__host__ __device__ inline size_t gather_offset(uint32_t img, uint32_t lane_in_image) {
  const uint32_t h2 = mix32(mix32(img * 2654435761u + 12345u) + lane_in_image * 0x9e3779b9u);
  const uint32_t x = 2 + h2 % (IMG_W - 12), y = 2 + (h2 >> 12) % (IMG_H - 8);
  return (size_t)img * IMG_BYTES + (size_t)y * IMG_W + x;
}

__global__ void calib_stream_read(const float4* src, size_t n4, float* sink) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = src[i];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 1234.5678f) *sink = acc;   // never true for the fill pattern: keeps the loads alive
}

__global__ void calib_gather_dword(const uint8_t* img, uint32_t n_images, uint32_t* sink) {
  uint32_t acc = 0;
  for (uint32_t k = 0; k < IMAGES_PER_BLOCK; ++k) {
    const uint32_t image = blockIdx.x * IMAGES_PER_BLOCK + k;
    if (image >= n_images) break;
    for (uint32_t l = threadIdx.x; l < LANES_PER_IMAGE; l += blockDim.x) {
      const size_t off = gather_offset(image, l);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const size_t a = (off + (size_t)r * IMG_W) & ~(size_t)3;
        acc += *reinterpret_cast<const uint32_t*>(img + a) + *reinterpret_cast<const uint32_t*>(img + a + 4);
      }
    }
  }
  if (acc == 0xdeadbeefu) *sink = acc;
}

__global__ void calib_stream_write(float4* dst, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = make_float4((float)i, 1.f, 2.f, 3.f);
}

static size_t distinct_blocks(const std::vector<size_t>& addrs, size_t total_bytes, int shift) {
  std::vector<uint8_t> seen((total_bytes >> shift) / 8 + 2, 0);
  size_t n = 0;
  for (size_t a : addrs) {
    for (size_t b : { a >> shift, (a + 7) >> shift }) {
      if (!(seen[b >> 3] & (1u << (b & 7)))) { seen[b >> 3] |= (uint8_t)(1u << (b & 7)); ++n; }
    }
  }
  return n;
}

int main() {
  const size_t stream_bytes = (size_t)2 << 30;             // 2 GiB streamed
  const uint32_t n_images = 16384;                        // 16384 x 75 KB = 1.2 GB of level images
  const uint32_t n_lanes = n_images * LANES_PER_IMAGE;    // 16 M gather lanes
  const size_t img_bytes = (size_t)n_images * IMG_BYTES + 256;
  float4* buf; uint8_t* img; float* sink; uint32_t* sink2;
  CHECK(hipMalloc(&buf, stream_bytes));
  CHECK(hipMalloc(&img, img_bytes));
  CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&sink2, 4));
  CHECK(hipMemset(buf, 0x3c, stream_bytes));
  CHECK(hipMemset(img, 0x55, img_bytes));
  CHECK(hipDeviceSynchronize());
  const size_t n4 = stream_bytes / 16;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_stream_read, dim3(256 * 16), dim3(256), 0, 0, buf, n4, sink);
    hipLaunchKernelGGL(calib_gather_dword, dim3((n_images + IMAGES_PER_BLOCK - 1) / IMAGES_PER_BLOCK), dim3(128), 0, 0, img, n_images, sink2);
    hipLaunchKernelGGL(calib_stream_write, dim3(256 * 16), dim3(256), 0, 0, buf, n4);
    CHECK(hipDeviceSynchronize());
  }
  // host replay of the gather: the bytes a perfect memory system would fetch at each block size
  std::vector<size_t> addrs;
  addrs.reserve((size_t)n_lanes * 3);
  for (uint32_t image = 0; image < n_images; ++image)
    for (uint32_t l = 0; l < LANES_PER_IMAGE; ++l) {
      const size_t off = gather_offset(image, l);
      for (int r = 0; r < 3; ++r) addrs.push_back((off + (size_t)r * IMG_W) & ~(size_t)3);
    }
  const size_t d32 = distinct_blocks(addrs, img_bytes, 5), d64 = distinct_blocks(addrs, img_bytes, 6), d128 = distinct_blocks(addrs, img_bytes, 7);
  printf("{\"stream_read_bytes\": %zu, \"stream_write_bytes\": %zu, \"gather_lanes\": %u, \"gather_requested_bytes\": %zu, "
         "\"gather_distinct_32B_blocks_bytes\": %zu, \"gather_distinct_64B_blocks_bytes\": %zu, \"gather_distinct_128B_blocks_bytes\": %zu}\n",
         stream_bytes, stream_bytes, n_lanes, (size_t)n_lanes * 24, d32 * 32, d64 * 64, d128 * 128);
  return 0;
}
