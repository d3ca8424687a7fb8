// pyramid_kernels.hip -- 2x2 half-sampling of u8 images on the device (SURVEY 8f "next" #1).
// Replaces vk::halfSample ([ext] vikit/vision.h), called by frame_utils::createImgPyramid
// (src/frame.cpp:171-180).  HBM-bound byte kernel: each lane reads 2 x 8 contiguous bytes and writes
// 4 bytes, fully coalesced; one launch handles one level of a whole batch of slots.
//   rounding 0: vikit's SSE2 path  avg(avg(a,c), avg(b,d)) with rounding averages (x+y+1)>>1
//   rounding 1: vikit's scalar path (a+b+c+d)/4 truncating
#include <hip/hip_runtime.h>

#include "plsvo_dev.hpp"

namespace plsvo_hip {

__device__ __forceinline__ uint32_t half4(uint32_t t0, uint32_t t1, uint32_t b0, uint32_t b1, int rounding) {
  // t0,t1: 8 bytes of the top row; b0,b1: 8 bytes of the bottom row -> 4 output bytes
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t tw = (k < 2) ? t0 : t1, bw = (k < 2) ? b0 : b1;
    const int sh = (k & 1) * 16;
    const uint32_t a = (tw >> sh) & 0xff, b = (tw >> (sh + 8)) & 0xff, c = (bw >> sh) & 0xff, d = (bw >> (sh + 8)) & 0xff;
    uint32_t v;
    if (rounding == 0) { const uint32_t ac = (a + c + 1) >> 1, bd = (b + d + 1) >> 1; v = (ac + bd + 1) >> 1; }
    else v = (a + b + c + d) >> 2;
    out |= v << (8 * k);
  }
  return out;
}

// grid: (ceil(ow/4 * oh / 256), n_slots); src/dst: slot s at base + s*pitch
__global__ __launch_bounds__(256) void halfsample_kernel(const uint8_t* src, size_t src_pitch, int in_w, int in_h, int in_stride,
                                                         uint8_t* dst, size_t dst_pitch, int rounding) {
  const int ow = in_w >> 1, oh = in_h >> 1;
  const int qw = (ow + 3) >> 2;  // quads per output row
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= qw * oh) return;
  const int y = idx / qw, q = idx - y * qw;
  const uint8_t* s = src + (size_t)blockIdx.y * src_pitch;
  uint8_t* d = dst + (size_t)blockIdx.y * dst_pitch;
  const uint8_t* r0 = s + (size_t)(2 * y) * in_stride + 8 * q;
  const uint8_t* r1 = r0 + in_stride;
  uint8_t* o = d + (size_t)y * ow + 4 * q;
  const bool fast = (4 * q + 4 <= ow) && ((in_stride & 7) == 0) && ((reinterpret_cast<uintptr_t>(s) & 7) == 0) &&
                    ((ow & 3) == 0) && ((reinterpret_cast<uintptr_t>(d) & 3) == 0);
  if (fast) {
    const uint2 t = *reinterpret_cast<const uint2*>(r0);
    const uint2 bt = *reinterpret_cast<const uint2*>(r1);
    *reinterpret_cast<uint32_t*>(o) = half4(t.x, t.y, bt.x, bt.y, rounding);
  } else {
    for (int k = 0; k < 4 && 4 * q + k < ow; ++k) {
      const uint32_t a = r0[2 * k], b = r0[2 * k + 1], c = r1[2 * k], dd = r1[2 * k + 1];
      uint32_t v;
      if (rounding == 0) { const uint32_t ac = (a + c + 1) >> 1, bd = (b + dd + 1) >> 1; v = (ac + bd + 1) >> 1; }
      else v = (a + b + c + dd) >> 2;
      o[k] = (uint8_t)v;
    }
  }
}

// strided copy of level 0 into the slab (rows tightened)
__global__ __launch_bounds__(256) void copy_level0_kernel(const uint8_t* src, size_t src_pitch, int w, int h, int stride,
                                                          uint8_t* dst, size_t dst_pitch) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int qw = (w + 15) >> 4;
  if (idx >= qw * h) return;
  const int y = idx / qw, q = idx - y * qw;
  const uint8_t* s = src + (size_t)blockIdx.y * src_pitch + (size_t)y * stride + 16 * q;
  uint8_t* d = dst + (size_t)blockIdx.y * dst_pitch + (size_t)y * w + 16 * q;
  const bool fast = (16 * q + 16 <= w) && ((reinterpret_cast<uintptr_t>(s) & 15) == 0) && ((reinterpret_cast<uintptr_t>(d) & 15) == 0);
  if (fast) *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
  else for (int k = 0; k < 16 && 16 * q + k < w; ++k) d[k] = s[k];
}

// row-major level -> 16 x 8 tiles (plsvo_dev.hpp): one thread per 16-byte tile row; pixels beyond the image edge are written as 0
__global__ __launch_bounds__(256) void tile_level_kernel(const uint8_t* src, size_t src_pitch, int w, int h, uint8_t* dst, size_t dst_pitch) {
  const int tiles_x = (w + 15) >> 4, tiles_y = (h + 7) >> 3;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;      // tile-row index: (tile, row inside the tile)
  if (idx >= tiles_x * tiles_y * 8) return;
  const int tile = idx >> 3, r = idx & 7;
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int y = ty * 8 + r, x0 = tx * 16;
  const uint8_t* s = src + (size_t)blockIdx.y * src_pitch + (size_t)y * w + x0;
  uint8_t* d = dst + (size_t)blockIdx.y * dst_pitch + ((size_t)tile << 7) + (r << 4);
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (y < h) {
    if (x0 + 16 <= w && (reinterpret_cast<uintptr_t>(s) & 15) == 0) v = *reinterpret_cast<const uint4*>(s);
    else {
      uint32_t q[4] = { 0u, 0u, 0u, 0u };
      for (int k = 0; k < 16 && x0 + k < w; ++k) q[k >> 2] |= (uint32_t)s[k] << (8 * (k & 3));
      v = make_uint4(q[0], q[1], q[2], q[3]);
    }
  }
  *reinterpret_cast<uint4*>(d) = v;
}

hipError_t launch_tile_level(const uint8_t* src, size_t src_pitch, int w, int h, uint8_t* dst, size_t dst_pitch, int n_slots, hipStream_t stream) {
  const int work = ((w + 15) >> 4) * ((h + 7) >> 3) * 8;
  if (work <= 0 || n_slots <= 0) return hipSuccess;
  hipLaunchKernelGGL(tile_level_kernel, dim3((work + 255) / 256, n_slots), dim3(256), 0, stream, src, src_pitch, w, h, dst, dst_pitch);
  return hipGetLastError();
}

hipError_t launch_halfsample(const uint8_t* src, size_t src_pitch, int in_w, int in_h, int in_stride, uint8_t* dst,
                             size_t dst_pitch, int n_slots, int rounding, hipStream_t stream) {
  const int ow = in_w >> 1, oh = in_h >> 1;
  const int work = ((ow + 3) >> 2) * oh;
  if (work <= 0 || n_slots <= 0) return hipSuccess;
  hipLaunchKernelGGL(halfsample_kernel, dim3((work + 255) / 256, n_slots), dim3(256), 0, stream, src, src_pitch, in_w, in_h,
                     in_stride, dst, dst_pitch, rounding);
  return hipGetLastError();
}

hipError_t launch_copy_level0(const uint8_t* src, size_t src_pitch, int w, int h, int stride, uint8_t* dst, size_t dst_pitch,
                              int n_slots, hipStream_t stream) {
  const int work = ((w + 15) >> 4) * h;
  if (work <= 0 || n_slots <= 0) return hipSuccess;
  hipLaunchKernelGGL(copy_level0_kernel, dim3((work + 255) / 256, n_slots), dim3(256), 0, stream, src, src_pitch, w, h, stride,
                     dst, dst_pitch);
  return hipGetLastError();
}

}  // namespace plsvo_hip
