// Stands in for <hip/hip_runtime.h> when the device sources of pl-svo_amd/csrc/ are compiled for the HOST (test infrastructure:
// tests/host/emu/wave_emu.hpp runs a workgroup on cooperative fibres).  Two parts: the device language (qualifiers, threadIdx, vector
// types, intrinsics mapped onto the emulator) and a minimal runtime (device memory = host memory, a stream = immediate execution, a kernel
// launch = the workgroups run one after the other).
#pragma once
#include <chrono>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "../wave_emu.hpp"

#define PLSVO_WAVE_EMU 1
inline void wave_emu_yield_thread() { std::this_thread::yield(); }   // a workgroup polling its peer (two workgroups of a frame run on two OS threads)

// ---- device language -----------------------------------------------------------------------------------------------------------
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __noinline__
#define __launch_bounds__(...)
#define __align__(n) __attribute__((aligned(n)))
#define __shared__ thread_local   // every fibre of a workgroup runs on the same OS thread: one copy per workgroup
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct WaveEmuTid { operator unsigned() const { return wave_emu::S().tid_base + (unsigned)wave_emu::fibre(); } };
struct WaveEmuTid3 { WaveEmuTid x; unsigned y = 0, z = 0; };
static const WaveEmuTid3 threadIdx = {};
namespace wave_emu { inline dim3& block_idx() { static thread_local dim3 v; return v; } inline dim3& block_dim() { static dim3 v; return v; } inline dim3& grid_dim() { static dim3 v; return v; } }
#define blockIdx (wave_emu::block_idx())
#define blockDim (wave_emu::block_dim())
#define gridDim (wave_emu::grid_dim())
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct int2 { int x, y; };
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; };
inline float2 make_float2(float x, float y) { float2 r = { x, y }; return r; }
inline float4 make_float4(float x, float y, float z, float w) { float4 r = { x, y, z, w }; return r; }
inline double2 make_double2(double x, double y) { double2 r = { x, y }; return r; }
inline int2 make_int2(int x, int y) { int2 r = { x, y }; return r; }
inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 r = { x, y }; return r; }
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 r = { x, y, z, w }; return r; }

inline int __double2loint(double v) { uint64_t b; memcpy(&b, &v, 8); return (int)(uint32_t)b; }
inline int __double2hiint(double v) { uint64_t b; memcpy(&b, &v, 8); return (int)(uint32_t)(b >> 32); }
inline double __hiloint2double(int hi, int lo) { const uint64_t b = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double v; memcpy(&v, &b, 8); return v; }
inline double __longlong_as_double(long long i) { double v; memcpy(&v, &i, 8); return v; }
inline long long __double_as_longlong(double v) { long long i; memcpy(&i, &v, 8); return i; }
inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fsqrt_rn(float a) { return sqrtf(a); }
inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline long long min(long long a, long long b) { return a < b ? a : b; }
inline long long max(long long a, long long b) { return a > b ? a : b; }
inline unsigned long min(unsigned long a, unsigned long b) { return a < b ? a : b; }
inline unsigned long max(unsigned long a, unsigned long b) { return a > b ? a : b; }
inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }
inline double min(double a, double b) { return fmin(a, b); }
inline double max(double a, double b) { return fmax(a, b); }
namespace wave_emu {
inline uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (sh & 3))); }
inline uint32_t udot4(uint32_t a, uint32_t b, uint32_t c, bool) {
  for (int k = 0; k < 4; ++k) c += ((a >> (8 * k)) & 0xffu) * ((b >> (8 * k)) & 0xffu);
  return c;
}
}
#define __builtin_amdgcn_update_dpp wave_emu::update_dpp
#define __builtin_amdgcn_readlane wave_emu::readlane
#define __builtin_amdgcn_ds_bpermute wave_emu::ds_bpermute
#define __builtin_amdgcn_wave_barrier() (wave_emu::set_site(__FILE__, __LINE__), wave_emu::wave_barrier())
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_rsq(x) (1.0 / sqrt(x))
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_alignbyte wave_emu::alignbyte
#define __builtin_amdgcn_udot4 wave_emu::udot4
#define __builtin_amdgcn_s_memtime() 0ull
#define __syncthreads() (wave_emu::set_site(__FILE__, __LINE__), wave_emu::syncthreads())
#define __ballot(p) (wave_emu::set_site(__FILE__, __LINE__), wave_emu::ballot((p) != 0))
#define __any(p) (wave_emu::set_site(__FILE__, __LINE__), (int)(wave_emu::ballot((p) != 0) != 0ull))
#define __all(p) (wave_emu::set_site(__FILE__, __LINE__), (int)(wave_emu::ballot((p) == 0) == 0ull))
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
// one fibre runs at a time: a read-modify-write is atomic by construction
template <class T> inline T atomicAdd(T* p, T v) { const T o = *p; *p = o + v; return o; }
template <class T> inline T atomicMin(T* p, T v) { const T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicMax(T* p, T v) { const T o = *p; if (v > o) *p = v; return o; }

// ---- runtime ---------------------------------------------------------------------------------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef void* hipStream_t;
struct WaveEmuEvent { std::chrono::steady_clock::time_point t; };
typedef WaveEmuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipDeviceAttribute_t { hipDeviceAttributeMaxSharedMemoryPerBlock = 1 };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t sharedMemPerBlock; size_t totalGlobalMem; };
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "wave_emu: error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 8; return hipSuccess; }   // (an 8-GPU node: one emulated device per rank of a dry run)
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  snprintf(p->name, sizeof(p->name), "wave_emu host emulation"); snprintf(p->gcnArchName, sizeof(p->gcnArchName), "host");
  p->multiProcessorCount = 256; p->sharedMemPerBlock = 65536; p->totalGlobalMem = (size_t)1 << 34;
  return hipSuccess;
}
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 160 * 1024; return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); if (*p) memset(*p, 0xA5, n); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc(reinterpret_cast<void**>(p), n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc(reinterpret_cast<void**>(p), n, f); }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t = nullptr) {
  for (size_t y = 0; y < h; ++y) memmove((char*)d + y * dp, (const char*)s + y * sp, w);
  return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new WaveEmuEvent(); return hipSuccess; }
#define hipEventDisableTiming 2u
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new WaveEmuEvent(); return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (void*)1; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // (streams execute at once: nothing to wait for)
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
// a launch: the workgroups run one after the other, each on its own set of fibres; dynamic LDS is re-poisoned for every workgroup.
// Workgroups that WAIT FOR EACH OTHER inside the kernel (the two workgroups of a frame, align_kernels.hip: blocks q and q + stride of
// every group of 2 * stride, announced through wave_emu::pair_stride()) run concurrently: the second of a pair on a helper OS thread --
// the emulator's state, blockIdx and LDS are per thread -- so that each finds its partner's granules arriving while it polls.
namespace wave_emu {
void poison_dynamic_lds(size_t bytes);
constexpr size_t DYNAMIC_LDS_BYTES = 160 * 1024;
inline int& pair_stride() { static int v = 0; return v; }
struct PairWorker {
  std::thread th; std::mutex m; std::condition_variable cv; std::function<void()> task; bool has = false, done = true, quit = false;
  PairWorker() {
    th = std::thread([this] {
      for (;;) {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return has || quit; });
        if (quit) return;
        std::function<void()> t = std::move(task);
        has = false;
        l.unlock(); t(); l.lock();
        done = true; cv.notify_all();
      }
    });
  }
  ~PairWorker() { { std::lock_guard<std::mutex> l(m); quit = true; } cv.notify_all(); th.join(); }
  void start(std::function<void()> t) { std::lock_guard<std::mutex> l(m); task = std::move(t); has = true; done = false; cv.notify_all(); }
  void wait() { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return done; }); }
};
inline PairWorker& pair_worker() { static PairWorker w; return w; }
}
template <class K, class... A>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t, A... args) {
  blockDim = block; gridDim = grid;
  const int T = (int)(block.x * block.y * block.z);
  auto one = [&](unsigned bx, unsigned by, unsigned bz) {
    blockIdx = dim3(bx, by, bz);
    if (lds) wave_emu::poison_dynamic_lds(lds);
    wave_emu::run_block(T, [&]() { kernel(args...); });
  };
  const unsigned ps = (unsigned)wave_emu::pair_stride();
  if (ps > 0 && grid.y == 1 && grid.z == 1 && grid.x % (2 * ps) == 0) {
    for (unsigned g = 0; g < grid.x; g += 2 * ps)
      for (unsigned q = 0; q < ps; ++q) {
        wave_emu::pair_worker().start([&, g, q] { one(g + q + ps, 0, 0); });
        one(g + q, 0, 0);
        wave_emu::pair_worker().wait();
      }
    return;
  }
  for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) one(bx, by, bz);
}
