// plsvo_capi.hip -- implementation of the C ABI declared in include/plsvo_hip.h.
// Host side only: context, HBM buffers, staging of flattened features, kernel launches, result fetch,
// hipEvent timing, RCCL gather.  All device work is enqueued on the ctx stream; nothing here computes
// any part of the hot path on the CPU (there is no fallback).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/plsvo_hip.h"
#include "plsvo_dev.hpp"
#include "plsvo_math.hpp"

namespace plsvo_hip {
// kernels (align_kernels.hip, poseopt_kernels.hip, pyramid_kernels.hip)
size_t align_level_lds_bytes(int threads, int cap, int scap, int chi_lds_pts);
hipError_t launch_align_reorder(const int* work_key, int n, int* order_out, int shift, hipStream_t stream);   // (the pose optimiser's batches use it too)
hipError_t launch_align_levels(const AlignBatchDev& b, int cap, int scap, int level_hi, int level_lo, int do_init, int threads, size_t lds,
                               hipStream_t stream);
hipError_t launch_pose_opt(const PoseBatchDev& b, double* d_poses, int threads, hipStream_t stream);
hipError_t launch_structopt(const StructBatchDev& s, hipStream_t stream);
hipError_t launch_match_direct(const MatchBatchDev& b, hipStream_t stream);
hipError_t launch_reproject(const ReprojBatchDev& b, hipStream_t stream);
hipError_t launch_update_seeds(const SeedsBatchDev& b, hipStream_t stream);
hipError_t launch_halfsample(const uint8_t* src, size_t src_pitch, int in_w, int in_h, int in_stride, uint8_t* dst,
                             size_t dst_pitch, int n_slots, int rounding, hipStream_t stream);
hipError_t launch_copy_level0(const uint8_t* src, size_t src_pitch, int w, int h, int stride, uint8_t* dst, size_t dst_pitch,
                              int n_slots, hipStream_t stream);
hipError_t launch_chain_pose(const ChainBatchDev& b, hipStream_t stream);
hipError_t launch_pack_pose_records(const AlignStateDev* ast, const PoseStateDev* pst, int n, plsvo_pose_record* dst, hipStream_t stream);
hipError_t launch_chain_active(const ChainBatchDev& b, hipStream_t stream);
hipError_t launch_chain_select(const ChainBatchDev& b, hipStream_t stream);
hipError_t launch_tile_level(const uint8_t* src, size_t src_pitch, int w, int h, uint8_t* dst, size_t dst_pitch, int n_slots, hipStream_t stream);
}  // namespace plsvo_hip

using namespace plsvo_hip;

static thread_local std::string g_create_error;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
    size_t want = std::max(bytes, (size_t)256);
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct EventPair { hipEvent_t a, b; };

// All input arrays of a staged batch travel as ONE host-to-device copy: sections of a single blob, each 256-byte aligned.
// (A single-frame call used to pay a dozen small pageable copies; small blobs go through a pinned staging buffer.)
struct Blob {
  std::vector<uint8_t> host;
  template <typename T> size_t add(const std::vector<T>& v) {
    const size_t off = (host.size() + 255) & ~(size_t)255;
    host.resize(off + std::max(v.size() * sizeof(T), (size_t)16));
    if (!v.empty()) memcpy(host.data() + off, v.data(), v.size() * sizeof(T));
    return off;
  }
};

}  // namespace

struct plsvo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  int cu_count = 0;
  size_t lds_per_block = 65536;
  // experiment switches, read from the environment ONCE, at plsvo_hip_create (never on a launch path)
  int env_align_threads = 0, env_align_lds_pad = 0, env_poseopt_threads = 0;
  bool env_align_per_level = false, env_align_no_lpt = false, env_host_timing = false;
  int ldlt_flavour = 320;   // plsvo_hip_set_option(PLSVO_OPT_LDLT_FLAVOUR)

  // pyramids (row-major slab: half-sampler, matcher, depth filter, download) and their tiled mirror (alignment kernel)
  DevBuf pyr_tiled;
  DevBuf pyr_slab;
  PyrDesc pyr{};
  DevBuf pyr_upload;  // staging for level-0 uploads

  // alignment batch
  int a_n = 0;
  bool a_staged = false;
  std::vector<AlignJobDev> a_jobs;
  std::vector<int> a_nseg_off;  // per job seg offset (host copy)
  int a_total_seg = 0;
  int a_gmax = -1, a_gmin = 99;
  int a_cap[PLSVO_MAX_LEVELS]{};
  int a_scap = 4;   // max segments of one job
  int a_trace_cap = 0;
  DevBuf a_d_state, a_d_alive;   // (inputs: a_d_blob)
  DevBuf a_d_pxyz, a_d_puv, a_d_cref, a_d_chi, a_d_log, a_d_poses;
  size_t a_patch_total = 0;                 // patch slots of the staged batch (all jobs, all levels' maximum)
  int a_seg_align = 32;                     // the staged layout's segment alignment (64: two workgroups per frame are possible)
  DevBuf a_d_workkey;                       // per job: the patch-iterations its last launch evaluated
  DevBuf a_d_order[2];                      // launch order of a RE-RUN resident batch: sorted on the device by the last launch's measured work
  int a_order_next = 0;                     //   (align_kernels.hip::align_reorder_kernel); the buffer the next reorder writes
  bool env_align_no_reorder = false;
  const int* a_stage_order = nullptr;          // the staged batch's own order (most patches first), inside a_d_blob
  bool a_one_shot = false, p_one_shot = false;   // set around the run of a one-shot batch call: its launch order is never consumed
  int env_align_reorder_min = 0;            //   PLSVO_ALIGN_REORDER_MIN: smallest batch that is re-ordered (tests; default 16 frames per CU)
  DevBuf a_d_xbuf;                          // two workgroups per frame: their exchange granules (2 KB per frame, zeroed once)
  unsigned int x_launch = 0;                // launches that used it (tags = launch << 10 | exchange: never repeated)
  bool env_align_no_pair = false;
  AlignBatchDev a_b{};

  // pose-opt batch
  int p_n = 0;
  bool p_staged = false;
  std::vector<PoseJobDev> p_jobs;
  int p_total_pt = 0, p_total_seg = 0;
  int p_trace_cap = 0;
  DevBuf p_d_state, p_d_ptkeep, p_d_segkeep;   // (inputs: p_d_blob)
  DevBuf p_d_s32, p_d_s64, p_d_log, p_d_poses;
  PoseBatchDev p_b{};
  DevBuf p_d_workkey, p_d_order[2];         // as a_d_workkey / a_d_order: the launch order of a re-run staged batch, from its last launch's feature-iterations
  int p_order_next = 0, p_key_shift = 0;
  bool env_poseopt_no_reorder = false;
  int env_poseopt_reorder_min = 0;          //   PLSVO_POSEOPT_REORDER_MIN (tests)

  // resident frame step (plsvo_chain_*): candidates, glue state, pose-optimiser input written on the device
  bool ch_staged = false;
  int ch_n = 0, ch_ncand = 0, ch_npt_cap = 0, ch_nseg_cap = 0;
  std::vector<ChainJobDev> ch_jobs;
  DevBuf ch_d_blob, ch_d_work, ch_d_po;
  ChainBatchDev ch_b{};
  MatchBatchDev ch_match{};
  ReprojBatchDev ch_reproj{};
  PoseBatchDev ch_pose{};
  DevBuf ch_d_state, ch_d_ptkeep, ch_d_segkeep, ch_d_s32, ch_d_s64, ch_d_poses;
  DevBuf rec_d;   // plsvo_fetch_pose_records
  unsigned long long run_seq = 0, a_run_seq = 0, p_run_seq = 0, ch_run_seq = 0;   // which resident batch ran last, 0 = not since it was staged (plsvo_pack_pose_records)

  // structure optimisation (one-shot batches)
  DevBuf s_d_in, s_d_out;

  // staging: one blob per batch type (Blob), pinned bounce buffer for small ones
  DevBuf a_d_blob, p_d_blob;
  // pinned bounce buffers of the stage calls: two, used in turn, each guarded by an event recorded behind its last DMA
  void* pinned[2] = { nullptr, nullptr };
  size_t pinned_cap[2] = { 0, 0 };
  hipEvent_t pinned_done[2] = { nullptr, nullptr };
  int pinned_next = 0;
  // the same for whole pyramids (plsvo_hip_upload_pyramid packs a frame's levels into one pinned image of its slot: ONE DMA, no wait)
  void* pyr_pinned[2] = { nullptr, nullptr };
  size_t pyr_pinned_cap[2] = { 0, 0 };
  hipEvent_t pyr_pinned_done[2] = { nullptr, nullptr };
  int pyr_pinned_next = 0;
  // slots whose TILED mirror is stale: an uploaded pyramid is re-tiled only when a launch that reads the mirror (the one-wave-per-frame
  // shape of the alignment) is about to use it -- a per-frame caller never pays the four tile launches
  std::vector<uint8_t> tiled_stale;
  int tiled_stale_count = 0;
  // results of a fetch come back through a pinned buffer (device-to-PAGEABLE copies are staged and waited for one by one by the driver)
  void* dl_pinned = nullptr;
  size_t dl_pinned_cap = 0;

  // profiling
  bool profiling = false;
  std::vector<EventPair> ev[PLSVO_K_COUNT];
  std::vector<EventPair> ev_pool;
  double ev_ms[PLSVO_K_COUNT]{};
  int64_t ev_launches[PLSVO_K_COUNT]{};
};

#define CTX_CHECK(ctx) do { if (!(ctx)) return PLSVO_E_INVALID; } while (0)
#define HIP_TRY(ctx, expr)                                                                      \
  do {                                                                                          \
    hipError_t e__ = (expr);                                                                    \
    if (e__ != hipSuccess) {                                                                    \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                          \
      return PLSVO_E_HIP;                                                                       \
    }                                                                                           \
  } while (0)
static int fail(plsvo_ctx* ctx, int code, const std::string& msg) { ctx->err = msg; return code; }

// ---- profiling helpers ----------------------------------------------------------------------
static void prof_begin(plsvo_ctx* c, int k, EventPair* ep) {
  if (!c->profiling) return;
  if (!c->ev_pool.empty()) { *ep = c->ev_pool.back(); c->ev_pool.pop_back(); }
  else { (void)hipEventCreate(&ep->a); (void)hipEventCreate(&ep->b); }
  (void)hipEventRecord(ep->a, c->stream);
  (void)k;
}
static void prof_end(plsvo_ctx* c, int k, EventPair* ep) {
  if (!c->profiling) return;
  (void)hipEventRecord(ep->b, c->stream);
  c->ev[k].push_back(*ep);
}
static void prof_collect(plsvo_ctx* c) {
  for (int k = 0; k < PLSVO_K_COUNT; ++k) {
    for (auto& ep : c->ev[k]) {
      float ms = 0.f;
      if (hipEventSynchronize(ep.b) == hipSuccess && hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) {
        c->ev_ms[k] += (double)ms; c->ev_launches[k] += 1;
      }
      c->ev_pool.push_back(ep);
    }
    c->ev[k].clear();
  }
}

// ---- context ---------------------------------------------------------------------------------
extern "C" const char* plsvo_hip_version(void) { return "plsvo_hip 0.1 (gfx950)"; }

static int create_ctx(int device_id, void* stream, bool use_given_stream, plsvo_ctx** out);
static bool env_flag(const char* name) { const char* s = getenv(name); return s && *s && strcmp(s, "0") != 0; }   // set and not "0"
extern "C" int plsvo_hip_create(int device_id, void* stream, plsvo_ctx** out) { return create_ctx(device_id, stream, stream != nullptr, out); }
extern "C" int plsvo_hip_create_on_stream(int device_id, void* stream, plsvo_ctx** out) { return create_ctx(device_id, stream, true, out); }

static int create_ctx(int device_id, void* stream, bool use_given_stream, plsvo_ctx** out) {
  if (!out) return PLSVO_E_INVALID;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "count=0") +
                     "); plsvo_hip has no CPU fallback";
    return PLSVO_E_NODEVICE;
  }
  if (device_id < 0 || device_id >= n) { g_create_error = "device_id out of range"; return PLSVO_E_INVALID; }
  if ((e = hipSetDevice(device_id)) != hipSuccess) { g_create_error = hipGetErrorString(e); return PLSVO_E_HIP; }
  plsvo_ctx* c = new plsvo_ctx();
  c->device = device_id;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
    c->cu_count = prop.multiProcessorCount;
    c->lds_per_block = prop.sharedMemPerBlock;
    int optin = 0;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id) == hipSuccess && optin > 0)
      c->lds_per_block = std::max(c->lds_per_block, (size_t)optin);
  }
  if (use_given_stream) { c->stream = reinterpret_cast<hipStream_t>(stream); c->own_stream = false; }   // NULL = HIP's default stream
  else {
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
      g_create_error = hipGetErrorString(e); delete c; return PLSVO_E_HIP;
    }
    c->own_stream = true;
  }
  if (const char* s = getenv("PLSVO_LDS_LIMIT")) c->lds_per_block = (size_t)atol(s);
  if (const char* s = getenv("PLSVO_ALIGN_THREADS")) { const int v = atoi(s); if (v == 64 || v == 128 || v == 256 || v == 512) c->env_align_threads = v; }
  if (const char* s = getenv("PLSVO_ALIGN_LDS_PAD")) c->env_align_lds_pad = std::max(0, atoi(s));
  if (const char* s = getenv("PLSVO_POSEOPT_THREADS")) { const int v = atoi(s); if (v == 16 || v == 64 || v == 256 || v == 512) c->env_poseopt_threads = v; }
  if (const char* s = getenv("PLSVO_ALIGN_PER_LEVEL")) c->env_align_per_level = atoi(s) != 0;
  c->env_align_no_lpt = env_flag("PLSVO_ALIGN_NO_LPT");
  c->env_align_no_pair = env_flag("PLSVO_ALIGN_NO_PAIR");   // (A/B: one workgroup per frame also for small batches)
  c->env_align_no_reorder = env_flag("PLSVO_ALIGN_NO_REORDER");   // (A/B: keep the stage call's patch-count order for every launch)
  if (const char* s = getenv("PLSVO_ALIGN_REORDER_MIN")) c->env_align_reorder_min = atoi(s);
  c->env_poseopt_no_reorder = env_flag("PLSVO_POSEOPT_NO_REORDER");
  if (const char* s = getenv("PLSVO_POSEOPT_REORDER_MIN")) c->env_poseopt_reorder_min = atoi(s);
  c->env_host_timing = getenv("PLSVO_HOST_TIMING") != nullptr;
  *out = c;
  return PLSVO_OK;
}

extern "C" void plsvo_hip_destroy(plsvo_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  prof_collect(c);
  for (auto& ep : c->ev_pool) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
  c->a_d_xbuf.release(); c->a_d_order[0].release(); c->a_d_order[1].release(); c->a_d_workkey.release(); c->p_d_workkey.release(); c->p_d_order[0].release(); c->p_d_order[1].release();
  DevBuf* bufs[] = { &c->pyr_slab, &c->pyr_tiled, &c->pyr_upload, &c->a_d_blob, &c->a_d_state, &c->a_d_alive, &c->a_d_pxyz, &c->a_d_puv, &c->a_d_cref,
                     &c->a_d_chi, &c->a_d_log, &c->a_d_poses, &c->p_d_blob, &c->p_d_state, &c->p_d_ptkeep, &c->p_d_segkeep, &c->p_d_s32, &c->p_d_s64,
                     &c->p_d_log, &c->p_d_poses, &c->s_d_in, &c->s_d_out, &c->ch_d_blob, &c->ch_d_work, &c->ch_d_po, &c->ch_d_state,
                     &c->ch_d_ptkeep, &c->ch_d_segkeep, &c->ch_d_s32, &c->ch_d_s64, &c->ch_d_poses, &c->rec_d };
  for (DevBuf* b : bufs) b->release();
  for (int k = 0; k < 2; ++k) { if (c->pinned[k]) (void)hipHostFree(c->pinned[k]); if (c->pinned_done[k]) (void)hipEventDestroy(c->pinned_done[k]); }
  for (int k = 0; k < 2; ++k) { if (c->pyr_pinned[k]) (void)hipHostFree(c->pyr_pinned[k]); if (c->pyr_pinned_done[k]) (void)hipEventDestroy(c->pyr_pinned_done[k]); }
  if (c->dl_pinned) (void)hipHostFree(c->dl_pinned);
  if (c->own_stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" int plsvo_hip_set_option(plsvo_ctx* c, int option, int value) {
  CTX_CHECK(c);
  if (option == PLSVO_OPT_LDLT_FLAVOUR) {
    if (value != 320 && value != 330) return fail(c, PLSVO_E_INVALID, "set_option: the LDLT flavour is 320 (Eigen 3.1 ... 3.2.1) or 330 (Eigen 3.2.2 and later)");
    c->ldlt_flavour = value;
    c->a_staged = false; c->p_staged = false;   // the rule travels with the staged jobs
    return PLSVO_OK;
  }
  if (option == PLSVO_OPT_ALIGN_THREADS) {
    if (value != 0 && value != 64 && value != 128 && value != 256 && value != 512) return fail(c, PLSVO_E_INVALID, "set_option: alignment launch shape is 0 (automatic), 64, 128, 256 or 512 threads per frame");
    c->env_align_threads = value;
    return PLSVO_OK;
  }
  if (option == PLSVO_OPT_POSEOPT_THREADS) {
    if (value != 0 && value != 16 && value != 64 && value != 256 && value != 512) return fail(c, PLSVO_E_INVALID, "set_option: pose-optimiser launch shape is 0 (automatic), 16 (a 16-lane row per frame, four frames per wave), 64, 256 or 512 threads per frame");
    c->env_poseopt_threads = value;
    return PLSVO_OK;
  }
  if (option == PLSVO_OPT_ALIGN_REORDER || option == PLSVO_OPT_POSEOPT_REORDER) {
    if (value != 0 && value != 1) return fail(c, PLSVO_E_INVALID, "set_option: the launch-order refresh is 0 (keep the stage order) or 1 (refresh from the last launch's measured work)");
    if (option == PLSVO_OPT_ALIGN_REORDER) {
      c->env_align_no_reorder = value == 0;
      if (value == 0 && c->a_staged && c->a_d_blob.p) c->a_b.order = c->a_stage_order;   // back to the stage call's order
    } else {
      c->env_poseopt_no_reorder = value == 0;
    }
    return PLSVO_OK;
  }
  return fail(c, PLSVO_E_INVALID, "set_option: unknown option");
}

extern "C" const char* plsvo_hip_last_error(const plsvo_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }
extern "C" void* plsvo_hip_stream(plsvo_ctx* c) { return c ? reinterpret_cast<void*>(c->stream) : nullptr; }
extern "C" int plsvo_hip_synchronize(plsvo_ctx* c) { CTX_CHECK(c); HIP_TRY(c, hipStreamSynchronize(c->stream)); return PLSVO_OK; }

extern "C" int plsvo_hip_device_info(plsvo_ctx* c, char* name, int name_len, int* cu_count, size_t* hbm_bytes) {
  CTX_CHECK(c);
  hipDeviceProp_t prop;
  HIP_TRY(c, hipGetDeviceProperties(&prop, c->device));
  if (name && name_len > 0) { snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName); }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return PLSVO_OK;
}

// ---- pyramids ----------------------------------------------------------------------------------
extern "C" int plsvo_hip_config_pyramids(plsvo_ctx* c, int n_slots, int width, int height, int n_levels) {
  CTX_CHECK(c);
  if (n_slots <= 0 || width <= 0 || height <= 0 || n_levels <= 0 || n_levels > PLSVO_MAX_LEVELS) return fail(c, PLSVO_E_INVALID, "config_pyramids: bad arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  PyrDesc d{};
  size_t off = 0;
  for (int l = 0; l < n_levels; ++l) {
    d.w[l] = width >> l; d.h[l] = height >> l;
    if (d.w[l] <= 0 || d.h[l] <= 0) return fail(c, PLSVO_E_INVALID, "config_pyramids: too many levels for this image size");
    d.off[l] = pyr_level_offset(width, height, l);
    off = (size_t)pyr_level_offset(width, height, l + 1);           // >= 64 bytes of slack after every level (gather over-read)
  }
  d.slot_bytes = off; d.n_slots = n_slots; d.n_levels = n_levels;
  d.tslot_bytes = pyr_tiled_level_offset(width, height, n_levels);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, c->pyr_slab.ensure((size_t)n_slots * off + 256));
  HIP_TRY(c, hipMemsetAsync(c->pyr_slab.p, 0, (size_t)n_slots * off + 256, c->stream));
  HIP_TRY(c, c->pyr_tiled.ensure((size_t)n_slots * d.tslot_bytes + 256));
  HIP_TRY(c, hipMemsetAsync(c->pyr_tiled.p, 0, (size_t)n_slots * d.tslot_bytes + 256, c->stream));
  d.base = c->pyr_slab.as<uint8_t>();
  d.tbase = c->pyr_tiled.as<uint8_t>();
  c->pyr = d;
  c->tiled_stale.assign((size_t)n_slots, 0); c->tiled_stale_count = 0;
  c->a_staged = false;
  c->ch_staged = false;
  return PLSVO_OK;
}

static int check_slot(plsvo_ctx* c, int slot) {
  if (!c->pyr.base) return fail(c, PLSVO_E_STATE, "pyramids not configured");
  if (slot < 0 || slot >= c->pyr.n_slots) return fail(c, PLSVO_E_CAPACITY, "pyramid slot out of range");
  return PLSVO_OK;
}

// refresh the tiled mirror of slots [first_slot, first_slot + n), levels [0, n_levels): enqueued after whatever wrote the slab
static int retile(plsvo_ctx* c, int first_slot, int n, int n_levels) {
  const uint8_t* src = c->pyr_slab.as<uint8_t>() + (size_t)first_slot * c->pyr.slot_bytes;
  uint8_t* dst = c->pyr_tiled.as<uint8_t>() + (size_t)first_slot * c->pyr.tslot_bytes;
  for (int l = 0; l < n_levels; ++l)
    HIP_TRY(c, launch_tile_level(src + c->pyr.off[l], c->pyr.slot_bytes, c->pyr.w[l], c->pyr.h[l],
                                 dst + pyr_tiled_level_offset(c->pyr.w[0], c->pyr.h[0], l), c->pyr.tslot_bytes, n, c->stream));
  return PLSVO_OK;
}

// the tiled mirror of every slot in [first, first + n) is current again (whoever calls this has just enqueued retile() for them)
static void mark_tiled_fresh(plsvo_ctx* c, int first, int n) {
  if (c->tiled_stale_count == 0) return;
  for (int s = first; s < first + n && s < (int)c->tiled_stale.size(); ++s)
    if (c->tiled_stale[(size_t)s]) { c->tiled_stale[(size_t)s] = 0; --c->tiled_stale_count; }
}
// re-tile the stale slots among `slots` (a launch that reads the mirror is about to be enqueued behind this)
static int retile_stale(plsvo_ctx* c, const std::vector<int>& slots) {
  if (c->tiled_stale_count == 0) return PLSVO_OK;
  for (int sl : slots) {
    if (sl < 0 || sl >= (int)c->tiled_stale.size() || !c->tiled_stale[(size_t)sl]) continue;
    int rc = retile(c, sl, 1, c->pyr.n_levels); if (rc) return rc;
    c->tiled_stale[(size_t)sl] = 0; --c->tiled_stale_count;
    if (c->tiled_stale_count == 0) break;
  }
  return PLSVO_OK;
}

extern "C" int plsvo_hip_upload_pyramid(plsvo_ctx* c, int slot, int n_levels, const uint8_t* const* level_ptr, const int* width,
                                        const int* height, const int* stride_bytes) {
  CTX_CHECK(c);
  int rc = check_slot(c, slot); if (rc) return rc;
  if (!level_ptr || !width || !height || !stride_bytes || n_levels <= 0 || n_levels > c->pyr.n_levels) return fail(c, PLSVO_E_INVALID, "upload_pyramid: bad arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  for (int l = 0; l < n_levels; ++l) {
    if (!level_ptr[l] || width[l] != c->pyr.w[l] || height[l] != c->pyr.h[l] || stride_bytes[l] < width[l])
      return fail(c, PLSVO_E_INVALID, "upload_pyramid: level size does not match the configured pyramid");
  }
  uint8_t* const slot_dst = c->pyr_slab.as<uint8_t>() + (size_t)slot * c->pyr.slot_bytes;
  // The levels are packed into a pinned image of the slot (rows tight, every level at its offset) and cross PCIe as ONE copy enqueued on
  // the stream: the caller's buffers are free when the call returns, nothing is waited for, and whatever is launched next on the stream
  // reads the new pyramid.  (Until round 4: one pageable 2-D copy per level -- each staged by the driver -- four tile launches and a stream
  // synchronisation, ~45 us of a 0.4 ms SparseImgAlign::run.)  Two pinned images, each guarded by an event behind its last DMA.
  const size_t image_bytes = (size_t)c->pyr.off[n_levels - 1] + (size_t)width[n_levels - 1] * (size_t)height[n_levels - 1];
  const int k = c->pyr_pinned_next;
  if (!c->pyr_pinned_done[k] && hipEventCreate(&c->pyr_pinned_done[k]) != hipSuccess) c->pyr_pinned_done[k] = nullptr;
  bool packed = false;
  if (c->pyr_pinned_done[k]) {
    HIP_TRY(c, hipEventSynchronize(c->pyr_pinned_done[k]));   // the DMA that last read this image has landed (no-op when never recorded)
    if (c->pyr_pinned_cap[k] < image_bytes) {
      if (c->pyr_pinned[k]) (void)hipHostFree(c->pyr_pinned[k]);
      c->pyr_pinned[k] = nullptr; c->pyr_pinned_cap[k] = 0;
      if (hipHostMalloc(&c->pyr_pinned[k], (size_t)c->pyr.slot_bytes, hipHostMallocDefault) == hipSuccess) {
        c->pyr_pinned_cap[k] = (size_t)c->pyr.slot_bytes;
        memset(c->pyr_pinned[k], 0, c->pyr_pinned_cap[k]);   // (the slack between the levels travels with them)
      }
    }
    if (c->pyr_pinned[k]) {
      uint8_t* const img = static_cast<uint8_t*>(c->pyr_pinned[k]);
      for (int l = 0; l < n_levels; ++l) {
        uint8_t* d = img + c->pyr.off[l];
        const size_t w = (size_t)width[l];
        if ((size_t)stride_bytes[l] == w) memcpy(d, level_ptr[l], w * (size_t)height[l]);
        else for (int y = 0; y < height[l]; ++y) memcpy(d + (size_t)y * w, level_ptr[l] + (size_t)y * (size_t)stride_bytes[l], w);
      }
      HIP_TRY(c, hipMemcpyAsync(slot_dst, img, image_bytes, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(c, hipEventRecord(c->pyr_pinned_done[k], c->stream));
      c->pyr_pinned_next = k ^ 1;
      packed = true;
    }
  }
  if (!packed) {   // no pinned memory: straight from the caller's buffers, which must stay untouched until the copies have landed
    for (int l = 0; l < n_levels; ++l)
      HIP_TRY(c, hipMemcpy2DAsync(slot_dst + c->pyr.off[l], (size_t)width[l], level_ptr[l], (size_t)stride_bytes[l], (size_t)width[l], (size_t)height[l],
                                  hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (!c->tiled_stale[(size_t)slot]) { c->tiled_stale[(size_t)slot] = 1; ++c->tiled_stale_count; }
  return PLSVO_OK;
}

static int build_levels(plsvo_ctx* c, int first_slot, int n, int rounding) {
  uint8_t* base = c->pyr_slab.as<uint8_t>() + (size_t)first_slot * c->pyr.slot_bytes;
  for (int l = 1; l < c->pyr.n_levels; ++l) {
    EventPair ep{}; prof_begin(c, PLSVO_K_HALFSAMPLE, &ep);
    HIP_TRY(c, launch_halfsample(base + c->pyr.off[l - 1], c->pyr.slot_bytes, c->pyr.w[l - 1], c->pyr.h[l - 1], c->pyr.w[l - 1],
                                 base + c->pyr.off[l], c->pyr.slot_bytes, n, rounding, c->stream));
    prof_end(c, PLSVO_K_HALFSAMPLE, &ep);
  }
  // (the slots count as fresh only once their tile launches are enqueued: a failed launch leaves them stale and the next launch that
  //  reads the mirror tries again -- ADVICE r05)
  const int rc_t = retile(c, first_slot, n, c->pyr.n_levels);
  if (rc_t) {
    for (int s = first_slot; s < first_slot + n && s < (int)c->tiled_stale.size(); ++s)
      if (!c->tiled_stale[(size_t)s]) { c->tiled_stale[(size_t)s] = 1; ++c->tiled_stale_count; }
    return rc_t;
  }
  mark_tiled_fresh(c, first_slot, n);
  return PLSVO_OK;
}

extern "C" int plsvo_hip_build_pyramid(plsvo_ctx* c, int slot, const uint8_t* level0, int stride_bytes, int rounding) {
  CTX_CHECK(c);
  int rc = check_slot(c, slot); if (rc) return rc;
  if (!level0 || stride_bytes < c->pyr.w[0]) return fail(c, PLSVO_E_INVALID, "build_pyramid: bad arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  uint8_t* dst = c->pyr_slab.as<uint8_t>() + (size_t)slot * c->pyr.slot_bytes + c->pyr.off[0];
  HIP_TRY(c, hipMemcpy2DAsync(dst, (size_t)c->pyr.w[0], level0, (size_t)stride_bytes, (size_t)c->pyr.w[0], (size_t)c->pyr.h[0],
                              hipMemcpyHostToDevice, c->stream));
  rc = build_levels(c, slot, 1, rounding); if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return PLSVO_OK;
}

extern "C" int plsvo_hip_build_pyramids_dev(plsvo_ctx* c, int first_slot, int n, const void* d_level0, int stride_bytes,
                                            size_t image_pitch_bytes, int rounding) {
  CTX_CHECK(c);
  if (n <= 0 || !d_level0) return fail(c, PLSVO_E_INVALID, "build_pyramids_dev: bad arguments");
  int rc = check_slot(c, first_slot); if (rc) return rc;
  rc = check_slot(c, first_slot + n - 1); if (rc) return rc;
  if (stride_bytes < c->pyr.w[0]) return fail(c, PLSVO_E_INVALID, "build_pyramids_dev: stride smaller than width");
  HIP_TRY(c, hipSetDevice(c->device));
  uint8_t* base = c->pyr_slab.as<uint8_t>() + (size_t)first_slot * c->pyr.slot_bytes;
  HIP_TRY(c, launch_copy_level0(reinterpret_cast<const uint8_t*>(d_level0), image_pitch_bytes, c->pyr.w[0], c->pyr.h[0], stride_bytes,
                                base + c->pyr.off[0], c->pyr.slot_bytes, n, c->stream));
  return build_levels(c, first_slot, n, rounding);
}

// "cur becomes ref" for a block of streams without crossing PCIe: device-to-device copy of whole slots, row-major slab and tiled
// mirror alike (src/frame_handler_mono.cpp:272-274: the frame aligned against in the next call is the one just tracked)
extern "C" int plsvo_hip_copy_slots(plsvo_ctx* c, int dst_first, int src_first, int n) {
  CTX_CHECK(c);
  if (n <= 0) return fail(c, PLSVO_E_INVALID, "copy_slots: bad arguments");
  int rc = check_slot(c, dst_first); if (rc) return rc;
  rc = check_slot(c, dst_first + n - 1); if (rc) return rc;
  rc = check_slot(c, src_first); if (rc) return rc;
  rc = check_slot(c, src_first + n - 1); if (rc) return rc;
  if (dst_first < src_first + n && src_first < dst_first + n) return fail(c, PLSVO_E_INVALID, "copy_slots: the two slot ranges overlap");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(c->pyr_slab.as<uint8_t>() + (size_t)dst_first * c->pyr.slot_bytes, c->pyr_slab.as<uint8_t>() + (size_t)src_first * c->pyr.slot_bytes,
                            (size_t)n * c->pyr.slot_bytes, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->pyr_tiled.as<uint8_t>() + (size_t)dst_first * c->pyr.tslot_bytes, c->pyr_tiled.as<uint8_t>() + (size_t)src_first * c->pyr.tslot_bytes,
                            (size_t)n * c->pyr.tslot_bytes, hipMemcpyDeviceToDevice, c->stream));
  for (int k = 0; k < n; ++k) {   // a stale mirror stays stale in its copy (and a fresh one replaces a stale one)
    const uint8_t st = c->tiled_stale[(size_t)(src_first + k)];
    uint8_t& dt = c->tiled_stale[(size_t)(dst_first + k)];
    if (st != dt) { c->tiled_stale_count += st ? 1 : -1; dt = st; }
  }
  return PLSVO_OK;
}

extern "C" int plsvo_hip_download_level(plsvo_ctx* c, int slot, int level, uint8_t* out) {
  CTX_CHECK(c);
  int rc = check_slot(c, slot); if (rc) return rc;
  if (level < 0 || level >= c->pyr.n_levels || !out) return fail(c, PLSVO_E_INVALID, "download_level: bad arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  const uint8_t* src = c->pyr_slab.as<uint8_t>() + (size_t)slot * c->pyr.slot_bytes + c->pyr.off[level];
  HIP_TRY(c, hipMemcpyAsync(out, src, (size_t)c->pyr.w[level] * c->pyr.h[level], hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return PLSVO_OK;
}

// ---- alignment -----------------------------------------------------------------------------------
template <typename T>
static int upload(plsvo_ctx* c, DevBuf& buf, const std::vector<T>& v) {
  HIP_TRY(c, buf.ensure(std::max(v.size(), (size_t)1) * sizeof(T)));
  if (!v.empty()) HIP_TRY(c, hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
  return PLSVO_OK;
}

// Static patch-slot layout of one job at one level (align_kernels.hip header).  Points own slots [0, n_pts).  The kernel works through the
// slots in WAVE-ROUNDS of 64 (slots [64 r, 64 r + 64) go to the 64 lanes of one wave) and weights a line from the residual sums its samples
// left in LDS behind a wave-level fence, so all samples of a segment with N <= 64 must sit inside ONE round; longer segments make the
// level a two-pass level (workgroup barrier between the passes).  Every round costs the same whether its lanes hold patches or holes, so
// the segments are PACKED into the rounds: first-fit in decreasing N (ties in feature order), starting in the round the points leave
// partly empty.  (Rounds 1-5 placed the segments in feature order from the next multiple of 32 behind the points and kept them from
// straddling multiples of 32 -- a left-over of the lane-pair kernel: 6.0 / 6.7 / 8.75 rounds per pass at levels 3 / 2 / 1 of BASELINE
// configs[1] where the patches fill 5 / 6 / 8.)  Segments without a landmark on entry, or whose end points fail the 3-pixel border test of
// the level (src/sparse_img_align.cpp:299-301), get no slots (code -1).  Host-only arithmetic; no result depends on where a segment sits.
// seg_align: 64 = small batches, whose point and segment slots may go to two workgroups (the segments start at a multiple of 64);
//            otherwise the segments start right behind the points.
static int slot_layout(const plsvo_align_in* in, int level, int seg_align, int32_t* seg_code, int32_t* n_slots, int32_t* long_lines, long long* n_patches) {
  if (!in || level < 0 || level >= PLSVO_MAX_LEVELS || in->n_pts < 0 || in->n_seg < 0 || (in->n_seg > 0 && !seg_code)) return PLSVO_E_INVALID;
  const plsvo_align_in& a = *in;
  constexpr int kRound = 64;
  long long n_real = a.n_pts;
  int any_long = 0;
  const double scale = 1.0 / (double)(1 << level);
  const int cw = a.cam.width / (1 << level), ch = a.cam.height / (1 << level);
  std::vector<int> shorts, longs;     // segments with slots: N <= 64 / longer
  for (int s = 0; s < a.n_seg; ++s) {
    seg_code[s] = -1;
    const bool alive_on_entry = a.seg_alive_in ? (a.seg_alive_in[s] != 0) : true;
    const double sx = a.seg_spx[2 * s], sy = a.seg_spx[2 * s + 1], ex = a.seg_epx[2 * s], ey = a.seg_epx[2 * s + 1];
    const int isx = (int)(sx * scale), isy = (int)(sy * scale), iex = (int)(ex * scale), iey = (int)(ey * scale);
    const bool vis = isx >= 3 && isx < cw - 3 && isy >= 3 && isy < ch - 3 && iex >= 3 && iex < cw - 3 && iey >= 3 && iey < ch - 3;
    if (alive_on_entry && vis) {
      const int N = seg_num_samples(sx, sy, ex, ey, a.seg_len[s], level);
      if (N > 2047) return PLSVO_E_CAPACITY;
      seg_code[s] = N << 20;             // (the first slot is filled in below)
      (N <= kRound ? shorts : longs).push_back(s);
      n_real += N;
    }
  }
  const long long seg0 = (seg_align == 64 && a.n_seg > 0) ? (((long long)a.n_pts + 63) & ~63LL) : (long long)a.n_pts;   // first slot a segment may take
  long long used = a.n_pts;
  if (!shorts.empty() || !longs.empty()) {
    std::stable_sort(shorts.begin(), shorts.end(), [&](int x, int y) { return (seg_code[x] >> 20) > (seg_code[y] >> 20); });
    const long long r0 = seg0 / kRound;                                      // the round the first segment may share with the last points
    std::vector<int> fill(1, (int)(seg0 - r0 * kRound));                     // slots taken in round r0 + k
    for (int s : shorts) {
      const int N = seg_code[s] >> 20;
      size_t k = 0;
      while (k < fill.size() && fill[k] + N > kRound) ++k;
      if (k == fill.size()) fill.push_back(0);
      const long long first = (r0 + (long long)k) * kRound + fill[k];
      if (first + N > (1 << 20) - 8) return PLSVO_E_CAPACITY;
      seg_code[s] |= (int)first;
      fill[k] += N;
      used = std::max(used, first + N);
    }
    long long cur = shorts.empty() ? ((seg0 + kRound - 1) / kRound) * kRound : (r0 + (long long)fill.size()) * kRound;   // behind the packed rounds
    for (int s : longs) {                                                    // two-pass level: any contiguous run of slots will do
      const int N = seg_code[s] >> 20;
      if (cur + N > (1 << 20) - 8) return PLSVO_E_CAPACITY;
      seg_code[s] |= (int)cur;
      cur += N; used = cur;
      any_long = 1;
    }
  }
  if (n_slots) *n_slots = (int32_t)used;
  if (long_lines) *long_lines = any_long;
  if (n_patches) *n_patches = n_real;
  return PLSVO_OK;
}
extern "C" int plsvo_align_slot_layout(const plsvo_align_in* in, int level, int32_t* seg_code, int32_t* n_slots, int32_t* long_lines,
                                       long long* n_patches) {
  return slot_layout(in, level, 32, seg_code, n_slots, long_lines, n_patches);
}

// Up to three device ranges to the host with ONE wait: through the context's pinned download buffer when they fit (a per-frame caller's
// results: a few hundred bytes of state + the flags), straight into the caller's (pageable) buffers otherwise.  On return h[k] points at
// range k's bytes: inside the pinned buffer (valid until the next fetch) or inside `fallback`.
// (`fallback` is sized HERE whenever the pinned path is not taken -- too large, or no pinned memory to be had -- so the decision and the
//  storage can never disagree: ADVICE r05)
static int download_ranges(plsvo_ctx* c, int n_ranges, const void* const* src, const size_t* bytes, std::vector<uint8_t>& fallback, const uint8_t** h) {
  size_t total = 0, off[3] = { 0, 0, 0 };
  for (int k = 0; k < n_ranges; ++k) { off[k] = total; total += (bytes[k] + 63) & ~(size_t)63; }
  bool pinned = total <= ((size_t)4 << 20);
  if (pinned && c->dl_pinned_cap < total) {
    if (c->dl_pinned) (void)hipHostFree(c->dl_pinned);
    c->dl_pinned = nullptr; c->dl_pinned_cap = 0;
    const size_t want = std::max(total * 2, (size_t)64 << 10);
    if (hipHostMalloc(&c->dl_pinned, want, hipHostMallocDefault) == hipSuccess) c->dl_pinned_cap = want; else { c->dl_pinned = nullptr; pinned = false; }
  }
  if (!pinned) fallback.resize(std::max(total, (size_t)64));
  uint8_t* const base = pinned ? static_cast<uint8_t*>(c->dl_pinned) : fallback.data();
  for (int k = 0; k < n_ranges; ++k) {
    void* to = static_cast<void*>(base + off[k]);
    h[k] = static_cast<const uint8_t*>(to);
    if (bytes[k]) HIP_TRY(c, hipMemcpyAsync(to, src[k], bytes[k], hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return PLSVO_OK;
}

static const size_t kPinnedMax = (size_t)8 << 20;   // blobs up to 8 MB (hundreds of frames) bounce through pinned memory
static int upload_blob(plsvo_ctx* c, DevBuf& buf, const Blob& blob) {
  const size_t bytes = std::max(blob.host.size(), (size_t)256);
  HIP_TRY(c, buf.ensure(bytes));
  if (blob.host.empty()) return PLSVO_OK;
  // Small blobs (a frame, a few hundred frames) bounce through one of two pinned buffers: the DMA of a stage call is awaited only when
  // ITS buffer comes up for reuse two stage calls later (an event recorded behind the copy), so a per-frame caller's stage -> run -> fetch
  // makes ONE synchronisation, the fetch's.  (Round 3 waited for the stream inside every stage call.)
  if (blob.host.size() <= kPinnedMax) {
    const int k = c->pinned_next;
    if (!c->pinned_done[k] && hipEventCreate(&c->pinned_done[k]) != hipSuccess) c->pinned_done[k] = nullptr;
    if (c->pinned_done[k]) {
      HIP_TRY(c, hipEventSynchronize(c->pinned_done[k]));        // the copy that last read this buffer has landed (no-op when never recorded)
      if (c->pinned_cap[k] < blob.host.size()) {
        if (c->pinned[k]) (void)hipHostFree(c->pinned[k]);
        c->pinned[k] = nullptr; c->pinned_cap[k] = 0;
        const size_t want = std::max(blob.host.size() * 2, (size_t)1 << 20);
        if (hipHostMalloc(&c->pinned[k], want, hipHostMallocDefault) == hipSuccess) c->pinned_cap[k] = want; else c->pinned[k] = nullptr;
      }
      if (c->pinned[k]) {
        memcpy(c->pinned[k], blob.host.data(), blob.host.size());
        HIP_TRY(c, hipMemcpyAsync(buf.p, c->pinned[k], blob.host.size(), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipEventRecord(c->pinned_done[k], c->stream));
        c->pinned_next = k ^ 1;
        return PLSVO_OK;
      }
    }
  }
  // large blobs (or no pinned memory): straight from the caller's Blob, which must not be touched while the DMA is in flight -- wait here
  HIP_TRY(c, hipMemcpyAsync(buf.p, blob.host.data(), blob.host.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return PLSVO_OK;
}

extern "C" int plsvo_align_stage(plsvo_ctx* c, int n, const plsvo_align_in* in) {
  CTX_CHECK(c);
  if (n <= 0 || !in) return fail(c, PLSVO_E_INVALID, "align_stage: bad arguments");
  if (!c->pyr.base) return fail(c, PLSVO_E_STATE, "align_stage: pyramids not configured");
  HIP_TRY(c, hipSetDevice(c->device));
  c->a_staged = false;
  c->ch_staged = false;   // a resident frame step is built on the staged alignment batch: staging another one invalidates it
  std::vector<AlignJobDev> jobs((size_t)n);
  std::vector<double> T0((size_t)n * 7), ptpx, ptxyz, spx, epx, len, sp, sq;
  std::vector<uint8_t> alive;
  int gmax = -1, gmin = 99;
  // small batches may run two workgroups per frame (points | segments: plsvo_align_run decides from the launch shape): their segments start
  // at a multiple of 64 slots
  const int cus_stage = c->cu_count > 0 ? c->cu_count : 256;
  const int seg_align = (2 * n <= cus_stage) ? 64 : 32;
  int caps[PLSVO_MAX_LEVELS] = { 0 };
  std::vector<long long> work((size_t)n, 0);   // patches summed over the levels, per job: the launch-order key
  std::vector<int> seg_slot[PLSVO_MAX_LEVELS];   // per level, one entry per segment of the batch
  size_t patch_total = 0;
  for (int j = 0; j < n; ++j) {
    const plsvo_align_in& a = in[j];
    if (a.n_pts < 0 || a.n_seg < 0 || a.max_level < a.min_level || a.min_level < 0 || a.max_level >= c->pyr.n_levels || a.n_iter < 0)
      return fail(c, PLSVO_E_INVALID, "align_stage: bad job parameters (levels must exist in the configured pyramid)");
    if (a.ref_slot < 0 || a.ref_slot >= c->pyr.n_slots || a.cur_slot < 0 || a.cur_slot >= c->pyr.n_slots)
      return fail(c, PLSVO_E_CAPACITY, "align_stage: pyramid slot out of range");
    if (a.cam.width != c->pyr.w[0] || a.cam.height != c->pyr.h[0])
      return fail(c, PLSVO_E_INVALID, "align_stage: camera size does not match the configured pyramid");
    if ((a.n_pts > 0 && (!a.pt_px || !a.pt_xyz_ref)) || (a.n_seg > 0 && (!a.seg_spx || !a.seg_epx || !a.seg_len || !a.seg_p_ref || !a.seg_q_ref)))
      return fail(c, PLSVO_E_INVALID, "align_stage: null feature array");
    AlignJobDev& J = jobs[(size_t)j];
    J.ref_slot = a.ref_slot; J.cur_slot = a.cur_slot;
    J.fx = a.cam.fx; J.fy = a.cam.fy; J.cx = a.cam.cx; J.cy = a.cam.cy; J.width = a.cam.width; J.height = a.cam.height;
    J.max_level = a.max_level; J.min_level = a.min_level; J.n_iter = a.n_iter; J.eps = a.eps;
    J.skip = (a.n_pts == 0 && a.n_seg == 0) ? 1 : 0;
    J.pt_off = (int)(ptpx.size() / 2); J.n_pts = a.n_pts; J.seg_off = (int)len.size(); J.n_seg = a.n_seg;
    for (int k = 0; k < 7; ++k) T0[(size_t)j * 7 + k] = a.T_cur_from_ref[k];
    ptpx.insert(ptpx.end(), a.pt_px, a.pt_px + 2 * (size_t)a.n_pts);
    ptxyz.insert(ptxyz.end(), a.pt_xyz_ref, a.pt_xyz_ref + 3 * (size_t)a.n_pts);
    spx.insert(spx.end(), a.seg_spx, a.seg_spx + 2 * (size_t)a.n_seg);
    epx.insert(epx.end(), a.seg_epx, a.seg_epx + 2 * (size_t)a.n_seg);
    len.insert(len.end(), a.seg_len, a.seg_len + (size_t)a.n_seg);
    sp.insert(sp.end(), a.seg_p_ref, a.seg_p_ref + 3 * (size_t)a.n_seg);
    sq.insert(sq.end(), a.seg_q_ref, a.seg_q_ref + 3 * (size_t)a.n_seg);
    for (int s = 0; s < a.n_seg; ++s) alive.push_back(a.seg_alive_in ? (a.seg_alive_in[s] ? 1 : 0) : 1);
    // static patch-slot layout, per level (slot_layout above): points own slots [0, n_pts), the segments are packed behind them into
    // the kernel's wave-rounds of 64 slots; segments without a landmark on entry, or whose end points fail the 3-pixel border test of
    // the level (src/sparse_img_align.cpp:299-301), get no slots.
    int ub_max = 0;
    long long ub_sum = 0;
    J.long_mask = 0; J.ldlt_flavour = c->ldlt_flavour;
    for (int l = 0; l < PLSVO_MAX_LEVELS; ++l) J.n_slots[l] = 0;
    std::vector<int> codes((size_t)std::max(a.n_seg, 1));
    for (int l = a.max_level; l >= a.min_level; --l) {
      int n_slots = 0, long_lines = 0; long long n_real = 0;
      const int lrc = slot_layout(&a, l, seg_align, codes.data(), &n_slots, &long_lines, &n_real);
      if (lrc == PLSVO_E_CAPACITY) return fail(c, PLSVO_E_CAPACITY, "align_stage: a segment with more than 2047 samples or more than 2^20 patch slots in one job");
      if (lrc != PLSVO_OK) return fail(c, lrc, "align_stage: slot layout failed");
      seg_slot[(size_t)l].insert(seg_slot[(size_t)l].end(), codes.begin(), codes.begin() + a.n_seg);
      if (long_lines) J.long_mask |= 1 << l;
      const int slots4 = (n_slots + 3) & ~3;
      J.n_slots[l] = n_slots;
      ub_sum += n_real;
      if (!J.skip) caps[l] = std::max(caps[l], slots4);
      ub_max = std::max(ub_max, slots4);
    }
    J.patch_off = (int)patch_total; J.patch_cap = ub_max;
    work[(size_t)j] = J.skip ? 0 : ub_sum;
    patch_total += (size_t)ub_max;
    if (patch_total > (size_t)0x7fffffff / 16) return fail(c, PLSVO_E_CAPACITY, "align_stage: batch too large (patch index overflow)");
    if (!J.skip) { gmax = std::max(gmax, a.max_level); gmin = std::min(gmin, a.min_level); }
    // levels this job does not run still need an entry per segment: the table is indexed [level][segment of the batch]
    for (int l = 0; l < PLSVO_MAX_LEVELS; ++l)
      if (l > a.max_level || l < a.min_level) seg_slot[(size_t)l].insert(seg_slot[(size_t)l].end(), (size_t)a.n_seg, -1);
  }
  // launch order: most patches (summed over the levels) first; stable, so equal jobs keep their batch order
  std::vector<int> order((size_t)n);
  for (int j = 0; j < n; ++j) order[(size_t)j] = j;
  if (!c->env_align_no_lpt) std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return work[(size_t)x] > work[(size_t)y]; });
  int rc;
  const int slot_level0 = (gmax >= gmin && gmax >= 0) ? gmin : 0;
  const size_t total_seg = len.size();
  std::vector<int> slot_table;
  for (int l = slot_level0; l <= std::max(gmax, slot_level0); ++l) slot_table.insert(slot_table.end(), seg_slot[(size_t)l].begin(), seg_slot[(size_t)l].end());
  if (slot_table.empty()) slot_table.push_back(-1);
  Blob blob;
  const size_t o_slot = blob.add(slot_table), o_order = blob.add(order), o_jobs = blob.add(jobs), o_T0 = blob.add(T0), o_ptpx = blob.add(ptpx),
               o_ptxyz = blob.add(ptxyz), o_spx = blob.add(spx), o_epx = blob.add(epx), o_len = blob.add(len), o_sp = blob.add(sp),
               o_sq = blob.add(sq), o_alive = blob.add(alive);
  if ((rc = upload_blob(c, c->a_d_blob, blob))) return rc;
  HIP_TRY(c, c->a_d_alive.ensure(std::max(alive.size(), (size_t)1)));
  HIP_TRY(c, c->a_d_state.ensure((size_t)n * sizeof(AlignStateDev)));
  HIP_TRY(c, c->a_d_poses.ensure((size_t)n * 7 * sizeof(double)));
  const size_t pt = std::max(patch_total, (size_t)4);
  HIP_TRY(c, c->a_d_pxyz.ensure(pt * 3 * sizeof(double)));
  HIP_TRY(c, c->a_d_puv.ensure(pt * 2 * sizeof(float)));
  HIP_TRY(c, c->a_d_cref.ensure(pt * 64));   // one 64-byte record per patch slot (latency shapes: 192 B of float rows, sized in plsvo_align_run)
  c->a_patch_total = pt;
  const size_t npt_total = ptpx.size() / 2 + 32;
  HIP_TRY(c, c->a_d_chi.ensure(2 * npt_total * 16 * sizeof(float)));   // two planes of the points' per-pixel chi2 terms
  if (c->a_trace_cap > 0) HIP_TRY(c, c->a_d_log.ensure((size_t)n * c->a_trace_cap * sizeof(plsvo_align_iterlog)));

  AlignBatchDev& b = c->a_b;
  uint8_t* const base = c->a_d_blob.as<uint8_t>();
  b.jobs = reinterpret_cast<const AlignJobDev*>(base + o_jobs); b.state = c->a_d_state.as<AlignStateDev>();
  b.T0 = reinterpret_cast<const double*>(base + o_T0);
  b.pt_px = reinterpret_cast<const double*>(base + o_ptpx); b.pt_xyz = reinterpret_cast<const double*>(base + o_ptxyz);
  b.seg_spx = reinterpret_cast<const double*>(base + o_spx); b.seg_epx = reinterpret_cast<const double*>(base + o_epx);
  b.seg_len = reinterpret_cast<const double*>(base + o_len);
  b.seg_p = reinterpret_cast<const double*>(base + o_sp); b.seg_q = reinterpret_cast<const double*>(base + o_sq);
  b.seg_alive_in = base + o_alive; b.seg_alive = c->a_d_alive.as<uint8_t>();
  b.patch_xyz = c->a_d_pxyz.as<double>(); b.patch_uvref = c->a_d_puv.as<float>();
  b.cache_ref = c->a_d_cref.as<float>();
  b.chi_terms = c->a_d_chi.as<float>(); b.chi_plane = (unsigned long long)npt_total * 16;
  b.seg_slot = reinterpret_cast<const int*>(base + o_slot); b.slot_level0 = slot_level0; b.slot_stride = (int)total_seg;
  b.poses = c->a_d_poses.as<double>();
  b.pyr = c->pyr;
  b.log = c->a_trace_cap > 0 ? c->a_d_log.as<plsvo_align_iterlog>() : nullptr;
  b.log_cap = c->a_trace_cap;
  b.n_jobs = n;
  b.pair = 0; b.xseq0 = 0; b.xbuf = nullptr; b.work_key = nullptr;   // (plsvo_align_run decides)
  b.order = reinterpret_cast<const int*>(base + o_order);
  c->a_stage_order = b.order;
  c->a_jobs.swap(jobs);
  c->a_n = n; c->a_total_seg = (int)alive.size(); c->a_gmax = gmax; c->a_gmin = gmin;
  for (int l = 0; l < PLSVO_MAX_LEVELS; ++l) c->a_cap[l] = caps[l];
  c->a_scap = 4;
  for (int j = 0; j < n; ++j) c->a_scap = std::max(c->a_scap, in[j].n_seg);
  c->a_seg_align = seg_align;
  c->a_staged = true; c->a_run_seq = 0;
  return PLSVO_OK;
}

// Launch configuration of the fused alignment kernel (measured on MI355X).  A frame pair is one workgroup (two for the
// smallest batches, plsvo_align_run); how many threads it gets depends on how many frames there are to fill the chip with:
//   >= 64 frames per CU (throughput): 64 threads -- one wave per frame, eight frames per CU hide each other's serial solve / update tails;
//   > 4 / > 1 frames per CU:          128 / 256 threads;
//   at most one frame per CU (latency): 512 threads, the whole CU works on one frame (its points; a second CU on its segments).
// Environment overrides (experiments only, read once at plsvo_hip_create): PLSVO_ALIGN_THREADS, PLSVO_ALIGN_PER_LEVEL, PLSVO_ALIGN_LDS_PAD,
// PLSVO_ALIGN_NO_PAIR, PLSVO_ALIGN_NO_REORDER / PLSVO_POSEOPT_NO_REORDER (launch order of a re-run staged batch), PLSVO_ALIGN_REORDER_MIN /
// PLSVO_POSEOPT_REORDER_MIN (its threshold, tests).  A switch set to "0" is off.
static void pick_align_config(const plsvo_ctx* c, int n_jobs, int cap, int scap, int max_pts, int* threads, size_t* lds, int* chi_lds_pts) {
  const int cus = c->cu_count > 0 ? c->cu_count : 256;
  int t = 64;                            // >= 64 frames per CU: one wave per frame, no workgroup barrier at all
  if (n_jobs <= cus) t = 512;
  else if (n_jobs <= 4 * cus) t = 256;
  else if (n_jobs < 64 * cus) t = 128;
  if (c->env_align_threads) t = c->env_align_threads;
  // chi2 terms: in LDS when a workgroup owns (most of) a CU -- a global store is acknowledged from the memory side (~1 us) and a
  // lone frame has no other wave to hide that behind -- in HBM planes when eight frames share the CU's LDS
  int pts = t >= 256 ? ((max_pts + 3) & ~3) : 0;
  if (pts > 0 && align_level_lds_bytes(t, cap, scap, pts) > c->lds_per_block) pts = 0;
  *chi_lds_pts = pts;
  *threads = t; *lds = align_level_lds_bytes(t, cap, scap, pts);
  *lds += (size_t)c->env_align_lds_pad;   // occupancy experiments: unused LDS bytes per workgroup
}

// roctx range around the host side of an ABI call (rocprofv3 --marker-trace shows it).  The marker library is looked up at run
// time, once: the product library neither links against the profiler SDK nor fails to load without it.
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_LAZY | RTLD_GLOBAL);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_LAZY | RTLD_GLOBAL);
    if (!h) return;
    push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
    pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!push || !pop) { push = nullptr; pop = nullptr; }
  }
};
const Roctx& roctx() { static const Roctx r; return r; }
struct RoctxRange {
  explicit RoctxRange(const char* name) { if (roctx().push) roctx().push(name); }
  ~RoctxRange() { if (roctx().pop) roctx().pop(); }
};
}  // namespace

extern "C" int plsvo_align_run(plsvo_ctx* c) {
  CTX_CHECK(c);
  RoctxRange range("sparse_img_align");
  if (!c->a_staged) return fail(c, PLSVO_E_STATE, "align_run: no staged batch");
  HIP_TRY(c, hipSetDevice(c->device));
  // one launch: the workgroup of a job resets its solver state, runs its levels and writes its pose
  // (jobs without features, src/sparse_img_align.cpp:58-62, only do the first and the last)
  int cap = 4;
  const bool have_levels = c->a_gmax >= c->a_gmin && c->a_gmax >= 0;
  if (have_levels) for (int l = c->a_gmin; l <= c->a_gmax; ++l) cap = std::max(cap, c->a_cap[l]);
  int threads; size_t lds;
  const int scap = (c->a_scap + 3) & ~3;
  int max_pts = 0, chi_lds_pts = 0;
  for (const AlignJobDev& J : c->a_jobs) max_pts = std::max(max_pts, J.n_pts);
  pick_align_config(c, c->a_n, cap, scap, max_pts, &threads, &lds, &chi_lds_pts);
  c->a_b.chi_lds_pts = chi_lds_pts;
  if (lds > c->lds_per_block) return fail(c, PLSVO_E_CAPACITY, "align_run: slot tables do not fit in LDS (too many features in one job)");
  if (threads <= 64 && c->tiled_stale_count > 0) {   // the one-wave-per-frame shape reads the tiled mirror: bring the uploaded slots' tiles up to date first
    std::vector<int> slots;
    slots.reserve(2 * c->a_jobs.size());
    for (const AlignJobDev& J : c->a_jobs) { slots.push_back(J.ref_slot); slots.push_back(J.cur_slot); }
    const int rc_t = retile_stale(c, slots); if (rc_t) return rc_t;
  }
  // TWO WORKGROUPS PER FRAME: a frame that has a CU to itself is bound by that CU's instruction issue (two 64-slot wave-rounds per SIMD per
  // pass at 380 slots); with at most cu_count / 2 frames a second CU takes the segment slots, the two exchange 32 partial sums per
  // iteration through L2 (~1 us) and both run the identical solver.  Needs the latency shape, the chi2 planes in LDS (no deferred
  // decisions) and the layout staged for it; the per-level debug launches keep one workgroup.
  const int cus_run = c->cu_count > 0 ? c->cu_count : 256;
  // (exchange tags are launch << 10 | exchange, at most two exchanges per Gauss-Newton iteration -- the totals and a near tie's float sums:
  //  a launch that could need more than 1023 of them keeps one workgroup per frame, so a tag never runs into the next launch's range)
  int max_iter = 0;
  for (const AlignJobDev& J : c->a_jobs) max_iter = std::max(max_iter, J.n_iter);
  const long long max_exchanges = have_levels ? (long long)(c->a_gmax - c->a_gmin + 1) * (long long)max_iter * 2 : 0;
  const bool pair = threads >= kQuadMinThreads && chi_lds_pts > 0 && c->a_seg_align == 64 && 2 * c->a_n <= cus_run && !c->env_align_no_pair &&
                    !c->env_align_per_level && have_levels && max_exchanges < 1024;
  c->a_b.pair = pair ? 1 : 0;
  if (pair) {
    const size_t xbytes = (size_t)c->a_n * 256 * sizeof(unsigned long long);
    if (c->a_d_xbuf.cap < xbytes) {
      HIP_TRY(c, c->a_d_xbuf.ensure(std::max(xbytes, (size_t)128 * 2048)));
      HIP_TRY(c, hipMemsetAsync(c->a_d_xbuf.p, 0, c->a_d_xbuf.cap, c->stream));   // tag 0 is never used
    }
    c->a_b.xbuf = c->a_d_xbuf.as<unsigned long long>();
    c->x_launch = (c->x_launch + 1u) & 0x3fffffu;
    if (c->x_launch == 0) {   // the 22-bit launch counter wrapped: old tags could repeat -- zero the granules again (stream-ordered behind the last launch)
      HIP_TRY(c, hipMemsetAsync(c->a_d_xbuf.p, 0, c->a_d_xbuf.cap, c->stream));
      c->x_launch = 1;
    }
    c->a_b.xseq0 = c->x_launch << 10;
  }
  if (threads >= kQuadMinThreads) {   // latency shapes keep the reference patches as float rows: 192 B per slot (small batches only: <= 4 frames per CU)
    HIP_TRY(c, c->a_d_cref.ensure(c->a_patch_total * 192));
    c->a_b.cache_ref = c->a_d_cref.as<float>();
  }
  const bool per_level = c->env_align_per_level;
  // (a one-shot call -- plsvo_sparse_align_batch -- never re-runs its batch: no key, no sort)
  const bool reorder = have_levels && !per_level && c->a_n > (c->env_align_reorder_min > 0 ? c->env_align_reorder_min - 1 : cus_run * 512 / threads) &&   // more frames than resident workgroups (eight waves per CU)
                       !c->env_align_no_reorder && !c->env_align_no_lpt && !c->a_one_shot;
  c->a_b.work_key = nullptr;
  if (reorder) {
    if (c->a_d_workkey.cap < (size_t)c->a_n * sizeof(int)) {   // a fresh key buffer starts at zero work: jobs that leave the kernel early never write theirs
      HIP_TRY(c, c->a_d_workkey.ensure((size_t)c->a_n * sizeof(int)));
      HIP_TRY(c, hipMemsetAsync(c->a_d_workkey.p, 0, c->a_d_workkey.cap, c->stream));
    }
    c->a_b.work_key = c->a_d_workkey.as<int>();
  }
  if (!per_level || !have_levels) {
    EventPair ep{}; prof_begin(c, PLSVO_K_ALIGN_LEVEL, &ep);
    HIP_TRY(c, launch_align_levels(c->a_b, cap, scap, have_levels ? c->a_gmax : 0, have_levels ? c->a_gmin : 0, 1, threads, lds, c->stream));
    prof_end(c, PLSVO_K_ALIGN_LEVEL, &ep);
    // a batch with more frames than resident slots: the NEXT launch of this staged batch starts its frames longest-first by what they cost in
    // this one (one small kernel behind the launch; the order buffers alternate, the running launch's is never written)
    if (reorder) {
      DevBuf& ob = c->a_d_order[c->a_order_next];
      HIP_TRY(c, ob.ensure((size_t)c->a_n * sizeof(int)));
      HIP_TRY(c, launch_align_reorder(c->a_b.work_key, c->a_n, ob.as<int>(), 7, c->stream));
      c->a_b.order = ob.as<int>();
      c->a_order_next ^= 1;
    }
  } else {
    for (int level = c->a_gmax; level >= c->a_gmin; --level) {   // debug: one launch per level, state carried in HBM
      EventPair ep{}; prof_begin(c, PLSVO_K_ALIGN_LEVEL, &ep);
      HIP_TRY(c, launch_align_levels(c->a_b, cap, scap, level, level, level == c->a_gmax ? 1 : 0, threads, lds, c->stream));
      prof_end(c, PLSVO_K_ALIGN_LEVEL, &ep);
    }
  }
  c->a_run_seq = ++c->run_seq;
  return PLSVO_OK;
}

extern "C" int plsvo_align_fetch(plsvo_ctx* c, int n, plsvo_align_out* out) {
  CTX_CHECK(c);
  if (!c->a_staged) return fail(c, PLSVO_E_STATE, "align_fetch: no staged batch");
  if (n != c->a_n || !out) return fail(c, PLSVO_E_INVALID, "align_fetch: n does not match the staged batch");
  HIP_TRY(c, hipSetDevice(c->device));
  std::vector<uint8_t> big;   // (large batches, or no pinned memory: download_ranges sizes it)
  const size_t rb[2] = { (size_t)n * sizeof(AlignStateDev), (size_t)std::max(c->a_total_seg, 0) };
  const void* const rs[2] = { c->a_d_state.p, c->a_d_alive.p };
  const uint8_t* rh[2] = { nullptr, nullptr };
  { const int rc_d = download_ranges(c, 2, rs, rb, big, rh); if (rc_d) return rc_d; }
  const AlignStateDev* const st = reinterpret_cast<const AlignStateDev*>(rh[0]);
  const uint8_t* const alive = rh[1];
  int dev_err = 0;
  for (int j = 0; j < n; ++j) {
    const AlignStateDev& s = st[(size_t)j];
    plsvo_align_out& o = out[j];
    uint8_t* alive_out = o.seg_alive_out;
    memset(&o, 0, sizeof(o));
    o.seg_alive_out = alive_out;
    for (int k = 0; k < 7; ++k) o.T_cur_from_ref[k] = s.T[k];
    o.n_meas = s.n_meas; o.n_tracked = s.n_meas / PLSVO_PATCH_AREA;
    for (int k = 0; k < 36; ++k) o.H[k] = s.H[k];
    o.chi2 = s.chi2;
    for (int l = 0; l < PLSVO_MAX_LEVELS; ++l) o.iters_per_level[l] = s.iters[l];
    o.status = s.stop ? 1 : 0;
    if (alive_out && c->a_jobs[(size_t)j].n_seg > 0)
      memcpy(alive_out, alive + c->a_jobs[(size_t)j].seg_off, (size_t)c->a_jobs[(size_t)j].n_seg);
    if (s.error) dev_err = s.error;
  }
  if (dev_err) return fail(c, PLSVO_E_CAPACITY, "align: device-side capacity check failed (code " + std::to_string(dev_err) + ")");
  return PLSVO_OK;
}

extern "C" int plsvo_sparse_align_batch(plsvo_ctx* c, int n, const plsvo_align_in* in, plsvo_align_out* out) {
  const bool host_timing = c && c->env_host_timing;   // debug: wall time of the three steps on stderr
  const auto t0 = std::chrono::steady_clock::now();
  int rc = plsvo_align_stage(c, n, in); if (rc) return rc;
  const auto t1 = std::chrono::steady_clock::now();
  c->a_one_shot = true;
  rc = plsvo_align_run(c);
  c->a_one_shot = false;
  if (rc) return rc;
  const auto t2 = std::chrono::steady_clock::now();
  rc = plsvo_align_fetch(c, n, out);
  if (host_timing) {
    const auto t3 = std::chrono::steady_clock::now();
    fprintf(stderr, "[plsvo_hip] sparse_align_batch n=%d: stage %.1f us, run (enqueue) %.1f us, fetch (incl. wait) %.1f us\n", n,
            std::chrono::duration<double, std::micro>(t1 - t0).count(), std::chrono::duration<double, std::micro>(t2 - t1).count(),
            std::chrono::duration<double, std::micro>(t3 - t2).count());
  }
  return rc;
}
extern "C" int plsvo_sparse_align(plsvo_ctx* c, const plsvo_align_in* in, plsvo_align_out* out) {
  return plsvo_sparse_align_batch(c, 1, in, out);
}

extern "C" int plsvo_align_set_trace(plsvo_ctx* c, int max_records_per_job) {
  CTX_CHECK(c);
  if (max_records_per_job < 0) return fail(c, PLSVO_E_INVALID, "set_trace: negative capacity");
  c->a_trace_cap = max_records_per_job;
  c->a_staged = false;  // the trace buffer is sized at stage time
  return PLSVO_OK;
}

extern "C" int plsvo_align_fetch_trace(plsvo_ctx* c, int job, plsvo_align_iterlog* out, int max_records, int* n_records) {
  CTX_CHECK(c);
  if (!c->a_staged || c->a_trace_cap <= 0) return fail(c, PLSVO_E_STATE, "fetch_trace: tracing not enabled for the staged batch");
  if (job < 0 || job >= c->a_n || !out || !n_records) return fail(c, PLSVO_E_INVALID, "fetch_trace: bad arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  AlignStateDev s;
  HIP_TRY(c, hipMemcpyAsync(&s, c->a_d_state.as<AlignStateDev>() + job, sizeof(s), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const int have = std::min(s.log_count, c->a_trace_cap);
  const int nrec = std::min(have, max_records);
  if (nrec > 0) {
    HIP_TRY(c, hipMemcpyAsync(out, c->a_d_log.as<plsvo_align_iterlog>() + (size_t)job * c->a_trace_cap, (size_t)nrec * sizeof(plsvo_align_iterlog),
                              hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  *n_records = nrec;
  return PLSVO_OK;
}

extern "C" const double* plsvo_align_poses_dev(plsvo_ctx* c) { return (c && c->a_staged) ? c->a_d_poses.as<double>() : nullptr; }

// debug: per-phase s_memtime ticks summed over the batch (all zero unless built with -DPLSVO_TIMING); not in the public header
extern "C" int plsvo_align_phase_ticks(plsvo_ctx* c, unsigned long long* out8) {
  CTX_CHECK(c);
  if (!c->a_staged || !out8) return fail(c, PLSVO_E_STATE, "phase_ticks: no staged batch");
  std::vector<AlignStateDev> st((size_t)c->a_n);
  HIP_TRY(c, hipMemcpyAsync(st.data(), c->a_d_state.p, (size_t)c->a_n * sizeof(AlignStateDev), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < 8; ++k) { out8[k] = 0; for (auto& s : st) out8[k] += s.phase_ticks[k]; }
  return PLSVO_OK;
}

extern "C" int plsvo_poseopt_phase_ticks(plsvo_ctx* c, unsigned long long* out8) {
  CTX_CHECK(c);
  if (!c->p_staged || !out8) return fail(c, PLSVO_E_STATE, "phase_ticks: no staged batch");
  std::vector<PoseStateDev> st((size_t)c->p_n);
  HIP_TRY(c, hipMemcpyAsync(st.data(), c->p_d_state.p, (size_t)c->p_n * sizeof(PoseStateDev), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < 8; ++k) { out8[k] = 0; for (auto& s : st) out8[k] += s.phase_ticks[k]; }
  return PLSVO_OK;
}

extern "C" int plsvo_align_copy_poses(plsvo_ctx* c, double* d_dst) {
  CTX_CHECK(c);
  if (!c->a_staged || !d_dst) return fail(c, PLSVO_E_STATE, "align_copy_poses: no staged batch / null destination");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(d_dst, c->a_d_poses.p, (size_t)c->a_n * 7 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  return PLSVO_OK;
}

extern "C" int plsvo_align_work(plsvo_ctx* c, uint64_t* patch_levels, uint64_t* patch_iters) {
  CTX_CHECK(c);
  if (!c->a_staged) return fail(c, PLSVO_E_STATE, "align_work: no staged batch");
  HIP_TRY(c, hipSetDevice(c->device));
  std::vector<AlignStateDev> st((size_t)c->a_n);
  HIP_TRY(c, hipMemcpyAsync(st.data(), c->a_d_state.p, (size_t)c->a_n * sizeof(AlignStateDev), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  uint64_t pl = 0, pi = 0;
  for (auto& s : st) { pl += s.patch_levels; pi += s.patch_iters; }
  if (patch_levels) *patch_levels = pl;
  if (patch_iters) *patch_iters = pi;
  return PLSVO_OK;
}

extern "C" int plsvo_align_work_points(plsvo_ctx* c, uint64_t* point_patch_iters) {
  CTX_CHECK(c);
  if (!c->a_staged) return fail(c, PLSVO_E_STATE, "align_work_points: no staged batch");
  HIP_TRY(c, hipSetDevice(c->device));
  std::vector<AlignStateDev> st((size_t)c->a_n);
  HIP_TRY(c, hipMemcpyAsync(st.data(), c->a_d_state.p, (size_t)c->a_n * sizeof(AlignStateDev), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  uint64_t n = 0;
  for (auto& s : st) n += s.patch_iters_pt;
  if (point_patch_iters) *point_patch_iters = n;
  return PLSVO_OK;
}

extern "C" int plsvo_align_launch_order(plsvo_ctx* c, int n, int32_t* order) {
  CTX_CHECK(c);
  if (!c->a_staged) return fail(c, PLSVO_E_STATE, "align_launch_order: no staged batch");
  if (n != c->a_n || !order) return fail(c, PLSVO_E_INVALID, "align_launch_order: n does not match the staged batch");
  HIP_TRY(c, hipSetDevice(c->device));
  if (!c->a_b.order) { for (int j = 0; j < n; ++j) order[j] = j; return PLSVO_OK; }
  HIP_TRY(c, hipMemcpyAsync(order, c->a_b.order, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return PLSVO_OK;
}

extern "C" int plsvo_align_chi2_ties(plsvo_ctx* c, uint64_t* iterations, uint64_t* ties, uint64_t* near_ties_without_terms) {
  CTX_CHECK(c);
  if (!c->a_staged) return fail(c, PLSVO_E_STATE, "align_chi2_ties: no staged batch");
  HIP_TRY(c, hipSetDevice(c->device));
  std::vector<AlignStateDev> st((size_t)c->a_n);
  HIP_TRY(c, hipMemcpyAsync(st.data(), c->a_d_state.p, (size_t)c->a_n * sizeof(AlignStateDev), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  uint64_t it = 0, t = 0, un = 0;
  for (auto& s : st) { t += (uint64_t)s.chi2_ties; un += (uint64_t)s.chi2_unarmed; for (int l = 0; l < PLSVO_MAX_LEVELS; ++l) it += (uint64_t)s.iters[l]; }
  if (iterations) *iterations = it;
  if (ties) *ties = t;
  if (near_ties_without_terms) *near_ties_without_terms = un;
  return PLSVO_OK;
}

// ---- pose optimisation ---------------------------------------------------------------------------
extern "C" int plsvo_poseopt_stage(plsvo_ctx* c, int n, const plsvo_poseopt_in* in) {
  CTX_CHECK(c);
  if (n <= 0 || !in) return fail(c, PLSVO_E_INVALID, "poseopt_stage: bad arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  c->p_staged = false;
  std::vector<PoseJobDev> jobs((size_t)n);
  std::vector<double> f, pos, line, spos, epos;
  std::vector<int> plev, slev;
  for (int j = 0; j < n; ++j) {
    const plsvo_poseopt_in& a = in[j];
    if (a.n_pts < 0 || a.n_seg < 0 || a.n_iter < 0) return fail(c, PLSVO_E_INVALID, "poseopt_stage: bad job parameters");
    if ((a.n_pts > 0 && (!a.pt_f || !a.pt_pos || !a.pt_level)) || (a.n_seg > 0 && (!a.seg_line || !a.seg_spos || !a.seg_epos || !a.seg_level)))
      return fail(c, PLSVO_E_INVALID, "poseopt_stage: null feature array");
    PoseJobDev& J = jobs[(size_t)j];
    for (int k = 0; k < 7; ++k) J.T0[k] = a.T_f_w[k];
    J.fx = a.fx; J.reproj_thresh = a.reproj_thresh; J.n_iter = a.n_iter; J.n_iter_ref = a.n_iter_ref; J.ldlt_flavour = c->ldlt_flavour; J.reserved0 = 0;
    J.pt_off = (int)plev.size(); J.n_pts = a.n_pts; J.seg_off = (int)slev.size(); J.n_seg = a.n_seg;
    f.insert(f.end(), a.pt_f, a.pt_f + 3 * (size_t)a.n_pts);
    pos.insert(pos.end(), a.pt_pos, a.pt_pos + 3 * (size_t)a.n_pts);
    plev.insert(plev.end(), a.pt_level, a.pt_level + (size_t)a.n_pts);
    line.insert(line.end(), a.seg_line, a.seg_line + 3 * (size_t)a.n_seg);
    spos.insert(spos.end(), a.seg_spos, a.seg_spos + 3 * (size_t)a.n_seg);
    epos.insert(epos.end(), a.seg_epos, a.seg_epos + 3 * (size_t)a.n_seg);
    slev.insert(slev.end(), a.seg_level, a.seg_level + (size_t)a.n_seg);
    for (int i = 0; i < a.n_pts; ++i) if (a.pt_level[i] < 0 || a.pt_level[i] > 30) return fail(c, PLSVO_E_INVALID, "poseopt_stage: feature level out of range");
    for (int i = 0; i < a.n_seg; ++i) if (a.seg_level[i] < 0 || a.seg_level[i] > 30) return fail(c, PLSVO_E_INVALID, "poseopt_stage: feature level out of range");
  }
  int rc;
  Blob blob;
  const size_t o_jobs = blob.add(jobs), o_f = blob.add(f), o_pos = blob.add(pos), o_plev = blob.add(plev), o_line = blob.add(line),
               o_spos = blob.add(spos), o_epos = blob.add(epos), o_slev = blob.add(slev);
  if ((rc = upload_blob(c, c->p_d_blob, blob))) return rc;
  const size_t npt = plev.size(), nsg = slev.size(), nft = std::max(npt + nsg, (size_t)1);
  HIP_TRY(c, c->p_d_ptkeep.ensure(std::max(npt, (size_t)1)));
  HIP_TRY(c, c->p_d_segkeep.ensure(std::max(nsg, (size_t)1)));
  HIP_TRY(c, c->p_d_s32.ensure(nft * sizeof(float)));
  HIP_TRY(c, c->p_d_s64.ensure(nft * 5 * sizeof(double)));
  HIP_TRY(c, c->p_d_state.ensure((size_t)n * sizeof(PoseStateDev)));
  HIP_TRY(c, c->p_d_poses.ensure((size_t)n * 7 * sizeof(double)));
  if (c->p_trace_cap > 0) HIP_TRY(c, c->p_d_log.ensure((size_t)n * c->p_trace_cap * sizeof(plsvo_poseopt_iterlog)));
  PoseBatchDev& b = c->p_b;
  uint8_t* const base = c->p_d_blob.as<uint8_t>();
  b.jobs = reinterpret_cast<const PoseJobDev*>(base + o_jobs); b.state = c->p_d_state.as<PoseStateDev>();
  b.pt_f = reinterpret_cast<const double*>(base + o_f); b.pt_pos = reinterpret_cast<const double*>(base + o_pos);
  b.pt_level = reinterpret_cast<const int*>(base + o_plev);
  b.seg_line = reinterpret_cast<const double*>(base + o_line); b.seg_spos = reinterpret_cast<const double*>(base + o_spos);
  b.seg_epos = reinterpret_cast<const double*>(base + o_epos); b.seg_level = reinterpret_cast<const int*>(base + o_slev);
  b.pt_keep = c->p_d_ptkeep.as<uint8_t>(); b.seg_keep = c->p_d_segkeep.as<uint8_t>();
  b.scratch_f32 = c->p_d_s32.as<float>(); b.scratch_f64 = c->p_d_s64.as<double>();
  b.log = c->p_trace_cap > 0 ? c->p_d_log.as<plsvo_poseopt_iterlog>() : nullptr;
  b.log_cap = c->p_trace_cap; b.n_jobs = n;
  b.order = nullptr; b.work_key = nullptr;   // a new batch: no measured work yet (plsvo_poseopt_run)
  {   // sort-key resolution: the most feature-iterations a frame of this batch can evaluate, in 1024 bins
    long key_max = 1;
    for (int j = 0; j < n; ++j) key_max = std::max(key_max, (long)(jobs[j].n_pts + jobs[j].n_seg) * (long)(std::max(jobs[j].n_iter, 0) + std::max(jobs[j].n_iter_ref, 0)));
    c->p_key_shift = 0;
    while ((key_max >> c->p_key_shift) > 1023) ++c->p_key_shift;
  }
  c->p_jobs.swap(jobs);
  c->p_n = n; c->p_total_pt = (int)npt; c->p_total_seg = (int)nsg;
  c->p_staged = true; c->p_run_seq = 0;
  return PLSVO_OK;
}

extern "C" int plsvo_poseopt_run(plsvo_ctx* c) {
  CTX_CHECK(c);
  RoctxRange range("pose_optimizer");
  if (!c->p_staged) return fail(c, PLSVO_E_STATE, "poseopt_run: no staged batch");
  HIP_TRY(c, hipSetDevice(c->device));
  // Large batches: a 16-lane row per frame, four frames per wave (the serial solve + update of four frames share one instruction stream:
  // poseopt_kernels.hip) while a frame has a few hundred features -- measured on MI355X, 32768 frames: 200 + 80 features 1.86 ms against
  // 2.03 ms for a wave per frame, 500 + 200 features 4.55 against 3.77 (a row then needs 44 dependent feature rounds per pass).  The
  // crossover, measured in round 5 (tools/poseopt_crossover.py, 16384 frames): 420 features 1.31 vs 1.47 ms, 550 features 1.72 vs 1.79,
  // 620 features 1.91 vs 1.80, 700 features 2.19 vs 1.95 -- frames above 580 features keep the wave-per-frame shape.  Too few frames to fill the chip: four waves per frame (the Gauss-Newton loop of
  // one frame is then ~2x shorter).  PLSVO_POSEOPT_THREADS / PLSVO_OPT_POSEOPT_THREADS override (tests and measurements).
  const int cus = c->cu_count > 0 ? c->cu_count : 256;
  const long feats = (long)c->p_total_pt + (long)c->p_total_seg;
  int threads = c->p_n <= 2 * cus ? 256 : (feats <= 580l * c->p_n ? 16 : 64);
  if (c->env_poseopt_threads) threads = c->env_poseopt_threads;
  // batches larger than the resident waves: this launch records what every frame cost, the next one takes them most-expensive first (see
  // plsvo_align_run; for the rows kernel the sort also puts frames that stop together into the same wave)
  const bool reorder = threads != 256 && !c->env_poseopt_no_reorder && !c->p_one_shot &&
                       c->p_n > (c->env_poseopt_reorder_min > 0 ? c->env_poseopt_reorder_min - 1 : (threads == 16 ? 4 : 8) * cus);   // (rows: the grouping pays before the launch has a tail)
  c->p_b.work_key = nullptr;
  if (reorder) {
    if (c->p_d_workkey.cap < (size_t)c->p_n * sizeof(int)) {
      HIP_TRY(c, c->p_d_workkey.ensure((size_t)c->p_n * sizeof(int)));
      HIP_TRY(c, hipMemsetAsync(c->p_d_workkey.p, 0, c->p_d_workkey.cap, c->stream));
    }
    c->p_b.work_key = c->p_d_workkey.as<int>();
  } else c->p_b.order = nullptr;
  EventPair ep{}; prof_begin(c, PLSVO_K_POSEOPT, &ep);
  HIP_TRY(c, launch_pose_opt(c->p_b, c->p_d_poses.as<double>(), threads, c->stream));
  c->p_run_seq = ++c->run_seq;
  prof_end(c, PLSVO_K_POSEOPT, &ep);
  if (reorder) {
    DevBuf& ob = c->p_d_order[c->p_order_next];
    HIP_TRY(c, ob.ensure((size_t)c->p_n * sizeof(int)));
    HIP_TRY(c, launch_align_reorder(c->p_b.work_key, c->p_n, ob.as<int>(), c->p_key_shift, c->stream));
    c->p_b.order = ob.as<int>();
    c->p_order_next ^= 1;
  }
  return PLSVO_OK;
}

extern "C" int plsvo_poseopt_fetch(plsvo_ctx* c, int n, plsvo_poseopt_out* out) {
  CTX_CHECK(c);
  if (!c->p_staged) return fail(c, PLSVO_E_STATE, "poseopt_fetch: no staged batch");
  if (n != c->p_n || !out) return fail(c, PLSVO_E_INVALID, "poseopt_fetch: n does not match the staged batch");
  HIP_TRY(c, hipSetDevice(c->device));
  std::vector<uint8_t> big;   // (large batches, or no pinned memory: download_ranges sizes it)
  const size_t rb[3] = { (size_t)n * sizeof(PoseStateDev), (size_t)std::max(c->p_total_pt, 0), (size_t)std::max(c->p_total_seg, 0) };
  const void* const rs[3] = { c->p_d_state.p, c->p_d_ptkeep.p, c->p_d_segkeep.p };
  const uint8_t* rh[3] = { nullptr, nullptr, nullptr };
  { const int rc_d = download_ranges(c, 3, rs, rb, big, rh); if (rc_d) return rc_d; }
  const PoseStateDev* const st = reinterpret_cast<const PoseStateDev*>(rh[0]);
  const uint8_t* const pk = rh[1]; const uint8_t* const sk = rh[2];
  for (int j = 0; j < n; ++j) {
    const PoseStateDev& s = st[(size_t)j];
    const PoseJobDev& J = c->p_jobs[(size_t)j];
    plsvo_poseopt_out& o = out[j];
    uint8_t* pko = o.pt_keep; uint8_t* sko = o.seg_keep;
    memset(&o, 0, sizeof(o));
    o.pt_keep = pko; o.seg_keep = sko;
    for (int k = 0; k < 7; ++k) o.T_f_w[k] = s.T[k];
    for (int k = 0; k < 36; ++k) o.cov[k] = s.cov[k];
    o.estimated_scale = s.estimated_scale; o.error_init = s.error_init; o.error_final = s.error_final;
    o.num_obs_pt = s.num_obs_pt; o.num_obs_ls = s.num_obs_ls;
    o.iters = s.iters; o.iters_ref = s.iters_ref; o.status = s.status;
    if (pko && J.n_pts > 0) memcpy(pko, pk + J.pt_off, (size_t)J.n_pts);
    if (sko && J.n_seg > 0) memcpy(sko, sk + J.seg_off, (size_t)J.n_seg);
  }
  return PLSVO_OK;
}

extern "C" int plsvo_pose_optimize_batch(plsvo_ctx* c, int n, const plsvo_poseopt_in* in, plsvo_poseopt_out* out) {
  int rc = plsvo_poseopt_stage(c, n, in); if (rc) return rc;
  c->p_one_shot = true;
  rc = plsvo_poseopt_run(c);
  c->p_one_shot = false;
  if (rc) return rc;
  return plsvo_poseopt_fetch(c, n, out);
}
extern "C" int plsvo_pose_optimize(plsvo_ctx* c, const plsvo_poseopt_in* in, plsvo_poseopt_out* out) {
  return plsvo_pose_optimize_batch(c, 1, in, out);
}

extern "C" int plsvo_poseopt_set_trace(plsvo_ctx* c, int max_records_per_job) {
  CTX_CHECK(c);
  if (max_records_per_job < 0) return fail(c, PLSVO_E_INVALID, "set_trace: negative capacity");
  c->p_trace_cap = max_records_per_job;
  c->p_staged = false;
  return PLSVO_OK;
}

extern "C" int plsvo_poseopt_fetch_trace(plsvo_ctx* c, int job, plsvo_poseopt_iterlog* out, int max_records, int* n_records) {
  CTX_CHECK(c);
  if (!c->p_staged || c->p_trace_cap <= 0) return fail(c, PLSVO_E_STATE, "fetch_trace: tracing not enabled for the staged batch");
  if (job < 0 || job >= c->p_n || !out || !n_records) return fail(c, PLSVO_E_INVALID, "fetch_trace: bad arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  PoseStateDev s;
  HIP_TRY(c, hipMemcpyAsync(&s, c->p_d_state.as<PoseStateDev>() + job, sizeof(s), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const int nrec = std::min(std::min(s.log_count, c->p_trace_cap), max_records);
  if (nrec > 0) {
    HIP_TRY(c, hipMemcpyAsync(out, c->p_d_log.as<plsvo_poseopt_iterlog>() + (size_t)job * c->p_trace_cap, (size_t)nrec * sizeof(plsvo_poseopt_iterlog),
                              hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  *n_records = nrec;
  return PLSVO_OK;
}

extern "C" const double* plsvo_poseopt_poses_dev(plsvo_ctx* c) { return (c && c->p_staged) ? c->p_d_poses.as<double>() : nullptr; }

extern "C" int plsvo_poseopt_copy_poses(plsvo_ctx* c, double* d_dst) {
  CTX_CHECK(c);
  if (!c->p_staged || !d_dst) return fail(c, PLSVO_E_STATE, "poseopt_copy_poses: no staged batch / null destination");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipMemcpyAsync(d_dst, c->p_d_poses.p, (size_t)c->p_n * 7 * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  return PLSVO_OK;
}

extern "C" int plsvo_poseopt_work(plsvo_ctx* c, uint64_t* pt_iters, uint64_t* seg_iters) {
  CTX_CHECK(c);
  if (!c->p_staged) return fail(c, PLSVO_E_STATE, "poseopt_work: no staged batch");
  HIP_TRY(c, hipSetDevice(c->device));
  std::vector<PoseStateDev> st((size_t)c->p_n);
  HIP_TRY(c, hipMemcpyAsync(st.data(), c->p_d_state.p, (size_t)c->p_n * sizeof(PoseStateDev), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  uint64_t a = 0, b2 = 0;
  for (auto& s : st) { a += s.pt_iters; b2 += s.seg_iters; }
  if (pt_iters) *pt_iters = a;
  if (seg_iters) *seg_iters = b2;
  return PLSVO_OK;
}

// ---- structure optimisation ------------------------------------------------------------------------
extern "C" int plsvo_structure_optimize(plsvo_ctx* c, const plsvo_structopt_in* in, plsvo_structopt_out* out) {
  CTX_CHECK(c);
  if (!in || !out || in->n_pts < 0 || in->n_seg < 0 || in->n_frames < 0 || in->n_iter_pts < 0 || in->n_iter_segs < 0)
    return fail(c, PLSVO_E_INVALID, "structure_optimize: bad arguments");
  if ((in->n_pts > 0 && (!in->pt_pos || !in->pt_obs_off)) || (in->n_seg > 0 && (!in->seg_spos || !in->seg_epos || !in->seg_obs_off)))
    return fail(c, PLSVO_E_INVALID, "structure_optimize: null landmark array");
  if (in->n_pts + in->n_seg == 0) return PLSVO_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const int np = in->n_pts, ns = in->n_seg;
  const int npo = np > 0 ? in->pt_obs_off[np] : 0, nso = ns > 0 ? in->seg_obs_off[ns] : 0;
  if (npo < 0 || nso < 0 || (npo > 0 && (!in->pt_obs_frame || !in->pt_obs_f)) || (nso > 0 && (!in->seg_obs_frame || !in->seg_obs_sf || !in->seg_obs_ef)) ||
      ((npo > 0 || nso > 0) && (!in->frame_T || in->n_frames <= 0)))
    return fail(c, PLSVO_E_INVALID, "structure_optimize: bad observation tables");
  for (int i = 0; i < npo; ++i) if (in->pt_obs_frame[i] < 0 || in->pt_obs_frame[i] >= in->n_frames) return fail(c, PLSVO_E_INVALID, "structure_optimize: frame index out of range");
  for (int i = 0; i < nso; ++i) if (in->seg_obs_frame[i] < 0 || in->seg_obs_frame[i] >= in->n_frames) return fail(c, PLSVO_E_INVALID, "structure_optimize: frame index out of range");
  // one packed upload: doubles first, then ints
  std::vector<double> d;
  auto putd = [&](const double* p, size_t n) { const size_t o = d.size(); if (n) d.insert(d.end(), p, p + n); return o; };
  const size_t o_T = putd(in->frame_T, (size_t)in->n_frames * 7), o_pp = putd(in->pt_pos, (size_t)np * 3), o_pf = putd(in->pt_obs_f, (size_t)npo * 3);
  const size_t o_ss = putd(in->seg_spos, (size_t)ns * 3), o_se = putd(in->seg_epos, (size_t)ns * 3);
  const size_t o_sf = putd(in->seg_obs_sf, (size_t)nso * 3), o_ef = putd(in->seg_obs_ef, (size_t)nso * 3);
  std::vector<int> iv;
  auto puti = [&](const int32_t* p, size_t n) { const size_t o = iv.size(); if (n) iv.insert(iv.end(), p, p + n); return o; };
  const size_t o_po = puti(in->pt_obs_off, np > 0 ? (size_t)np + 1 : 0), o_pfr = puti(in->pt_obs_frame, (size_t)npo);
  const size_t o_so = puti(in->seg_obs_off, ns > 0 ? (size_t)ns + 1 : 0), o_sfr = puti(in->seg_obs_frame, (size_t)nso);
  const size_t dbytes = (d.size() + 1) * sizeof(double), ibytes = (iv.size() + 1) * sizeof(int);
  HIP_TRY(c, c->s_d_in.ensure(dbytes + ibytes));
  HIP_TRY(c, hipMemcpyAsync(c->s_d_in.p, d.data(), d.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (!iv.empty()) HIP_TRY(c, hipMemcpyAsync(reinterpret_cast<char*>(c->s_d_in.p) + dbytes, iv.data(), iv.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  const size_t out_d = (size_t)(np + 2 * ns) * 3, out_i = (size_t)(np + ns);
  HIP_TRY(c, c->s_d_out.ensure((out_d + 1) * sizeof(double) + (out_i + 1) * sizeof(int)));
  const double* dd = c->s_d_in.as<double>();
  const int* di = reinterpret_cast<const int*>(reinterpret_cast<const char*>(c->s_d_in.p) + dbytes);
  double* od = c->s_d_out.as<double>();
  int* oi = reinterpret_cast<int*>(reinterpret_cast<char*>(c->s_d_out.p) + (out_d + 1) * sizeof(double));
  StructBatchDev s{};
  s.frame_T = dd + o_T; s.pt_pos = dd + o_pp; s.pt_obs_f = dd + o_pf; s.seg_spos = dd + o_ss; s.seg_epos = dd + o_se;
  s.seg_obs_sf = dd + o_sf; s.seg_obs_ef = dd + o_ef;
  s.pt_obs_off = di + o_po; s.pt_obs_frame = di + o_pfr; s.seg_obs_off = di + o_so; s.seg_obs_frame = di + o_sfr;
  s.pt_pos_out = od; s.seg_spos_out = od + (size_t)np * 3; s.seg_epos_out = od + (size_t)(np + ns) * 3;
  s.pt_iters = oi; s.seg_iters = oi + np;
  s.n_pts = np; s.n_seg = ns; s.n_iter_pts = in->n_iter_pts; s.n_iter_segs = in->n_iter_segs;
  {
    EventPair ep{}; prof_begin(c, PLSVO_K_STRUCTOPT, &ep);
    HIP_TRY(c, launch_structopt(s, c->stream));
    prof_end(c, PLSVO_K_STRUCTOPT, &ep);
  }
  std::vector<double> hd(out_d + 1);
  std::vector<int> hi(out_i + 1);
  HIP_TRY(c, hipMemcpyAsync(hd.data(), od, out_d * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(hi.data(), oi, out_i * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (out->pt_pos && np) memcpy(out->pt_pos, hd.data(), (size_t)np * 3 * sizeof(double));
  if (out->seg_spos && ns) memcpy(out->seg_spos, hd.data() + (size_t)np * 3, (size_t)ns * 3 * sizeof(double));
  if (out->seg_epos && ns) memcpy(out->seg_epos, hd.data() + (size_t)(np + ns) * 3, (size_t)ns * 3 * sizeof(double));
  if (out->pt_iters && np) memcpy(out->pt_iters, hi.data(), (size_t)np * sizeof(int));
  if (out->seg_iters && ns) memcpy(out->seg_iters, hi.data() + np, (size_t)ns * sizeof(int));
  return PLSVO_OK;
}

// ---- direct feature matching -------------------------------------------------------------------------
extern "C" int plsvo_match_direct(plsvo_ctx* c, const plsvo_match_in* in, plsvo_match_out* out) {
  CTX_CHECK(c);
  if (!in || !out || in->n < 0 || in->n_frames < 0 || in->n_pyr_levels < 1 || in->align_max_iter < 0)
    return fail(c, PLSVO_E_INVALID, "match_direct: bad arguments");
  const int n = in->n, nf = in->n_frames;
  if (n == 0) return PLSVO_OK;
  if (!c->pyr.base) return fail(c, PLSVO_E_STATE, "match_direct: pyramids not configured");
  if (nf <= 0 || !in->frame_T || !in->frame_slot || !in->cur_frame || !in->ref_frame || !in->ref_px || !in->ref_f || !in->ref_level ||
      !in->ref_type || !in->pos || !in->px_cur)
    return fail(c, PLSVO_E_INVALID, "match_direct: null input array");
  if (in->n_pyr_levels > c->pyr.n_levels) return fail(c, PLSVO_E_INVALID, "match_direct: n_pyr_levels exceeds the configured pyramid");
  if (in->cam.width != c->pyr.w[0] || in->cam.height != c->pyr.h[0]) return fail(c, PLSVO_E_INVALID, "match_direct: camera size does not match the configured pyramid");
  for (int k = 0; k < nf; ++k) if (in->frame_slot[k] < 0 || in->frame_slot[k] >= c->pyr.n_slots) return fail(c, PLSVO_E_CAPACITY, "match_direct: pyramid slot out of range");
  bool any_edgelet = false;
  for (int i = 0; i < n; ++i) {
    if (in->cur_frame[i] < 0 || in->cur_frame[i] >= nf || in->ref_frame[i] < 0 || in->ref_frame[i] >= nf) return fail(c, PLSVO_E_INVALID, "match_direct: frame index out of range");
    if (in->ref_level[i] < 0 || in->ref_level[i] >= c->pyr.n_levels) return fail(c, PLSVO_E_INVALID, "match_direct: ref_level outside the configured pyramid");
    if (in->ref_type[i] == PLSVO_FTR_EDGELET) any_edgelet = true;
    else if (in->ref_type[i] != PLSVO_FTR_CORNER) return fail(c, PLSVO_E_INVALID, "match_direct: unknown feature type");
  }
  if (any_edgelet && !in->ref_grad) return fail(c, PLSVO_E_INVALID, "match_direct: edgelets without ref_grad");
  HIP_TRY(c, hipSetDevice(c->device));
  // one packed upload: doubles, then ints, then bytes
  std::vector<double> d;
  auto putd = [&](const double* p, size_t k) { const size_t o = d.size(); if (p) d.insert(d.end(), p, p + k); else d.resize(d.size() + k, 0.0); return o; };
  const size_t o_T = putd(in->frame_T, (size_t)nf * 7), o_px = putd(in->ref_px, (size_t)n * 2), o_f = putd(in->ref_f, (size_t)n * 3);
  const size_t o_g = putd(in->ref_grad, (size_t)n * 2), o_pos = putd(in->pos, (size_t)n * 3), o_pc = putd(in->px_cur, (size_t)n * 2);
  std::vector<int> iv;
  auto puti = [&](const int32_t* p, size_t k) { const size_t o = iv.size(); iv.insert(iv.end(), p, p + k); return o; };
  const size_t o_slot = puti(in->frame_slot, (size_t)nf), o_cf = puti(in->cur_frame, (size_t)n), o_rf = puti(in->ref_frame, (size_t)n), o_lv = puti(in->ref_level, (size_t)n);
  const size_t dbytes = d.size() * sizeof(double), ibytes = ((iv.size() * sizeof(int) + 15) / 16) * 16, bbytes = (size_t)n;
  HIP_TRY(c, c->s_d_in.ensure(dbytes + ibytes + bbytes + 16));
  char* din = reinterpret_cast<char*>(c->s_d_in.p);
  HIP_TRY(c, hipMemcpyAsync(din, d.data(), dbytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(din + dbytes, iv.data(), iv.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(din + dbytes + ibytes, in->ref_type, bbytes, hipMemcpyHostToDevice, c->stream));
  const size_t out_d = (size_t)n * 2 * sizeof(double), out_i = (size_t)n * 2 * sizeof(int), out_b = (size_t)n;
  HIP_TRY(c, c->s_d_out.ensure(out_d + out_i + out_b + 16));
  char* dout = reinterpret_cast<char*>(c->s_d_out.p);
  const double* dd = reinterpret_cast<const double*>(din);
  const int* di = reinterpret_cast<const int*>(din + dbytes);
  MatchBatchDev b{};
  b.pyr_base = c->pyr.base; b.slot_bytes = c->pyr.slot_bytes; b.width = c->pyr.w[0]; b.height = c->pyr.h[0];
  b.fx = in->cam.fx; b.fy = in->cam.fy; b.cx = in->cam.cx; b.cy = in->cam.cy; b.cam_width = in->cam.width; b.cam_height = in->cam.height;
  b.n = n; b.n_pyr_levels = in->n_pyr_levels; b.align_max_iter = in->align_max_iter;
  b.frame_T = dd + o_T; b.ref_px = dd + o_px; b.ref_f = dd + o_f; b.ref_grad = dd + o_g; b.pos = dd + o_pos; b.px_cur = dd + o_pc;
  b.frame_slot = di + o_slot; b.cur_frame = di + o_cf; b.ref_frame = di + o_rf; b.ref_level = di + o_lv;
  b.ref_type = reinterpret_cast<const uint8_t*>(din + dbytes + ibytes);
  b.px_out = reinterpret_cast<double*>(dout);
  b.search_level = reinterpret_cast<int*>(dout + out_d); b.n_iter = b.search_level + n;
  b.found = reinterpret_cast<uint8_t*>(dout + out_d + out_i);
  {
    EventPair ep{}; prof_begin(c, PLSVO_K_MATCH, &ep);
    HIP_TRY(c, launch_match_direct(b, c->stream));
    prof_end(c, PLSVO_K_MATCH, &ep);
  }
  std::vector<char> h(out_d + out_i + out_b);
  HIP_TRY(c, hipMemcpyAsync(h.data(), dout, h.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (out->px_cur) memcpy(out->px_cur, h.data(), out_d);
  if (out->search_level) memcpy(out->search_level, h.data() + out_d, (size_t)n * sizeof(int));
  if (out->n_iter) memcpy(out->n_iter, h.data() + out_d + (size_t)n * sizeof(int), (size_t)n * sizeof(int));
  if (out->found) memcpy(out->found, h.data() + out_d + out_i, out_b);
  return PLSVO_OK;
}

// ---- reprojection of landmarks (candidates for the direct matcher) ------------------------------------
extern "C" int plsvo_reproject(plsvo_ctx* c, const plsvo_reproject_in* in, plsvo_reproject_out* out) {
  CTX_CHECK(c);
  if (!in || !out || in->n < 0 || in->n_frames < 0 || in->cell_size <= 0 || in->grid_n_cols <= 0 || in->boundary < 0)
    return fail(c, PLSVO_E_INVALID, "reproject: bad arguments");
  const int n = in->n, nf = in->n_frames;
  if (n == 0) return PLSVO_OK;
  if (nf <= 0 || !in->frame_T || !in->frame || !in->pos) return fail(c, PLSVO_E_INVALID, "reproject: null input array");
  for (int i = 0; i < n; ++i) if (in->frame[i] < 0 || in->frame[i] >= nf) return fail(c, PLSVO_E_INVALID, "reproject: frame index out of range");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t dT = (size_t)nf * 7, dP = (size_t)n * 3;
  const size_t dbytes = (dT + dP) * sizeof(double), ibytes = (size_t)n * sizeof(int);
  HIP_TRY(c, c->s_d_in.ensure(dbytes + ibytes + 16));
  char* din = reinterpret_cast<char*>(c->s_d_in.p);
  HIP_TRY(c, hipMemcpyAsync(din, in->frame_T, dT * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(din + dT * sizeof(double), in->pos, dP * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(din + dbytes, in->frame, ibytes, hipMemcpyHostToDevice, c->stream));
  const size_t out_d = (size_t)n * 2 * sizeof(double), out_i = (size_t)n * sizeof(int);
  HIP_TRY(c, c->s_d_out.ensure(out_d + out_i + 16));
  char* dout = reinterpret_cast<char*>(c->s_d_out.p);
  ReprojBatchDev b{};
  b.fx = in->cam.fx; b.fy = in->cam.fy; b.cx = in->cam.cx; b.cy = in->cam.cy; b.cam_width = in->cam.width; b.cam_height = in->cam.height;
  b.n = n; b.cell_size = in->cell_size; b.grid_n_cols = in->grid_n_cols; b.boundary = in->boundary;
  b.frame_T = reinterpret_cast<const double*>(din); b.pos = b.frame_T + dT; b.frame = reinterpret_cast<const int*>(din + dbytes);
  b.px = reinterpret_cast<double*>(dout); b.cell = reinterpret_cast<int*>(dout + out_d);
  {
    EventPair ep{}; prof_begin(c, PLSVO_K_MATCH, &ep);
    HIP_TRY(c, launch_reproject(b, c->stream));
    prof_end(c, PLSVO_K_MATCH, &ep);
  }
  std::vector<char> h(out_d + out_i);
  HIP_TRY(c, hipMemcpyAsync(h.data(), dout, h.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (out->px) memcpy(out->px, h.data(), out_d);
  if (out->cell) memcpy(out->cell, h.data() + out_d, out_i);
  return PLSVO_OK;
}

// ---- resident frame step ------------------------------------------------------------------------------
namespace {
// carve typed arrays out of one device allocation (256-byte aligned sections)
struct Carver {
  size_t off = 0;
  template <typename T> size_t take(size_t n) { const size_t o = (off + 255) & ~(size_t)255; off = o + std::max(n, (size_t)1) * sizeof(T); return o; }
};
}  // namespace

extern "C" int plsvo_chain_stage(plsvo_ctx* c, int n, const plsvo_chain_in* in, const plsvo_chain_params* pr) {
  CTX_CHECK(c);
  if (n <= 0 || !in || !pr) return fail(c, PLSVO_E_INVALID, "chain_stage: bad arguments");
  if (pr->cell_size <= 0 || pr->n_pyr_levels < 1 || pr->align_max_iter < 0 || pr->poseopt_n_iter < 0 || pr->max_fts < 0)
    return fail(c, PLSVO_E_INVALID, "chain_stage: bad parameters");
  if (!c->pyr.base) return fail(c, PLSVO_E_STATE, "chain_stage: pyramids not configured");
  if (pr->n_pyr_levels > c->pyr.n_levels) return fail(c, PLSVO_E_INVALID, "chain_stage: n_pyr_levels exceeds the configured pyramid");
  if (pr->cam.width != c->pyr.w[0] || pr->cam.height != c->pyr.h[0]) return fail(c, PLSVO_E_INVALID, "chain_stage: camera size does not match the configured pyramid");
  c->ch_staged = false;
  // the alignment part is an ordinary staged alignment batch
  std::vector<plsvo_align_in> ain((size_t)n);
  for (int j = 0; j < n; ++j) ain[(size_t)j] = in[j].align;
  int rc = plsvo_align_stage(c, n, ain.data()); if (rc) return rc;
  const int grid_n_cols = (pr->cam.width + pr->cell_size - 1) / pr->cell_size, grid_n_rows = (pr->cam.height + pr->cell_size - 1) / pr->cell_size;
  const int n_cells = grid_n_cols * grid_n_rows;
  const bool seg_grid = pr->cell_rule && pr->seg_cell_size > 0;
  if (pr->seg_cell_size < 0 || (seg_grid && pr->max_fts_segs < 0)) return fail(c, PLSVO_E_INVALID, "chain_stage: bad segment-grid parameters");
  const int seg_n_cols = seg_grid ? (pr->cam.width + pr->seg_cell_size - 1) / pr->seg_cell_size : 0;
  const int seg_n_cells = seg_grid ? seg_n_cols * ((pr->cam.height + pr->seg_cell_size - 1) / pr->seg_cell_size) : 0;
  if (pr->cell_rule && ((size_t)n_cells + (size_t)seg_n_cells) * sizeof(int) > 60000) return fail(c, PLSVO_E_CAPACITY, "chain_stage: reprojection grid too fine for the selection kernel");
  if (seg_grid && pr->seg_cell_order)
    for (int k = 0; k < seg_n_cells; ++k) if (pr->seg_cell_order[k] < 0 || pr->seg_cell_order[k] >= seg_n_cells) return fail(c, PLSVO_E_INVALID, "chain_stage: seg_cell_order entry out of range");
  if (pr->cell_rule && pr->cell_order)
    for (int k = 0; k < n_cells; ++k) if (pr->cell_order[k] < 0 || pr->cell_order[k] >= n_cells) return fail(c, PLSVO_E_INVALID, "chain_stage: cell_order entry out of range");
  std::vector<ChainJobDev> jobs((size_t)n);
  std::vector<double> pos, rpx, rf, rgrad;
  std::vector<int> rlevel, cand_job, frame_cur, frame_ref, order, seg_order;
  std::vector<uint8_t> rtype, active;
  bool any_active = false, any_edgelet = false;
  int npt_total = 0, nseg_total = 0;
  for (int j = 0; j < n; ++j) {
    const plsvo_chain_in& a = in[j];
    const int nc = a.n_cand_pt + 2 * a.n_cand_seg;
    if (a.n_cand_pt < 0 || a.n_cand_seg < 0) return fail(c, PLSVO_E_INVALID, "chain_stage: negative candidate count");
    if (nc > 0 && (!a.pos || !a.ref_px || !a.ref_f || !a.ref_level)) return fail(c, PLSVO_E_INVALID, "chain_stage: null candidate array");
    if (a.kf_slot < 0 || a.kf_slot >= c->pyr.n_slots) return fail(c, PLSVO_E_CAPACITY, "chain_stage: keyframe slot out of range");
    ChainJobDev& J = jobs[(size_t)j];
    for (int k = 0; k < 7; ++k) { J.T_prev[k] = a.T_prev_w[k]; J.T_kf[k] = a.T_kf_w[k]; }
    J.kf_slot = a.kf_slot; J.cur_slot = a.align.cur_slot;
    J.cand_off = (int)cand_job.size(); J.n_pt = a.n_cand_pt; J.n_seg = a.n_cand_seg;
    J.po_pt_off = npt_total; J.po_seg_off = nseg_total; J.reserved0 = 0;
    npt_total += a.n_cand_pt; nseg_total += (seg_grid ? 2 : 1) * a.n_cand_seg;   // (a segment that wins both of its cells is a feature twice)
    pos.insert(pos.end(), a.pos, a.pos + 3 * (size_t)nc);
    rpx.insert(rpx.end(), a.ref_px, a.ref_px + 2 * (size_t)nc);
    rf.insert(rf.end(), a.ref_f, a.ref_f + 3 * (size_t)nc);
    for (int k = 0; k < nc; ++k) {
      if (a.ref_level[k] < 0 || a.ref_level[k] >= c->pyr.n_levels) return fail(c, PLSVO_E_INVALID, "chain_stage: ref_level outside the configured pyramid");
      const uint8_t ty = (a.ref_type && k < a.n_cand_pt) ? a.ref_type[k] : (uint8_t)PLSVO_FTR_CORNER;
      if (ty == PLSVO_FTR_EDGELET) any_edgelet = true; else if (ty != PLSVO_FTR_CORNER) return fail(c, PLSVO_E_INVALID, "chain_stage: unknown feature type");
      if (ty == PLSVO_FTR_EDGELET && !a.ref_grad) return fail(c, PLSVO_E_INVALID, "chain_stage: edgelets without ref_grad");
      rtype.push_back(ty);
      const bool has_grad = a.ref_grad && k < a.n_cand_pt;   // ref_type / ref_grad are sized by the POINT candidates (header); end points: 0
      rgrad.push_back(has_grad ? a.ref_grad[2 * k] : 0.0); rgrad.push_back(has_grad ? a.ref_grad[2 * k + 1] : 0.0);
      rlevel.push_back(a.ref_level[k]);
      active.push_back(a.active ? (a.active[k] ? 1 : 0) : 1);
      if (a.active) any_active = true;
      cand_job.push_back(j); frame_ref.push_back(2 * j); frame_cur.push_back(2 * j + 1);
    }
  }
  (void)any_edgelet;
  const size_t NC = cand_job.size();
  if (pr->cell_rule) { order.resize((size_t)n_cells); for (int k = 0; k < n_cells; ++k) order[(size_t)k] = pr->cell_order ? pr->cell_order[k] : k; }
  if (seg_grid) { seg_order.resize((size_t)seg_n_cells); for (int k = 0; k < seg_n_cells; ++k) seg_order[(size_t)k] = pr->seg_cell_order ? pr->seg_cell_order[k] : k; }
  Blob blob;
  const size_t o_jobs = blob.add(jobs), o_pos = blob.add(pos), o_rpx = blob.add(rpx), o_rf = blob.add(rf), o_rgrad = blob.add(rgrad), o_rlevel = blob.add(rlevel),
               o_cjob = blob.add(cand_job), o_fcur = blob.add(frame_cur), o_fref = blob.add(frame_ref), o_rtype = blob.add(rtype), o_active = blob.add(active),
               o_order = blob.add(order), o_sorder = blob.add(seg_order);
  if ((rc = upload_blob(c, c->ch_d_blob, blob))) return rc;
  // work arrays
  Carver w;
  const size_t w_frameT = w.take<double>((size_t)n * 14), w_fslot = w.take<int>((size_t)n * 2), w_px = w.take<double>(NC * 2), w_cell = w.take<int>(NC),
               w_act = w.take<uint8_t>(NC), w_mpx = w.take<double>(NC * 2), w_found = w.take<uint8_t>(NC), w_slevel = w.take<int>(NC), w_niter = w.take<int>(NC),
               w_selpt = w.take<int>((size_t)npt_total), w_selseg = w.take<int>((size_t)nseg_total), w_nsel = w.take<int>((size_t)n * 2);
  HIP_TRY(c, c->ch_d_work.ensure(w.off + 256));
  Carver p;
  const size_t p_jobs = p.take<PoseJobDev>((size_t)n), p_ptf = p.take<double>((size_t)npt_total * 3), p_ptpos = p.take<double>((size_t)npt_total * 3),
               p_ptlev = p.take<int>((size_t)npt_total), p_line = p.take<double>((size_t)nseg_total * 3), p_spos = p.take<double>((size_t)nseg_total * 3),
               p_epos = p.take<double>((size_t)nseg_total * 3), p_slev = p.take<int>((size_t)nseg_total);
  HIP_TRY(c, c->ch_d_po.ensure(p.off + 256));
  const size_t nft = std::max((size_t)npt_total + (size_t)nseg_total, (size_t)1);
  HIP_TRY(c, c->ch_d_ptkeep.ensure(std::max((size_t)npt_total, (size_t)1)));
  HIP_TRY(c, c->ch_d_segkeep.ensure(std::max((size_t)nseg_total, (size_t)1)));
  HIP_TRY(c, c->ch_d_s32.ensure(nft * sizeof(float)));
  HIP_TRY(c, c->ch_d_s64.ensure(nft * 5 * sizeof(double)));
  HIP_TRY(c, c->ch_d_state.ensure((size_t)n * sizeof(PoseStateDev)));
  HIP_TRY(c, c->ch_d_poses.ensure((size_t)n * 7 * sizeof(double)));
  uint8_t* const B = c->ch_d_blob.as<uint8_t>();
  uint8_t* const Wk = c->ch_d_work.as<uint8_t>();
  uint8_t* const Po = c->ch_d_po.as<uint8_t>();
  ChainBatchDev& b = c->ch_b;
  b = ChainBatchDev{};
  b.jobs = reinterpret_cast<const ChainJobDev*>(B + o_jobs); b.n_jobs = n; b.n_cand = (int)NC;
  b.align_poses = c->a_d_poses.as<double>();
  b.frame_T = reinterpret_cast<double*>(Wk + w_frameT); b.frame_slot = reinterpret_cast<int*>(Wk + w_fslot);
  b.cand_job = reinterpret_cast<const int*>(B + o_cjob); b.pos = reinterpret_cast<const double*>(B + o_pos);
  b.active_in = any_active ? B + o_active : nullptr;
  b.cell = reinterpret_cast<const int*>(Wk + w_cell); b.active = Wk + w_act;
  b.m_px = reinterpret_cast<const double*>(Wk + w_mpx); b.found = Wk + w_found; b.search_level = reinterpret_cast<const int*>(Wk + w_slevel);
  b.fx = pr->cam.fx; b.fy = pr->cam.fy; b.cx = pr->cam.cx; b.cy = pr->cam.cy;
  b.n_cells = n_cells; b.cell_rule = pr->cell_rule ? 1 : 0; b.max_fts = pr->max_fts;
  b.cell_order = pr->cell_rule ? reinterpret_cast<const int*>(B + o_order) : nullptr;
  b.proj_px = reinterpret_cast<const double*>(Wk + w_px);
  b.seg_cell_size = seg_grid ? pr->seg_cell_size : 0; b.seg_n_cols = seg_n_cols; b.seg_n_cells = seg_n_cells; b.max_fts_segs = pr->max_fts_segs;
  b.seg_cell_order = seg_grid ? reinterpret_cast<const int*>(B + o_sorder) : nullptr;
  b.po_jobs = reinterpret_cast<PoseJobDev*>(Po + p_jobs);
  b.pt_f = reinterpret_cast<double*>(Po + p_ptf); b.pt_pos = reinterpret_cast<double*>(Po + p_ptpos); b.pt_level = reinterpret_cast<int*>(Po + p_ptlev);
  b.seg_line = reinterpret_cast<double*>(Po + p_line); b.seg_spos = reinterpret_cast<double*>(Po + p_spos); b.seg_epos = reinterpret_cast<double*>(Po + p_epos);
  b.seg_level = reinterpret_cast<int*>(Po + p_slev);
  b.sel_pt = reinterpret_cast<int*>(Wk + w_selpt); b.sel_seg = reinterpret_cast<int*>(Wk + w_selseg); b.n_sel = reinterpret_cast<int*>(Wk + w_nsel);
  b.reproj_thresh = pr->reproj_thresh; b.po_n_iter = pr->poseopt_n_iter; b.ldlt_flavour = c->ldlt_flavour;
  ReprojBatchDev& r = c->ch_reproj;
  r = ReprojBatchDev{};
  r.fx = pr->cam.fx; r.fy = pr->cam.fy; r.cx = pr->cam.cx; r.cy = pr->cam.cy; r.cam_width = pr->cam.width; r.cam_height = pr->cam.height;
  r.n = (int)NC; r.cell_size = pr->cell_size; r.grid_n_cols = grid_n_cols; r.boundary = 8;
  r.frame_T = b.frame_T; r.frame = reinterpret_cast<const int*>(B + o_fcur); r.pos = b.pos;
  r.px = reinterpret_cast<double*>(Wk + w_px); r.cell = reinterpret_cast<int*>(Wk + w_cell);
  MatchBatchDev& m = c->ch_match;
  m = MatchBatchDev{};
  m.pyr_base = c->pyr.base; m.slot_bytes = c->pyr.slot_bytes; m.width = c->pyr.w[0]; m.height = c->pyr.h[0];
  m.fx = pr->cam.fx; m.fy = pr->cam.fy; m.cx = pr->cam.cx; m.cy = pr->cam.cy; m.cam_width = pr->cam.width; m.cam_height = pr->cam.height;
  m.n = (int)NC; m.n_pyr_levels = pr->n_pyr_levels; m.align_max_iter = pr->align_max_iter;
  m.frame_T = b.frame_T; m.frame_slot = b.frame_slot; m.cur_frame = r.frame; m.ref_frame = reinterpret_cast<const int*>(B + o_fref);
  m.ref_px = reinterpret_cast<const double*>(B + o_rpx); m.ref_f = reinterpret_cast<const double*>(B + o_rf); m.ref_level = reinterpret_cast<const int*>(B + o_rlevel);
  m.ref_type = B + o_rtype; m.ref_grad = reinterpret_cast<const double*>(B + o_rgrad); m.pos = b.pos; m.px_cur = r.px;
  m.px_out = reinterpret_cast<double*>(Wk + w_mpx); m.found = Wk + w_found; m.search_level = reinterpret_cast<int*>(Wk + w_slevel); m.n_iter = reinterpret_cast<int*>(Wk + w_niter);
  m.active = b.active;
  PoseBatchDev& q = c->ch_pose;
  q = PoseBatchDev{};
  q.jobs = b.po_jobs; q.state = c->ch_d_state.as<PoseStateDev>();
  q.pt_f = b.pt_f; q.pt_pos = b.pt_pos; q.pt_level = b.pt_level; q.seg_line = b.seg_line; q.seg_spos = b.seg_spos; q.seg_epos = b.seg_epos; q.seg_level = b.seg_level;
  q.pt_keep = c->ch_d_ptkeep.as<uint8_t>(); q.seg_keep = c->ch_d_segkeep.as<uint8_t>();
  q.scratch_f32 = c->ch_d_s32.as<float>(); q.scratch_f64 = c->ch_d_s64.as<double>();
  q.log = nullptr; q.log_cap = 0; q.n_jobs = n;
  c->ch_jobs.swap(jobs);
  c->ch_n = n; c->ch_ncand = (int)NC; c->ch_npt_cap = npt_total; c->ch_nseg_cap = nseg_total;
  c->ch_staged = true; c->ch_run_seq = 0;
  return PLSVO_OK;
}

extern "C" int plsvo_chain_run(plsvo_ctx* c) {
  CTX_CHECK(c);
  RoctxRange range("frame_step");
  if (!c->ch_staged || !c->a_staged || c->a_n != c->ch_n) return fail(c, PLSVO_E_STATE, "chain_run: no staged frame step (or the alignment batch was re-staged since)");
  int rc = plsvo_align_run(c); if (rc) return rc;
  HIP_TRY(c, launch_chain_pose(c->ch_b, c->stream));
  { EventPair ep{}; prof_begin(c, PLSVO_K_MATCH, &ep);
    HIP_TRY(c, launch_reproject(c->ch_reproj, c->stream));
    HIP_TRY(c, launch_chain_active(c->ch_b, c->stream));
    HIP_TRY(c, launch_match_direct(c->ch_match, c->stream));
    HIP_TRY(c, launch_chain_select(c->ch_b, c->stream));
    prof_end(c, PLSVO_K_MATCH, &ep); }
  const int cus = c->cu_count > 0 ? c->cu_count : 256;
  int threads = c->ch_n <= 2 * cus ? 256 : ((long)c->ch_b.n_cand <= 580l * c->ch_n ? 16 : 64);   // (selected features <= candidates; see plsvo_poseopt_run)
  if (c->env_poseopt_threads) threads = c->env_poseopt_threads;
  EventPair ep{}; prof_begin(c, PLSVO_K_POSEOPT, &ep);
  HIP_TRY(c, launch_pose_opt(c->ch_pose, c->ch_d_poses.as<double>(), threads, c->stream));
  prof_end(c, PLSVO_K_POSEOPT, &ep);
  c->ch_run_seq = ++c->run_seq;
  return PLSVO_OK;
}

extern "C" const double* plsvo_chain_poses_dev(plsvo_ctx* c) { return (c && c->ch_staged) ? c->ch_d_poses.as<double>() : nullptr; }

extern "C" int plsvo_chain_fetch(plsvo_ctx* c, int n, plsvo_chain_out* out) {
  CTX_CHECK(c);
  if (!c->ch_staged) return fail(c, PLSVO_E_STATE, "chain_fetch: no staged frame step");
  if (n != c->ch_n || !out) return fail(c, PLSVO_E_INVALID, "chain_fetch: n does not match the staged batch");
  // the alignment results through the ordinary fetch (the caller's seg_alive_out buffers are honoured)
  std::vector<plsvo_align_out> ao((size_t)n);
  for (int j = 0; j < n; ++j) ao[(size_t)j].seg_alive_out = out[j].align.seg_alive_out;
  int rc = plsvo_align_fetch(c, n, ao.data()); if (rc) return rc;
  const size_t NC = (size_t)c->ch_ncand, NP = (size_t)c->ch_npt_cap, NS = (size_t)c->ch_nseg_cap;
  std::vector<PoseStateDev> st((size_t)n);
  std::vector<uint8_t> found(std::max(NC, (size_t)1)), pk(std::max(NP, (size_t)1)), sk(std::max(NS, (size_t)1));
  std::vector<double> mpx(std::max(NC * 2, (size_t)1));
  std::vector<int> slev(std::max(NC, (size_t)1)), selp(std::max(NP, (size_t)1)), sels(std::max(NS, (size_t)1)), nsel((size_t)n * 2);
  const ChainBatchDev& b = c->ch_b;
  HIP_TRY(c, hipMemcpyAsync(st.data(), c->ch_d_state.p, (size_t)n * sizeof(PoseStateDev), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(nsel.data(), b.n_sel, (size_t)n * 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  if (NC) {
    HIP_TRY(c, hipMemcpyAsync(found.data(), b.found, NC, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(mpx.data(), b.m_px, NC * 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(slev.data(), b.search_level, NC * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  }
  if (NP) {
    HIP_TRY(c, hipMemcpyAsync(pk.data(), c->ch_d_ptkeep.p, NP, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(selp.data(), b.sel_pt, NP * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  }
  if (NS) {
    HIP_TRY(c, hipMemcpyAsync(sk.data(), c->ch_d_segkeep.p, NS, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(sels.data(), b.sel_seg, NS * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int j = 0; j < n; ++j) {
    const ChainJobDev& J = c->ch_jobs[(size_t)j];
    const PoseStateDev& s = st[(size_t)j];
    plsvo_chain_out& o = out[j];
    o.align = ao[(size_t)j];
    uint8_t* pko = o.pose.pt_keep; uint8_t* sko = o.pose.seg_keep;
    memset(&o.pose, 0, sizeof(o.pose));
    o.pose.pt_keep = pko; o.pose.seg_keep = sko;
    for (int k = 0; k < 7; ++k) o.pose.T_f_w[k] = s.T[k];
    for (int k = 0; k < 36; ++k) o.pose.cov[k] = s.cov[k];
    o.pose.estimated_scale = s.estimated_scale; o.pose.error_init = s.error_init; o.pose.error_final = s.error_final;
    o.pose.num_obs_pt = s.num_obs_pt; o.pose.num_obs_ls = s.num_obs_ls;
    o.pose.iters = s.iters; o.pose.iters_ref = s.iters_ref; o.pose.status = s.status;
    o.n_sel_pt = nsel[(size_t)j * 2]; o.n_sel_seg = nsel[(size_t)j * 2 + 1];
    const int nc = J.n_pt + 2 * J.n_seg;
    if (pko && o.n_sel_pt > 0) memcpy(pko, pk.data() + J.po_pt_off, (size_t)o.n_sel_pt);
    if (sko && o.n_sel_seg > 0) memcpy(sko, sk.data() + J.po_seg_off, (size_t)o.n_sel_seg);
    if (o.found && nc) memcpy(o.found, found.data() + J.cand_off, (size_t)nc);
    if (o.px && nc) memcpy(o.px, mpx.data() + 2 * (size_t)J.cand_off, (size_t)nc * 2 * sizeof(double));
    if (o.search_level && nc) memcpy(o.search_level, slev.data() + J.cand_off, (size_t)nc * sizeof(int));
    if (o.sel_pt && o.n_sel_pt > 0) memcpy(o.sel_pt, selp.data() + J.po_pt_off, (size_t)o.n_sel_pt * sizeof(int));
    if (o.sel_seg && o.n_sel_seg > 0) memcpy(o.sel_seg, sels.data() + J.po_seg_off, (size_t)o.n_sel_seg * sizeof(int));
  }
  return PLSVO_OK;
}

extern "C" int plsvo_frame_step_batch(plsvo_ctx* c, int n, const plsvo_chain_in* in, const plsvo_chain_params* params, plsvo_chain_out* out) {
  int rc = plsvo_chain_stage(c, n, in, params); if (rc) return rc;
  rc = plsvo_chain_run(c); if (rc) return rc;
  return plsvo_chain_fetch(c, n, out);
}

// ---- trajectory record (host only; app/run_pipeline.cpp:425-451) ---------------------------------------
extern "C" int plsvo_trajectory_record(const double T_f_w[7], const double cov[36], double out7[7]) {
  if (!T_f_w || !cov || !out7) return 0;
  bool skip_frame = false;
  for (int i = 0; i < 36; ++i) if (!((1.e-16 < std::fabs(cov[i])) && (std::fabs(cov[i]) < 1.e+16))) skip_frame = true;
  const SE3d W = se3_inv(se3_load(T_f_w));   // world_transf = T_f_w_.inverse()
  if ((W.t[0] == 0. && W.t[1] == 0. && W.t[2] == 0.) && (W.q.x == -0. && W.q.y == -0. && W.q.z == -0. && W.q.w == 1.)) skip_frame = true;
  out7[0] = W.t[0]; out7[1] = W.t[1]; out7[2] = W.t[2]; out7[3] = W.q.x; out7[4] = W.q.y; out7[5] = W.q.z; out7[6] = W.q.w;
  return skip_frame ? 0 : 1;
}

// ---- depth-filter seed update ----------------------------------------------------------------------------
extern "C" int plsvo_update_seeds(plsvo_ctx* c, const plsvo_seeds_in* in, plsvo_seeds_out* out) {
  CTX_CHECK(c);
  if (!in || !out || in->n_pt < 0 || in->n_seg < 0 || in->n_frames < 0 || in->n_pyr_levels < 1 || in->align_max_iter < 0 || in->max_epi_search_steps < 0)
    return fail(c, PLSVO_E_INVALID, "update_seeds: bad arguments");
  const int np = in->n_pt, ns = in->n_seg, nf = in->n_frames;
  if (np + ns == 0) return PLSVO_OK;
  if (!c->pyr.base) return fail(c, PLSVO_E_STATE, "update_seeds: pyramids not configured");
  if (nf <= 0 || !in->frame_T || !in->frame_slot) return fail(c, PLSVO_E_INVALID, "update_seeds: no frames");
  if (in->n_pyr_levels > c->pyr.n_levels) return fail(c, PLSVO_E_INVALID, "update_seeds: n_pyr_levels exceeds the configured pyramid");
  if (in->cam.width != c->pyr.w[0] || in->cam.height != c->pyr.h[0]) return fail(c, PLSVO_E_INVALID, "update_seeds: camera size does not match the configured pyramid");
  if (np > 0 && (!in->pt_ref_frame || !in->pt_cur_frame || !in->pt_px || !in->pt_f || !in->pt_level || !in->pt_type || !in->pt_a || !in->pt_b || !in->pt_mu || !in->pt_z_range || !in->pt_sigma2))
    return fail(c, PLSVO_E_INVALID, "update_seeds: null point-seed array");
  if (ns > 0 && (!in->seg_ref_frame || !in->seg_cur_frame || !in->seg_px || !in->seg_f || !in->seg_sf || !in->seg_ef || !in->seg_level || !in->seg_a || !in->seg_b ||
                 !in->seg_mu_s || !in->seg_mu_e || !in->seg_z_range_s || !in->seg_z_range_e || !in->seg_sigma2_s || !in->seg_sigma2_e))
    return fail(c, PLSVO_E_INVALID, "update_seeds: null line-seed array");
  for (int k = 0; k < nf; ++k) if (in->frame_slot[k] < 0 || in->frame_slot[k] >= c->pyr.n_slots) return fail(c, PLSVO_E_CAPACITY, "update_seeds: pyramid slot out of range");
  bool any_edgelet = false;
  for (int i = 0; i < np; ++i) {
    if (in->pt_ref_frame[i] < 0 || in->pt_ref_frame[i] >= nf || in->pt_cur_frame[i] < 0 || in->pt_cur_frame[i] >= nf) return fail(c, PLSVO_E_INVALID, "update_seeds: frame index out of range");
    if (in->pt_level[i] < 0 || in->pt_level[i] >= c->pyr.n_levels) return fail(c, PLSVO_E_INVALID, "update_seeds: feature level outside the configured pyramid");
    if (in->pt_type[i] == PLSVO_FTR_EDGELET) any_edgelet = true;
  }
  if (any_edgelet && in->edgelet_filtering && !in->pt_grad) return fail(c, PLSVO_E_INVALID, "update_seeds: edgelets without pt_grad");
  for (int i = 0; i < ns; ++i) {
    if (in->seg_ref_frame[i] < 0 || in->seg_ref_frame[i] >= nf || in->seg_cur_frame[i] < 0 || in->seg_cur_frame[i] >= nf) return fail(c, PLSVO_E_INVALID, "update_seeds: frame index out of range");
    if (in->seg_level[i] < 0 || in->seg_level[i] >= c->pyr.n_levels) return fail(c, PLSVO_E_INVALID, "update_seeds: feature level outside the configured pyramid");
  }
  HIP_TRY(c, hipSetDevice(c->device));
  // one packed upload per element type
  std::vector<double> d; std::vector<int> iv; std::vector<float> fv;
  auto putd = [&](const double* p, size_t k) { const size_t o = d.size(); if (p) d.insert(d.end(), p, p + k); else d.resize(d.size() + k, 0.0); return o; };
  auto puti = [&](const int32_t* p, size_t k) { const size_t o = iv.size(); iv.insert(iv.end(), p, p + k); return o; };
  auto putf = [&](const float* p, size_t k) { const size_t o = fv.size(); fv.insert(fv.end(), p, p + k); return o; };
  const size_t NP = (size_t)np, NS = (size_t)ns;
  const size_t o_T = putd(in->frame_T, (size_t)nf * 7), o_ppx = putd(in->pt_px, NP * 2), o_pf = putd(in->pt_f, NP * 3), o_pg = putd(in->pt_grad, NP * 2);
  const size_t o_spx = putd(in->seg_px, NS * 2), o_sf0 = putd(in->seg_f, NS * 3), o_ssf = putd(in->seg_sf, NS * 3), o_sef = putd(in->seg_ef, NS * 3);
  const size_t o_slot = puti(in->frame_slot, (size_t)nf), o_prf = puti(in->pt_ref_frame, NP), o_pcf = puti(in->pt_cur_frame, NP), o_plv = puti(in->pt_level, NP);
  const size_t o_srf = puti(in->seg_ref_frame, NS), o_scf = puti(in->seg_cur_frame, NS), o_slv = puti(in->seg_level, NS);
  const size_t o_pa = putf(in->pt_a, NP), o_pb = putf(in->pt_b, NP), o_pmu = putf(in->pt_mu, NP), o_pzr = putf(in->pt_z_range, NP), o_ps2 = putf(in->pt_sigma2, NP);
  const size_t o_sa = putf(in->seg_a, NS), o_sb = putf(in->seg_b, NS), o_smus = putf(in->seg_mu_s, NS), o_smue = putf(in->seg_mu_e, NS);
  const size_t o_szrs = putf(in->seg_z_range_s, NS), o_szre = putf(in->seg_z_range_e, NS), o_ss2s = putf(in->seg_sigma2_s, NS), o_ss2e = putf(in->seg_sigma2_e, NS);
  auto pad16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
  const size_t dbytes = pad16(d.size() * sizeof(double)), ibytes = pad16(iv.size() * sizeof(int)), fbytes = pad16(fv.size() * sizeof(float)), bbytes = pad16(NP);
  HIP_TRY(c, c->s_d_in.ensure(dbytes + ibytes + fbytes + bbytes + 16));
  char* din = reinterpret_cast<char*>(c->s_d_in.p);
  HIP_TRY(c, hipMemcpyAsync(din, d.data(), d.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(din + dbytes, iv.data(), iv.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  if (!fv.empty()) HIP_TRY(c, hipMemcpyAsync(din + dbytes + ibytes, fv.data(), fv.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
  if (np > 0) HIP_TRY(c, hipMemcpyAsync(din + dbytes + ibytes + fbytes, in->pt_type, NP, hipMemcpyHostToDevice, c->stream));
  // outputs: doubles, then floats, then ints
  const size_t od_n = NP * 6 + NS * 8, of_n = NP * 4 + NS * 6, oi_n = NP + NS;
  const size_t od_b = pad16(od_n * sizeof(double)), of_b = pad16(of_n * sizeof(float)), oi_b = pad16(oi_n * sizeof(int));
  HIP_TRY(c, c->s_d_out.ensure(od_b + of_b + oi_b + 16));
  char* dout = reinterpret_cast<char*>(c->s_d_out.p);
  const double* dd = reinterpret_cast<const double*>(din);
  const int* di = reinterpret_cast<const int*>(din + dbytes);
  const float* df = reinterpret_cast<const float*>(din + dbytes + ibytes);
  double* od = reinterpret_cast<double*>(dout);
  float* of = reinterpret_cast<float*>(dout + od_b);
  int* oi = reinterpret_cast<int*>(dout + od_b + of_b);
  SeedsBatchDev b{};
  b.pyr_base = c->pyr.base; b.slot_bytes = c->pyr.slot_bytes; b.width = c->pyr.w[0]; b.height = c->pyr.h[0];
  b.fx = in->cam.fx; b.fy = in->cam.fy; b.cx = in->cam.cx; b.cy = in->cam.cy; b.cam_width = in->cam.width; b.cam_height = in->cam.height;
  b.n_pyr_levels = in->n_pyr_levels; b.align_max_iter = in->align_max_iter; b.max_epi_search_steps = in->max_epi_search_steps;
  b.edgelet_filtering = in->edgelet_filtering; b.edgelet_max_angle = in->edgelet_max_angle;
  b.px_error_angle = std::atan(in->px_noise / (2.0 * std::fabs(in->cam.fx))) * 2.0;   // depth_filter.cpp:279-280 (host libm, like the reference)
  b.convergence_sigma2_thresh = in->convergence_sigma2_thresh;
  b.n_pt = np; b.n_seg = ns;
  b.frame_T = dd + o_T; b.frame_slot = di + o_slot;
  b.pt_ref_frame = di + o_prf; b.pt_cur_frame = di + o_pcf; b.pt_level = di + o_plv; b.pt_px = dd + o_ppx; b.pt_f = dd + o_pf;
  b.pt_grad = in->pt_grad ? dd + o_pg : nullptr; b.pt_type = reinterpret_cast<const uint8_t*>(din + dbytes + ibytes + fbytes);
  b.pt_a = df + o_pa; b.pt_b = df + o_pb; b.pt_mu = df + o_pmu; b.pt_z_range = df + o_pzr; b.pt_sigma2 = df + o_ps2;
  b.seg_ref_frame = di + o_srf; b.seg_cur_frame = di + o_scf; b.seg_level = di + o_slv;
  b.seg_px = dd + o_spx; b.seg_f = dd + o_sf0; b.seg_sf = dd + o_ssf; b.seg_ef = dd + o_sef;
  b.seg_a = df + o_sa; b.seg_b = df + o_sb; b.seg_mu_s = df + o_smus; b.seg_mu_e = df + o_smue; b.seg_z_range_s = df + o_szrs;
  b.seg_z_range_e = df + o_szre; b.seg_sigma2_s = df + o_ss2s; b.seg_sigma2_e = df + o_ss2e;
  b.o_pt_xyz = od; b.o_pt_px = od + NP * 3; b.o_pt_depth = od + NP * 5;
  double* ods = od + NP * 6;
  b.o_seg_xyz_s = ods; b.o_seg_xyz_e = ods + NS * 3; b.o_seg_depth_s = ods + NS * 6; b.o_seg_depth_e = ods + NS * 7;
  b.o_pt_a = of; b.o_pt_b = of + NP; b.o_pt_mu = of + NP * 2; b.o_pt_sigma2 = of + NP * 3;
  float* ofs = of + NP * 4;
  b.o_seg_a = ofs; b.o_seg_b = ofs + NS; b.o_seg_mu_s = ofs + NS * 2; b.o_seg_mu_e = ofs + NS * 3; b.o_seg_sigma2_s = ofs + NS * 4; b.o_seg_sigma2_e = ofs + NS * 5;
  b.o_pt_status = oi; b.o_seg_status = oi + NP;
  {
    EventPair ep{}; prof_begin(c, PLSVO_K_SEEDS, &ep);
    HIP_TRY(c, launch_update_seeds(b, c->stream));
    prof_end(c, PLSVO_K_SEEDS, &ep);
  }
  std::vector<char> h(od_b + of_b + oi_b);
  HIP_TRY(c, hipMemcpyAsync(h.data(), dout, h.size(), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  const double* hd = reinterpret_cast<const double*>(h.data());
  const float* hf = reinterpret_cast<const float*>(h.data() + od_b);
  const int* hi = reinterpret_cast<const int*>(h.data() + od_b + of_b);
  auto cpd = [&](double* dst, const double* src, size_t k) { if (dst && k) memcpy(dst, src, k * sizeof(double)); };
  auto cpf = [&](float* dst, const float* src, size_t k) { if (dst && k) memcpy(dst, src, k * sizeof(float)); };
  cpd(out->pt_xyz_world, hd, NP * 3); cpd(out->pt_px_cur, hd + NP * 3, NP * 2); cpd(out->pt_depth, hd + NP * 5, NP);
  const double* hds = hd + NP * 6;
  cpd(out->seg_xyz_world_s, hds, NS * 3); cpd(out->seg_xyz_world_e, hds + NS * 3, NS * 3); cpd(out->seg_depth_s, hds + NS * 6, NS); cpd(out->seg_depth_e, hds + NS * 7, NS);
  cpf(out->pt_a, hf, NP); cpf(out->pt_b, hf + NP, NP); cpf(out->pt_mu, hf + NP * 2, NP); cpf(out->pt_sigma2, hf + NP * 3, NP);
  const float* hfs = hf + NP * 4;
  cpf(out->seg_a, hfs, NS); cpf(out->seg_b, hfs + NS, NS); cpf(out->seg_mu_s, hfs + NS * 2, NS); cpf(out->seg_mu_e, hfs + NS * 3, NS);
  cpf(out->seg_sigma2_s, hfs + NS * 4, NS); cpf(out->seg_sigma2_e, hfs + NS * 5, NS);
  if (out->pt_status && np) memcpy(out->pt_status, hi, NP * sizeof(int));
  if (out->seg_status && ns) memcpy(out->seg_status, hi + NP, NS * sizeof(int));
  return PLSVO_OK;
}

// ---- multi-GPU gather ------------------------------------------------------------------------------
// which resident state the records come from, and how many there are -- decided BEFORE anything is sized or launched
namespace {
struct RecordSource { const AlignStateDev* ast = nullptr; const PoseStateDev* pst = nullptr; int n = 0; const char* why_not = nullptr; };
RecordSource pick_record_source(plsvo_ctx* c) {
  RecordSource r;
  if (c->ch_staged && c->a_staged && c->a_n == c->ch_n) {   // the resident frame step: its own pose-optimisation state
    if (c->ch_run_seq == 0) { r.why_not = "pack_pose_records: the staged frame step has not run"; return r; }
    r.ast = c->a_d_state.as<AlignStateDev>(); r.pst = c->ch_d_state.as<PoseStateDev>(); r.n = c->ch_n;
    return r;
  }
  // both batches when they describe the same streams (same size); otherwise the one that ran last
  const bool use_a = c->a_staged && c->a_run_seq > 0, use_p = c->p_staged && c->p_run_seq > 0;
  const bool both = use_a && use_p && c->a_n == c->p_n;
  if (use_a && (both || !use_p || c->a_run_seq > c->p_run_seq)) { r.ast = c->a_d_state.as<AlignStateDev>(); r.n = c->a_n; }
  if (use_p && (both || !use_a || c->p_run_seq > c->a_run_seq)) { r.pst = c->p_d_state.as<PoseStateDev>(); r.n = c->p_n; }
  if (!r.ast && !r.pst) r.why_not = "pack_pose_records: no staged batch has run";
  return r;
}
}  // namespace

extern "C" int plsvo_pack_pose_records(plsvo_ctx* c, plsvo_pose_record* d_dst, int* n_out) {
  CTX_CHECK(c);
  if (!d_dst) return fail(c, PLSVO_E_INVALID, "pack_pose_records: null destination");
  const RecordSource src = pick_record_source(c);
  if (src.why_not) return fail(c, PLSVO_E_STATE, src.why_not);
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, launch_pack_pose_records(src.ast, src.pst, src.n, d_dst, c->stream));
  if (n_out) *n_out = src.n;
  return PLSVO_OK;
}

extern "C" int plsvo_fetch_pose_records(plsvo_ctx* c, int n, plsvo_pose_record* out) {
  CTX_CHECK(c);
  if (!out || n <= 0) return fail(c, PLSVO_E_INVALID, "fetch_pose_records: bad arguments");
  const RecordSource src = pick_record_source(c);
  if (src.why_not) return fail(c, PLSVO_E_STATE, src.why_not);
  // (the pack kernel writes one record per RESIDENT stream: a caller's n that is not that count is rejected before the buffer is sized)
  if (src.n != n) return fail(c, PLSVO_E_INVALID, "fetch_pose_records: n does not match the resident batch");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, c->rec_d.ensure((size_t)n * sizeof(plsvo_pose_record)));
  HIP_TRY(c, launch_pack_pose_records(src.ast, src.pst, src.n, c->rec_d.as<plsvo_pose_record>(), c->stream));
  HIP_TRY(c, hipMemcpyAsync(out, c->rec_d.p, (size_t)n * sizeof(plsvo_pose_record), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return PLSVO_OK;
}

extern "C" int plsvo_gather_poses(plsvo_ctx* c, void* rccl_comm, const plsvo_pose_record* d_local, int n_local, plsvo_pose_record* d_all) {
  CTX_CHECK(c);
  if (!rccl_comm || !d_local || !d_all || n_local <= 0) return fail(c, PLSVO_E_INVALID, "gather_poses: bad arguments");
  HIP_TRY(c, hipSetDevice(c->device));
  static_assert(sizeof(plsvo_pose_record) == 96, "plsvo_pose_record is a 96-byte wire record");
  ncclResult_t r = ncclAllGather(d_local, d_all, (size_t)n_local * (sizeof(plsvo_pose_record) / sizeof(double)), ncclDouble,
                                 reinterpret_cast<ncclComm_t>(rccl_comm), c->stream);
  if (r != ncclSuccess) return fail(c, PLSVO_E_RCCL, std::string("ncclAllGather: ") + ncclGetErrorString(r));
  return PLSVO_OK;
}

// ---- timing ------------------------------------------------------------------------------------------
extern "C" int plsvo_hip_set_profiling(plsvo_ctx* c, int enable) { CTX_CHECK(c); c->profiling = enable != 0; return PLSVO_OK; }
extern "C" int plsvo_hip_kernel_time(plsvo_ctx* c, int k, double* total_ms, int64_t* launches) {
  CTX_CHECK(c);
  if (k < 0 || k >= PLSVO_K_COUNT) return fail(c, PLSVO_E_INVALID, "kernel_time: bad kernel family");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  prof_collect(c);
  if (total_ms) *total_ms = c->ev_ms[k];
  if (launches) *launches = c->ev_launches[k];
  return PLSVO_OK;
}
extern "C" int plsvo_hip_reset_profiling(plsvo_ctx* c) {
  CTX_CHECK(c);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  prof_collect(c);
  for (int k = 0; k < PLSVO_K_COUNT; ++k) { c->ev_ms[k] = 0.0; c->ev_launches[k] = 0; }
  return PLSVO_OK;
}
