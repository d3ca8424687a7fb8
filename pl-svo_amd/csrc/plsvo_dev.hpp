// plsvo_dev.hpp -- device-side data layout (HBM) shared by the kernels and the C-ABI host code.
//
// Layout principles (MI355X-first):
//   * every batch is a set of flat SoA arrays; a job is an (offset, count) pair into them -- no pointers
//     chased on the device, fully coalesced feature reads;
//   * pyramids live in one slab: slot s, level l at base + s*slot_bytes + off[l], rows tight
//     (stride == level width), each level 256-byte aligned, so staging a level into LDS is one
//     linear 16-byte-per-lane copy;
//   * per-level alignment cache (reference-patch intensity + image gradient, 3 floats per pixel, and the
//     per-patch 3-D point) is written once per level and re-read every Gauss-Newton iteration by the
//     same lanes in the same order (coalesced float4).
#pragma once
#include <stdint.h>

#include "../../include/plsvo_hip.h"

namespace plsvo_hip {

// byte offset of level `level` inside a pyramid slot: every level is followed by >= 64 bytes of slack and
// starts on a 256-byte boundary (one formula for host and device)
#if defined(__HIPCC__)
__host__ __device__
#endif
inline unsigned int pyr_level_offset(int width, int height, int level) {
  unsigned long long off = 0;
  for (int l = 0; l < level; ++l) off += (((unsigned long long)(width >> l) * (unsigned long long)(height >> l) + 64) + 255) & ~255ull;
  return (unsigned int)off;
}

// TILED MIRROR of the pyramids, read by the alignment kernel (written by tile_level_kernel whenever a slot changes).  A level is
// cut into tiles of 16 x 8 pixels = 128 B = one L2 line, tiles row-major, pixel (x, y) at
//     ((y >> 3) * tiles_x + (x >> 4)) * 128 + (y & 7) * 16 + (x & 15),        tiles_x = ceil(w / 16)
// A 5x5 window of a row-major level touches five lines (one per image row: 640 B fetched for 25 B used); in tiles it touches
// (1 + 4/16)(1 + 4/8) = 1.9 lines on average.  The launch runs within a few per cent of the achievable HBM rate, so lines are time.
#if defined(__HIPCC__)
__host__ __device__
#endif
inline unsigned int pyr_tiled_level_offset(int width, int height, int level) {
  unsigned long long off = 0;
  for (int l = 0; l < level; ++l) {
    const unsigned long long tiles = (unsigned long long)(((width >> l) + 15) >> 4) * (unsigned long long)(((height >> l) + 7) >> 3);
    off += (tiles * 128 + 255) & ~255ull;
  }
  return (unsigned int)off;
}
#if defined(__HIPCC__)
__device__ __forceinline__ int tiled_row_offset(int tiles_x, int y) { return (((y >> 3) * tiles_x) << 7) + ((y & 7) << 4); }
__device__ __forceinline__ int tiled_col_offset(int x) { return ((x >> 4) << 7) + (x & 15); }
#endif

// alignment launch shapes from this many threads per frame on are LATENCY shapes (a frame owns most of a CU): four lanes per slot in the
// pixel pass, the reference patches kept as FLOAT rows (192 B per slot, L2-resident) instead of the 64-byte record the throughput shapes rebuild
constexpr int kQuadMinThreads = 256;

struct PyrDesc {
  const uint8_t* base;
  unsigned long long slot_bytes;
  const uint8_t* tbase;               // tiled mirror: slot s at tbase + s * tslot_bytes, level l at + pyr_tiled_level_offset(w, h, l)
  unsigned long long tslot_bytes;
  int n_slots, n_levels;
  int w[PLSVO_MAX_LEVELS], h[PLSVO_MAX_LEVELS];
  unsigned int off[PLSVO_MAX_LEVELS];
};

struct AlignJobDev {
  int ref_slot, cur_slot;
  double fx, fy, cx, cy;
  int width, height;
  int max_level, min_level, n_iter, skip;   // skip: no features at all (src/sparse_img_align.cpp:58-62)
  double eps;
  int pt_off, n_pts, seg_off, n_seg;        // into the batch feature arrays
  int patch_off, patch_cap;                 // into the batch patch-cache arrays (slots)
  int n_slots[PLSVO_MAX_LEVELS];            // patch slots of the static layout, per level (host: align_slot_layout)
  int long_mask;                            // bit l: some segment has more than 64 samples at level l (two-pass level)
  int ldlt_flavour;                         // 320 / 330: zero-pivot rule of Eigen's LDLT (plsvo_wave.hpp::wave_solve6_core)
};

struct AlignStateDev {
  double T[7];                 // current model T_cur_from_ref
  double chi2;                 // solver chi2_
  unsigned long long n_meas;   // n_meas_ of the last computeResiduals
  double H[36];                // H_ of the last computeResiduals
  int stop;                    // solver stop_
  int log_count;
  int iters[PLSVO_MAX_LEVELS];
  unsigned long long patch_levels, patch_iters;  // work counters
  unsigned long long patch_iters_pt;             // of patch_iters, those of point features that wrote 64 B of chi2 terms to HBM
  int error;                   // device-side capacity/consistency error
  int chi2_ties;               // Gauss-Newton iterations whose accept / roll-back decision was taken on the exact float chi2 sums
  int chi2_unarmed;            // near ties met while the per-pixel terms of one of the two iterations had not been kept
  int reserved2;
  unsigned long long phase_ticks[8];  // only filled by -DPLSVO_TIMING builds (s_memtime ticks per phase)
};

struct AlignBatchDev {
  const AlignJobDev* jobs;
  AlignStateDev* state;
  const double* T0;            // 7 per job
  const double* pt_px;         // 2 per point
  const double* pt_xyz;        // 3 per point
  const double* seg_spx;       // 2 per segment
  const double* seg_epx;
  const double* seg_len;       // 1
  const double* seg_p;         // 3
  const double* seg_q;         // 3
  const uint8_t* seg_alive_in; // 1 (may be null)
  uint8_t* seg_alive;          // 1, working copy / result
  // static slot layout of the segments: seg_slot[(level - slot_level0) * slot_stride + segment] = first slot | N << 20, or -1
  const int* seg_slot;
  int slot_level0, slot_stride;
  // per-level patch cache (capacity = sum over jobs of patch_cap slots)
  double* patch_xyz;           // 3 per slot: 3-D point in the ref frame
  float* patch_uvref;          // 2 per slot: ref pixel position at the level (float, as Patch::setPosition)
  float* cache_ref;            // 64 bytes per slot: the reference patch's byte record (7 rows of 8 image bytes + the two sub-pixel fractions, align_refpatch.hpp);
                               // latency shapes (>= kQuadMinThreads threads per frame): 192 bytes per slot, four rows of {ref[4], dx[4], dy[4]} floats
  // per-pixel terms of the solver's chi2 for the POINT features, double-buffered by iteration parity (plane 0 / 1, chi_plane
  // floats apart): 16 per point of the batch, res*res*w (src/sparse_img_align.cpp:484).  Written by every iteration, read only when
  // two successive chi2 values are too close for the double-precision sums to order them the way the reference's sequential
  // float sums do (align_kernels.hip::exact_chi2_pair).
  float* chi_terms;
  unsigned long long chi_plane;
  int chi_lds_pts;             // > 0: the two planes live in LDS instead (capacity in points per plane): small batches, where a
  int reserved_lds;            //      workgroup has LDS to spare and nothing to hide a global store's acknowledge behind
  double* poses;               // 7 per job: final model, contiguous (what a device-side consumer / the RCCL gather reads)
  PyrDesc pyr;
  plsvo_align_iterlog* log;    // log_cap per job, or null
  int log_cap;
  int n_jobs;
  const int* order;            // launch order: workgroup w works on job order[w] (jobs with the most patches first), or null
  // TWO WORKGROUPS PER FRAME (latency shapes, at most cu_count / 2 frames): rank 0 owns the point slots, rank 1 the segment slots (the host
  // layout starts the segments at a multiple of 64); every iteration the two exchange their 32 partial sums through `xbuf` (64 tagged
  // 8-byte granules per rank and parity, plsvo_wave.hpp::pair_allgather32) and both run the solver on the identical totals.
  int* work_key;               // 1 per job, or null: the patch-iterations the job's last launch evaluated (what the next launch's order is sorted by)
  int pair;                    // 0 = one workgroup per frame, 1 = two
  unsigned int xseq0;          // first sequence number of this launch's exchanges (launch number << 10: never repeats in the buffer's life)
  unsigned long long* xbuf;    // 2 (parity) x 2 (rank) x 64 granules per frame
};

struct PoseJobDev {
  double T0[7];
  double fx, reproj_thresh;
  int n_iter, n_iter_ref;
  int pt_off, n_pts, seg_off, n_seg;
  int ldlt_flavour, reserved0;
};

struct PoseStateDev {
  double T[7];
  double cov[36];
  double estimated_scale, error_init, error_final;
  unsigned long long num_obs_pt, num_obs_ls;
  int iters, iters_ref, status, log_count;
  unsigned long long pt_iters, seg_iters;   // work counters
  unsigned long long phase_ticks[8];        // only filled by -DPLSVO_TIMING builds (s_memtime ticks per phase)
};

struct PoseBatchDev {
  const PoseJobDev* jobs;
  PoseStateDev* state;
  const double* pt_f;          // 3 per point
  const double* pt_pos;        // 3
  const int* pt_level;
  const double* seg_line;      // 3 per segment
  const double* seg_spos;
  const double* seg_epos;
  const int* seg_level;
  uint8_t* pt_keep;
  uint8_t* seg_keep;
  float* scratch_f32;          // per feature, for the float medians (errors)
  double* scratch_f64;         // 5 per feature: chi2_vec_init (2x), chi2_vec_final, and two per point for its normalised observation
  plsvo_poseopt_iterlog* log;
  int log_cap;
  int n_jobs;
  const int* order;            // launch slot -> job, or null (identity): a RE-RUN of a staged batch takes its frames sorted by work_key, most first
  int* work_key;               // 1 per job, or null: the feature-iterations the job's last launch evaluated
};

struct StructBatchDev {
  const double* frame_T;
  const double* pt_pos; const int* pt_obs_off; const int* pt_obs_frame; const double* pt_obs_f;
  const double* seg_spos; const double* seg_epos; const int* seg_obs_off; const int* seg_obs_frame;
  const double* seg_obs_sf; const double* seg_obs_ef;
  double* pt_pos_out; double* seg_spos_out; double* seg_epos_out; int* pt_iters; int* seg_iters;
  int n_pts, n_seg, n_iter_pts, n_iter_segs;
};

// direct feature matching: flat candidate arrays (include/plsvo_hip.h plsvo_match_in), one lane per candidate
struct MatchBatchDev {
  const uint8_t* pyr_base; unsigned long long slot_bytes;
  int width, height;                // level-0 size of every pyramid slot
  double fx, fy, cx, cy; int cam_width, cam_height;
  int n, n_pyr_levels, align_max_iter;
  const double* frame_T; const int* frame_slot; const int* cur_frame; const int* ref_frame;
  const double* ref_px; const double* ref_f; const int* ref_level; const uint8_t* ref_type; const double* ref_grad;
  const double* pos; const double* px_cur;
  double* px_out; uint8_t* found; int* search_level; int* n_iter;
  const uint8_t* active;            // optional (resident frame step): candidates with 0 are reported "not found" without any work
};

// depth-filter seeds (include/plsvo_hip.h plsvo_seeds_in / plsvo_seeds_out), one lane per seed
struct SeedsBatchDev {
  const uint8_t* pyr_base; unsigned long long slot_bytes;
  int width, height;
  double fx, fy, cx, cy; int cam_width, cam_height;
  int n_pyr_levels, align_max_iter, max_epi_search_steps, edgelet_filtering;
  double edgelet_max_angle, px_error_angle, convergence_sigma2_thresh;
  int n_pt, n_seg;
  const double* frame_T; const int* frame_slot;
  const int* pt_ref_frame; const int* pt_cur_frame; const double* pt_px; const double* pt_f; const int* pt_level; const uint8_t* pt_type;
  const double* pt_grad; const float* pt_a; const float* pt_b; const float* pt_mu; const float* pt_z_range; const float* pt_sigma2;
  const int* seg_ref_frame; const int* seg_cur_frame; const double* seg_px; const double* seg_f; const double* seg_sf; const double* seg_ef;
  const int* seg_level; const float* seg_a; const float* seg_b; const float* seg_mu_s; const float* seg_mu_e; const float* seg_z_range_s;
  const float* seg_z_range_e; const float* seg_sigma2_s; const float* seg_sigma2_e;
  // outputs
  int* o_pt_status; float* o_pt_a; float* o_pt_b; float* o_pt_mu; float* o_pt_sigma2; double* o_pt_xyz; double* o_pt_px; double* o_pt_depth;
  int* o_seg_status; float* o_seg_a; float* o_seg_b; float* o_seg_mu_s; float* o_seg_mu_e; float* o_seg_sigma2_s; float* o_seg_sigma2_e;
  double* o_seg_xyz_s; double* o_seg_xyz_e; double* o_seg_depth_s; double* o_seg_depth_e;
};

// resident frame step (chain_kernels.hip): one job = one stream's new frame
struct ChainJobDev {
  double T_prev[7], T_kf[7];        // previous frame's and keyframe's T_f_w
  int kf_slot, cur_slot;            // pyramid slots
  int cand_off, n_pt, n_seg;        // candidates of the job: [points | segment start points | segment end points] from cand_off
  int po_pt_off, po_seg_off;        // where its selected features go in the pose optimiser's input arrays
  int reserved0;
};
struct ChainBatchDev {
  const ChainJobDev* jobs; int n_jobs, n_cand;
  const double* align_poses;        // 7 per job: T_cur_from_ref of the alignment launch
  double* frame_T; int* frame_slot; // 2 per job: [keyframe, new frame]
  const int* cand_job; const double* pos; const uint8_t* active_in;
  const int* cell; uint8_t* active;                                  // reprojection result / worth matching
  const double* m_px; const uint8_t* found; const int* search_level; // matcher result
  double fx, fy, cx, cy;
  int n_cells, cell_rule, max_fts; const int* cell_order;
  // the segments' grid (gridls_): 0 cells = every matched segment becomes a feature
  const double* proj_px;            // 2 per candidate: the projection the reprojection kernel wrote (what files a segment in its cells)
  int seg_cell_size, seg_n_cols, seg_n_cells, max_fts_segs; const int* seg_cell_order;
  PoseJobDev* po_jobs; double* pt_f; double* pt_pos; int* pt_level; double* seg_line; double* seg_spos; double* seg_epos; int* seg_level;
  int* sel_pt; int* sel_seg; int* n_sel;
  double reproj_thresh; int po_n_iter, ldlt_flavour;
};

struct ReprojBatchDev {
  double fx, fy, cx, cy; int cam_width, cam_height;
  int n, cell_size, grid_n_cols, boundary;
  const double* frame_T; const int* frame; const double* pos;
  double* px; int* cell;
};

}  // namespace plsvo_hip
