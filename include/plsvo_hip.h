/*
 * plsvo_hip.h -- C ABI of the MI355X (gfx950) hot path of PL-SVO:
 *   sparse image alignment (points + sampled line segments) and motion-only pose optimisation,
 *   plus the steps either side of it: device pyramids, map reprojection, direct feature
 *   matching, structure optimisation, depth-filter seed updates, trajectory records.
 *
 * This header IS the drop-in boundary.  The reference has no FFI layer; its boundary is two C++
 * call signatures (reference file:line given per entry point below).  The C++ adapter in
 * pl-svo_amd/host/plsvo/ re-creates those signatures on top of this ABI.
 *
 * Conventions
 *   - every function returns 0 on success and a negative PLSVO_E_* code on failure; nothing throws,
 *     nothing aborts; plsvo_hip_last_error() gives a human-readable string for the last failure.
 *   - plain pointers and sizes only; no C++/torch types.  Pointers are HOST pointers unless the
 *     parameter name starts with d_ (device pointer, HBM of the ctx's device).
 *   - a ctx is bound to one device and one HIP stream; use one ctx per calling thread.
 *   - poses travel as double[7] = { qx, qy, qz, qw, tx, ty, tz } (unit quaternion + translation),
 *     the storage of the reference's Sophus::SE3 (SO3 as unit quaternion, Vector3d translation).
 *   - there is NO CPU fallback behind this ABI: without a gfx950 device plsvo_hip_create() fails.
 */
#ifndef PLSVO_HIP_H_
#define PLSVO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLSVO_MAX_LEVELS 8
#define PLSVO_PATCH_SIZE 4   /* reference: include/plsvo/sparse_img_align.h:48-50 (halfsize 2, size 4, area 16) */
#define PLSVO_PATCH_AREA 16

/* error codes */
#define PLSVO_OK              0
#define PLSVO_E_INVALID      -1  /* bad argument */
#define PLSVO_E_NODEVICE     -2  /* no usable HIP device (there is no CPU fallback) */
#define PLSVO_E_HIP          -3  /* a HIP runtime call failed; see plsvo_hip_last_error */
#define PLSVO_E_CAPACITY     -4  /* a ctx capacity (slots, features, patches) would be exceeded */
#define PLSVO_E_STATE        -5  /* call order violated (e.g. run before stage) */
#define PLSVO_E_RCCL         -6  /* an RCCL call failed */

typedef struct plsvo_ctx plsvo_ctx;

/* undistorted pinhole camera: the only model the reference demo hands the VO
 * (app/run_pipeline.cpp:786-795; vk::PinholeCamera with zero distortion, [ext] vikit) */
typedef struct plsvo_pinhole {
  double fx, fy, cx, cy;
  int32_t width, height;      /* level-0 image size */
} plsvo_pinhole;

/* ------------------------------------------------------------------------------------------ */
/* context                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* device_id: HIP ordinal.  stream: a hipStream_t (as void*) to enqueue on, or NULL to let the
 * ctx create its own non-blocking stream. */
int plsvo_hip_create(int device_id, void* stream, plsvo_ctx** out);
/* The same, but `stream` is ALWAYS the stream to enqueue on -- including the NULL handle, which is HIP's default (legacy)
 * stream and what e.g. torch.cuda.current_stream().cuda_stream returns when no other stream is current.  A caller that
 * orders other work (an RCCL collective, its own kernels) against the library's launches must use this form, or pass a
 * non-NULL stream: plsvo_hip_create(.., NULL, ..) enqueues on a private stream the caller's work is not ordered with. */
int plsvo_hip_create_on_stream(int device_id, void* stream, plsvo_ctx** out);
void plsvo_hip_destroy(plsvo_ctx* ctx);
/* Options of a context.  PLSVO_OPT_LDLT_FLAVOUR selects the zero-pivot rule of the 6x6 `H.ldlt().solve()` the two optimisers call
 * (src/sparse_img_align.cpp:699, src/pose_optimizer.cpp:170), which changed between the Eigen releases PL-SVO's named platforms
 * ship: 320 (default) = Eigen 3.1 ... 3.2.1 (Ubuntu 12.04 / 14.04: stop at eps * max|A_ii|, drop |d| <= eps * max|D|),
 * 330 = Eigen 3.2.2 and later (Ubuntu 16.04's 3.3-beta: exact zeros only).  Identical arithmetic on every full-rank system; they
 * differ with fewer than three point observations.  Takes effect at the next *_stage call. */
#define PLSVO_OPT_LDLT_FLAVOUR 1
/* Launch shapes (threads per frame) of the two hot kernels: 0 = chosen from the batch size (default), or a fixed 64 / 128 / 256 / 512
 * (alignment), 16 / 64 / 256 / 512 (pose optimiser; 16 = a 16-lane row per frame, four frames per wave: the large-batch shape).  For tests and measurements: every shape computes the same thing. */
#define PLSVO_OPT_ALIGN_THREADS 2
#define PLSVO_OPT_POSEOPT_THREADS 3
/* Launch order of a RE-RUN staged batch (plsvo_align_run / plsvo_poseopt_run called again without a new stage call): 1 (default) =
 * the frames start longest-first by the work the PREVIOUS launch of the batch measured (a counting sort on the device behind every
 * launch), 0 = every launch keeps the stage call's order (the alignment's: most patches first).  Scheduling only -- no result depends
 * on it.  The environment switches PLSVO_ALIGN_NO_REORDER / PLSVO_POSEOPT_NO_REORDER set the initial value to 0. */
#define PLSVO_OPT_ALIGN_REORDER 4
#define PLSVO_OPT_POSEOPT_REORDER 5
int plsvo_hip_set_option(plsvo_ctx* ctx, int option, int value);
const char* plsvo_hip_last_error(const plsvo_ctx* ctx);   /* ctx may be NULL: last create error */
void* plsvo_hip_stream(plsvo_ctx* ctx);                   /* the hipStream_t all work is enqueued on */
int plsvo_hip_synchronize(plsvo_ctx* ctx);

/* ------------------------------------------------------------------------------------------ */
/* image pyramids (input of both frames; replaces Frame::img_pyr_, include/plsvo/frame.h:64,   */
/* filled by frame_utils::createImgPyramid, src/frame.cpp:171-180)                             */
/* ------------------------------------------------------------------------------------------ */

/* Allocate n_slots pyramid slots of n_levels u8 images, level l = (width>>l) x (height>>l),
 * rows stored tightly (stride == width of the level).  Re-configuring frees the old slab. */
int plsvo_hip_config_pyramids(plsvo_ctx* ctx, int n_slots, int width, int height, int n_levels);

/* Upload an existing host pyramid (what the reference keeps in Frame::img_pyr_) into a slot.
 * The call returns WITHOUT a stream synchronisation: the levels are packed into a pinned image of the slot and cross PCIe as one
 * copy enqueued on the context's stream -- the caller's buffers are free on return, and whatever is launched on the stream afterwards
 * reads the new pyramid (a caller that reads the slot from ANOTHER stream orders it with plsvo_hip_synchronize).  The slot's tiled
 * mirror (read by the one-wave-per-frame shape of the alignment only) is refreshed LAZILY: the slot is marked stale here and re-tiled
 * by the first launch that reads the mirror; plsvo_hip_build_pyramid / _build_pyramids_dev / _copy_slots refresh it eagerly. */
int plsvo_hip_upload_pyramid(plsvo_ctx* ctx, int slot, int n_levels,
                             const uint8_t* const* level_ptr, const int* width, const int* height,
                             const int* stride_bytes);

/* Upload level 0 only and build levels 1.. on the device with the 2x2 half-sampler
 * (replaces vk::halfSample, [ext] vikit/vision.h, called from src/frame.cpp:178).
 * rounding: 0 = vikit SSE2 path  avg(avg(a,c),avg(b,d)) with (x+y+1)>>1
 *           1 = vikit scalar path (a+b+c+d)/4 truncating. */
int plsvo_hip_build_pyramid(plsvo_ctx* ctx, int slot, const uint8_t* level0, int stride_bytes,
                            int rounding);
/* Same, level 0 already in HBM (tight rows or stride_bytes), for n consecutive slots starting at
 * first_slot; image i is at d_level0 + i*image_pitch_bytes. */
int plsvo_hip_build_pyramids_dev(plsvo_ctx* ctx, int first_slot, int n, const void* d_level0,
                                 int stride_bytes, size_t image_pitch_bytes, int rounding);
/* Device-to-device copy of n whole pyramid slots (src_first.. -> dst_first.., ranges must not overlap), enqueued on the ctx stream:
 * "the current frame becomes the reference frame" (src/frame_handler_mono.cpp:272-274) for a resident batch without crossing PCIe. */
int plsvo_hip_copy_slots(plsvo_ctx* ctx, int dst_first, int src_first, int n);
/* Read one level of a slot back to the host (tight rows); for tests. */
int plsvo_hip_download_level(plsvo_ctx* ctx, int slot, int level, uint8_t* out);

/* ------------------------------------------------------------------------------------------ */
/* sparse image alignment                                                                      */
/* replaces plsvo::SparseImgAlign::run   (include/plsvo/sparse_img_align.h:64-66,              */
/*                                        src/sparse_img_align.cpp:54-95; call sites           */
/*                                        src/frame_handler_mono.cpp:272-274, 418-420)         */
/* ------------------------------------------------------------------------------------------ */

/* One alignment job = one SparseImgAlign::run(ref_frame, cur_frame).
 * Features are those of the REFERENCE frame, flattened (the adapter walks the std::lists once):
 *   points  (ref_frame->pt_fts_ with feat3D != NULL):
 *     pt_px      2*n_pts  Feature::px, level-0 pixels                (include/plsvo/feature.h:42)
 *     pt_xyz_ref 3*n_pts  f * ||feat3D->pos_ - ref_frame->pos()||    (src/sparse_img_align.cpp:229-230)
 *   segments (ref_frame->seg_fts_, ALL of them, so indices stay stable):
 *     seg_spx/seg_epx 2*n_seg  LineFeat::spx / epx                   (include/plsvo/feature.h:85-86)
 *     seg_len    n_seg    LineFeat::length                           (include/plsvo/feature.h:92)
 *     seg_p_ref  3*n_seg  sf * ||feat3D->spos_ - ref_pos||           (src/sparse_img_align.cpp:327-328)
 *     seg_q_ref  3*n_seg  ef * ||feat3D->epos_ - ref_pos||           (src/sparse_img_align.cpp:329-330)
 *     seg_alive_in n_seg  1 if feat3D != NULL, 0 otherwise; NULL = all alive
 */
typedef struct plsvo_align_in {
  int32_t ref_slot, cur_slot;       /* pyramid slots of ref_frame / cur_frame */
  plsvo_pinhole cam;                /* cur_frame->cam_ == ref_frame->cam_ */
  int32_t max_level, min_level;     /* SparseImgAlign ctor (src/sparse_img_align.cpp:40-52) */
  int32_t n_iter;                   /* n_iter_ (30 at the call sites) */
  int32_t reserved0;
  double eps;                       /* eps_ = 1e-6 (src/sparse_img_align.cpp:51) */
  double T_cur_from_ref[7];         /* cur.T_f_w * ref.T_f_w^-1 (src/sparse_img_align.cpp:80) */
  int32_t n_pts, n_seg;
  const double* pt_px;
  const double* pt_xyz_ref;
  const double* seg_spx;
  const double* seg_epx;
  const double* seg_len;
  const double* seg_p_ref;
  const double* seg_q_ref;
  const uint8_t* seg_alive_in;
} plsvo_align_in;

typedef struct plsvo_align_out {
  double T_cur_from_ref[7];         /* after all levels (src/sparse_img_align.cpp:92 multiplies by ref.T_f_w) */
  uint64_t n_meas;                  /* n_meas_ of the last computeResiduals call */
  uint64_t n_tracked;               /* run()'s return value n_meas_/16 (src/sparse_img_align.cpp:94) */
  double H[36];                     /* H_ of the last computeResiduals (row-major; getFisherInformation, :97-102) */
  double chi2;                      /* chi2_ of the solver ([ext] vk::NLLSSolver) */
  uint8_t* seg_alive_out;           /* caller buffer of n_seg bytes or NULL; 0 = LineFeat::feat3D set to NULL
                                       (src/sparse_img_align.cpp:687-688) */
  int32_t iters_per_level[PLSVO_MAX_LEVELS];  /* #computeResiduals calls in the GN loop, index = level */
  int32_t status;                   /* bit 0: solver stop_ flag was raised (NaN in solve, :700) */
  int32_t reserved0;
} plsvo_align_out;

/* per-iteration trace (debug/parity): one record per GN iteration, in execution order */
typedef struct plsvo_align_iterlog {
  int32_t level, iter;
  int32_t accepted;                 /* 1: update applied; 0: rolled back / stopped at this iteration */
  int32_t stop;                     /* solver stop_ flag after this iteration */
  uint64_t n_meas;
  double new_chi2;                  /* value returned by computeResiduals (float chi2 / n_meas) */
  double H[36], Jres[6], x[6];
  double T_after[7];                /* model after this iteration's accept/rollback decision */
} plsvo_align_iterlog;

/* one job, synchronous: stage + run + fetch (the drop-in call) */
int plsvo_sparse_align(plsvo_ctx* ctx, const plsvo_align_in* in, plsvo_align_out* out);
/* n independent jobs in one batch, synchronous (BASELINE config 4: many streams) */
int plsvo_sparse_align_batch(plsvo_ctx* ctx, int n, const plsvo_align_in* in, plsvo_align_out* out);

/* the same, split so that a batch can stay resident in HBM and be re-run (bench, pipelining):
 *   stage: copy job descriptors + features to the device (replaces any previously staged batch)
 *   run  : enqueue the kernels on the ctx stream (asynchronous); re-initialises poses and alive
 *          masks from the staged inputs each time it is called
 *   fetch: wait for the stream and copy the results back */
int plsvo_align_stage(plsvo_ctx* ctx, int n, const plsvo_align_in* in);
int plsvo_align_run(plsvo_ctx* ctx);
int plsvo_align_fetch(plsvo_ctx* ctx, int n, plsvo_align_out* out);

/* per-iteration trace: enable before plsvo_align_run; max_records_per_job bounds the trace */
/* Host-only helper (no device needed): the static patch-slot layout plsvo_align_stage gives one job at one pyramid level.
 * Points own slots [0, n_pts); the segments are packed behind them into the kernel's wave-rounds of 64 slots, first-fit in
 * decreasing N (N = 1 + (N0-1)/2^level, LineFeat::setupSampling src/feature.cpp:160-173, src/sparse_img_align.cpp:320; ties in
 * feature order), starting in the round the points leave partly empty; a segment with N <= 64 samples never straddles a multiple
 * of 64, longer ones follow behind the packed rounds.  seg_code[s] = first slot | N << 20, or -1 for a segment without landmark on
 * entry or with an end point inside the 3-pixel border of the level (src/sparse_img_align.cpp:299-301).  n_slots: slots in use;
 * long_lines: some N > 64 (the level then runs in two passes); n_patches: points + samples (holes not counted).  No result depends
 * on where a segment sits.  Returns PLSVO_E_CAPACITY for N > 2047 or more than 2^20 slots. */
int plsvo_align_slot_layout(const plsvo_align_in* in, int level, int32_t* seg_code, int32_t* n_slots, int32_t* long_lines,
                            long long* n_patches);

int plsvo_align_set_trace(plsvo_ctx* ctx, int max_records_per_job);
int plsvo_align_fetch_trace(plsvo_ctx* ctx, int job, plsvo_align_iterlog* out, int max_records,
                            int* n_records);

/* device pointer to the staged batch's result poses, n*7 doubles (for a device-side gather) */
const double* plsvo_align_poses_dev(plsvo_ctx* ctx);
/* enqueue a device-to-device copy of those n*7 doubles into caller-owned HBM (e.g. a torch tensor) */
int plsvo_align_copy_poses(plsvo_ctx* ctx, double* d_dst);

/* work counters of the last plsvo_align_run (for the roofline accounting):
 *   patch_levels = sum over jobs and levels of patches precomputed (497 B each)
 *   patch_iters  = sum over jobs, levels and GN iterations of patches evaluated (485 B each) */
int plsvo_align_work(plsvo_ctx* ctx, uint64_t* patch_levels, uint64_t* patch_iters);
/* The order in which the NEXT plsvo_align_run of the resident batch starts its jobs (block w works on job order[w]): the stage call's
   (most patches first) until the batch has run; after a run of a batch larger than the device's resident slots, the jobs sorted by the
   patch-iterations that run measured, longest first (align_kernels.hip::align_reorder_kernel).  Scheduling only: a job's results do not
   depend on its place in the launch.  Tests and measurements. */
int plsvo_align_launch_order(plsvo_ctx* ctx, int n, int32_t* order);
/* of patch_iters, the evaluations of POINT patches that also wrote their 64 B of per-pixel chi2 terms to HBM (see plsvo_align_chi2_ties) */
int plsvo_align_work_points(plsvo_ctx* ctx, uint64_t* point_patch_iters);

/* parity accounting of the last plsvo_align_run: Gauss-Newton iterations in total, and how many of them had their
 * `new_chi2 > chi2_` decision ([ext] vk::NLLSSolver::optimizeGaussNewton) taken on the reference's own sequential float sums
 * (src/sparse_img_align.cpp:484, 683, 171, 192) because the two chi2 values were closer than the rounding noise of those sums;
 * near_ties_without_terms: such iterations whose per-pixel terms had not been kept (large batches keep them only once the solver's
 * steps are small) -- decided on the exactly-rounded sums instead.  Any output may be NULL. */
int plsvo_align_chi2_ties(plsvo_ctx* ctx, uint64_t* iterations, uint64_t* ties, uint64_t* near_ties_without_terms);

/* ------------------------------------------------------------------------------------------ */
/* pose optimisation                                                                           */
/* replaces plsvo::pose_optimizer::optimizeGaussNewton (include/plsvo/pose_optimizer.h:47-64,  */
/*          src/pose_optimizer.cpp:38-260 and :262-582; call site frame_handler_mono.cpp:327)  */
/* ------------------------------------------------------------------------------------------ */

/* Features are those of the frame being optimised, flattened, only entries with feat3D != NULL:
 *   pt_f   3*n_pts  Feature::f (bearing)         pt_pos  3*n_pts  feat3D->pos_ (world)
 *   pt_level n_pts  Feature::level
 *   seg_line 3*n_seg LineFeat::line (src/feature.cpp:103-104)
 *   seg_spos/seg_epos 3*n_seg feat3D->spos_/epos_ (world)     seg_level n_seg */
typedef struct plsvo_poseopt_in {
  double T_f_w[7];                  /* frame->T_f_w_ on entry */
  double fx;                        /* frame->cam_->errorMultiplier2() = |fx| ([ext] vikit) */
  double reproj_thresh;             /* 2.0 at the call site (src/config.cpp:102) */
  int32_t n_iter;                   /* 10 at the call site (src/config.cpp:103) */
  int32_t n_iter_ref;               /* <0: 9-argument overload (:38); >=0: 10-argument overload (:262) */
  int32_t n_pts, n_seg;
  const double* pt_f;
  const double* pt_pos;
  const int32_t* pt_level;
  const double* seg_line;
  const double* seg_spos;
  const double* seg_epos;
  const int32_t* seg_level;
} plsvo_poseopt_in;

typedef struct plsvo_poseopt_out {
  double T_f_w[7];                  /* frame->T_f_w_ on exit */
  double cov[36];                   /* frame->Cov_ (src/pose_optimizer.cpp:198-199), row-major */
  double estimated_scale, error_init, error_final;
  uint64_t num_obs_pt, num_obs_ls;
  uint8_t* pt_keep;                 /* caller buffers (n_pts / n_seg bytes) or NULL; 0 = feat3D set to NULL */
  uint8_t* seg_keep;                /*   (src/pose_optimizer.cpp:218, 239) */
  int32_t iters;                    /* GN iterations executed in the first loop */
  int32_t iters_ref;                /* ... in the refinement loop of the 10-argument overload */
  int32_t status;                   /* bit 0: early return, nothing written (errors.empty(), :88-89) */
  int32_t reserved0;
} plsvo_poseopt_out;

typedef struct plsvo_poseopt_iterlog {
  int32_t phase;                    /* 0: first loop, 1: refinement loop */
  int32_t iter;
  int32_t accepted;
  int32_t reserved0;
  double new_chi2;
  double A[36], b[6], dT[6];
  double T_after[7];
} plsvo_poseopt_iterlog;

int plsvo_pose_optimize(plsvo_ctx* ctx, const plsvo_poseopt_in* in, plsvo_poseopt_out* out);
int plsvo_pose_optimize_batch(plsvo_ctx* ctx, int n, const plsvo_poseopt_in* in, plsvo_poseopt_out* out);
int plsvo_poseopt_stage(plsvo_ctx* ctx, int n, const plsvo_poseopt_in* in);
int plsvo_poseopt_run(plsvo_ctx* ctx);
int plsvo_poseopt_fetch(plsvo_ctx* ctx, int n, plsvo_poseopt_out* out);
int plsvo_poseopt_set_trace(plsvo_ctx* ctx, int max_records_per_job);
int plsvo_poseopt_fetch_trace(plsvo_ctx* ctx, int job, plsvo_poseopt_iterlog* out, int max_records,
                              int* n_records);
const double* plsvo_poseopt_poses_dev(plsvo_ctx* ctx);
int plsvo_poseopt_copy_poses(plsvo_ctx* ctx, double* d_dst);
/* feature-iterations of the last run: points (24 B each) and lines (40 B each), SURVEY 8d */
int plsvo_poseopt_work(plsvo_ctx* ctx, uint64_t* pt_iters, uint64_t* seg_iters);

/* ------------------------------------------------------------------------------------------ */
/* structure optimisation (hot-path contract row (f) "next" #3)                                */
/* replaces plsvo::Point::optimize / plsvo::LineSeg::optimize (src/feature3D_impl.cpp:36-95,   */
/* :97-174), called from FrameHandlerBase::optimizeStructure (src/frame_handler_base.cpp:      */
/* 202-237; call site src/frame_handler_mono.cpp:340)                                          */
/* ------------------------------------------------------------------------------------------ */

/* A batch of independent 3-D landmarks, each refined by its own 3x3 Gauss-Newton over its observations
 * (Feature3D::obs_, include/plsvo/feature3D.h): observation = (frame pose T_f_w, unit bearing f).
 *   frame_T        7*n_frames   poses of every frame referenced by an observation
 *   pt_pos         3*n_pts      Point::pos_ on entry
 *   pt_obs_off     n_pts+1      observations of point i are [pt_obs_off[i], pt_obs_off[i+1])
 *   pt_obs_frame   n_pt_obs     index into frame_T            (Feature::frame)
 *   pt_obs_f       3*n_pt_obs   Feature::f
 *   seg_*                       the same for LineSeg::spos_/epos_ with LineFeat::sf / ef
 * The landmark selection (nth_element on last_structure_optim_) stays on the host. */
typedef struct plsvo_structopt_in {
  int32_t n_frames;
  int32_t n_iter_pts;               /* Config::structureOptimNumIter() */
  int32_t n_iter_segs;              /* Config::structureOptimNumIterSegs() */
  int32_t n_pts, n_seg;
  int32_t reserved0;
  const double* frame_T;
  const double* pt_pos;
  const int32_t* pt_obs_off;
  const int32_t* pt_obs_frame;
  const double* pt_obs_f;
  const double* seg_spos;
  const double* seg_epos;
  const int32_t* seg_obs_off;
  const int32_t* seg_obs_frame;
  const double* seg_obs_sf;
  const double* seg_obs_ef;
} plsvo_structopt_in;

typedef struct plsvo_structopt_out {   /* caller buffers; any of them may be NULL */
  double* pt_pos;                   /* 3*n_pts  Point::pos_ on exit */
  double* seg_spos;                 /* 3*n_seg */
  double* seg_epos;                 /* 3*n_seg */
  int32_t* pt_iters;                /* n_pts: residual evaluations executed (1..n_iter) */
  int32_t* seg_iters;               /* n_seg */
} plsvo_structopt_out;

int plsvo_structure_optimize(plsvo_ctx* ctx, const plsvo_structopt_in* in, plsvo_structopt_out* out);

/* ------------------------------------------------------------------------------------------ */
/* direct feature matching (hot-path contract row (f) "next" #2)                               */
/* replaces plsvo::Matcher::findMatchDirect for points (src/matcher.cpp:159-207) and for line  */
/* segments (:232-275), with everything they call: warp::getWarpMatrixAffine (:44-71),         */
/* warp::getBestSearchLevel (:73-86), warp::warpAffine (:88-129),                              */
/* Matcher::createPatchFromPatchWithBorder (:148-157), Matcher::precomputeRefPatch (:209-230), */
/* feature_alignment::align1D (src/feature_alignment.cpp:41-158) and align2D (:160-290).       */
/* Call sites: Reprojector::refineBestCandidate (src/reprojector.cpp:288, :348).               */
/* ------------------------------------------------------------------------------------------ */

#define PLSVO_FTR_CORNER  0         /* PointFeat::CORNER, and both end points of a LineFeat */
#define PLSVO_FTR_EDGELET 1         /* PointFeat::EDGELET: 1-D alignment along the warped gradient */

/* A batch of n independent match candidates.  One candidate = one 2-D position to refine in the image of
 * frame cur_frame[i], starting from px_cur[i] (the landmark's projection, written by Reprojector), against the
 * 8x8 patch around the landmark's closest-view observation (Point::getCloseViewObs, chosen by the host).
 * A line segment contributes two candidates (start and end point, each with its own pos / px / f); the
 * caller ANDs the two `found` flags like matcher.cpp:253-274.  Images are the ctx's pyramid slots.
 *   frame_T      7*n_frames  Frame::T_f_w_ of every frame referenced
 *   frame_slot   n_frames    pyramid slot holding that frame's Frame::img_pyr_
 *   cur_frame    n           index of the frame matched into
 *   ref_frame    n           index of ref_ftr_->frame
 *   ref_px       2*n         ref_ftr_->px (level-0 pixels)       ref_f   3*n   ref_ftr_->f
 *   ref_level    n           ref_ftr_->level                     ref_type n    PLSVO_FTR_*
 *   ref_grad     2*n         PointFeat::grad (read for edgelets only; may be NULL when there are none)
 *   pos          3*n         Point::pos_ (LineSeg::spos_ / epos_)
 *   px_cur       2*n         initial estimate in level-0 pixels of the current image */
typedef struct plsvo_match_in {
  plsvo_pinhole cam;                /* ref_ftr_->frame->cam_ == cur_frame.cam_ (one camera) */
  int32_t n_pyr_levels;             /* Config::nPyrLevels(): search level <= n_pyr_levels-1 (matcher.cpp:177) */
  int32_t align_max_iter;           /* Matcher::Options::align_max_iter (10, include/plsvo/matcher.h:98) */
  int32_t n_frames;
  int32_t n;
  const double* frame_T;
  const int32_t* frame_slot;
  const int32_t* cur_frame;
  const int32_t* ref_frame;
  const double* ref_px;
  const double* ref_f;
  const int32_t* ref_level;
  const uint8_t* ref_type;
  const double* ref_grad;
  const double* pos;
  const double* px_cur;
} plsvo_match_in;

typedef struct plsvo_match_out {    /* caller buffers; any of them may be NULL */
  double* px_cur;                   /* 2*n  refined position (level-0 pixels); the input value when the
                                       reference observation is too close to the border (matcher.cpp:168-170) */
  uint8_t* found;                   /* n    findMatchDirect's return value */
  int32_t* search_level;            /* n    Matcher::search_level_ (-1 when rejected before the warp) */
  int32_t* n_iter;                  /* n    residual passes executed by align1D/align2D */
} plsvo_match_out;

int plsvo_match_direct(plsvo_ctx* ctx, const plsvo_match_in* in, plsvo_match_out* out);

/* Reprojector::reproject(frame, Point*) / (frame, LineSeg*) (src/reprojector.cpp:389-423): the projection of map
 * landmarks into a frame that produces the candidates (and their initial px_cur) for plsvo_match_direct.
 *   px[i]   = frame->w2c(pos[i]) = world2cam(T_f_w * pos[i])                      (include/plsvo/frame.h:113)
 *   cell[i] = (int)(px.y / cell_size) * grid_n_cols + (int)(px.x / cell_size)    when isInFrame(px.cast<int>(), 8),
 *             -1 otherwise (the reference then rejects the landmark; a segment needs both end points in frame and
 *             is filed under both cells, :405-421 -- the caller ANDs, like for plsvo_match_direct)
 * The grid itself (cells, random visit order, one match per cell) is host control flow and stays with the caller. */
typedef struct plsvo_reproject_in {
  plsvo_pinhole cam;
  int32_t n_frames;                 /* poses referenced */
  int32_t n;                        /* landmark positions (points, segment start points, segment end points) */
  int32_t cell_size;                /* Config::gridSize() (points) or gridSizeSegs() (segments) */
  int32_t grid_n_cols;              /* ceil(width / cell_size)  (reprojector.cpp:59, :71) */
  int32_t boundary;                 /* 8: "the patch size in the matcher" (:393) */
  int32_t reserved0;
  const double* frame_T;            /* 7*n_frames  Frame::T_f_w_ */
  const int32_t* frame;             /* n: index of the frame each position is projected into */
  const double* pos;                /* 3*n world positions */
} plsvo_reproject_in;

typedef struct plsvo_reproject_out { /* caller buffers; either may be NULL */
  double* px;                       /* 2*n */
  int32_t* cell;                    /* n */
} plsvo_reproject_out;

int plsvo_reproject(plsvo_ctx* ctx, const plsvo_reproject_in* in, plsvo_reproject_out* out);

/* ------------------------------------------------------------------------------------------ */
/* resident frame step (hot-path contract row (f) "next": the callers either side of the path) */
/* FrameHandlerMono::processFrame runs alignment -> Reprojector::reprojectMap -> pose          */
/* optimisation back to back (src/frame_handler_mono.cpp:263-345).  plsvo_chain_* runs the     */
/* same sequence for a batch of streams in ONE call: the alignment result is composed into the */
/* new frame's pose (:92), the map candidates are projected with it, matched                   */
/* (Matcher::findMatchDirect), selected, turned into bearings / line equations                 */
/* (src/feature.cpp:103-104) and handed to the pose optimiser without leaving the device.      */
/* ------------------------------------------------------------------------------------------ */

/* One stream.  Candidates are the landmarks of ONE keyframe (pose T_kf_w, pyramid slot kf_slot) with their observation in it, in
 * the caller's order of preference (the reference sorts a cell's candidates by landmark quality, reprojector.cpp:225):
 * n_cand_pt points, then the start points of n_cand_seg segments, then their end points -- every array below has
 * n_cand_pt + 2 * n_cand_seg entries in that order (ref_type / ref_grad are read for points only). */
typedef struct plsvo_chain_in {
  plsvo_align_in align;             /* previous frame -> new frame; ref_slot / cur_slot are the two frames' pyramid slots */
  double T_prev_w[7];               /* previous frame's T_f_w_ */
  double T_kf_w[7];                 /* keyframe's T_f_w_ */
  int32_t kf_slot;
  int32_t n_cand_pt, n_cand_seg;
  int32_t reserved0;
  const double* pos;                /* 3 per candidate: Point::pos_ / LineSeg::spos_ / LineSeg::epos_ */
  const double* ref_px;             /* 2: the keyframe observation (as plsvo_match_in) */
  const double* ref_f;              /* 3 */
  const int32_t* ref_level;
  const uint8_t* ref_type;          /* PLSVO_FTR_*; may be NULL (all corners) */
  const double* ref_grad;           /* 2; may be NULL without edgelets */
  const uint8_t* active;            /* may be NULL: 0 = leave this candidate out (e.g. a landmark not yet in the map) */
} plsvo_chain_in;

typedef struct plsvo_chain_params {
  plsvo_pinhole cam;                /* one camera for the batch */
  int32_t n_pyr_levels;             /* Config::nPyrLevels() (matcher) */
  int32_t align_max_iter;           /* Matcher::Options::align_max_iter (10) */
  int32_t cell_size;                /* Config::gridSize(): the reprojection grid of the points (reprojector.cpp:57-66) */
  int32_t cell_rule;                /* 0: every matched candidate becomes a feature; 1: the reference's rule -- per cell the first
                                       candidate that matches, cells in cell_order, stop after the match that makes the count exceed
                                       max_fts (reprojector.cpp:188-199, :222-243).  Segments follow their own grid when seg_cell_size > 0
                                       (below); with seg_cell_size == 0 every segment whose two end points match becomes a feature */
  int32_t max_fts;                  /* Config::maxFts() */
  int32_t poseopt_n_iter;           /* 10 (src/config.cpp:103) */
  const int32_t* cell_order;        /* grid_n_cols * grid_n_rows cell indices (Grid::cell_order, shuffled once, :63-66); NULL = 0,1,2,.. */
  double reproj_thresh;             /* 2.0 (src/config.cpp:102) */
  /* the segments' grid, gridls_ (reprojector.cpp:68-79, :200-207, :256-275, :405-421), used when cell_rule != 0 and seg_cell_size > 0:
   * a segment is filed under the cell of its projected start point AND under the cell of its projected end point; cells are visited
   * in seg_cell_order, per cell the first segment (caller's order = quality order) whose findMatchDirect succeeded becomes a feature,
   * and the visit stops after the match that makes the count exceed max_fts_segs.  A segment that wins both of its cells becomes a
   * feature TWICE, as in the reference (refine() adds a LineFeat per success): sel_seg / seg_keep hold up to 2 * n_cand_seg entries. */
  int32_t seg_cell_size;            /* Config::gridSizeSegs(); 0 = no segment grid */
  int32_t max_fts_segs;             /* Config::maxFtsSegs() (100, src/config.cpp:76) */
  const int32_t* seg_cell_order;    /* ceil(width / seg_cell_size) * ceil(height / seg_cell_size) cell indices; NULL = 0,1,2,.. */
} plsvo_chain_params;

typedef struct plsvo_chain_out {
  plsvo_align_out align;            /* as plsvo_align_fetch */
  plsvo_poseopt_out pose;           /* as plsvo_poseopt_fetch; pt_keep / seg_keep index the SELECTED features (sel_pt / sel_seg order) */
  int32_t n_sel_pt, n_sel_seg;      /* features the new frame received */
  /* caller buffers, any may be NULL */
  uint8_t* found;                   /* per candidate: findMatchDirect's result (0 for candidates left out) */
  double* px;                       /* 2 per candidate: refined pixel (the projection for candidates left out / not found) */
  int32_t* search_level;            /* per candidate */
  int32_t* sel_pt;                  /* n_cand_pt: candidate index of selected point feature k, k < n_sel_pt */
  int32_t* sel_seg;                 /* n_cand_seg (2 * n_cand_seg with a segment grid): segment index of selected segment feature k, k < n_sel_seg */
} plsvo_chain_out;

int plsvo_chain_stage(plsvo_ctx* ctx, int n, const plsvo_chain_in* in, const plsvo_chain_params* params);
int plsvo_chain_run(plsvo_ctx* ctx);     /* enqueue only: alignment, pose composition, reprojection, matching, selection, pose optimisation */
int plsvo_chain_fetch(plsvo_ctx* ctx, int n, plsvo_chain_out* out);
int plsvo_frame_step_batch(plsvo_ctx* ctx, int n, const plsvo_chain_in* in, const plsvo_chain_params* params, plsvo_chain_out* out);
const double* plsvo_chain_poses_dev(plsvo_ctx* ctx);   /* n*7 doubles: the optimised T_f_w of the staged streams, on the device */

/* TUM-style trajectory record of a frame (app/run_pipeline.cpp:425-451): the camera pose in the world,
 * T_f_w^-1, as tx ty tz qx qy qz qw.  Returns 1 and fills out7 when the reference would write the line, 0 when
 * it skips the frame (a covariance entry outside (1e-16, 1e16), or an exactly-identity pose).  Host-only helper:
 * no device work, may be called with ctx == NULL. */
int plsvo_trajectory_record(const double T_f_w[7], const double cov[36], double out7[7]);

/* ------------------------------------------------------------------------------------------ */
/* depth-filter seed update (hot-path contract row (f) "next" #4, last item)                   */
/* replaces the per-seed bodies of DepthFilter::updatePointSeeds / updateLineSeeds             */
/* (src/depth_filter.cpp:270-365, :367-471) with everything they call:                         */
/* Matcher::findEpipolarMatchDirect (src/matcher.cpp:277-420) and                              */
/* findEpipolarMatchDirectSegmentEndpoint (:422-586), depthFromTriangulation (:133-146),       */
/* [ext] vk::patch_score::ZMSSD<4>, DepthFilter::computeTau (src/depth_filter.cpp:568-584),    */
/* updatePointSeed (:489-512), updateLineSeed (:514-566).                                      */
/* The seed lists, the batch-age test (:289-292), the converged-seed callbacks and the         */
/* detector's grid occupancy stay on the host; this call maps seed state -> seed state.        */
/* ------------------------------------------------------------------------------------------ */

#define PLSVO_SEED_NOT_VISIBLE 0    /* behind the camera or outside the image: unchanged (:296-304) */
#define PLSVO_SEED_NO_MATCH    1    /* epipolar search failed: b += 1 (:311-318) */
#define PLSVO_SEED_UPDATED     2    /* Bayesian update applied, seed stays */
#define PLSVO_SEED_CONVERGED   3    /* updated and sqrt(sigma2) < z_range/thresh: the host creates the landmark from xyz_world and removes the seed (:334-355) */
#define PLSVO_SEED_NAN         4    /* updated but z_inv_min was NaN: the host removes the seed (:356-360) */

/* float fields are the reference's float members of PointSeed / LineSeed (include/plsvo/depth_filter.h:60-96).
 * Point seed i: ref feature (frame pt_ref_frame[i], px, f, level, type, grad) + Beta/normal parameters.
 * Line seed i: Feature::px / f (what the reference hands to the end-point search for BOTH end points, :411-414),
 * LineFeat::sf / ef (used for visibility, tau and the landmark), level, and the two-ended parameters. */
typedef struct plsvo_seeds_in {
  plsvo_pinhole cam;
  int32_t n_pyr_levels;             /* Config::nPyrLevels() */
  int32_t align_max_iter;           /* Matcher::Options::align_max_iter (10) */
  int32_t max_epi_search_steps;     /* Matcher::Options::max_epi_search_steps (1000) */
  int32_t edgelet_filtering;        /* Matcher::Options::epi_search_edgelet_filtering (1) */
  double edgelet_max_angle;         /* Matcher::Options::epi_search_edgelet_max_angle (0.7) */
  double px_noise;                  /* 1.0 (:279-280) */
  double convergence_sigma2_thresh; /* DepthFilter::Options::seed_convergence_sigma2_thresh (200.0) */
  int32_t n_frames;
  int32_t n_pt;
  int32_t n_seg;
  int32_t reserved0;
  const double* frame_T;            /* 7*n_frames */
  const int32_t* frame_slot;        /* n_frames */
  /* point seeds */
  const int32_t* pt_ref_frame;      /* n_pt: it->ftr->frame */
  const int32_t* pt_cur_frame;      /* n_pt: the frame the seed is updated with */
  const double* pt_px;              /* 2*n_pt */
  const double* pt_f;               /* 3*n_pt */
  const int32_t* pt_level;          /* n_pt */
  const uint8_t* pt_type;           /* n_pt PLSVO_FTR_* */
  const double* pt_grad;            /* 2*n_pt (edgelets; may be NULL without edgelets) */
  const float* pt_a; const float* pt_b; const float* pt_mu; const float* pt_z_range; const float* pt_sigma2;
  /* line seeds */
  const int32_t* seg_ref_frame;
  const int32_t* seg_cur_frame;
  const double* seg_px;             /* 2*n_seg Feature::px */
  const double* seg_f;              /* 3*n_seg Feature::f  */
  const double* seg_sf;             /* 3*n_seg */
  const double* seg_ef;             /* 3*n_seg */
  const int32_t* seg_level;
  const float* seg_a; const float* seg_b; const float* seg_mu_s; const float* seg_mu_e;
  const float* seg_z_range_s; const float* seg_z_range_e; const float* seg_sigma2_s; const float* seg_sigma2_e;
} plsvo_seeds_in;

typedef struct plsvo_seeds_out {     /* caller buffers; any of them may be NULL */
  int32_t* pt_status;               /* n_pt PLSVO_SEED_* */
  float* pt_a; float* pt_b; float* pt_mu; float* pt_sigma2;
  double* pt_xyz_world;             /* 3*n_pt, valid for PLSVO_SEED_CONVERGED */
  double* pt_px_cur;                /* 2*n_pt Matcher::px_cur_ after a successful match (for setGridOccpuancy, :327-331) */
  double* pt_depth;                 /* n_pt   the triangulated depth z of a successful match */
  int32_t* seg_status;
  float* seg_a; float* seg_b; float* seg_mu_s; float* seg_mu_e; float* seg_sigma2_s; float* seg_sigma2_e;
  double* seg_xyz_world_s; double* seg_xyz_world_e;   /* 3*n_seg each */
  double* seg_depth_s; double* seg_depth_e;           /* n_seg each */
} plsvo_seeds_out;

int plsvo_update_seeds(plsvo_ctx* ctx, const plsvo_seeds_in* in, plsvo_seeds_out* out);

/* ------------------------------------------------------------------------------------------ */
/* multi-GPU: gather of per-stream pose records (new; the reference is single-process)         */
/* ------------------------------------------------------------------------------------------ */

/* The per-stream record a rank publishes after a frame step: what FrameHandlerMono::processFrame decides on
 * afterwards -- the pose it keeps (src/frame_handler_mono.cpp:92, :327-329), SparseImgAlign::run's return value (:272-274), the pose
 * optimiser's surviving observations (sfba_n_edges_final = num_obs_pt + num_obs_ls, :327-335) -- so that the rank that holds the
 * gathered table sees which streams lost tracking without a second exchange.  Fixed size: 96 bytes. */
#define PLSVO_REC_ALIGN      0x01   /* an alignment batch contributed (n_tracked, PLSVO_REC_ALIGN_STOP, PLSVO_REC_ALIGN_ERROR valid) */
#define PLSVO_REC_ALIGN_STOP 0x02   /* the alignment's solver raised stop_ (NaN in solve, src/sparse_img_align.cpp:700) */
#define PLSVO_REC_ALIGN_ERROR 0x04  /* device-side capacity / consistency error (plsvo_align_out.status bit 1) */
#define PLSVO_REC_POSEOPT    0x08   /* a pose-optimisation batch contributed (T_f_w is its result; num_obs_*, error_final valid) */
#define PLSVO_REC_POSEOPT_EMPTY 0x10 /* optimizeGaussNewton returned early: no observation (src/pose_optimizer.cpp:88-89) */
typedef struct plsvo_pose_record {
  double T_f_w[7];                  /* qx qy qz qw tx ty tz: pose_optimizer's T_f_w when PLSVO_REC_POSEOPT, else the alignment's T_cur_from_ref */
  uint64_t n_tracked;               /* SparseImgAlign::run's return value, n_meas_ / patch_area_ (src/sparse_img_align.cpp:94) */
  uint64_t num_obs_pt;              /* optimizeGaussNewton's num_obs_pt (src/pose_optimizer.cpp:218-226) */
  uint64_t num_obs_ls;              /* ... num_obs_ls (:239-245) */
  double error_final;               /* ... error_final (:247-251) */
  int32_t status;                   /* PLSVO_REC_* */
  int32_t stream;                   /* index of the stream in the publishing context's batch */
} plsvo_pose_record;

/* Writes one record per stream of the resident batch into device memory (d_dst: n records), enqueued on the ctx stream after the
 * launches that produce them: the resident frame step (plsvo_chain_stage) if one is staged, else the staged alignment and
 * pose-optimisation batches that have run -- both when they have the same number of jobs (one job of each per stream), otherwise the
 * one that ran last.  *n_out (may be NULL) receives the record count. */
int plsvo_pack_pose_records(plsvo_ctx* ctx, plsvo_pose_record* d_dst, int* n_out);
/* The same records in host memory (out: n records, n = the resident batch's size): packs into a ctx-owned device buffer, copies,
 * synchronises the ctx stream.  For a single-process host that wants the table without a communicator. */
int plsvo_fetch_pose_records(plsvo_ctx* ctx, int n, plsvo_pose_record* out);

/* All-gather of n_local pose records (device memory) over an RCCL communicator (ncclComm_t passed as void*), enqueued on the ctx
 * stream: d_all receives world_size*n_local records, rank-major (96 B per stream: 768 B per rank for BASELINE configs[3]).  No other
 * collective exists on this path (streams are independent). */
int plsvo_gather_poses(plsvo_ctx* ctx, void* rccl_comm, const plsvo_pose_record* d_local, int n_local,
                       plsvo_pose_record* d_all);

/* ------------------------------------------------------------------------------------------ */
/* timing (hipEvent pairs recorded on the ctx stream around each kernel family)                */
/* ------------------------------------------------------------------------------------------ */
#define PLSVO_K_ALIGN_INIT    0
#define PLSVO_K_ALIGN_LEVEL   1   /* the residual/Jacobian + GN kernel (dominant) */
#define PLSVO_K_POSEOPT       2
#define PLSVO_K_HALFSAMPLE    3
#define PLSVO_K_STRUCTOPT     4
#define PLSVO_K_MATCH         5
#define PLSVO_K_SEEDS         6
#define PLSVO_K_COUNT         7
int plsvo_hip_set_profiling(plsvo_ctx* ctx, int enable);
/* accumulated GPU time and launch count of kernel family k since the last reset (synchronises) */
int plsvo_hip_kernel_time(plsvo_ctx* ctx, int k, double* total_ms, int64_t* launches);
int plsvo_hip_reset_profiling(plsvo_ctx* ctx);

/* library / device info */
const char* plsvo_hip_version(void);
/* compile-time experiment switches of THIS build of the alignment kernel, space-separated ("" for the default build; "byte_cache",
 * "lds_img": pl-svo_amd/csrc/Makefile's A/B targets).  bench.py prices its roofline with the bytes the build really moves. */
const char* plsvo_hip_build_flags(void);
int plsvo_hip_device_info(plsvo_ctx* ctx, char* name, int name_len, int* cu_count, size_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PLSVO_HIP_H_ */
