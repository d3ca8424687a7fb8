// chain_kernels.hip -- the glue that keeps one frame step resident on the device.
//
// FrameHandlerMono::processFrame (src/frame_handler_mono.cpp:263-345) runs, back to back on one thread,
//     sparse image alignment  ->  Reprojector::reprojectMap (project the map, file candidates in grid cells, refine one per cell with
//     Matcher::findMatchDirect)  ->  pose_optimizer::optimizeGaussNewton
// and every arrow is a few host-side lines: compose the pose (:92), test visibility, pick the cell winners, turn refined pixels into
// bearings / line equations (src/feature.cpp:103-104).  Called one ABI entry point at a time those lines sit on the host between four
// synchronous calls with PCIe copies either side.  The kernels here do them on the device, between the alignment, reprojection,
// matching and pose-optimisation kernels of the same stream (plsvo_capi.hip::plsvo_chain_run):
//   chain_pose_kernel     T_f_w(new) = T_cur_from_ref * T_f_w(prev)  -> frame table [keyframe, new frame] of every stream
//   chain_active_kernel   candidate is worth matching: caller's mask, in frame (8-px border), a segment needs both end points
//   chain_select_kernel   which matches become features of the new frame -- all of them, or the reference's rule: per grid cell the first
//                         candidate (caller's order = its quality order) that matched, cells visited in the caller's (random) order until
//                         more than max_fts have matched (src/reprojector.cpp:188-199, refineBestCandidate :222-243).  Matching every
//                         candidate and selecting afterwards equals the reference's early exits because a match has no side effect on
//                         other matches.  Segments follow their own grid (gridls_: a segment filed under both end-point cells, :405-421,
//                         max_fts_segs, its own cell order) when the caller gives one.  Then the selected features are written as
//                         pose-optimiser input, in selection order.
#include <hip/hip_runtime.h>

#include "../../include/plsvo_hip.h"
#include "plsvo_dev.hpp"
#include "plsvo_math.hpp"

namespace plsvo_hip {

// The record a rank publishes per stream (include/plsvo_hip.h: plsvo_pose_record): pose + what the host decides "tracking
// lost" on (src/frame_handler_mono.cpp:272-274, :327-335), packed from the resident state of the alignment / pose-optimisation launches
// that precede it on the stream.  One lane per stream; 96 B written per record.
__global__ void __launch_bounds__(64) pack_pose_records_kernel(const AlignStateDev* ast, const PoseStateDev* pst, int n, plsvo_pose_record* dst) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= n) return;
  plsvo_pose_record r;
  int status = 0;
  r.n_tracked = 0; r.num_obs_pt = 0; r.num_obs_ls = 0; r.error_final = 0.0;
  for (int k = 0; k < 7; ++k) r.T_f_w[k] = k == 3 ? 1.0 : 0.0;
  if (ast) {
    const AlignStateDev& a = ast[j];
    status |= PLSVO_REC_ALIGN | (a.stop ? PLSVO_REC_ALIGN_STOP : 0) | (a.error ? PLSVO_REC_ALIGN_ERROR : 0);
    r.n_tracked = a.n_meas / PLSVO_PATCH_AREA;                       // run() returns n_meas_ / patch_area_  :94
    for (int k = 0; k < 7; ++k) r.T_f_w[k] = a.T[k];
  }
  if (pst) {
    const PoseStateDev& q = pst[j];
    status |= PLSVO_REC_POSEOPT | ((q.status & 1) ? PLSVO_REC_POSEOPT_EMPTY : 0);
    r.num_obs_pt = q.num_obs_pt; r.num_obs_ls = q.num_obs_ls; r.error_final = q.error_final;
    for (int k = 0; k < 7; ++k) r.T_f_w[k] = q.T[k];
  }
  r.status = status; r.stream = j;
  dst[j] = r;
}
hipError_t launch_pack_pose_records(const AlignStateDev* ast, const PoseStateDev* pst, int n, plsvo_pose_record* dst, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(pack_pose_records_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, ast, pst, n, dst);
  return hipGetLastError();
}

__global__ void __launch_bounds__(64) chain_pose_kernel(const ChainBatchDev b) {
  const int j = blockIdx.x * 64 + threadIdx.x;
  if (j >= b.n_jobs) return;
  const ChainJobDev& J = b.jobs[j];
  const SE3d T_k = se3_mul(se3_load(b.align_poses + 7 * j), se3_load(J.T_prev));   // cur_frame->T_f_w_ = T_cur_from_ref * ref_frame->T_f_w_
  se3_store(se3_load(J.T_kf), b.frame_T + 14 * j);
  se3_store(T_k, b.frame_T + 14 * j + 7);
  b.frame_slot[2 * j] = J.kf_slot;
  b.frame_slot[2 * j + 1] = J.cur_slot;
}

__global__ void __launch_bounds__(256) chain_active_kernel(const ChainBatchDev b) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= b.n_cand) return;
  const ChainJobDev& J = b.jobs[b.cand_job[i]];
  const int k = i - J.cand_off;
  bool a = (b.active_in ? b.active_in[i] != 0 : true) && b.cell[i] >= 0;
  if (k >= J.n_pt) {   // a segment end point: the other end must be in frame (and wanted) too (src/reprojector.cpp:405-421)
    const int other = k < J.n_pt + J.n_seg ? i + J.n_seg : i - J.n_seg;
    a = a && (b.active_in ? b.active_in[other] != 0 : true) && b.cell[other] >= 0;
  }
  b.active[i] = a ? 1 : 0;
}

// in-order compaction by ONE wave: flag[i] for i in [0, n) -> dst[rank] = value(i); returns the count (wave-uniform).  limit < 0: no limit
template <class Flag, class Emit>
__device__ __forceinline__ int wave_compact(int n, int limit, Flag flag, Emit emit) {
  const int lane = threadIdx.x & 63;
  int total = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const bool f = i < n && flag(i);
    const unsigned long long m = __ballot(f);
    const int rank = total + __popcll(m & ((1ull << lane) - 1ull));
    if (f && (limit < 0 || rank < limit)) emit(i, rank);
    total += __popcll(m);
    if (limit >= 0 && total >= limit) { total = limit; break; }
  }
  return total;
}

__global__ void __launch_bounds__(256) chain_select_kernel(const ChainBatchDev b) {
  extern __shared__ int s_winner[];   // n_cells (cell rule only), then seg_n_cells (segment grid only)
  __shared__ int s_n[2];
  const int j = blockIdx.x, tid = threadIdx.x;
  const ChainJobDev J = b.jobs[j];
  const int c0 = J.cand_off;
  int* const sel_pt = b.sel_pt + J.po_pt_off;
  int* const sel_seg = b.sel_seg + J.po_seg_off;
  auto matched = [&](int k) { return b.active[c0 + k] && b.found[c0 + k]; };

  if (b.cell_rule) {
    for (int c = tid; c < b.n_cells; c += 256) s_winner[c] = 0x7fffffff;
    __syncthreads();
    for (int k = tid; k < J.n_pt; k += 256)
      if (matched(k)) atomicMin(&s_winner[b.cell[c0 + k]], k);        // first candidate of the cell, in the caller's order, that matched
    __syncthreads();
  }
  // the segments' own grid (gridls_, src/reprojector.cpp:405-421): a segment is filed under the cell of its projected start point and under
  // the cell of its projected end point; per cell the first segment, in the caller's order, whose findMatchDirect succeeded
  const bool seg_grid = b.cell_rule && b.seg_n_cells > 0;
  int* const s_wseg = s_winner + (b.cell_rule ? b.n_cells : 0);
  if (seg_grid) {
    for (int c = tid; c < b.seg_n_cells; c += 256) s_wseg[c] = 0x7fffffff;
    __syncthreads();
    for (int s = tid; s < J.n_seg; s += 256) {
      const int cs = c0 + J.n_pt + s, ce = cs + J.n_seg;
      if (matched(J.n_pt + s) && matched(J.n_pt + J.n_seg + s)) {
        const int sk = (int)(b.proj_px[2 * cs + 1] / b.seg_cell_size) * b.seg_n_cols + (int)(b.proj_px[2 * cs] / b.seg_cell_size);
        const int ek = (int)(b.proj_px[2 * ce + 1] / b.seg_cell_size) * b.seg_n_cols + (int)(b.proj_px[2 * ce] / b.seg_cell_size);
        atomicMin(&s_wseg[sk], s);
        atomicMin(&s_wseg[ek], s);
      }
    }
    __syncthreads();
  }
  if (tid < 64) {
    int n_pt_sel;
    if (b.cell_rule)    // cells in visit order; the reference stops AFTER the match that makes n_matches_ exceed max_fts
      n_pt_sel = wave_compact(b.n_cells, b.max_fts + 1,
                              [&](int r) { return s_winner[b.cell_order ? b.cell_order[r] : r] != 0x7fffffff; },
                              [&](int r, int rank) { sel_pt[rank] = s_winner[b.cell_order ? b.cell_order[r] : r]; });
    else
      n_pt_sel = wave_compact(J.n_pt, -1, matched, [&](int k, int rank) { sel_pt[rank] = k; });
    int n_seg_sel;
    if (seg_grid)       // cells in visit order, stop AFTER the match that makes n_ls_matches_ exceed max_fts_segs (:200-207); a segment that
                        // wins both of its cells is emitted twice (refine() adds a LineFeat per success, :361-363)
      n_seg_sel = wave_compact(b.seg_n_cells, b.max_fts_segs + 1,
                               [&](int r) { return s_wseg[b.seg_cell_order ? b.seg_cell_order[r] : r] != 0x7fffffff; },
                               [&](int r, int rank) { sel_seg[rank] = s_wseg[b.seg_cell_order ? b.seg_cell_order[r] : r]; });
    else
      n_seg_sel = wave_compact(J.n_seg, -1, [&](int s) { return matched(J.n_pt + s) && matched(J.n_pt + J.n_seg + s); },
                               [&](int s, int rank) { sel_seg[rank] = s; });
    if (tid == 0) { s_n[0] = n_pt_sel; s_n[1] = n_seg_sel; }
  }
  __syncthreads();
  const int n_pt_sel = s_n[0], n_seg_sel = s_n[1];
  // unit bearing of a pixel ([ext] vk::PinholeCamera::cam2world, normalised as Feature's constructor does)
  auto bearing = [&](const double* px, double* f) {
    const double x = (px[0] - b.cx) / b.fx, y = (px[1] - b.cy) / b.fy;
    const double n = sqrt(x * x + y * y + 1.0);
    f[0] = x / n; f[1] = y / n; f[2] = 1.0 / n;
  };
  for (int k = tid; k < n_pt_sel; k += 256) {
    const int c = c0 + sel_pt[k], o = J.po_pt_off + k;
    double f[3];
    bearing(b.m_px + 2 * c, f);
    for (int d = 0; d < 3; ++d) { b.pt_f[3 * o + d] = f[d]; b.pt_pos[3 * o + d] = b.pos[3 * c + d]; }
    b.pt_level[o] = max(b.search_level[c], 0);
  }
  for (int k = tid; k < n_seg_sel; k += 256) {
    const int cs = c0 + J.n_pt + sel_seg[k], ce = cs + J.n_seg, o = J.po_seg_off + k;
    double sf[3], ef[3];
    bearing(b.m_px + 2 * cs, sf); bearing(b.m_px + 2 * ce, ef);
    // LineFeat: line = sf x ef, scaled to a unit normal in the image plane (src/feature.cpp:103-104)
    double l[3] = { sf[1] * ef[2] - sf[2] * ef[1], sf[2] * ef[0] - sf[0] * ef[2], sf[0] * ef[1] - sf[1] * ef[0] };
    const double n = sqrt(l[0] * l[0] + l[1] * l[1]);
    for (int d = 0; d < 3; ++d) { b.seg_line[3 * o + d] = l[d] / n; b.seg_spos[3 * o + d] = b.pos[3 * cs + d]; b.seg_epos[3 * o + d] = b.pos[3 * ce + d]; }
    b.seg_level[o] = max(b.search_level[cs], 0);
  }
  if (tid == 0) {
    PoseJobDev& P = b.po_jobs[j];
    for (int d = 0; d < 7; ++d) P.T0[d] = b.frame_T[14 * j + 7 + d];
    P.fx = fabs(b.fx); P.reproj_thresh = b.reproj_thresh; P.n_iter = b.po_n_iter; P.n_iter_ref = -1;
    P.pt_off = J.po_pt_off; P.n_pts = n_pt_sel; P.seg_off = J.po_seg_off; P.n_seg = n_seg_sel;
    P.ldlt_flavour = b.ldlt_flavour; P.reserved0 = 0;
    b.n_sel[2 * j] = n_pt_sel; b.n_sel[2 * j + 1] = n_seg_sel;
  }
}

hipError_t launch_chain_pose(const ChainBatchDev& b, hipStream_t stream) {
  if (b.n_jobs <= 0) return hipSuccess;
  hipLaunchKernelGGL(chain_pose_kernel, dim3((b.n_jobs + 63) / 64), dim3(64), 0, stream, b);
  return hipGetLastError();
}
hipError_t launch_chain_active(const ChainBatchDev& b, hipStream_t stream) {
  if (b.n_cand <= 0) return hipSuccess;
  hipLaunchKernelGGL(chain_active_kernel, dim3((b.n_cand + 255) / 256), dim3(256), 0, stream, b);
  return hipGetLastError();
}
hipError_t launch_chain_select(const ChainBatchDev& b, hipStream_t stream) {
  if (b.n_jobs <= 0) return hipSuccess;
  const size_t lds = b.cell_rule ? ((size_t)b.n_cells + (size_t)b.seg_n_cells) * sizeof(int) : 0;
  hipLaunchKernelGGL(chain_select_kernel, dim3(b.n_jobs), dim3(256), lds, stream, b);
  return hipGetLastError();
}

}  // namespace plsvo_hip
