// hip_adapter.hpp -- C++ host side of the drop-in: re-creates the reference's two call signatures
//
//     plsvo::SparseImgAlign(max_level, min_level, n_iter, method, display, verbose).run(ref_frame, cur_frame)
//                                   (include/plsvo/sparse_img_align.h:46-66, src/sparse_img_align.cpp:40-95)
//     plsvo::pose_optimizer::optimizeGaussNewton(reproj_thresh, n_iter, verbose, frame, estimated_scale,
//                                   error_init, error_final, num_obs_pt, num_obs_ls)
//                                   (include/plsvo/pose_optimizer.h:47-64, src/pose_optimizer.cpp:38-260, :262-582)
//
// on top of the C ABI (include/plsvo_hip.h).  Header-only and duck-typed on the frame/feature types: it reads
// exactly the members the reference's hot path reads (include/plsvo/frame.h:58-71, feature.h:41-92,
// feature3D.h:103,149-150) and reproduces every mutation the reference performs:
//     cur_frame->T_f_w_ (sparse_img_align.cpp:92), LineFeat::feat3D = NULL for culled segments (:687-688),
//     frame->T_f_w_, frame->Cov_ and feat3D = NULL for culled observations (pose_optimizer.cpp:183-199,218,239),
//     the scalar outputs.
// It compiles against the reference's own headers (Eigen/Sophus/OpenCV types expose the accessors used here)
// and against the dependency-free look-alike types in mini_types.hpp used by this repo's tests.
// The few places where the reference's third-party types need help (pinhole intrinsics sit behind
// vk::AbstractCamera*) are customisation points in plsvo_hip_adapter::traits -- see INTEGRATION.md.
//
// There is NO CPU fallback: if the C ABI reports an error the adapter prints it and behaves like the
// reference's own failure paths (run() returns 0; optimizeGaussNewton leaves its outputs untouched).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <list>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../../include/plsvo_hip.h"
#include "../../csrc/plsvo_math.hpp"

namespace plsvo_hip_adapter {

// ---- customisation points --------------------------------------------------------------------
// true when the camera exposes radial/tangential coefficients d0()..d3() (vk::PinholeCamera does) and one of them is not
// zero: the device projects with the UNDISTORTED pinhole model only (what app/run_pipeline.cpp:786-795 hands the VO), so a
// distorted camera must be refused, not silently projected wrongly
template <class Cam>
inline auto camera_is_distorted(const Cam& c, int) -> decltype(c.d0(), c.d1(), c.d2(), c.d3(), bool()) {
  return c.d0() != 0.0 || c.d1() != 0.0 || c.d2() != 0.0 || c.d3() != 0.0;
}
template <class Cam>
inline bool camera_is_distorted(const Cam&, long) { return false; }   // no coefficients to look at: the caller vouches for it

template <class Cam>
struct camera_traits {  // default: the camera type has fx()/fy()/cx()/cy()/width()/height() (vk::PinholeCamera does)
  static plsvo_pinhole get(const Cam& c) {
    plsvo_pinhole p;
    p.fx = c.fx(); p.fy = c.fy(); p.cx = c.cx(); p.cy = c.cy(); p.width = c.width(); p.height = c.height();
    return p;
  }
  /// false -> the adapter refuses the camera (prints why, behaves like the reference's failure path)
  static bool supported(const Cam& c) {
    if (camera_is_distorted(c, 0)) {
      std::fprintf(stderr, "[plsvo_hip] camera has non-zero distortion coefficients: only the undistorted pinhole model is implemented "
                           "(hand the VO the undistorted camera, as app/run_pipeline.cpp does)\n");
      return false;
    }
    return true;
  }
};
// camera_traits<Cam>::supported(cam) if the (possibly user-specialised) traits define it, true otherwise: a specialisation that
// follows INTEGRATION.md's get()-only recipe must keep compiling
template <class Cam, class = void>
struct camera_supported_ {
  static bool check(const Cam&) { return true; }
};
template <class Cam>
struct camera_supported_<Cam, decltype((void)camera_traits<Cam>::supported(std::declval<const Cam&>()))> {
  static bool check(const Cam& c) { return camera_traits<Cam>::supported(c); }
};
template <class Cam>
inline bool camera_supported(const Cam& c) { return camera_supported_<Cam>::check(c); }

template <class Img>
struct image_traits {  // default: cv::Mat-like (data, cols, rows, step)
  static const uint8_t* data(const Img& m) { return reinterpret_cast<const uint8_t*>(m.data); }
  static int cols(const Img& m) { return m.cols; }
  static int rows(const Img& m) { return m.rows; }
  static int stride(const Img& m) { return static_cast<int>(static_cast<size_t>(m.step)); }
};
template <class SE3T>
struct se3_traits {  // default: non-templated Sophus::SE3 (unit_quaternion(), translation(), ctor from both)
  static void get(const SE3T& T, double out[7]) {
    const auto& q = T.unit_quaternion();
    const auto& t = T.translation();
    out[0] = q.x(); out[1] = q.y(); out[2] = q.z(); out[3] = q.w(); out[4] = t[0]; out[5] = t[1]; out[6] = t[2];
  }
  static void set(SE3T& T, const double in[7]) {
    typedef typename std::decay<decltype(T.unit_quaternion())>::type Q;   // Eigen::Quaterniond: ctor (w, x, y, z)
    typedef typename std::decay<decltype(T.translation())>::type V;       // Eigen::Vector3d: ctor (x, y, z)
    T = SE3T(Q(in[3], in[0], in[1], in[2]), V(in[4], in[5], in[6]));
  }
};
template <class Mat66>
struct mat66_traits {  // default: Eigen-like operator()(row, col)
  static void set(Mat66& M, const double* row_major) {
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) M(i, j) = row_major[i * 6 + j];
  }
};

// ---- one context per calling thread -------------------------------------------------------------
struct Context {
  plsvo_ctx* ctx = nullptr;
  int width = 0, height = 0, levels = 0;
  // pyramid slots: 0/1 belong to SparseImgAlign::run (ref/cur); slots 2.. cache the pyramids of the frames the
  // direct matcher reads (keyframes are immutable once created, so (address, id_) identifies the pixels)
  struct CachedFrame { const void* frame; long id; unsigned long stamp; };
  std::vector<CachedFrame> cache;
  unsigned long clock = 0;
  unsigned long generation = 0;   // bumped whenever the pyramid slab is re-allocated: every slot index handed out before is void
  ~Context() { if (ctx) plsvo_hip_destroy(ctx); }
  static int cache_slots() { const char* e = std::getenv("PLSVO_KF_SLOTS"); const int n = e ? std::atoi(e) : 16; return n > 1 ? n : 2; }
  /// slot holding `frame`'s pyramid, or -1 on a miss; on a miss *victim gets the least recently used slot not stamped
  /// in the current batch (or -1 when every slot is in use by this batch)
  int lookup(const void* frame, long id, unsigned long batch_start, int* victim) {
    int lru = -1;
    for (size_t k = 0; k < cache.size(); ++k) {
      if (cache[k].frame == frame && cache[k].id == id) { cache[k].stamp = ++clock; return 2 + (int)k; }
      if (cache[k].stamp < batch_start && (lru < 0 || cache[k].stamp < cache[(size_t)lru].stamp)) lru = (int)k;
    }
    *victim = lru;
    return -1;
  }
  bool ensure(int w, int h, int n_levels) {
    if (!ctx) {
      const char* dev = std::getenv("PLSVO_DEVICE");
      if (plsvo_hip_create(dev ? std::atoi(dev) : 0, nullptr, &ctx) != PLSVO_OK) {
        std::fprintf(stderr, "[plsvo_hip] %s\n", plsvo_hip_last_error(nullptr));
        ctx = nullptr;
        return false;
      }
    }
    if (w != width || h != height || n_levels > levels) {   // n_levels is a minimum: fewer levels never reconfigure
      const int n_cache = cache_slots();
      if (plsvo_hip_config_pyramids(ctx, 2 + n_cache, w, h, n_levels) != PLSVO_OK) {
        std::fprintf(stderr, "[plsvo_hip] %s\n", plsvo_hip_last_error(ctx));
        return false;
      }
      width = w; height = h; levels = n_levels;
      cache.assign((size_t)n_cache, CachedFrame{nullptr, 0, 0ul});   // reconfiguring drops the slab
      ++generation;
    }
    return true;
  }
  /// pyramid slot of `fr` (uploaded on first sight, cached by (address, id_)); -1 on failure.  `batch_start` protects the
  /// slots already handed out to the same batch from eviction (0 = start a new batch at the current clock).
  template <class FrameT>
  int slot_for(const FrameT* fr, int n_levels, unsigned long* batch_start);
};
inline Context& default_context() { static thread_local Context c; return c; }

template <class V>
inline void copy3(const V& v, double* o) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; }

template <class Frame>
inline bool upload_frame_pyramid(Context& c, int slot, const Frame& f, int n_levels) {
  typedef typename std::remove_reference<decltype(f.img_pyr_[0])>::type Img;
  std::vector<const uint8_t*> ptr((size_t)n_levels);
  std::vector<int> w((size_t)n_levels), h((size_t)n_levels), s((size_t)n_levels);
  for (int l = 0; l < n_levels; ++l) {
    const Img& m = f.img_pyr_[(size_t)l];
    ptr[(size_t)l] = image_traits<Img>::data(m); w[(size_t)l] = image_traits<Img>::cols(m);
    h[(size_t)l] = image_traits<Img>::rows(m); s[(size_t)l] = image_traits<Img>::stride(m);
  }
  if (plsvo_hip_upload_pyramid(c.ctx, slot, n_levels, ptr.data(), w.data(), h.data(), s.data()) != PLSVO_OK) {
    std::fprintf(stderr, "[plsvo_hip] %s\n", plsvo_hip_last_error(c.ctx));
    return false;
  }
  return true;
}

template <class FrameT>
int Context::slot_for(const FrameT* fr, int n_levels, unsigned long* batch_start) {
  if (*batch_start == 0) *batch_start = clock + 1;
  int victim = -1;
  int slot = lookup((const void*)fr, (long)fr->id_, *batch_start, &victim);
  if (slot >= 0) return slot;
  if (victim < 0) {
    std::fprintf(stderr, "[plsvo_hip] more than %d distinct frames in one batch (raise PLSVO_KF_SLOTS)\n", (int)cache.size());
    return -1;
  }
  slot = 2 + victim;
  if (!upload_frame_pyramid(*this, slot, *fr, n_levels)) { cache[(size_t)victim] = CachedFrame{nullptr, 0, 0ul}; return -1; }
  cache[(size_t)victim] = CachedFrame{(const void*)fr, (long)fr->id_, ++clock};
  return slot;
}

}  // namespace plsvo_hip_adapter

namespace plsvo {

/// Optimize the pose of the frame by minimizing the photometric error of feature patches
/// (same public surface as include/plsvo/sparse_img_align.h:46-70; the vk::NLLSSolver base is gone because the
/// Gauss-Newton loop runs on the device).
template <class FramePtrT>
class SparseImgAlignT {
 public:
  enum Method { GaussNewton, LevenbergMarquardt };  // vk::NLLSSolver::Method; the call sites pass GaussNewton

  SparseImgAlignT(int max_level, int min_level, int n_iter, Method method, bool display, bool verbose)
      : max_level_(max_level), min_level_(min_level), n_iter_(n_iter), method_(method), display_(display), verbose_(verbose),
        eps_(0.000001) {
    for (int k = 0; k < 36; ++k) H_[k] = 0.0;
  }

  /// returns the number of tracked features, n_meas_/patch_area_ (src/sparse_img_align.cpp:94)
  size_t run(FramePtrT ref_frame, FramePtrT cur_frame) {
    using namespace plsvo_hip_adapter;
    using namespace plsvo_hip;
    if (ref_frame->pt_fts_.empty() && ref_frame->seg_fts_.empty()) {  // :58-62
      std::fprintf(stderr, "\033[0;33m[WARN] SparseImgAlign: no features (points or segments) to track!\033[0;0m\n");
      return 0;
    }
    if (method_ != GaussNewton) {
      std::fprintf(stderr, "[plsvo_hip] SparseImgAlign: only GaussNewton is implemented (the reference's call sites use it)\n");
      return 0;
    }
    typedef typename std::remove_reference<decltype(*ref_frame->cam_)>::type Cam;
    plsvo_align_in in;
    if (!camera_supported<Cam>(*ref_frame->cam_)) return 0;
    in.cam = camera_traits<Cam>::get(*ref_frame->cam_);
    const int n_levels = max_level_ + 1;
    Context& c = default_context();
    if ((int)ref_frame->img_pyr_.size() < n_levels || (int)cur_frame->img_pyr_.size() < n_levels) return 0;
    // the slab holds every level a frame has, so that the matcher / depth filter can share the cached pyramids
    const int slab_levels = (int)ref_frame->img_pyr_.size() > n_levels ? (int)ref_frame->img_pyr_.size() : n_levels;
    if (!c.ensure(in.cam.width, in.cam.height, slab_levels)) return 0;
    // pyramids: cached on the device by (frame address, id_).  The previous call's cur_frame is this call's ref_frame
    // (src/frame_handler_mono.cpp:272-274), so in steady state only ONE pyramid crosses PCIe per frame.
    unsigned long batch_start = 0;
    const int ref_levels = (int)ref_frame->img_pyr_.size() < c.levels ? (int)ref_frame->img_pyr_.size() : c.levels;
    const int cur_levels = (int)cur_frame->img_pyr_.size() < c.levels ? (int)cur_frame->img_pyr_.size() : c.levels;
    const int ref_slot = c.slot_for(&*ref_frame, ref_levels, &batch_start);
    const int cur_slot = ref_slot < 0 ? -1 : c.slot_for(&*cur_frame, cur_levels, &batch_start);
    if (ref_slot < 0 || cur_slot < 0) return 0;

    // poses: T_cur_from_ref = cur.T_f_w * ref.T_f_w^-1 (:80); ref_pos = ref.T_f_w^-1 translation (frame.h:131)
    double Tr[7], Tc[7];
    typedef typename std::remove_reference<decltype(ref_frame->T_f_w_)>::type SE3T;
    se3_traits<SE3T>::get(ref_frame->T_f_w_, Tr);
    se3_traits<SE3T>::get(cur_frame->T_f_w_, Tc);
    const SE3d T_ref = se3_load(Tr), T_cur = se3_load(Tc);
    const SE3d T_ref_inv = se3_inv(T_ref);
    const SE3d T_cfr = se3_mul(T_cur, T_ref_inv);
    const double* ref_pos = T_ref_inv.t;

    // flatten the reference frame's features (the only walk over the std::lists)
    std::vector<double> pt_px, pt_xyz, spx, epx, len, pref, qref;
    std::vector<uint8_t> alive;
    for (auto it = ref_frame->pt_fts_.begin(); it != ref_frame->pt_fts_.end(); ++it) {
      if ((*it)->feat3D == NULL) continue;
      double pos[3], f[3];
      copy3((*it)->feat3D->pos_, pos); copy3((*it)->f, f);
      const double d0 = pos[0] - ref_pos[0], d1 = pos[1] - ref_pos[1], d2 = pos[2] - ref_pos[2];
      const double depth = std::sqrt(d0 * d0 + d1 * d1 + d2 * d2);          // :229
      pt_px.push_back((*it)->px[0]); pt_px.push_back((*it)->px[1]);
      pt_xyz.push_back(f[0] * depth); pt_xyz.push_back(f[1] * depth); pt_xyz.push_back(f[2] * depth);  // :230
    }
    std::vector<decltype(&**ref_frame->seg_fts_.begin())> seg_ptr;
    for (auto it = ref_frame->seg_fts_.begin(); it != ref_frame->seg_fts_.end(); ++it) {
      auto* s = &**it;
      seg_ptr.push_back(s);
      spx.push_back(s->spx[0]); spx.push_back(s->spx[1]); epx.push_back(s->epx[0]); epx.push_back(s->epx[1]);
      len.push_back(s->length);
      double p[3] = { 0, 0, 1 }, q[3] = { 0, 0, 1 };
      if (s->feat3D != NULL) {
        double sp[3], ep[3], sf[3], ef[3];
        copy3(s->feat3D->spos_, sp); copy3(s->feat3D->epos_, ep); copy3(s->sf, sf); copy3(s->ef, ef);
        const double pd = std::sqrt((sp[0] - ref_pos[0]) * (sp[0] - ref_pos[0]) + (sp[1] - ref_pos[1]) * (sp[1] - ref_pos[1]) + (sp[2] - ref_pos[2]) * (sp[2] - ref_pos[2]));
        const double qd = std::sqrt((ep[0] - ref_pos[0]) * (ep[0] - ref_pos[0]) + (ep[1] - ref_pos[1]) * (ep[1] - ref_pos[1]) + (ep[2] - ref_pos[2]) * (ep[2] - ref_pos[2]));
        for (int k = 0; k < 3; ++k) { p[k] = sf[k] * pd; q[k] = ef[k] * qd; }   // :327-330
      }
      for (int k = 0; k < 3; ++k) { pref.push_back(p[k]); qref.push_back(q[k]); }
      alive.push_back(s->feat3D != NULL ? 1 : 0);
    }
    in.ref_slot = ref_slot; in.cur_slot = cur_slot;
    in.max_level = max_level_; in.min_level = min_level_; in.n_iter = n_iter_; in.reserved0 = 0; in.eps = eps_;
    se3_store(T_cfr, in.T_cur_from_ref);
    in.n_pts = (int)(pt_px.size() / 2); in.n_seg = (int)len.size();
    in.pt_px = pt_px.data(); in.pt_xyz_ref = pt_xyz.data();
    in.seg_spx = spx.data(); in.seg_epx = epx.data(); in.seg_len = len.data();
    in.seg_p_ref = pref.data(); in.seg_q_ref = qref.data(); in.seg_alive_in = alive.data();
    if (in.n_pts == 0 && in.n_seg == 0) return 0;

    plsvo_align_out out;
    std::vector<uint8_t> alive_out(alive.size() ? alive.size() : 1, 1);
    out.seg_alive_out = alive_out.data();
    const int n_trace = verbose_ ? (max_level_ - min_level_ + 1) * (n_iter_ + 1) : 0;   // verbose: per-iteration records come back
    if (verbose_) plsvo_align_set_trace(c.ctx, n_trace);
    const int rc = plsvo_sparse_align(c.ctx, &in, &out);
    if (rc == PLSVO_OK && verbose_) print_trace(c.ctx, n_trace);
    if (verbose_) plsvo_align_set_trace(c.ctx, 0);
    if (rc != PLSVO_OK) {
      std::fprintf(stderr, "[plsvo_hip] sparse_align failed: %s\n", plsvo_hip_last_error(c.ctx));
      return 0;
    }
    // write back exactly what the reference mutates
    const SE3d T_new = se3_mul(se3_load(out.T_cur_from_ref), T_ref);             // :92
    double Tn[7];
    se3_store(T_new, Tn);
    se3_traits<SE3T>::set(cur_frame->T_f_w_, Tn);
    for (size_t s = 0; s < seg_ptr.size(); ++s)
      if (alive[s] && !alive_out[s]) seg_ptr[s]->feat3D = NULL;                    // :687-688
    for (int k = 0; k < 36; ++k) H_[k] = out.H[k];
    n_meas_ = out.n_meas; chi2_ = out.chi2; stop_ = (out.status & 1) != 0;
    for (int l = 0; l < PLSVO_MAX_LEVELS; ++l) iters_per_level_[l] = out.iters_per_level[l];
    return (size_t)out.n_tracked;                                                  // :94
  }

  /// Fisher information H_/sigma_i^2 (src/sparse_img_align.cpp:97-102), returned BY VALUE like the reference's
  /// `Matrix<double,6,6> getFisherInformation()` (include/plsvo/sparse_img_align.h:69); the matrix type is the frame's own
  /// 6x6 type (Frame::Cov_, include/plsvo/frame.h:67)
  typedef typename std::decay<decltype(std::declval<FramePtrT>()->Cov_)>::type Matrix66;
  Matrix66 getFisherInformation() const {
    Matrix66 I;
    getFisherInformation(I);
    return I;
  }
  /// the same into a caller-provided matrix of any type mat66_traits knows
  template <class Mat66>
  void getFisherInformation(Mat66& I) const {
    const double sigma_i_sq = 5e-4 * 255 * 255;
    double tmp[36];
    for (int k = 0; k < 36; ++k) tmp[k] = H_[k] / sigma_i_sq;
    plsvo_hip_adapter::mat66_traits<Mat66>::set(I, tmp);
  }

  /// verbose_: what the reference prints while it optimises -- the level banner of run() (src/sparse_img_align.cpp:88-89) and
  /// the per-iteration lines of [ext] vk::NLLSSolver::optimizeGaussNewton ("It. k  Success/Failure  new_chi2 = ..  n_meas = ..
  /// x_norm = ..", restated from vikit's nlls_solver_impl.hpp) -- reconstructed after the launch from the device's trace
  static void print_trace(plsvo_ctx* ctx, int cap) {
    std::vector<plsvo_align_iterlog> log((size_t)(cap > 0 ? cap : 1));
    int n = 0;
    if (plsvo_align_fetch_trace(ctx, 0, log.data(), cap, &n) != PLSVO_OK) return;
    int level = -1;
    for (int k = 0; k < n; ++k) {
      const plsvo_align_iterlog& r = log[(size_t)k];
      if (r.level != level) { level = r.level; std::printf("\nPYRAMID LEVEL %i\n---------------\n", level); }
      double xn = 0;
      for (int j = 0; j < 6; ++j) xn = std::fabs(r.x[j]) > xn ? std::fabs(r.x[j]) : xn;
      if (r.accepted)
        std::printf("It. %d\t Success\t new_chi2 = %g\t n_meas = %llu\t x_norm = %g\n", r.iter, r.new_chi2, (unsigned long long)r.n_meas, xn);
      else
        std::printf("It. %d\t Failure\t new_chi2 = %g\t Error increased. Stop optimizing.\n", r.iter, r.new_chi2);
    }
  }

  // solver state the reference exposes through vk::NLLSSolver
  size_t n_meas_ = 0;
  double chi2_ = 1e10;
  bool stop_ = false;
  int iters_per_level_[PLSVO_MAX_LEVELS] = { 0 };

 private:
  int max_level_, min_level_, n_iter_;
  Method method_;
  bool display_, verbose_;   // display_ (the reference's cv::Mat residual image for a debug window, :121-122, :493-497) is accepted and ignored
  double eps_;
  double H_[36];
};

namespace pose_optimizer {

/// Motion-only bundle adjustment (include/plsvo/pose_optimizer.h:47-64).  n_iter_ref < 0 selects the
/// 9-argument overload's behaviour, >= 0 the 10-argument one.
template <class FramePtrT>
void optimizeGaussNewtonImpl(const double reproj_thresh, const size_t n_iter, const long n_iter_ref, const bool verbose,
                             FramePtrT& frame, double& estimated_scale, double& error_init, double& error_final,
                             size_t& num_obs_pt, size_t& num_obs_ls) {
  using namespace plsvo_hip_adapter;
  Context& c = default_context();
  if (!c.ctx && !c.ensure(64, 64, 1)) return;
  typedef typename std::remove_reference<decltype(frame->T_f_w_)>::type SE3T;
  plsvo_poseopt_in in;
  se3_traits<SE3T>::get(frame->T_f_w_, in.T_f_w);
  in.fx = frame->cam_->errorMultiplier2();
  in.reproj_thresh = reproj_thresh; in.n_iter = (int)n_iter; in.n_iter_ref = (int)n_iter_ref;
  std::vector<double> f, pos, line, spos, epos;
  std::vector<int32_t> plev, slev;
  std::vector<decltype(&**frame->pt_fts_.begin())> pt_ptr;
  std::vector<decltype(&**frame->seg_fts_.begin())> seg_ptr;
  for (auto it = frame->pt_fts_.begin(); it != frame->pt_fts_.end(); ++it) {
    if ((*it)->feat3D == NULL) continue;
    pt_ptr.push_back(&**it);
    for (int k = 0; k < 3; ++k) { f.push_back((*it)->f[k]); pos.push_back((*it)->feat3D->pos_[k]); }
    plev.push_back((*it)->level);
  }
  for (auto it = frame->seg_fts_.begin(); it != frame->seg_fts_.end(); ++it) {
    if ((*it)->feat3D == NULL) continue;
    seg_ptr.push_back(&**it);
    for (int k = 0; k < 3; ++k) { line.push_back((*it)->line[k]); spos.push_back((*it)->feat3D->spos_[k]); epos.push_back((*it)->feat3D->epos_[k]); }
    slev.push_back((*it)->level);
  }
  in.n_pts = (int)plev.size(); in.n_seg = (int)slev.size();
  in.pt_f = f.data(); in.pt_pos = pos.data(); in.pt_level = plev.data();
  in.seg_line = line.data(); in.seg_spos = spos.data(); in.seg_epos = epos.data(); in.seg_level = slev.data();
  plsvo_poseopt_out out;
  std::vector<uint8_t> pk(pt_ptr.size() ? pt_ptr.size() : 1, 1), sk(seg_ptr.size() ? seg_ptr.size() : 1, 1);
  out.pt_keep = pk.data(); out.seg_keep = sk.data();
  const int n_trace = verbose ? (int)n_iter + (n_iter_ref > 0 ? (int)n_iter_ref : 0) + 2 : 0;
  if (verbose) plsvo_poseopt_set_trace(c.ctx, n_trace);
  const int rc = plsvo_pose_optimize(c.ctx, &in, &out);
  if (rc == PLSVO_OK && verbose) {   // the per-iteration lines of src/pose_optimizer.cpp:175-190, from the device's trace
    std::vector<plsvo_poseopt_iterlog> log((size_t)n_trace);
    int n = 0;
    if (plsvo_poseopt_fetch_trace(c.ctx, 0, log.data(), n_trace, &n) == PLSVO_OK)
      for (int k = 0; k < n; ++k) {
        double dn = 0;
        for (int j = 0; j < 6; ++j) dn = std::fabs(log[(size_t)k].dT[j]) > dn ? std::fabs(log[(size_t)k].dT[j]) : dn;
        if (log[(size_t)k].accepted) std::printf("it %d\t Success \t new_chi2 = %g\t norm(dT) = %g\n", log[(size_t)k].iter, log[(size_t)k].new_chi2, dn);
        else std::printf("it %d\t FAILURE \t new_chi2 = %g\n", log[(size_t)k].iter, log[(size_t)k].new_chi2);
      }
  }
  if (verbose) plsvo_poseopt_set_trace(c.ctx, 0);
  if (rc != PLSVO_OK) {
    std::fprintf(stderr, "[plsvo_hip] pose_optimize failed: %s\n", plsvo_hip_last_error(c.ctx));
    return;
  }
  num_obs_pt = (size_t)out.num_obs_pt;                                             // :71 (set before the early return)
  if (out.status & 1) return;                                                     // errors.empty() :88-89
  se3_traits<SE3T>::set(frame->T_f_w_, out.T_f_w);
  mat66_traits<typename std::remove_reference<decltype(frame->Cov_)>::type>::set(frame->Cov_, out.cov);  // :199
  for (size_t i = 0; i < pt_ptr.size(); ++i) if (!pk[i]) pt_ptr[i]->feat3D = NULL;                        // :218
  for (size_t i = 0; i < seg_ptr.size(); ++i) if (!sk[i]) seg_ptr[i]->feat3D = NULL;                      // :239
  estimated_scale = out.estimated_scale; error_init = out.error_init; error_final = out.error_final;
  num_obs_ls = (size_t)out.num_obs_ls;
  if (verbose)
    std::printf("n deleted obs = %zu points \t %zu lines\t scale = %g\t error init = %g\t error end = %g\n",
                pt_ptr.size() - (size_t)out.num_obs_pt, seg_ptr.size() - (size_t)out.num_obs_ls, estimated_scale, error_init, error_final);
}

template <class FramePtrT>
void optimizeGaussNewton(const double reproj_thresh, const size_t n_iter, const bool verbose, FramePtrT& frame,
                         double& estimated_scale, double& error_init, double& error_final, size_t& num_obs_pt, size_t& num_obs_ls) {
  optimizeGaussNewtonImpl(reproj_thresh, n_iter, -1L, verbose, frame, estimated_scale, error_init, error_final, num_obs_pt, num_obs_ls);
}
template <class FramePtrT>
void optimizeGaussNewton(const double reproj_thresh, const size_t n_iter, const size_t n_iter_ref, const bool verbose, FramePtrT& frame,
                         double& estimated_scale, double& error_init, double& error_final, size_t& num_obs_pt, size_t& num_obs_ls) {
  optimizeGaussNewtonImpl(reproj_thresh, n_iter, (long)n_iter_ref, verbose, frame, estimated_scale, error_init, error_final, num_obs_pt, num_obs_ls);
}

}  // namespace pose_optimizer
}  // namespace plsvo

namespace plsvo_hip_adapter {
// distinct frames referenced by the observations of a landmark batch, in first-seen order (C++11: no generic lambdas)
struct FrameTable {
  std::vector<const void*> frames;
  std::vector<double>& poses;
  explicit FrameTable(std::vector<double>& p) : poses(p) {}
  template <class Frame>
  int32_t index(const Frame* fr) {
    for (size_t k = 0; k < frames.size(); ++k) if (frames[k] == (const void*)fr) return (int32_t)k;
    frames.push_back((const void*)fr);
    double T[7];
    typedef typename std::remove_cv<typename std::remove_reference<decltype(fr->T_f_w_)>::type>::type SE3T;
    se3_traits<SE3T>::get(fr->T_f_w_, T);
    poses.insert(poses.end(), T, T + 7);
    return (int32_t)(frames.size() - 1);
  }
};
}  // namespace plsvo_hip_adapter

namespace plsvo {
namespace structure_optimizer {

/// Batched replacement of the two loops of FrameHandlerBase::optimizeStructure (src/frame_handler_base.cpp:213-236):
///     for (it in selected points)  (*it)->optimize(max_iter);          -> Point::optimize    (src/feature3D_impl.cpp:36-95)
///     for (it in selected segs)    (*it)->optimize(max_iter_segs);     -> LineSeg::optimize  (src/feature3D_impl.cpp:97-174)
/// The selection (nth_element on last_structure_optim_) and the last_structure_optim_ bookkeeping stay with the
/// caller.  Iterators range over Point* / LineSeg*; observations are read from Feature3D::obs_
/// (include/plsvo/feature3D.h) as (feature->frame->T_f_w_, feature->f | sf | ef).  Writes pos_ / spos_ / epos_.
template <class PointIt, class SegIt>
bool optimize(PointIt pts_begin, PointIt pts_end, size_t n_iter, SegIt segs_begin, SegIt segs_end, size_t n_iter_segs) {
  using namespace plsvo_hip_adapter;
  Context& c = default_context();
  if (!c.ctx && !c.ensure(64, 64, 1)) return false;
  std::vector<double> frame_T, pt_pos, pt_f, sp, ep, sf, ef;
  std::vector<int32_t> pt_off(1, 0), pt_fr, sg_off(1, 0), sg_fr;
  FrameTable frames_tab(frame_T);   // frame identity -> index into frame_T
#define frame_index(fr) frames_tab.index(fr)
  for (PointIt it = pts_begin; it != pts_end; ++it) {
    for (int k = 0; k < 3; ++k) pt_pos.push_back((*it)->pos_[k]);
    for (auto o = (*it)->obs_.begin(); o != (*it)->obs_.end(); ++o) {
      pt_fr.push_back(frame_index((*o)->frame));
      for (int k = 0; k < 3; ++k) pt_f.push_back((*o)->f[k]);
    }
    pt_off.push_back((int32_t)pt_fr.size());
  }
  for (SegIt it = segs_begin; it != segs_end; ++it) {
    for (int k = 0; k < 3; ++k) { sp.push_back((*it)->spos_[k]); ep.push_back((*it)->epos_[k]); }
    for (auto o = (*it)->obs_.begin(); o != (*it)->obs_.end(); ++o) {
      sg_fr.push_back(frame_index((*o)->frame));
      for (int k = 0; k < 3; ++k) { sf.push_back((*o)->sf[k]); ef.push_back((*o)->ef[k]); }
    }
    sg_off.push_back((int32_t)sg_fr.size());
  }
  plsvo_structopt_in in;
#undef frame_index
  in.n_frames = (int32_t)frames_tab.frames.size(); in.n_iter_pts = (int32_t)n_iter; in.n_iter_segs = (int32_t)n_iter_segs;
  in.n_pts = (int32_t)pt_off.size() - 1; in.n_seg = (int32_t)sg_off.size() - 1; in.reserved0 = 0;
  in.frame_T = frame_T.data(); in.pt_pos = pt_pos.data(); in.pt_obs_off = pt_off.data(); in.pt_obs_frame = pt_fr.data(); in.pt_obs_f = pt_f.data();
  in.seg_spos = sp.data(); in.seg_epos = ep.data(); in.seg_obs_off = sg_off.data(); in.seg_obs_frame = sg_fr.data();
  in.seg_obs_sf = sf.data(); in.seg_obs_ef = ef.data();
  std::vector<double> o_pt(pt_pos.size() + 1), o_s(sp.size() + 1), o_e(ep.size() + 1);
  plsvo_structopt_out out;
  out.pt_pos = o_pt.data(); out.seg_spos = o_s.data(); out.seg_epos = o_e.data(); out.pt_iters = nullptr; out.seg_iters = nullptr;
  if (plsvo_structure_optimize(c.ctx, &in, &out) != PLSVO_OK) {
    std::fprintf(stderr, "[plsvo_hip] structure_optimize failed: %s\n", plsvo_hip_last_error(c.ctx));
    return false;
  }
  size_t i = 0;
  for (PointIt it = pts_begin; it != pts_end; ++it, ++i) for (int k = 0; k < 3; ++k) (*it)->pos_[k] = o_pt[3 * i + k];
  i = 0;
  for (SegIt it = segs_begin; it != segs_end; ++it, ++i) for (int k = 0; k < 3; ++k) { (*it)->spos_[k] = o_s[3 * i + k]; (*it)->epos_[k] = o_e[3 * i + k]; }
  return true;
}

}  // namespace structure_optimizer
}  // namespace plsvo

namespace plsvo_hip_adapter {
// Frames referenced by one batch of the direct matcher / the seed update: index -> (pose, pyramid slot).  The pyramids are
// uploaded on first sight and cached on the device by (address, id_) in the context's slots 2.. (Context::lookup).
struct FrameRegistry {
  std::vector<const void*> frames;
  std::vector<double> frame_T;
  std::vector<int32_t> frame_slot;
  plsvo_pinhole cam;
  bool ok;
  unsigned long batch_start;
  unsigned long generation;   // Context::generation when the first frame was registered
  FrameRegistry() : ok(true), batch_start(0), generation(0) {}
  void clear() { frames.clear(); frame_T.clear(); frame_slot.clear(); ok = true; batch_start = 0; generation = 0; }
  /// the slots handed out are only valid while the slab they index has not been re-allocated
  bool valid() const { return ok && (frames.empty() || generation == default_context().generation); }
  template <class FrameT>
  int index(const FrameT* fr) {
    for (size_t k = 0; k < frames.size(); ++k) if (frames[k] == (const void*)fr) return (int)k;
    Context& c = default_context();
    typedef typename std::remove_reference<decltype(*fr->cam_)>::type Cam;
    if (!camera_supported<Cam>(*fr->cam_)) ok = false;
    const plsvo_pinhole cm = camera_traits<Cam>::get(*fr->cam_);
    const int n_levels = (int)fr->img_pyr_.size();
    if (frames.empty()) cam = cm;
    if (!c.ensure(cm.width, cm.height, n_levels)) ok = false;
    if (frames.empty()) generation = c.generation;
    else if (generation != c.generation) {
      // a frame with a different size or a deeper pyramid re-allocated the slab: the slots of the frames registered
      // earlier in this batch now point at zeroed memory
      if (ok) std::fprintf(stderr, "[plsvo_hip] frames of one batch differ in image size or pyramid depth (the slab was re-allocated): batch refused\n");
      ok = false;
    }
    int slot = 0;
    if (ok) {
      slot = c.slot_for(fr, n_levels < c.levels ? n_levels : c.levels, &batch_start);
      if (slot < 0) { ok = false; slot = 0; }
    }
    frames.push_back((const void*)fr);
    double T[7];
    typedef typename std::remove_cv<typename std::remove_reference<decltype(fr->T_f_w_)>::type>::type SE3T;
    se3_traits<SE3T>::get(fr->T_f_w_, T);
    frame_T.insert(frame_T.end(), T, T + 7);
    frame_slot.push_back(slot);
    return (int)frames.size() - 1;
  }
};
}  // namespace plsvo_hip_adapter

namespace plsvo {

/// Batched replacement of Matcher::findMatchDirect (include/plsvo/matcher.h:120-131, src/matcher.cpp:159-275), the
/// call Reprojector::refineBestCandidate makes once per candidate (src/reprojector.cpp:288, :348):
///
///     DirectMatcher m(Config::nPyrLevels());                       // options_.align_max_iter = 10
///     for every candidate:   if (pt->getCloseViewObs(frame->pos(), ref_ftr))  h = m.addPoint(pt->pos_, ref_ftr, *frame, px_est);
///                            ... m.addSegment(ls->spos_, ls->epos_, ref_ftr, *frame, spx_est, epx_est);
///     m.run();                                                     // ONE kernel launch for all of them
///     for every candidate, in the original order:  if (m.found(h)) { m.px(h, px_est); ... }
///
/// The closest-view choice (Point::getCloseViewObs, map logic) and the per-cell "first success wins" bookkeeping stay
/// with the caller: evaluating every queued candidate and then walking them in the reference's order gives the
/// reference's result, because candidates do not influence one another.  Reads, duck-typed: Feature::frame / px / f /
/// level, PointFeat::type / grad, LineFeat::spx / epx / sf / ef, Frame::T_f_w_ / cam_ / img_pyr_ / id_.
class DirectMatcher {
 public:
  explicit DirectMatcher(int n_pyr_levels, int align_max_iter = 10) : n_pyr_levels_(n_pyr_levels), align_max_iter_(align_max_iter) {}

  void clear() {
    reg_.clear(); cur_frame_.clear(); ref_frame_.clear(); ref_px_.clear(); ref_f_.clear();
    ref_level_.clear(); ref_type_.clear(); ref_grad_.clear(); pos_.clear(); px_cur_.clear(); items_.clear(); out_px_.clear(); out_found_.clear();
    out_level_.clear();
  }

  /// queue findMatchDirect(pt, cur_frame, px_cur); ref_ftr is what pt.getCloseViewObs(cur_frame.pos(), ref_ftr) returned
  template <class Pos, class PointFeatT, class FrameT, class Px>
  int addPoint(const Pos& pos, const PointFeatT* ref_ftr, const FrameT& cur_frame, const Px& px_cur) {
    const int cf = frame_index(&cur_frame), rf = frame_index(ref_ftr->frame);
    const bool edgelet = (int)ref_ftr->type == (int)PointFeatT::EDGELET;
    push(cf, rf, ref_ftr->px, ref_ftr->f, ref_ftr->level, edgelet ? PLSVO_FTR_EDGELET : PLSVO_FTR_CORNER, ref_ftr->grad[0], ref_ftr->grad[1], pos, px_cur);
    items_.push_back(Item{(int)ref_level_.size() - 1, 1});
    return (int)items_.size() - 1;
  }
  /// queue findMatchDirect(ls, cur_frame, spx_cur, epx_cur)
  template <class Pos, class LineFeatT, class FrameT, class Px>
  int addSegment(const Pos& spos, const Pos& epos, const LineFeatT* ref_ftr, const FrameT& cur_frame, const Px& spx_cur, const Px& epx_cur) {
    const int cf = frame_index(&cur_frame), rf = frame_index(ref_ftr->frame);
    push(cf, rf, ref_ftr->spx, ref_ftr->sf, ref_ftr->level, PLSVO_FTR_CORNER, 0.0, 0.0, spos, spx_cur);
    push(cf, rf, ref_ftr->epx, ref_ftr->ef, ref_ftr->level, PLSVO_FTR_CORNER, 0.0, 0.0, epos, epx_cur);
    items_.push_back(Item{(int)ref_level_.size() - 2, 2});
    return (int)items_.size() - 1;
  }

  /// one launch over everything queued; false (and found() == false everywhere) when the device path failed
  bool run() {
    using namespace plsvo_hip_adapter;
    const size_t n = ref_level_.size();
    out_px_.assign(px_cur_.begin(), px_cur_.end()); out_found_.assign(n ? n : 1, 0); out_level_.assign(n ? n : 1, -1);
    if (n == 0) return true;
    if (!reg_.valid()) return false;
    plsvo_match_in in;
    in.cam = reg_.cam; in.n_pyr_levels = n_pyr_levels_; in.align_max_iter = align_max_iter_;
    in.n_frames = (int32_t)reg_.frames.size(); in.n = (int32_t)n;
    in.frame_T = reg_.frame_T.data(); in.frame_slot = reg_.frame_slot.data(); in.cur_frame = cur_frame_.data(); in.ref_frame = ref_frame_.data();
    in.ref_px = ref_px_.data(); in.ref_f = ref_f_.data(); in.ref_level = ref_level_.data(); in.ref_type = ref_type_.data();
    in.ref_grad = ref_grad_.data(); in.pos = pos_.data(); in.px_cur = px_cur_.data();
    plsvo_match_out out;
    out.px_cur = out_px_.data(); out.found = out_found_.data(); out.search_level = out_level_.data(); out.n_iter = nullptr;
    Context& c = default_context();
    if (plsvo_match_direct(c.ctx, &in, &out) != PLSVO_OK) {
      std::fprintf(stderr, "[plsvo_hip] match_direct failed: %s\n", plsvo_hip_last_error(c.ctx));
      out_px_.assign(px_cur_.begin(), px_cur_.end()); out_found_.assign(n, 0);
      return false;
    }
    return true;
  }

  /// findMatchDirect's return value: a point's flag, or start & end for a segment (matcher.cpp:253-274)
  bool found(int h) const {
    const Item& it = items_[(size_t)h];
    bool f = true;
    for (int k = 0; k < it.n; ++k) f = f & (out_found_[(size_t)(it.first + k)] != 0);
    return f;
  }
  /// refined position of a point candidate (px_cur on return from findMatchDirect)
  template <class Px> void px(int h, Px& out) const { const size_t i = (size_t)items_[(size_t)h].first; out[0] = out_px_[2 * i]; out[1] = out_px_[2 * i + 1]; }
  /// refined end points of a segment candidate (spx_cur / epx_cur)
  template <class Px> void segment_px(int h, Px& spx, Px& epx) const {
    const size_t i = (size_t)items_[(size_t)h].first;
    spx[0] = out_px_[2 * i]; spx[1] = out_px_[2 * i + 1]; epx[0] = out_px_[2 * i + 2]; epx[1] = out_px_[2 * i + 3];
  }
  /// Matcher::search_level_ after the call (for a segment: of its end point, like the member the reference leaves behind)
  int search_level(int h) const { const Item& it = items_[(size_t)h]; return out_level_[(size_t)(it.first + it.n - 1)]; }
  size_t size() const { return items_.size(); }

 private:
  struct Item { int first, n; };

  template <class FrameT>
  int frame_index(const FrameT* fr) { return reg_.index(fr); }
  template <class V2, class V3, class Pos, class Px>
  void push(int cf, int rf, const V2& px, const V3& f, int level, uint8_t type, double g0, double g1, const Pos& pos, const Px& px_cur) {
    cur_frame_.push_back(cf); ref_frame_.push_back(rf); ref_level_.push_back(level); ref_type_.push_back(type);
    ref_px_.push_back(px[0]); ref_px_.push_back(px[1]);
    for (int k = 0; k < 3; ++k) { ref_f_.push_back(f[k]); pos_.push_back(pos[k]); }
    ref_grad_.push_back(g0); ref_grad_.push_back(g1);
    px_cur_.push_back(px_cur[0]); px_cur_.push_back(px_cur[1]);
  }

  int n_pyr_levels_, align_max_iter_;
  plsvo_hip_adapter::FrameRegistry reg_;
  std::vector<double> ref_px_, ref_f_, ref_grad_, pos_, px_cur_, out_px_;
  std::vector<int32_t> cur_frame_, ref_frame_, ref_level_, out_level_;
  std::vector<uint8_t> ref_type_, out_found_;
  std::vector<Item> items_;
};

}  // namespace plsvo

namespace plsvo {
namespace depth_filter {

/// The reference's defaults (DepthFilter::Options include/plsvo/depth_filter.h:113-131, Matcher::Options matcher.h:88-106)
struct SeedUpdateOptions {
  int n_pyr_levels;                  // Config::nPyrLevels()
  int max_n_kfs;                     // DepthFilter::Options::max_n_kfs
  double seed_convergence_sigma2_thresh;
  int align_max_iter, max_epi_search_steps;
  bool epi_search_edgelet_filtering;
  double epi_search_edgelet_max_angle;
  SeedUpdateOptions() : n_pyr_levels(3), max_n_kfs(3), seed_convergence_sigma2_thresh(200.0), align_max_iter(10), max_epi_search_steps(1000),
                        epi_search_edgelet_filtering(true), epi_search_edgelet_max_angle(0.7) {}
};

/// Batched replacement of DepthFilter::updateSeeds(frame) (src/depth_filter.cpp:262-471): every seed of both lists is
/// updated against `frame` in ONE launch, then the lists are walked once in the reference's order to apply exactly its
/// mutations: seeds older than max_n_kfs batches are erased (:289-292, :386-389), `b++` on a failed search, the new
/// a/b/mu/sigma2, converged seeds are handed to the callbacks with their world position(s) and erased (:335-358,
/// :438-462), NaN seeds are erased.  The callbacks create the landmark (`new Point(xyz_world, it->ftr)`,
/// `it->ftr->feat3D = point`, seed_converged_cb_) -- map objects stay host business:
///     on_point(seed, xyz_world[3]);   on_segment(seed, xyz_world_s[3], xyz_world_e[3]);
/// Seed types are duck-typed on the reference's members (batch_id, ftr, a, b, mu, z_range, sigma2 / the _s _e pairs).
/// Not reproduced: seeds_updating_halt_ (a launch is not interruptible) and the detector's setGridOccpuancy
/// (:327-331; feature detection is outside this path).
template <class FrameT, class PointSeedList, class LineSeedList, class OnPoint, class OnSegment>
bool updateSeeds(const FrameT& frame, PointSeedList& pt_seeds, LineSeedList& seg_seeds, int batch_counter, const SeedUpdateOptions& opt,
                 OnPoint on_point, OnSegment on_segment) {
  using namespace plsvo_hip_adapter;
  for (auto it = pt_seeds.begin(); it != pt_seeds.end();) { if ((batch_counter - it->batch_id) > opt.max_n_kfs) it = pt_seeds.erase(it); else ++it; }
  for (auto it = seg_seeds.begin(); it != seg_seeds.end();) { if ((batch_counter - it->batch_id) > opt.max_n_kfs) it = seg_seeds.erase(it); else ++it; }
  FrameRegistry reg;
  std::vector<int32_t> prf, pcf, plv, srf, scf, slv;
  std::vector<double> ppx, pf, pg, spx, sfc, ssf, sef;
  std::vector<uint8_t> ptype;
  std::vector<float> pa, pb, pmu, pzr, ps2, sa, sb, smus, smue, szrs, szre, ss2s, ss2e;
  const int cf = pt_seeds.empty() && seg_seeds.empty() ? 0 : reg.index(&frame);
  for (auto it = pt_seeds.begin(); it != pt_seeds.end(); ++it) {
    typedef typename std::remove_reference<decltype(*it->ftr)>::type Ftr;
    prf.push_back(reg.index(it->ftr->frame)); pcf.push_back(cf); plv.push_back(it->ftr->level);
    ppx.push_back(it->ftr->px[0]); ppx.push_back(it->ftr->px[1]);
    for (int k = 0; k < 3; ++k) pf.push_back(it->ftr->f[k]);
    pg.push_back(it->ftr->grad[0]); pg.push_back(it->ftr->grad[1]);
    ptype.push_back((int)it->ftr->type == (int)Ftr::EDGELET ? PLSVO_FTR_EDGELET : PLSVO_FTR_CORNER);
    pa.push_back(it->a); pb.push_back(it->b); pmu.push_back(it->mu); pzr.push_back(it->z_range); ps2.push_back(it->sigma2);
  }
  for (auto it = seg_seeds.begin(); it != seg_seeds.end(); ++it) {
    srf.push_back(reg.index(it->ftr->frame)); scf.push_back(cf); slv.push_back(it->ftr->level);
    spx.push_back(it->ftr->px[0]); spx.push_back(it->ftr->px[1]);
    for (int k = 0; k < 3; ++k) { sfc.push_back(it->ftr->f[k]); ssf.push_back(it->ftr->sf[k]); sef.push_back(it->ftr->ef[k]); }
    sa.push_back(it->a); sb.push_back(it->b); smus.push_back(it->mu_s); smue.push_back(it->mu_e); szrs.push_back(it->z_range_s); szre.push_back(it->z_range_e);
    ss2s.push_back(it->sigma2_s); ss2e.push_back(it->sigma2_e);
  }
  const size_t np = pa.size(), ns = sa.size();
  if (np + ns == 0) return true;
  if (!reg.valid()) return false;
  plsvo_seeds_in in;
  in.cam = reg.cam; in.n_pyr_levels = opt.n_pyr_levels; in.align_max_iter = opt.align_max_iter; in.max_epi_search_steps = opt.max_epi_search_steps;
  in.edgelet_filtering = opt.epi_search_edgelet_filtering ? 1 : 0; in.edgelet_max_angle = opt.epi_search_edgelet_max_angle; in.px_noise = 1.0;
  in.convergence_sigma2_thresh = opt.seed_convergence_sigma2_thresh;
  in.n_frames = (int32_t)reg.frames.size(); in.n_pt = (int32_t)np; in.n_seg = (int32_t)ns; in.reserved0 = 0;
  in.frame_T = reg.frame_T.data(); in.frame_slot = reg.frame_slot.data();
  in.pt_ref_frame = prf.data(); in.pt_cur_frame = pcf.data(); in.pt_px = ppx.data(); in.pt_f = pf.data(); in.pt_level = plv.data(); in.pt_type = ptype.data();
  in.pt_grad = pg.data(); in.pt_a = pa.data(); in.pt_b = pb.data(); in.pt_mu = pmu.data(); in.pt_z_range = pzr.data(); in.pt_sigma2 = ps2.data();
  in.seg_ref_frame = srf.data(); in.seg_cur_frame = scf.data(); in.seg_px = spx.data(); in.seg_f = sfc.data(); in.seg_sf = ssf.data(); in.seg_ef = sef.data();
  in.seg_level = slv.data(); in.seg_a = sa.data(); in.seg_b = sb.data(); in.seg_mu_s = smus.data(); in.seg_mu_e = smue.data();
  in.seg_z_range_s = szrs.data(); in.seg_z_range_e = szre.data(); in.seg_sigma2_s = ss2s.data(); in.seg_sigma2_e = ss2e.data();
  std::vector<int32_t> pst(np + 1), sst(ns + 1);
  std::vector<float> oa(np + 1), ob(np + 1), omu(np + 1), os2(np + 1), osa(ns + 1), osb(ns + 1), osmus(ns + 1), osmue(ns + 1), oss2s(ns + 1), oss2e(ns + 1);
  std::vector<double> oxyz(3 * np + 3), oxs(3 * ns + 3), oxe(3 * ns + 3);
  plsvo_seeds_out out;
  out.pt_status = pst.data(); out.pt_a = oa.data(); out.pt_b = ob.data(); out.pt_mu = omu.data(); out.pt_sigma2 = os2.data(); out.pt_xyz_world = oxyz.data();
  out.pt_px_cur = nullptr; out.pt_depth = nullptr;
  out.seg_status = sst.data(); out.seg_a = osa.data(); out.seg_b = osb.data(); out.seg_mu_s = osmus.data(); out.seg_mu_e = osmue.data();
  out.seg_sigma2_s = oss2s.data(); out.seg_sigma2_e = oss2e.data(); out.seg_xyz_world_s = oxs.data(); out.seg_xyz_world_e = oxe.data();
  out.seg_depth_s = nullptr; out.seg_depth_e = nullptr;
  Context& c = default_context();
  if (plsvo_update_seeds(c.ctx, &in, &out) != PLSVO_OK) {
    std::fprintf(stderr, "[plsvo_hip] update_seeds failed: %s\n", plsvo_hip_last_error(c.ctx));
    return false;
  }
  size_t i = 0;
  for (auto it = pt_seeds.begin(); it != pt_seeds.end(); ++i) {
    it->a = oa[i]; it->b = ob[i]; it->mu = omu[i]; it->sigma2 = os2[i];
    if (pst[i] == PLSVO_SEED_CONVERGED) { on_point(*it, &oxyz[3 * i]); it = pt_seeds.erase(it); }
    else if (pst[i] == PLSVO_SEED_NAN) it = pt_seeds.erase(it);
    else ++it;
  }
  i = 0;
  for (auto it = seg_seeds.begin(); it != seg_seeds.end(); ++i) {
    it->a = osa[i]; it->b = osb[i]; it->mu_s = osmus[i]; it->mu_e = osmue[i]; it->sigma2_s = oss2s[i]; it->sigma2_e = oss2e[i];
    if (sst[i] == PLSVO_SEED_CONVERGED) { on_segment(*it, &oxs[3 * i], &oxe[3 * i]); it = seg_seeds.erase(it); }
    else if (sst[i] == PLSVO_SEED_NAN) it = seg_seeds.erase(it);
    else ++it;
  }
  return true;
}

}  // namespace depth_filter
}  // namespace plsvo

namespace svo = plsvo;  // BASELINE.json spells the upstream name svo::SparseImgAlign
