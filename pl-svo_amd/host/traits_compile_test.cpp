// traits_compile_test.cpp -- compile-time check of the integration recipe in INTEGRATION.md 1.1: the reference's Frame::cam_ is a
// pointer to an ABSTRACT camera (vk::AbstractCamera*, include/plsvo/frame.h:61), so a real build specialises
// plsvo_hip_adapter::camera_traits for the base class.  The recipe defines get() only; this file does exactly that, with look-alike
// types, and instantiates every adapter entry point that reads the camera.  It is built by the host Makefile (a build failure here
// means the documented recipe no longer compiles); running it does nothing.
#include <memory>

#include "plsvo/mini_types.hpp"
#include "plsvo/hip_adapter.hpp"

namespace absd {
struct AbstractCamera {                       // vk::AbstractCamera look-alike: no intrinsics on the base class
  virtual ~AbstractCamera() {}
  virtual double errorMultiplier2() const = 0;
};
struct PinholeCamera : AbstractCamera {       // vk::PinholeCamera look-alike
  double fx_, fy_, cx_, cy_, d_[5]; int w_, h_;
  PinholeCamera() : fx_(300), fy_(300), cx_(160), cy_(120), d_{0, 0, 0, 0, 0}, w_(320), h_(240) {}
  double fx() const { return fx_; } double fy() const { return fy_; } double cx() const { return cx_; } double cy() const { return cy_; }
  int width() const { return w_; } int height() const { return h_; }
  double errorMultiplier2() const override { return fx_ < 0 ? -fx_ : fx_; }
};
struct Frame {                                 // plsvo::Frame's members the hot path reads, cam_ through the abstract base
  int id_ = 0;
  AbstractCamera* cam_ = nullptr;
  mini::SE3 T_f_w_;
  mini::Mat66 Cov_;
  std::vector<mini::Image> img_pyr_;
  std::list<mini::PointFeat*> pt_fts_;
  std::list<mini::LineFeat*> seg_fts_;
};
typedef std::shared_ptr<Frame> FramePtr;
}  // namespace absd

namespace plsvo_hip_adapter {
// INTEGRATION.md 1.1, verbatim in shape: get() and nothing else
template <> struct camera_traits<absd::AbstractCamera> {
  static plsvo_pinhole get(const absd::AbstractCamera& c) {
    const absd::PinholeCamera& p = dynamic_cast<const absd::PinholeCamera&>(c);
    plsvo_pinhole o; o.fx = p.fx(); o.fy = p.fy(); o.cx = p.cx(); o.cy = p.cy(); o.width = p.width(); o.height = p.height();
    return o;
  }
};
}  // namespace plsvo_hip_adapter

int main(int argc, char**) {
  if (argc > 1000) {   // never true: the calls below only have to instantiate
    absd::FramePtr ref(new absd::Frame()), cur(new absd::Frame());
    plsvo::SparseImgAlignT<absd::FramePtr> align(2, 0, 30, plsvo::SparseImgAlignT<absd::FramePtr>::GaussNewton, false, false);
    (void)align.run(ref, cur);
    double thresh = 0, e0 = 0, e1 = 0; size_t n_pt = 0, n_ls = 0;
    plsvo::pose_optimizer::optimizeGaussNewton(2.0, (size_t)10, false, cur, thresh, e0, e1, n_pt, n_ls);
  }
  return 0;
}
