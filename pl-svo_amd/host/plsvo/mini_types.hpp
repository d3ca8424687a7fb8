// mini_types.hpp -- dependency-free look-alikes of the reference's data model, with the SAME member names the
// hot path reads (include/plsvo/frame.h:58-71, feature.h:37-92, feature3D.h:98-150), so that the adapter in
// hip_adapter.hpp can be compiled and tested in an image that has no Eigen / Sophus / OpenCV / boost.
// Not part of the product interface: a PL-SVO build uses its own types.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <list>
#include <memory>
#include <vector>

namespace mini {

struct Vec2 { double v[2]; Vec2() : v{0, 0} {} Vec2(double a, double b) : v{a, b} {} double& operator[](int i) { return v[i]; } const double& operator[](int i) const { return v[i]; } };
struct Vec3 { double v[3]; Vec3() : v{0, 0, 0} {} Vec3(double a, double b, double c) : v{a, b, c} {} double& operator[](int i) { return v[i]; } const double& operator[](int i) const { return v[i]; } };
struct Quat {  // Eigen::Quaterniond look-alike: ctor order (w, x, y, z)
  double x_, y_, z_, w_;
  Quat() : x_(0), y_(0), z_(0), w_(1) {}
  Quat(double w, double x, double y, double z) : x_(x), y_(y), z_(z), w_(w) {}
  double x() const { return x_; } double y() const { return y_; } double z() const { return z_; } double w() const { return w_; }
};
struct SE3 {  // Sophus::SE3 (non-templated) look-alike
  Quat q; Vec3 t;
  SE3() {}
  SE3(const Quat& q_, const Vec3& t_) : q(q_), t(t_) {}
  const Quat& unit_quaternion() const { return q; }
  const Vec3& translation() const { return t; }
};
struct Mat66 { double m[36]; Mat66() : m{} {} double& operator()(int i, int j) { return m[i * 6 + j]; } double operator()(int i, int j) const { return m[i * 6 + j]; } };
struct Image {  // cv::Mat (CV_8U) look-alike
  uint8_t* data = nullptr; int cols = 0, rows = 0; size_t step = 0;
  std::vector<uint8_t> store;
  void alloc(int w, int h) { cols = w; rows = h; step = (size_t)w; store.assign((size_t)w * h, 0); data = store.data(); }
};
struct Camera {  // vk::PinholeCamera look-alike (no distortion)
  double fx_, fy_, cx_, cy_; int w_, h_;
  double fx() const { return fx_; } double fy() const { return fy_; } double cx() const { return cx_; } double cy() const { return cy_; }
  int width() const { return w_; } int height() const { return h_; }
  double errorMultiplier2() const { return std::fabs(fx_); }
};
struct Frame;
struct PointFeat;
struct LineFeat;
struct Point { Vec3 pos_; std::list<PointFeat*> obs_; };              // Feature3D<PointFeat>::obs_
struct LineSeg { Vec3 spos_, epos_; std::list<LineFeat*> obs_; };     // Feature3D<LineFeat>::obs_
struct Feature { Frame* frame = nullptr; Vec2 px; Vec3 f; int level = 0; };
struct PointFeat : Feature { enum FeatureType { CORNER, EDGELET }; FeatureType type = CORNER; Vec2 grad; Point* feat3D = nullptr; };
struct LineFeat : Feature { Vec2 spx, epx, grad; Vec3 sf, ef, line; LineSeg* feat3D = nullptr; double length = 0; };
struct Frame {
  int id_ = 0;
  Camera* cam_ = nullptr;
  SE3 T_f_w_;
  Mat66 Cov_;
  std::vector<Image> img_pyr_;
  std::list<PointFeat*> pt_fts_;
  std::list<LineFeat*> seg_fts_;
};
// depth-filter seeds (include/plsvo/depth_filter.h:47-101)
struct PointSeed { int batch_id = 0, id = 0; PointFeat* ftr = nullptr; float a = 10, b = 10, mu = 0, z_range = 0, sigma2 = 0; };
struct LineSeed { int batch_id = 0, id = 0; LineFeat* ftr = nullptr; float a = 10, b = 10, mu_s = 0, mu_e = 0, z_range_s = 0, z_range_e = 0, sigma2_s = 0, sigma2_e = 0; };
typedef std::shared_ptr<Frame> FramePtr;

}  // namespace mini
