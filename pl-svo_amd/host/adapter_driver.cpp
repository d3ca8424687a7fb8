// adapter_driver.cpp -- exercises the C++ adapter (hip_adapter.hpp) exactly the way
// FrameHandlerMono::processFrame does (src/frame_handler_mono.cpp:266-274, 327-329), on frames built from a
// binary dump written by tests/test_gpu_adapter.py.  Prints the mutated state for the test to compare with
// the oracle.  Usage: adapter_driver <input.bin> <output.txt> [structure.bin]
//             adapter_driver --bench K <input.bin>     per-call wall time of run() and optimizeGaussNewton() over K calls,
//                                                      every call with a NEW current frame (so its pyramid is uploaded, as
//                                                      in a live pipeline; the reference frame is the cached previous one)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "plsvo/hip_adapter.hpp"
#include "plsvo/mini_types.hpp"

typedef plsvo::SparseImgAlignT<mini::FramePtr> SparseImgAlign;

static std::vector<double> read_doubles(FILE* f, size_t n) { std::vector<double> v(n); if (n && fread(v.data(), 8, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } return v; }

static double median_of(std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; }

int main(int argc, char** argv) {
  int bench_calls = 0;
  if (argc >= 4 && !strcmp(argv[1], "--bench")) { bench_calls = atoi(argv[2]); argv += 2; argc -= 2; }
  if (argc < (bench_calls ? 2 : 3)) { fprintf(stderr, "usage: %s in.bin out.txt [structure.bin] | --bench K in.bin\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("open"); return 2; }
  std::vector<double> hdr = read_doubles(f, 12);
  const int W = (int)hdr[0], H = (int)hdr[1], n_levels = (int)hdr[2], max_level = (int)hdr[3], min_level = (int)hdr[4];
  const int n_pts = (int)hdr[5], n_seg = (int)hdr[6], po_pts = (int)hdr[7], po_seg = (int)hdr[8], n_dead_seg = (int)hdr[9];
  mini::Camera cam;
  std::vector<double> c = read_doubles(f, 4);
  cam.fx_ = c[0]; cam.fy_ = c[1]; cam.cx_ = c[2]; cam.cy_ = c[3]; cam.w_ = W; cam.h_ = H;
  std::vector<double> Tr = read_doubles(f, 7), Tc = read_doubles(f, 7);
  mini::FramePtr ref(new mini::Frame()), cur(new mini::Frame());
  ref->cam_ = cur->cam_ = &cam;
  ref->T_f_w_ = mini::SE3(mini::Quat(Tr[3], Tr[0], Tr[1], Tr[2]), mini::Vec3(Tr[4], Tr[5], Tr[6]));
  cur->T_f_w_ = mini::SE3(mini::Quat(Tc[3], Tc[0], Tc[1], Tc[2]), mini::Vec3(Tc[4], Tc[5], Tc[6]));
  for (int which = 0; which < 2; ++which) {
    mini::Frame& fr = which ? *cur : *ref;
    fr.img_pyr_.resize((size_t)n_levels);
    for (int l = 0; l < n_levels; ++l) {
      fr.img_pyr_[(size_t)l].alloc(W >> l, H >> l);
      const size_t nb = (size_t)(W >> l) * (H >> l);
      if (fread(fr.img_pyr_[(size_t)l].data, 1, nb, f) != nb) { fprintf(stderr, "short image read\n"); return 2; }
    }
  }
  std::vector<mini::Point> pts((size_t)n_pts); std::vector<mini::PointFeat> pfs((size_t)n_pts);
  std::vector<double> pd = read_doubles(f, (size_t)n_pts * 8);
  for (int i = 0; i < n_pts; ++i) {
    const double* d = &pd[(size_t)i * 8];
    pfs[(size_t)i].px = mini::Vec2(d[0], d[1]); pfs[(size_t)i].f = mini::Vec3(d[2], d[3], d[4]);
    pts[(size_t)i].pos_ = mini::Vec3(d[5], d[6], d[7]); pfs[(size_t)i].feat3D = &pts[(size_t)i];
    ref->pt_fts_.push_back(&pfs[(size_t)i]);
  }
  std::vector<mini::LineSeg> lss((size_t)n_seg); std::vector<mini::LineFeat> lfs((size_t)n_seg);
  std::vector<double> sd = read_doubles(f, (size_t)n_seg * 17);
  for (int i = 0; i < n_seg; ++i) {
    const double* d = &sd[(size_t)i * 17];
    mini::LineFeat& L = lfs[(size_t)i];
    L.spx = mini::Vec2(d[0], d[1]); L.epx = mini::Vec2(d[2], d[3]); L.sf = mini::Vec3(d[4], d[5], d[6]); L.ef = mini::Vec3(d[7], d[8], d[9]);
    lss[(size_t)i].spos_ = mini::Vec3(d[10], d[11], d[12]); lss[(size_t)i].epos_ = mini::Vec3(d[13], d[14], d[15]);
    L.length = d[16];
    L.feat3D = (i < n_dead_seg) ? nullptr : &lss[(size_t)i];   // the first n_dead_seg segments have no landmark
    ref->seg_fts_.push_back(&L);
  }
  if (bench_calls > 0) {
    // ---- per-call latency of the drop-in, as FrameHandlerMono::processFrame pays it (src/frame_handler_mono.cpp:266-274, 327-329) ----
    std::vector<double> Tp = read_doubles(f, 7);
    mini::FramePtr fr(new mini::Frame());
    fr->cam_ = &cam;
    std::vector<mini::Point> qpts((size_t)po_pts); std::vector<mini::PointFeat> qpfs((size_t)po_pts);
    std::vector<double> qd = read_doubles(f, (size_t)po_pts * 7);
    std::vector<mini::LineSeg> qls((size_t)po_seg); std::vector<mini::LineFeat> qlf((size_t)po_seg);
    std::vector<double> qs = read_doubles(f, (size_t)po_seg * 10);
    fclose(f);
    for (int i = 0; i < po_pts; ++i) {
      const double* d = &qd[(size_t)i * 7];
      qpfs[(size_t)i].f = mini::Vec3(d[0], d[1], d[2]); qpts[(size_t)i].pos_ = mini::Vec3(d[3], d[4], d[5]); qpfs[(size_t)i].level = (int)d[6];
      fr->pt_fts_.push_back(&qpfs[(size_t)i]);
    }
    for (int i = 0; i < po_seg; ++i) {
      const double* d = &qs[(size_t)i * 10];
      qlf[(size_t)i].line = mini::Vec3(d[0], d[1], d[2]); qls[(size_t)i].spos_ = mini::Vec3(d[3], d[4], d[5]); qls[(size_t)i].epos_ = mini::Vec3(d[6], d[7], d[8]);
      qlf[(size_t)i].level = (int)d[9];
      fr->seg_fts_.push_back(&qlf[(size_t)i]);
    }
    ref->id_ = 1;
    std::vector<double> t_run, t_opt;
    size_t tracked = 0;
    for (int k = -3; k < bench_calls; ++k) {   // three untimed warm-up calls (context creation, first allocations)
      cur->id_ = 1000 + k + 3;                 // a NEW frame identity: its pyramid is uploaded; ref stays cached
      cur->T_f_w_ = mini::SE3(mini::Quat(Tc[3], Tc[0], Tc[1], Tc[2]), mini::Vec3(Tc[4], Tc[5], Tc[6]));
      for (int i = 0; i < n_seg; ++i) lfs[(size_t)i].feat3D = (i < n_dead_seg) ? nullptr : &lss[(size_t)i];
      fr->T_f_w_ = mini::SE3(mini::Quat(Tp[3], Tp[0], Tp[1], Tp[2]), mini::Vec3(Tp[4], Tp[5], Tp[6]));
      for (int i = 0; i < po_pts; ++i) qpfs[(size_t)i].feat3D = &qpts[(size_t)i];
      for (int i = 0; i < po_seg; ++i) qlf[(size_t)i].feat3D = &qls[(size_t)i];
      const auto t0 = std::chrono::steady_clock::now();
      SparseImgAlign img_align(max_level, min_level, 30, SparseImgAlign::GaussNewton, false, false);
      tracked = img_align.run(ref, cur);
      const auto t1 = std::chrono::steady_clock::now();
      size_t n_pt = 0, n_ls = 0; double thresh = 0, e0 = 0, e1 = 0;
      plsvo::pose_optimizer::optimizeGaussNewton(2.0, (size_t)10, false, fr, thresh, e0, e1, n_pt, n_ls);
      const auto t2 = std::chrono::steady_clock::now();
      if (k >= 0) {
        t_run.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        t_opt.push_back(std::chrono::duration<double, std::micro>(t2 - t1).count());
      }
    }
    double m_run = 0, m_opt = 0;
    for (double v : t_run) m_run += v;
    for (double v : t_opt) m_opt += v;
    printf("{\"calls\": %d, \"n_tracked\": %zu, \"run_us_mean\": %.1f, \"run_us_median\": %.1f, \"poseopt_us_mean\": %.1f, \"poseopt_us_median\": %.1f}\n",
           bench_calls, tracked, m_run / t_run.size(), median_of(t_run), m_opt / t_opt.size(), median_of(t_opt));
    return 0;
  }
  FILE* o = fopen(argv[2], "w");
  const bool verbose = getenv("PLSVO_DRIVER_VERBOSE") != nullptr;   // the reference's verbose flags (its call sites pass false)
  // ---- step 2 of processFrame: sparse image alignment ----
  SparseImgAlign img_align(max_level, min_level, 30, SparseImgAlign::GaussNewton, false, verbose);
  const size_t n_tracked = img_align.run(ref, cur);
  fprintf(o, "n_tracked %zu\n", n_tracked);
  fprintf(o, "T_cur %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", cur->T_f_w_.q.x(), cur->T_f_w_.q.y(), cur->T_f_w_.q.z(), cur->T_f_w_.q.w(),
          cur->T_f_w_.t[0], cur->T_f_w_.t[1], cur->T_f_w_.t[2]);
  fprintf(o, "alive");
  for (int i = 0; i < n_seg; ++i) fprintf(o, " %d", lfs[(size_t)i].feat3D != nullptr ? 1 : 0);
  fprintf(o, "\n");
  const mini::Mat66 I = img_align.getFisherInformation();   // by value, as include/plsvo/sparse_img_align.h:69 declares it
  mini::Mat66 I2; img_align.getFisherInformation(I2);
  fprintf(o, "fisher00 %.17g\n", I(0, 0));
  if (I(0, 0) != I2(0, 0) || I(5, 5) != I2(5, 5)) { fprintf(stderr, "getFisherInformation overloads disagree\n"); return 3; }

  // ---- step 4 of processFrame: pose optimisation on a second frame ----
  std::vector<double> Tp = read_doubles(f, 7);
  mini::FramePtr fr(new mini::Frame());
  fr->cam_ = &cam;
  fr->T_f_w_ = mini::SE3(mini::Quat(Tp[3], Tp[0], Tp[1], Tp[2]), mini::Vec3(Tp[4], Tp[5], Tp[6]));
  std::vector<mini::Point> qpts((size_t)po_pts); std::vector<mini::PointFeat> qpfs((size_t)po_pts);
  std::vector<double> qd = read_doubles(f, (size_t)po_pts * 7);
  for (int i = 0; i < po_pts; ++i) {
    const double* d = &qd[(size_t)i * 7];
    qpfs[(size_t)i].f = mini::Vec3(d[0], d[1], d[2]); qpts[(size_t)i].pos_ = mini::Vec3(d[3], d[4], d[5]); qpfs[(size_t)i].level = (int)d[6];
    qpfs[(size_t)i].feat3D = &qpts[(size_t)i];
    fr->pt_fts_.push_back(&qpfs[(size_t)i]);
  }
  std::vector<mini::LineSeg> qls((size_t)po_seg); std::vector<mini::LineFeat> qlf((size_t)po_seg);
  std::vector<double> qs = read_doubles(f, (size_t)po_seg * 10);
  for (int i = 0; i < po_seg; ++i) {
    const double* d = &qs[(size_t)i * 10];
    qlf[(size_t)i].line = mini::Vec3(d[0], d[1], d[2]); qls[(size_t)i].spos_ = mini::Vec3(d[3], d[4], d[5]); qls[(size_t)i].epos_ = mini::Vec3(d[6], d[7], d[8]);
    qlf[(size_t)i].level = (int)d[9]; qlf[(size_t)i].feat3D = &qls[(size_t)i];
    fr->seg_fts_.push_back(&qlf[(size_t)i]);
  }
  fclose(f);
  size_t sfba_n_edges_final_pt = 0, sfba_n_edges_final_ls = 0;
  double sfba_thresh = 0, sfba_error_init = 0, sfba_error_final = 0;
  plsvo::pose_optimizer::optimizeGaussNewton(2.0, (size_t)10, verbose, fr, sfba_thresh, sfba_error_init, sfba_error_final,
                                             sfba_n_edges_final_pt, sfba_n_edges_final_ls);
  fprintf(o, "T_opt %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", fr->T_f_w_.q.x(), fr->T_f_w_.q.y(), fr->T_f_w_.q.z(), fr->T_f_w_.q.w(),
          fr->T_f_w_.t[0], fr->T_f_w_.t[1], fr->T_f_w_.t[2]);
  fprintf(o, "scalars %.17g %.17g %.17g %zu %zu %.17g\n", sfba_thresh, sfba_error_init, sfba_error_final, sfba_n_edges_final_pt,
          sfba_n_edges_final_ls, fr->Cov_(0, 0));
  fprintf(o, "pt_keep");
  for (int i = 0; i < po_pts; ++i) fprintf(o, " %d", qpfs[(size_t)i].feat3D != nullptr ? 1 : 0);
  fprintf(o, "\nseg_keep");
  for (int i = 0; i < po_seg; ++i) fprintf(o, " %d", qlf[(size_t)i].feat3D != nullptr ? 1 : 0);
  fprintf(o, "\n");

  // ---- step 5 of processFrame: structure optimisation (optional third input file) ----
  if (argc > 3) {
    FILE* g = fopen(argv[3], "rb");
    if (!g) { perror("open"); return 2; }
    std::vector<double> h = read_doubles(g, 4);
    const int nfr = (int)h[0], nsp = (int)h[1], nss = (int)h[2], n_it = (int)h[3];
    std::vector<mini::Frame> kfs((size_t)nfr);
    for (int k = 0; k < nfr; ++k) { std::vector<double> T = read_doubles(g, 7); kfs[(size_t)k].T_f_w_ = mini::SE3(mini::Quat(T[3], T[0], T[1], T[2]), mini::Vec3(T[4], T[5], T[6])); }
    std::vector<mini::Point> lp((size_t)nsp); std::vector<mini::LineSeg> ll((size_t)nss);
    std::list<mini::PointFeat> pf_store; std::list<mini::LineFeat> lf_store;
    for (int i = 0; i < nsp; ++i) {
      std::vector<double> d = read_doubles(g, 4);
      lp[(size_t)i].pos_ = mini::Vec3(d[0], d[1], d[2]);
      for (int k = 0; k < (int)d[3]; ++k) {
        std::vector<double> ob = read_doubles(g, 4);
        pf_store.emplace_back(); mini::PointFeat& F = pf_store.back();
        F.frame = &kfs[(size_t)ob[0]]; F.f = mini::Vec3(ob[1], ob[2], ob[3]);
        lp[(size_t)i].obs_.push_back(&F);
      }
    }
    for (int i = 0; i < nss; ++i) {
      std::vector<double> d = read_doubles(g, 7);
      ll[(size_t)i].spos_ = mini::Vec3(d[0], d[1], d[2]); ll[(size_t)i].epos_ = mini::Vec3(d[3], d[4], d[5]);
      for (int k = 0; k < (int)d[6]; ++k) {
        std::vector<double> ob = read_doubles(g, 7);
        lf_store.emplace_back(); mini::LineFeat& F = lf_store.back();
        F.frame = &kfs[(size_t)ob[0]]; F.sf = mini::Vec3(ob[1], ob[2], ob[3]); F.ef = mini::Vec3(ob[4], ob[5], ob[6]);
        ll[(size_t)i].obs_.push_back(&F);
      }
    }
    fclose(g);
    std::vector<mini::Point*> pp; std::vector<mini::LineSeg*> sp;
    for (auto& x : lp) pp.push_back(&x);
    for (auto& x : ll) sp.push_back(&x);
    plsvo::structure_optimizer::optimize(pp.begin(), pp.end(), (size_t)n_it, sp.begin(), sp.end(), (size_t)n_it);
    for (int i = 0; i < nsp; ++i) fprintf(o, "spt %.17g %.17g %.17g\n", lp[(size_t)i].pos_[0], lp[(size_t)i].pos_[1], lp[(size_t)i].pos_[2]);
    for (int i = 0; i < nss; ++i) fprintf(o, "sseg %.17g %.17g %.17g %.17g %.17g %.17g\n", ll[(size_t)i].spos_[0], ll[(size_t)i].spos_[1], ll[(size_t)i].spos_[2],
                                          ll[(size_t)i].epos_[0], ll[(size_t)i].epos_[1], ll[(size_t)i].epos_[2]);
  }
  fclose(o);
  return 0;
}
