// match_driver.cpp -- exercises plsvo::DirectMatcher (hip_adapter.hpp) the way Reprojector::refineBestCandidate uses
// Matcher::findMatchDirect (src/reprojector.cpp:288, :348): a keyframe and a current frame built from a binary dump
// written by tests/test_gpu_adapter.py, candidates queued from PointFeat / LineFeat objects, one batched run, results
// printed for the test to compare with the oracle.  The batch is run twice: the second pass must hit the adapter's
// keyframe-pyramid cache and give the same answer.  With a third file the depth-filter adapter
// (plsvo::depth_filter::updateSeeds) is run on seed lists built from the same features.
// Usage: match_driver <input.bin> <output.txt> [seeds.bin]
#include <cstdio>
#include <cstdlib>
#include <list>
#include <vector>

#include "plsvo/hip_adapter.hpp"
#include "plsvo/mini_types.hpp"

static std::vector<double> read_doubles(FILE* f, size_t n) { std::vector<double> v(n); if (n && fread(v.data(), 8, n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } return v; }

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s in.bin out.txt\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("open"); return 2; }
  std::vector<double> hdr = read_doubles(f, 6);
  const int W = (int)hdr[0], H = (int)hdr[1], n_levels = (int)hdr[2], n_pts = (int)hdr[3], n_seg = (int)hdr[4], n_pyr_levels = (int)hdr[5];
  mini::Camera cam;
  std::vector<double> c = read_doubles(f, 4);
  cam.fx_ = c[0]; cam.fy_ = c[1]; cam.cx_ = c[2]; cam.cy_ = c[3]; cam.w_ = W; cam.h_ = H;
  mini::Frame kf, cur;
  kf.id_ = 10; cur.id_ = 11;
  for (int which = 0; which < 2; ++which) {
    mini::Frame& fr = which ? cur : kf;
    std::vector<double> T = read_doubles(f, 7);
    fr.cam_ = &cam;
    fr.T_f_w_ = mini::SE3(mini::Quat(T[3], T[0], T[1], T[2]), mini::Vec3(T[4], T[5], T[6]));
  }
  for (int which = 0; which < 2; ++which) {
    mini::Frame& fr = which ? cur : kf;
    fr.img_pyr_.resize((size_t)n_levels);
    for (int l = 0; l < n_levels; ++l) {
      fr.img_pyr_[(size_t)l].alloc(W >> l, H >> l);
      const size_t nb = (size_t)(W >> l) * (H >> l);
      if (fread(fr.img_pyr_[(size_t)l].data, 1, nb, f) != nb) { fprintf(stderr, "short image read\n"); return 2; }
    }
  }
  std::vector<mini::Point> pts((size_t)n_pts); std::vector<mini::PointFeat> pfs((size_t)n_pts); std::vector<mini::Vec2> px_est((size_t)n_pts);
  std::vector<double> pd = read_doubles(f, (size_t)n_pts * 14);
  for (int i = 0; i < n_pts; ++i) {
    const double* d = &pd[(size_t)i * 14];
    mini::PointFeat& F = pfs[(size_t)i];
    F.frame = &kf; F.px = mini::Vec2(d[0], d[1]); F.f = mini::Vec3(d[2], d[3], d[4]); F.level = (int)d[5];
    F.type = d[6] != 0.0 ? mini::PointFeat::EDGELET : mini::PointFeat::CORNER; F.grad = mini::Vec2(d[7], d[8]);
    pts[(size_t)i].pos_ = mini::Vec3(d[9], d[10], d[11]); F.feat3D = &pts[(size_t)i];
    px_est[(size_t)i] = mini::Vec2(d[12], d[13]);
  }
  std::vector<mini::LineSeg> lss((size_t)n_seg); std::vector<mini::LineFeat> lfs((size_t)n_seg); std::vector<mini::Vec2> spx_est((size_t)n_seg), epx_est((size_t)n_seg);
  std::vector<double> sd = read_doubles(f, (size_t)n_seg * 21);
  for (int i = 0; i < n_seg; ++i) {
    const double* d = &sd[(size_t)i * 21];
    mini::LineFeat& L = lfs[(size_t)i];
    L.frame = &kf; L.spx = mini::Vec2(d[0], d[1]); L.epx = mini::Vec2(d[2], d[3]); L.sf = mini::Vec3(d[4], d[5], d[6]); L.ef = mini::Vec3(d[7], d[8], d[9]);
    L.level = (int)d[10];
    lss[(size_t)i].spos_ = mini::Vec3(d[11], d[12], d[13]); lss[(size_t)i].epos_ = mini::Vec3(d[14], d[15], d[16]); L.feat3D = &lss[(size_t)i];
    spx_est[(size_t)i] = mini::Vec2(d[17], d[18]); epx_est[(size_t)i] = mini::Vec2(d[19], d[20]);
  }
  fclose(f);
  FILE* o = fopen(argv[2], "w");
  plsvo::DirectMatcher m(n_pyr_levels);
  for (int pass = 0; pass < 2; ++pass) {
    m.clear();
    std::vector<int> hp, hs;
    for (int i = 0; i < n_pts; ++i) hp.push_back(m.addPoint(pts[(size_t)i].pos_, &pfs[(size_t)i], cur, px_est[(size_t)i]));
    for (int i = 0; i < n_seg; ++i) hs.push_back(m.addSegment(lss[(size_t)i].spos_, lss[(size_t)i].epos_, &lfs[(size_t)i], cur, spx_est[(size_t)i], epx_est[(size_t)i]));
    if (!m.run()) { fprintf(stderr, "DirectMatcher::run failed\n"); return 3; }
    for (int i = 0; i < n_pts; ++i) {
      mini::Vec2 p = px_est[(size_t)i];
      m.px(hp[(size_t)i], p);
      fprintf(o, "pt%d %d %d %.17g %.17g\n", pass, m.found(hp[(size_t)i]) ? 1 : 0, m.search_level(hp[(size_t)i]), p[0], p[1]);
    }
    for (int i = 0; i < n_seg; ++i) {
      mini::Vec2 s, e;
      m.segment_px(hs[(size_t)i], s, e);
      fprintf(o, "seg%d %d %d %.17g %.17g %.17g %.17g\n", pass, m.found(hs[(size_t)i]) ? 1 : 0, m.search_level(hs[(size_t)i]), s[0], s[1], e[0], e[1]);
    }
  }
  // ---- DepthFilter::updateSeeds(frame) through the adapter (optional third input file) ----
  if (argc > 3) {
    FILE* g = fopen(argv[3], "rb");
    if (!g) { perror("open"); return 2; }
    std::vector<double> h = read_doubles(g, 3);
    const int nps = (int)h[0], nss = (int)h[1], batch_counter = (int)h[2];
    std::list<mini::PointSeed> pt_seeds; std::list<mini::LineSeed> seg_seeds;
    for (int i = 0; i < nps; ++i) {
      std::vector<double> d = read_doubles(g, 7);
      mini::PointSeed sd; sd.ftr = &pfs[(size_t)d[0]]; sd.batch_id = (int)d[1]; sd.id = i;
      sd.a = (float)d[2]; sd.b = (float)d[3]; sd.mu = (float)d[4]; sd.z_range = (float)d[5]; sd.sigma2 = (float)d[6];
      pt_seeds.push_back(sd);
    }
    for (int i = 0; i < nss; ++i) {
      std::vector<double> d = read_doubles(g, 15);
      mini::LineSeed sd; mini::LineFeat& L = lfs[(size_t)d[0]];
      L.px = mini::Vec2(d[10], d[11]); L.f = mini::Vec3(d[12], d[13], d[14]);       // Feature::px / f of the segment feature
      sd.ftr = &L; sd.batch_id = (int)d[1]; sd.id = i;
      sd.a = (float)d[2]; sd.b = (float)d[3]; sd.mu_s = (float)d[4]; sd.mu_e = (float)d[5]; sd.z_range_s = (float)d[6]; sd.z_range_e = (float)d[7];
      sd.sigma2_s = (float)d[8]; sd.sigma2_e = (float)d[9];
      seg_seeds.push_back(sd);
    }
    fclose(g);
    plsvo::depth_filter::SeedUpdateOptions opt;
    opt.n_pyr_levels = n_pyr_levels;
    FILE* oo = o;
    const bool ok = plsvo::depth_filter::updateSeeds(
        cur, pt_seeds, seg_seeds, batch_counter, opt,
        [oo](mini::PointSeed& sd, const double* xyz) { fprintf(oo, "pconv %d %.17g %.17g %.17g %.9g\n", sd.id, xyz[0], xyz[1], xyz[2], (double)sd.sigma2); },
        [oo](mini::LineSeed& sd, const double* xs, const double* xe) {
          fprintf(oo, "sconv %d %.17g %.17g %.17g %.17g %.17g %.17g\n", sd.id, xs[0], xs[1], xs[2], xe[0], xe[1], xe[2]); });
    if (!ok) { fprintf(stderr, "depth_filter::updateSeeds failed\n"); return 3; }
    for (auto& sd : pt_seeds) fprintf(o, "pseed %d %.9g %.9g %.9g %.9g\n", sd.id, (double)sd.a, (double)sd.b, (double)sd.mu, (double)sd.sigma2);
    for (auto& sd : seg_seeds) fprintf(o, "sseed %d %.9g %.9g %.9g %.9g %.9g %.9g\n", sd.id, (double)sd.a, (double)sd.b, (double)sd.mu_s, (double)sd.mu_e,
                                       (double)sd.sigma2_s, (double)sd.sigma2_e);
  }
  fclose(o);
  return 0;
}
