// Drives the C++ host mirror of the photometric classes (mimosa_amd/host/mimosa_hip/photometric.hpp) through the
// reference's call order
//   preprocess(frame 0) -> updateMap (detectFeatures) -> preprocess(frame 1) -> getFactors -> linearize -> clone ->
//   updateMap (bookkeeping from the statuses + re-detection)
// on inputs written by tests/test_gpu_host_cpp.py, and prints the results as JSON for comparison with the oracle.
#include <cstdio>
#include <fstream>
#include <iostream>

#include "../../mimosa_amd/host/mimosa_hip/binio.hpp"

using namespace mimosa_hip;
using namespace mimosa_hip::lidar;
using mimosa_hip::binio::pose_from;
using mimosa_hip::binio::read_vec;

static void dump(const char * name, const double * v, int n)
{
  std::printf("\"%s\": [", name);
  for (int i = 0; i < n; ++i) std::printf("%.17g%s", v[i], i + 1 < n ? ", " : "");
  std::printf("],\n");
}

int main(int argc, char ** argv)
{
  if (argc < 2) return 2;
  std::ifstream f(argv[1], std::ios::binary);
  const PhotometricConfig cfg = binio::read_photo_config(f);
  const auto bias = read_vec<double>(f);
  std::vector<V3D> bias_directions;
  for (size_t i = 0; i + 2 < bias.size(); i += 3) bias_directions.push_back(V3D(bias[i], bias[i + 1], bias[i + 2]));
  try {
    auto ctx = std::make_shared<Context>(0);
    Photometric photo(ctx, cfg);
    Values values;
    std::printf("{\n");
    for (int k = 0; k < 2; ++k) {
      auto raw = read_vec<Point>(f);
      auto desk = read_vec<Point>(f);
      const auto ns = read_vec<uint32_t>(f);
      const auto T = read_vec<double>(f);
      const auto pose = read_vec<double>(f);
      std::vector<std::pair<uint32_t, Pose3>> interp(ns.size());
      for (size_t g = 0; g < ns.size(); ++g) interp[g] = {ns[g], pose3(&T[12 * g], &T[12 * g + 9])};
      const Key Xk = X(10 + k);
      values.insert(Xk, pose3(pose.data(), pose.data() + 9));
      photo.preprocess(raw, desk, interp, 0.1 * k, Xk);
      double isum = 0;  // corrected intensities were written back into the deskewed cloud (photometric.cpp:307-314)
      for (const Point & p : desk) isum += p.intensity;
      std::printf("\"intensity_sum_%d\": %.17g,\n", k, isum);
      NonlinearFactorGraph graph;
      photo.getFactors(values, graph);  // frame 0: no features yet -> no factor (photometric.cpp:381)
      std::printf("\"n_factors_%d\": %zu,\n", k, graph.size());
      if (!graph.empty()) {
        auto h = std::static_pointer_cast<HessianFactor>(graph.at(0)->linearize(values));
        double H36[36], g6[6];
        auto flat = [](const HessianFactor & hf, double * Hm, double * gv) {
          const gtsam::Matrix G = hf.information();
          const gtsam::Vector g = hf.linearTerm();
          for (int r = 0; r < 6; ++r) {
            for (int c = 0; c < 6; ++c) Hm[6 * r + c] = G(r, c);
            gv[r] = g(r);
          }
        };
        flat(*h, H36, g6);
        dump("H", H36, 36);
        dump("g", g6, 6);
        std::printf("\"f\": %.17g,\n", h->constantTerm());
        std::printf("\"status_hist\": [");
        for (int i = 0; i < 9; ++i) std::printf("%d%s", photo.factor()->lastResult().status_hist[i], i < 8 ? ", " : "");
        std::printf("],\n");
        V3D tf, rf;
        M33 et, er;
        photo.factor()->getLocalizabilities(tf, rf, et, er);
        const double tf3[3] = {tf(0), tf(1), tf(2)};
        dump("loc_trans_final", tf3, 3);
        auto cl = graph.at(0)->clone();  // ISAM2 clones factors
        auto hc = std::static_pointer_cast<HessianFactor>(cl->linearize(values));
        double Hc[36], gc[6];
        flat(*hc, Hc, gc);
        std::printf("\"clone_equal\": %d,\n", (!std::memcmp(Hc, H36, sizeof(Hc)) && !std::memcmp(gc, g6, sizeof(gc)) && hc->constantTerm() == h->constantTerm()) ? 1 : 0);
        int valid = 0;
        for (const auto s : photo.factor()->getStatuses()) valid += s == PhotometricFactor::RejectStatus::Valid;
        std::printf("\"n_valid\": %d,\n", valid);
      }
      photo.updateMap(values, bias_directions);
      const auto feats = photo.features();
      std::printf("\"n_features_%d\": %zu,\n", k, feats.size());
      double csum = 0;
      for (const auto & ft : feats) csum += ft.center[0] + 1e-3 * ft.center[1] + ft.life_time;
      std::printf("\"feature_sum_%d\": %.17g,\n", k, csum);
    }
    std::printf("\"ok\": 1\n}\n");
  } catch (const std::exception & e) {
    std::fprintf(stderr, "photo_pipeline: %s\n", e.what());
    return 1;
  }
  return 0;
}
