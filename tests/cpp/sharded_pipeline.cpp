// Drives the C++ host mirror of the map-sharded factor (mimosa_amd/host/mimosa_hip/sharded.hpp): ShardCommunicator,
// ShardedVoxelMap, ShardedICPFactor — the sharded factor through gtsam::Values / gtsam::HessianFactor next to the unsharded
// ICPFactor on the full map, over a pose sequence.  Modes:  local <world>  (ranks = threads, in-process transport),
// rccl  (one rank over RCCL, the full exchange protocol forced).  Prints JSON; inputs from tests/test_gpu_host_cpp.py.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <thread>

#include "../../mimosa_amd/host/mimosa_hip/sharded.hpp"

using namespace mimosa_hip;
using namespace mimosa_hip::lidar;

template <typename T>
static std::vector<T> read_vec(std::ifstream & f)
{
  uint64_t n = 0;
  f.read(reinterpret_cast<char *>(&n), 8);
  std::vector<T> v(n);
  f.read(reinterpret_cast<char *>(v.data()), static_cast<std::streamsize>(n * sizeof(T)));
  return v;
}
struct Flat
{
  double H[36], g[6], f;
};
static Flat flat(const GaussianFactor & gf)
{
  const HessianFactor & h = static_cast<const HessianFactor &>(gf);
  Flat o;
  const gtsam::Matrix G = h.information();
  const gtsam::Vector g = h.linearTerm();
  for (int r = 0; r < 6; ++r) {
    for (int c = 0; c < 6; ++c) o.H[6 * r + c] = G(r, c);
    o.g[r] = g(r);
  }
  o.f = h.constantTerm();
  return o;
}
static GeometricConfig enwide()
{
  GeometricConfig cfg;  // config/enwide/params.yaml:76-100
  cfg.lru_horizon = 1000;
  cfg.neighbor_voxel_mode = 19;
  cfg.scan_to_map.source_voxel_grid_min_dist_in_voxel = 0.15f;
  cfg.scan_to_map.target_ivox_map_min_dist_in_voxel = 0.15f;
  cfg.scan_to_map.max_corres_distance = 1.0f;
  cfg.scan_to_map.plane_validity_distance = 0.07f;
  cfg.scan_to_map.lidar_point_noise_std_dev = 0.07f;
  cfg.scan_to_map.project_on_degneneracy = 0;
  return cfg;
}

int main(int argc, char ** argv)
{
  if (argc < 3) return 2;
  const std::string mode = argv[2];
  const int world = mode == "local" && argc > 3 ? std::atoi(argv[3]) : 1;
  std::ifstream f(argv[1], std::ios::binary);
  const auto map_xyz = read_vec<float>(f);
  const auto scan = read_vec<Point>(f);
  const auto poses12 = read_vec<double>(f);  // n poses: R(9) t(3)
  const size_t n_poses = poses12.size() / 12;
  const GeometricConfig cfg = enwide();
  try {
    // the unsharded factor on the full map
    auto ctx0 = std::make_shared<Context>(0);
    auto full = std::make_shared<IncrementalVoxelMapPCL>(ctx0, cfg.scan_to_map.target_ivox_map_leaf_size);
    full->set_lru_horizon(cfg.lru_horizon);
    full->set_neighbor_voxel_mode(cfg.neighbor_voxel_mode);
    full->set_min_dist_in_cell(cfg.scan_to_map.target_ivox_map_min_dist_in_voxel);
    full->insert(map_xyz.data(), map_xyz.size() / 3);
    const Key X1 = X(1);
    ICPFactor ref(X1, full, scan, cfg.scan_to_map);
    std::vector<Flat> want(n_poses);
    std::vector<std::vector<int>> want_hist(n_poses, std::vector<int>(9));
    for (size_t k = 0; k < n_poses; ++k) {
      Values v;
      v.insert(G(0), Unit3(0.0, 0.0, -1.0));
      v.insert(X1, pose3(&poses12[12 * k], &poses12[12 * k + 9]));
      want[k] = flat(*ref.linearize(v));
      for (int i = 0; i < 9; ++i) want_hist[k][i] = ref.lastResult().status_hist[i];
    }
    // the throughput forms, in the order the ranks run them below: a window batch of the factor and a SECOND, fresh factor
    // (key X(2)) at the last pose, then one enqueued call of the first factor collected by wait()
    const Key X2 = X(2);
    Values v_last;
    v_last.insert(G(0), Unit3(0.0, 0.0, -1.0));
    v_last.insert(X1, pose3(&poses12[12 * (n_poses - 1)], &poses12[12 * (n_poses - 1) + 9]));
    v_last.insert(X2, pose3(&poses12[12 * (n_poses - 1)], &poses12[12 * (n_poses - 1) + 9]));
    ICPFactor ref2(X2, full, scan, cfg.scan_to_map);
    const Flat want_b1 = flat(*ref.linearize(v_last)), want_b2 = flat(*ref2.linearize(v_last)), want_async = flat(*ref.linearize(v_last));
    std::vector<Flat> got_b1(world), got_b2(world), got_async(world);
    // the sharded factor: `world` ranks
    std::vector<std::shared_ptr<Context>> ctxs;
    for (int r = 0; r < world; ++r) ctxs.push_back(r == 0 ? ctx0 : std::make_shared<Context>(0));
    std::vector<ShardCommunicator::Ptr> comms;
    if (mode == "rccl")
      comms.push_back(ShardCommunicator::rcclFromEnv(ctx0));
    else
      comms = ShardCommunicator::local(ctxs);
    std::vector<std::vector<Flat>> got(world, std::vector<Flat>(n_poses));
    std::vector<std::vector<int>> hist_ok(world, std::vector<int>(n_poses, 0));
    std::vector<mh_shard_stats> stats(world);
    std::vector<std::string> errors(world);
    auto body = [&](int r) {
      try {
        auto shard = std::make_shared<ShardedVoxelMap>(comms[r], cfg);
        shard->insert(map_xyz.data(), map_xyz.size() / 3);
        const size_t lo = scan.size() * r / world, hi = scan.size() * (r + 1) / world;
        PointCloud share(scan.begin() + lo, scan.begin() + hi);
        ShardedICPFactor fac(X1, shard, share, cfg.scan_to_map, mode == "rccl" || world == 1);
        for (size_t k = 0; k < n_poses; ++k) {
          Values v;
          v.insert(G(0), Unit3(0.0, 0.0, -1.0));
          v.insert(X1, pose3(&poses12[12 * k], &poses12[12 * k + 9]));
          got[r][k] = flat(*fac.linearize(v));
          hist_ok[r][k] = !std::memcmp(fac.lastResult().status_hist, want_hist[k].data(), 9 * sizeof(int));
        }
        stats[r] = fac.stats();
        auto fac_p = std::static_pointer_cast<ShardedICPFactor>(fac.clone());  // shares the device state with `fac`
        auto fac2 = std::make_shared<ShardedICPFactor>(X2, shard, share, cfg.scan_to_map, mode == "rccl" || world == 1);
        const auto both = ShardedICPFactor::linearizeBatch({fac_p, fac2}, v_last);
        got_b1[r] = flat(*both[0]);
        got_b2[r] = flat(*both[1]);
        fac.linearizeAsync(v_last);
        got_async[r] = flat(*fac.wait());
      } catch (const std::exception & e) {
        errors[r] = e.what();
      }
    };
    std::vector<std::thread> th;
    for (int r = 0; r < world; ++r) th.emplace_back(body, r);
    for (auto & t : th) t.join();
    for (int r = 0; r < world; ++r)
      if (!errors[r].empty()) throw std::runtime_error("rank " + std::to_string(r) + ": " + errors[r]);
    double worst = 0;
    int hist_all = 1, ranks_equal = 1;
    for (int r = 0; r < world; ++r)
      for (size_t k = 0; k < n_poses; ++k) {
        double num = 0, den = 0;
        for (int i = 0; i < 36; ++i) {
          num += (got[r][k].H[i] - want[k].H[i]) * (got[r][k].H[i] - want[k].H[i]);
          den += want[k].H[i] * want[k].H[i];
        }
        worst = std::max(worst, std::sqrt(num / den));
        worst = std::max(worst, std::fabs(got[r][k].f - want[k].f) / std::fabs(want[k].f));
        hist_all &= hist_ok[r][k];
        ranks_equal &= !std::memcmp(&got[r][k], &got[0][k], sizeof(Flat));  // every rank holds the SAME global result, bit for bit
      }
    auto rel_of = [](const Flat & a, const Flat & b) {
      double num = 0, den = 0;
      for (int i = 0; i < 36; ++i) {
        num += (a.H[i] - b.H[i]) * (a.H[i] - b.H[i]);
        den += b.H[i] * b.H[i];
      }
      return std::max(std::sqrt(num / den), std::fabs(a.f - b.f) / std::fabs(b.f));
    };
    double worst_batch = 0, worst_async = 0;
    for (int r = 0; r < world; ++r) {
      worst_batch = std::max(worst_batch, std::max(rel_of(got_b1[r], want_b1), rel_of(got_b2[r], want_b2)));
      worst_async = std::max(worst_async, rel_of(got_async[r], want_async));
    }
    std::printf("{\"batch_worst_rel\": %.3g, \"async_worst_rel\": %.3g, ", worst_batch, worst_async);
    std::printf("\"mode\": \"%s\", \"world\": %d, \"backend\": \"%s\", \"n_poses\": %zu, \"worst_rel\": %.3g, \"hist_equal\": %d, \"ranks_equal\": %d,\n", mode.c_str(),
                world, comms[0]->backend().c_str(), n_poses, worst, hist_all, ranks_equal);
    std::printf("\"collective\": %d, \"collectives_last\": %u, \"points_held\": [", stats[0].collective, stats[0].collectives_last);
    for (int r = 0; r < world; ++r) std::printf("%llu%s", static_cast<unsigned long long>(stats[r].n_live), r + 1 < world ? ", " : "");
    std::printf("], \"H00\": %.17g, \"f\": %.17g}\n", got[0][0].H[0], got[0][0].f);
  } catch (const std::exception & e) {
    std::fprintf(stderr, "sharded_pipeline: %s\n", e.what());
    return 1;
  }
  return 0;
}
