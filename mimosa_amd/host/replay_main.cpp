// Native sequence replay: drives mimosa_hip::replay::FixedLagReplay (host/mimosa_hip/replay.hpp) on an input file written
// by mimosa_amd/replay.py:write_native_input and prints one JSON object (estimated poses, per-stage seconds, scans/s).
//   replay_native <input.bin> [repeats] [manager | sequential | sharded <world> | sharded-rccl]
//     repeats > 1: the whole sequence again, timing of the last pass is reported
//     sharded <world>: the map sharded over <world> ranks INSIDE this process (one host thread and one context each, in-process
//       transport: what a one-GPU box can run); sharded-rccl: this process is one rank of a torch.distributed.run-style launch
//       (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT), RCCL over xGMI.  The JSON carries rank 0's trajectory and
//       `max_rank_deviation_m`, the largest difference between any rank's positions and rank 0's (in-process form).
#include <cstdio>
#include <cstring>

#include "mimosa_hip/binio.hpp"
#include "mimosa_hip/replay.hpp"
#include "mimosa_hip/sharded_replay.hpp"

using namespace mimosa_hip;
using binio::read_vec;

int main(int argc, char ** argv)
{
  if (argc < 2) {
    std::fprintf(stderr, "usage: replay_native <input.bin> [repeats] [manager | sequential]\n");
    return 2;
  }
  const int repeats = argc > 2 ? std::atoi(argv[2]) : 1;
  const bool through_manager = argc > 3 && std::string(argv[3]) == "manager";  // the same sequence through lidar::Manager::callback
  const bool sequential = argc > 3 && std::string(argv[3]) == "sequential";   // FixedLagReplay without the cross-scan overlap
  const bool sharded_local = argc > 4 && std::string(argv[3]) == "sharded";
  const bool sharded_rccl = argc > 3 && std::string(argv[3]) == "sharded-rccl";
  const int world = sharded_local ? std::atoi(argv[4]) : 1;
  try {
    std::ifstream f(argv[1], std::ios::binary);
    if (!f) throw std::runtime_error("cannot open the input file");
    const auto I = read_vec<int32_t>(f);   // window, update_iters, photometric, neighbour mode, lru_horizon
    const auto D = read_vec<double>(f);    // between sigmas (rot, trans), keyframe thresholds (trans, rot deg), gravity xyz
    const auto regb = read_vec<uint8_t>(f);
    const auto inpb = read_vec<uint8_t>(f);
    if (I.size() != 5 || D.size() != 7 || regb.size() != sizeof(mh_reg_config) || inpb.size() != sizeof(mh_input_config))
      throw std::runtime_error("configuration block has the wrong shape");
    replay::Config cfg;
    cfg.window = I[0];
    cfg.update_iters = I[1];
    cfg.photometric = I[2] != 0;
    cfg.neighbor_voxel_mode = static_cast<size_t>(I[3]);
    cfg.between_sigma_rot = D[0];
    cfg.between_sigma_trans = D[1];
    cfg.keyframe_trans_thresh = D[2];
    cfg.keyframe_rot_thresh_deg = D[3];
    cfg.gravity = {D[4], D[5], D[6]};
    std::memcpy(&cfg.reg, regb.data(), sizeof(cfg.reg));
    std::memcpy(&cfg.input, inpb.data(), sizeof(cfg.input));
    if (cfg.photometric) cfg.photo = binio::read_photo_config(f);
    cfg.pipeline = !sequential;
    const auto bias = read_vec<double>(f);
    for (size_t i = 0; i + 2 < bias.size(); i += 3) cfg.bias_directions.push_back(V3D(bias[i], bias[i + 1], bias[i + 2]));
    const auto seed = read_vec<float>(f);
    const auto s0 = read_vec<double>(f);  // R (9), t (3), velocity (3): the state at the first IMU sample of the first sweep
    const auto nsc = read_vec<int32_t>(f);
    std::vector<replay::ScanInput> scans(static_cast<size_t>(nsc.at(0)));
    for (auto & sc : scans) {
      sc.raw = read_vec<lidar::PointOuster>(f);
      sc.imu.ts = read_vec<double>(f);
      const auto gy = read_vec<double>(f), ac = read_vec<double>(f);
      for (size_t j = 0; j < sc.imu.ts.size(); ++j) {
        sc.imu.gyro.push_back({gy[3 * j], gy[3 * j + 1], gy[3 * j + 2]});
        sc.imu.acc.push_back({ac[3 * j], ac[3 * j + 1], ac[3 * j + 2]});
      }
      sc.header_ts = read_vec<double>(f).at(0);
    }
    replay::State st0;
    if (s0.size() < 15) throw std::runtime_error("the initial state needs 15 doubles");
    for (int i = 0; i < 9; ++i) st0.T.R[i] = s0[i];
    for (int i = 0; i < 3; ++i) st0.T.t[i] = s0[9 + i];
    st0.vel = {s0[12], s0[13], s0[14]};
    const char * lr = std::getenv("LOCAL_RANK");
    auto ctx = std::make_shared<lidar::Context>(sharded_rccl && lr ? std::atoi(lr) : 0);
    replay::Result r;
    double max_dev = 0.0;
    int n_ranks = 1;
    for (int rep = 0; rep < repeats; ++rep) {
      if (sharded_local) {
        // `world` ranks = host threads of this process, one context each on the one device
        if (world < 1 || world > 64) throw std::runtime_error("sharded: world in 1..64");
        n_ranks = world;
        std::vector<std::shared_ptr<lidar::Context>> ctxs{ctx};
        for (int q = 1; q < world; ++q) ctxs.push_back(std::make_shared<lidar::Context>(0));
        const auto comms = lidar::ShardCommunicator::local(ctxs);
        std::vector<replay::Result> rr(static_cast<size_t>(world));
        std::vector<std::string> err(static_cast<size_t>(world));
        std::vector<std::thread> th;
        for (int q = 0; q < world; ++q)
          th.emplace_back([&, q] {
            try {
              replay::ShardedFixedLagReplay run(comms[static_cast<size_t>(q)], cfg, static_cast<size_t>(I[4]), 3, world == 1);
              run.seedMap(seed.data(), seed.size() / 3);
              rr[static_cast<size_t>(q)] = run.run(scans, st0);
            } catch (const std::exception & e) {
              err[static_cast<size_t>(q)] = e.what();
            }
          });
        for (auto & t : th) t.join();
        for (int q = 0; q < world; ++q)
          if (!err[static_cast<size_t>(q)].empty()) throw std::runtime_error("rank " + std::to_string(q) + ": " + err[static_cast<size_t>(q)]);
        r = rr[0];
        max_dev = 0.0;
        for (int q = 1; q < world; ++q) {
          r.seconds = std::max(r.seconds, rr[static_cast<size_t>(q)].seconds);
          for (size_t k = 0; k < r.poses.size(); ++k)
            for (int i = 0; i < 3; ++i) max_dev = std::max(max_dev, std::fabs(rr[static_cast<size_t>(q)].poses[k].t[i] - r.poses[k].t[i]));
        }
      } else if (sharded_rccl) {
        const auto comm = lidar::ShardCommunicator::rcclFromEnv(ctx);
        n_ranks = comm->world();
        replay::ShardedFixedLagReplay run(comm, cfg, static_cast<size_t>(I[4]), 3, comm->world() == 1);
        run.seedMap(seed.data(), seed.size() / 3);
        r = run.run(scans, st0);
      } else if (through_manager) {
        replay::ManagerReplay run(ctx, cfg, static_cast<size_t>(I[4]));
        r = run.run(scans, st0, seed.data(), seed.size() / 3);
      } else {
        replay::FixedLagReplay run(ctx, cfg, static_cast<size_t>(I[4]));
        run.seedMap(seed.data(), seed.size() / 3);
        r = run.run(scans, st0);
      }
    }
    std::printf("{\"scans\": %zu, \"seconds\": %.9f, \"scans_per_s\": %.3f, \"n_keyframes\": %d, \"n_ranks\": %d, \"max_rank_deviation_m\": %.3e,\n", scans.size(),
                r.seconds, static_cast<double>(scans.size()) / r.seconds, r.n_keyframes, n_ranks, max_dev);
    std::printf("\"stage_s\": {\"front_end\": %.9f, \"imu\": %.9f, \"factor_create\": %.9f, \"optimise\": %.9f, \"update_map\": %.9f},\n",
                r.stage[0], r.stage[1], r.stage[2], r.stage[3], r.stage[4]);
    {
      const char * names[12] = {"stage_wait", "prepare_incl_stage_wait", "deskew", "geo_preprocess", "icp_create", "photo_wait", "photo_preprocess", "photo_factor",
                                "optimise", "keyframe_map", "photo_final_linearize", "photo_tail_incl_final_linearize"};
      std::printf("\"detail_s\": {");
      for (int i = 0; i < 12; ++i) std::printf("\"%s\": %.9f, ", names[i], r.detail[i]);
      std::printf("\"stager_start_latency\": %.9f, \"stager_prefetch\": %.9f, \"photo_worker_start_latency\": %.9f, \"photo_worker_update_map\": %.9f},\n", r.worker[0],
                  r.worker[1], r.worker[2], r.worker[3]);
    }
    std::printf("\"photo_valid\": [");
    for (size_t i = 0; i < r.photo_valid.size(); ++i) std::printf("%d%s", r.photo_valid[i], i + 1 < r.photo_valid.size() ? ", " : "");
    std::printf("],\n\"first_costs\": [");
    for (size_t i = 0; !r.costs.empty() && i < r.costs.at(0).size(); ++i) std::printf("%.17g%s", r.costs[0][i], i + 1 < r.costs[0].size() ? ", " : "");
    std::printf("],\n\"poses\": [");
    for (size_t k = 0; k < r.poses.size(); ++k) {
      std::printf("[");
      for (int i = 0; i < 9; ++i) std::printf("%.17g, ", r.poses[k].R[i]);
      std::printf("%.17g, %.17g, %.17g]%s", r.poses[k].t[0], r.poses[k].t[1], r.poses[k].t[2], k + 1 < r.poses.size() ? ", " : "");
    }
    std::printf("]}\n");
  } catch (const std::exception & e) {
    std::fprintf(stderr, "replay_native: %s\n", e.what());
    return 1;
  }
  return 0;
}
