// C++ host mirror of the map-sharded scan-to-map factor (BASELINE configs[2]) over the C ABI's mh_shard_*
// group (include/mimosa_hip.h; implementation mimosa_amd/csrc/shard_api.hip).
//
// The reference is single-process, so there is no class to mirror; what is kept is ICPFactor's surface
// (include/mimosa/lidar/geometric_factor.hpp:25-72, :119-174, :231): a gtsam::NonlinearFactor whose linearize(Values) returns
// the gtsam::HessianFactor of the WHOLE scan against the WHOLE map — identical on every rank — so the smoother of each
// process adds it to its graph exactly like an ICPFactor.  One process per GPU; the exchange (ncclAllToAll of fixed-size
// segments, ncclAllReduce of the Hessian sums over xGMI) happens inside the library, this header only owns handles.
//
//   ShardCommunicator   mh_shard_comm: RCCL (rendezvous of the 128-byte ncclUniqueId over a TCP socket on the node, or
//                       handed in by the caller) or the in-process test transport
//   ShardedVoxelMap     this rank's shard of IncrementalVoxelMapPCL: blocks it owns + their one-voxel halo
//   ShardedICPFactor    the factor
#pragma once

#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <chrono>
#include <cstdlib>
#include <thread>

#include "lidar.hpp"

namespace mimosa_hip
{
namespace lidar
{
class ShardCommunicator
{
public:
  using Ptr = std::shared_ptr<ShardCommunicator>;
  ~ShardCommunicator() { mh_shard_comm_destroy(comm_); }
  ShardCommunicator(const ShardCommunicator &) = delete;
  ShardCommunicator & operator=(const ShardCommunicator &) = delete;

  // RCCL communicator from an id the caller distributed itself (MPI, a file, torch.distributed ...)
  static Ptr rccl(const std::shared_ptr<Context> & ctx, const std::array<char, MH_SHARD_UNIQUE_ID_BYTES> & id, int world, int rank)
  {
    Ptr c(new ShardCommunicator(ctx));
    ctx->check(mh_shard_comm_init_rccl(ctx->get(), id.data(), world, rank, &c->comm_), "mh_shard_comm_init_rccl");
    return c;
  }
  // ... with the id passed from rank 0 to the others over a TCP socket (ranks of one node: host = 127.0.0.1).  Rank 0 listens
  // on `host`:`port` ONLY (not on every interface) until ranks 1 .. world - 1 have each fetched the id; a peer says who it is
  // first ("MHID" + its rank), anything else — a stray connection, a rank twice — is dropped without costing a real rank its
  // slot.
  static Ptr rcclOverTcp(const std::shared_ptr<Context> & ctx, int world, int rank, const std::string & host, int port, double timeout_s = 120.0)
  {
    std::array<char, MH_SHARD_UNIQUE_ID_BYTES> id{};
    if (rank == 0) {
      ctx->check(mh_shard_unique_id(id.data()), "mh_shard_unique_id");
      if (world > 1) serveId(id, world, host, port, timeout_s);
    } else {
      fetchId(id, rank, host, port, timeout_s);
    }
    return rccl(ctx, id, world, rank);
  }
  // RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as `python -m torch.distributed.run` (or any launcher) exports them; the id
  // travels on MASTER_PORT + 1
  static Ptr rcclFromEnv(const std::shared_ptr<Context> & ctx)
  {
    auto env = [](const char * k, const char * d) {
      const char * v = std::getenv(k);
      return std::string(v && *v ? v : d);
    };
    const int world = std::atoi(env("WORLD_SIZE", "1").c_str()), rank = std::atoi(env("RANK", "0").c_str());
    return rcclOverTcp(ctx, world, rank, env("MASTER_ADDR", "127.0.0.1"), std::atoi(env("MASTER_PORT", "29500").c_str()) + 1);
  }
  // `world` ranks inside this process (one host thread each, one device): what the tests use on a one-GPU box
  static std::vector<Ptr> local(const std::vector<std::shared_ptr<Context>> & ctxs)
  {
    std::vector<mh_shard_comm *> raw(ctxs.size(), nullptr);
    if (mh_shard_comm_init_local(static_cast<int>(ctxs.size()), raw.data()) != MH_OK)
      throw std::runtime_error(std::string("mh_shard_comm_init_local: ") + mh_last_error(nullptr));
    std::vector<Ptr> out;
    for (size_t r = 0; r < ctxs.size(); ++r) {
      Ptr c(new ShardCommunicator(ctxs[r]));
      c->comm_ = raw[r];
      out.push_back(c);
    }
    return out;
  }
  int world() const { return mh_shard_comm_world(comm_); }
  int rank() const { return mh_shard_comm_rank(comm_); }
  std::string backend() const { return mh_shard_comm_backend(comm_); }
  mh_shard_comm * underlying() const { return comm_; }
  const std::shared_ptr<Context> & context() const { return ctx_; }

private:
  explicit ShardCommunicator(const std::shared_ptr<Context> & ctx) : ctx_(ctx) {}
  static void serveId(const std::array<char, MH_SHARD_UNIQUE_ID_BYTES> & id, int world, const std::string & host, int port, double timeout_s)
  {
    const int ls = ::socket(AF_INET, SOCK_STREAM, 0);
    if (ls < 0) throw std::runtime_error("ShardCommunicator: socket() failed");
    int one = 1;
    ::setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_port = htons(static_cast<uint16_t>(port));
    if (::inet_pton(AF_INET, host.c_str(), &a.sin_addr) != 1) {
      ::close(ls);
      throw std::runtime_error("ShardCommunicator: MASTER_ADDR must be an IPv4 address");
    }
    if (::bind(ls, reinterpret_cast<sockaddr *>(&a), sizeof(a)) != 0 || ::listen(ls, world) != 0) {
      ::close(ls);
      throw std::runtime_error("ShardCommunicator: cannot listen on " + host + ":" + std::to_string(port));
    }
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<char> served(static_cast<size_t>(world), 0);
    int left = world - 1;
    while (left > 0) {
      const double remaining = timeout_s - std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (remaining <= 0) {
        ::close(ls);
        throw std::runtime_error("ShardCommunicator: a peer did not fetch the communicator id in time");
      }
      timeval tv{static_cast<time_t>(remaining), static_cast<suseconds_t>((remaining - static_cast<double>(static_cast<time_t>(remaining))) * 1e6)};
      ::setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
      const int s = ::accept(ls, nullptr, nullptr);
      if (s < 0) continue;  // timed out or interrupted: the loop head decides
      timeval hv{2, 0};     // a peer has two seconds to say who it is
      ::setsockopt(s, SOL_SOCKET, SO_RCVTIMEO, &hv, sizeof(hv));
      char hello[8] = {0};
      size_t got = 0;
      while (got < sizeof(hello)) {
        const ssize_t r = ::recv(s, hello + got, sizeof(hello) - got, 0);
        if (r <= 0) break;
        got += static_cast<size_t>(r);
      }
      int32_t peer = -1;
      if (got == sizeof(hello) && std::memcmp(hello, "MHID", 4) == 0) std::memcpy(&peer, hello + 4, 4);
      if (peer >= 1 && peer < world && !served[static_cast<size_t>(peer)]) {
        size_t off = 0;
        while (off < id.size()) {
          const ssize_t w = ::send(s, id.data() + off, id.size() - off, 0);
          if (w <= 0) break;
          off += static_cast<size_t>(w);
        }
        if (off == id.size()) {
          served[static_cast<size_t>(peer)] = 1;
          --left;
        }
      }
      ::close(s);
    }
    ::close(ls);
  }
  static void fetchId(std::array<char, MH_SHARD_UNIQUE_ID_BYTES> & id, int rank, const std::string & host, int port, double timeout_s)
  {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const int s = ::socket(AF_INET, SOCK_STREAM, 0);
      if (s < 0) throw std::runtime_error("ShardCommunicator: socket() failed");
      sockaddr_in a{};
      a.sin_family = AF_INET;
      a.sin_port = htons(static_cast<uint16_t>(port));
      if (::inet_pton(AF_INET, host.c_str(), &a.sin_addr) != 1) {
        ::close(s);
        throw std::runtime_error("ShardCommunicator: MASTER_ADDR must be an IPv4 address");
      }
      if (::connect(s, reinterpret_cast<sockaddr *>(&a), sizeof(a)) == 0) {
        char hello[8] = {'M', 'H', 'I', 'D', 0, 0, 0, 0};
        const int32_t me = rank;
        std::memcpy(hello + 4, &me, 4);
        (void)::send(s, hello, sizeof(hello), 0);
        size_t off = 0;
        while (off < id.size()) {
          const ssize_t r = ::recv(s, id.data() + off, id.size() - off, 0);
          if (r <= 0) break;
          off += static_cast<size_t>(r);
        }
        ::close(s);
        if (off == id.size()) return;
      } else {
        ::close(s);
      }
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
        throw std::runtime_error("ShardCommunicator: rank 0 did not serve the communicator id in time");
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
  }
  std::shared_ptr<Context> ctx_;
  mh_shard_comm * comm_ = nullptr;
};

// This rank's shard of the voxel map: IncrementalVoxelMapPCL::insert (incremental_voxel_map.cpp:19-24) of a batch that is
// IDENTICAL on every rank keeps the points of owned shard blocks plus their one-voxel halo, in the original order — every
// voxel a rank stores has the contents it has in the unsharded map.
class ShardedVoxelMap
{
public:
  using Ptr = std::shared_ptr<ShardedVoxelMap>;
  ShardedVoxelMap(const ShardCommunicator::Ptr & comm, const GeometricConfig & cfg, int block_log2 = 3) : comm_(comm), block_log2_(block_log2)
  {
    map_ = std::make_shared<IncrementalVoxelMapPCL>(comm->context(), cfg.scan_to_map.target_ivox_map_leaf_size);
    map_->set_lru_horizon(cfg.lru_horizon);  // geometric.cpp:23-28
    map_->set_neighbor_voxel_mode(cfg.neighbor_voxel_mode);
    map_->set_min_dist_in_cell(cfg.scan_to_map.target_ivox_map_min_dist_in_voxel);
  }
  void insert(const float * xyz, size_t n)
  {
    comm_->context()->check(mh_map_insert_shard(map_->underlying(), xyz, n, 3, comm_->world(), comm_->rank(), block_log2_), "mh_map_insert_shard");
  }
  void insert(const PointCloud & cloud)
  {
    if (!cloud.empty())
      comm_->context()->check(mh_map_insert_shard(map_->underlying(), &cloud[0].x, cloud.size(), sizeof(Point) / sizeof(float), comm_->world(), comm_->rank(), block_log2_),
                              "mh_map_insert_shard");
  }
  // Geometric::updateMap's insert (geometric.cpp:483-495) of a device-resident scan's Be_cloud_: f32 world transform, shard
  // filter and greedy insert on the GPU — this rank's part of IncrementalVoxelMapPCL::insertBodyCloud
  void insertBodyCloud(mh_scan * scan, const Pose3 & T_W_Be)
  {
    float Rt[12];
    toFloat12(T_W_Be, Rt);
    comm_->context()->check(mh_map_insert_shard_from_scan(map_->underlying(), scan, Rt, Rt + 9, comm_->world(), comm_->rank(), block_log2_), "mh_map_insert_shard_from_scan");
  }
  // Successor for copy-then-insert (Geometric::updateMap, geometric.cpp:494-495): a device-to-device copy of THIS rank's shard;
  // this object stays valid and unchanged for the factors that hold it.  Every rank forks at the same keyframes.
  Ptr fork() const
  {
    Ptr next(new ShardedVoxelMap(comm_, block_log2_));
    next->map_ = map_->fork();
    return next;
  }
  const IncrementalVoxelMapPCL::Ptr & map() const { return map_; }
  const ShardCommunicator::Ptr & communicator() const { return comm_; }
  int blockLog2() const { return block_log2_; }

private:
  ShardedVoxelMap(const ShardCommunicator::Ptr & comm, int block_log2) : comm_(comm), block_log2_(block_log2) {}
  ShardCommunicator::Ptr comm_;
  int block_log2_;
  IncrementalVoxelMapPCL::Ptr map_;
};

class ShardedICPFactor : public NonlinearFactor
{
public:
  using Ptr = std::shared_ptr<ShardedICPFactor>;
  // unary (geometric_factor.hpp:119-129): cloud_share = this rank's part of the scan (any split; the first linearize routes
  // every point to the owner of its centre voxel).  Collective: every rank constructs its factor at the same time.
  ShardedICPFactor(const Key key_source, const ShardedVoxelMap::Ptr & shard, const PointCloud & cloud_share, const RegistrationConfig & config,
                   bool force_collectives = false)
  : NonlinearFactor(KeyVector{key_source}), is_binary_(false), impl_(std::make_shared<Impl>(shard))
  {
    create(cloud_share, config, force_collectives);
  }
  // unary, the rank's share already on the device (a slice of a resident scan's sm_Be_cloud_ds_: mh_scan_device_points)
  ShardedICPFactor(const Key key_source, const ShardedVoxelMap::Ptr & shard, const mh_point32 * d_share, size_t n_share, const RegistrationConfig & config,
                   bool force_collectives = false)
  : NonlinearFactor(KeyVector{key_source}), is_binary_(false), impl_(std::make_shared<Impl>(shard))
  {
    create(d_share, n_share, true, config, force_collectives);
  }
  // binary (:131-142)
  ShardedICPFactor(const Key key_source, const Key key_target, const ShardedVoxelMap::Ptr & shard, const PointCloud & cloud_share,
                   const RegistrationConfig & config, bool force_collectives = false)
  : NonlinearFactor(KeyVector{key_source, key_target}), is_binary_(true), impl_(std::make_shared<Impl>(shard))
  {
    create(cloud_share, config, force_collectives);
  }

  // ISAM2 clones factors; a deep copy of a sharded factor would be a collective of its own, so the clone SHARES the device
  // state (and its data-association cache) with the original — one of the two is to be linearized from then on.
  NonlinearFactor::shared_ptr clone() const override { return std::shared_ptr<ShardedICPFactor>(new ShardedICPFactor(*this)); }
  size_t dim() const override { return 6; }                  // :166
  double error(const Values &) const override { return 0.0; }  // :168-174
  void computeComponents(bool on) { ctx().check(mh_shard_icp_set_components(impl_->icp, on ? 1 : 0), "mh_shard_icp_set_components"); }

  // :231-562 on the global cloud and map.  Collective: every rank calls it with the same Values.
  std::shared_ptr<GaussianFactor> linearize(const Values & c) const override
  {
    const PoseRM Ts = rowMajor(c.at<Pose3>(keys()[0]));
    PoseRM Tt{};
    if (is_binary_) Tt = rowMajor(c.at<Pose3>(keys()[1]));
    const A3 g = toArray(c.at<Unit3>(G(0)).unitVector());
    mh_icp_result r;
    ctx().check(mh_shard_icp_linearize(impl_->icp, Ts.R.data(), Ts.t.data(), is_binary_ ? Tt.R.data() : nullptr, is_binary_ ? Tt.t.data() : nullptr, g.data(), &r),
                "mh_shard_icp_linearize");
    impl_->last = r;
    return hessianFrom(keys(), is_binary_, r);
  }
  // The throughput forms (graph::Manager re-linearizes every live factor per update, src/graph/manager.cpp:585-588).
  // linearizeAsync only enqueues the factor's protocol round; wait() completes every round in flight on the communicator and
  // returns this factor's last enqueued result.  Collective like linearize(); keep one call in flight per factor where the
  // order of data-association cache updates matters (include/mimosa_hip.h).
  void linearizeAsync(const Values & c) const
  {
    const PoseRM Ts = rowMajor(c.at<Pose3>(keys()[0]));
    PoseRM Tt{};
    if (is_binary_) Tt = rowMajor(c.at<Pose3>(keys()[1]));
    const A3 g = toArray(c.at<Unit3>(G(0)).unitVector());
    ctx().check(mh_shard_icp_linearize_async(impl_->icp, Ts.R.data(), Ts.t.data(), is_binary_ ? Tt.R.data() : nullptr, is_binary_ ? Tt.t.data() : nullptr, g.data(),
                                             &impl_->last),
                "mh_shard_icp_linearize_async");
  }
  std::shared_ptr<GaussianFactor> wait() const
  {
    ctx().check(mh_shard_icp_wait(impl_->icp), "mh_shard_icp_wait");
    return hessianFrom(keys(), is_binary_, impl_->last);
  }
  // All factors of the window (one communicator) in ONE protocol round: one all-to-all carrying every factor's movers, one
  // K3b / K4b launch pair per kernel instantiation, one all-reduce of n x 168 doubles.  Unary and binary factors may be mixed.
  static std::vector<std::shared_ptr<GaussianFactor>> linearizeBatch(const std::vector<Ptr> & factors, const Values & c)
  {
    std::vector<std::shared_ptr<GaussianFactor>> out;
    if (factors.empty()) return out;
    const size_t n = factors.size();
    bool any_binary = false;
    for (const Ptr & f : factors) any_binary = any_binary || f->is_binary_;
    std::vector<mh_shard_icp *> h(n);
    std::vector<double> Rs(9 * n), ts(3 * n), Rt(any_binary ? 9 * n : 0), tt(any_binary ? 3 * n : 0), g(3 * n);
    for (size_t i = 0; i < n; ++i) {
      const ShardedICPFactor & f = *factors[i];
      h[i] = f.impl_->icp;
      const PoseRM Ts = rowMajor(c.at<Pose3>(f.keys()[0]));
      std::memcpy(&Rs[9 * i], Ts.R.data(), 72);
      std::memcpy(&ts[3 * i], Ts.t.data(), 24);
      if (f.is_binary_) {
        const PoseRM Tt = rowMajor(c.at<Pose3>(f.keys()[1]));
        std::memcpy(&Rt[9 * i], Tt.R.data(), 72);
        std::memcpy(&tt[3 * i], Tt.t.data(), 24);
      }
      const A3 gu = toArray(c.at<Unit3>(G(0)).unitVector());
      std::memcpy(&g[3 * i], gu.data(), 24);
    }
    std::vector<mh_icp_result> r(n);
    factors[0]->ctx().check(mh_shard_icp_linearize_batch(h.data(), n, Rs.data(), ts.data(), any_binary ? Rt.data() : nullptr, any_binary ? tt.data() : nullptr, g.data(), r.data()),
                            "mh_shard_icp_linearize_batch");
    for (size_t i = 0; i < n; ++i) {
      factors[i]->impl_->last = r[i];
      out.push_back(hessianFrom(factors[i]->keys(), factors[i]->is_binary_, r[i]));
    }
    return out;
  }
  void getLocalizabilities(V3D & trans_comp, V3D & rot_comp, V3D & trans_final, V3D & rot_final, M33 & eigenvectors_trans, M33 & eigenvectors_rot)
  {
    const mh_icp_result & l = impl_->last;
    trans_comp = vector3(l.loc_trans_comp);
    rot_comp = vector3(l.loc_rot_comp);
    trans_final = vector3(l.loc_trans_final);
    rot_final = vector3(l.loc_rot_final);
    eigenvectors_trans = matrix3(l.eigvec_trans);
    eigenvectors_rot = matrix3(l.eigvec_rot);
  }
  int getLinearizeCount() const { return impl_->last.linearize_count; }
  const mh_icp_result & lastResult() const { return impl_->last; }
  mh_shard_stats stats() const
  {
    mh_shard_stats s;
    ctx().check(mh_shard_icp_stats(impl_->icp, &s), "mh_shard_icp_stats");
    return s;
  }

private:
  struct Impl
  {
    explicit Impl(const ShardedVoxelMap::Ptr & s) : shard(s) { std::memset(&last, 0, sizeof(last)); }
    ~Impl() { mh_shard_icp_destroy(icp); }
    ShardedVoxelMap::Ptr shard;
    mh_shard_icp * icp = nullptr;
    mh_icp_result last;
  };
  ShardedICPFactor(const ShardedICPFactor & o) : NonlinearFactor(KeyVector(o.keys())), is_binary_(o.is_binary_), impl_(o.impl_) {}
  void create(const PointCloud & cloud, const RegistrationConfig & config, bool force) { create(cloud.data(), cloud.size(), false, config, force); }
  void create(const mh_point32 * points, size_t n, bool on_device, const RegistrationConfig & config, bool force)
  {
    mh_shard_config sc{};
    sc.block_log2 = impl_->shard->blockLog2();
    sc.force_collectives = force ? 1 : 0;
    ctx().check(mh_shard_icp_create(ctx().get(), impl_->shard->communicator()->underlying(), impl_->shard->map()->underlying(), points, n, on_device ? 1 : 0, &config,
                                    is_binary_ ? 1 : 0, &sc, &impl_->icp),
                "mh_shard_icp_create");
  }
  const Context & ctx() const { return *impl_->shard->communicator()->context(); }
  const bool is_binary_;
  std::shared_ptr<Impl> impl_;
};

}  // namespace lidar
}  // namespace mimosa_hip
