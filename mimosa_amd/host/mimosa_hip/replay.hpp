// Sequence replay on the host mirror (SURVEY.md §8 "next" row f-4, BASELINE configs[4]): the native counterpart of
// mimosa_amd/replay.py — same loop, same arithmetic, no Python between the library calls.
//
// Per scan, in the reference's LiDAR call order (src/lidar/manager.cpp:45-147):
//   prepareInput -> IMU propagation over the distinct timestamps (manager.cpp:455-499) -> deskewPoints ->
//   Photometric::preprocess -> Geometric::preprocess -> ICPFactor / PhotometricFactor ctors -> smoother update with
//   EVERY live ICPFactor re-linearized per iteration (src/graph/manager.cpp:585-588; here ONE mh_icp_linearize_batch call)
//   -> Geometric::updateMap (keyframe test geometric.cpp:445-478, copy-then-insert on the device) + Photometric::updateMap.
// What stands in for GTSAM / ISAM2 (out of scope): a dense Gauss-Newton over the `window` most recent poses — unary ICP
// Hessian factors, the photometric factor on the newest pose, between factors from the IMU propagation, a prior on the
// oldest pose.  Retraction T <- T Exp(xi), xi = (omega, v): the perturbation the reference's Jacobians are taken against
// (geometric_factor.hpp:341-355, photometric_factor.hpp:262-279).
#pragma once

#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

#include "manager.hpp"

namespace mimosa_hip
{
namespace replay
{
using lidar::ICPFactor;
using lidar::IncrementalVoxelMapPCL;
using lidar::Photometric;
using lidar::PhotometricFactor;
using lidar::ScanFrontEnd;

// The harness's own arithmetic runs on plain row-major arrays (bit for bit what mimosa_amd/replay.py computes with numpy);
// gtsam::Pose3 / Values / HessianFactor appear where the mirror's classes are called, as they would in the reference.
using A36 = std::array<double, 36>;
struct RT
{
  A9 R{1, 0, 0, 0, 1, 0, 0, 0, 1};
  A3 t{0, 0, 0};
};
inline Pose3 toPose3(const RT & T) { return pose3(T.R.data(), T.t.data()); }
inline A9 so3Expmap(const A3 & w)  // gtsam::Rot3::Expmap (Rodrigues)
{
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = std::sqrt(th2);
  const double K[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double A, B;
  if (th < 1e-10) {
    A = 1.0 - th2 / 6.0;
    B = 0.5 - th2 / 24.0;
  } else {
    A = std::sin(th) / th;
    B = (1.0 - std::cos(th)) / th2;
  }
  A9 R{1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double kk = 0;
      for (int m = 0; m < 3; ++m) kk += K[3 * i + m] * K[3 * m + j];
      R[3 * i + j] += A * K[3 * i + j] + B * kk;
    }
  return R;
}


inline A9 matmul(const A9 & a, const A9 & b)
{
  A9 c{};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
  return c;
}
inline A9 transpose(const A9 & a) { return {a[0], a[3], a[6], a[1], a[4], a[7], a[2], a[5], a[8]}; }
inline A3 matvec(const A9 & a, const A3 & v)
{
  return {a[0] * v[0] + a[1] * v[1] + a[2] * v[2], a[3] * v[0] + a[4] * v[1] + a[5] * v[2], a[6] * v[0] + a[7] * v[1] + a[8] * v[2]};
}
inline A9 hat(const A3 & v) { return {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0}; }
inline A3 so3Log(const A9 & R)
{
  double c = (R[0] + R[4] + R[8] - 1.0) / 2.0;
  c = c < -1.0 ? -1.0 : (c > 1.0 ? 1.0 : c);
  const double th = std::acos(c);
  const double s = th < 1e-9 ? 0.5 : th / (2.0 * std::sin(th));
  return {(R[7] - R[5]) * s, (R[2] - R[6]) * s, (R[3] - R[1]) * s};
}
// T <- T * Exp(xi), first order in the translation
inline void retract(RT & T, const double * xi)
{
  const A3 d = matvec(T.R, {xi[3], xi[4], xi[5]});
  T.R = matmul(T.R, so3Expmap({xi[0], xi[1], xi[2]}));
  for (int i = 0; i < 3; ++i) T.t[i] += d[i];
}
inline RT between(const RT & a, const RT & b)
{
  RT r;
  const A9 Rt = transpose(a.R);
  r.R = matmul(Rt, b.R);
  r.t = matvec(Rt, {b.t[0] - a.t[0], b.t[1] - a.t[1], b.t[2] - a.t[2]});
  return r;
}
// Ad(R, t) in (rotation, translation) tangent order
inline A36 adjoint(const A9 & R, const A3 & t)
{
  A36 A{};
  const A9 hR = matmul(hat(t), R);
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      A[6 * i + j] = R[3 * i + j];
      A[6 * (3 + i) + 3 + j] = R[3 * i + j];
      A[6 * (3 + i) + j] = hR[3 * i + j];
    }
  return A;
}
// A x = b, Gaussian elimination with partial pivoting (what LAPACK's gesv does); A is n x n row-major, overwritten
inline std::vector<double> solve(std::vector<double> A, std::vector<double> b)
{
  const size_t n = b.size();
  for (size_t c = 0; c < n; ++c) {
    size_t piv = c;
    for (size_t r = c + 1; r < n; ++r)
      if (std::fabs(A[r * n + c]) > std::fabs(A[piv * n + c])) piv = r;
    if (A[piv * n + c] == 0.0) throw std::runtime_error("replay::solve: singular system");
    if (piv != c) {
      for (size_t j = 0; j < n; ++j) std::swap(A[c * n + j], A[piv * n + j]);
      std::swap(b[c], b[piv]);
    }
    for (size_t r = c + 1; r < n; ++r) {
      const double f = A[r * n + c] / A[c * n + c];
      if (f == 0.0) continue;
      for (size_t j = c; j < n; ++j) A[r * n + j] -= f * A[c * n + j];
      b[r] -= f * b[c];
    }
  }
  for (size_t i = n; i-- > 0;) {
    double s = b[i];
    for (size_t j = i + 1; j < n; ++j) s -= A[i * n + j] * b[j];
    b[i] = s / A[i * n + i];
  }
  return b;
}

struct ImuSamples
{
  std::vector<double> ts;
  std::vector<A3> gyro, acc;
};
struct State
{
  RT T;
  A3 vel{0, 0, 0};
};

// Manager::deskewPoints' host part (src/lidar/manager.cpp:455-499): states at the IMU sample times by integrating sample
// to sample, then constant-acc / omega extrapolation to every distinct timestamp.  `s0` = state at the first sample.
// Returns T_W_Bt per timestamp; `last` = state at the last sample.
inline std::vector<RT> propagate(const State & s0, const ImuSamples & imu, const double header_ts, const std::vector<uint32_t> & unique_ns,
                                    const A3 & gravity, State & last)
{
  const size_t m = imu.ts.size();
  if (m < 2) throw std::runtime_error("Preintegration not possible as there are less than 2 measurements P1");  // :442-446
  std::vector<State> st(m);
  st[0] = s0;
  for (size_t c = 0; c + 1 < m; ++c) {
    const double d = imu.ts[c + 1] - imu.ts[c];
    const A3 Ra = matvec(st[c].T.R, imu.acc[c]);
    const A3 aw{Ra[0] + gravity[0], Ra[1] + gravity[1], Ra[2] + gravity[2]};
    st[c + 1].T.R = matmul(st[c].T.R, so3Expmap({imu.gyro[c][0] * d, imu.gyro[c][1] * d, imu.gyro[c][2] * d}));
    for (int i = 0; i < 3; ++i) {
      st[c + 1].T.t[i] = st[c].T.t[i] + st[c].vel[i] * d + 0.5 * aw[i] * d * d;
      st[c + 1].vel[i] = st[c].vel[i] + aw[i] * d;
    }
  }
  std::vector<RT> out(unique_ns.size());
  size_t c = 0;
  for (size_t u = 0; u < unique_ns.size(); ++u) {
    const double tq = header_ts + unique_ns[u] * 1.0e-9;
    while (c + 2 < m && imu.ts[c + 1] < tq) ++c;  // interval with ts[c] < tq <= ts[c + 1] (:470-476)
    const double d = tq - imu.ts[c];
    const A9 E = so3Expmap({imu.gyro[c][0] * d, imu.gyro[c][1] * d, imu.gyro[c][2] * d});
    out[u].R = matmul(st[c].T.R, E);
    const A3 Ra = matvec(st[c].T.R, imu.acc[c]);
    for (int i = 0; i < 3; ++i) out[u].t[i] = st[c].T.t[i] + st[c].vel[i] * d + 0.5 * Ra[i] * d * d + 0.5 * gravity[i] * d * d;
  }
  last = st[m - 1];
  return out;
}

struct Config
{
  int window = 5;
  int update_iters = 6;
  double between_sigma_rot = 2e-3, between_sigma_trans = 1e-2;
  double keyframe_trans_thresh = 1.0, keyframe_rot_thresh_deg = 20.0;
  bool photometric = true;
  // FixedLagReplay: overlap what does not depend on each other across scans — the next cloud's staging (pinned copy +
  // upload on a copy stream) and the photometric map update of scan k (feature detection: a host-side selection over
  // device-compacted candidates, on the photometric context's own stream) run on two worker threads beside the geometric
  // path of scan k + 1.  Same calls on the same handles in the same order per handle: the trajectory does not change by a bit.
  bool pipeline = true;
  A3 gravity{0.0, 0.0, -9.81};
  lidar::RegistrationConfig reg = lidar::defaultRegistrationConfig();
  lidar::ManagerInputConfig input = lidar::defaultManagerInputConfig();
  size_t neighbor_voxel_mode = 19;
  lidar::PhotometricConfig photo;
  std::vector<V3D> bias_directions;
};

struct ScanInput
{
  std::vector<lidar::PointOuster> raw;
  ImuSamples imu;
  double header_ts = 0;
};

struct Result
{
  std::vector<RT> poses;
  std::vector<int> photo_valid;
  std::vector<std::vector<double>> costs;
  int n_keyframes = 0;
  double seconds = 0, stage[5] = {0, 0, 0, 0, 0};  // front_end, imu, factor_create, optimise, update_map
  // finer split of the main thread's time: stage_wait, prepare, deskew, geo_preprocess, icp_create, photo_wait, photo_preprocess,
  // photo_factor, optimise, keyframe_map, photo_final_linearize, photo_update_or_submit
  double detail[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  double worker[4] = {0, 0, 0, 0};  // stager: start latency, duration; photometric worker: start latency, duration (summed over the scans)
};

// One host thread that runs the jobs it is given in order (the pipelined replay's helpers).
class Worker
{
public:
  Worker() : th_([this] { loop(); }) {}
  ~Worker()
  {
    {
      std::lock_guard<std::mutex> g(mu_);
      stop_ = true;
    }
    cv_.notify_all();
    th_.join();
  }
  void submit(std::function<void()> job)
  {
    {
      std::lock_guard<std::mutex> g(mu_);
      jobs_.push_back(std::move(job));
      ++pending_;
    }
    cv_.notify_all();
  }
  // blocks until every submitted job has run; rethrows the first exception a job ended with
  void wait()
  {
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this] { return pending_ == 0; });
    if (error_) {
      std::exception_ptr e = error_;
      error_ = nullptr;
      std::rethrow_exception(e);
    }
  }

private:
  void loop()
  {
    for (;;) {
      std::function<void()> job;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
        if (jobs_.empty()) return;
        job = std::move(jobs_.front());
        jobs_.pop_front();
      }
      try {
        job();
      } catch (...) {
        std::lock_guard<std::mutex> g(mu_);
        if (!error_) error_ = std::current_exception();
      }
      {
        std::lock_guard<std::mutex> g(mu_);
        --pending_;
      }
      done_cv_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  std::deque<std::function<void()>> jobs_;
  int pending_ = 0;
  bool stop_ = false;
  std::exception_ptr error_;
  std::thread th_;
};

// The stand-in for GTSAM / ISAM2 (out of scope): a dense Gauss-Newton over the `window` most recent poses — unary ICP Hessian
// factors (every live one re-linearized per iteration through ICPFactor::linearizeBatch, what smoother_->update +
// additional_update_iterations do, src/graph/manager.cpp:585-588), the photometric factor on the newest pose, between
// factors from the IMU propagation, a prior on the oldest pose.  Shared by FixedLagReplay and by the graph-manager stand-in
// that lidar::Manager is driven with (ManagerReplay below).
// FactorT: lidar::ICPFactor, or lidar::ShardedICPFactor (sharded.hpp) — the same surface: a static linearizeBatch over the
// window's factors returning one HessianFactor each, identical on every rank in the sharded case.
template <class FactorT>
class WindowSmootherT
{
public:
  struct Live
  {
    size_t k;
    RT T;
    typename FactorT::Ptr f;
    bool has_Z;
    RT Z;
  };
  WindowSmootherT(int window, int update_iters, double between_sigma_rot, double between_sigma_trans)
  : window_(window), update_iters_(update_iters)
  {
    const double wr = 1.0 / (between_sigma_rot * between_sigma_rot), wt = 1.0 / (between_sigma_trans * between_sigma_trans);
    for (int i = 0; i < 3; ++i) {
      Wb_[i] = wr;
      Wb_[3 + i] = wt;
    }
  }
  void push(const Live & lv)
  {
    if (!pushed_) {
      first_k_ = lv.k;  // the first pose the smoother ever sees: its prior stays loose until the window slides past it
      pushed_ = true;
    }
    win.push_back(lv);
    if (static_cast<int>(win.size()) > window_) win.pop_front();
  }
  const RT & newest() const { return win.back().T; }
  void clear() { win.clear(); }
  // update_iters Gauss-Newton iterations at scan k; returns the cost before each iteration
  std::vector<double> optimise(const size_t k, const PhotometricFactor::Ptr & pf)
  {
    const double * Wb = Wb_;
    const struct
    {
      int window, update_iters;
    } cfg_{window_, update_iters_};
      const size_t nW = win.size(), dim = 6 * nW;
      std::vector<double> fs;
      for (int it = 0; it < cfg_.update_iters; ++it) {
        std::vector<typename FactorT::Ptr> factors(nW);
        Values v;
        v.insert(G(0), Unit3(0.0, 0.0, -1.0));  // the gravity direction ICPFactor::linearize reads (geometric_factor.hpp:257)
        for (size_t i = 0; i < nW; ++i) {
          factors[i] = win[i].f;
          v.insert(X(win[i].k), toPose3(win[i].T));
        }
        if (pf) pf->linearizeAsync(v);  // queued ahead of the window: one wait for both
        const auto lin = FactorT::linearizeBatch(factors, v);
        std::vector<double> A(dim * dim, 0.0), g(dim, 0.0);
        double cost = 0.0;
        for (size_t i = 0; i < nW; ++i) {
          const auto & h = *std::static_pointer_cast<HessianFactor>(lin[i]);
          const gtsam::Matrix Gi = h.information();
          const gtsam::Vector gi = h.linearTerm();
          for (int r = 0; r < 6; ++r) {
            for (int c = 0; c < 6; ++c) A[(6 * i + r) * dim + 6 * i + c] += Gi(r, c);
            g[6 * i + r] += -gi(r);  // the HessianFactor carries -b
          }
          cost += h.constantTerm();
        }
        if (pf) {
          const auto hp = std::static_pointer_cast<HessianFactor>(pf->collect());
          bool finite = pf->lastResult().status_hist[8] > 0;
          const gtsam::Matrix Gp = hp->information();
          const gtsam::Vector gp = hp->linearTerm();
          for (int q = 0; q < 36 && finite; ++q) finite = std::isfinite(Gp(q / 6, q % 6));
          for (int q = 0; q < 6 && finite; ++q) finite = std::isfinite(gp(q));
          if (finite) {
            const size_t o = 6 * (nW - 1);
            for (int r = 0; r < 6; ++r) {
              for (int c = 0; c < 6; ++c) A[(o + r) * dim + o + c] += Gp(r, c);
              g[o + r] += -gp(r);
            }
            cost += hp->constantTerm();
          }
        }
        for (size_t i = 1; i < nW; ++i) {
          if (!win[i].has_Z) continue;
          const RT ab = between(win[i - 1].T, win[i].T);
          const A9 Rzt = transpose(win[i].Z.R);
          const A9 Re = matmul(Rzt, ab.R);
          const A3 te = matvec(Rzt, {ab.t[0] - win[i].Z.t[0], ab.t[1] - win[i].Z.t[1], ab.t[2] - win[i].Z.t[2]});  // Z^-1 * between
          const A3 lr = so3Log(Re);
          const double r[6] = {lr[0], lr[1], lr[2], te[0], te[1], te[2]};
          const A9 Rabt = transpose(ab.R);
          const A3 tinv = matvec(Rabt, {-ab.t[0], -ab.t[1], -ab.t[2]});
          const A36 Ad = adjoint(Rabt, tinv);  // J_a = -Ad(between^-1), J_b = I
          // A += J^T W J, g += J^T W r over the two 6-blocks (a = i - 1, b = i)
          const size_t oa = 6 * (i - 1), ob = 6 * i;
          for (int p = 0; p < 6; ++p)
            for (int q = 0; q < 6; ++q) {
              double aa = 0;
              for (int m = 0; m < 6; ++m) aa += Ad[6 * m + p] * Wb[m] * Ad[6 * m + q];
              A[(oa + p) * dim + oa + q] += aa;
              A[(oa + p) * dim + ob + q] += -Ad[6 * q + p] * Wb[q];
              A[(ob + p) * dim + oa + q] += -Wb[p] * Ad[6 * p + q];
            }
          for (int p = 0; p < 6; ++p) {
            A[(ob + p) * dim + ob + p] += Wb[p];
            g[ob + p] += Wb[p] * r[p];
            double ga = 0;
            for (int m = 0; m < 6; ++m) ga += -Ad[6 * m + p] * Wb[m] * r[m];
            g[oa + p] += ga;
            cost += r[p] * Wb[p] * r[p];
          }
        }
        // what marginalisation leaves on the oldest pose; loose while that pose has never been optimised
        const bool loose = win[0].k == first_k_ && static_cast<int>(k - first_k_) < cfg_.window;
        const double sr = loose ? 0.017453292519943295 : 1e-4, st = loose ? 0.1 : 1e-4;
        for (int p = 0; p < 3; ++p) {
          A[p * dim + p] += 1.0 / (sr * sr);
          A[(3 + p) * dim + 3 + p] += 1.0 / (st * st);
        }
        for (size_t p = 0; p < dim; ++p) {
          A[p * dim + p] += 1e-9;
          g[p] = -g[p];
        }
        const std::vector<double> xi = solve(std::move(A), std::move(g));
        for (size_t i = 0; i < nW; ++i) retract(win[i].T, &xi[6 * i]);
        fs.push_back(cost);
      }
    return fs;
  }
  std::deque<Live> win;

private:
  int window_, update_iters_;
  double Wb_[6];
  size_t first_k_ = 0;
  bool pushed_ = false;
};

using WindowSmoother = WindowSmootherT<ICPFactor>;

// The geometric side of the replay loop — what Geometric owns in the reference (src/lidar/geometric.cpp): the map, the factor
// of a scan (getFactors, :185-226) and the keyframe's copy-then-insert (updateMap, :483-495).  PlainGeometric: one GPU holds
// the map.  ShardedGeometric (sharded_replay.hpp): the map sharded over the ranks of a communicator.
class PlainGeometric
{
public:
  using Factor = ICPFactor;
  PlainGeometric(const std::shared_ptr<lidar::Context> & ctx, const Config & cfg, size_t lru_horizon)
  {
    map_ = std::make_shared<IncrementalVoxelMapPCL>(ctx, cfg.reg.target_ivox_map_leaf_size);
    map_->set_lru_horizon(lru_horizon);
    map_->set_neighbor_voxel_mode(cfg.neighbor_voxel_mode);
    map_->set_min_dist_in_cell(cfg.reg.target_ivox_map_min_dist_in_voxel);
  }
  void seed(const float * xyz, size_t n) { map_->insert(xyz, n); }
  Factor::Ptr makeFactor(const Key Xk, ScanFrontEnd & scan, const lidar::RegistrationConfig & reg) { return std::make_shared<ICPFactor>(Xk, map_, scan, reg); }
  void keyframe(ScanFrontEnd & scan, const Pose3 & T_W_Be)
  {
    map_ = map_->fork();  // copy-then-insert (geometric.cpp:494-495): live factors keep the map they were built on
    map_->insertBodyCloud(scan.underlying(), T_W_Be);
  }

private:
  IncrementalVoxelMapPCL::Ptr map_;
};

template <class Geo>
class FixedLagReplayT
{
public:
  // geo: the geometric back end (its map lives on ctx)
  FixedLagReplayT(const std::shared_ptr<lidar::Context> & ctx, const Config & cfg, std::unique_ptr<Geo> geo) : ctx_(ctx), cfg_(cfg), scan_(ctx), scan_b_(ctx), geo_(std::move(geo))
  {
    if (cfg_.photometric) {
      // its own context = its own HIP stream: the patch factor (60 waves, a 27 us latency chain) and the window's ICP batch
      // (a few hundred waves) are independent work of one smoother iteration and run side by side instead of one behind
      // the other; every hand-over between the two contexts happens at a call that synchronises anyway
      photo_ctx_ = std::make_shared<lidar::Context>(ctx_->device());
      photo_.reset(new Photometric(photo_ctx_, cfg_.photo));
      scan_.keepRaw(true);
      scan_b_.keepRaw(true);
    }
  }
  void seedMap(const float * xyz, size_t n) { geo_->seed(xyz, n); }

  // state0: the state at the first IMU sample of the first sweep (the caller's first guess)
  Result run(const std::vector<ScanInput> & scans, const State & state0)
  {
    using clk = std::chrono::steady_clock;
    auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    Result res;
    using Smoother = WindowSmootherT<typename Geo::Factor>;
    using Live = typename Smoother::Live;
    Smoother smoother(cfg_.window, cfg_.update_iters, cfg_.between_sigma_rot, cfg_.between_sigma_trans);
    std::deque<Live> & win = smoother.win;
    std::vector<RT> kf_poses;
    State prev = state0;
    bool have_prev = false;
    const float I3f[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z3f[3] = {0, 0, 0};
    // Pipelined (cfg_.pipeline): `stager` copies cloud k + 1 into the idle front end's pinned buffer and starts its upload while
    // scan k is processed; `photo_worker` runs Photometric::updateMap of scan k (feature detection on the photometric context's
    // stream) beside the geometric path of scan k + 1, which joins it right before it needs the photometric frame itself.
    ScanFrontEnd * bufs[2] = {&scan_, &scan_b_};
    std::unique_ptr<Worker> stager, photo_worker;
    if (cfg_.pipeline) {
      stager.reset(new Worker);
      if (photo_) photo_worker.reset(new Worker);
      if (!scans.empty()) stager->submit([&scans, bufs] { bufs[0]->prefetch(scans[0].raw.data(), scans[0].raw.size()); });
    }
    const auto t_begin = clk::now();
    for (size_t k = 0; k < scans.size(); ++k) {
      const ScanInput & sc = scans[k];
      ScanFrontEnd & scan_ = cfg_.pipeline ? *bufs[k & 1] : this->scan_;
      const auto a0 = clk::now();
      if (cfg_.pipeline) {
        stager->wait();  // cloud k sits in this front end's staging buffer, its upload is on the copy stream
        res.detail[0] += secs(a0, clk::now());
        scan_.prepareInputPrefetched(cfg_.input, sc.header_ts);
        if (k + 1 < scans.size()) {
          const ScanInput * nx = &scans[k + 1];
          ScanFrontEnd * nb = bufs[(k + 1) & 1];
          const auto ts = clk::now();
          double * wk = res.worker;
          stager->submit([nx, nb, ts, wk] {
            const auto t0 = clk::now();
            nb->prefetch(nx->raw.data(), nx->raw.size());
            wk[0] += std::chrono::duration<double>(t0 - ts).count();
            wk[1] += std::chrono::duration<double>(clk::now() - t0).count();
          });
        }
      } else {
        scan_.prepareInput(sc.raw.data(), sc.raw.size(), cfg_.input, sc.header_ts);
      }
      const auto a1 = clk::now();
      State pred;
      const std::vector<RT> T_W_Bt = propagate(prev, sc.imu, sc.header_ts, scan_.uniqueNs(), cfg_.gravity, pred);
      // pose of every column in the scan-end frame, laid out as the C ABI takes poses (R row-major, t): the deskew and the
      // photometric frame read the same 12 doubles a Pose3 would hand back
      std::vector<double> T_Le_Lt12(12 * T_W_Bt.size());
      {
        const A9 Rt = transpose(pred.T.R);
        for (size_t g = 0; g < T_W_Bt.size(); ++g) {
          const A9 R = matmul(Rt, T_W_Bt[g].R);
          const A3 t = matvec(Rt, {T_W_Bt[g].t[0] - pred.T.t[0], T_W_Bt[g].t[1] - pred.T.t[1], T_W_Bt[g].t[2] - pred.T.t[2]});
          std::memcpy(&T_Le_Lt12[12 * g], R.data(), 72);
          std::memcpy(&T_Le_Lt12[12 * g + 9], t.data(), 24);
        }
      }
      const auto a2 = clk::now();
      scan_.deskewPoints(T_Le_Lt12.data(), T_W_Bt.size());
      const auto b0 = clk::now();
      res.detail[2] += secs(a2, b0);
      const Key Xk = X(k);
      // Pipelined: the frame of scan k is ENQUEUED here (behind the deskew, on the photometric stream; the scan is only read)
      // and runs beside the down-sampler and the ICP factor below, while the worker may still be inside updateMap of scan k - 1
      // (it reads the frame and the tracked features of k - 1; building touches neither); it becomes current — and the scan's
      // cloud receives the corrected intensities — at the commit, once that update has returned.
      if (photo_ && !cfg_.pipeline) {
        std::vector<Pose3> T_Le_Lt(T_W_Bt.size());
        for (size_t g = 0; g < T_W_Bt.size(); ++g) T_Le_Lt[g] = pose3(&T_Le_Lt12[12 * g], &T_Le_Lt12[12 * g + 9]);
        photo_->preprocess(scan_, T_Le_Lt, sc.header_ts, Xk);
      }
      if (photo_ && cfg_.pipeline) {
        const auto c0 = clk::now();
        photo_->preprocessBegin(scan_, T_Le_Lt12.data(), T_W_Bt.size());
        res.detail[6] += secs(c0, clk::now());
      }
      const auto b0g = clk::now();
      ctx_->check(mh_scan_preprocess_geometric(scan_.underlying(), I3f, z3f, cfg_.reg.source_voxel_grid_filter_leaf_size, 20,
                                               cfg_.reg.source_voxel_grid_min_dist_in_voxel, &scan_.mutableInfo()),
                  "mh_scan_preprocess_geometric");
      const auto a3 = clk::now();
      res.detail[3] += secs(b0g, a3);
      Live lv;
      lv.k = k;
      lv.T = pred.T;
      lv.f = geo_->makeFactor(Xk, scan_, cfg_.reg);
      lv.f->computeComponents(false);  // the loop below only takes H, b, f
      const auto b1 = clk::now();
      res.detail[4] += secs(a3, b1);
      lv.has_Z = have_prev;
      if (have_prev) lv.Z = between(prev.T, pred.T);
      Values values;
      NonlinearFactorGraph photo_graph;
      if (photo_) {
        if (cfg_.pipeline) {
          const auto c1 = clk::now();
          photo_worker->wait();  // Photometric::updateMap of scan k - 1
          photo_->preprocessCommit(sc.header_ts, Xk);
          photo_->detectPrefetch();  // candidate pixels of frame k: on the device while the smoother iterates
          res.detail[5] += secs(c1, clk::now());
        }
        const auto c2 = clk::now();
        values.insert(Xk, toPose3(pred.T));
        photo_->getFactors(values, photo_graph);  // no factor while nothing is tracked (photometric.cpp:381)
        res.detail[7] += secs(c2, clk::now());
      }
      PhotometricFactor::Ptr pf = photo_ ? photo_->factor() : nullptr;
      smoother.push(lv);
      const auto a4 = clk::now();
      const std::vector<double> fs = smoother.optimise(k, pf);
      res.costs.push_back(fs);
      const auto a5 = clk::now();
      const RT T = win.back().T;
      // ---- Geometric::updateMap's keyframe test (geometric.cpp:445-478)
      bool is_kf = true;
      if (!kf_poses.empty()) {
        size_t j = 0;
        double best = std::numeric_limits<double>::max();
        for (size_t i = 0; i < kf_poses.size(); ++i) {
          const double dx = T.t[0] - kf_poses[i].t[0], dy = T.t[1] - kf_poses[i].t[1], dz = T.t[2] - kf_poses[i].t[2];
          const double d = std::sqrt(dx * dx + dy * dy + dz * dz);
          if (d < best) {
            best = d;
            j = i;
          }
        }
        const A9 d = matmul(transpose(kf_poses[j].R), T.R);
        const double yaw = std::atan2(d[3], d[0]), pitch = std::atan2(-d[6], std::hypot(d[7], d[8])), roll = std::atan2(d[7], d[8]);
        const double ypr = std::max(std::fabs(yaw), std::max(std::fabs(pitch), std::fabs(roll)));
        is_kf = best > cfg_.keyframe_trans_thresh || ypr > cfg_.keyframe_rot_thresh_deg * 0.017453293;
      }
      if (is_kf) {
        geo_->keyframe(scan_, toPose3(T));
        kf_poses.push_back(T);
        ++res.n_keyframes;
      }
      const auto d0 = clk::now();
      res.detail[9] += secs(a5, d0);
      if (photo_) {
        values.update(Xk, toPose3(T));
        if (pf) {
          (void)pf->linearize(values);  // statuses / centres at the final pose feed the bookkeeping
          res.photo_valid.push_back(pf->lastResult().status_hist[8]);
        }
        const auto d1 = clk::now();
        res.detail[10] += secs(d0, d1);
        if (cfg_.pipeline) {
          Photometric * ph = photo_.get();
          const std::vector<V3D> * bias = &cfg_.bias_directions;
          const auto ts = clk::now();
          double * wk = res.worker;
          photo_worker->submit([ph, values, bias, ts, wk] {
            const auto t0 = clk::now();
            ph->updateMap(values, *bias);
            wk[2] += std::chrono::duration<double>(t0 - ts).count();
            wk[3] += std::chrono::duration<double>(clk::now() - t0).count();
          });
        } else {
          photo_->updateMap(values, cfg_.bias_directions);
        }
      }
      const auto a6 = clk::now();
      res.detail[1] += secs(a0, a1);
      res.detail[8] += secs(a4, a5);
      res.detail[11] += secs(d0, a6);
      res.stage[0] += secs(a0, a1) + secs(a2, a3);
      res.stage[1] += secs(a1, a2);
      res.stage[2] += secs(a3, a4);
      res.stage[3] += secs(a4, a5);
      res.stage[4] += secs(a5, a6);
      res.poses.push_back(T);
      // velocity: the propagated one, carried into the corrected attitude
      prev.T = T;
      prev.vel = matvec(T.R, matvec(transpose(pred.T.R), pred.vel));
      have_prev = true;
    }
    if (photo_worker) photo_worker->wait();
    win.clear();
    res.seconds = secs(t_begin, clk::now());
    return res;
  }

private:
  std::shared_ptr<lidar::Context> ctx_;
  Config cfg_;
  ScanFrontEnd scan_, scan_b_;  // double-buffered: scan k + 1 is staged into the one scan k is not using
  std::unique_ptr<Geo> geo_;
  std::shared_ptr<lidar::Context> photo_ctx_;  // declared before photo_: destroyed after it
  std::unique_ptr<Photometric> photo_;
};

class FixedLagReplay : public FixedLagReplayT<PlainGeometric>
{
public:
  FixedLagReplay(const std::shared_ptr<lidar::Context> & ctx, const Config & cfg, size_t lru_horizon = 1000)
  : FixedLagReplayT<PlainGeometric>(ctx, cfg, std::unique_ptr<PlainGeometric>(new PlainGeometric(ctx, cfg, lru_horizon)))
  {
  }
};

// ---- the same sequence through lidar::Manager::callback (manager.hpp) -------------------------------------------------
// Stand-ins for what the reference's Manager talks to and this build leaves to GTSAM: the IMU side integrates the scan's
// samples with constant acceleration / rate per sample (the arithmetic of propagate() above), the graph side keeps the
// fixed-lag window of WindowSmoother.  Reference semantics, which differ from FixedLagReplay in one place: the FIRST cloud
// initialises (its pose is the graph's initial pose, it is not registered, manager.cpp:111-121) — so the initial state handed
// in should be a good one.
class SampleImu : public imu::ManagerInterface
{
public:
  explicit SampleImu(const A3 & gravity) : g_(gravity) {}
  void load(const ImuSamples * s) { cur_samples_ = s; }
  void getInterpolatedMeasurements(const double, const double, ImuBuffer & out, const bool) override
  {
    out.clear();
    for (size_t j = 0; j < cur_samples_->ts.size(); ++j) {
      V6D m;
      for (int i = 0; i < 3; ++i) {
        m(i) = cur_samples_->acc[j][i];
        m(3 + i) = cur_samples_->gyro[j][i];
      }
      out.emplace_back(cur_samples_->ts[j], m);
    }
  }
  double gravityNorm() const override { return std::sqrt(g_[0] * g_[0] + g_[1] * g_[1] + g_[2] * g_[2]); }
  void resetIntegrationAndSetBias(const mimosa_hip::State & state) override
  {
    const PoseRM p = rowMajor(state.navState().pose());
    st_.T.R = p.R;
    st_.T.t = p.t;
    st_.vel = toArray(state.navState().velocity());
  }
  void integrateMeasurement(const V3D & acc, const V3D & gyro, const double d) override
  {
    const A3 Ra = matvec(st_.T.R, toArray(acc));
    const A3 aw{Ra[0] + g_[0], Ra[1] + g_[1], Ra[2] + g_[2]};
    const A9 Rn = matmul(st_.T.R, so3Expmap({gyro(0) * d, gyro(1) * d, gyro(2) * d}));
    for (int i = 0; i < 3; ++i) {
      st_.T.t[i] = st_.T.t[i] + st_.vel[i] * d + 0.5 * aw[i] * d * d;
      st_.vel[i] = st_.vel[i] + aw[i] * d;
    }
    st_.T.R = Rn;
  }
  gtsam::NavState predict(const mimosa_hip::State &) override { return gtsam::NavState(toPose3(st_.T), V3D(st_.vel[0], st_.vel[1], st_.vel[2])); }
  const State & integrated() const { return st_; }
  Unit3 gravityDirection() const
  {
    const double n = gravityNorm();
    return Unit3(g_[0] / n, g_[1] / n, g_[2] / n);
  }

private:
  A3 g_;
  const ImuSamples * cur_samples_ = nullptr;
  State st_;
};

class WindowGraph : public graph::ManagerInterface
{
public:
  WindowGraph(const Config & cfg, SampleImu & imu, const State & state0)
  : cfg_(cfg), imu_(imu), smoother_(cfg.window, cfg.update_iters, cfg.between_sigma_rot, cfg.between_sigma_trans), prev_(state0)
  {
  }
  void beginScan(const ScanInput & sc) { cur_ = &sc; }
  // the state "up to" the cloud's header stamp: the last optimised pose with the propagated velocity (first cloud: the caller's)
  void getStateUpto(const double, mimosa_hip::State & state) override
  {
    state.update(X(have_prev_ ? k_ : 0), cur_->imu.ts.front(), gtsam::NavState(toPose3(prev_.T), V3D(prev_.vel[0], prev_.vel[1], prev_.vel[2])), V3D(0, 0, 0),
                 V3D(0, 0, 0), imu_.gravityDirection());
  }
  graph::DeclarationResult declare(const double, size_t & new_key, const bool use_to_init) override
  {
    // the graph predicts the new state from its own IMU factor: the same integration the LiDAR manager runs for deskewing
    propagate(prev_, cur_->imu, cur_->header_ts, {}, cfg_.gravity, pred_);
    if (!have_prev_) {
      if (!use_to_init) return graph::DeclarationResult::FAILURE_CANNOT_INIT_ON_MODALITY;
      new_key = k_ = 0;
      return graph::DeclarationResult::SUCCESS_INITIALIZED;
    }
    new_key = ++k_;
    return graph::DeclarationResult::SUCCESS_NORMAL;
  }
  Pose3 getPoseAt(const size_t) override { return toPose3(pred_.T); }
  Values getCurrentOptimizedValues() override
  {
    Values v;
    v.insert(G(0), imu_.gravityDirection());
    return v;
  }
  // smoother_->update(): the new ICP factor joins the window, every live one is re-linearized per iteration
  void define(const NonlinearFactorGraph & new_factors, Values & optimized_values, const graph::DeclarationResult) override
  {
    WindowSmoother::Live lv;
    lv.k = k_;
    lv.T = pred_.T;
    lv.has_Z = have_prev_;
    if (have_prev_) lv.Z = between(prev_.T, pred_.T);
    PhotometricFactor::Ptr pf;
    for (const auto & f : new_factors) {
      if (auto icp = std::dynamic_pointer_cast<ICPFactor>(f)) lv.f = icp;
      if (auto ph = std::dynamic_pointer_cast<PhotometricFactor>(f)) pf = ph;
    }
    if (!lv.f) throw std::runtime_error("WindowGraph::define: no ICPFactor among the new factors");
    smoother_.push(lv);
    costs.push_back(smoother_.optimise(k_, pf));
    for (const auto & l : smoother_.win) optimized_values.insert_or_assign(X(l.k), toPose3(l.T));
    finish(smoother_.newest());
  }
  // the first cloud is not optimised: its pose becomes the previous state as it is
  void initialised() { finish(pred_.T); }
  std::vector<std::vector<double>> costs;

private:
  void finish(const RT & T)
  {
    prev_.vel = matvec(T.R, matvec(transpose(pred_.T.R), pred_.vel));  // the propagated velocity, carried into the corrected attitude
    prev_.T = T;
    have_prev_ = true;
  }
  Config cfg_;
  SampleImu & imu_;
  WindowSmoother smoother_;
  State prev_, pred_;
  const ScanInput * cur_ = nullptr;
  size_t k_ = 0;
  bool have_prev_ = false;
};

class ManagerReplay
{
public:
  ManagerReplay(const std::shared_ptr<lidar::Context> & ctx, const Config & cfg, size_t lru_horizon = 1000) : ctx_(ctx), cfg_(cfg), imu_(cfg.gravity)
  {
    lidar::ManagerConfig mc;
    mc.range_min = cfg.input.range_min;
    mc.range_max = cfg.input.range_max;
    mc.intensity_min = cfg.input.intensity_min;
    mc.intensity_max = cfg.input.intensity_max;
    mc.ns_max = cfg.input.ns_max;
    mc.z_offset = cfg.input.z_offset;
    mc.create_full_res_pointcloud = cfg.input.create_full_res_pointcloud != 0;
    gc_.point_skip_divisor = cfg.input.point_skip_divisor;
    gc_.ring_skip_divisor = cfg.input.ring_skip_divisor;
    gc_.map_keyframe_trans_thresh = static_cast<float>(cfg.keyframe_trans_thresh);
    gc_.map_keyframe_rot_thresh_deg = static_cast<float>(cfg.keyframe_rot_thresh_deg);
    gc_.initial_clouds_to_force_map_update = 0;
    gc_.lru_horizon = lru_horizon;
    gc_.neighbor_voxel_mode = cfg.neighbor_voxel_mode;
    gc_.scan_to_map = cfg.reg;
    mc_ = mc;
    pc_ = cfg.photo;
    pc_.enabled = cfg.photometric;
    if (cfg.photometric) photo_ctx_ = std::make_shared<lidar::Context>(ctx_->device());
  }
  Result run(const std::vector<ScanInput> & scans, const State & state0, const float * seed_xyz = nullptr, size_t n_seed = 0)
  {
    using clk = std::chrono::steady_clock;
    WindowGraph graph(cfg_, imu_, state0);
    lidar::Manager manager(ctx_, mc_, gc_, pc_, graph, imu_, photo_ctx_);
    if (n_seed) manager.geometric().map()->insert(seed_xyz, n_seed);
    Result res;
    const auto t0 = clk::now();
    for (size_t k = 0; k < scans.size(); ++k) {
      imu_.load(&scans[k].imu);
      graph.beginScan(scans[k]);
      manager.callback(scans[k].raw.data(), scans[k].raw.size(), scans[k].header_ts);
      if (k == 0) graph.initialised();
      const PoseRM p = rowMajor(manager.lastPose());
      RT T;
      T.R = p.R;
      T.t = p.t;
      res.poses.push_back(T);
      res.n_keyframes += manager.geometric().debug().map_updated ? 1 : 0;
      if (manager.photometric().debug().n_features_in_factor) res.photo_valid.push_back(manager.photometric().debug().n_status[8]);
      res.stage[0] += (manager.debug().t_deskew + manager.debug().t_preprocess_geo_photo) * 1e-3;
      res.stage[2] += manager.debug().t_factor_prep * 1e-3;
      res.stage[3] += manager.debug().t_define * 1e-3;
      res.stage[4] += manager.debug().t_post_define_update * 1e-3;
    }
    res.costs = graph.costs;
    res.seconds = std::chrono::duration<double>(clk::now() - t0).count();
    return res;
  }

private:
  std::shared_ptr<lidar::Context> ctx_;
  Config cfg_;
  SampleImu imu_;
  lidar::ManagerConfig mc_;
  lidar::GeometricConfig gc_;
  lidar::PhotometricConfig pc_;
  std::shared_ptr<lidar::Context> photo_ctx_;
};

}  // namespace replay
}  // namespace mimosa_hip
