// Minimal value types for the host mirror of the reference's LiDAR classes.
//
// The reference's ICPFactor is a gtsam::NonlinearFactor and returns a gtsam::HessianFactor
// (include/mimosa/lidar/geometric_factor.hpp:25, :459-462, :559-560).  GTSAM is not available in this
// build environment, so the mirror is written against this small interface whose names and
// signatures follow GTSAM's: compile with -DMIMOSA_HIP_WITH_GTSAM and provide the aliases below from
// <gtsam/...> to drop the classes into a real factor graph.
#pragma once

#include <array>
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mimosa_hip
{
using Key = std::uint64_t;
using V3D = std::array<double, 3>;
using M33 = std::array<double, 9>;   // row-major
using M66 = std::array<double, 36>;  // row-major
using V6D = std::array<double, 6>;

// gtsam::Pose3 subset (rotation matrix + translation)
struct Pose3
{
  M33 R{1, 0, 0, 0, 1, 0, 0, 0, 1};
  V3D t{0, 0, 0};
  static Pose3 Identity() { return Pose3(); }
  Pose3 inverse() const
  {
    Pose3 r;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) r.R[3 * i + j] = R[3 * j + i];
    for (int i = 0; i < 3; ++i) r.t[i] = -(r.R[3 * i] * t[0] + r.R[3 * i + 1] * t[1] + r.R[3 * i + 2] * t[2]);
    return r;
  }
  Pose3 operator*(const Pose3 & o) const
  {
    Pose3 r;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        r.R[3 * i + j] = R[3 * i] * o.R[j] + R[3 * i + 1] * o.R[3 + j] + R[3 * i + 2] * o.R[6 + j];
    for (int i = 0; i < 3; ++i) r.t[i] = t[i] + R[3 * i] * o.t[0] + R[3 * i + 1] * o.t[1] + R[3 * i + 2] * o.t[2];
    return r;
  }
  const V3D & translation() const { return t; }
  const M33 & rotation() const { return R; }
};

// gtsam::Values subset: poses by key + the gravity direction Unit3 stored under G(0)
// (linearize reads it unconditionally, geometric_factor.hpp:257)
class Values
{
public:
  void insert(Key k, const Pose3 & p) { poses_[k] = p; }
  void update(Key k, const Pose3 & p) { poses_[k] = p; }
  const Pose3 & atPose3(Key k) const
  {
    auto it = poses_.find(k);
    if (it == poses_.end()) throw std::out_of_range("Values: no Pose3 for key");
    return it->second;
  }
  void setGravity(const V3D & unit) { g_ = unit; }
  const V3D & gravityUnit() const { return g_; }

private:
  std::map<Key, Pose3> poses_;
  V3D g_{0, 0, -1};
};

// gtsam::HessianFactor as ICPFactor constructs it: unary (key, G, g, f) or binary
// (k1, k2, G11, G12, g1, G22, g2, f); error = 0.5 x'Gx - x'g + 0.5 f
struct GaussianFactor
{
  virtual ~GaussianFactor() = default;
};
struct HessianFactor : GaussianFactor
{
  std::vector<Key> keys;
  M66 G11{}, G12{}, G22{};
  V6D g1{}, g2{};
  double f = 0.0;
};

class NonlinearFactor
{
public:
  using shared_ptr = std::shared_ptr<NonlinearFactor>;
  explicit NonlinearFactor(std::vector<Key> keys) : keys_(std::move(keys)) {}
  virtual ~NonlinearFactor() = default;
  const std::vector<Key> & keys() const { return keys_; }
  virtual std::shared_ptr<GaussianFactor> linearize(const Values & c) const = 0;
  virtual shared_ptr clone() const = 0;
  virtual size_t dim() const = 0;
  virtual double error(const Values & c) const = 0;

private:
  std::vector<Key> keys_;
};

struct NonlinearFactorGraph
{
  std::vector<NonlinearFactor::shared_ptr> factors;
  void add(const NonlinearFactor::shared_ptr & f) { factors.push_back(f); }
};

}  // namespace mimosa_hip
