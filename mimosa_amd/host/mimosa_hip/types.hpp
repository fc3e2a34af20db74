// Value types of the host mirror: GTSAM's own.
//
// The reference's ICPFactor is a gtsam::NonlinearFactor that reads c.at<gtsam::Pose3>(key) / c.at<gtsam::Unit3>(G(0)) and
// returns a gtsam::HessianFactor (include/mimosa/lidar/geometric_factor.hpp:25, :247-257, :459-462, :559-560); its V3D / M33 /
// M66 / V6D are Eigen typedefs (include/mimosa/utils.hpp).  The mirror is written against exactly those headers and
// signatures — there is no #ifdef and no second code path.  What <gtsam/...> resolves to is decided by the include path:
// a GTSAM installation in a deployment, or host/gtsam_sig (a signature stub: same paths, namespaces and member
// signatures, just enough behaviour for the tests) in this build environment, where GTSAM and Eigen are absent.
//
// The C ABI underneath speaks plain row-major double arrays; the helpers at the end convert (element-wise: Eigen matrices
// are column-major, nothing here relies on a memory layout).
#pragma once

#include <gtsam/base/Matrix.h>
#include <gtsam/base/Vector.h>
#include <gtsam/geometry/Pose3.h>
#include <gtsam/geometry/Unit3.h>
#include <gtsam/inference/Symbol.h>
#include <gtsam/linear/HessianFactor.h>
#include <gtsam/navigation/NavState.h>
#include <gtsam/nonlinear/NonlinearFactor.h>
#include <gtsam/nonlinear/NonlinearFactorGraph.h>
#include <gtsam/nonlinear/Values.h>

#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mimosa_hip
{
using gtsam::GaussianFactor;
using gtsam::HessianFactor;
using gtsam::Key;
using gtsam::KeyVector;
using gtsam::NonlinearFactor;
using gtsam::NonlinearFactorGraph;
using gtsam::Point3;
using gtsam::Pose3;
using gtsam::Rot3;
using gtsam::Unit3;
using gtsam::Values;
using gtsam::symbol_shorthand::G;  // the gravity direction lives under G(0) (geometric_factor.hpp:257)
using gtsam::symbol_shorthand::X;

// include/mimosa/utils.hpp: V3D = Eigen::Vector3d, M33 = Eigen::Matrix3d, M66, V6D
using V3D = gtsam::Vector3;
using M33 = gtsam::Matrix3;
using M66 = gtsam::Matrix6;
using V6D = gtsam::Vector6;

// ---- plain row-major arrays <-> GTSAM / Eigen values (the C ABI's side of the boundary) ------------------------------
using A3 = std::array<double, 3>;
using A9 = std::array<double, 9>;  // row-major 3 x 3

inline A9 rowMajor(const gtsam::Matrix3 & M)
{
  A9 a;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) a[3 * r + c] = M(r, c);
  return a;
}
inline A3 toArray(const gtsam::Vector3 & v) { return A3{v(0), v(1), v(2)}; }
inline M33 matrix3(const double * rm)
{
  M33 M;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) M(r, c) = rm[3 * r + c];
  return M;
}
inline V3D vector3(const double * v) { return V3D(v[0], v[1], v[2]); }
inline M66 matrix6(const double * rm)
{
  M66 M;
  for (int r = 0; r < 6; ++r)
    for (int c = 0; c < 6; ++c) M(r, c) = rm[6 * r + c];
  return M;
}
// a pose as the C ABI takes it: R row-major, t
struct PoseRM
{
  A9 R;
  A3 t;
};
inline PoseRM rowMajor(const Pose3 & T) { return PoseRM{rowMajor(T.rotation().matrix()), toArray(T.translation())}; }
inline Pose3 pose3(const double * R_rm, const double * t) { return Pose3(Rot3(matrix3(R_rm)), Point3(t[0], t[1], t[2])); }
inline void toFloat12(const Pose3 & T, float * Rt12)  // {R row-major 9, t 3} in float: mh_deskew / mh_transform_f32
{
  const PoseRM p = rowMajor(T);
  for (int i = 0; i < 9; ++i) Rt12[i] = static_cast<float>(p.R[i]);
  for (int i = 0; i < 3; ++i) Rt12[9 + i] = static_cast<float>(p.t[i]);
}

}  // namespace mimosa_hip
