// gtsam_sig: stand-in for <gtsam/geometry/Pose3.h>: the members the host mirror calls.  NOT GTSAM.
#pragma once
#include <gtsam/geometry/Rot3.h>

namespace gtsam
{
class Pose3
{
public:
  Pose3() = default;
  Pose3(const Rot3 & R, const Point3 & t) : R_(R), t_(t) {}
  static Pose3 Identity() { return Pose3(); }
  const Rot3 & rotation() const { return R_; }
  const Point3 & translation() const { return t_; }
  Pose3 inverse() const
  {
    const Rot3 Ri = R_.inverse();
    return Pose3(Ri, Point3(-(Ri.matrix() * t_)));
  }
  Pose3 operator*(const Pose3 & o) const { return Pose3(R_ * o.R_, Point3(t_ + R_.matrix() * o.t_)); }
  Pose3 between(const Pose3 & o) const { return inverse() * o; }
  Point3 transformFrom(const Point3 & p) const { return Point3(R_.matrix() * p + t_); }

private:
  Rot3 R_;
  Point3 t_;
};
}  // namespace gtsam
