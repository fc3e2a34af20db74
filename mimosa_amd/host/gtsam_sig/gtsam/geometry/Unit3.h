// gtsam_sig: stand-in for <gtsam/geometry/Unit3.h>.  NOT GTSAM.
#pragma once
#include <gtsam/geometry/Point3.h>

namespace gtsam
{
class Unit3
{
public:
  Unit3() : p_(1.0, 0.0, 0.0) {}
  explicit Unit3(const Vector3 & p) : p_(p)
  {
    const double n = p.norm();
    if (n > 0.0) p_ = Vector3(p * (1.0 / n));
  }
  Unit3(double x, double y, double z) : Unit3(Vector3(x, y, z)) {}
  Vector3 unitVector() const { return p_; }
  const Vector3 & point3() const { return p_; }

private:
  Vector3 p_;
};
}  // namespace gtsam
