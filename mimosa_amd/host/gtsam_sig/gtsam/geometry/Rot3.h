// gtsam_sig: stand-in for <gtsam/geometry/Rot3.h>: the members the host mirror calls.  NOT GTSAM.
#pragma once
#include <gtsam/geometry/Point3.h>

namespace gtsam
{
class Rot3
{
public:
  Rot3() : R_(Matrix3::Identity()) {}
  explicit Rot3(const Matrix3 & R) : R_(R) {}
  static Rot3 Identity() { return Rot3(); }
  const Matrix3 & matrix() const { return R_; }
  Rot3 inverse() const { return Rot3(Matrix3(R_.transpose())); }
  Rot3 operator*(const Rot3 & o) const { return Rot3(Matrix3(R_ * o.R_)); }
  Point3 rotate(const Point3 & p) const { return Point3(R_ * p); }
  Point3 operator*(const Point3 & p) const { return rotate(p); }
  Rot3 between(const Rot3 & o) const { return inverse() * o; }
  static Rot3 Expmap(const Vector3 & w)  // Rodrigues
  {
    const double th2 = w.dot(w), th = std::sqrt(th2);
    Matrix3 K;
    K(0, 1) = -w(2);
    K(0, 2) = w(1);
    K(1, 0) = w(2);
    K(1, 2) = -w(0);
    K(2, 0) = -w(1);
    K(2, 1) = w(0);
    double A, B;
    if (th < 1e-10) {
      A = 1.0 - th2 / 6.0;
      B = 0.5 - th2 / 24.0;
    } else {
      A = std::sin(th) / th;
      B = (1.0 - std::cos(th)) / th2;
    }
    return Rot3(Matrix3(Matrix3::Identity() + K * A + (K * K) * B));
  }

private:
  Matrix3 R_;
};
}  // namespace gtsam
