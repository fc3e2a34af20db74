// gtsam_sig: stand-in for <gtsam/base/Vector.h>.  NOT GTSAM — see ../../README.md.
#pragma once
#include <cmath>

#include <gtsam/base/Matrix.h>

namespace gtsam_sig
{
inline double VecX::norm() const { return std::sqrt(dot(*this)); }
}  // namespace gtsam_sig

namespace gtsam
{
typedef gtsam_sig::VecX Vector;
typedef gtsam_sig::VecF<3> Vector3;
typedef gtsam_sig::VecF<6> Vector6;
}  // namespace gtsam
