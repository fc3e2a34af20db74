// gtsam_sig: stand-in for <gtsam/geometry/Point3.h> (GTSAM 4.x: Point3 is Vector3).  NOT GTSAM.
#pragma once
#include <gtsam/base/Vector.h>

namespace gtsam
{
typedef Vector3 Point3;
typedef Vector3 Velocity3;
}  // namespace gtsam
