// gtsam_sig: stand-in for <gtsam/navigation/NavState.h>.  NOT GTSAM.
#pragma once
#include <gtsam/geometry/Pose3.h>

namespace gtsam
{
class NavState
{
public:
  NavState() = default;
  NavState(const Pose3 & pose, const Velocity3 & v) : pose_(pose), v_(v) {}
  const Pose3 & pose() const { return pose_; }
  const Velocity3 & velocity() const { return v_; }
  const Velocity3 & v() const { return v_; }

private:
  Pose3 pose_;
  Velocity3 v_;
};
}  // namespace gtsam
