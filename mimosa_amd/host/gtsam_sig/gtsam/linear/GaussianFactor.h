// gtsam_sig: stand-in for <gtsam/linear/GaussianFactor.h>.  NOT GTSAM.
#pragma once
#include <memory>

#include <gtsam/base/Matrix.h>
#include <gtsam/base/Vector.h>
#include <gtsam/inference/Key.h>

namespace gtsam
{
class GaussianFactor
{
public:
  typedef std::shared_ptr<GaussianFactor> shared_ptr;
  virtual ~GaussianFactor() = default;
  const KeyVector & keys() const { return keys_; }
  size_t size() const { return keys_.size(); }
  virtual Matrix information() const = 0;  // the quadratic term over all keys

protected:
  KeyVector keys_;
};
}  // namespace gtsam
