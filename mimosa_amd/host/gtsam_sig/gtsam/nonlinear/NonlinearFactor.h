// gtsam_sig: stand-in for <gtsam/nonlinear/NonlinearFactor.h>: the virtuals ICPFactor / PhotometricFactor override
// (geometric_factor.hpp:160-174, :231).  NOT GTSAM.
#pragma once
#include <memory>

#include <gtsam/linear/GaussianFactor.h>
#include <gtsam/nonlinear/Values.h>

namespace gtsam
{
class NonlinearFactor
{
public:
  typedef std::shared_ptr<NonlinearFactor> shared_ptr;
  virtual ~NonlinearFactor() = default;
  const KeyVector & keys() const { return keys_; }
  size_t size() const { return keys_.size(); }
  virtual double error(const Values & c) const = 0;
  virtual size_t dim() const = 0;
  virtual std::shared_ptr<GaussianFactor> linearize(const Values & c) const = 0;
  virtual shared_ptr clone() const = 0;

protected:
  NonlinearFactor() = default;
  template <typename CONTAINER>
  explicit NonlinearFactor(const CONTAINER & keys) : keys_(keys.begin(), keys.end())
  {
  }
  KeyVector keys_;
};
}  // namespace gtsam
