// gtsam_sig: stand-in for <gtsam/nonlinear/NonlinearFactorGraph.h>.  NOT GTSAM.
#pragma once
#include <vector>

#include <gtsam/nonlinear/NonlinearFactor.h>

namespace gtsam
{
class NonlinearFactorGraph
{
public:
  typedef std::vector<NonlinearFactor::shared_ptr>::const_iterator const_iterator;
  void add(const NonlinearFactor::shared_ptr & f) { factors_.push_back(f); }
  void push_back(const NonlinearFactor::shared_ptr & f) { factors_.push_back(f); }
  size_t size() const { return factors_.size(); }
  bool empty() const { return factors_.empty(); }
  const NonlinearFactor::shared_ptr & at(size_t i) const { return factors_.at(i); }
  const NonlinearFactor::shared_ptr & operator[](size_t i) const { return factors_[i]; }
  const_iterator begin() const { return factors_.begin(); }
  const_iterator end() const { return factors_.end(); }

private:
  std::vector<NonlinearFactor::shared_ptr> factors_;
};
}  // namespace gtsam
