// gtsam_sig: stand-in for <gtsam/linear/HessianFactor.h>: the two constructors the reference uses
// (geometric_factor.hpp:459-462 binary, :559-560 unary) and the read accessors.  error(x) = 0.5 x'Gx - x'g + 0.5 f.  NOT GTSAM.
#pragma once
#include <gtsam/linear/GaussianFactor.h>

namespace gtsam
{
class HessianFactor : public GaussianFactor
{
public:
  typedef std::shared_ptr<HessianFactor> shared_ptr;
  HessianFactor(Key j, const Matrix & G, const Vector & g, double f) : G_(G), g_(g), f_(f) { keys_ = {j}; }
  HessianFactor(Key j1, Key j2, const Matrix & G11, const Matrix & G12, const Vector & g1, const Matrix & G22, const Vector & g2, double f)
  : G_(G11.rows() + G22.rows(), G11.cols() + G22.cols()), g_(g1.size() + g2.size()), f_(f)
  {
    keys_ = {j1, j2};
    const int n1 = G11.rows(), n2 = G22.rows();
    for (int r = 0; r < n1; ++r) {
      for (int c = 0; c < n1; ++c) G_(r, c) = G11(r, c);
      for (int c = 0; c < n2; ++c) G_(r, n1 + c) = G_(n1 + c, r) = G12(r, c);
      g_(r) = g1(r);
    }
    for (int r = 0; r < n2; ++r) {
      for (int c = 0; c < n2; ++c) G_(n1 + r, n1 + c) = G22(r, c);
      g_(n1 + r) = g2(r);
    }
  }
  Matrix information() const override { return G_; }
  Vector linearTerm() const { return g_; }
  double constantTerm() const { return f_; }

private:
  Matrix G_;
  Vector g_;
  double f_;
};
}  // namespace gtsam
