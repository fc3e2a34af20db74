// gtsam_sig: stand-in for <gtsam/base/Matrix.h> (GTSAM: typedefs of Eigen matrices).  NOT GTSAM — see ../../README.md.
#pragma once
#include <cassert>
#include <cstddef>
#include <initializer_list>
#include <vector>

namespace gtsam_sig
{
// dynamic column-major matrix with the few Eigen spellings the host mirror uses
class MatX
{
public:
  MatX() = default;
  MatX(int r, int c) : r_(r), c_(c), d_(static_cast<size_t>(r) * c, 0.0) {}
  double & operator()(int i, int j) { return d_[static_cast<size_t>(j) * r_ + i]; }
  double operator()(int i, int j) const { return d_[static_cast<size_t>(j) * r_ + i]; }
  int rows() const { return r_; }
  int cols() const { return c_; }
  void setZero() { d_.assign(d_.size(), 0.0); }
  static MatX Zero(int r, int c) { return MatX(r, c); }
  static MatX Identity(int r, int c)
  {
    MatX m(r, c);
    for (int i = 0; i < r && i < c; ++i) m(i, i) = 1.0;
    return m;
  }
  MatX transpose() const
  {
    MatX t(c_, r_);
    for (int i = 0; i < r_; ++i)
      for (int j = 0; j < c_; ++j) t(j, i) = (*this)(i, j);
    return t;
  }
  MatX operator*(const MatX & o) const
  {
    assert(c_ == o.r_);
    MatX m(r_, o.c_);
    for (int i = 0; i < r_; ++i)
      for (int j = 0; j < o.c_; ++j) {
        double s = 0.0;
        for (int k = 0; k < c_; ++k) s += (*this)(i, k) * o(k, j);
        m(i, j) = s;
      }
    return m;
  }
  MatX operator+(const MatX & o) const
  {
    MatX m(*this);
    for (size_t i = 0; i < d_.size(); ++i) m.d_[i] += o.d_[i];
    return m;
  }
  MatX operator-(const MatX & o) const
  {
    MatX m(*this);
    for (size_t i = 0; i < d_.size(); ++i) m.d_[i] -= o.d_[i];
    return m;
  }
  MatX operator-() const
  {
    MatX m(*this);
    for (double & v : m.d_) v = -v;
    return m;
  }
  MatX operator*(double s) const
  {
    MatX m(*this);
    for (double & v : m.d_) v *= s;
    return m;
  }

protected:
  int r_ = 0, c_ = 0;
  std::vector<double> d_;
};

class VecX : public MatX
{
public:
  VecX() = default;
  explicit VecX(int n) : MatX(n, 1) {}
  VecX(const MatX & m) : MatX(m) { assert(m.cols() == 1 || m.rows() == 0); }
  double & operator()(int i) { return d_[i]; }
  double operator()(int i) const { return d_[i]; }
  double & operator[](int i) { return d_[i]; }
  double operator[](int i) const { return d_[i]; }
  int size() const { return r_; }
  static VecX Zero(int n) { return VecX(n); }
  double dot(const VecX & o) const
  {
    double s = 0.0;
    for (int i = 0; i < r_; ++i) s += d_[i] * o.d_[i];
    return s;
  }
  double norm() const;
};

template <int R, int C>
class MatF : public MatX
{
public:
  MatF() : MatX(R, C) {}
  MatF(const MatX & m) : MatX(m) { assert(m.rows() == R && m.cols() == C); }
  static MatF Zero() { return MatF(); }
  static MatF Identity() { return MatF(MatX::Identity(R, C)); }
};

template <int N>
class VecF : public VecX
{
public:
  VecF() : VecX(N) {}
  VecF(const MatX & m) : VecX(m) { assert(m.rows() == N); }
  VecF(double x, double y, double z) : VecX(N)
  {
    static_assert(N == 3, "three-argument constructor is Vector3's");
    d_[0] = x;
    d_[1] = y;
    d_[2] = z;
  }
  static VecF Zero() { return VecF(); }
};
}  // namespace gtsam_sig

namespace gtsam
{
typedef gtsam_sig::MatX Matrix;
typedef gtsam_sig::MatF<3, 3> Matrix3;
typedef gtsam_sig::MatF<6, 6> Matrix6;
}  // namespace gtsam
