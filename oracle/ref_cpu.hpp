// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// CPU restatement (C++17, zero dependencies) of the LiDAR geometric-factor hot
// path of ntnu-arl/mimosa, used as (a) the parity checker for the HIP path and
// (b) the timed "port" CPU baseline in bench.py.  Only tests/, bench.py's
// cpu_baseline leg and __graft_entry__.smoke() may load this.
//
// PARITY UNPINNED: the reference has no tests / golden vectors and
// cannot be compiled here (needs ROS, GTSAM, gtsam_points, PCL, Eigen — F6).  The
// k-NN arithmetic lives in ntnu-arl/gtsam_points@minimal_updated (no commit pin)
// and the 3x3 eigensolver in Eigen 3.3.7; both are absent from /root/reference
// and are restated here from their published algorithms.  The restatement is
// pinned only against an independent numpy restatement (oracle/numpy_ref.py) and
// the fixtures it generated (tests/golden/).
//
// Every function cites the reference file:line it follows (paths relative to
// /root/reference/mimosa/).
#pragma once

#include <algorithm>
#include <type_traits>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <unordered_map>
#include <utility>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace refcpu
{
// ----------------------------------------------------------------------------------------------
// Tiny linear algebra (row-major 3x3).  Stand-in for the Eigen fixed-size types the reference
// uses (include/mimosa/utils.hpp:57-110 typedefs V3D, M33, M66 ...).
// ----------------------------------------------------------------------------------------------
struct V3
{
  double x = 0, y = 0, z = 0;
  double & operator[](int i) { return (&x)[i]; }
  double operator[](int i) const { return (&x)[i]; }
};
inline V3 operator+(const V3 & a, const V3 & b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(const V3 & a, const V3 & b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, const V3 & a) { return {s * a.x, s * a.y, s * a.z}; }
inline V3 operator-(const V3 & a) { return {-a.x, -a.y, -a.z}; }
// Eigen's unrolled 3-element reduction is p0 + (p1 + p2) (redux_novec_unroller splits Length/2).
inline double dot(const V3 & a, const V3 & b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline V3 cross(const V3 & a, const V3 & b)
{
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline double norm(const V3 & a) { return std::sqrt(dot(a, a)); }

struct M3
{
  double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // row-major
  double & operator()(int r, int c) { return m[3 * r + c]; }
  double operator()(int r, int c) const { return m[3 * r + c]; }
  static M3 identity()
  {
    M3 r;
    r(0, 0) = r(1, 1) = r(2, 2) = 1;
    return r;
  }
};
inline V3 operator*(const M3 & A, const V3 & v)
{
  return {
    A(0, 0) * v.x + (A(0, 1) * v.y + A(0, 2) * v.z), A(1, 0) * v.x + (A(1, 1) * v.y + A(1, 2) * v.z),
    A(2, 0) * v.x + (A(2, 1) * v.y + A(2, 2) * v.z)};
}
inline M3 operator*(const M3 & A, const M3 & B)
{
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r(i, j) = A(i, 0) * B(0, j) + (A(i, 1) * B(1, j) + A(i, 2) * B(2, j));
  return r;
}
inline M3 operator-(const M3 & A, const M3 & B)
{
  M3 r;
  for (int i = 0; i < 9; ++i) r.m[i] = A.m[i] - B.m[i];
  return r;
}
inline M3 transpose(const M3 & A)
{
  M3 r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r(i, j) = A(j, i);
  return r;
}
// Eigen fixed-size 3x3 inverse = cofactors / determinant (Eigen/src/LU/InverseImpl.h).
inline M3 inverse(const M3 & A)
{
  M3 c;
  c(0, 0) = A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1);
  c(1, 0) = A(1, 2) * A(2, 0) - A(1, 0) * A(2, 2);
  c(2, 0) = A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0);
  const double det = A(0, 0) * c(0, 0) + A(0, 1) * c(1, 0) + A(0, 2) * c(2, 0);
  const double inv = 1.0 / det;
  M3 r;
  r(0, 0) = c(0, 0) * inv;
  r(1, 0) = c(1, 0) * inv;
  r(2, 0) = c(2, 0) * inv;
  r(0, 1) = (A(0, 2) * A(2, 1) - A(0, 1) * A(2, 2)) * inv;
  r(1, 1) = (A(0, 0) * A(2, 2) - A(0, 2) * A(2, 0)) * inv;
  r(2, 1) = (A(0, 1) * A(2, 0) - A(0, 0) * A(2, 1)) * inv;
  r(0, 2) = (A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1)) * inv;
  r(1, 2) = (A(0, 2) * A(1, 0) - A(0, 0) * A(1, 2)) * inv;
  r(2, 2) = (A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0)) * inv;
  return r;
}

// gtsam::Pose3 subset: compose / inverse / act (used at geometric_factor.hpp:247-253).
struct Pose
{
  M3 R = M3::identity();
  V3 t;
  Pose inverse() const
  {
    Pose r;
    r.R = transpose(R);
    r.t = -(r.R * t);
    return r;
  }
  Pose operator*(const Pose & o) const
  {
    Pose r;
    r.R = R * o.R;
    r.t = t + R * o.t;
    return r;
  }
  V3 operator*(const V3 & p) const { return R * p + t; }
};

// ----------------------------------------------------------------------------------------------
// Eigen::SelfAdjointEigenSolver<Matrix3d>::compute restated (Eigen 3.3.7,
// Eigen/src/Eigenvalues/SelfAdjointEigenSolver.h + Tridiagonalization.h + Jacobi.h):
// scale to [-1,1] -> 3x3 Householder tridiagonalisation special case -> implicit symmetric QR
// with Wilkinson shift -> ascending selection sort.  Used at geometric_factor.hpp:196 and
// include/mimosa/utils.hpp:308-313.  evecs is row-major with eigenvectors in COLUMNS.
// Returns false on NoConvergence (-> RejectStatus::EigenSolverFail).
// ----------------------------------------------------------------------------------------------
inline void make_givens(double p, double q, double & c, double & s)
{
  if (q == 0.0) {
    c = p < 0.0 ? -1.0 : 1.0;
    s = 0.0;
  } else if (p == 0.0) {
    c = 0.0;
    s = q < 0.0 ? 1.0 : -1.0;
  } else if (std::abs(p) > std::abs(q)) {
    const double t = q / p;
    double u = std::sqrt(1.0 + t * t);
    if (p < 0.0) u = -u;
    c = 1.0 / u;
    s = -t * c;
  } else {
    const double t = p / q;
    double u = std::sqrt(1.0 + t * t);
    if (q < 0.0) u = -u;
    s = -1.0 / u;
    c = -t * s;
  }
}

inline bool self_adjoint_eigen3(const M3 & A, V3 & evals, M3 & evecs)
{
  // Lower triangle only, scaled by the max abs coefficient.
  double mat[3][3];
  double scale = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j <= i; ++j) scale = std::max(scale, std::abs(A(i, j)));
  if (scale == 0.0) scale = 1.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) mat[i][j] = (j <= i) ? A(i, j) / scale : 0.0;

  double diag[3], sub[2];
  double Q[3][3];
  {  // tridiagonalization_inplace_selector<MatrixType,3,false>::run
    const double tol = std::numeric_limits<double>::min();
    diag[0] = mat[0][0];
    const double v1norm2 = mat[2][0] * mat[2][0];
    if (v1norm2 <= tol) {
      diag[1] = mat[1][1];
      diag[2] = mat[2][2];
      sub[0] = mat[1][0];
      sub[1] = mat[2][1];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Q[i][j] = (i == j) ? 1.0 : 0.0;
    } else {
      const double beta = std::sqrt(mat[1][0] * mat[1][0] + v1norm2);
      const double invBeta = 1.0 / beta;
      const double m01 = mat[1][0] * invBeta;
      const double m02 = mat[2][0] * invBeta;
      const double q = 2.0 * m01 * mat[2][1] + m02 * (mat[2][2] - mat[1][1]);
      diag[1] = mat[1][1] + m02 * q;
      diag[2] = mat[2][2] - m02 * q;
      sub[0] = beta;
      sub[1] = mat[2][1] - m01 * q;
      const double q0[3][3] = {{1, 0, 0}, {0, m01, m02}, {0, m02, -m01}};
      std::memcpy(Q, q0, sizeof(Q));
    }
  }

  // computeFromTridiagonal_impl, m_maxIterations = 30
  const int n = 3;
  int end = n - 1, start = 0, iter = 0;
  const int max_iter = 30;
  const double consider_as_zero = std::numeric_limits<double>::min();
  const double precision = 2.0 * std::numeric_limits<double>::epsilon();
  while (end > 0) {
    for (int i = start; i < end; ++i) {
      if (
        std::abs(sub[i]) <= (std::abs(diag[i]) + std::abs(diag[i + 1])) * precision ||
        std::abs(sub[i]) <= consider_as_zero)
        sub[i] = 0.0;
    }
    while (end > 0 && sub[end - 1] == 0.0) end--;
    if (end <= 0) break;
    iter++;
    if (iter > max_iter * n) break;
    start = end - 1;
    while (start > 0 && sub[start - 1] != 0.0) start--;

    // tridiagonal_qr_step
    const double td = (diag[end - 1] - diag[end]) * 0.5;
    const double e = sub[end - 1];
    double mu = diag[end];
    if (td == 0.0) {
      mu -= std::abs(e);
    } else {
      const double e2 = e * e;
      const double h = std::hypot(td, e);
      if (e2 == 0.0)
        mu -= (e / (td + (td > 0.0 ? 1.0 : -1.0))) * (e / h);
      else
        mu -= e2 / (td + (td > 0.0 ? h : -h));
    }
    double x = diag[start] - mu;
    double z = sub[start];
    for (int k = start; k < end; ++k) {
      double c, s;
      make_givens(x, z, c, s);
      const double sdk = s * diag[k] + c * sub[k];
      const double dkp1 = s * sub[k] + c * diag[k + 1];
      diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
      diag[k + 1] = s * sdk + c * dkp1;
      sub[k] = c * sdk - s * dkp1;
      if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
      x = sub[k];
      if (k < end - 1) {
        z = -s * sub[k + 1];
        sub[k + 1] = c * sub[k + 1];
      }
      // Q = Q * G  (applyOnTheRight(k, k+1, rot))
      for (int i = 0; i < 3; ++i) {
        const double xi = Q[i][k], yi = Q[i][k + 1];
        Q[i][k] = c * xi - s * yi;
        Q[i][k + 1] = s * xi + c * yi;
      }
    }
  }
  const bool ok = iter <= max_iter * n;
  if (ok) {
    for (int i = 0; i < n - 1; ++i) {
      int k = 0;
      for (int j = 1; j < n - i; ++j)
        if (diag[i + j] < diag[i + k]) k = j;
      if (k > 0) {
        std::swap(diag[i], diag[k + i]);
        for (int r = 0; r < 3; ++r) std::swap(Q[r][i], Q[r][k + i]);
      }
    }
  }
  for (int i = 0; i < 3; ++i) {
    evals[i] = diag[i] * scale;
    for (int j = 0; j < 3; ++j) evecs(i, j) = Q[i][j];
  }
  return ok;
}

// include/mimosa/utils.hpp:308-313
inline void compute_localizability(const M3 & JtJ, V3 & localizability, M3 & eigenvectors)
{
  V3 ev;
  self_adjoint_eigen3(JtJ, ev, eigenvectors);
  localizability = {std::sqrt(ev.x), std::sqrt(ev.y), std::sqrt(ev.z)};
}

// include/mimosa/lidar/utils.hpp:191-213
inline bool get_projection_matrix(
  const V3 & localizability, double thresh, const M3 & eigenvectors, M3 & P, V3 & degenerate_axes)
{
  if (localizability.x > thresh && localizability.y > thresh && localizability.z > thresh) {
    P = M3::identity();
    degenerate_axes = {0, 0, 0};
    return false;
  }
  P = M3();
  degenerate_axes = {1, 1, 1};
  for (int i = 0; i < 3; ++i) {
    if (localizability[i] > thresh) {
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) P(r, c) += eigenvectors(r, i) * eigenvectors(c, i);
      degenerate_axes[i] = 0;
    }
  }
  return true;
}

// ----------------------------------------------------------------------------------------------
// Voxel hashing primitives  (include/mimosa/lidar/utils.hpp:218-238; identical helpers live in
// gtsam_points/util/fast_floor.hpp and vector3i_hash.hpp)
// ----------------------------------------------------------------------------------------------
inline int fast_floor(double v)
{
  const int n = static_cast<int>(v);
  return n - (v < static_cast<double>(n) ? 1 : 0);
}
struct Coord
{
  int x, y, z;
  bool operator==(const Coord & o) const { return x == o.x && y == o.y && z == o.z; }
};
struct XORVector3iHash
{
  size_t operator()(const Coord & c) const
  {
    const size_t p1 = 9132043225175502913ull;
    const size_t p2 = 7277549399757405689ull;
    const size_t p3 = 6673468629021231217ull;
    return static_cast<size_t>((c.x * p1) ^ (c.y * p2) ^ (c.z * p3));
  }
};

inline std::vector<Coord> neighbor_offsets(int mode)
{
  // gtsam_points/ann/incremental_voxelmap.hpp neighbor_offsets() — see SURVEY.md Appendix B
  switch (mode) {
    case 1:
      return {{0, 0, 0}};
    case 7:
      return {{0, 0, 0}, {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    case 19:
    case 27: {
      std::vector<Coord> o;
      for (int i = -1; i <= 1; ++i)
        for (int j = -1; j <= 1; ++j)
          for (int k = -1; k <= 1; ++k) {
            if (mode == 19 && std::abs(i) == 1 && std::abs(j) == 1 && std::abs(k) == 1) continue;
            o.push_back({i, j, k});
          }
      return o;
    }
    default:
      return {};
  }
}

// ----------------------------------------------------------------------------------------------
// gtsam_points::iVox = IncrementalVoxelMap<FlatContainer> restated with the reference's data
// structure shape (unordered_map -> shared_ptr voxel -> 32-byte Vector4d points), so the timed CPU
// baseline is neither straw-manned nor steel-manned.  Call sites: incremental_voxel_map.cpp:14-62,
// geometric.cpp:23-28,491-495, geometric_factor.hpp:184,294.
// ----------------------------------------------------------------------------------------------
struct P4
{
  double x, y, z, w;
};

struct FlatContainer
{
  std::vector<P4> points;
};

struct VoxelInfo
{
  Coord coord;
  size_t lru;
};

class IVox
{
public:
  explicit IVox(double leaf_size) : inv_leaf_size_(1.0 / leaf_size), offsets_(neighbor_offsets(7)) {}

  // Copy: mimosa declares "Deep copy constructor ... assumes that gtsam_points::iVox has a proper
  // copy constructor" (include/mimosa/lidar/incremental_voxel_map.hpp:33-42) and relies on it so
  // that live factors keep an immutable snapshot (src/lidar/geometric.cpp:494).  Whether the
  // ntnu-arl fork really deep-copies is unverifiable here; the stated intent
  // — a deep copy — is what both this oracle and the HIP path implement.
  IVox(const IVox & o)
  : inv_leaf_size_(o.inv_leaf_size_), lru_horizon_(o.lru_horizon_), lru_clear_cycle_(o.lru_clear_cycle_),
    lru_counter_(o.lru_counter_), min_sq_dist_in_cell_(o.min_sq_dist_in_cell_),
    max_num_points_in_cell_(o.max_num_points_in_cell_), offsets_(o.offsets_), voxels_(o.voxels_)
  {
    flat_voxels_.reserve(o.flat_voxels_.size());
    for (const auto & v : o.flat_voxels_)
      flat_voxels_.push_back(std::make_shared<std::pair<VoxelInfo, FlatContainer>>(*v));
  }

  void set_lru_horizon(size_t h) { lru_horizon_ = h; }
  void set_lru_clear_cycle(size_t c) { lru_clear_cycle_ = c; }
  void set_neighbor_voxel_mode(int mode) { offsets_ = neighbor_offsets(mode); }
  void set_min_dist_in_cell(double d) { min_sq_dist_in_cell_ = d * d; }
  void set_max_num_points_in_cell(size_t n) { max_num_points_in_cell_ = n; }

  // insert(): points are float xyz promoted to double (PointCloudCPU(vector<Vector3f>) stores
  // Vector4d(x,y,z,1)).  incremental_voxel_map.cpp:19-24.
  void insert(const float * xyz, size_t n, size_t stride_floats = 3)
  {
    for (size_t i = 0; i < n; ++i) {
      const P4 pt{
        static_cast<double>(xyz[i * stride_floats + 0]), static_cast<double>(xyz[i * stride_floats + 1]),
        static_cast<double>(xyz[i * stride_floats + 2]), 1.0};
      const Coord coord{
        fast_floor(pt.x * inv_leaf_size_), fast_floor(pt.y * inv_leaf_size_),
        fast_floor(pt.z * inv_leaf_size_)};
      auto found = voxels_.find(coord);
      if (found == voxels_.end()) {
        auto voxel = std::make_shared<std::pair<VoxelInfo, FlatContainer>>(
          VoxelInfo{coord, lru_counter_}, FlatContainer());
        found = voxels_.emplace_hint(found, coord, flat_voxels_.size());
        flat_voxels_.emplace_back(voxel);
      }
      auto & entry = *flat_voxels_[found->second];
      entry.first.lru = lru_counter_;
      add(entry.second, pt);
    }
    if ((++lru_counter_) % lru_clear_cycle_ == 0) {
      auto rm = std::remove_if(
        flat_voxels_.begin(), flat_voxels_.end(),
        [&](const std::shared_ptr<std::pair<VoxelInfo, FlatContainer>> & v) {
          return v->first.lru + lru_horizon_ < lru_counter_;
        });
      flat_voxels_.erase(rm, flat_voxels_.end());
      voxels_.clear();
      for (size_t i = 0; i < flat_voxels_.size(); ++i) voxels_[flat_voxels_[i]->first.coord] = i;
    }
  }

  // knn_search(): returns the number found; the mimosa wrapper demands == k
  // (incremental_voxel_map.cpp:26-32).  Global id = (voxel_index << 32) | point_index.
  size_t knn_search(
    const double pt[3], size_t k, size_t * k_indices, double * k_sq_dists,
    double max_sq_dist = std::numeric_limits<double>::max(), size_t * n_candidates = nullptr) const
  {
    const Coord center{
      fast_floor(pt[0] * inv_leaf_size_), fast_floor(pt[1] * inv_leaf_size_),
      fast_floor(pt[2] * inv_leaf_size_)};
    // KnnResult<-1>: distances initialised to max_sq_dist, strict '<' insertion sort.
    for (size_t i = 0; i < k; ++i) {
      k_indices[i] = static_cast<size_t>(-1);
      k_sq_dists[i] = max_sq_dist;
    }
    size_t num_found = 0, cand = 0;
    for (const auto & off : offsets_) {
      const Coord coord{center.x + off.x, center.y + off.y, center.z + off.z};
      const auto found = voxels_.find(coord);
      if (found == voxels_.end()) continue;
      const size_t voxel_index = found->second;
      const auto & pts = flat_voxels_[voxel_index]->second.points;
      cand += pts.size();
      for (size_t j = 0; j < pts.size(); ++j) {
        const double dx = pts[j].x - pt[0], dy = pts[j].y - pt[1], dz = pts[j].z - pt[2];
        // Eigen SSE2 Vector4d squaredNorm: (dx2 + dz2) + (dy2 + dw2), dw = 0
        const double d = (dx * dx + dz * dz) + (dy * dy + 0.0);
        if (d >= k_sq_dists[k - 1]) continue;
        int loc = static_cast<int>(std::min(num_found, k - 1));
        for (; loc > 0 && d < k_sq_dists[loc - 1]; --loc) {
          k_indices[loc] = k_indices[loc - 1];
          k_sq_dists[loc] = k_sq_dists[loc - 1];
        }
        k_indices[loc] = (voxel_index << 32) | j;
        k_sq_dists[loc] = d;
        num_found = std::min(num_found + 1, k);
      }
    }
    if (n_candidates) *n_candidates = cand;
    return num_found;
  }

  const P4 & point(size_t id) const { return flat_voxels_[id >> 32]->second.points[id & 0xffffffffull]; }

  size_t num_voxels() const { return flat_voxels_.size(); }
  size_t num_points() const
  {
    size_t n = 0;
    for (const auto & v : flat_voxels_) n += v->second.points.size();
    return n;
  }
  // voxel_data(): all points, voxel order (incremental_voxel_map.cpp:34-38)
  void voxel_data(std::vector<float> & xyz) const
  {
    xyz.clear();
    for (const auto & v : flat_voxels_)
      for (const auto & p : v->second.points) {
        xyz.push_back(static_cast<float>(p.x));
        xyz.push_back(static_cast<float>(p.y));
        xyz.push_back(static_cast<float>(p.z));
      }
  }
  // Export (coord, points) per voxel in flat order — lets tests mirror the exact map content.
  void export_voxels(std::vector<int> & coords, std::vector<int> & counts, std::vector<float> & xyz) const
  {
    coords.clear();
    counts.clear();
    xyz.clear();
    for (const auto & v : flat_voxels_) {
      coords.push_back(v->first.coord.x);
      coords.push_back(v->first.coord.y);
      coords.push_back(v->first.coord.z);
      counts.push_back(static_cast<int>(v->second.points.size()));
      for (const auto & p : v->second.points) {
        xyz.push_back(static_cast<float>(p.x));
        xyz.push_back(static_cast<float>(p.y));
        xyz.push_back(static_cast<float>(p.z));
      }
    }
  }
  double inv_leaf_size() const { return inv_leaf_size_; }

private:
  // FlatContainer::add — same rule mimosa restates at include/mimosa/lidar/utils.hpp:260-278
  void add(FlatContainer & c, const P4 & pt) const
  {
    if (c.points.size() >= max_num_points_in_cell_) return;
    for (const auto & e : c.points) {
      const double dx = e.x - pt.x, dy = e.y - pt.y, dz = e.z - pt.z;
      const double d = (dx * dx + dz * dz) + (dy * dy + 0.0);
      if (d < min_sq_dist_in_cell_) return;
    }
    c.points.push_back(pt);
  }

  double inv_leaf_size_;
  size_t lru_horizon_ = 100;
  size_t lru_clear_cycle_ = 10;
  size_t lru_counter_ = 0;
  double min_sq_dist_in_cell_ = 0.1 * 0.1;
  size_t max_num_points_in_cell_ = 20;
  std::vector<Coord> offsets_;
  std::unordered_map<Coord, size_t, XORVector3iHash> voxels_;
  std::vector<std::shared_ptr<std::pair<VoxelInfo, FlatContainer>>> flat_voxels_;
};

// ----------------------------------------------------------------------------------------------
// lidar::Point (include/mimosa/lidar/point.hpp:18-39): 32 bytes
// ----------------------------------------------------------------------------------------------
struct Point32
{
  float x, y, z, pad;
  float intensity;
  uint32_t t;
  uint32_t idx;
  float range;
};
static_assert(sizeof(Point32) == 32, "lidar::Point must be 32 bytes");

// include/mimosa/lidar/geometric_config.hpp:17-33 (defaults = struct defaults)
struct RegistrationConfig
{
  float source_voxel_grid_filter_leaf_size = 0.5f;
  float source_voxel_grid_min_dist_in_voxel = 0.1f;
  float target_ivox_map_leaf_size = 0.5f;
  float target_ivox_map_min_dist_in_voxel = 0.1f;
  uint64_t num_corres_points = 5;
  float max_corres_distance = 2.24f;
  float plane_validity_distance = 0.04f;
  float lidar_point_noise_std_dev = 0.02f;
  int32_t use_huber = 1;
  float huber_threshold = 1.345f;
  int32_t reg_4_dof = 0;
  int32_t project_on_degneneracy = 1;
  float degen_thresh_rot = 10;
  float degen_thresh_trans = 15;
};

enum RejectStatus : int32_t {
  Unprocessed = 0,
  InsufficientCorresPoints,
  CorresMaxDist,
  EigenSolverFail,
  MinEigenValueLow,
  Line,
  CorresPlaneInvalid,
  MaxError,
  Valid
};

// What gtsam::HessianFactor(key[,key2], G11,[G12,] g1,[G22, g2,] f) receives
// (geometric_factor.hpp:459-462, 559-560), plus every getter-visible side output.
struct LinearizeResult
{
  double H_ss[36];  // J_source^T J_source (row-major 6x6), after optional 4-DoF / degeneracy projection
  double H_st[36];  // binary only
  double H_tt[36];  // binary only
  double b_s[6];    // J_source^T e   (HessianFactor gets -b_s)
  double b_t[6];    // binary only
  double f;
  double loc_trans_comp[3], loc_rot_comp[3], loc_trans_final[3], loc_rot_final[3];
  double eigvec_trans[9], eigvec_rot[9];
  double degen_rot[3], degen_trans[3], degen_eigvec_rot[9], degen_eigvec_trans[9];
  int32_t status_hist[9];
  int32_t linearize_count;
  double mean_candidates;  // mean C_q over points that ran k-NN this call
  int64_t n_knn;           // number of points that ran k-NN this call
};

// ----------------------------------------------------------------------------------------------
// lidar::ICPFactor restated (include/mimosa/lidar/geometric_factor.hpp:25-563)
// ----------------------------------------------------------------------------------------------
class ICPFactor
{
public:
  ICPFactor(
    bool is_binary, std::shared_ptr<IVox> target, const Point32 * cloud, size_t n,
    const RegistrationConfig & config)
  : is_binary_(is_binary), ivox_target_(std::move(target)), cloud_source_(cloud, cloud + n), config_(config)
  {
    // commonConstructor(), :144-156
    transed_point_target_.assign(n, V3());
    transed_point_target_da_.assign(n, V3());
    corres_means_target_.assign(n, V3());
    corres_normals_target_.assign(n, V3());
    statuses_.assign(n, Unprocessed);
    localizabilities_trans_body_.assign(n, V3());
    localizabilities_rot_body_.assign(n, V3());
    linearize_count_ = 0;
  }

  ICPFactor(const ICPFactor &) = default;  // clone() deep-copies per-point state, shares the map (:160-164)

  size_t size() const { return cloud_source_.size(); }
  size_t dim() const { return 6; }                 // :166
  double error() const { return 0.0; }             // :168-174 (prints, returns 0)
  int n_threads = 4;                               // :261 (hard-coded 4 in the reference)

  const std::vector<int32_t> & statuses() const { return statuses_; }
  const std::vector<V3> & means() const { return corres_means_target_; }
  const std::vector<V3> & normals() const { return corres_normals_target_; }
  const std::vector<V3> & transed() const { return transed_point_target_; }
  const std::vector<V3> & transed_da() const { return transed_point_target_da_; }
  // test tooling for the map-sharded layer: a point that migrates to another rank takes its association state along
  void set_state(const int32_t * st, const double * means, const double * normals, const double * q_da, int linearize_count)
  {
    const size_t n = size();
    for (size_t i = 0; i < n; ++i) {
      statuses_[i] = st[i];
      corres_means_target_[i] = {means[3 * i], means[3 * i + 1], means[3 * i + 2]};
      corres_normals_target_[i] = {normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
      transed_point_target_da_[i] = {q_da[3 * i], q_da[3 * i + 1], q_da[3 * i + 2]};
    }
    linearize_count_ = linearize_count;
  }

  // estimatePlane, :176-229
  bool estimate_plane(size_t i, const size_t * idx, const V3 & source_origin_in_target)
  {
    const size_t k = config_.num_corres_points;
    std::vector<V3> pts(k);
    for (size_t j = 0; j < k; ++j) {
      const P4 & p = ivox_target_->point(idx[j]);
      pts[j] = {p.x, p.y, p.z};
    }
    V3 mean;
    for (size_t j = 0; j < k; ++j) mean = mean + pts[j];
    mean = (1.0 / static_cast<double>(k)) * mean;  // colwise().mean() = sum / rows
    corres_means_target_[i] = mean;

    std::vector<V3> centered(k);
    for (size_t j = 0; j < k; ++j) centered[j] = pts[j] - mean;
    M3 cov;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        double s = 0;
        for (size_t j = 0; j < k; ++j) s += centered[j][r] * centered[j][c];
        cov(r, c) = s / static_cast<double>(k - 1);
      }

    V3 evals;
    M3 evecs;
    if (!self_adjoint_eigen3(cov, evals, evecs)) {
      statuses_[i] = EigenSolverFail;
      return false;
    }
    if (evals.x < 1e-6) {
      statuses_[i] = MinEigenValueLow;
      return false;
    }
    if (evals.z > 3 * evals.y) {
      statuses_[i] = Line;
      return false;
    }
    V3 nrm{evecs(0, 0), evecs(1, 0), evecs(2, 0)};
    if (dot(nrm, source_origin_in_target - mean) < 0) nrm = -nrm;
    corres_normals_target_[i] = nrm;
    for (size_t j = 0; j < k; ++j) {
      if (std::abs(dot(centered[j], nrm)) > static_cast<double>(config_.plane_validity_distance)) {
        statuses_[i] = CorresPlaneInvalid;
        return false;
      }
    }
    return true;
  }

  // linearize, :231-562.  T_src = Values[keys[0]]; T_tgt = binary ? Values[keys[1]] : identity;
  // g_unit = Values[G(0)] unit vector (read unconditionally, :257).
  void linearize(const Pose & T_src, const Pose * T_tgt, const V3 & g_unit, LinearizeResult & out)
  {
    linearize_count_++;
    const Pose T_W_B_target = (is_binary_ && T_tgt) ? *T_tgt : Pose();
    const Pose delta_pose = T_W_B_target.inverse() * T_src;
    const V3 source_origin_in_target = delta_pose * V3();
    const V3 global_z = -g_unit;
    const M3 Rt = transpose(delta_pose.R);
    const V3 local_z = Rt * global_z;
    M3 rot_projection_mat;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) rot_projection_mat(r, c) = local_z[r] * local_z[c];

    const size_t N = cloud_source_.size();
    const int nt = std::max(1, n_threads);
    struct alignas(128) Acc  // one thread's sums: its own cache lines (unpadded, neighbouring threads shared a line — an artefact of this port)
    {
      double ss[36], st[36], tt[36], bs[6], bt[6], f;
      int64_t n_knn, n_cand;
    };
    std::vector<Acc> acc(nt);
    for (auto & a : acc) std::memset(&a, 0, sizeof(Acc));

    const double sigma = static_cast<double>(config_.lidar_point_noise_std_dev);
    const size_t k = config_.num_corres_points;

#ifdef _OPENMP
#pragma omp parallel for num_threads(nt) schedule(static)
#endif
    for (size_t i = 0; i < N; ++i) {
#ifdef _OPENMP
      const int tid = omp_get_thread_num();
#else
      const int tid = 0;
#endif
      Acc & A = acc[tid];
      const Point32 & sp = cloud_source_[i];
      const V3 p{static_cast<double>(sp.x), static_cast<double>(sp.y), static_cast<double>(sp.z)};
      transed_point_target_[i] = delta_pose * p;

      bool update_correspondance = false;
      if (
        norm(transed_point_target_[i] - transed_point_target_da_[i]) >
        static_cast<double>(config_.target_ivox_map_min_dist_in_voxel / 4)) {
        update_correspondance = true;
        transed_point_target_da_[i] = transed_point_target_[i];
      }

      if (update_correspondance) {
        statuses_[i] = Unprocessed;
        size_t k_indices[64];
        double sq_dists[64];
        const double q[3] = {transed_point_target_[i].x, transed_point_target_[i].y, transed_point_target_[i].z};
        size_t cand = 0;
        const size_t found = ivox_target_->knn_search(
          q, k, k_indices, sq_dists, std::numeric_limits<double>::max(), &cand);
        A.n_knn++;
        A.n_cand += static_cast<int64_t>(cand);
        if (found != k) {
          statuses_[i] = InsufficientCorresPoints;
          continue;
        }
        if (
          sq_dists[k - 1] >
          static_cast<double>(config_.max_corres_distance * config_.max_corres_distance)) {
          statuses_[i] = CorresMaxDist;
          continue;
        }
        if (!estimate_plane(i, k_indices, source_origin_in_target)) continue;
      } else {
        if (statuses_[i] <= CorresPlaneInvalid) continue;
      }

      double e = dot(corres_normals_target_[i], corres_means_target_[i] - transed_point_target_[i]);
      const double s = 1 - 0.9 * std::fabs(e) / std::sqrt(norm(p));
      if (s < 0.9) {
        statuses_[i] = MaxError;
        continue;
      }
      double sqrt_weight = 1.0;
      if (config_.use_huber) {
        const double whitened_error = e / sigma;
        if (std::fabs(whitened_error) > static_cast<double>(config_.huber_threshold))
          sqrt_weight = std::sqrt(static_cast<double>(config_.huber_threshold) / std::fabs(whitened_error));
      }
      e *= sqrt_weight / sigma;

      const V3 ns = Rt * corres_normals_target_[i];
      const V3 cr = cross(ns, p);
      double J[6] = {cr.x, cr.y, cr.z, -ns.x, -ns.y, -ns.z};
      const double nr = std::sqrt(cr.x * cr.x + cr.y * cr.y + cr.z * cr.z);
      // Eigen normalized(): returns the vector unchanged when the squared norm is 0
      localizabilities_rot_body_[i] = nr > 0 ? V3{cr.x / nr, cr.y / nr, cr.z / nr} : cr;
      localizabilities_trans_body_[i] = {J[3], J[4], J[5]};
      const double w = sqrt_weight / sigma;
      for (double & v : J) v *= w;

      for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) A.ss[6 * r + c] += J[r] * J[c];
        A.bs[r] += J[r] * e;
      }
      A.f += e * e;

      if (is_binary_) {
        const V3 ct = cross(transed_point_target_[i], corres_normals_target_[i]);
        const V3 & nt_ = corres_normals_target_[i];
        double Jt[6] = {ct.x, ct.y, ct.z, nt_.x, nt_.y, nt_.z};
        for (double & v : Jt) v *= w;
        for (int r = 0; r < 6; ++r) {
          for (int c = 0; c < 6; ++c) {
            A.st[6 * r + c] += J[r] * Jt[c];
            A.tt[6 * r + c] += Jt[r] * Jt[c];
          }
          A.bt[r] += Jt[r] * e;
        }
      }
      statuses_[i] = Valid;
    }

    // :389-403 merge in thread order
    Acc T;
    std::memset(&T, 0, sizeof(T));
    for (int t = 0; t < nt; ++t) {
      for (int j = 0; j < 36; ++j) {
        T.ss[j] += acc[t].ss[j];
        T.st[j] += acc[t].st[j];
        T.tt[j] += acc[t].tt[j];
      }
      for (int j = 0; j < 6; ++j) {
        T.bs[j] += acc[t].bs[j];
        T.bt[j] += acc[t].bt[j];
      }
      T.f += acc[t].f;
      T.n_knn += acc[t].n_knn;
      T.n_cand += acc[t].n_cand;
    }

    auto block = [&](const double * H, int r0, int c0) {
      M3 B;
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) B(r, c) = H[6 * (r0 + r) + (c0 + c)];
      return B;
    };
    // :405-428
    M3 Hrr = block(T.ss, 0, 0), Hrt = block(T.ss, 0, 3), Htr = block(T.ss, 3, 0), Htt = block(T.ss, 3, 3);
    compute_localizability(Hrr, loc_rot_final_, eig_rot_);
    compute_localizability(Htt, loc_trans_final_, eig_trans_);
    const M3 Sigma_rr = inverse(Hrr - Hrt * inverse(Htt) * Htr);
    const M3 Sigma_tt = inverse(Htt - Htr * inverse(Hrr) * Hrt);
    compute_localizability(Sigma_rr, degen_rot_, degen_eig_rot_);
    compute_localizability(Sigma_tt, degen_trans_, degen_eig_trans_);
    const double rad2deg = 57.29578;  // RAD2DEG is PCL's macro ((x)*57.29578), pcl/pcl_macros.h — the reference defines none
    degen_rot_ = rad2deg * degen_rot_;

    // :434-457 component localizabilities
    loc_trans_comp_ = V3();
    loc_rot_comp_ = V3();
    for (size_t i = 0; i < N; ++i) {
      if (statuses_[i] != Valid) continue;
      for (int c = 0; c < 3; ++c) {
        const V3 et{eig_trans_(0, c), eig_trans_(1, c), eig_trans_(2, c)};
        const V3 er{eig_rot_(0, c), eig_rot_(1, c), eig_rot_(2, c)};
        const double tc = std::fabs(dot(localizabilities_trans_body_[i], et));
        const double rc = std::fabs(dot(localizabilities_rot_body_[i], er));
        if (tc >= 0.5) loc_trans_comp_[c] += tc;
        if (rc >= 0.5) loc_rot_comp_[c] += rc;
      }
    }

    // :459-561
    if (!is_binary_) {
      if (config_.reg_4_dof) {
        const M3 & P = rot_projection_mat;
        const M3 a = P * Hrr * P, b = P * Hrt, c = Htr * P;
        for (int r = 0; r < 3; ++r)
          for (int cc = 0; cc < 3; ++cc) {
            T.ss[6 * r + cc] = a(r, cc);
            T.ss[6 * r + 3 + cc] = b(r, cc);
            T.ss[6 * (3 + r) + cc] = c(r, cc);
          }
        const V3 br = P * V3{T.bs[0], T.bs[1], T.bs[2]};
        T.bs[0] = br.x;
        T.bs[1] = br.y;
        T.bs[2] = br.z;
      }
      if (config_.project_on_degneneracy) {
        M3 P_rot, P_trans;
        V3 ax_rot, ax_trans;
        const bool rot_degen = get_projection_matrix(
          loc_rot_final_, static_cast<double>(config_.degen_thresh_rot), eig_rot_, P_rot, ax_rot);
        const bool trans_degen = get_projection_matrix(
          loc_trans_final_, static_cast<double>(config_.degen_thresh_trans), eig_trans_, P_trans, ax_trans);
        if (rot_degen || trans_degen) {
          // Reference quirk: H,b are rebuilt from J_source_whitened_weighted_arr /
          // e_whitened_weigehted_arr, which are allocated zero (:270-271) and never written, so
          // the rebuilt H and b are exactly zero (:496-532), then localizabilities are recomputed
          // from that zero matrix (:550-555).
          std::memset(T.ss, 0, sizeof(T.ss));
          std::memset(T.bs, 0, sizeof(T.bs));
          compute_localizability(M3(), loc_rot_final_, eig_rot_);
          compute_localizability(M3(), loc_trans_final_, eig_trans_);
        }
      }
    }

    std::memcpy(out.H_ss, T.ss, sizeof(T.ss));
    std::memcpy(out.H_st, T.st, sizeof(T.st));
    std::memcpy(out.H_tt, T.tt, sizeof(T.tt));
    std::memcpy(out.b_s, T.bs, sizeof(T.bs));
    std::memcpy(out.b_t, T.bt, sizeof(T.bt));
    out.f = T.f;
    for (int c = 0; c < 3; ++c) {
      out.loc_trans_comp[c] = loc_trans_comp_[c];
      out.loc_rot_comp[c] = loc_rot_comp_[c];
      out.loc_trans_final[c] = loc_trans_final_[c];
      out.loc_rot_final[c] = loc_rot_final_[c];
      out.degen_rot[c] = degen_rot_[c];
      out.degen_trans[c] = degen_trans_[c];
    }
    std::memcpy(out.eigvec_trans, eig_trans_.m, sizeof(double) * 9);
    std::memcpy(out.eigvec_rot, eig_rot_.m, sizeof(double) * 9);
    std::memcpy(out.degen_eigvec_rot, degen_eig_rot_.m, sizeof(double) * 9);
    std::memcpy(out.degen_eigvec_trans, degen_eig_trans_.m, sizeof(double) * 9);
    std::memset(out.status_hist, 0, sizeof(out.status_hist));
    for (size_t i = 0; i < N; ++i) out.status_hist[statuses_[i]]++;
    out.linearize_count = linearize_count_;
    out.n_knn = T.n_knn;
    out.mean_candidates = T.n_knn ? static_cast<double>(T.n_cand) / static_cast<double>(T.n_knn) : 0.0;
  }

  // Per-point whitened residual / Jacobian row for the CURRENT cached association at a given pose —
  // test helper (recomputes steps 5-7 of SURVEY.md Appendix A without touching state).
  bool point_row(size_t i, const Pose & delta_pose, double & e_out, double J_out[6]) const
  {
    if (statuses_[i] != Valid) return false;
    const Point32 & sp = cloud_source_[i];
    const V3 p{static_cast<double>(sp.x), static_cast<double>(sp.y), static_cast<double>(sp.z)};
    const V3 q = delta_pose * p;
    double e = dot(corres_normals_target_[i], corres_means_target_[i] - q);
    const double sigma = static_cast<double>(config_.lidar_point_noise_std_dev);
    double sw = 1.0;
    if (config_.use_huber) {
      const double we = e / sigma;
      if (std::fabs(we) > static_cast<double>(config_.huber_threshold))
        sw = std::sqrt(static_cast<double>(config_.huber_threshold) / std::fabs(we));
    }
    e *= sw / sigma;
    const V3 ns = transpose(delta_pose.R) * corres_normals_target_[i];
    const V3 cr = cross(ns, p);
    const double w = sw / sigma;
    J_out[0] = cr.x * w;
    J_out[1] = cr.y * w;
    J_out[2] = cr.z * w;
    J_out[3] = -ns.x * w;
    J_out[4] = -ns.y * w;
    J_out[5] = -ns.z * w;
    e_out = e;
    return true;
  }

private:
  bool is_binary_;
  std::shared_ptr<IVox> ivox_target_;
  std::vector<Point32> cloud_source_;
  RegistrationConfig config_;
  std::vector<V3> transed_point_target_, transed_point_target_da_, corres_means_target_, corres_normals_target_;
  std::vector<int32_t> statuses_;
  std::vector<V3> localizabilities_trans_body_, localizabilities_rot_body_;
  V3 loc_trans_comp_, loc_rot_comp_, loc_trans_final_, loc_rot_final_;
  M3 eig_trans_ = M3::identity(), eig_rot_ = M3::identity();
  V3 degen_rot_, degen_trans_;
  M3 degen_eig_rot_, degen_eig_trans_;
  int linearize_count_ = 0;
};

// ----------------------------------------------------------------------------------------------
// f32 rigid transform p <- R*p + t, the reference's operation order without FMA:
// Eigen lazy 3x3*3x1 coefficient product = r0*x + (r1*y + r2*z), then + t.
// Deskew hot loop src/lidar/manager.cpp:498-509; body transform src/lidar/geometric.cpp:154-161;
// world transform src/lidar/geometric.cpp:483-490.
// ----------------------------------------------------------------------------------------------
inline void transform_f32(const float R[9], const float t[3], float & x, float & y, float & z)
{
  const float px = x, py = y, pz = z;
  x = (R[0] * px + (R[1] * py + R[2] * pz)) + t[0];
  y = (R[3] * px + (R[4] * py + R[5] * pz)) + t[1];
  z = (R[6] * px + (R[7] * py + R[8] * pz)) + t[2];
}

// PointOuster (include/mimosa/lidar/point.hpp:42-50) and the configuration fields prepareInput reads
// (include/mimosa/lidar/manager.hpp:24-41, geometric_config.hpp:43-44).
struct PointOusterIn
{
  float x, y, z, pad;
  float intensity;
  uint32_t t;
  uint16_t reflectivity, ring;
  uint32_t pad2;
};
struct InputConfig
{
  float range_min, range_max, intensity_min, intensity_max, ns_max, z_offset;
  int32_t create_full_res_pointcloud, point_skip_divisor, ring_skip_divisor;
};
struct PreparedInput
{
  std::vector<Point32> points_full;                  // manager.cpp:312-313
  std::vector<uint64_t> geometric_point_idxs;        // :334
  std::vector<uint32_t> unique_ns;                   // :344-368
  std::vector<std::vector<uint64_t>> idxs_at_unique_ns;
  uint32_t last_point_ns = 0;                        // :310, corrected_ts_ = header + last * 1e-9 (:336)
};

// Manager::prepareInput<PointOuster> (src/lidar/manager.cpp:149-383) without the ROS / PCL conversion,
// the transpose / organise branches (other sensors) and the debug timers.
inline void prepare_input(const PointOusterIn * in, size_t n, const InputConfig & cfg, PreparedInput & out)
{
  out = PreparedInput();
  const float range_min_sq = cfg.range_min * cfg.range_min;  // manager.cpp:19-20
  const float range_max_sq = cfg.range_max * cfg.range_max;
  const size_t point_skip = static_cast<size_t>(cfg.point_skip_divisor);
  const size_t skip_divisor = cfg.create_full_res_pointcloud ? 1 : point_skip;  // :244-245
  std::vector<std::pair<uint32_t, uint64_t>> ns_idx_pairs;
  uint32_t last_point_ns = 0;
  for (size_t i = 0; i < n; i = i + skip_divisor) {
    const PointOusterIn & pin = in[i];
    if (std::isnan(pin.x) || std::isnan(pin.y) || std::isnan(pin.z)) continue;  // :253
    if (std::isnan(pin.intensity) || pin.intensity < cfg.intensity_min || pin.intensity > cfg.intensity_max) continue;  // :272-276
    const float range_sq = pin.x * pin.x + pin.y * pin.y + pin.z * pin.z;  // :281
    if (range_sq < range_min_sq || range_sq > range_max_sq) continue;      // :282
    const uint32_t t_ns = pin.t;                                           // :289
    if (t_ns > cfg.ns_max) continue;                                       // :306 (uint32 -> float comparison)
    last_point_ns = std::max(last_point_ns, t_ns);
    Point32 p{};
    p.x = pin.x;
    p.y = pin.y;
    p.z = pin.z + cfg.z_offset;
    p.intensity = pin.intensity;
    p.t = t_ns;
    p.idx = static_cast<uint32_t>(i);
    p.range = std::sqrt(range_sq);
    out.points_full.push_back(p);
    const uint64_t new_idx = out.points_full.size() - 1;
    ns_idx_pairs.emplace_back(t_ns, new_idx);
    if (i % point_skip != 0) continue;                                                  // :318
    if (pin.ring % static_cast<uint16_t>(cfg.ring_skip_divisor) != 0) continue;          // :331
    out.geometric_point_idxs.push_back(new_idx);
  }
  out.last_point_ns = last_point_ns;
  // :340-342 std::sort on the timestamp only: the order inside one timestamp is unspecified there; a stable
  // sort is one admissible outcome (only the membership of each group is consumed, :504-508)
  std::stable_sort(ns_idx_pairs.begin(), ns_idx_pairs.end(), [](const auto & a, const auto & b) { return a.first < b.first; });
  for (size_t i = 0; i < ns_idx_pairs.size(); ++i) {
    if (i == 0 || ns_idx_pairs[i].first != ns_idx_pairs[i - 1].first) {
      out.unique_ns.push_back(ns_idx_pairs[i].first);
      out.idxs_at_unique_ns.emplace_back();
    }
    out.idxs_at_unique_ns.back().push_back(ns_idx_pairs[i].second);
  }
}

// ---- the reference's other point types (include/mimosa/lidar/point.hpp:52-131), restated with the same member order
// and 16-byte alignment (PCL_ADD_POINT4D = float x, y, z + one float of padding) ---------------------------------
struct alignas(16) PointOusterOdysseyIn
{
  float x, y, z, pad;
  uint32_t t;
  uint16_t reflectivity, near_ir;
};
struct alignas(16) PointOusterR8In
{
  float x, y, z, pad;
  float intensity;
  uint32_t t;
  uint16_t reflectivity;
  uint8_t ring;
};
struct alignas(16) PointHesaiIn
{
  float x, y, z, pad;
  float intensity;
  double timestamp;
  uint16_t ring;
};
struct alignas(16) PointLivoxIn
{
  float x, y, z, pad;
  float intensity;
  uint8_t tag, line;
  double timestamp;
};
struct alignas(16) PointLivoxFromCustom2In
{
  float x, y, z;
  uint32_t t;
  float intensity;
  uint8_t tag, line;
};
struct alignas(16) PointVelodyneIn
{
  float x, y, z, pad;
  float intensity;
  uint16_t ring;
  float time;
};
struct alignas(16) PointVelodyneAnyboticsIn
{
  float x, y, z, pad;
  float intensity;
  float ring;
  float time;
};
struct alignas(16) PointRslidarIn
{
  float x, y, z, pad;
  float intensity;
  uint16_t ring;
  double timestamp;
};

struct InputOrder  // lidar/manager.hpp:28-29 + the PointCloud2 shape
{
  uint32_t width = 0, height = 1;
  bool transpose_pointcloud = false, organize_pointcloud_by_ring = false;
  double header_ts = 0.0;
};

// Manager::prepareInput<PointT> (src/lidar/manager.cpp:149-383) for every point type, branch for branch.
template <typename PointT>
inline void prepare_input_typed(const PointT * in, size_t n, const InputConfig & cfg, const InputOrder & ord, PreparedInput & out)
{
  out = PreparedInput();
  std::vector<PointT> cloud(in, in + n);
  uint32_t width = ord.width, height = ord.height;
  if constexpr (std::is_same<PointT, PointRslidarIn>::value || std::is_same<PointT, PointVelodyneAnyboticsIn>::value) {  // :177-203
    if (ord.transpose_pointcloud) {
      std::vector<PointT> tr(cloud.size());
      const uint32_t tw = height, th = width;
      for (size_t i = 0; i < cloud.size(); ++i) {
        const size_t current_row = i / width, current_col = i % width;
        tr[current_col * tw + current_row] = cloud[i];
      }
      cloud = tr;
      width = tw;
      height = th;
    }
  }
  if constexpr (!std::is_same<PointT, PointLivoxIn>::value && !std::is_same<PointT, PointLivoxFromCustom2In>::value &&
                !std::is_same<PointT, PointOusterOdysseyIn>::value) {  // :205-241
    if (ord.organize_pointcloud_by_ring && height == 1) {
      constexpr uint32_t num_rings = 128;
      std::array<size_t, num_rings> ring_counts{};
      for (const auto & point : cloud) ++ring_counts[static_cast<size_t>(point.ring)];
      std::array<size_t, num_rings> ring_offsets{};
      size_t offset = 0;
      for (uint32_t i = 0; i < num_rings; ++i) {
        ring_offsets[i] = offset;
        offset += ring_counts[i];
      }
      std::vector<PointT> organized(cloud.size());
      std::array<size_t, num_rings> ring_cursors = ring_offsets;
      for (const auto & point : cloud) organized[ring_cursors[static_cast<size_t>(point.ring)]++] = point;
      cloud = std::move(organized);
    }
  }
  const float range_min_sq = cfg.range_min * cfg.range_min;  // manager.cpp:19-20
  const float range_max_sq = cfg.range_max * cfg.range_max;
  const size_t point_skip = static_cast<size_t>(cfg.point_skip_divisor);
  const size_t skip_divisor = cfg.create_full_res_pointcloud ? 1 : point_skip;  // :244-245
  std::vector<std::pair<uint32_t, uint64_t>> ns_idx_pairs;
  uint32_t last_point_ns = 0;
  for (size_t i = 0; i < cloud.size(); i = i + skip_divisor) {
    const PointT & pin = cloud[i];
    if (std::isnan(pin.x) || std::isnan(pin.y) || std::isnan(pin.z)) continue;  // :253
    if constexpr (std::is_same<PointT, PointLivoxIn>::value || std::is_same<PointT, PointLivoxFromCustom2In>::value) {  // :256-262
      if (!((pin.tag & 0x30) == 0x10 || (pin.tag & 0x30) == 0x00)) continue;
    }
    float intensity = 0.0f;
    if constexpr (std::is_same<PointT, PointOusterOdysseyIn>::value) {  // :265-271
      if (pin.reflectivity < cfg.intensity_min || pin.reflectivity > cfg.intensity_max) continue;
      intensity = static_cast<float>(pin.reflectivity);
    } else {
      if (std::isnan(pin.intensity) || pin.intensity < cfg.intensity_min || pin.intensity > cfg.intensity_max) continue;  // :272-276
      intensity = pin.intensity;
    }
    const float range_sq = pin.x * pin.x + pin.y * pin.y + pin.z * pin.z;  // :281
    if (range_sq < range_min_sq || range_sq > range_max_sq) continue;      // :282
    uint32_t t_ns;  // :285-304
    if constexpr (std::is_same<PointT, PointOusterIn>::value || std::is_same<PointT, PointOusterOdysseyIn>::value ||
                  std::is_same<PointT, PointOusterR8In>::value || std::is_same<PointT, PointLivoxFromCustom2In>::value) {
      t_ns = pin.t;
    } else if constexpr (std::is_same<PointT, PointHesaiIn>::value || std::is_same<PointT, PointRslidarIn>::value) {
      t_ns = static_cast<uint32_t>((pin.timestamp - ord.header_ts) * 1e9);
    } else if constexpr (std::is_same<PointT, PointLivoxIn>::value) {
      t_ns = static_cast<uint32_t>(pin.timestamp - ord.header_ts * 1e9);
    } else {
      t_ns = static_cast<uint32_t>(pin.time * 1e9);  // PointVelodyne, PointVelodyneAnybotics: float * double
    }
    if (t_ns > cfg.ns_max) continue;  // :306
    last_point_ns = std::max(last_point_ns, t_ns);
    Point32 p{};
    p.x = pin.x;
    p.y = pin.y;
    p.z = pin.z + cfg.z_offset;
    p.intensity = intensity;
    p.t = t_ns;
    p.idx = static_cast<uint32_t>(i);
    p.range = std::sqrt(range_sq);
    out.points_full.push_back(p);
    const uint64_t new_idx = out.points_full.size() - 1;
    ns_idx_pairs.emplace_back(t_ns, new_idx);
    if (i % point_skip != 0) continue;  // :318
    if constexpr (!std::is_same<PointT, PointLivoxIn>::value && !std::is_same<PointT, PointLivoxFromCustom2In>::value &&
                  !std::is_same<PointT, PointVelodyneAnyboticsIn>::value && !std::is_same<PointT, PointOusterOdysseyIn>::value) {  // :321-332
      if (pin.ring % cfg.ring_skip_divisor != 0) continue;
    }
    out.geometric_point_idxs.push_back(new_idx);
  }
  out.last_point_ns = last_point_ns;
  std::stable_sort(ns_idx_pairs.begin(), ns_idx_pairs.end(), [](const auto & a, const auto & b) { return a.first < b.first; });  // :340-342
  for (size_t i = 0; i < ns_idx_pairs.size(); ++i) {
    if (i == 0 || ns_idx_pairs[i].first != ns_idx_pairs[i - 1].first) {
      out.unique_ns.push_back(ns_idx_pairs[i].first);
      out.idxs_at_unique_ns.emplace_back();
    }
    out.idxs_at_unique_ns.back().push_back(ns_idx_pairs[i].second);
  }
}

// deskewPoints (iii): every point whose t equals unique_ns[g] gets pose g.  Rt12 = per group
// row-major R (9 floats) then t (3 floats), already cast from fp64 (manager.cpp:502-503).
inline void deskew(
  Point32 * pts, size_t n, const uint32_t * unique_ns, const float * Rt12, size_t n_groups)
{
  for (size_t i = 0; i < n; ++i) {
    const uint32_t * it = std::lower_bound(unique_ns, unique_ns + n_groups, pts[i].t);
    if (it == unique_ns + n_groups || *it != pts[i].t) continue;  // not in any group: untouched
    const float * P = Rt12 + 12 * static_cast<size_t>(it - unique_ns);
    transform_f32(P, P + 9, pts[i].x, pts[i].y, pts[i].z);
  }
}

// Geometric::downsample (src/lidar/geometric.cpp:55-126) with FlatContainerMinimal::add
// (include/mimosa/lidar/utils.hpp:260-278).  Returns kept indices: voxels in first-seen order,
// points within a voxel in acceptance order.
inline void downsample(
  const Point32 * pts, size_t n, double leaf_size, size_t max_points_per_voxel,
  double min_dist_in_voxel, std::vector<uint32_t> & kept)
{
  const double inv_leaf = 1.0 / leaf_size;
  const double min_sq = min_dist_in_voxel * min_dist_in_voxel;
  struct Cell
  {
    std::vector<V3> pts;
    std::vector<uint32_t> idx;
  };
  std::vector<Cell> flat;
  std::unordered_map<Coord, size_t, XORVector3iHash> voxels;
  voxels.reserve(n / 2);
  for (size_t i = 0; i < n; ++i) {
    const V3 p{static_cast<double>(pts[i].x), static_cast<double>(pts[i].y), static_cast<double>(pts[i].z)};
    const Coord c{fast_floor(p.x * inv_leaf), fast_floor(p.y * inv_leaf), fast_floor(p.z * inv_leaf)};
    auto f = voxels.find(c);
    if (f == voxels.end()) {
      f = voxels.emplace_hint(f, c, flat.size());
      flat.emplace_back();
    }
    Cell & cell = flat[f->second];
    if (cell.pts.size() >= max_points_per_voxel) continue;
    bool close = false;
    for (const auto & e : cell.pts) {
      const double dx = e.x - p.x, dy = e.y - p.y, dz = e.z - p.z;
      if (dx * dx + (dy * dy + dz * dz) < min_sq) {  // Vector3d squaredNorm: p0 + (p1 + p2)
        close = true;
        break;
      }
    }
    if (close) continue;
    cell.pts.push_back(p);
    cell.idx.push_back(static_cast<uint32_t>(i));
  }
  kept.clear();
  for (const auto & cell : flat) kept.insert(kept.end(), cell.idx.begin(), cell.idx.end());
}

}  // namespace refcpu
