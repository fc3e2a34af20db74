// Pin kit (tools/pin_reference/README.md): runs the REAL mimosa::lidar::ICPFactor::linearize and
// IncrementalVoxelMapPCL::knn_search on the inputs of this repository's golden cases and writes their outputs, so that the
// CPU oracle (oracle/ref_cpu.hpp) and the HIP path can be held to the reference itself.  Compiled INSIDE a mimosa catkin
// workspace (Eigen, GTSAM, gtsam_points, PCL present); it contains no code of the build under test.
//
// Input file  (uint64 length-prefixed little-endian vectors, in this order):
//   int32  I[6]    : neighbor_voxel_mode, lru_horizon, is_binary, num_corres_points, use_huber | reg_4_dof << 1 |
//                    project_on_degneneracy << 2, reserved
//   double D[12]   : source_voxel_grid_filter_leaf_size, source_voxel_grid_min_dist_in_voxel, target_ivox_map_leaf_size,
//                    target_ivox_map_min_dist_in_voxel, max_corres_distance, plane_validity_distance,
//                    lidar_point_noise_std_dev, huber_threshold, degen_thresh_rot, degen_thresh_trans, reserved, reserved
//   float  map[3 M]: the map cloud, inserted in ONE insert() call
//   float  src[3 N]: the source cloud (body frame, deskewed)
//   double poses[] : pass 1 source pose (R row-major 9, t 3), pass 2 source pose (12), target pose (12; identity when unary)
//   double g[3]    : the Unit3 stored under G(0)
// Output file:
//   int32  checks[2]            : abs(double) is the double overload, the map copy is deep
//   per pass p = 1, 2:  double H[n*n] (n = 6 unary, 12 binary; the HessianFactor's information block, row-major),
//                       double g[n] (linear term as stored: -J^T b), double f, int32 status[N], double means[3N],
//                       double normals[3N], double loc[12] (trans_comp, rot_comp, trans_final, rot_final),
//                       double eig[18] (eigenvectors_trans, eigenvectors_rot, row-major), double degen[6] (rot, trans)
//   knn (pass-1 pose, every 8th source point): int32 found[Q], double sq_dists[Q*k], double points[Q*k*3]
#include <gtsam/geometry/Pose3.h>
#include <gtsam/geometry/Unit3.h>
#include <gtsam/linear/HessianFactor.h>
#include <gtsam/nonlinear/Values.h>

#include <cstdint>
#include <fstream>
#include <iostream>
#include <type_traits>
#include <vector>

#include "mimosa/lidar/geometric_factor.hpp"

namespace mimosa
{
namespace lidar
{
// the expression of geometric_factor.hpp:323,334 — an unqualified abs on a double — looked up from the same namespace,
// behind the same includes
static_assert(std::is_same<decltype(abs(std::declval<double>())), double>::value, "abs(double) does not resolve to the double overload");
inline bool abs_probe()
{
  const double e = 0.4;
  return abs(e) == 0.4 && abs(-e) == 0.4;
}
}  // namespace lidar
}  // namespace mimosa

namespace
{
template <typename T>
std::vector<T> read_vec(std::ifstream & f)
{
  uint64_t n = 0;
  f.read(reinterpret_cast<char *>(&n), 8);
  std::vector<T> v(n);
  f.read(reinterpret_cast<char *>(v.data()), static_cast<std::streamsize>(n * sizeof(T)));
  if (!f) throw std::runtime_error("pin: truncated input");
  return v;
}
template <typename T>
void write_vec(std::ofstream & f, const std::vector<T> & v)
{
  const uint64_t n = v.size();
  f.write(reinterpret_cast<const char *>(&n), 8);
  f.write(reinterpret_cast<const char *>(v.data()), static_cast<std::streamsize>(n * sizeof(T)));
}
gtsam::Pose3 pose_from(const std::vector<double> & p, size_t o)
{
  gtsam::Matrix3 R;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R(r, c) = p[o + 3 * r + c];
  return gtsam::Pose3(gtsam::Rot3(R), gtsam::Point3(p[o + 9], p[o + 10], p[o + 11]));
}
pcl::PointCloud<mimosa::lidar::Point> cloud_from(const std::vector<float> & xyz)
{
  pcl::PointCloud<mimosa::lidar::Point> c;
  c.resize(xyz.size() / 3);
  for (size_t i = 0; i < c.size(); ++i) {
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    c.points[i] = mimosa::lidar::Point(x, y, z, 0.f, 0u, static_cast<uint32_t>(i), std::sqrt(x * x + y * y + z * z));
  }
  return c;
}
void put3(std::vector<double> & out, const mimosa::V3D & v)
{
  out.push_back(v.x());
  out.push_back(v.y());
  out.push_back(v.z());
}
void put33(std::vector<double> & out, const mimosa::M33 & m)
{
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) out.push_back(m(r, c));
}
}  // namespace

int main(int argc, char ** argv)
{
  using namespace mimosa;
  using namespace mimosa::lidar;
  if (argc < 3) {
    std::cerr << "usage: mimosa_pin <case.in> <case.out>\n";
    return 2;
  }
  std::ifstream in(argv[1], std::ios::binary);
  if (!in) {
    std::cerr << "cannot open " << argv[1] << "\n";
    return 2;
  }
  const auto I = read_vec<int32_t>(in);
  const auto D = read_vec<double>(in);
  const auto map_xyz = read_vec<float>(in);
  const auto src_xyz = read_vec<float>(in);
  const auto poses = read_vec<double>(in);
  const auto gz = read_vec<double>(in);
  if (I.size() < 5 || D.size() < 10 || poses.size() < 36 || gz.size() < 3) throw std::runtime_error("pin: malformed input");

  RegistrationConfig cfg;
  cfg.source_voxel_grid_filter_leaf_size = static_cast<float>(D[0]);
  cfg.source_voxel_grid_min_dist_in_voxel = static_cast<float>(D[1]);
  cfg.target_ivox_map_leaf_size = static_cast<float>(D[2]);
  cfg.target_ivox_map_min_dist_in_voxel = static_cast<float>(D[3]);
  cfg.num_corres_points = static_cast<size_t>(I[3]);
  cfg.max_corres_distance = static_cast<float>(D[4]);
  cfg.plane_validity_distance = static_cast<float>(D[5]);
  cfg.lidar_point_noise_std_dev = static_cast<float>(D[6]);
  cfg.use_huber = (I[4] & 1) != 0;
  cfg.huber_threshold = static_cast<float>(D[7]);
  cfg.reg_4_dof = (I[4] & 2) != 0;
  cfg.project_on_degneneracy = (I[4] & 4) != 0;
  cfg.degen_thresh_rot = static_cast<float>(D[8]);
  cfg.degen_thresh_trans = static_cast<float>(D[9]);
  const bool binary = I[2] != 0;

  // the map, set up as lidar::Geometric does (src/lidar/geometric.cpp:25-30)
  auto map = std::make_shared<IncrementalVoxelMapPCL>(cfg.target_ivox_map_leaf_size);
  map->underlying()->set_lru_horizon(static_cast<size_t>(I[1]));
  map->underlying()->set_neighbor_voxel_mode(static_cast<size_t>(I[0]));
  map->underlying()->voxel_insertion_setting().set_min_dist_in_cell(cfg.target_ivox_map_min_dist_in_voxel);
  map->insert(cloud_from(map_xyz));

  std::vector<int32_t> checks(2, 0);
  checks[0] = abs_probe() ? 1 : 0;
  {
    // geometric.cpp:494 — "Reset the map to a new map": is the copy deep?
    const size_t before = map->getCloud()->size();
    auto copy = std::make_shared<IncrementalVoxelMapPCL>(*map);
    pcl::PointCloud<Point> far;
    far.push_back(Point(1.0e4f, 1.0e4f, 1.0e4f, 0.f, 0u, 0u, 0.f));
    copy->insert(far);
    checks[1] = (map->getCloud()->size() == before && copy->getCloud()->size() == before + 1) ? 1 : 0;
  }

  const auto source = cloud_from(src_xyz);
  const gtsam::Pose3 T1 = pose_from(poses, 0), T2 = pose_from(poses, 12), Tt = pose_from(poses, 24);
  ICPFactor::Ptr factor = binary ? std::make_shared<ICPFactor>(X(1), X(0), map, source, cfg)
                                 : std::make_shared<ICPFactor>(X(1), map, source, cfg);

  std::ofstream out(argv[2], std::ios::binary);
  write_vec(out, checks);
  const size_t N = source.size();
  for (int pass = 0; pass < 2; ++pass) {
    gtsam::Values values;
    values.insert(X(1), pass == 0 ? T1 : T2);
    if (binary) values.insert(X(0), Tt);
    values.insert(G(0), gtsam::Unit3(gz[0], gz[1], gz[2]));
    const auto gf = factor->linearize(values);
    const auto hf = std::dynamic_pointer_cast<gtsam::HessianFactor>(gf);
    if (!hf) throw std::runtime_error("pin: linearize did not return a HessianFactor");
    const gtsam::Matrix Hm = hf->information();     // G
    const gtsam::Vector gv = hf->linearTerm();      // g (= -J^T b)
    const int n = static_cast<int>(Hm.rows());
    std::vector<double> H(static_cast<size_t>(n) * n), g(n);
    for (int r = 0; r < n; ++r) {
      g[r] = gv(r);
      for (int c = 0; c < n; ++c) H[static_cast<size_t>(r) * n + c] = Hm(r, c);
    }
    write_vec(out, H);
    write_vec(out, g);
    write_vec(out, std::vector<double>{hf->constantTerm()});
    std::vector<int32_t> st(N);
    std::vector<double> means, normals;
    for (size_t i = 0; i < N; ++i) {
      st[i] = static_cast<int32_t>(factor->getStatuses()[i]);
      put3(means, factor->getCorresMeansTarget()[i]);
      put3(normals, factor->getCorresNormalsTarget()[i]);
    }
    write_vec(out, st);
    write_vec(out, means);
    write_vec(out, normals);
    V3D tc, rc, tf, rf, dr, dt;
    M33 Et, Er, Dr, Dt;
    factor->getLocalizabilities(tc, rc, tf, rf, Et, Er);
    factor->getDegenInfo(dr, Dr, dt, Dt);
    std::vector<double> loc, eig, degen;
    put3(loc, tc);
    put3(loc, rc);
    put3(loc, tf);
    put3(loc, rf);
    put33(eig, Et);
    put33(eig, Er);
    put3(degen, dr);
    put3(degen, dt);
    write_vec(out, loc);
    write_vec(out, eig);
    write_vec(out, degen);
  }

  // knn_search at the pass-1 pose (what linearize does at geometric_factor.hpp:292-302), every 8th source point
  {
    const gtsam::Pose3 delta = Tt.inverse() * T1;
    const size_t k = cfg.num_corres_points;
    std::vector<int32_t> found;
    std::vector<double> sq, pts;
    for (size_t i = 0; i < N; i += 8) {
      const V3D q = delta * V3D(source.points[i].x, source.points[i].y, source.points[i].z);
      std::vector<size_t> idx(k, 0);
      std::vector<double> d(k, 0.0);
      const bool ok = map->knn_search(q, k, idx, d);
      found.push_back(ok ? 1 : 0);
      for (size_t j = 0; j < k; ++j) {
        sq.push_back(ok ? d[j] : 0.0);
        const Eigen::Vector4d p = ok ? map->underlying()->point(idx[j]) : Eigen::Vector4d::Zero();
        pts.push_back(p.x());
        pts.push_back(p.y());
        pts.push_back(p.z());
      }
    }
    write_vec(out, found);
    write_vec(out, sq);
    write_vec(out, pts);
  }
  std::cout << "mimosa_pin: wrote " << argv[2] << "  checks abs(double) " << checks[0] << "  deep copy " << checks[1] << "\n";
  return (checks[0] && checks[1]) ? 0 : 1;
}
