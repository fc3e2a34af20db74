// CPU-only driver for mimosa_hip::lidar::computeDeskewPoses (Manager::deskewPoints' pose part,
// src/lidar/manager.cpp:455-499): reads the inputs written by tests/test_deskew_poses.py, prints the poses.
#include <cstdio>
#include <fstream>

#include "../../mimosa_amd/host/mimosa_hip/lidar.hpp"

using namespace mimosa_hip;
using namespace mimosa_hip::lidar;

template <typename T>
static std::vector<T> read_vec(std::ifstream & f)
{
  uint64_t n = 0;
  f.read(reinterpret_cast<char *>(&n), 8);
  std::vector<T> v(n);
  f.read(reinterpret_cast<char *>(v.data()), static_cast<std::streamsize>(n * sizeof(T)));
  return v;
}

int main(int argc, char ** argv)
{
  if (argc < 2) return 2;
  std::ifstream f(argv[1], std::ios::binary);
  const auto imu_t = read_vec<double>(f);
  const auto meas = read_vec<double>(f);   // per sample: acc(3) gyro(3)
  const auto navd = read_vec<double>(f);   // per sample: R(9) p(3) v(3)
  const auto misc = read_vec<double>(f);   // bias_acc(3) bias_gyro(3) g_unit(3) g_norm header_ts T_B_S R(9) t(3)
  const auto uns = read_vec<uint32_t>(f);
  std::vector<V3D> acc(imu_t.size()), gyro(imu_t.size());
  std::vector<NavState> nav(imu_t.size());
  for (size_t j = 0; j < imu_t.size(); ++j) {
    acc[j] = vector3(&meas[6 * j]);
    gyro[j] = vector3(&meas[6 * j + 3]);
    nav[j] = NavState(pose3(&navd[15 * j], &navd[15 * j + 9]), vector3(&navd[15 * j + 12]));
  }
  const Pose3 T_B_S = pose3(&misc[11], &misc[20]);
  try {
    const auto poses = computeDeskewPoses(imu_t, acc, gyro, nav, vector3(&misc[0]), vector3(&misc[3]), vector3(&misc[6]), misc[9], uns, misc[10], T_B_S);
    std::printf("[");
    for (size_t g = 0; g < poses.size(); ++g) {
      std::printf("%s[", g ? ",\n" : "");
      const PoseRM p = rowMajor(poses[g]);
      for (int i = 0; i < 9; ++i) std::printf("%.17g, ", p.R[i]);
      for (int i = 0; i < 3; ++i) std::printf("%.17g%s", p.t[i], i < 2 ? ", " : "");
      std::printf("]");
    }
    std::printf("]\n");
  } catch (const std::exception & e) {
    std::fprintf(stderr, "deskew_poses: %s\n", e.what());
    return 1;
  }
  return 0;
}
