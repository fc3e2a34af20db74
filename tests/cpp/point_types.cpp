// Driver for ScanFrontEnd::prepareInput<PointT> (the host mirror of Manager::prepareInput for every sensor point type):
// reads {kind, width, height, flags, header_ts, config, records} written by tests/test_gpu_host_cpp.py, writes
// n_full n_geometric last_point_ns + points_full_ + unique_ns_ as raw bytes to argv[2].
#include <cstdio>
#include <fstream>

#include "../../mimosa_amd/host/mimosa_hip/lidar.hpp"

using namespace mimosa_hip;
using namespace mimosa_hip::lidar;

struct Header
{
  int32_t kind;
  uint32_t width, height;
  int32_t transpose, organize;
  uint32_t pad;
  double header_ts;
  uint64_t n;
  ManagerInputConfig cfg;
};

template <typename PointT>
static void run(ScanFrontEnd & fe, const Header & h, const std::vector<char> & bytes)
{
  static_assert(sizeof(PointT) % 16 == 0, "EIGEN_ALIGN16");
  CloudOrder o;
  o.width = h.width;
  o.height = h.height;
  o.transpose_pointcloud = h.transpose != 0;
  o.organize_pointcloud_by_ring = h.organize != 0;
  if (bytes.size() != h.n * sizeof(PointT)) throw std::runtime_error("record size mismatch");
  fe.prepareInput(reinterpret_cast<const PointT *>(bytes.data()), h.n, h.cfg, h.header_ts, o);
}

int main(int argc, char ** argv)
{
  if (argc < 3) return 2;
  std::ifstream f(argv[1], std::ios::binary);
  Header h{};
  f.read(reinterpret_cast<char *>(&h), sizeof(h));
  uint64_t nbytes = 0;
  f.read(reinterpret_cast<char *>(&nbytes), 8);
  std::vector<char> bytes(nbytes);
  f.read(bytes.data(), static_cast<std::streamsize>(nbytes));
  try {
    auto ctx = std::make_shared<Context>(0);
    ScanFrontEnd fe(ctx);
    switch (h.kind) {
      case 0: run<PointOuster>(fe, h, bytes); break;
      case 1: run<PointOusterOdyssey>(fe, h, bytes); break;
      case 2: run<PointOusterR8>(fe, h, bytes); break;
      case 3: run<PointHesai>(fe, h, bytes); break;
      case 4: run<PointLivox>(fe, h, bytes); break;
      case 5: run<PointLivoxFromCustom2>(fe, h, bytes); break;
      case 6: run<PointVelodyne>(fe, h, bytes); break;
      case 7: run<PointVelodyneAnybotics>(fe, h, bytes); break;
      case 8: run<PointRslidar>(fe, h, bytes); break;
      default: return 2;
    }
    const PointCloud full = fe.download(0);
    std::ofstream o(argv[2], std::ios::binary);
    const uint64_t head[4] = {fe.info().n_full, fe.info().n_geometric, fe.info().last_point_ns, fe.uniqueNs().size()};
    o.write(reinterpret_cast<const char *>(head), sizeof(head));
    o.write(reinterpret_cast<const char *>(full.data()), static_cast<std::streamsize>(full.size() * sizeof(full[0])));
    o.write(reinterpret_cast<const char *>(fe.uniqueNs().data()), static_cast<std::streamsize>(fe.uniqueNs().size() * 4));
    std::printf("%.9f\n", fe.correctedTs());
  } catch (const std::exception & e) {
    std::fprintf(stderr, "point_types: %s\n", e.what());
    return 1;
  }
  return 0;
}
