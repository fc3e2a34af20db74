// Type-check of lidar::Manager (mimosa_amd/host/mimosa_hip/manager.hpp) for every sensor point type the reference's callback
// dispatches on (src/lidar/manager.cpp:55-86): Manager::callback<PointT> is instantiated for all nine; one of them is run on
// an empty cloud when a device is present.  Compiled by tests/test_gpu_host_cpp.py::test_host_layer_compiles.
#include <cstdio>

#include "../../mimosa_amd/host/mimosa_hip/manager.hpp"

using namespace mimosa_hip;
using namespace mimosa_hip::lidar;

struct NoGraph : graph::ManagerInterface
{
  void getStateUpto(const double, State &) override {}
  graph::DeclarationResult declare(const double, size_t & k, const bool) override
  {
    k = 0;
    return graph::DeclarationResult::FAILURE_CANNOT_INIT_ON_MODALITY;  // every message is skipped after prepareInput
  }
  Pose3 getPoseAt(const size_t) override { return Pose3(); }
  Values getCurrentOptimizedValues() override { return Values(); }
  void define(const NonlinearFactorGraph &, Values &, const graph::DeclarationResult) override {}
};
struct NoImu : imu::ManagerInterface
{
  void getInterpolatedMeasurements(const double, const double, ImuBuffer &, const bool) override {}
  double gravityNorm() const override { return 9.81; }
  void resetIntegrationAndSetBias(const State &) override {}
  void integrateMeasurement(const V3D &, const V3D &, const double) override {}
  gtsam::NavState predict(const State & s) override { return s.navState(); }
};

template <typename PointT>
static void one(Manager & m, int n)
{
  std::vector<PointT> cloud(static_cast<size_t>(n));
  CloudOrder order;
  order.width = static_cast<uint32_t>(n);
  m.callback(cloud.data(), cloud.size(), 100.0, order);
}

int main(int argc, char **)
{
  try {
    auto ctx = std::make_shared<Context>(0);
    NoGraph g;
    NoImu i;
    PhotometricConfig pc;
    pc.enabled = false;
    Manager m(ctx, ManagerConfig(), GeometricConfig(), pc, g, i);
    const int n = argc > 1 ? 64 : 0;
    one<PointOuster>(m, n);
    one<PointOusterOdyssey>(m, n);
    one<PointOusterR8>(m, n);
    one<PointHesai>(m, n);
    one<PointLivox>(m, n);
    one<PointLivoxFromCustom2>(m, n);
    one<PointVelodyne>(m, n);
    one<PointVelodyneAnybotics>(m, n);
    one<PointRslidar>(m, n);
    std::printf("{\"ok\": 1, \"last_key\": %zu}\n", m.lastKey());
  } catch (const std::exception & e) {
    std::fprintf(stderr, "manager_types: %s\n", e.what());
    return 1;
  }
  return 0;
}
