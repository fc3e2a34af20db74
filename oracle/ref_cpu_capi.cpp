// ORACLE — TEST INFRASTRUCTURE ONLY (see ref_cpu.hpp header).  PARITY UNPINNED.
// extern "C" surface over the CPU restatement so tests/ and bench.py's cpu_baseline leg can drive
// it through ctypes.  Never linked into, or loaded by, the shipped library.
#include <chrono>
#include <cstdio>

#include "ref_cpu.hpp"

using namespace refcpu;

struct ref_map
{
  std::shared_ptr<IVox> ivox;
};
struct ref_icp
{
  std::unique_ptr<ICPFactor> f;
};

static Pose pose_from(const double * R, const double * t)
{
  Pose p;
  std::memcpy(p.R.m, R, sizeof(double) * 9);
  p.t = {t[0], t[1], t[2]};
  return p;
}

extern "C" {

int ref_abi_version() { return 1; }

// ---- map -----------------------------------------------------------------------------------
ref_map * ref_map_create(double leaf, double min_dist_in_cell, int max_pts, int neighbor_mode, int lru_horizon)
{
  auto * m = new ref_map;
  m->ivox = std::make_shared<IVox>(leaf);
  m->ivox->set_min_dist_in_cell(min_dist_in_cell);
  m->ivox->set_max_num_points_in_cell(static_cast<size_t>(max_pts));
  m->ivox->set_neighbor_voxel_mode(neighbor_mode);
  m->ivox->set_lru_horizon(static_cast<size_t>(lru_horizon));
  return m;
}
void ref_map_set_lru_clear_cycle(ref_map * m, int cycle) { m->ivox->set_lru_clear_cycle(static_cast<size_t>(cycle)); }
// Geometric::updateMap's "copy then insert" (geometric.cpp:494): shallow-per-voxel copy.
ref_map * ref_map_copy(const ref_map * o)
{
  auto * m = new ref_map;
  m->ivox = std::make_shared<IVox>(*o->ivox);
  return m;
}
void ref_map_destroy(ref_map * m) { delete m; }
void ref_map_insert(ref_map * m, const float * xyz, int64_t n) { m->ivox->insert(xyz, static_cast<size_t>(n)); }
int64_t ref_map_num_voxels(const ref_map * m) { return static_cast<int64_t>(m->ivox->num_voxels()); }
int64_t ref_map_num_points(const ref_map * m) { return static_cast<int64_t>(m->ivox->num_points()); }
// coords: 3*nv ints; counts: nv ints; xyz: 3*np floats.  Call with nullptrs to size first.
void ref_map_export(const ref_map * m, int32_t * coords, int32_t * counts, float * xyz)
{
  std::vector<int> c, k;
  std::vector<float> p;
  m->ivox->export_voxels(c, k, p);
  if (coords) std::memcpy(coords, c.data(), c.size() * sizeof(int));
  if (counts) std::memcpy(counts, k.data(), k.size() * sizeof(int));
  if (xyz) std::memcpy(xyz, p.data(), p.size() * sizeof(float));
}
// knn for n queries (fp64 xyz).  idx: n*k int64 global ids, sq: n*k, found: n, ncand: n.
void ref_map_knn(
  const ref_map * m, const double * q, int64_t n, int k, int64_t * idx, double * sq, int32_t * found,
  int32_t * ncand)
{
  for (int64_t i = 0; i < n; ++i) {
    size_t ids[64];
    double d[64];
    size_t c = 0;
    const size_t f = m->ivox->knn_search(q + 3 * i, static_cast<size_t>(k), ids, d, std::numeric_limits<double>::max(), &c);
    for (int j = 0; j < k; ++j) {
      idx[i * k + j] = static_cast<int64_t>(ids[j]);
      sq[i * k + j] = d[j];
    }
    found[i] = static_cast<int32_t>(f);
    if (ncand) ncand[i] = static_cast<int32_t>(c);
  }
}
// coordinates of a global id
void ref_map_point(const ref_map * m, int64_t id, double * xyz)
{
  const P4 & p = m->ivox->point(static_cast<size_t>(id));
  xyz[0] = p.x;
  xyz[1] = p.y;
  xyz[2] = p.z;
}

// ---- factor --------------------------------------------------------------------------------
ref_icp * ref_icp_create(int is_binary, ref_map * map, const Point32 * pts, int64_t n, const RegistrationConfig * cfg)
{
  auto * h = new ref_icp;
  h->f = std::make_unique<ICPFactor>(is_binary != 0, map->ivox, pts, static_cast<size_t>(n), *cfg);
  return h;
}
ref_icp * ref_icp_clone(const ref_icp * o)
{
  auto * h = new ref_icp;
  h->f = std::make_unique<ICPFactor>(*o->f);
  return h;
}
void ref_icp_destroy(ref_icp * h) { delete h; }
void ref_icp_set_threads(ref_icp * h, int n) { h->f->n_threads = n; }

// R/t row-major fp64.  R_tgt/t_tgt may be null (unary).  g_unit = Values[G(0)].unitVector().
void ref_icp_linearize(
  ref_icp * h, const double * R_src, const double * t_src, const double * R_tgt, const double * t_tgt,
  const double * g_unit, LinearizeResult * out)
{
  const Pose Ts = pose_from(R_src, t_src);
  Pose Tt;
  if (R_tgt && t_tgt) Tt = pose_from(R_tgt, t_tgt);
  h->f->linearize(Ts, (R_tgt && t_tgt) ? &Tt : nullptr, {g_unit[0], g_unit[1], g_unit[2]}, *out);
}
void ref_icp_get_da(const ref_icp * h, double * q_da) { std::memcpy(q_da, h->f->transed_da().data(), h->f->size() * 3 * sizeof(double)); }
void ref_icp_set_state(ref_icp * h, const int32_t * status, const double * means, const double * normals, const double * q_da, int count)
{
  h->f->set_state(status, means, normals, q_da, count);
}
void ref_icp_get_state(const ref_icp * h, int32_t * status, double * means, double * normals, double * transed)
{
  const size_t n = h->f->size();
  if (status) std::memcpy(status, h->f->statuses().data(), n * sizeof(int32_t));
  if (means) std::memcpy(means, h->f->means().data(), n * 3 * sizeof(double));
  if (normals) std::memcpy(normals, h->f->normals().data(), n * 3 * sizeof(double));
  if (transed) std::memcpy(transed, h->f->transed().data(), n * 3 * sizeof(double));
}
// Per-point whitened residual + Jacobian rows at delta pose (R,t): e[n], J[n*6], valid[n]
void ref_icp_point_rows(const ref_icp * h, const double * R, const double * t, double * e, double * J, int32_t * valid)
{
  const Pose d = pose_from(R, t);
  const size_t n = h->f->size();
  for (size_t i = 0; i < n; ++i) {
    double ei = 0, Ji[6] = {0, 0, 0, 0, 0, 0};
    valid[i] = h->f->point_row(i, d, ei, Ji) ? 1 : 0;
    e[i] = ei;
    std::memcpy(J + 6 * i, Ji, sizeof(Ji));
  }
}

// Timed cold linearizes: a fresh factor per iteration (constructor included, as in
// Geometric::getFactors geometric.cpp:194-196); returns seconds per iteration in secs[iters].
void ref_icp_time_cold(
  ref_map * map, const Point32 * pts, int64_t n, const RegistrationConfig * cfg, const double * R_src,
  const double * t_src, const double * g_unit, int n_threads, int iters, double * secs, LinearizeResult * last)
{
  const Pose Ts = pose_from(R_src, t_src);
  for (int it = 0; it < iters; ++it) {
    const auto t0 = std::chrono::steady_clock::now();
    ICPFactor f(false, map->ivox, pts, static_cast<size_t>(n), *cfg);
    f.n_threads = n_threads;
    LinearizeResult r;
    f.linearize(Ts, nullptr, {g_unit[0], g_unit[1], g_unit[2]}, r);
    const auto t1 = std::chrono::steady_clock::now();
    secs[it] = std::chrono::duration<double>(t1 - t0).count();
    if (last) *last = r;
  }
}

// ---- small kernels -------------------------------------------------------------------------
int ref_eigen3(const double * A, double * evals, double * evecs)
{
  M3 a, v;
  std::memcpy(a.m, A, sizeof(double) * 9);
  V3 e;
  const bool ok = self_adjoint_eigen3(a, e, v);
  evals[0] = e.x;
  evals[1] = e.y;
  evals[2] = e.z;
  std::memcpy(evecs, v.m, sizeof(double) * 9);
  return ok ? 1 : 0;
}
void ref_deskew(Point32 * pts, int64_t n, const uint32_t * unique_ns, const float * Rt12, int64_t n_groups)
{
  deskew(pts, static_cast<size_t>(n), unique_ns, Rt12, static_cast<size_t>(n_groups));
}
void ref_transform_f32(Point32 * pts, int64_t n, const float * R, const float * t)
{
  for (int64_t i = 0; i < n; ++i) transform_f32(R, t, pts[i].x, pts[i].y, pts[i].z);
}
// Manager::prepareInput.  Outputs sized by the caller for n points; returns n_full.  counts[0..3] = n_full,
// n_geometric, n_unique_ns, last_point_ns.  group_offsets has n_unique_ns + 1 entries into group_idxs.
int64_t ref_prepare_input(
  const PointOusterIn * in, int64_t n, const InputConfig * cfg, Point32 * points_full, uint64_t * geometric_idxs,
  uint32_t * unique_ns, uint64_t * group_offsets, uint64_t * group_idxs, uint64_t * counts)
{
  PreparedInput o;
  prepare_input(in, static_cast<size_t>(n), *cfg, o);
  std::memcpy(points_full, o.points_full.data(), o.points_full.size() * sizeof(Point32));
  std::memcpy(geometric_idxs, o.geometric_point_idxs.data(), o.geometric_point_idxs.size() * sizeof(uint64_t));
  std::memcpy(unique_ns, o.unique_ns.data(), o.unique_ns.size() * sizeof(uint32_t));
  uint64_t off = 0;
  for (size_t g = 0; g < o.idxs_at_unique_ns.size(); ++g) {
    group_offsets[g] = off;
    for (const uint64_t j : o.idxs_at_unique_ns[g]) group_idxs[off++] = j;
  }
  group_offsets[o.idxs_at_unique_ns.size()] = off;
  counts[0] = o.points_full.size();
  counts[1] = o.geometric_point_idxs.size();
  counts[2] = o.unique_ns.size();
  counts[3] = o.last_point_ns;
  return static_cast<int64_t>(o.points_full.size());
}
// Manager::prepareInput<PointT> for the other point types.  kind: 0 Ouster, 1 OusterOdyssey, 2 OusterR8, 3 Hesai, 4 Livox,
// 5 LivoxFromCustom2, 6 Velodyne, 7 VelodyneAnybotics, 8 Rslidar.  ref_point_sizeof(kind) = sizeof of the restated struct.
int64_t ref_point_sizeof(int kind)
{
  switch (kind) {
    case 0: return sizeof(PointOusterIn);
    case 1: return sizeof(PointOusterOdysseyIn);
    case 2: return sizeof(PointOusterR8In);
    case 3: return sizeof(PointHesaiIn);
    case 4: return sizeof(PointLivoxIn);
    case 5: return sizeof(PointLivoxFromCustom2In);
    case 6: return sizeof(PointVelodyneIn);
    case 7: return sizeof(PointVelodyneAnyboticsIn);
    case 8: return sizeof(PointRslidarIn);
  }
  return -1;
}
int64_t ref_prepare_input_typed(int kind, const void * in, int64_t n, uint32_t width, uint32_t height, int transpose, int organize,
                                double header_ts, const InputConfig * cfg, Point32 * points_full, uint64_t * geometric_idxs,
                                uint32_t * unique_ns, uint64_t * counts)
{
  PreparedInput o;
  InputOrder ord;
  ord.width = width;
  ord.height = height;
  ord.transpose_pointcloud = transpose != 0;
  ord.organize_pointcloud_by_ring = organize != 0;
  ord.header_ts = header_ts;
  const size_t m = static_cast<size_t>(n);
  switch (kind) {
    case 0: prepare_input_typed(static_cast<const PointOusterIn *>(in), m, *cfg, ord, o); break;
    case 1: prepare_input_typed(static_cast<const PointOusterOdysseyIn *>(in), m, *cfg, ord, o); break;
    case 2: prepare_input_typed(static_cast<const PointOusterR8In *>(in), m, *cfg, ord, o); break;
    case 3: prepare_input_typed(static_cast<const PointHesaiIn *>(in), m, *cfg, ord, o); break;
    case 4: prepare_input_typed(static_cast<const PointLivoxIn *>(in), m, *cfg, ord, o); break;
    case 5: prepare_input_typed(static_cast<const PointLivoxFromCustom2In *>(in), m, *cfg, ord, o); break;
    case 6: prepare_input_typed(static_cast<const PointVelodyneIn *>(in), m, *cfg, ord, o); break;
    case 7: prepare_input_typed(static_cast<const PointVelodyneAnyboticsIn *>(in), m, *cfg, ord, o); break;
    case 8: prepare_input_typed(static_cast<const PointRslidarIn *>(in), m, *cfg, ord, o); break;
    default: return -1;
  }
  std::memcpy(points_full, o.points_full.data(), o.points_full.size() * sizeof(Point32));
  std::memcpy(geometric_idxs, o.geometric_point_idxs.data(), o.geometric_point_idxs.size() * sizeof(uint64_t));
  std::memcpy(unique_ns, o.unique_ns.data(), o.unique_ns.size() * sizeof(uint32_t));
  counts[0] = o.points_full.size();
  counts[1] = o.geometric_point_idxs.size();
  counts[2] = o.unique_ns.size();
  counts[3] = o.last_point_ns;
  return static_cast<int64_t>(o.points_full.size());
}
int64_t ref_downsample(
  const Point32 * pts, int64_t n, double leaf, int max_pts, double min_dist, uint32_t * kept)
{
  std::vector<uint32_t> k;
  downsample(pts, static_cast<size_t>(n), leaf, static_cast<size_t>(max_pts), min_dist, k);
  std::memcpy(kept, k.data(), k.size() * sizeof(uint32_t));
  return static_cast<int64_t>(k.size());
}
int ref_projection_matrix(const double * loc, double thresh, const double * evecs, double * P, double * axes)
{
  M3 E, p;
  std::memcpy(E.m, evecs, sizeof(double) * 9);
  V3 ax;
  const bool d = get_projection_matrix({loc[0], loc[1], loc[2]}, thresh, E, p, ax);
  std::memcpy(P, p.m, sizeof(double) * 9);
  axes[0] = ax.x;
  axes[1] = ax.y;
  axes[2] = ax.z;
  return d ? 1 : 0;
}
int ref_max_threads()
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
}  // extern "C"
