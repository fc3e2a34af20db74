// C ABI of the target map (include/mimosa_hip.h: mh_map_*): IncrementalVoxelMapPCL / gtsam_points::iVox, resident on
// and maintained by the device.
//
// Reference: include/mimosa/lidar/incremental_voxel_map.hpp:22-54, src/lidar/incremental_voxel_map.cpp:14-62,
// Geometric::updateMap src/lidar/geometric.cpp:483-495 (f32 world transform, copy-then-insert).
// The host side is bookkeeping: capacities, the three small read-backs an insert needs (new voxels, new blocks, totals),
// the LRU cadence (every lru_clear_cycle-th insert).  Kernels: map_kernels.hip.  No CPU fallback.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>

#include "mh_internal.hpp"

namespace
{

mh::MapArrays arrays_of(const mh_map * m)
{
  mh::MapArrays a;
  a.table = static_cast<int4 *>(m->d_table.p);
  a.table_mask = static_cast<uint32_t>(m->table_cap - 1);
  a.cells = static_cast<uint32_t *>(m->d_cells.p);
  a.buckets = static_cast<float4 *>(m->d_buckets.p);
  a.qbuckets = static_cast<uint32_t *>(m->d_qbuckets.p);
  a.vox = static_cast<int4 *>(m->d_vox.p);
  a.lru = static_cast<unsigned long long *>(m->d_lru.p);
  a.state = m->d_state;
  return a;
}

int fetch_state(mh_map * m)
{
  mh_ctx * ctx = m->ctx;
  MH_HIP(ctx, hipMemcpyAsync(m->h_state, m->d_state, sizeof(mh::MapState), hipMemcpyDeviceToHost, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return MH_OK;
}
int push_state(mh_map * m)
{
  mh_ctx * ctx = m->ctx;
  m->h_state->n_voxels = m->n_voxels;
  m->h_state->n_blocks = m->n_blocks;
  m->h_state->n_points = m->n_points;
  m->h_state->bad_coord = 0;
  MH_HIP(ctx, hipMemcpyAsync(m->d_state, m->h_state, sizeof(mh::MapState), hipMemcpyHostToDevice, ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // h_state is reused as the read-back target
  return MH_OK;
}

size_t grow(size_t have, size_t need) { return need <= have ? have : need + need / 2 + 64; }

// voxel arrays for at least n voxels (+ one bucket of slack: the k-NN kernels' branch-free loads may touch "slot 31 of
// the last voxel")
int ensure_voxels(mh_map * m, size_t n)
{
  if (n + 2 <= m->vox_cap) return MH_OK;
  mh_ctx * ctx = m->ctx;
  const size_t cap = grow(m->vox_cap, n + 2);
  MH_HIP(ctx, m->d_buckets.reserve(cap * mh::kBucketStride * sizeof(float4), ctx->stream, true));
  MH_HIP(ctx, m->d_qbuckets.reserve(cap * mh::kBucketStride * sizeof(uint32_t), ctx->stream, true));
  MH_HIP(ctx, m->d_vox.reserve(cap * sizeof(int4), ctx->stream, true));
  MH_HIP(ctx, m->d_lru.reserve(cap * sizeof(unsigned long long), ctx->stream, true));
  m->vox_cap = cap;
  return MH_OK;
}
// cell tables for at least n blocks; new tables start empty
int ensure_blocks(mh_map * m, size_t n, size_t n_initialised)
{
  mh_ctx * ctx = m->ctx;
  if (n + 1 > m->block_cap) {
    const size_t cap = grow(m->block_cap, n + 1);
    MH_HIP(ctx, m->d_cells.reserve(cap * mh::kCellsPerBlock * sizeof(uint32_t), ctx->stream, true));
    m->block_cap = cap;
  }
  static_assert(mh::kEmptyCell == 0u, "the byte pattern of the fill below");
  if (n > n_initialised)
    MH_HIP(ctx, hipMemsetAsync(static_cast<uint32_t *>(m->d_cells.p) + n_initialised * mh::kCellsPerBlock, 0x00,
                               (n - n_initialised) * mh::kCellsPerBlock * sizeof(uint32_t), ctx->stream));
  return MH_OK;
}
// hash table with load <= 0.5 for up to n_blocks_bound blocks
int ensure_table(mh_map * m, size_t n_blocks_bound)
{
  if (n_blocks_bound * 2 <= m->table_cap) return MH_OK;
  mh_ctx * ctx = m->ctx;
  size_t cap = m->table_cap ? m->table_cap : 1024;
  while (cap < n_blocks_bound * 2) cap *= 2;
  DevBuf nt;
  MH_HIP(ctx, nt.reserve(cap * sizeof(int4), ctx->stream, false));
  MH_HIP(ctx, hipMemsetAsync(nt.p, 0xFF, cap * sizeof(int4), ctx->stream));
  if (m->table_cap)
    MH_HIP(ctx, mh::launch_map_rehash(static_cast<const int4 *>(m->d_table.p), static_cast<uint32_t>(m->table_cap), static_cast<int4 *>(nt.p),
                                      static_cast<uint32_t>(cap), ctx->stream));
  MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
  m->d_table.release();
  m->d_table = nt;
  nt.p = nullptr;
  nt.cap = 0;
  m->table_cap = cap;
  return MH_OK;
}

int ensure_scratch(mh_map * m, size_t n)
{
  mh_ctx * ctx = m->ctx;
  const size_t k = n ? n : 1;
  MH_HIP(ctx, m->s_pts.reserve(k * sizeof(float4), ctx->stream, false));
  MH_HIP(ctx, m->s_group.reserve(mh::map_group_bytes(k), ctx->stream, false));
  MH_HIP(ctx, m->s_seg_vid.reserve(k * sizeof(uint32_t), ctx->stream, false));
  MH_HIP(ctx, m->s_seg_added.reserve(k * sizeof(uint32_t), ctx->stream, false));
  MH_HIP(ctx, m->s_blk_new.reserve(((k + 255) / 256 + 1) * sizeof(uint32_t), ctx->stream, false));
  return MH_OK;
}
mh::InsertScratch scratch_of(const mh_map * m)
{
  mh::InsertScratch s;
  s.pts = static_cast<float4 *>(m->s_pts.p);
  s.group = m->s_group.p;
  s.seg_vid = static_cast<uint32_t *>(m->s_seg_vid.p);
  s.seg_added = static_cast<uint32_t *>(m->s_seg_added.p);
  s.blk_new = static_cast<uint32_t *>(m->s_blk_new.p);
  return s;
}

// iVox's LRU purge: voxels with lru + horizon < counter are erased, the survivors keep their
// order; tables are rebuilt.  Everything stays on the device: the survivors are compacted into fresh arrays.
int purge_lru(mh_map * m)
{
  mh_ctx * ctx = m->ctx;
  if (m->n_voxels == 0) return MH_OK;
  const uint32_t nv = m->n_voxels;
  MH_HIP(ctx, m->s_flags.reserve(nv * sizeof(uint32_t), ctx->stream, false));
  MH_HIP(ctx, m->s_pos.reserve(nv * sizeof(uint32_t), ctx->stream, false));
  MH_HIP(ctx, m->s_temp.reserve(mh::map_temp_bytes(nv), ctx->stream, false));
  mh::MapArrays a = arrays_of(m);
  MH_HIP(ctx, mh::launch_map_purge_flags(a, nv, static_cast<unsigned long long>(m->cfg.lru_horizon), m->lru_counter,
                                         static_cast<uint32_t *>(m->s_flags.p), static_cast<uint32_t *>(m->s_pos.p), m->s_temp.p, m->s_temp.cap,
                                         ctx->stream));
  int rc = fetch_state(m);
  if (rc != MH_OK) return rc;
  const uint32_t keep = m->h_state->n_keep;
  if (keep == nv) return MH_OK;
  // fresh voxel arrays; the old ones are the compaction source
  DevBuf ob = m->d_buckets, oq = m->d_qbuckets, ov = m->d_vox, ol = m->d_lru;
  m->d_buckets = DevBuf{};
  m->d_qbuckets = DevBuf{};
  m->d_vox = DevBuf{};
  m->d_lru = DevBuf{};
  const size_t old_cap = m->vox_cap;
  m->vox_cap = 0;
  auto restore = [&]() {
    m->d_buckets.release();
    m->d_qbuckets.release();
    m->d_vox.release();
    m->d_lru.release();
    m->d_buckets = ob;
    m->d_qbuckets = oq;
    m->d_vox = ov;
    m->d_lru = ol;
    m->vox_cap = old_cap;
  };
  rc = ensure_voxels(m, keep);
  if (rc != MH_OK) {
    restore();
    return rc;
  }
  mh::MapArrays src = a, dst = arrays_of(m);
  src.buckets = static_cast<float4 *>(ob.p);
  src.qbuckets = static_cast<uint32_t *>(oq.p);
  src.vox = static_cast<int4 *>(ov.p);
  src.lru = static_cast<unsigned long long *>(ol.p);
  m->n_voxels = keep;
  m->n_blocks = 0;
  m->n_points = 0;
  rc = push_state(m);
  hipError_t e = hipSuccess;
  if (rc == MH_OK) e = mh::launch_map_purge_compact(src, dst, nv, static_cast<const uint32_t *>(m->s_flags.p), static_cast<const uint32_t *>(m->s_pos.p), ctx->stream);
  if (rc == MH_OK && e == hipSuccess) e = hipMemsetAsync(m->d_table.p, 0xFF, m->table_cap * sizeof(int4), ctx->stream);
  if (rc == MH_OK && e == hipSuccess) e = mh::launch_map_claim_blocks(dst, 0, keep, ctx->stream);
  if (rc == MH_OK && e == hipSuccess) rc = fetch_state(m);
  if (rc != MH_OK || e != hipSuccess) {
    // the fresh arrays and the table are half written and the counters already name the compacted map: there is no way back
    // to the old state from here.  The map is marked unusable instead of being left to answer from garbage ids.
    (void)hipStreamSynchronize(ctx->stream);
    ob.release();
    oq.release();
    ov.release();
    ol.release();
    m->poisoned = true;
    return rc != MH_OK ? rc : hip_fail(ctx, e, "mh_map_insert: LRU purge");
  }
  m->n_blocks = m->h_state->n_blocks;
  m->n_points = m->h_state->n_points;
  ob.release();
  oq.release();
  ov.release();
  ol.release();
  rc = ensure_blocks(m, m->n_blocks, 0);
  if (rc != MH_OK) {
    m->poisoned = true;  // voxels without their cell words
    return rc;
  }
  dst = arrays_of(m);
  {
    hipError_t e2 = mh::launch_map_write_words(dst, 0, keep, ctx->stream);
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);
    if (e2 != hipSuccess) {
      m->poisoned = true;
      return hip_fail(ctx, e2, "mh_map_insert: LRU purge (cell words)");
    }
  }
  m->purges++;
  return MH_OK;
}

// iVox::insert on a batch already on the device: n points `stride` floats apart, optional f32 transform (12 floats on
// the device).  Synchronous: the map is current when this returns.
int insert_device(mh_map * m, const float * d_src, size_t n, size_t stride, const float * d_Rt12)
{
  mh_ctx * ctx = m->ctx;
  if (n > 0x3fffffffu) return fail(ctx, MH_ERR_UNSUPPORTED, "mh_map_insert: batch too large");
  if (m->poisoned) return fail(ctx, MH_ERR_HIP, "mh_map_insert: the map is inconsistent after a failed mutation");
  // Factors of other contexts (HIP streams) may be reading this map: the arrays are modified in place and may be
  // reallocated, so the streams of the contexts that hold factors on it are waited for first (none in the usual
  // copy-then-insert: the fresh copy has no factor yet).
  MH_HIP(ctx, map_wait_readers(m));
  if (n) {
    int rc = ensure_scratch(m, n);
    if (rc != MH_OK) return rc;
    const mh::InsertScratch s = scratch_of(m);
    mh::MapArrays a = arrays_of(m);
    MH_HIP(ctx, mh::launch_map_insert_prepare(a, d_src, static_cast<uint32_t>(n), static_cast<uint32_t>(stride), d_Rt12, m->inv_leaf, s, ctx->stream));
    rc = fetch_state(m);
    if (rc != MH_OK) return rc;
    if (m->h_state->bad_coord) {
      (void)push_state(m);  // clears the flag; nothing was modified yet
      return fail(ctx, MH_ERR_UNSUPPORTED, "mh_map_insert: a point is NaN or its voxel coordinate exceeds +-2^20");
    }
    const uint32_t n_new = m->h_state->n_new_voxels, before = m->n_voxels;
    if (static_cast<uint64_t>(before) + n_new >= (1u << 27)) return fail(ctx, MH_ERR_UNSUPPORTED, "mh_map_insert: more than 2^27 voxels");
    rc = ensure_voxels(m, static_cast<size_t>(before) + n_new);
    if (rc == MH_OK && n_new) rc = ensure_table(m, static_cast<size_t>(m->n_blocks) + 8 * static_cast<size_t>(n_new));
    if (rc != MH_OK) return rc;
    a = arrays_of(m);
    if (n_new) {
      MH_HIP(ctx, mh::launch_map_create_voxels(a, static_cast<uint32_t>(n), before, m->lru_counter, s, ctx->stream));
      rc = fetch_state(m);
      if (rc != MH_OK) return rc;
      const uint32_t nb = m->h_state->n_blocks;
      rc = ensure_blocks(m, nb, m->n_blocks);
      if (rc != MH_OK) return rc;
      m->n_blocks = nb;
      a = arrays_of(m);
    }
    MH_HIP(ctx, mh::launch_map_insert_points(a, static_cast<uint32_t>(n), before + n_new, static_cast<uint32_t>(m->cfg.max_points_in_cell), m->min_sq,
                                             m->inv_leaf, m->lru_counter, s, ctx->stream));
    rc = fetch_state(m);
    if (rc != MH_OK) return rc;
    m->n_voxels = m->h_state->n_voxels;
    m->n_points = m->h_state->n_points;
  }
  m->inserts++;
  // ++lru_counter; every lru_clear_cycle inserts the stale voxels go
  if ((++m->lru_counter) % static_cast<uint64_t>(m->cfg.lru_clear_cycle) == 0) return purge_lru(m);
  return MH_OK;
}

void map_free(mh_map * m)
{
  for (DevBuf * b : {&m->d_table, &m->d_cells, &m->d_buckets, &m->d_qbuckets, &m->d_vox, &m->d_lru, &m->s_in, &m->s_pts, &m->s_group, &m->s_seg_vid,
                     &m->s_seg_added, &m->s_blk_new, &m->s_flags, &m->s_pos, &m->s_temp, &m->s_rt, &m->s_shard})
    b->release();
  if (m->d_state) dev_free(m->d_state);
  if (m->h_state) (void)hipHostFree(m->h_state);
  if (m->h_in) (void)hipHostFree(m->h_in);
  delete m;
}

int map_alloc(mh_ctx * ctx, const mh_map_config & cfg, mh_map ** out)
{
  mh_map * m = new mh_map;
  m->ctx = ctx;
  m->cfg = cfg;
  m->inv_leaf = 1.0 / cfg.leaf_size;
  m->min_sq = cfg.min_dist_in_cell * cfg.min_dist_in_cell;
  m->n_off = mh::neighbor_offsets(cfg.neighbor_voxel_mode, m->off);
  hipError_t e = dev_alloc(&m->d_state, sizeof(mh::MapState));
  if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&m->h_state), sizeof(mh::MapState), hipHostMallocDefault);
  if (e != hipSuccess) {
    map_free(m);
    return hip_fail(ctx, e, "mh_map_create");
  }
  std::memset(m->h_state, 0, sizeof(mh::MapState));
  *out = m;
  return MH_OK;
}
}  // namespace

extern "C" {

int mh_map_create(mh_ctx * ctx, const mh_map_config * cfg, mh_map ** out)
{
  if (!ctx || !cfg || !out) return fail(ctx, MH_ERR_INVALID_ARG, "mh_map_create: NULL argument");
  *out = nullptr;
  return guarded(ctx, "mh_map_create", [&]() -> int {
    if (!(cfg->leaf_size > 0)) return fail(ctx, MH_ERR_INVALID_ARG, "mh_map_create: leaf_size must be > 0");
    if (cfg->max_points_in_cell < 1 || cfg->max_points_in_cell > mh::kBucketStride)
      return fail(ctx, MH_ERR_UNSUPPORTED, "mh_map_create: max_points_in_cell must be in 1..20");
    const int md = cfg->neighbor_voxel_mode;
    if (md != 1 && md != 7 && md != 19 && md != 27) return fail(ctx, MH_ERR_INVALID_ARG, "mh_map_create: neighbor_voxel_mode must be 1, 7, 19 or 27");
    if (cfg->lru_clear_cycle < 1) return fail(ctx, MH_ERR_INVALID_ARG, "mh_map_create: lru_clear_cycle must be >= 1");
    if (cfg->lru_horizon < 0) return fail(ctx, MH_ERR_INVALID_ARG, "mh_map_create: lru_horizon must be >= 0");
    MH_HIP(ctx, mh_enter(ctx));
    mh_map * m = nullptr;
    int rc = map_alloc(ctx, *cfg, &m);
    if (rc != MH_OK) return rc;
    rc = ensure_table(m, 256);
    if (rc == MH_OK) rc = ensure_voxels(m, 64);
    if (rc == MH_OK) rc = ensure_blocks(m, 1, 0);  // block 0's table must exist: absent-block lookups read it
    if (rc == MH_OK) rc = push_state(m);
    if (rc != MH_OK) {
      map_free(m);
      return rc;
    }
    MH_HIP(ctx, hipMemsetAsync(m->d_buckets.p, 0, m->d_buckets.cap, ctx->stream));
    MH_HIP(ctx, hipMemsetAsync(m->d_qbuckets.p, 0, m->d_qbuckets.cap, ctx->stream));
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = m;
    return MH_OK;
  });
}

// the batch goes to the device through pinned staging as packed xyz (a pageable strided source would copy at a few GB/s)
static int stage_batch(mh_map * map, const float * xyz, size_t n, size_t stride_floats)
{
  mh_ctx * ctx = map->ctx;
  if (n > (size_t(1) << 40) / 12) return fail(ctx, MH_ERR_OOM, "mh_map_insert: batch too large");
  const size_t bytes = n * 3 * sizeof(float);
  if (map->h_in_cap < bytes) {
    if (map->h_in) (void)hipHostFree(map->h_in);
    map->h_in = nullptr;
    map->h_in_cap = 0;
    MH_HIP(ctx, hipHostMalloc(&map->h_in, bytes + bytes / 2, hipHostMallocDefault));
    map->h_in_cap = bytes + bytes / 2;
  }
  float * st = static_cast<float *>(map->h_in);
  if (stride_floats == 3) {
    std::memcpy(st, xyz, bytes);
  } else {
    for (size_t i = 0; i < n; ++i) {
      st[3 * i] = xyz[i * stride_floats];
      st[3 * i + 1] = xyz[i * stride_floats + 1];
      st[3 * i + 2] = xyz[i * stride_floats + 2];
    }
  }
  MH_HIP(ctx, map->s_in.reserve(bytes, ctx->stream, false));
  MH_HIP(ctx, hipMemcpyAsync(map->s_in.p, st, bytes, hipMemcpyHostToDevice, ctx->stream));
  map->upload_bytes += static_cast<int64_t>(bytes);
  return MH_OK;
}

int mh_map_insert(mh_map * map, const float * xyz, size_t n, size_t stride_floats)
{
  if (!map || (!xyz && n)) return fail(map ? map->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_map_insert: NULL argument");
  mh_ctx * ctx = map->ctx;
  return guarded(ctx, "mh_map_insert", [&]() -> int {
    if (stride_floats < 3) return fail(ctx, MH_ERR_INVALID_ARG, "mh_map_insert: stride_floats must be >= 3");
    MH_HIP(ctx, mh_enter(ctx));
    if (n) {
      const int rc = stage_batch(map, xyz, n, stride_floats);
      if (rc != MH_OK) return rc;
    }
    return insert_device(map, static_cast<const float *>(map->s_in.p), n, 3, nullptr);
  });
}

/* One rank's share of an insert into a map that is sharded by spatial hash: of the batch every rank
 * receives, this rank keeps the points whose voxel lies in one of its shard blocks or in their one-voxel halo, in
 * the original order (iVox's insertion rule is per voxel), and inserts those. */
int mh_map_insert_shard(mh_map * map, const float * xyz, size_t n, size_t stride_floats, int world, int rank, int block_log2)
{
  if (!map || (!xyz && n)) return fail(map ? map->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_map_insert_shard: NULL argument");
  mh_ctx * ctx = map->ctx;
  return guarded(ctx, "mh_map_insert_shard", [&]() -> int {
    if (stride_floats < 3) return fail(ctx, MH_ERR_INVALID_ARG, "mh_map_insert_shard: stride_floats must be >= 3");
    if (world < 1 || world > 64 || rank < 0 || rank >= world || block_log2 < 0 || block_log2 > 10)
      return fail(ctx, MH_ERR_INVALID_ARG, "mh_map_insert_shard: world in 1..64, 0 <= rank < world, block_log2 in 0..10");
    MH_HIP(ctx, mh_enter(ctx));
    size_t kept = 0;
    if (n) {
      if (n > 0x3fffffffu) return fail(ctx, MH_ERR_UNSUPPORTED, "mh_map_insert_shard: batch too large");
      int rc = stage_batch(map, xyz, n, stride_floats);
      if (rc != MH_OK) return rc;
      MH_HIP(ctx, map->s_flags.reserve(n * sizeof(uint32_t), ctx->stream, false));
      MH_HIP(ctx, map->s_pos.reserve(n * sizeof(uint32_t), ctx->stream, false));
      MH_HIP(ctx, map->s_temp.reserve(mh::shard_temp_bytes(n) > mh::map_temp_bytes(n) ? mh::shard_temp_bytes(n) : mh::map_temp_bytes(n), ctx->stream, false));
      MH_HIP(ctx, map->s_shard.reserve(n * 3 * sizeof(float), ctx->stream, false));
      MH_HIP(ctx, mh::launch_shard_filter(static_cast<const float *>(map->s_in.p), static_cast<uint32_t>(n), 3, map->inv_leaf, static_cast<uint32_t>(world),
                                          static_cast<uint32_t>(rank), block_log2, static_cast<uint32_t *>(map->s_flags.p),
                                          static_cast<uint32_t *>(map->s_pos.p), static_cast<float *>(map->s_shard.p), &map->d_state->n_keep, map->s_temp.p,
                                          map->s_temp.cap, ctx->stream));
      rc = fetch_state(map);
      if (rc != MH_OK) return rc;
      kept = map->h_state->n_keep;
    }
    return insert_device(map, static_cast<const float *>(map->s_shard.p), kept, 3, nullptr);
  });
}

// This rank's share of Geometric::updateMap's insert of a device-resident scan: the f32 world transform of the body cloud
// (mh_transform_f32's kernel, the arithmetic of geometric.cpp:483-490), the shard filter and the greedy insert, all on the
// device — what mh_scan_get_points + mh_transform_f32 + mh_map_insert_shard do through the host, without the three copies.
int mh_map_insert_shard_from_scan(mh_map * map, const mh_scan * scan, const float R_W_Be[9], const float t_W_Be[3], int world, int rank, int block_log2)
{
  if (!map || !scan || !R_W_Be || !t_W_Be) return fail(map ? map->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_map_insert_shard_from_scan: NULL argument");
  mh_ctx * ctx = map->ctx;
  return guarded(ctx, "mh_map_insert_shard_from_scan", [&]() -> int {
    if (!scan->preprocessed) return fail(ctx, MH_ERR_INVALID_ARG, "mh_map_insert_shard_from_scan: no mh_scan_preprocess_geometric before");
    if (scan->ctx->device != ctx->device) return fail(ctx, MH_ERR_INVALID_ARG, "mh_map_insert_shard_from_scan: scan lives on another device");
    if (world < 1 || world > 64 || rank < 0 || rank >= world || block_log2 < 0 || block_log2 > 10)
      return fail(ctx, MH_ERR_INVALID_ARG, "mh_map_insert_shard_from_scan: world in 1..64, 0 <= rank < world, block_log2 in 0..10");
    MH_HIP(ctx, mh_enter(ctx));
    const size_t n = scan->n_body;
    size_t kept = 0;
    if (n) {
      if (n > 0x3fffffffu) return fail(ctx, MH_ERR_UNSUPPORTED, "mh_map_insert_shard_from_scan: batch too large");
      float rt[12];
      std::memcpy(rt, R_W_Be, 9 * sizeof(float));
      std::memcpy(rt + 9, t_W_Be, 3 * sizeof(float));
      MH_HIP(ctx, map->s_rt.reserve(sizeof(rt), ctx->stream, false));
      MH_HIP(ctx, hipMemcpyAsync(map->s_rt.p, rt, sizeof(rt), hipMemcpyHostToDevice, ctx->stream));
      MH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // rt is a stack buffer
      // a copy of the body cloud (the scan keeps its own), transformed in place; the filter reads xyz at the records' stride
      MH_HIP(ctx, map->s_in.reserve(n * sizeof(mh_point32), ctx->stream, false));
      MH_HIP(ctx, hipMemcpyAsync(map->s_in.p, scan->d_body.p, n * sizeof(mh_point32), hipMemcpyDeviceToDevice, ctx->stream));
      MH_HIP(ctx, mh::launch_transform(static_cast<mh_point32 *>(map->s_in.p), static_cast<int>(n), static_cast<const float *>(map->s_rt.p), ctx->stream));
      MH_HIP(ctx, map->s_flags.reserve(n * sizeof(uint32_t), ctx->stream, false));
      MH_HIP(ctx, map->s_pos.reserve(n * sizeof(uint32_t), ctx->stream, false));
      MH_HIP(ctx, map->s_temp.reserve(mh::shard_temp_bytes(n) > mh::map_temp_bytes(n) ? mh::shard_temp_bytes(n) : mh::map_temp_bytes(n), ctx->stream, false));
      MH_HIP(ctx, map->s_shard.reserve(n * 3 * sizeof(float), ctx->stream, false));
      MH_HIP(ctx, mh::launch_shard_filter(static_cast<const float *>(map->s_in.p), static_cast<uint32_t>(n), static_cast<uint32_t>(sizeof(mh_point32) / sizeof(float)),
                                          map->inv_leaf, static_cast<uint32_t>(world), static_cast<uint32_t>(rank), block_log2,
                                          static_cast<uint32_t *>(map->s_flags.p), static_cast<uint32_t *>(map->s_pos.p), static_cast<float *>(map->s_shard.p),
                                          &map->d_state->n_keep, map->s_temp.p, map->s_temp.cap, ctx->stream));
      const int rc = fetch_state(map);
      if (rc != MH_OK) return rc;
      kept = map->h_state->n_keep;
    }
    return insert_device(map, static_cast<const float *>(map->s_shard.p), kept, 3, nullptr);
  });
}

int mh_map_insert_device(mh_map * map, const void * d_points, size_t n, size_t stride_floats, const float * R, const float * t)
{
  if (!map || (!d_points && n)) return fail(map ? map->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_map_insert_device: NULL argument");
  mh_ctx * ctx = map->ctx;
  return guarded(ctx, "mh_map_insert_device", [&]() -> int {
    if (stride_floats < 3) return fail(ctx, MH_ERR_INVALID_ARG, "mh_map_insert_device: stride_floats must be >= 3");
    if ((R == nullptr) != (t == nullptr)) return fail(ctx, MH_ERR_INVALID_ARG, "mh_map_insert_device: R and t go together");
    MH_HIP(ctx, mh_enter(ctx));
    const float * d_rt = nullptr;
    if (R) {
      float rt[12];
      std::memcpy(rt, R, 9 * sizeof(float));
      std::memcpy(rt + 9, t, 3 * sizeof(float));
      MH_HIP(ctx, map->s_rt.reserve(sizeof(rt), ctx->stream, false));
      MH_HIP(ctx, hipMemcpyAsync(map->s_rt.p, rt, sizeof(rt), hipMemcpyHostToDevice, ctx->stream));
      MH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // rt is a stack buffer
      d_rt = static_cast<const float *>(map->s_rt.p);
    }
    return insert_device(map, static_cast<const float *>(d_points), n, stride_floats, d_rt);
  });
}

int mh_map_insert_from_scan(mh_map * map, const mh_scan * scan, const float R_W_Be[9], const float t_W_Be[3])
{
  if (!map || !scan) return fail(map ? map->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_map_insert_from_scan: NULL argument");
  if (!scan->preprocessed) return fail(map->ctx, MH_ERR_INVALID_ARG, "mh_map_insert_from_scan: no mh_scan_preprocess_geometric before");
  if (scan->ctx->device != map->ctx->device) return fail(map->ctx, MH_ERR_INVALID_ARG, "mh_map_insert_from_scan: scan lives on another device");
  return mh_map_insert_device(map, scan->d_body.p, scan->n_body, sizeof(mh_point32) / sizeof(float), R_W_Be, t_W_Be);
}

int mh_map_copy(const mh_map * src, mh_map ** out)
{
  if (!src || !out) return fail(src ? src->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_map_copy: NULL argument");
  *out = nullptr;
  mh_ctx * ctx = src->ctx;
  return guarded(ctx, "mh_map_copy", [&]() -> int {
    if (src->poisoned) return fail(ctx, MH_ERR_HIP, "mh_map_copy: the map is inconsistent after a failed mutation");
    MH_HIP(ctx, mh_enter(ctx));
    mh_map * m = nullptr;
    int rc = map_alloc(ctx, src->cfg, &m);
    if (rc != MH_OK) return rc;
    // deep copy, device to device, with growth headroom (Geometric::updateMap inserts right after copying)
    m->table_cap = src->table_cap;
    hipError_t e = m->d_table.reserve(src->table_cap * sizeof(int4), ctx->stream, false);
    if (e == hipSuccess) e = hipMemcpyAsync(m->d_table.p, src->d_table.p, src->table_cap * sizeof(int4), hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) {
      rc = ensure_voxels(m, static_cast<size_t>(src->n_voxels) + src->n_voxels / 16 + 1024);
      if (rc == MH_OK) rc = ensure_blocks(m, static_cast<size_t>(src->n_blocks) + src->n_blocks / 16 + 256, static_cast<size_t>(src->n_blocks) + src->n_blocks / 16 + 256);
    }
    const size_t nv = static_cast<size_t>(src->n_voxels) + 1;  // + the slack bucket
    auto cp = [&](const DevBuf & a, DevBuf & b, size_t bytes) {
      return bytes && e == hipSuccess && rc == MH_OK ? hipMemcpyAsync(b.p, a.p, bytes < a.cap ? bytes : a.cap, hipMemcpyDeviceToDevice, ctx->stream) : e;
    };
    e = cp(src->d_buckets, m->d_buckets, nv * mh::kBucketStride * sizeof(float4));
    e = cp(src->d_qbuckets, m->d_qbuckets, nv * mh::kBucketStride * sizeof(uint32_t));
    e = cp(src->d_vox, m->d_vox, static_cast<size_t>(src->n_voxels) * sizeof(int4));
    e = cp(src->d_lru, m->d_lru, static_cast<size_t>(src->n_voxels) * sizeof(unsigned long long));
    e = cp(src->d_cells, m->d_cells, (static_cast<size_t>(src->n_blocks) + 1) * mh::kCellsPerBlock * sizeof(uint32_t));
    m->n_voxels = src->n_voxels;
    m->n_blocks = src->n_blocks;
    m->n_points = src->n_points;
    m->lru_counter = src->lru_counter;
    if (e == hipSuccess && rc == MH_OK) rc = push_state(m);
    if (e != hipSuccess || rc != MH_OK) {
      (void)hipStreamSynchronize(ctx->stream);
      map_free(m);
      return rc != MH_OK ? rc : hip_fail(ctx, e, "mh_map_copy");
    }
    *out = m;
    return MH_OK;
  });
}

int mh_map_retain(mh_map * map)
{
  if (!map) return fail(nullptr, MH_ERR_INVALID_ARG, "mh_map_retain: map is NULL");
  map->refs.fetch_add(1);
  return MH_OK;
}

void mh_map_release(mh_map * map)
{
  if (!map) return;
  if (map->refs.fetch_sub(1) == 1) {
    (void)mh_enter(map->ctx);
    (void)hipStreamSynchronize(map->ctx->stream);  // the last factor is gone (each waited for its own stream): only the map's own work can be in flight
    map_free(map);
  }
}

int mh_map_get_stats(const mh_map * map, mh_map_stats * out)
{
  if (!map || !out) return fail(map ? map->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_map_get_stats: NULL argument");
  out->n_voxels = map->n_voxels;
  out->n_points = static_cast<int64_t>(map->n_points);
  out->n_blocks = map->n_blocks;
  out->device_bytes = static_cast<int64_t>(map->d_table.cap + map->d_cells.cap + map->d_buckets.cap + map->d_qbuckets.cap + map->d_vox.cap + map->d_lru.cap);
  out->uploads = map->inserts;
  out->upload_bytes = map->upload_bytes;
  out->delta_uploads = map->inserts;
  out->full_uploads = 0;
  return MH_OK;
}

int mh_map_get_cloud(const mh_map * cmap, float * xyz, size_t capacity_points, size_t * n_out)
{
  if (!cmap || !n_out) return fail(cmap ? cmap->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_map_get_cloud: NULL argument");
  mh_map * map = const_cast<mh_map *>(cmap);  // scratch buffers only
  mh_ctx * ctx = map->ctx;
  return guarded(ctx, "mh_map_get_cloud", [&]() -> int {
    if (map->poisoned) return fail(ctx, MH_ERR_HIP, "mh_map_get_cloud: the map is inconsistent after a failed mutation");
    *n_out = static_cast<size_t>(map->n_points);
    if (!xyz || map->n_points == 0) return MH_OK;
    MH_HIP(ctx, mh_enter(ctx));
    const uint32_t nv = map->n_voxels;
    const size_t np = static_cast<size_t>(map->n_points);
    MH_HIP(ctx, map->s_flags.reserve(nv * sizeof(uint32_t), ctx->stream, false));
    MH_HIP(ctx, map->s_pos.reserve(nv * sizeof(uint32_t), ctx->stream, false));
    MH_HIP(ctx, map->s_temp.reserve(mh::map_temp_bytes(nv), ctx->stream, false));
    MH_HIP(ctx, map->s_in.reserve(np * 3 * sizeof(float), ctx->stream, false));
    MH_HIP(ctx, mh::launch_map_cloud(arrays_of(map), nv, static_cast<uint32_t *>(map->s_flags.p), static_cast<uint32_t *>(map->s_pos.p),
                                     static_cast<float *>(map->s_in.p), np, map->s_temp.p, map->s_temp.cap, ctx->stream));
    const size_t take = np < capacity_points ? np : capacity_points;
    MH_HIP(ctx, hipMemcpyAsync(xyz, map->s_in.p, take * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MH_OK;
  });
}

int mh_map_knn(mh_map * map, const double * queries, size_t n, int k, double * point_xyz, double * sq_dists, int32_t * found)
{
  if (!map || !queries || !point_xyz || !sq_dists || !found)
    return fail(map ? map->ctx : nullptr, MH_ERR_INVALID_ARG, "mh_map_knn: NULL argument");
  mh_ctx * ctx = map->ctx;
  return guarded(ctx, "mh_map_knn", [&]() -> int {
    if (k < 1 || k > 8) return fail(ctx, MH_ERR_UNSUPPORTED, "mh_map_knn: k must be in 1..8");
    if (map->poisoned) return fail(ctx, MH_ERR_HIP, "mh_map_knn: the map is inconsistent after a failed mutation");
    MH_HIP(ctx, mh_enter(ctx));
    if (n == 0) return MH_OK;
    DevTemp<double> d_q, d_p, d_s;
    DevTemp<int32_t> d_f;
    MH_HIP(ctx, d_q.alloc(n * 3 * sizeof(double)));
    MH_HIP(ctx, d_p.alloc(n * k * 3 * sizeof(double)));
    MH_HIP(ctx, d_s.alloc(n * k * sizeof(double)));
    MH_HIP(ctx, d_f.alloc(n * sizeof(int32_t)));
    MH_HIP(ctx, hipMemcpyAsync(d_q, queries, n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    MH_HIP(ctx, mh::launch_map_knn(map_view(map), d_q, static_cast<int>(n), k, d_p, d_s, d_f, ctx->stream));
    MH_HIP(ctx, hipMemcpyAsync(point_xyz, d_p, n * k * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MH_HIP(ctx, hipMemcpyAsync(sq_dists, d_s, n * k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MH_HIP(ctx, hipMemcpyAsync(found, d_f, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    MH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MH_OK;
  });
}

}  // extern "C"
