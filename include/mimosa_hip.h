/*
 * mimosa_hip.h — C ABI of libmimosa_hip.so: the MI355X-native (gfx950, HIP) implementation of the
 * LiDAR geometric-factor hot path of ntnu-arl/mimosa.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ / torch / GTSAM
 * types.  Every entry point names the reference interface it replaces (paths relative to the
 * reference's mimosa/ package).  The C++ host layer in mimosa_amd/host/ (ICPFactor, Geometric,
 * IncrementalVoxelMap, Manager) is a thin mirror of the reference classes over this ABI.
 *
 * Conventions
 *  - every function returns an mh_status (0 = MH_OK); nothing throws across the boundary; per-point
 *    failures are RejectStatus values (same enum values 0..8 as geometric_factor.hpp:35-46), never
 *    errors.
 *  - matrices are row-major doubles; poses are (R[9], t[3]).
 *  - caller-owned host buffers are only read/written for the duration of the call; device memory is
 *    owned by the library behind opaque handles.
 *  - handles are not individually thread-safe; different handles may be used from different host
 *    threads (the library sets the device per call; one HIP stream per context).
 */
#ifndef MIMOSA_HIP_H
#define MIMOSA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MH_ABI_VERSION 3  /* 3: mh_set_overlap is gone with the component server (round 6); 2: the caller-driven sharding entry points,
                           * mh_map_fork and mh_map_sync went (round 5) */

typedef enum mh_status {
  MH_OK = 0,
  MH_ERR_INVALID_ARG = 1,
  MH_ERR_HIP = 2,        /* a HIP runtime call failed; see mh_last_error() */
  MH_ERR_NO_DEVICE = 3,  /* no usable gfx950 device (the library never falls back to the CPU) */
  MH_ERR_OOM = 4,
  MH_ERR_UNSUPPORTED = 5
} mh_status;

/* ICPFactor::RejectStatus, include/mimosa/lidar/geometric_factor.hpp:35-46 */
typedef enum mh_reject_status {
  MH_UNPROCESSED = 0,
  MH_INSUFFICIENT_CORRES_POINTS = 1,
  MH_CORRES_MAX_DIST = 2,
  MH_EIGEN_SOLVER_FAIL = 3,
  MH_MIN_EIGEN_VALUE_LOW = 4,
  MH_LINE = 5,
  MH_CORRES_PLANE_INVALID = 6,
  MH_MAX_ERROR = 7,
  MH_VALID = 8
} mh_reject_status;

/* lidar::Point, include/mimosa/lidar/point.hpp:18-39 (32 bytes, 16-byte aligned) */
typedef struct mh_point32 {
  float x, y, z, pad;
  float intensity;
  uint32_t t;   /* ns since the beginning of the scan */
  uint32_t idx; /* index in the original cloud */
  float range;
} mh_point32;

/* lidar::PointOuster, include/mimosa/lidar/point.hpp:42-50 (32 bytes, 16-byte aligned): what
 * Manager::prepareInput<PointOuster> reads from the sensor_msgs::PointCloud2 (lidar/manager.cpp:155-157) */
typedef struct mh_ouster_point {
  float x, y, z, pad;
  float intensity;
  uint32_t t; /* ns since the beginning of the scan */
  uint16_t reflectivity;
  uint16_t ring;
  uint32_t pad2;
} mh_ouster_point;

/* The fields of lidar::ManagerConfig (include/mimosa/lidar/manager.hpp:24-41) and GeometricConfig
 * (geometric_config.hpp:43-44) that Manager::prepareInput reads. */
typedef struct mh_input_config {
  float range_min, range_max;         /* metres; compared as squares in float (manager.cpp:19-20, :281-282) */
  float intensity_min, intensity_max; /* manager.cpp:272-276 */
  float ns_max;                       /* manager.cpp:306 */
  float z_offset;                     /* -lidar_to_sensor_transform[11] / 1000 (manager.cpp:18, :313) */
  int32_t create_full_res_pointcloud; /* loop stride 1 instead of point_skip_divisor (manager.cpp:244-246) */
  int32_t point_skip_divisor;         /* geometric subset: raw index % divisor == 0 (manager.cpp:318) */
  int32_t ring_skip_divisor;          /* geometric subset: ring % divisor == 0 (manager.cpp:331) */
} mh_input_config;

typedef struct mh_scan_info {
  uint64_t n_in;          /* raw points given to mh_scan_prepare_input */
  uint64_t n_full;        /* points_full_: points that passed the filters */
  uint64_t n_geometric;   /* geometric_point_idxs_ */
  uint64_t n_unique_ns;   /* unique_ns_ */
  uint64_t n_body;        /* Be_cloud_ (after mh_scan_preprocess_geometric) */
  uint64_t n_downsampled; /* sm_Be_cloud_ds_ */
  uint32_t last_point_ns; /* corrected_ts_ = header_ts + last_point_ns * 1e-9 (manager.cpp:336) */
  uint32_t pad;
} mh_scan_info;

/* lidar::RegistrationConfig, include/mimosa/lidar/geometric_config.hpp:17-33 (same field order) */
typedef struct mh_reg_config {
  float source_voxel_grid_filter_leaf_size;
  float source_voxel_grid_min_dist_in_voxel;
  float target_ivox_map_leaf_size;
  float target_ivox_map_min_dist_in_voxel;
  uint64_t num_corres_points; /* 2..8 supported */
  float max_corres_distance;
  float plane_validity_distance;
  float lidar_point_noise_std_dev;
  int32_t use_huber;
  float huber_threshold;
  int32_t reg_4_dof;
  int32_t project_on_degneneracy; /* (sic) spelling follows the reference */
  float degen_thresh_rot;
  float degen_thresh_trans;
} mh_reg_config;

/* gtsam_points::iVox settings as configured by Geometric's constructor, src/lidar/geometric.cpp:23-28 */
typedef struct mh_map_config {
  double leaf_size;            /* scan_to_map.target_ivox_map_leaf_size */
  double min_dist_in_cell;     /* scan_to_map.target_ivox_map_min_dist_in_voxel */
  int32_t max_points_in_cell;  /* FlatContainer default 20 (<= 20 supported) */
  int32_t neighbor_voxel_mode; /* 1, 7, 19 or 27 */
  int64_t lru_horizon;         /* GeometricConfig::lru_horizon */
  int32_t lru_clear_cycle;     /* iVox default 10 */
  int32_t reserved;
} mh_map_config;

typedef struct mh_map_stats {
  int64_t n_voxels;
  int64_t n_points;
  int64_t n_blocks;          /* 4x4x4-voxel blocks in the device block table */
  int64_t device_bytes;      /* HBM held by this map */
  int64_t uploads;           /* insert calls so far */
  int64_t upload_bytes;      /* bytes of point batches pushed host->device (nothing else ever travels) */
  int64_t delta_uploads;     /* == uploads: an insert only ever moves its own batch */
  int64_t full_uploads;      /* always 0 since the map is maintained on the device (kept for ABI stability) */
} mh_map_stats;

/*
 * What gtsam::HessianFactor(key[, key2], G11, [G12,] g1, [G22, g2,] f) receives from
 * ICPFactor::linearize (geometric_factor.hpp:459-462, 559-560) plus every getter-visible side
 * output (getLocalizabilities :52-62, getDegenInfo :64-70, getLinearizeCount :72) and the status
 * histogram Geometric::getFactors builds (src/lidar/geometric.cpp:280-323).
 * The Hessian factor is HessianFactor(key, H_ss, -b_s, f).
 */
typedef struct mh_icp_result {
  double H_ss[36]; /* J_s^T J_s, row-major 6x6, rotation block first (GTSAM Pose3 tangent order) */
  double H_st[36]; /* binary factor only */
  double H_tt[36]; /* binary factor only */
  double b_s[6];   /* J_s^T e */
  double b_t[6];   /* binary factor only */
  double f;        /* sum e^2 */
  double loc_trans_comp[3], loc_rot_comp[3], loc_trans_final[3], loc_rot_final[3];
  double eigvec_trans[9], eigvec_rot[9]; /* eigenvectors in columns, row-major storage */
  double degen_rot[3], degen_trans[3], degen_eigvec_rot[9], degen_eigvec_trans[9];
  int32_t status_hist[9];
  int32_t linearize_count;
  double mean_candidates; /* mean number of map points in the occupied neighbour voxels of a query that ran k-NN
                             (what the reference scans; SURVEY.md §8(d)'s C_q) */
  double mean_scanned;    /* mean number the device actually scanned after exact box-distance pruning */
  int64_t n_knn;          /* queries that ran k-NN in this call (the rest hit the DA cache) */
  int64_t n_exact_fallback; /* of those, queries whose coarse f32 scan could not be proven exact and were redone in fp64 */
  /* device-side timing of this call, ms; -1 unless this call was timed (mh_set_profiling) */
  float gpu_ms_linearize; /* icp_linearize kernel (K3) */
  float gpu_ms_localizability; /* component-localizability kernel (K4) */
} mh_icp_result;

typedef struct mh_ctx mh_ctx;
typedef struct mh_map mh_map;
typedef struct mh_icp mh_icp;
typedef struct mh_scan mh_scan;

/* ---- context ------------------------------------------------------------------------------ */
int mh_abi_version(void);
/* Binds a context to HIP device `device` and creates its stream.  MH_ERR_NO_DEVICE if there is no
 * GPU: there is no CPU fallback. */
int mh_init(int device, mh_ctx ** out);
void mh_shutdown(mh_ctx * ctx);
/* Last error text for this context (or the calling thread when ctx == NULL). */
const char * mh_last_error(const mh_ctx * ctx);
/* Per-kernel HIP-event timing inside mh_icp_linearize / mh_icp_linearize_async: 0 = off (default), n >= 1 =
 * bracket the kernels of every n-th linearize call of a factor (each event record costs ~4 us of stream time,
 * so a throughput measurement samples).  Untimed calls report gpu_ms_* = -1. */
int mh_set_profiling(mh_ctx * ctx, int every);
/* hipStream_t of the context, for callers that want to order their own work / events on it. */
void * mh_stream(mh_ctx * ctx);
int mh_synchronize(mh_ctx * ctx);
/* Event pair on the context stream: begin/end bracket, returns elapsed ms (synchronises). */
int mh_timer_begin(mh_ctx * ctx);
int mh_timer_end(mh_ctx * ctx, float * ms);

/* ---- target map: IncrementalVoxelMapPCL / gtsam_points::iVox --------------------------------
 * replaces include/mimosa/lidar/incremental_voxel_map.hpp:22-54, src/lidar/incremental_voxel_map.cpp:14-62.
 * The map lives on the device and is maintained there: insertion (greedy first-come-first-kept per voxel, min-distance
 * rule, per-voxel cap, voxels numbered in creation order), the LRU purge, getCloud and the copy are kernels; the
 * host keeps counters.  Mutations are synchronous and drain the device first, so factors of any context that share
 * the map see either the old or the new state, never a half-written one. */
int mh_map_create(mh_ctx * ctx, const mh_map_config * cfg, mh_map ** out); /* ctor, geometric.cpp:23-28 */
/* IncrementalVoxelMapPCL::insert (incremental_voxel_map.cpp:19-24).  xyz: n host points, `stride_floats` floats apart
 * (3 for packed xyz, 8 for mh_point32).  MH_ERR_UNSUPPORTED if a point is NaN or farther than 2^20 voxels from the origin. */
int mh_map_insert(mh_map * map, const float * xyz, size_t n, size_t stride_floats);
/* The same for a batch that is already on the map's device (d_points: device pointer).  R, t (both or neither): the
 * f32 rigid transform p <- R p + t applied first — Geometric::updateMap's world transform (geometric.cpp:483-490),
 * Eigen's evaluation order, no FMA. */
int mh_map_insert_device(mh_map * map, const void * d_points, size_t n, size_t stride_floats, const float * R, const float * t);
/* Geometric::updateMap's insert (geometric.cpp:483-495) straight from a device-resident scan: Be_cloud_ (the body-frame
 * geometric subset of mh_scan_preprocess_geometric) transformed by T_W_Be in f32 and inserted, no host round trip. */
int mh_map_insert_from_scan(mh_map * map, const mh_scan * scan, const float R_W_Be[9], const float t_W_Be[3]);
/* Deep copy — IncrementalVoxelMapPCL copy ctor (incremental_voxel_map.hpp:33-42) used by Geometric::updateMap's
 * copy-then-insert (geometric.cpp:494): device to device, both maps stay writable. */
int mh_map_copy(const mh_map * map, mh_map ** out);
/* shared_ptr semantics: factors retain the map they were built with. */
int mh_map_retain(mh_map * map);
void mh_map_release(mh_map * map);
int mh_map_get_stats(const mh_map * map, mh_map_stats * out);
/* IncrementalVoxelMapPCL::getCloud (incremental_voxel_map.cpp:34-38): all points in voxel order.
 * xyz may be NULL to query the size; returns the number of points through n_out. */
int mh_map_get_cloud(const mh_map * map, float * xyz, size_t capacity_points, size_t * n_out);
/* IncrementalVoxelMapPCL::knn_search (incremental_voxel_map.cpp:26-32) for a batch of fp64 queries,
 * run on the device.  point_xyz (n*k*3 doubles) receives the neighbour coordinates
 * (ivox->point(id), geometric_factor.hpp:184), sq_dists n*k ascending, found[n] = number found
 * (the reference returns found == k). */
int mh_map_knn(mh_map * map, const double * queries, size_t n, int k, double * point_xyz,
               double * sq_dists, int32_t * found);

/* ---- ICPFactor ------------------------------------------------------------------------------
 * replaces include/mimosa/lidar/geometric_factor.hpp:25-563 */
/* ctor (:119-156): copies the source cloud to the device, retains the map, allocates the per-point
 * data-association state zero-initialised.  `map` may belong to another context of the same device
 * (several contexts = several HIP streams sharing one read-only map); while factors of other contexts
 * are linearizing, do not insert into that map. */
int mh_icp_create(mh_ctx * ctx, mh_map * map, const mh_point32 * source, size_t n,
                  const mh_reg_config * cfg, int is_binary, mh_icp ** out);
/* clone() (:160-164): deep-copies the per-point state, shares the map. */
int mh_icp_clone(const mh_icp * icp, mh_icp ** out);
void mh_icp_destroy(mh_icp * icp);
/* linearize(Values) (:231-562).  T_src = Values[keys[0]]; (R_tgt, t_tgt) = Values[keys[1]] for a
 * binary factor, NULL for unary; g_unit = Values[G(0)].unitVector() (read unconditionally, :257).
 * Blocks until the result is on the host. */
int mh_icp_linearize(mh_icp * icp, const double R_src[9], const double t_src[3],
                     const double * R_tgt, const double * t_tgt, const double g_unit[3],
                     mh_icp_result * out);
/* Same work enqueued on the context stream without waiting; the result is written to *out (which
 * must stay valid) when mh_icp_wait / mh_synchronize returns. Up to 64 calls may be in flight. */
int mh_icp_linearize_async(mh_icp * icp, const double R_src[9], const double t_src[3],
                           const double * R_tgt, const double * t_tgt, const double g_unit[3],
                           mh_icp_result * out);
int mh_icp_wait(mh_icp * icp);
/* Every live ICPFactor of the sliding window re-linearized in ONE pass: what graph::Manager::defineNoLock's
 * smoother_->update() + additional_update_iterations (src/graph/manager.cpp:585-588) make GTSAM do one factor at
 * a time.  icps[f] is linearized at (R_src + 9 f, t_src + 3 f[, R_tgt + 9 f, t_tgt + 3 f], g_unit + 3 f) into out[f];
 * the per-point results (status, cached mean / normal of every point) are bit-identical to n_factors separate
 * mh_icp_linearize calls, and so are the sums when the factor runs the same LAUNCH CLASS in both — the class follows the
 * points of the whole launch (two lanes per point for k = 5 launches of up to 32 768 points, one lane up to 65 536, the
 * 512-thread class above); where the window's total moves a factor to another class its rows are added in another order
 * (H, b, f agree to ~1e-13 relative).  Factors that share a kernel instantiation — launch class, num_corres_points == 5 or
 * not, the map's neighbour mode, unary / binary — form one launch group: one K3 launch and one K4 launch per group, so a
 * window of like factors (the usual case) is one launch pair and any mix is accepted.  All factors must belong to one context and have no call in flight;
 * at most 64 per call.  R_tgt / t_tgt may be NULL when no factor is binary.  Blocks until every result is on the host. */
int mh_icp_linearize_batch(mh_icp * const * icps, size_t n_factors, const double * R_src, const double * t_src,
                           const double * R_tgt, const double * t_tgt, const double * g_unit, mh_icp_result * out);
/* getStatuses / getCorresMeansTarget / getCorresNormalsTarget (:48-50); any pointer may be NULL. */
int mh_icp_get_state(const mh_icp * icp, int32_t * status, double * means, double * normals);
/* Forget all data associations (== a freshly constructed factor): enqueued, no host sync. */
int mh_icp_reset(mh_icp * icp);
/* The component localizabilities and the status histogram (geometric_factor.hpp:430-457, geometric.cpp:280-323) need a second
 * pass over the points in the eigenbasis of the finished H — a kernel of its own (K4), a fifth of a call.  The reference
 * computes them in every linearize() but reads them in ONE place: Geometric::getFactors, right after its own linearize
 * (geometric.cpp:205-214, getLocalizabilities / the status log); the re-linearizations ISAM2 drives afterwards
 * (graph/manager.cpp:585-588) never have them looked at.  enabled = 0 skips the pass for this factor's following calls
 * (also in mh_icp_linearize_batch when no factor of the batch wants it): loc_trans_comp / loc_rot_comp come back NaN and
 * status_hist -1; H, b, f, the final localizabilities, eigenvectors and degeneracy info are bit-identical either way.
 * Default: enabled (the reference's behaviour). */
int mh_icp_set_components(mh_icp * icp, int enabled);
size_t mh_icp_size(const mh_icp * icp);

/* ---- deskew / rigid transforms ----------------------------------------------------------------
 * Manager::deskewPoints hot loop (src/lidar/manager.cpp:496-509): every point whose t equals
 * unique_ns[g] gets p <- R_g p + t_g in float (no FMA, Eigen's evaluation order).  Rt12 = n_groups x
 * {R row-major 9 floats, t 3 floats}.  If R_B_L != NULL the Geometric::preprocess body transform
 * (src/lidar/geometric.cpp:154-161) is applied afterwards in the same kernel. In place, host buffers. */
int mh_deskew(mh_ctx * ctx, mh_point32 * pts, size_t n, const uint32_t * unique_ns,
              const float * Rt12, size_t n_groups, const float * R_B_L, const float * t_B_L);
/* p <- R p + t in float for a whole cloud (body transform geometric.cpp:154-161, world transform
 * geometric.cpp:483-490). In place, host buffers. */
int mh_transform_f32(mh_ctx * ctx, mh_point32 * pts, size_t n, const float R[9], const float t[3]);


/* ---- device-resident scan front end ------------------------------------------------------------
 * The raw cloud is uploaded once; input filter, deskew, body-frame subset and voxel down-sampler run on
 * the device and the ICP factor takes its source cloud from there (no host round trip between
 * Manager::prepareInput and the first linearize).  Every stage reproduces the reference's sequential
 * result exactly, including output order.  Call order: prepare_input -> (caller propagates the IMU over
 * unique_ns) -> deskew -> preprocess_geometric -> mh_icp_create_from_scan. */
int mh_scan_create(mh_ctx * ctx, mh_scan ** out);
void mh_scan_destroy(mh_scan * scan);
/* Manager::prepareInput<PointOuster> (lidar/manager.cpp:149-383): NaN / intensity / range / ns_max filters,
 * z offset, points_full_ (order preserved), geometric subset indices, distinct timestamps (ascending). */
int mh_scan_prepare_input(mh_scan * scan, const mh_ouster_point * raw, size_t n, const mh_input_config * cfg,
                          mh_scan_info * info);
/* Manager::prepareInput<PointT> for the reference's OTHER point types (include/mimosa/lidar/point.hpp:52-131:
 * PointOusterOdyssey, PointOusterR8, PointHesai, PointLivox, PointLivoxFromCustom2, PointVelodyne,
 * PointVelodyneAnybotics, PointRslidar) — and PointOuster itself — described by where the fields sit in a record,
 * i.e. what the sensor_msgs::PointCloud2 `fields` array says.  The per-type branches of the reference are selected by the
 * layout: time decoding (lidar/manager.cpp:285-304), reflectivity as intensity (:265-271), the Livox tag filter
 * (:256-262), whether the ring filter applies (:321-332); `transpose` (:177-203, RSAiry / VelodyneAnybotics) and
 * `organize_by_ring` (:205-241, applied when height == 1 and the layout has a ring field, as there; ring numbers must be
 * < 128, the size of the reference's tables, else MH_ERR_UNSUPPORTED) reorder the cloud first.  width * height = n.
 * The double -> uint32 time conversions follow the reference's x86-64 build (truncate to 64 bits, keep the low word). */
typedef enum mh_time_kind {
  MH_TIME_U32_NS = 0,     /* t_ns = t                                    PointOuster*, PointLivoxFromCustom2 */
  MH_TIME_F64_S_ABS = 1,  /* t_ns = (timestamp - header_ts) * 1e9        PointHesai, PointRslidar */
  MH_TIME_F64_NS_ABS = 2, /* t_ns = timestamp - header_ts * 1e9          PointLivox */
  MH_TIME_F32_S = 3       /* t_ns = time * 1e9                           PointVelodyne, PointVelodyneAnybotics */
} mh_time_kind;
typedef enum mh_ring_kind { MH_RING_NONE = 0, MH_RING_U16 = 1, MH_RING_U8 = 2, MH_RING_F32 = 3 } mh_ring_kind;
typedef struct mh_point_layout {
  uint32_t stride;                  /* bytes per record */
  uint32_t off_x, off_y, off_z;     /* float */
  uint32_t off_intensity;           /* float intensity, or uint16 reflectivity when intensity_is_u16 (PointOusterOdyssey) */
  int32_t intensity_is_u16;
  uint32_t off_time;
  int32_t time_kind;                /* mh_time_kind */
  uint32_t off_ring;
  int32_t ring_kind;                /* mh_ring_kind: the field organize_by_ring and the ring filter read */
  int32_t ring_filter;              /* 0: `ring % ring_skip_divisor` is not applied (Livox, VelodyneAnybotics, OusterOdyssey) */
  uint32_t off_tag;                 /* uint8 Livox tag */
  int32_t has_tag;                  /* keep only (tag & 0x30) == 0x10 or 0x00 */
} mh_point_layout;
int mh_scan_prepare_input_layout(mh_scan * scan, const void * raw, size_t n, const mh_point_layout * layout, uint32_t width,
                                 uint32_t height, int transpose, int organize_by_ring, double header_ts,
                                 const mh_input_config * cfg, mh_scan_info * info);
/* Same, for a raw cloud that is already in device memory (a driver that DMAs packets to the GPU, or a benchmark that
 * wants the figure without the PCIe upload): d_raw must stay valid and unchanged until the call returns. */
int mh_scan_prepare_input_device(mh_scan * scan, const mh_ouster_point * d_raw, size_t n, const mh_input_config * cfg,
                                 mh_scan_info * info);
/* Pipelined input.  mh_scan_prefetch stages the NEXT cloud while another scan is being processed: raw[n] is copied into the
 * handle's pinned staging buffer and the host-to-device copy is enqueued on a copy stream of the handle's own; it returns as
 * soon as the caller's buffer may go away.  It may be called from another host thread than the one that runs the pipeline,
 * on a handle no other call is using at that moment.  mh_scan_prepare_input_prefetched is mh_scan_prepare_input on the staged
 * cloud: the context stream waits for the copy on the device, the host does not. */
int mh_scan_prefetch(mh_scan * scan, const mh_ouster_point * raw, size_t n);
int mh_scan_prepare_input_prefetched(mh_scan * scan, const mh_input_config * cfg, mh_scan_info * info);
/* unique_ns_ (lidar/manager.cpp:344-368), ascending; the caller's IMU propagation needs them on the host. */
int mh_scan_get_unique_ns(const mh_scan * scan, uint32_t * out, size_t capacity, size_t * n_out);
/* Manager::deskewPoints' per-point part (lidar/manager.cpp:496-509) on points_full_, in place:
 * Rt12[g] = pose (row-major R, then t, float) of the group with timestamp unique_ns[g].  Rt12 is read before the call returns;
 * the kernel itself is only ENQUEUED (every later call on the handle is ordered behind it on the context's stream). */
int mh_scan_deskew(mh_scan * scan, const float * Rt12, size_t n_groups);
/* Geometric::preprocess (geometric.cpp:128-183): Be_cloud_ = R_B_L * points_full_[geometric idx] + t_B_L
 * (f32), then Geometric::downsample (geometric.cpp:55-126) into sm_Be_cloud_ds_.  max_points_per_voxel <= 20
 * (the reference passes the literal 20). */
int mh_scan_preprocess_geometric(mh_scan * scan, const float R_B_L[9], const float t_B_L[3], double leaf_size,
                                 int max_points_per_voxel, double min_dist_in_voxel, mh_scan_info * info);
/* Copies a stage's cloud to the host.  which: 0 points_full_, 1 Be_cloud_, 2 sm_Be_cloud_ds_. */
int mh_scan_get_points(const mh_scan * scan, int which, mh_point32 * out, size_t capacity, size_t * n_out);
/* The DEVICE address of a stage's cloud (which as above; valid until the scan object is prepared again or destroyed) for callers
 * that hand it on without a copy — mh_shard_icp_create(points_on_device = 1) on a rank's share of sm_Be_cloud_ds_,
 * mh_map_insert_device.  No reference counterpart (the reference's clouds are host members of Manager / Geometric). */
int mh_scan_device_points(const mh_scan * scan, int which, const mh_point32 ** d_points, size_t * n_out);
/* geometric_point_idxs_ (indices into points_full_) / the indices into Be_cloud_ that the down-sampler kept. */
int mh_scan_get_indices(const mh_scan * scan, int which, uint32_t * out, size_t capacity, size_t * n_out);
/* ICPFactor ctor (geometric_factor.hpp:119-142) with sm_Be_cloud_ds_ taken from the device. */
int mh_icp_create_from_scan(mh_ctx * ctx, mh_map * map, const mh_scan * scan, const mh_reg_config * cfg,
                            int is_binary, mh_icp ** out);


/* ---- photometric path: lidar::Photometric + PhotometricFactor ---------------------------------------
 * replaces include/mimosa/lidar/photometric.hpp:22-86 (class Photometric), src/lidar/photometric.cpp,
 * include/mimosa/lidar/photometric_factor.hpp:22-357 and src/lidar/photometric_utils.cpp.
 * An mh_photo is one Photometric instance: configuration, the current Frame (device-resident images, yaw table,
 * proj_idx, pose table — photometric_utils.hpp:42-92) and the tracked features (map_Le_features_).  Frames are
 * reference-counted like the reference's shared_ptr<Frame>: a factor keeps the frame it was built on. */
typedef struct mh_photo mh_photo;
typedef struct mh_photo_factor mh_photo_factor;

/* lidar::PhotometricConfig, include/mimosa/lidar/photometric_config.hpp:15-87: the fields the arithmetic reads.
 * Arrays are copied by mh_photo_create.  The derived parameters fx, fy, cx, beam_offset_m
 * (src/lidar/photometric_config.cpp:98-110) are computed by the library. */
typedef struct mh_photo_config {
  int32_t rows, cols;                   /* sensor/lidar_data_format: pixels_per_column, columns_per_frame */
  int32_t destagger;
  const int32_t * pixel_shift_by_row;   /* rows entries */
  const float * beam_altitude_angles;   /* rows entries, degrees, descending */
  float range_min, range_max;
  int32_t erosion_buffer, patch_size, margin_size;
  float intensity_scale, intensity_gamma;
  int32_t remove_lines, filter_brightness, gaussian_blur, gaussian_blur_size; /* gaussian_blur_size: 3 supported */
  float gradient_threshold, max_dist_from_mean, max_dist_from_plane;
  int32_t nma_radius;
  int32_t num_features_detect;
  float occlusion_range_diff_threshold;
  int32_t max_feature_life_time;
  const double * high_pass_fir;         /* column kernel of removeLines (photometric.cpp:322-327) */
  int32_t n_high_pass;
  const double * low_pass_fir;          /* row kernel */
  int32_t n_low_pass;
  int32_t brightness_window_size[2];    /* cv::Size(width, height) */
  float lidar_origin_to_beam_origin_mm;
  int32_t rotate_patch_to_align_with_gradient; /* photometric.cpp:659-684: new features sample the pattern rotated into their edge frame */
  const int32_t * patch_offsets;        /* edgelet_patch_offsets: n_patch_offsets pairs (du, dv); <= 64 */
  int32_t n_patch_offsets;
  int32_t use_robust_cost_function;
  int32_t robust_cost_function;         /* 0 "huber", 1 "gemanmcclure" */
  double robust_cost_function_parameter, error_scale, max_error, sigma;
  double T_B_L_R[9], T_B_L_t[3];        /* lidar/T_B_S */
  const uint8_t * static_mask;          /* rows * cols, 0 = invalid, or NULL (static_mask_path == "") */
} mh_photo_config;

/* PhotometricFactor::RejectStatus, photometric_factor.hpp:36-47 */
typedef enum mh_photo_status {
  MH_PHOTO_UNPROCESSED = 0,
  MH_PHOTO_POINT_PROJECT_UNDISTORTED = 1,
  MH_PHOTO_POINT_RANGE = 2,
  MH_PHOTO_POINT_PROJECT = 3,
  MH_PHOTO_POINT_MASK = 4,
  MH_PHOTO_POINT_MASK_MARGIN = 5,
  MH_PHOTO_POINT_RANGE_DIFF = 6,
  MH_PHOTO_MAX_ERROR = 7,
  MH_PHOTO_VALID = 8
} mh_photo_status;

/* Feature, include/mimosa/lidar/photometric_utils.hpp:25-40 (per-point arrays travel separately) */
typedef struct mh_photo_feature {
  uint32_t id;
  int32_t life_time;
  int32_t n_points;
  int32_t pad;
  double center[2];
  double normal[3];
  double mean_intensity, sigma_intensity;
} mh_photo_feature;

/* What gtsam::HessianFactor receives from PhotometricFactor::linearize (photometric_factor.hpp:332-353):
 * unary HessianFactor(key, H_bb, -b_b, f); binary HessianFactor(key_b, key_a, H_bb, H_ba, -b_b, H_aa, -b_a, f). */
typedef struct mh_photo_result {
  double H_bb[36], H_ba[36], H_aa[36];
  double b_b[6], b_a[6];
  double f;
  double loc_trans_final[3], loc_rot_final[3], eigvec_trans[9], eigvec_rot[9]; /* getLocalizabilities :51-59 (unary) */
  int32_t status_hist[9];
  int32_t n_exceptions; /* features at which the reference would have THROWN (project(): invalid x coordinate,
                           photometric_utils.cpp:90-97; interpolated_map_T_Le_Lt.at()); reported as
                           PointProjectUndistorted, the host mirror turns a non-zero count into the exception */
  float gpu_ms;         /* kernel time of this call, -1 unless mh_set_profiling is on */
} mh_photo_result;

int mh_photo_create(mh_ctx * ctx, const mh_photo_config * cfg, mh_photo ** out); /* ctor, photometric.cpp:13-70 */
void mh_photo_destroy(mh_photo * photo);
/* Photometric::preprocess (photometric.cpp:92-320): yaw table from points_raw, image formation from points_deskewed,
 * proj_idx, intensity filter chain, Sobel, mask.  points_raw[i] / points_deskewed[i] are the same measurement before /
 * after Manager::deskewPoints; the corrected intensities are written back into points_deskewed (:307-314).
 * unique_ns / T_Le_Lt = interpolated_map_T_Le_Lt_ (lidar/manager.cpp:390-405, :501-503): n_groups ascending
 * timestamps, 12 doubles each (R row-major, then t).  Replaces the current frame. */
int mh_photo_preprocess(mh_photo * photo, const mh_point32 * points_raw, mh_point32 * points_deskewed, size_t n,
                        const uint32_t * unique_ns, const double * T_Le_Lt, size_t n_groups);
/* The same on a device-resident scan (mh_scan_*): points_raw = the scan's points_full_ as they were before
 * mh_scan_deskew (the scan keeps that copy once mh_scan_keep_raw(scan, 1) was called before mh_scan_deskew),
 * points_deskewed = its points_full_ now (corrected intensities are written there). */
int mh_scan_keep_raw(mh_scan * scan, int keep);
int mh_photo_preprocess_scan(mh_photo * photo, mh_scan * scan, const double * T_Le_Lt, size_t n_groups);
/* The same in two steps, for a caller that pipelines scans: _begin ENQUEUES the frame of scan k without making it current and
 * without waiting for anything (it queues behind the scan's deskew on the device, reads the scan, does not write it) — it touches
 * none of the state mh_photo_update_map / mh_photo_detect_features use (tracked features, current frame, their scratch), so one
 * host thread may run it while another is still inside mh_photo_update_map of scan k - 1 on the same object (the only concurrent
 * pair that is allowed; both enqueue on the object's context stream); _commit (after that update has returned) waits for the
 * frame, reports what _begin's kernels found (the project() error), writes the corrected intensities into the scan's cloud —
 * `scan` must still be alive and must not have been prepared again — and makes the frame current.
 * mh_photo_preprocess_scan == _begin + _commit.  A begun frame that is never committed is dropped by the next _begin or by
 * mh_photo_destroy. */
int mh_photo_preprocess_scan_begin(mh_photo * photo, mh_scan * scan, const double * T_Le_Lt, size_t n_groups);
int mh_photo_preprocess_commit(mh_photo * photo);
/* One image of the current frame, rows * cols elements.  which: 0 img_intensity (float), 1 img_range (float),
 * 2 img_dx (float), 3 img_dy (float), 4 img_mask (uint8), 5 img_deskewed_cloud_idx (int32), 6 yaw_angles (float),
 * 7 proj_idx (int32, x 10 per pixel), 8 gradient magnitude (uint8, photometric.cpp:536-540), 9 detection mask
 * before the per-feature circles (uint8, :524-525). */
int mh_photo_get_image(mh_photo * photo, int which, void * out, size_t capacity_bytes);
/* tracked features (map_Le_features_).  Per-point arrays: n_points entries per feature, concatenated in feature
 * order: Le_ps (3 doubles per point), intensities, psi_intensities.  Any output pointer may be NULL. */
int mh_photo_num_features(const mh_photo * photo, size_t * n_features, size_t * n_points_total);
int mh_photo_get_features(const mh_photo * photo, mh_photo_feature * features, double * Le_ps, double * intensities,
                          double * psi);
int mh_photo_set_features(mh_photo * photo, const mh_photo_feature * features, size_t n_features, const double * Le_ps,
                          const double * intensities, const double * psi);
/* Photometric::detectFeatures (photometric.cpp:516-745) on the current frame: up to num_to_detect new features are
 * appended to the tracked ones.  T_W_Be = Values[frame key]; bias_directions: n_directions x 3 (the degenerate
 * directions Manager::postDefineUpdate passes, lidar/manager.cpp:568-581).  The candidate sort (std::sort's exact result, tie order included)
 * runs on up to four host threads of its own for the duration of the call (MH_SORT_THREADS=1: the plain library call). */
int mh_photo_detect_features(mh_photo * photo, int num_to_detect, const double R_W_Be[9], const double t_W_Be[3],
                             const double * bias_directions, size_t n_directions);
/* Photometric::updateMap (photometric.cpp:396-514): feature bookkeeping from the factor's statuses (drop the
 * non-Valid ones, new centres, life_time, max_feature_life_time), then detectFeatures for the missing ones.
 * factor may be NULL (no factor was built for this frame). */
int mh_photo_update_map(mh_photo * photo, mh_photo_factor * factor, const double R_W_Be[9], const double t_W_Be[3],
                        const double * bias_directions, size_t n_directions);
/* Optional: enqueues the part of the next mh_photo_detect_features / mh_photo_update_map that depends on the current frame
 * alone (gradient magnitude, detection mask, candidate compaction, read-back of the candidate list) and returns without
 * waiting; the next detection on this frame then only waits for those copies.  Results are the same with or without it. */
int mh_photo_detect_prefetch(mh_photo * photo);
/* PhotometricFactor ctor (photometric_factor.hpp:86-124) from the current frame and the tracked features (copied).
 * VSVt: 6x6 row-major = V S V^T of Photometric::getFactors (photometric.cpp:373-394), NULL = identity; ignored for
 * the binary form. */
int mh_photo_factor_create(mh_photo * photo, const double * VSVt, int is_binary, mh_photo_factor ** out);
/* clone() (photometric_factor.hpp:120-124): member-wise copy — same frame (shared), own feature list and statuses. */
int mh_photo_factor_clone(const mh_photo_factor * factor, mh_photo_factor ** out);
void mh_photo_factor_destroy(mh_photo_factor * factor);
/* linearize(Values) (:136-355): T_b = Values[keys[0]] (the frame's pose), T_a = Values[keys[1]] for the binary
 * form (NULL otherwise).  Blocks until the result is on the host. */
int mh_photo_factor_linearize(mh_photo_factor * factor, const double R_b[9], const double t_b[3], const double * R_a,
                              const double * t_a, mh_photo_result * out);
/* The same enqueued on the context stream without waiting: the smoother can queue it next to mh_icp_linearize_batch and
 * collect both; mh_photo_factor_wait blocks and fills *out.  One call in flight per factor. */
int mh_photo_factor_linearize_async(mh_photo_factor * factor, const double R_b[9], const double t_b[3], const double * R_a,
                                    const double * t_a);
int mh_photo_factor_wait(mh_photo_factor * factor, mh_photo_result * out);
/* getStatuses / getFeatures().center after the last linearize; rows (optional, parity tooling): per feature and
 * patch point {whitened residual, J_b[6], valid} = 8 doubles, 64 points per feature. */
int mh_photo_factor_get_state(const mh_photo_factor * factor, int32_t * statuses, double * centers, double * rows);
size_t mh_photo_factor_size(const mh_photo_factor * factor);


/* ---- map sharded across GPUs (BASELINE configs[2]): what both the library's own exchange (below) and a
 * framework that owns the stream need ------------------------------------------------------------------------------------
 * No reference counterpart: the reference is single-process.  The map is partitioned into shard blocks of 2^block_log2 voxels
 * per axis; block (bx, by, bz) = voxel coordinates >> block_log2 is owned by rank (bx + A[world] by + B[world] bz) mod world, a
 * lattice colouring of the block grid (neighbouring blocks never share a rank; tools/lattice_table.py derives the two tables;
 * rounds 3-4 used the reference's XORVector3iHash, include/mimosa/lidar/utils.hpp:228-238, which put neighbouring heavy blocks
 * on one rank at random).  A caller that partitions points itself asks mh_shard_owner_of_block instead of re-implementing the
 * rule.  Every rank also stores the one-voxel halo of its blocks, so a query on the owner of its centre voxel finds all
 * 1/7/19/27 neighbour voxels locally.  (ABI version 1 also exported a caller-driven form of the exchange — mh_icp_shard_plan / _pack /
 * _unpack, mh_icp_linearize_begin[_device] / _finish[_device], mh_icp_global_epilogue — with the collectives in the caller's
 * hands; version 2 keeps ONE implementation, the native one below.) */
/* The owner function: rank (0 .. world - 1) of shard block (bx, by, bz); -1 if world is outside 1 .. 64. */
int mh_shard_owner_of_block(int bx, int by, int bz, int world);
/* A context on an existing HIP stream (not owned): kernels, the caller's collectives and its tensor ops are ordered by it. */
int mh_init_on_stream(int device, void * hip_stream, mh_ctx ** out);
/* This rank's share of IncrementalVoxelMapPCL::insert: of the batch (identical on every rank) the points of owned shard
 * blocks plus their one-voxel halo are inserted, in the original order. */
int mh_map_insert_shard(mh_map * map, const float * xyz, size_t n, size_t stride_floats, int world, int rank, int block_log2);
/* The same for the resident scan's Be_cloud_ with Geometric::updateMap's f32 world transform (geometric.cpp:483-495) applied
 * first: mh_map_insert_from_scan for ONE RANK's shard — transform, shard filter and insert on the device, nothing crosses PCIe. */
int mh_map_insert_shard_from_scan(mh_map * map, const mh_scan * scan, const float R_W_Be[9], const float t_W_Be[3], int world, int rank, int block_log2);
/* ICPFactor ctor (geometric_factor.hpp:119-142) from a cloud that is already on the device; the point order is kept. */
int mh_icp_create_from_device(mh_ctx * ctx, mh_map * map, const mh_point32 * d_points, size_t n, const mh_reg_config * cfg,
                              int is_binary, mh_icp ** out);
/* ---- native map-sharded factor: the exchange inside the library (RCCL over xGMI) ------------------------------------
 * No reference counterpart (the reference is single-process): what it must preserve is that the sharded factor equals
 * ICPFactor::linearize (include/mimosa/lidar/geometric_factor.hpp:231-562) point for point.  Same partition as above
 * (shard blocks by the lattice owner function, mh_shard_owner_of_block; one-voxel halo stored by mh_map_insert_shard).
 * One process per GPU; a linearize is ONE chain of enqueues — route kernels, ncclAllToAll of fixed-size per-peer segments
 * [count | records], append, K3, ncclAllReduce of the Hessian sums, (K4 + ncclAllReduce when the components are on), publish —
 * and one wait at its end: no count ever crosses the host, the factor's slot count lives on the device.  A segment that
 * was too small (every rank reads the same maxima from the all-reduce vector) makes every rank repeat the call with larger
 * segments.  With world == 1 and force_collectives == 0 the call is mh_icp_linearize.  librccl is resolved at run time
 * (dlopen; a copy already in the process, e.g. PyTorch's, is used; MH_RCCL_LIB overrides). */
typedef struct mh_shard_comm mh_shard_comm;
typedef struct mh_shard_icp mh_shard_icp;
#define MH_SHARD_UNIQUE_ID_BYTES 128
typedef struct mh_shard_config {
  int32_t block_log2;        /* shard blocks of 2^block_log2 voxels per axis (the map must have been built with the same value) */
  int32_t force_collectives; /* run the full exchange protocol even with one rank (tests, overhead measurement) */
} mh_shard_config;
typedef struct mh_shard_stats {
  uint64_t n_live, n_slots, slot_capacity, n_total; /* points held; slots in use incl. tombstones; capacity; points of all ranks */
  uint32_t segment_records;   /* per-peer segment capacity the next call will use */
  uint32_t last_max_movers;   /* max over ranks and destinations of the points that wanted to move in the last call */
  uint32_t retries_total, retries_last, compactions_total;
  uint32_t collectives_last;  /* collectives the last call entered */
  int32_t world, rank, collective, linearize_count;
} mh_shard_stats;
/* ncclGetUniqueId: one rank creates it, the caller hands it to the others (any channel: a file, MPI, torch.distributed). */
int mh_shard_unique_id(void * id128);
/* ncclCommInitRank on ctx's device; collective over all ranks. */
int mh_shard_comm_init_rccl(mh_ctx * ctx, const void * id128, int world, int rank, mh_shard_comm ** out);
/* Test transport: `world` ranks inside ONE process (one host thread per rank, all on one device); out_array[world]. */
int mh_shard_comm_init_local(int world, mh_shard_comm ** out_array);
void mh_shard_comm_destroy(mh_shard_comm * comm);
int mh_shard_comm_world(const mh_shard_comm * comm);
int mh_shard_comm_rank(const mh_shard_comm * comm);
const char * mh_shard_comm_backend(const mh_shard_comm * comm); /* "rccl" | "local" */
/* The communicator as the transport itself reports it: ranks_in_communicator = ncclCommCount (world for the local transport),
 * rccl_version = ncclGetVersion (0 for the local transport).  Either pointer may be NULL. */
int mh_shard_comm_info(const mh_shard_comm * comm, int * ranks_in_communicator, int * rccl_version);
/* ICPFactor ctor (geometric_factor.hpp:119-142) for this rank's share of the scan (any split; the first linearize routes
 * every point to the owner of its centre voxel).  points: host buffer, or device buffer when points_on_device != 0.
 * shard_map: this rank's shard (mh_map_insert_shard with the same world / rank / block_log2).  Collective. */
int mh_shard_icp_create(mh_ctx * ctx, mh_shard_comm * comm, mh_map * shard_map, const mh_point32 * points, size_t n_local,
                        int points_on_device, const mh_reg_config * cfg, int is_binary, const mh_shard_config * scfg, mh_shard_icp ** out);
/* ICPFactor::linearize of the WHOLE scan against the WHOLE map: every rank gets the same global result (H, b, f,
 * localizabilities, degeneracy info; 4-DoF / degeneracy projection applied once, on the global sums).  Collective; blocks. */
int mh_shard_icp_linearize(mh_shard_icp * icp, const double R_src[9], const double t_src[3], const double * R_tgt,
                           const double * t_tgt, const double g_unit[3], mh_icp_result * out);
/* Throughput forms.  The reference's second caller re-linearizes EVERY live ICPFactor per smoother update
 * (src/graph/manager.cpp:585-588); the unsharded path answers that with mh_icp_linearize_async / _batch, these are the
 * sharded counterparts.  A protocol ROUND carries any number of factors of one communicator and context: their movers
 * interleaved per peer in ONE ncclAllToAll, one route / append / K3b (/ K4b) / publish launch each, ONE ncclAllReduce of
 * n_factors x 168 doubles (+ one of n_factors x 16 when a factor's components are on).  _async / _batch_async only enqueue
 * (up to 32 rounds in flight per communicator; a further one first completes the oldest); mh_shard_icp_wait completes
 * every round in flight on the factor's communicator, in order, and fills their results — `out` must stay valid until
 * then.  All of them are collective: every rank issues the same sequence of calls.  A call whose movers did not fit the
 * per-peer segments is repeated (larger segments, same pose) at the next wait, or when its round leaves the ring — and
 * the calls of that factor enqueued behind it are repeated behind its repeat, so results and association cache end as
 * if the calls had run one after the other (the k-NN counters of repeated calls are statistics: a point may be counted
 * by the attempt and by the repeat).
 * Arrays are n_factors long: R_src[9 n], t_src[3 n], R_tgt / t_tgt (NULL without binary factors), g_unit[3 n], out[n]. */
int mh_shard_icp_linearize_async(mh_shard_icp * icp, const double R_src[9], const double t_src[3], const double * R_tgt,
                                 const double * t_tgt, const double g_unit[3], mh_icp_result * out);
int mh_shard_icp_linearize_batch(mh_shard_icp * const * icps, size_t n_factors, const double * R_src, const double * t_src,
                                 const double * R_tgt, const double * t_tgt, const double * g_unit, mh_icp_result * out); /* blocks */
int mh_shard_icp_linearize_batch_async(mh_shard_icp * const * icps, size_t n_factors, const double * R_src, const double * t_src,
                                       const double * R_tgt, const double * t_tgt, const double * g_unit, mh_icp_result * out);
int mh_shard_icp_wait(mh_shard_icp * icp);
int mh_shard_icp_reset(mh_shard_icp * icp);                       /* as mh_icp_reset; points stay where they are; stream-ordered */
int mh_shard_icp_set_components(mh_shard_icp * icp, int enabled); /* as mh_icp_set_components; must agree on all ranks */
/* Per-point state of the points this rank holds now (origin = first rank << 32 | index there); n_out alone may be asked for. */
int mh_shard_icp_get_state(mh_shard_icp * icp, uint64_t * origin, int32_t * status, double * means, double * normals,
                           size_t capacity, size_t * n_out);
int mh_shard_icp_stats(const mh_shard_icp * icp, mh_shard_stats * out);
void mh_shard_icp_destroy(mh_shard_icp * icp);

/* ---- diagnostics -------------------------------------------------------------------------------------------------------
 * With MH_ALLOC_CHECK=1 in the environment the device-allocation cache checks its hand-over rule (a block is re-used only
 * behind everything that was enqueued on it): freed blocks are poisoned behind their last use, verified when they are handed
 * out again.  blocks_verified / words_overwritten so far; MH_ERR_UNSUPPORTED when the check is off.  No reference counterpart. */
int mh_alloc_check_stats(unsigned long long * blocks_verified, unsigned long long * words_overwritten);

#ifdef __cplusplus
}
#endif
#endif /* MIMOSA_HIP_H */
