// Map-sharded factor: structs and launcher declarations shared by shard_kernels.hip and the C ABI.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "icp_device.hpp"

namespace mh
{
struct ShardPose
{
  double R[9], t[3];  // delta pose T_tgt^-1 * T_src of the factor
};

// the per-point arrays of an ICP factor that travel with a point
struct ShardArrays
{
  float4 * src;
  double * q_da;
  double * mean;
  double * normal;
  int32_t * status;
  unsigned long long * origin;  // (rank that first held the point) << 32 | its index there
};

// what a point that changes owner takes along: 112 bytes
struct ShardRecord
{
  float4 src;
  double q_da[3], mean[3], normal[3];
  int32_t status;
  uint32_t pad;
  unsigned long long origin;
};
static_assert(sizeof(ShardRecord) == 112, "ShardRecord layout");

// ---- native sharded factor (shard_api.hip): slots, tombstones, fixed-capacity exchange ------------------------------
// A sharded factor's per-point arrays are SLOTS: a point that leaves for another rank becomes a tombstone (origin == kShardTomb,
// status == -1, which carries kShardSkip), arrivals are appended behind the last slot.  Nothing about the exchange is
// known to the host in advance: every rank sends every peer one fixed-size segment — a 16-byte header with the record
// count, then up to `cap` records — so the collective needs no counts, and the slot count of the factor lives on the
// device (ping-pong entries: the kernels of call c read [c & 1] and write [(c + 1) & 1]).
constexpr unsigned long long kShardTomb = ~0ull;
constexpr int kShardMaxWorld = 64;
// The owner function's two tables (index = world size; tools/lattice_table.py derives and prints them): rank of block
// (bx, by, bz) = (bx + A[P] by + B[P] bz) mod P.  One initialiser for the device copy (shard_kernels.hip) and the host twin
// (mh_shard_owner_of_block, shard_api.hip).
#define MH_SHARD_OWNER_A {0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 4, 3, 3, 5, 7, 5, 6, 5, 4, 7, 5, 6, 6, 5, 5, 5, 5, 5, 6, 6, 6, 5, 5, 7, 6, 7, 5, 7, 6, 7, 8, 7, 6, 7, 7, 9, 8, 8, 7, 7, 10, 10, 7, 7, 8, 7, 8, 8, 8, 8}
#define MH_SHARD_OWNER_B {0, 0, 1, 1, 2, 2, 3, 3, 3, 3, 3, 3, 4, 4, 5, 6, 4, 5, 7, 8, 8, 8, 8, 10, 9, 7, 11, 8, 12, 11, 12, 6, 7, 14, 13, 10, 6, 7, 11, 16, 15, 6, 12, 7, 13, 17, 10, 14, 18, 11, 11, 14, 23, 22, 16, 12, 21, 8, 9, 13, 11, 11, 17, 14, 19}
constexpr int kShardSums = 104;                          // all-reduce vector: 95 sums / counters of a binary factor, padded
constexpr int kShardArLen = kShardSums + kShardMaxWorld;  // ... + one slot per rank (each rank fills its own): max movers per destination
struct ShardState
{
  uint32_t n_slots[2];  // slots in use, tombstones included
  uint32_t n_live[2];   // points this rank holds
  uint32_t sent_total;  // records that left in the current call
  uint32_t max_total;   // max over destinations of the points that WANT to leave (may exceed the segment capacity)
  uint32_t error;       // bit 0: arrivals beyond the slot capacity were dropped
  uint32_t pad;
};
struct ShardHdr
{
  uint32_t sent, total, pad0, pad1;  // records in this segment; points bound for this peer (> sent: capacity overflow)
};
static_assert(sizeof(ShardHdr) == 16, "segment header");
// what the last kernel of a collective call leaves in mapped pinned host memory
struct ShardPublish
{
  double ar[kShardArLen];  // all-reduced sums, counters, per-rank mover maxima
  double loc[16];          // all-reduced component localizabilities + status histogram
  uint32_t n_slots, n_live, error, max_total;
  uint32_t seq, pad[3];
};
__host__ __device__ inline size_t shard_segment_bytes(uint32_t cap) { return sizeof(ShardHdr) + static_cast<size_t>(cap) * 112; }

// One factor of a protocol round, as the route / append kernels see it.  A round of several factors (the live factors of the
// smoother window, src/graph/manager.cpp:585-588) shares ONE send and ONE receive buffer: peer p's block holds the factors'
// segments one after the other, so `send` / `recv` point at this factor's segment inside peer 0's block and `peer_stride`
// is the size of a whole peer block (= shard_segment_bytes(cap) for a round of one).
struct ShardFactorArgs
{
  ShardPose P;
  ShardArrays a;
  ShardState * st;
  double inv_leaf;
  uint8_t * dest;     // per slot: destination rank, 0xFF = stays
  uint32_t * hist;    // per 256-slot block and destination: movers
  char * send;
  const char * recv;
  size_t peer_stride;
  double * ar_slots;  // this factor's per-rank slots of the all-reduce vector (kShardSums ...)
  uint32_t cap, slot_capacity;
  int cur, log2;
};
constexpr int kShardBatchMax = 8;  // factors per batched launch (the argument blocks travel in the kernel-argument segment)
struct ShardBatch
{
  ShardFactorArgs f[kShardBatchMax];
  int route_start[kShardBatchMax + 1];   // exclusive prefix of the factors' route grids (256 slots per workgroup, >= 1 each)
  int append_start[kShardBatchMax + 1];  // ... and of their append grids (world * cap threads)
  int n;
  uint32_t world, rank;
};
struct ShardPublishBatch
{
  const ShardState * st[kShardBatchMax];
  int next[kShardBatchMax];
  ShardPublish * host;  // n consecutive entries of the round's publish slot
  const double * ar;    // n x kShardArLen
  const double * loc;   // n x 16, or null
  uint32_t seq;
  int n;
};

hipError_t launch_shard_state_init(ShardState * st, uint32_t n, hipStream_t stream);
// owner of every live slot at this pose -> dest[] (0xFF: stays / tombstone), per-block per-destination counts; then the
// movers' records into the per-peer segments in slot order (stable), tombstones behind them; the last block writes the
// segment headers, the state's sent / max counters and this rank's slot of the all-reduce vector
hipError_t launch_shard_route(const ShardFactorArgs & f, uint32_t n_bound, uint32_t world, uint32_t rank, hipStream_t stream);
// arrivals of all peers appended behind the last slot, in (peer, record) order; writes the next ping-pong entries
hipError_t launch_shard_append(const ShardFactorArgs & f, uint32_t world, hipStream_t stream);
// the same for up to kShardBatchMax factors per launch; shard_batch_grids fills the two prefix tables from the slot bounds
void shard_batch_grids(ShardBatch & b, const uint32_t * n_bound);
hipError_t launch_shard_route_batch(const ShardBatch & b, hipStream_t stream);
hipError_t launch_shard_append_batch(const ShardBatch & b, hipStream_t stream);
hipError_t launch_shard_publish_batch(const ShardPublishBatch & b, hipStream_t stream);
// stable compaction of the live slots into `out` (tombstones dropped); both ping-pong entries become n_live
hipError_t launch_shard_compact(const ShardArrays & in, const ShardArrays & out, ShardState * st, int cur, uint32_t n_bound, uint32_t * flags, uint32_t * pos,
                                void * temp, size_t temp_bytes, hipStream_t stream);
// forget every data association (== freshly constructed factor state) of the points held; tombstones stay
hipError_t launch_shard_reset(const ShardArrays & a, const ShardState * st, int cur, uint32_t n_bound, hipStream_t stream);
hipError_t launch_shard_publish(const double * ar, const double * loc, const ShardState * st, int next, ShardPublish * host, uint32_t seq, hipStream_t stream);

size_t shard_temp_bytes(size_t n);
hipError_t launch_shard_filter(const float * xyz, uint32_t n, uint32_t stride, double inv_leaf, uint32_t world, uint32_t rank, int log2,
                               uint32_t * flags, uint32_t * pos, float * out, uint32_t * n_out, void * temp, size_t temp_bytes, hipStream_t stream);
hipError_t launch_shard_origin(unsigned long long * origin, uint32_t n, uint32_t rank, hipStream_t stream);

}  // namespace mh
