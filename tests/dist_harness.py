"""Test harness (NOT product code): a host-staged, backend-agnostic restatement of the map-sharding protocol of SURVEY.md 8(e) over
torch.distributed — what the CPU suite runs at world size 2 over gloo with the oracle standing in for the device backend
(tests/test_dist_cpu.py).  The product's multi-GPU path is the NATIVE one behind the C ABI (mh_map_insert_shard + mh_shard_*,
mimosa_amd/csrc/shard_api.hip: RCCL inside the library); until round 5 this file lived in the package next to a device form that
drove caller-side collectives through C entry points which ABI version 2 no longer has.

Partition.  Voxels are grouped into shard blocks of 8 x 8 x 8 voxels (4 m cubes at the 0.5 m leaf); block b is owned by
rank  XORVector3iHash(b) mod P  (the reference's hash, include/mimosa/lidar/utils.hpp:228-238).  Besides the voxels of its
own blocks every rank also stores a ONE-VOXEL HALO: a point whose voxel is adjacent (27-neighbourhood) to an owned block is
inserted there too.  iVox's insertion rule is per voxel and order-dependent only within a voxel, and every rank inserts its
subsequence in the original order, so a voxel has identical contents on every rank that stores it.  A query on the owner
of its centre voxel therefore finds all 1/7/19/27 neighbour voxels locally, bit-identically.

Per linearize (the pose changes between Gauss-Newton iterations, so does the owner of points near block borders):
  plan     owner of every local point at the new pose -> counts                   C0  all_to_all of the counts (P int64)
  migrate  points that changed owner leave WITH their data-association state      C1  all_to_all(v) of the records
  K3       local linearize up to the raw Hessian sums                             C3a all_reduce(SUM) of 32 doubles
  K4       component localizabilities in the eigenbasis of the GLOBAL H           C3b all_reduce(SUM) of 16 doubles
"""
from __future__ import annotations

import ctypes as C

import numpy as np

SHARD_BLOCK_LOG2 = 3
RECORD_BYTES = 112
OWNER_A = [0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 3, 3, 3, 4, 3, 3, 5, 7, 5, 6, 5, 4, 7, 5, 6, 6, 5, 5, 5, 5, 5, 6, 6, 6, 5, 5, 7, 6, 7, 5, 7, 6, 7, 8, 7, 6, 7, 7, 9, 8, 8, 7, 7, 10, 10, 7, 7, 8, 7, 8, 8, 8, 8]
OWNER_B = [0, 0, 1, 1, 2, 2, 3, 3, 3, 3, 3, 3, 4, 4, 5, 6, 4, 5, 7, 8, 8, 8, 8, 10, 9, 7, 11, 8, 12, 11, 12, 6, 7, 14, 13, 10, 6, 7, 11, 16, 15, 6, 12, 7, 13, 17, 10, 14, 18, 11, 11, 14, 23, 22, 16, 12, 21, 8, 9, 13, 11, 11, 17, 14, 19]


def voxel_coords(xyz, leaf: float) -> np.ndarray:
    """fast_floor(double(p) * inv_leaf) per axis (include/mimosa/lidar/utils.hpp:218-222)."""
    v = np.asarray(xyz, dtype=np.float64) * (1.0 / leaf)
    n = np.trunc(v)
    return (n - (v < n)).astype(np.int64)


def owner_of_block(b: np.ndarray, world: int) -> np.ndarray:
    """The library's owner function (mimosa_amd/csrc/shard_kernels.hip owner_of_block): a lattice colouring of the block grid,
    rank = (bx + A[world] by + B[world] bz) mod world, tables from tools/lattice_table.py."""
    b = np.asarray(b, dtype=np.int64)
    return np.mod(b[..., 0] + OWNER_A[world] * b[..., 1] + OWNER_B[world] * b[..., 2], world).astype(np.int64)


def owner_of_voxel(v: np.ndarray, world: int) -> np.ndarray:
    return owner_of_block(np.asarray(v, np.int64) >> SHARD_BLOCK_LOG2, world)


def shard_insert_mask(xyz, leaf: float, world: int, rank: int) -> np.ndarray:
    """Which map points rank `rank` must insert: voxels of owned blocks plus the one-voxel halo."""
    v = voxel_coords(xyz, leaf)
    need = np.zeros(len(v), dtype=bool)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                need |= owner_of_voxel(v + np.array([dx, dy, dz]), world) == rank
    return need


def query_owner(pts_xyz_f32, R, t, leaf: float, world: int) -> np.ndarray:
    """Owner rank of each source point = owner of the centre voxel of q = R p + t (fp64, the reference's evaluation order
    r0 x + (r1 y + r2 z), geometric_factor.hpp:276-277)."""
    p = np.asarray(pts_xyz_f32, np.float32).astype(np.float64)
    R, t = np.asarray(R, float), np.asarray(t, float)
    q = np.stack([(R[k, 0] * p[:, 0] + (R[k, 1] * p[:, 1] + R[k, 2] * p[:, 2])) + t[k] for k in range(3)], 1)
    return owner_of_voxel(voxel_coords(q, leaf), world)


# ---- collectives that work for device tensors over RCCL and for the gloo test backend -------------------------------
def _staged(dist, comm, t):
    return dist.get_backend(comm) == "gloo" and t.is_cuda


def exchange_counts(dist, comm, send_counts, device):
    """All ranks' send-count vectors (world x world, row = sender).  Returns (recv_counts of this rank, total number of
    records moving anywhere): the total is what decides whether the record exchange — a collective every rank must enter
    or skip TOGETHER — runs at all; a rank's own row and column alone cannot tell (A idle while B sends to C)."""
    import torch
    world, rank = dist.get_world_size(comm), dist.get_rank(comm)
    m = torch.zeros((world, world), dtype=torch.int64)
    m[rank] = torch.as_tensor(np.asarray(send_counts, np.int64))
    if dist.get_backend(comm) != "gloo":
        m = m.to(device)
    dist.all_reduce(m, op=dist.ReduceOp.SUM, group=comm)   # every rank contributes its own row: the cheapest collective there is
    m = m.cpu().numpy()
    return m[:, rank].copy(), int(m.sum())


def all_to_all_rows(dist, comm, send, send_counts, recv_counts):
    """send: (sum(send_counts), W) uint8 tensor grouped by destination -> (sum(recv_counts), W) tensor on the same device."""
    import torch
    stage = _staged(dist, comm, send)
    s = send.cpu() if stage else send
    r = torch.empty((int(np.sum(recv_counts)), send.shape[1]), dtype=send.dtype, device=s.device)
    dist.all_to_all_single(r, s, output_split_sizes=[int(c) for c in recv_counts], input_split_sizes=[int(c) for c in send_counts], group=comm)
    return r.to(send.device) if stage else r


def all_reduce_sum(dist, comm, t):
    if _staged(dist, comm, t):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=comm)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=comm)
    return t


class ShardedICP:
    """Host-staged form of the same protocol, backend-agnostic: `make_map() -> map` and `make_factor(map, points) -> factor`
    are injected; the factor must offer linearize(R, t), da_state() and set_da_state() (the CPU oracle does: it is the
    checker-side stand-in for the device backend in the gloo tests).  Only configurations without the unary epilogue
    (reg_4_dof, project_on_degneneracy) are supported here: this path all-reduces the shard results as they come."""

    def __init__(self, comm, make_map, make_factor, leaf: float):
        import torch.distributed as dist
        self.dist, self.comm = dist, comm
        self.rank, self.world = dist.get_rank(comm), dist.get_world_size(comm)
        self.make_map, self.make_factor, self.leaf = make_map, make_factor, leaf
        self.map = None
        self.rec = None
        self.lin_count = 0

    def build_map(self, insert_batches):
        self.map = self.make_map()
        for xyz in insert_batches:
            xyz = np.ascontiguousarray(xyz, dtype=np.float32)
            self.map.insert(xyz[shard_insert_mask(xyz, self.leaf, self.world, self.rank)])
        return self.map

    def set_scan(self, my_points):
        pts = np.ascontiguousarray(my_points)
        n = len(pts)
        self.rec = dict(pts=pts, status=np.zeros(n, np.int32), mean=np.zeros((n, 3)), normal=np.zeros((n, 3)), q_da=np.zeros((n, 3)),
                        origin=(np.uint64(self.rank) << np.uint64(32)) | np.arange(n, dtype=np.uint64))

    def _migrate(self, R, t):
        import torch
        r = self.rec
        xyz = np.stack([r["pts"]["x"], r["pts"]["y"], r["pts"]["z"]], 1)
        own = query_owner(xyz, R, t, self.leaf, self.world)
        go = own != self.rank
        order = np.argsort(own[go], kind="stable")
        sc = np.bincount(own[go], minlength=self.world).astype(np.int64)
        rc, _ = exchange_counts(self.dist, self.comm, sc, None)
        packed = np.concatenate([r["pts"][go].view(np.uint8).reshape(-1, 32), r["status"][go].view(np.uint8).reshape(-1, 4),
                                 r["mean"][go].view(np.uint8).reshape(-1, 24), r["normal"][go].view(np.uint8).reshape(-1, 24),
                                 r["q_da"][go].view(np.uint8).reshape(-1, 24), r["origin"][go].view(np.uint8).reshape(-1, 8)], 1)[order]
        recv = all_to_all_rows(self.dist, self.comm, torch.from_numpy(np.ascontiguousarray(packed)), sc, rc).numpy()
        keep = ~go
        f = lambda a, dt, w: np.ascontiguousarray(a).view(dt).reshape(-1, w) if w > 1 else np.ascontiguousarray(a).view(dt).reshape(-1)
        self.rec = dict(
            pts=np.concatenate([r["pts"][keep], f(recv[:, :32], r["pts"].dtype, 1)]),
            status=np.concatenate([r["status"][keep], f(recv[:, 32:36], np.int32, 1)]),
            mean=np.concatenate([r["mean"][keep], f(recv[:, 36:60], np.float64, 3)]),
            normal=np.concatenate([r["normal"][keep], f(recv[:, 60:84], np.float64, 3)]),
            q_da=np.concatenate([r["q_da"][keep], f(recv[:, 84:108], np.float64, 3)]),
            origin=np.concatenate([r["origin"][keep], f(recv[:, 108:116], np.uint64, 1)]))
        return int(rc.sum())

    def linearize(self, R, t):
        import torch
        n_in = self._migrate(R, t)
        r = self.rec
        fac = self.make_factor(self.map, r["pts"])
        fac.set_da_state(r["status"], r["mean"], r["normal"], r["q_da"], self.lin_count)
        res = fac.linearize(R, t)
        self.lin_count = int(res["linearize_count"])
        r["status"], r["mean"], r["normal"], r["q_da"] = fac.da_state()
        n_knn = float(res["n_knn"])
        vec = np.concatenate([np.asarray(res["H_ss"], float).ravel(), np.asarray(res["b_s"], float), [float(res["f"])],
                              np.asarray(res["status_hist"], float), [n_knn, float(res["mean_candidates"]) * n_knn]])
        tv = torch.from_numpy(vec)
        self.dist.all_reduce(tv, op=self.dist.ReduceOp.SUM, group=self.comm)
        v = tv.numpy()
        return dict(H_ss=v[:36].reshape(6, 6), b_s=v[36:42], f=float(v[42]), status_hist=v[43:52].astype(np.int64), n_knn=int(v[52]),
                    mean_candidates=(v[53] / v[52] if v[52] else 0.0), n_local=len(r["pts"]), n_migrated_in=n_in)
