"""Multi-GPU layer for the geometric factor: map sharded by spatial hash, queries
routed to the owner of their centre voxel, partial Hessians combined with one small all-reduce.

Partition.  Voxels are grouped into shard blocks of 8 x 8 x 8 voxels (4 m cubes at the 0.5 m leaf);
block b is owned by rank  XORVector3iHash(b) mod P  (the reference's hash, include/mimosa/lidar/
utils.hpp:228-238).  Besides the voxels of its own blocks every rank also stores a ONE-VOXEL HALO: a
point whose voxel is adjacent (27-neighbourhood) to an owned block is inserted there too.  iVox's
insertion rule is per voxel and order-dependent only within a voxel, and every rank inserts its
subsequence in the original order, so a voxel has identical contents on every rank that stores it.
A query routed to the owner of its centre voxel therefore finds all 1/7/19/27 neighbour voxels
locally, bit-identically: the exchange step is one all-to-all(v) of the query points (16 B each), not
a per-neighbour candidate exchange, and interior memory overhead is the halo (~+40 % at 8^3 blocks on
planar content).

Collectives (torch.distributed; backend "nccl" == RCCL over xGMI on ROCm, "gloo" in the CPU tests):
  C1  all_to_all_single   source points -> owner ranks         (n x 32 B mh_point32 records)
  C3  all_reduce(SUM)     H (36) + b (6) + f (1) + status histogram (9) + counters (2) = 54 doubles
The component-localizability pass (geometric_factor.hpp:434-457) needs the eigenvectors of the GLOBAL
H, so it runs after C3 (the device path splits linearize at the K3 / K4 boundary — round 2).

This module is backend-agnostic: `make_map(points) -> map` and `make_factor(map, points) -> factor
with .linearize(R, t) -> dict` are injected (the HIP classes of mimosa_amd.capi on GPUs; the tests
inject the CPU oracle as the checker-side backend).
"""
from __future__ import annotations

import numpy as np

SHARD_BLOCK_LOG2 = 3
_P1, _P2, _P3 = np.uint64(9132043225175502913), np.uint64(7277549399757405689), np.uint64(6673468629021231217)


def voxel_coords(xyz, leaf: float) -> np.ndarray:
    """fast_floor(double(p) * inv_leaf) per axis (include/mimosa/lidar/utils.hpp:218-222)."""
    v = np.asarray(xyz, dtype=np.float64) * (1.0 / leaf)
    n = np.trunc(v)
    return (n - (v < n)).astype(np.int64)


def owner_of_block(b: np.ndarray, world: int) -> np.ndarray:
    b = np.asarray(b, dtype=np.int64).astype(np.uint64)  # two's complement, like the size_t casts
    with np.errstate(over="ignore"):
        h = (b[..., 0] * _P1) ^ (b[..., 1] * _P2) ^ (b[..., 2] * _P3)
    return (h % np.uint64(world)).astype(np.int64)


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
    """Owner rank of each source point = owner of the centre voxel of q = R p + t (fp64 like
    geometric_factor.hpp:276-277)."""
    q = np.asarray(pts_xyz_f32, np.float32).astype(np.float64) @ np.asarray(R, float).T + np.asarray(t, float)
    return owner_of_voxel(voxel_coords(q, leaf), world)


class ShardedICP:
    """One rank's view of a map-sharded scan-to-map factor."""

    def __init__(self, comm, make_map, make_factor, leaf: float):
        import torch.distributed as dist
        self.dist, self.comm = dist, comm
        self.rank, self.world = dist.get_rank(comm), dist.get_world_size(comm)
        self.make_map, self.make_factor, self.leaf = make_map, make_factor, leaf
        self.map = None
        self.factor = None
        self.n_local = 0

    def build_map(self, insert_batches):
        """insert_batches: iterable of float32 (n,3) arrays, identical on every rank (one iVox insert
        call each).  Each rank keeps only its shard + halo, in the original order."""
        self.map = self.make_map()
        self.halo_points = 0
        for xyz in insert_batches:
            xyz = np.ascontiguousarray(xyz, dtype=np.float32)
            m = shard_insert_mask(xyz, self.leaf, self.world, self.rank)
            self.map.insert(xyz[m])
        return self.map

    def scatter_scan(self, my_points, R, t, device=None):
        """C1: route this rank's slice of the scan to the owners; build the local factor."""
        import torch
        pts = np.ascontiguousarray(my_points)
        xyz = np.stack([pts["x"], pts["y"], pts["z"]], axis=1)
        own = query_owner(xyz, R, t, self.leaf, self.world)
        order = np.argsort(own, kind="stable")
        send_counts = np.bincount(own, minlength=self.world).astype(np.int64)
        send = torch.from_numpy(pts[order].view(np.uint8).reshape(-1, 32).copy())
        sc = torch.from_numpy(send_counts)
        rc = torch.empty_like(sc)
        if device is not None:
            send, sc, rc = send.to(device), sc.to(device), rc.to(device)
        self.dist.all_to_all_single(rc, sc, group=self.comm)
        recv = torch.empty((int(rc.sum().item()), 32), dtype=torch.uint8, device=send.device)
        self.dist.all_to_all_single(recv, send, output_split_sizes=rc.tolist(), input_split_sizes=sc.tolist(),
                                    group=self.comm)
        local = recv.cpu().numpy().reshape(-1).view(pts.dtype)
        self.n_local = len(local)
        self.factor = self.make_factor(self.map, local)
        return local

    def linearize(self, R, t, device=None):
        """Local linearize + C3 all-reduce of H, b, f, histogram and counters; when the backend offers the
        two-phase form (mh_icp_linearize_begin / _finish) the component localizabilities are computed in
        the GLOBAL eigenbasis and all-reduced too."""
        import torch
        two_phase = hasattr(self.factor, "linearize_begin")
        r = self.factor.linearize_begin(R, t) if two_phase else self.factor.linearize(R, t)
        n_knn = float(r["n_knn"])
        vec = np.concatenate([np.asarray(r["H_ss"], float).ravel(), np.asarray(r["b_s"], float), [float(r["f"])],
                              np.asarray(r["status_hist"], float), [n_knn, float(r["mean_candidates"]) * n_knn]])
        tv = torch.from_numpy(vec)
        if device is not None:
            tv = tv.to(device)
        self.dist.all_reduce(tv, op=self.dist.ReduceOp.SUM, group=self.comm)
        v = tv.cpu().numpy()
        extra = {}
        if two_phase:
            H = v[:36].reshape(6, 6)
            wr, Er = np.linalg.eigh(H[:3, :3])
            wt, Et = np.linalg.eigh(H[3:, 3:])
            tc, rc, _ = self.factor.linearize_finish(Er, Et)
            lv = torch.from_numpy(np.concatenate([tc, rc]))
            if device is not None:
                lv = lv.to(device)
            self.dist.all_reduce(lv, op=self.dist.ReduceOp.SUM, group=self.comm)
            lv = lv.cpu().numpy()
            with np.errstate(invalid="ignore"):
                extra = dict(loc_trans_comp=lv[:3], loc_rot_comp=lv[3:], eigvec_rot=Er, eigvec_trans=Et,
                             loc_rot_final=np.sqrt(wr), loc_trans_final=np.sqrt(wt))
        return dict(**extra, H_ss=v[:36].reshape(6, 6), b_s=v[36:42], f=float(v[42]), status_hist=v[43:52].astype(np.int64),
                    n_knn=int(v[52]), mean_candidates=(v[53] / v[52] if v[52] else 0.0), n_local=self.n_local)
