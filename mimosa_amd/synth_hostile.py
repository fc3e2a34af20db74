"""A second, hostile synthetic world for the geometric factor: the map is what a LiDAR pipeline would
actually have built — the union of past scans — instead of a uniformly sampled surface grid.

* rooms as in synth.py, but cluttered: boxes of all sizes on the floor, thin vertical plates (2-5 cm), poles (vertical
  cylinders down to 2 cm radius);
* the MAP is built by ray-casting full OS0-128 scans (real beam tables, range noise) from several past sensor poses per room
  and inserting the hit points, one insert per past scan: sampling density falls with 1 / r^2 and with the grazing angle, so
  voxels near a past pose saturate at the 20-point cap while far walls and shadowed regions stay sparse (queries there find
  their 5th neighbour a voxel away or not at all), and thin structures put two surfaces into one voxel;
* the query SCAN is cast the same way from a pose between the past ones.

Test / bench tooling only: deterministic (counter-based RNG of synth.py), nothing here is on the product path.
"""
from __future__ import annotations

import numpy as np

from . import os0_128, synth

ROOM = synth.ROOM
PATH_Y = 26.1          # the sensor travels along x at this room-local y (clutter keeps a corridor free)
CORRIDOR = 1.6
SENSOR_Z = 1.7


def _u(seed, stream, n, lo=0.0, hi=1.0):
    return lo + (hi - lo) * synth._uniform(seed, stream, np.arange(n, dtype=np.uint64))


def make_clutter(seed: int, rx: int, ry: int, n_boxes: int = 60, n_plates: int = 24, n_poles: int = 40, n_bushes: int = 24):
    """Room-local clutter of room (rx, ry): dict(boxes_lo, boxes_hi (m,3) incl. the thin plates; poles (k,5) = cx, cy, r, z0, z1;
    bush_lo, bush_hi (b,3): semi-transparent volumes — a ray that enters one is returned from a random depth inside it (mean
    free path BUSH_MFP), so their voxels fill up in all three dimensions and reach the 20-point cap, which a planar patch
    cannot (0.15 m minimum spacing: ~12 points per 0.5 m voxel of a wall))."""
    s = seed + 7919 * (rx * 1000 + ry)
    lo, hi = [], []

    def place(n, stream, size_lo, size_hi, height_lo, height_hi, thin=False):
        cx, cy = _u(s, stream, n, 2.0, ROOM[0] - 2.0), _u(s, stream + 1, n, 2.0, ROOM[1] - 2.0)
        sx, sy = _u(s, stream + 2, n, size_lo, size_hi), _u(s, stream + 3, n, size_lo, size_hi)
        h = _u(s, stream + 4, n, height_lo, height_hi)
        if thin:  # a plate: one footprint side is 2-5 cm
            t = _u(s, stream + 5, n, 0.02, 0.05)
            flip = _u(s, stream + 6, n) < 0.5
            sx, sy = np.where(flip, t, sx), np.where(flip, sy, t)
        for i in range(n):
            if abs(cy[i] - PATH_Y) < CORRIDOR + 0.5 * sy[i]:
                cy[i] = PATH_Y + np.sign(cy[i] - PATH_Y + 1e-9) * (CORRIDOR + 0.5 * sy[i] + 0.2)
            lo.append([cx[i] - 0.5 * sx[i], cy[i] - 0.5 * sy[i], 0.0])
            hi.append([cx[i] + 0.5 * sx[i], cy[i] + 0.5 * sy[i], h[i]])

    place(n_boxes, 100, 0.3, 4.0, 0.3, 6.0)
    place(n_plates, 200, 1.0, 6.0, 1.5, 8.0, thin=True)
    n_solid = len(lo)
    place(n_bushes, 400, 1.0, 4.0, 1.0, 5.0)
    px, py = _u(s, 300, n_poles, 1.5, ROOM[0] - 1.5), _u(s, 301, n_poles, 1.5, ROOM[1] - 1.5)
    pr = np.exp(_u(s, 302, n_poles, np.log(0.02), np.log(0.3)))
    ph = _u(s, 303, n_poles, 1.0, 12.0)
    near = np.abs(py - PATH_Y) < CORRIDOR + pr
    py = np.where(near, PATH_Y + np.sign(py - PATH_Y + 1e-9) * (CORRIDOR + pr + 0.2), py)
    return dict(boxes_lo=np.array(lo[:n_solid]), boxes_hi=np.array(hi[:n_solid]), bush_lo=np.array(lo[n_solid:]), bush_hi=np.array(hi[n_solid:]),
                poles=np.stack([px, py, pr, np.zeros(n_poles), ph], 1))


BUSH_MFP = 0.5


def raycast(o: np.ndarray, d: np.ndarray, clutter: dict, room=ROOM, seed: int = 0) -> np.ndarray:
    """Distance along every ray (room-local origin o (n,3), unit direction d (n,3)) to the first return: room walls from
    the inside, boxes / plates / poles from the outside, a random depth inside a bush."""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t = np.where(d > 0, (room - o) * inv, np.where(d < 0, -o * inv, np.inf)).min(axis=1)
        for lo, hi in zip(clutter["boxes_lo"], clutter["boxes_hi"]):
            t1, t2 = (lo - o) * inv, (hi - o) * inv
            tn, tf = np.minimum(t1, t2), np.maximum(t1, t2)
            tn = np.where(np.isnan(tn), -np.inf, tn).max(axis=1)
            tf = np.where(np.isnan(tf), np.inf, tf).min(axis=1)
            hit = (tf >= tn) & (tn > 1e-6)
            t = np.where(hit & (tn < t), tn, t)
        idx = np.arange(len(o), dtype=np.uint64)
        for b, (lo, hi) in enumerate(zip(clutter["bush_lo"], clutter["bush_hi"])):
            t1, t2 = (lo - o) * inv, (hi - o) * inv
            tn, tf = np.minimum(t1, t2), np.maximum(t1, t2)
            tn = np.maximum(np.where(np.isnan(tn), -np.inf, tn).max(axis=1), 0.0)
            tf = np.where(np.isnan(tf), np.inf, tf).min(axis=1)
            depth = tn - BUSH_MFP * np.log(1.0 - synth._uniform(seed, 500 + b, idx))
            hit = (tf > tn) & (depth < tf) & (depth > 1e-6)
            t = np.where(hit & (depth < t), depth, t)
        for cx, cy, r, z0, z1 in clutter["poles"]:
            ox, oy = o[:, 0] - cx, o[:, 1] - cy
            a = d[:, 0] ** 2 + d[:, 1] ** 2
            b = 2.0 * (ox * d[:, 0] + oy * d[:, 1])
            c = ox * ox + oy * oy - r * r
            disc = b * b - 4.0 * a * c
            tc = (-b - np.sqrt(np.where(disc > 0, disc, np.nan))) / (2.0 * a)
            z = o[:, 2] + tc * d[:, 2]
            hit = (disc > 0) & (tc > 1e-6) & (z >= z0) & (z <= z1)
            t = np.where(hit & (tc < t), tc, t)
    return t


def path_pose(rx: int, ry: int, s: float):
    """Sensor pose (R_W_L, t_W_L) at parameter s in [0, 1] along the corridor of room (rx, ry): drives along x, weaves a
    little in y / z, yaws slowly."""
    x = 6.0 + s * (ROOM[0] - 12.0)
    local = np.array([x, PATH_Y + 0.4 * np.sin(7.0 * s), SENSOR_Z + 0.15 * np.sin(11.0 * s)])
    return synth.rot_z(0.3 + 1.3 * s), synth.room_origin(rx, ry) + local


def cast_scan(R_W_L, t_W_L, clutter, rx, ry, seed: int, n_rows: int = 128, n_cols: int = synth.N_COLS, sigma: float = 0.02):
    """One deskewed OS0-128 scan from (R_W_L, t_W_L): points[POINT_DTYPE] in the sensor frame (raw index order) + world hits."""
    alt = np.deg2rad(os0_128.altitude_angles(n_rows))
    shift = os0_128.pixel_shifts(n_rows, n_cols).astype(np.int64)
    az = -2.0 * np.pi * (np.arange(n_cols) / n_cols)
    rows, cols = np.meshgrid(np.arange(n_rows), np.arange(n_cols), indexing="ij")
    rows, cols = rows.ravel(), cols.ravel()
    look = (cols + shift[rows]) % n_cols
    ca, sa, cz, sz = np.cos(alt[rows]), np.sin(alt[rows]), np.cos(az[look]), np.sin(az[look])
    d_s = np.stack([ca * cz, ca * sz, sa], 1)
    o_s = np.stack([synth.BEAM_ORIGIN_M * cz, synth.BEAM_ORIGIN_M * sz, np.zeros_like(cz)], 1)
    org = synth.room_origin(rx, ry)
    o_w = t_W_L + o_s @ R_W_L.T
    d_w = d_s @ R_W_L.T
    rng = raycast(o_w - org, d_w, clutter, seed=seed)
    n = rows.size
    rng = rng + synth._normal(seed, 7, np.arange(n, dtype=np.uint64)) * sigma
    p_s = o_s + rng[:, None] * d_s
    pts = np.zeros(n, dtype=synth.POINT_DTYPE)
    pts["x"], pts["y"], pts["z"] = p_s[:, 0], p_s[:, 1], p_s[:, 2]
    pts["intensity"] = 100.0
    pts["t"] = (cols * synth.COL_NS * (synth.N_COLS // n_cols)).astype(np.uint32)
    pts["idx"] = (rows * n_cols + cols).astype(np.uint32)
    pts["range"] = np.sqrt(pts["x"] ** 2 + pts["y"] ** 2 + pts["z"] ** 2)
    hits_w = (o_w + rng[:, None] * d_w).astype(np.float32)
    return pts, hits_w


def _one_map_scan(task):
    rx, ry, k, poses_per_room, seed, n_rows = task
    cl = make_clutter(seed, rx, ry)
    R, t = path_pose(rx, ry, (k + 0.5) / poses_per_room)
    _, hits = cast_scan(R, t, cl, rx, ry, seed + 31 * (rx * 100 + ry) + k, n_rows=n_rows)
    return hits


def make_map_scans(n_rx: int, n_ry: int, poses_per_room: int = 5, seed: int = synth.BASE_SEED + 500, n_rows: int = 128, workers: int = 0):
    """Yield (rx, ry, k, float32 (n,3) world points) — one past scan each = one iVox insert() call, in the order a robot
    driving through the rooms would have made them.  workers > 1: the scans are ray-cast by a pool of spawned processes
    (a scan is ~1.5 s of numpy on one core); the order of the results is the same."""
    tasks = [(rx, ry, k, poses_per_room, seed, n_rows) for rx in range(n_rx) for ry in range(n_ry) for k in range(poses_per_room)]
    if workers > 1:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(max_workers=min(workers, len(tasks)), mp_context=mp.get_context("spawn")) as ex:
            for task, hits in zip(tasks, ex.map(_one_map_scan, tasks)):
                yield task[0], task[1], task[2], hits
    else:
        for task in tasks:
            yield task[0], task[1], task[2], _one_map_scan(task)


def make_query_scan(s: float = 0.37, rx: int = 0, ry: int = 0, seed: int = synth.BASE_SEED + 500, n_rows: int = 128, n_cols: int = synth.N_COLS):
    """The scan to register: cast from a pose BETWEEN the past ones.  Returns (points, aux with the ground-truth pose)."""
    cl = make_clutter(seed, rx, ry)
    R, t = path_pose(rx, ry, s)
    pts, _ = cast_scan(R, t, cl, rx, ry, seed + 999_983 + int(round(s * 1e6)), n_rows=n_rows, n_cols=n_cols)
    return pts, {"R_W_L": R, "t_W_L": t}


def candidate_stats(map_xyz: np.ndarray, q_world: np.ndarray, leaf: float = 0.5, mode: int = 19) -> dict:
    """What the reference's k-NN scans per query: map points (and 16-byte quads of four) in the occupied voxels of the
    1 / 7 / 19 / 27-neighbourhood of the query's voxel (gtsam_points' offsets).  Percentiles over the queries — the tail
    is what sets the dominant kernel's slowest wave."""
    def key(v):
        return (v[:, 0] + (1 << 20)) | ((v[:, 1] + (1 << 20)) << 21) | ((v[:, 2] + (1 << 20)) << 42)
    def xyz(a):  # plain (n, >= 3) arrays and the library's point records alike
        a = np.asarray(a)
        return np.stack([a["x"], a["y"], a["z"]], 1).astype(np.float64) if a.dtype.names else np.asarray(a, np.float64)[:, :3]
    map_xyz, q_world = xyz(map_xyz), xyz(q_world)
    mv = np.floor(np.asarray(map_xyz, np.float64) / leaf).astype(np.int64)
    keys, counts = np.unique(key(mv), return_counts=True)
    qv = np.floor(np.asarray(q_world, np.float64) / leaf).astype(np.int64)
    offs = [(i, j, k) for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)
            if abs(i) + abs(j) + abs(k) <= {1: 0, 7: 1, 19: 2, 27: 3}[mode]]
    cand = np.zeros(len(qv), np.int64)
    quads = np.zeros(len(qv), np.int64)
    for o in offs:
        kq = key(qv + np.array(o, np.int64))
        at = np.minimum(np.searchsorted(keys, kq), len(keys) - 1)
        c = np.where(keys[at] == kq, counts[at], 0)
        cand += c
        quads += (c + 3) // 4
    pc = lambda x, q: float(np.percentile(x, q))
    return dict(mean=float(cand.mean()), p50=pc(cand, 50), p90=pc(cand, 90), p99=pc(cand, 99), max=int(cand.max()),
                quads_p50=pc(quads, 50), quads_p90=pc(quads, 90), quads_p99=pc(quads, 99), quads_max=int(quads.max()),
                empty_centre_voxel_share=float((np.where(keys[np.minimum(np.searchsorted(keys, key(qv)), len(keys) - 1)] == key(qv), 1, 0) == 0).mean()))


def voxel_fill_stats(map_xyz: np.ndarray, leaf: float = 0.5, cap: int = 20) -> dict:
    """How full the map's voxels are (from mh_map_get_cloud): points per occupied voxel and the share at the cap."""
    v = np.floor(np.asarray(map_xyz, np.float64) / leaf).astype(np.int64)
    key = (v[:, 0] + (1 << 20)) | ((v[:, 1] + (1 << 20)) << 21) | ((v[:, 2] + (1 << 20)) << 42)
    _, counts = np.unique(key, return_counts=True)
    return dict(n_voxels=int(len(counts)), points_per_voxel_mean=float(counts.mean()), points_per_voxel_p50=float(np.percentile(counts, 50)),
                points_per_voxel_p99=float(np.percentile(counts, 99)), share_at_cap=float((counts >= cap).mean()),
                share_below_5=float((counts < 5).mean()))
