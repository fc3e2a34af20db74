"""CPU suite, world_size 2 over gloo: the map-sharding protocol as restated in tests/dist_harness.py — spatial-hash
partition with a one-voxel halo, all-to-all routing of the scan, all-reduce of the partial Hessians —
reproduces the unsharded result.  The CPU oracle stands in for the per-rank device backend here (it is
the checker; on GPUs the same layer drives mimosa_amd.capi)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


POSE_STEPS = [  # cumulative perturbations of the query pose: a cache-hit step, then centimetres / a degree away
    (np.zeros(3), np.array([0.004, 0.003, -0.002])),
    (np.array([0.0, 0.0, 0.012]), np.array([0.06, -0.05, 0.02])),
    (np.array([0.004, -0.003, 0.02]), np.array([-0.09, 0.11, 0.03])),
]


def _poses(R, t):
    from mimosa_amd import synth
    out = [(R, t)]
    for w, d in POSE_STEPS:
        out.append((R @ synth.so3_exp(w), t + d))
    return out


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import dist_harness as mdist
    from mimosa_amd import synth
    from oracle import ref_cpu

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    room = np.array([20.0, 14.0, 3.0])  # several 4 m shard blocks in x and y
    map_xyz = synth.make_room(4321, 0, 0, room=room)
    scan, aux = synth.make_scan(n_rows=32, seed=99, n_cols=128, room=room, sensor_local=np.array([9.3, 6.6, 1.2]))
    R, t = synth.query_pose(aux["R_W_L"], aux["t_W_L"])
    cfg = synth.enwide_config()

    def make_map():
        return ref_cpu.Map()

    def make_factor(m, pts):
        return ref_cpu.ICP(m, pts, ref_cpu.make_config(**cfg))

    sh = mdist.ShardedICP(dist.group.WORLD, make_map, make_factor, leaf=0.5)
    sh.build_map(np.array_split(map_xyz, 3))
    sh.set_scan(np.array_split(scan, world)[rank])  # every rank starts with a contiguous slice of the scan
    out = {}
    for k, (Rk, tk) in enumerate(_poses(R, t)):
        res = sh.linearize(Rk, tk)
        out.update({f"H{k}": res["H_ss"], f"b{k}": res["b_s"], f"f{k}": res["f"], f"hist{k}": res["status_hist"], f"n_knn{k}": res["n_knn"],
                    f"cq{k}": res["mean_candidates"], f"n_local{k}": res["n_local"], f"moved{k}": res["n_migrated_in"]})
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), map_points=sh.map.num_points, origin=sh.rec["origin"], status=sh.rec["status"],
             mean=sh.rec["mean"], **out)
    dist.barrier()
    dist.destroy_process_group()


def test_partition_properties():
    import dist_harness as mdist
    from mimosa_amd import synth
    m = synth.make_room(4321, 0, 0, room=np.array([20.0, 14.0, 3.0]))
    masks = [mdist.shard_insert_mask(m, 0.5, 4, r) for r in range(4)]
    assert np.all(np.sum(masks, axis=0) >= 1)          # every point lives somewhere
    own = mdist.owner_of_voxel(mdist.voxel_coords(m, 0.5), 4)
    for r in range(4):
        assert np.all(masks[r][own == r])              # owners hold their own voxels
        assert masks[r].sum() > (own == r).sum()       # ...plus a halo
    assert len(set(own.tolist())) == 4
    # the lattice colouring: neighbouring blocks never share a rank (world > 1), negative coordinates wrap like positive ones,
    # and the tables are the ones tools/lattice_table.py's rule gives (spot checks: the full search takes a quarter of a minute)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lattice_table as lt
    assert mdist.OWNER_A == lt.TABLE_A and mdist.OWNER_B == lt.TABLE_B
    for P in (2, 3, 8, 12):
        assert lt.pick(P) == (lt.TABLE_A[P], lt.TABLE_B[P])
    for world in (2, 3, 5, 8, 64):
        b = np.array([[-9, 2, -3], [7, 7, 7], [0, 0, 0]])
        o = mdist.owner_of_block(b, world)
        assert np.all((o >= 0) & (o < world))
        for d in ([1, 0, 0], [0, 1, 0], [0, 0, 1]):
            assert np.all(mdist.owner_of_block(b + np.array(d), world) != o)
        assert np.array_equal(mdist.owner_of_block(b + world * np.array([3, -5, 7]), world), o)


def test_sharded_equals_unsharded_world2(tmp_path):
    """Cold linearize, a cache-hit re-linearization and two re-linearizations centimetres / a degree away: at every pose
    the sharded factor (points re-routed with their data-association state) equals the unsharded one."""
    import torch.multiprocessing as mp
    from mimosa_amd import synth
    from oracle import ref_cpu
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity import rel

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))

    room = np.array([20.0, 14.0, 3.0])
    map_xyz = synth.make_room(4321, 0, 0, room=room)
    scan, aux = synth.make_scan(n_rows=32, seed=99, n_cols=128, room=room, sensor_local=np.array([9.3, 6.6, 1.2]))
    R, t = synth.query_pose(aux["R_W_L"], aux["t_W_L"])
    M = ref_cpu.Map()
    for c in np.array_split(map_xyz, 3):
        M.insert(c)
    F = ref_cpu.ICP(M, scan, ref_cpu.make_config(**synth.enwide_config()))
    moved_later = 0
    for k, (Rk, tk) in enumerate(_poses(R, t)):
        full = F.linearize(Rk, tk)
        for r in (r0, r1):  # both ranks hold the all-reduced result
            assert np.array_equal(r[f"hist{k}"], full["status_hist"]), k
            assert rel(r[f"H{k}"], full["H_ss"]) <= 1e-12 and rel(r[f"b{k}"], full["b_s"]) <= 1e-10, k
            assert abs(float(r[f"f{k}"]) - full["f"]) <= 1e-12 * full["f"]
            assert int(r[f"n_knn{k}"]) == full["n_knn"] and abs(float(r[f"cq{k}"]) - full["mean_candidates"]) < 1e-9
        assert int(r0[f"n_local{k}"]) + int(r1[f"n_local{k}"]) == len(scan) and min(int(r0[f"n_local{k}"]), int(r1[f"n_local{k}"])) > 0
        if k >= 2:
            moved_later += int(r0[f"moved{k}"]) + int(r1[f"moved{k}"])
    assert int(r0["n_knn1"]) == 0 and 0 < int(r0["n_knn2"]) < len(scan)   # step 1 hit the cache everywhere, step 2 did not
    assert moved_later > 0                                                 # points really changed owner between the poses
    # per-point state followed the points: the union of the ranks' records is the unsharded factor's state
    st, mean, _, _ = F.da_state()
    origin = np.concatenate([r0["origin"], r1["origin"]])
    split = np.array_split(np.arange(len(scan)), world)
    glob = np.array([split[int(o >> np.uint64(32))][int(o & np.uint64(0xFFFFFFFF))] for o in origin])
    assert sorted(glob.tolist()) == list(range(len(scan)))
    assert np.array_equal(np.concatenate([r0["status"], r1["status"]]), st[glob])
    assert np.abs(np.concatenate([r0["mean"], r1["mean"]]) - mean[glob]).max() <= 1e-12
    # each shard is smaller than the whole map but larger than half of it (halo)
    assert M.num_points // 2 < int(r0["map_points"]) < M.num_points
    assert int(r0["map_points"]) + int(r1["map_points"]) > M.num_points
