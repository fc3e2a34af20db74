"""CPU suite, world_size 2 over gloo: the map-sharding layer (mimosa_amd/dist.py) — spatial-hash
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


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from mimosa_amd import dist as mdist, synth
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
    batches = np.array_split(map_xyz, 3)
    sh.build_map(batches)
    mine = np.array_split(scan, world)[rank]  # every rank starts with a contiguous slice of the scan
    local = sh.scatter_scan(mine, R, t)
    res = sh.linearize(R, t)
    res2 = sh.linearize(R, t + np.array([0.004, 0.003, -0.002]))  # everything hits the DA cache
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), H=res["H_ss"], b=res["b_s"], f=res["f"], hist=res["status_hist"],
             n_knn=res["n_knn"], cq=res["mean_candidates"], n_local=res["n_local"], map_points=sh.map.num_points,
             H2=res2["H_ss"], n_knn2=res2["n_knn"], local_idx=local["idx"])
    dist.barrier()
    dist.destroy_process_group()


def test_partition_properties():
    from mimosa_amd import dist as mdist, synth
    m = synth.make_room(4321, 0, 0, room=np.array([20.0, 14.0, 3.0]))
    masks = [mdist.shard_insert_mask(m, 0.5, 4, r) for r in range(4)]
    assert np.all(np.sum(masks, axis=0) >= 1)          # every point lives somewhere
    own = mdist.owner_of_voxel(mdist.voxel_coords(m, 0.5), 4)
    for r in range(4):
        assert np.all(masks[r][own == r])              # owners hold their own voxels
        assert masks[r].sum() > (own == r).sum()       # ...plus a halo
    assert len(set(own.tolist())) == 4
    # negative coordinates hash like the reference's size_t arithmetic
    b = np.array([[-1, 2, -3]])
    with np.errstate(over="ignore"):
        h = (np.uint64(-1 & 0xFFFFFFFFFFFFFFFF) * mdist._P1) ^ (np.uint64(2) * mdist._P2) ^ (np.uint64(-3 & 0xFFFFFFFFFFFFFFFF) * mdist._P3)
    assert mdist.owner_of_block(b, 7)[0] == int(h % np.uint64(7))


def test_sharded_equals_unsharded_world2(tmp_path):
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
    full = ref_cpu.ICP(M, scan, ref_cpu.make_config(**synth.enwide_config())).linearize(R, t)

    for r in (r0, r1):  # both ranks hold the all-reduced result
        assert np.array_equal(r["hist"], full["status_hist"])
        assert rel(r["H"], full["H_ss"]) <= 1e-12 and rel(r["b"], full["b_s"]) <= 1e-10
        assert abs(float(r["f"]) - full["f"]) <= 1e-12 * full["f"]
        assert int(r["n_knn"]) == full["n_knn"] and abs(float(r["cq"]) - full["mean_candidates"]) < 1e-9
        assert int(r["n_knn2"]) == 0
    # the scan was really split across the ranks, nothing lost or duplicated, and each shard is smaller
    # than the whole map but larger than half of it (halo)
    assert int(r0["n_local"]) + int(r1["n_local"]) == len(scan) and min(int(r0["n_local"]), int(r1["n_local"])) > 0
    assert sorted(np.concatenate([r0["local_idx"], r1["local_idx"]]).tolist()) == sorted(scan["idx"].tolist())
    assert M.num_points // 2 < int(r0["map_points"]) < M.num_points
    assert int(r0["map_points"]) + int(r1["map_points"]) > M.num_points
