"""Shared body of the GPU sharding workers (tests/dist_gpu_worker.py: world 1 over RCCL; dist_gpu_worker2.py: world 2 over
gloo, both ranks on the one GPU of the test box): ShardedICPDevice — map shard built by mh_map_insert_shard, points routed
with their association state by mh_icp_shard_*, two-phase linearize on device buffers — against the UNSHARDED oracle at a
cold pose, a cache-hit pose and two poses centimetres / a degree away."""
import numpy as np

POSE_STEPS = [
    (np.zeros(3), np.array([0.004, 0.003, -0.002])),
    (np.array([0.0, 0.0, 0.012]), np.array([0.06, -0.05, 0.02])),
    (np.array([0.004, -0.003, 0.02]), np.array([-0.09, 0.11, 0.03])),
]


def run(dist, ctx, device, cfg_over=None):
    from mimosa_amd import capi, dist as mdist, synth
    from oracle import ref_cpu
    from parity import assert_result_parity

    rank, world = dist.get_rank(), dist.get_world_size()
    room = np.array([20.0, 14.0, 3.0])
    map_xyz = synth.make_room(4321, 0, 0, room=room)
    scan, aux = synth.make_scan(n_rows=32, seed=99, n_cols=128, room=room, sensor_local=np.array([9.3, 6.6, 1.2]))
    R, t = synth.query_pose(aux["R_W_L"], aux["t_W_L"])
    cfg = dict(synth.enwide_config(), **(cfg_over or {}))
    sh = mdist.ShardedICPDevice(dist.group.WORLD, ctx, 0.5, capi.make_reg_config(**cfg), device)
    sh.build_map(np.array_split(map_xyz, 3))
    split = np.array_split(np.arange(len(scan)), world)
    sh.set_scan(scan[split[rank]])
    M = ref_cpu.Map()
    for c in np.array_split(map_xyz, 3):
        M.insert(c)
    F = ref_cpu.ICP(M, scan, ref_cpu.make_config(**cfg))
    poses = [(R, t)] + [(R @ synth.so3_exp(w), t + d) for w, d in POSE_STEPS]
    moved = 0
    for k, (Rk, tk) in enumerate(poses):
        got, ref = sh.linearize(Rk, tk), F.linearize(Rk, tk)
        assert_result_parity(got, ref)                       # H, b, f, localizabilities, degeneracy info, histogram, counters
        if k >= 2:
            moved += got["n_migrated_in"]
        # the state of the points this rank holds now == the unsharded factor's state of those points
        origin, st, mean, nrm = sh.state()
        glob = np.array([split[int(o >> np.uint64(32))][int(o & np.uint64(0xFFFFFFFF))] for o in origin], np.int64)
        rs, rm, rn, _ = F.da_state()
        assert np.array_equal(st, rs[glob]) and np.abs(mean - rm[glob]).max() <= 1e-9 and np.abs(nrm - rn[glob]).max() <= 1e-9
    stats = sh.map.stats()
    if world > 1:
        assert 0 < stats["n_points"] < M.num_points           # a real shard (with halo), not the whole map
    else:
        assert stats["n_points"] == M.num_points
    sh.close()
    return moved


def run_random(dist, ctx, device, seed):
    """One random configuration of the sharded path against the unsharded oracle: room, map density, neighbour-independent
    registration options, how the scan is split over the ranks, and a pose walk from millimetres to decimetres (points change
    owner).  Every rank draws the same numbers."""
    from mimosa_amd import capi, dist as mdist, synth
    from oracle import ref_cpu
    from parity import assert_result_parity, assert_state_parity

    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(88000 + seed)
    room = np.array([rng.uniform(8, 30), rng.uniform(6, 20), rng.uniform(2.5, 4)])
    grid = float(rng.choice([0.11, 0.16, 0.3]))
    map_xyz = synth.make_room(6000 + seed, 0, 0, grid=grid, room=room)
    loc = np.array([rng.uniform(1.5, room[0] - 1.5), rng.uniform(1.5, room[1] - 1.5), rng.uniform(0.8, room[2] - 0.8)])
    scan, aux = synth.make_scan(n_rows=int(rng.choice([16, 32])), seed=7000 + seed, n_cols=int(rng.choice([64, 128])), room=room, sensor_local=loc)
    R, t = synth.query_pose(aux["R_W_L"], aux["t_W_L"])
    cfg = dict(synth.enwide_config(), use_huber=int(rng.integers(0, 2)), reg_4_dof=int(rng.integers(0, 2)),
               project_on_degneneracy=int(rng.integers(0, 2)), degen_thresh_trans=float(rng.choice([15.0, 40.0, 1e9])),
               max_corres_distance=float(rng.choice([0.5, 1.0])), plane_validity_distance=float(rng.choice([0.04, 0.07, 0.2])))
    sh = mdist.ShardedICPDevice(dist.group.WORLD, ctx, 0.5, capi.make_reg_config(**cfg), device)
    chunks = np.array_split(map_xyz, int(rng.integers(1, 5)))
    sh.build_map(chunks)
    cuts = np.sort(rng.integers(0, len(scan) + 1, world - 1)) if world > 1 else np.array([], int)   # uneven, possibly empty shares
    split = np.split(np.arange(len(scan)), cuts)
    sh.set_scan(scan[split[rank]])
    M = ref_cpu.Map()
    for c in chunks:
        M.insert(c)
    F = ref_cpu.ICP(M, scan, ref_cpu.make_config(**cfg))
    Rk, tk = R, t
    for k in range(4):
        got, ref = sh.linearize(Rk, tk), F.linearize(Rk, tk)
        assert_result_parity(got, ref, check_eigvec=False)
        origin, st, mean, nrm = sh.state()
        glob = np.array([split[int(o >> np.uint64(32))][int(o & np.uint64(0xFFFFFFFF))] for o in origin], np.int64)
        rs, rm, rn, _ = F.da_state()
        if len(glob):
            assert_state_parity((st, mean, nrm), (rs[glob], rm[glob], rn[glob]))
        scale = float(rng.choice([0.002, 0.03, 0.15]))
        Rk = Rk @ synth.so3_exp(rng.normal(0, 1.0, 3) * scale / 5.0)
        tk = tk + rng.normal(0, 1.0, 3) * scale
    sh.close()
