"""-m gpu: the device-resident incremental voxel map (row f-1 / a13): iVox insert, LRU purge, getCloud, copy and the
block / hash bookkeeping run as kernels; contents, voxel order and k-NN answers must equal the oracle's bit for bit."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _same(gm, rm, q=None):
    s = gm.stats()
    assert s["n_points"] == rm.num_points and s["n_voxels"] == rm.num_voxels, (s, rm.num_points, rm.num_voxels)
    assert np.array_equal(gm.get_cloud(), rm.export()[2])
    if q is not None and len(q):
        pts, sq, found = gm.knn(q, 5)
        idx, sq_r, found_r, _ = rm.knn(q, 5)
        assert np.array_equal(found, found_r)
        for i in range(len(q)):
            assert np.array_equal(sq[i, : found[i]], sq_r[i, : found[i]])


def test_incremental_inserts_match_oracle(ctx, room_world):
    """0.5 M points in ragged batches (1 point, 7, 5 000, 200 000, the rest): new voxels in old blocks, new blocks, several
    hash-table growths, voxels filling up to the cap across batches."""
    from mimosa_amd import capi
    from oracle import ref_cpu

    m = room_world["map_xyz"]
    gm, rm = capi.VoxelMap(ctx), ref_cpu.Map()
    rng = np.random.default_rng(3)
    q = m[rng.integers(0, len(m), 300)].astype(np.float64) + rng.normal(0, 0.15, (300, 3))
    cuts = [0, 1, 8, 5008, 205008, len(m)]
    for a, b in zip(cuts[:-1], cuts[1:]):
        gm.insert(m[a:b])
        rm.insert(m[a:b])
        _same(gm, rm, q[:60])
    _same(gm, rm, q)
    gm.insert(m[:0])                                            # an empty batch still counts as an insert (LRU cadence)
    rm.insert(m[:0])
    dense = (m[:3000] + rng.normal(0, 0.01, (3000, 3)).astype(np.float32))   # near-duplicates: min-distance rule + cap
    gm.insert(dense)
    rm.insert(dense)
    _same(gm, rm, q)
    assert gm.stats()["n_blocks"] > 512
    gm.release()


def test_lru_purge_on_the_device(ctx):
    """Voxels untouched for lru_horizon inserts disappear at every lru_clear_cycle-th insert; the survivors keep their
    order (getCloud), ids are renumbered, tables rebuilt — all on the device."""
    from mimosa_amd import capi, synth
    from oracle import ref_cpu

    room = np.array([10.0, 8.0, 3.0])
    base = synth.make_room(99, 0, 0, room=room)
    gm = capi.VoxelMap(ctx, lru_horizon=3, lru_clear_cycle=2)
    rm = ref_cpu.Map(lru_horizon=3)
    rm.set_lru_clear_cycle(2)
    rng = np.random.default_rng(8)
    purged = False
    for k in range(14):
        lo = k * 0.5                                           # a window sliding along x: old voxels age out
        sel = (base[:, 0] > lo) & (base[:, 0] < lo + 3.0)
        batch = base[sel] + rng.normal(0, 0.002, (sel.sum(), 3)).astype(np.float32)
        nv0 = gm.stats()["n_voxels"]
        gm.insert(batch)
        rm.insert(batch)
        purged |= gm.stats()["n_voxels"] < nv0
        q = batch[rng.integers(0, len(batch), 80)].astype(np.float64) + 0.02
        _same(gm, rm, q)
    assert purged
    g2 = gm.copy()                                              # the copy carries the LRU stamps and the counter
    r2 = rm.copy()
    for k in range(3):
        b = base[(base[:, 0] > 7.0 + k)][:2000]
        g2.insert(b)
        r2.insert(b)
        _same(g2, r2)
    gm.release()
    g2.release()


def test_bad_batches_are_rejected_without_touching_the_map(ctx, small_world):
    from mimosa_amd import capi
    from oracle import ref_cpu

    m = small_world["map_xyz"]
    gm, rm = capi.VoxelMap(ctx), ref_cpu.Map()
    gm.insert(m)
    rm.insert(m)
    bad = m[:100].copy()
    bad[37, 1] = np.nan
    with pytest.raises(capi.MhError):
        gm.insert(bad)
    far = m[:100].copy()
    far[5, 0] = 1.0e7                                           # 2e7 voxels from the origin: outside the 21-bit key
    with pytest.raises(capi.MhError):
        gm.insert(far)
    _same(gm, rm)
    gm.insert(m[:500] + np.float32(0.05))                       # and it still works afterwards
    rm.insert(m[:500] + np.float32(0.05))
    _same(gm, rm)
    # a 2^40-point batch must come back as an error code, not as an abort (ABI: nothing throws across the boundary)
    one = np.zeros(3, np.float32)
    rc = ctx.L.mh_map_insert(gm.h, one.ctypes.data_as(C.c_void_p), 1 << 40, 3)
    assert rc in (capi.MH_ERR_OOM, capi.MH_ERR_UNSUPPORTED, capi.MH_ERR_INVALID_ARG)
    _same(gm, rm)
    gm.release()


def test_insert_from_resident_scan(ctx, small_world):
    """mh_map_insert_from_scan: Be_cloud_ of a device-resident scan, f32 world transform on the device, no host round trip ==
    the host path (mh_transform_f32 on the downloaded cloud, then mh_map_insert) == the oracle."""
    from mimosa_amd import capi, synth
    from oracle import ref_cpu

    raw, aux = synth.make_raw_scan(16, n_cols=64, room=(6.0, 5.0, 3.0), sensor_local=np.array([2.3, 2.6, 1.2]))
    sc = capi.Scan(ctx)
    sc.prepare_input(raw, capi.make_input_config())
    uns = sc.unique_ns()
    sc.deskew(aux["Rt12"][np.searchsorted(aux["unique_ns"], uns)])
    sc.preprocess_geometric(np.eye(3, dtype=np.float32), np.zeros(3, np.float32))
    body = sc.points(capi.Scan.BODY)
    R, t = aux["R_W_L"], aux["t_W_L"]
    ga, gb, rm = capi.VoxelMap(ctx), capi.VoxelMap(ctx), ref_cpu.Map()
    for g in (ga, gb):
        g.insert(small_world["map_xyz"])
    rm.insert(small_world["map_xyz"])
    ga.insert_from_scan(sc, R, t)
    W = ctx.transform_f32(body, R.astype(np.float32), t.astype(np.float32))
    Wx = np.stack([W["x"], W["y"], W["z"]], 1)
    gb.insert(Wx)
    Wr = ref_cpu.transform_f32(body, R.astype(np.float32), t.astype(np.float32))
    rm.insert(np.stack([Wr["x"], Wr["y"], Wr["z"]], 1))
    assert np.array_equal(ga.get_cloud(), gb.get_cloud())
    _same(ga, rm)
    for g in (ga, gb):
        g.release()
    sc.destroy()


@pytest.mark.parametrize("world", [1, 2, 8])
def test_insert_shard_from_resident_scan(ctx, small_world, world):
    """mh_map_insert_shard_from_scan — one rank's share of Geometric::updateMap's insert, transform + shard filter + insert on the
    device — == the host route (download Be_cloud_, mh_transform_f32, mh_map_insert_shard), rank by rank; and the ranks' shards
    together hold the unsharded map's points."""
    from mimosa_amd import capi, synth

    raw, aux = synth.make_raw_scan(16, n_cols=64, room=(6.0, 5.0, 3.0), sensor_local=np.array([2.3, 2.6, 1.2]))
    sc = capi.Scan(ctx)
    sc.prepare_input(raw, capi.make_input_config())
    sc.deskew(aux["Rt12"][np.searchsorted(aux["unique_ns"], sc.unique_ns())])
    sc.preprocess_geometric(np.eye(3, dtype=np.float32), np.zeros(3, np.float32))
    body = sc.points(capi.Scan.BODY)
    R, t = aux["R_W_L"], aux["t_W_L"]
    W = ctx.transform_f32(body, R.astype(np.float32), t.astype(np.float32))
    Wx = np.stack([W["x"], W["y"], W["z"]], 1)
    whole = capi.VoxelMap(ctx)
    whole.insert(small_world["map_xyz"])
    whole.insert(Wx)
    want = {tuple(p) for p in whole.get_cloud()}
    seen = set()
    for rank in range(world):
        ga, gb = capi.VoxelMap(ctx), capi.VoxelMap(ctx)
        for g in (ga, gb):
            capi.map_insert_shard(ctx, g, small_world["map_xyz"], world, rank, 2)
        capi.map_insert_shard_from_scan(ctx, ga, sc, R, t, world, rank, 2)
        capi.map_insert_shard(ctx, gb, Wx, world, rank, 2)
        ca, cb = ga.get_cloud(), gb.get_cloud()
        assert np.array_equal(ca, cb), rank
        seen |= {tuple(p) for p in ca}
        ga.release()
        gb.release()
    assert seen == want          # owned blocks + halos cover the map; nothing a rank holds is foreign to it
    whole.release()
    sc.destroy()
