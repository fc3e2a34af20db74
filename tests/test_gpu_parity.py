"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs."""
import numpy as np
import pytest

from parity import assert_result_parity, assert_state_parity, rel

pytestmark = pytest.mark.gpu


def _mk(ctx, world, binary=False, mode=19, cfg_over=None, k=None):
    from mimosa_amd import capi
    from oracle import ref_cpu

    cfg = dict(world["cfg"])
    if cfg_over:
        cfg.update(cfg_over)
    if k is not None:
        cfg["num_corres_points"] = k
    gm = capi.VoxelMap(ctx, mode=mode)
    gm.insert(world["map_xyz"])
    rm = ref_cpu.Map(mode=mode)
    rm.insert(world["map_xyz"])
    gf = capi.ICPFactor(ctx, gm, world["pts"], capi.make_reg_config(**cfg), binary=binary)
    rf = ref_cpu.ICP(rm, world["pts"], ref_cpu.make_config(**cfg), binary=binary)
    return gm, rm, gf, rf


def test_map_matches_oracle(ctx, small_world):
    from mimosa_amd import capi
    from oracle import ref_cpu

    gm = capi.VoxelMap(ctx)
    rm = ref_cpu.Map()
    for chunk in np.array_split(small_world["map_xyz"], 3):
        gm.insert(chunk)
        rm.insert(chunk)
    s = gm.stats()
    assert s["n_voxels"] == rm.num_voxels and s["n_points"] == rm.num_points
    _, _, xyz = rm.export()
    assert np.array_equal(gm.get_cloud(), xyz)  # same voxel creation order, same kept points (bit-exact)


@pytest.mark.parametrize("k", [1, 5, 8])
def test_knn_matches_oracle(ctx, small_world, k):
    gm, rm, gf, rf = _mk(ctx, small_world)
    rng = np.random.default_rng(5)
    q = small_world["map_xyz"][rng.integers(0, len(small_world["map_xyz"]), 700)].astype(np.float64)
    q += rng.normal(0, 0.2, q.shape)
    q = np.concatenate([q, [[1e3, 1e3, 1e3], [-50.2, 3.3, 9.1]]])  # far away: nothing found
    pts, sq, found = gm.knn(q, k)
    idx, sq_r, found_r, _ = rm.knn(q, k)
    assert np.array_equal(found, found_r)
    for i in range(len(q)):
        f = found[i]
        assert np.array_equal(sq[i, :f], sq_r[i, :f]), (  # fp64 distances bit-exact, same order
            i, q[i].tolist(), (sq[i, :f] - sq_r[i, :f]).tolist(), sq[i, :f].view(np.int64) - sq_r[i, :f].view(np.int64))
        for j in range(f):
            assert np.array_equal(pts[i, j], rm.point(idx[i, j]))


def test_linearize_cold_and_relinearize(ctx, small_world):
    from mimosa_amd import synth
    gm, rm, gf, rf = _mk(ctx, small_world)
    R, t = small_world["R"], small_world["t"]
    g = gf.linearize(R, t)
    r = rf.linearize(R, t)
    assert_result_parity(g, r)
    assert_state_parity(gf.state(), rf.state())
    assert g["status_hist"][8] > 300  # a healthy share of valid correspondences
    # small move: everything hits the data-association cache (no k-NN)
    t2 = t + np.array([0.004, 0.003, -0.002])
    g2, r2 = gf.linearize(R, t2), rf.linearize(R, t2)
    assert g2["n_knn"] == 0
    assert_result_parity(g2, r2)
    # small rotation: far points move more than min_dist/4 and re-associate, near ones do not
    R3 = R @ synth.so3_exp(np.array([0.0, 0.0, 0.015]))
    g3, r3 = gf.linearize(R3, t2), rf.linearize(R3, t2)
    assert 0 < g3["n_knn"] < len(small_world["pts"])
    assert_result_parity(g3, r3)
    assert_state_parity(gf.state(), rf.state())


def test_reset_equals_fresh_factor(ctx, small_world):
    gm, rm, gf, rf = _mk(ctx, small_world)
    R, t = small_world["R"], small_world["t"]
    first = gf.linearize(R, t)
    gf.linearize(R, t + 0.05)
    gf.reset()
    again = gf.linearize(R, t)
    for k in ("H_ss", "b_s", "f", "status_hist", "loc_trans_comp"):
        assert np.array_equal(first[k], again[k]), k  # deterministic reduction order: bit-identical
    from oracle import ref_cpu
    rf2 = ref_cpu.ICP(rm, small_world["pts"], ref_cpu.make_config(**small_world["cfg"]))
    rf2.linearize(R, t)
    assert_state_parity(gf.state(), rf2.state())


def test_clone_is_deep(ctx, small_world):
    gm, rm, gf, rf = _mk(ctx, small_world)
    R, t = small_world["R"], small_world["t"]
    gf.linearize(R, t)
    rf.linearize(R, t)
    gc, rc = gf.clone(), rf.clone()
    t2 = t + np.array([0.03, -0.02, 0.01])
    assert_result_parity(gc.linearize(R, t2), rc.linearize(R, t2))
    # the original still holds its own association state
    t3 = t + np.array([0.001, 0.0, 0.0])
    assert_result_parity(gf.linearize(R, t3), rf.linearize(R, t3))


def test_binary_factor(ctx, small_world):
    gm, rm, gf, rf = _mk(ctx, small_world, binary=True)
    R, t = small_world["R"], small_world["t"]
    from mimosa_amd import synth
    Rt = synth.so3_exp(np.array([0.01, -0.02, 0.015]))
    tt = np.array([0.02, -0.01, 0.03])
    Rs, ts = Rt @ R, Rt @ t + tt  # so that delta = T_tgt^-1 T_src equals the unary query pose
    g = gf.linearize(Rs, ts, R_tgt=Rt, t_tgt=tt)
    r = rf.linearize(Rs, ts, R_tgt=Rt, t_tgt=tt)
    assert_result_parity(g, r, binary=True)


@pytest.mark.parametrize("mode", [1, 7, 27])
def test_neighbor_modes(ctx, small_world, mode):
    gm, rm, gf, rf = _mk(ctx, small_world, mode=mode)
    assert_result_parity(gf.linearize(small_world["R"], small_world["t"]), rf.linearize(small_world["R"], small_world["t"]))


@pytest.mark.parametrize("k", [3, 4, 6, 8])
def test_num_corres_points(ctx, small_world, k):
    gm, rm, gf, rf = _mk(ctx, small_world, k=k)
    assert_result_parity(gf.linearize(small_world["R"], small_world["t"]), rf.linearize(small_world["R"], small_world["t"]))


def test_reg_4_dof_and_degeneracy_projection(ctx, small_world):
    R, t = small_world["R"], small_world["t"]
    g_unit = np.array([0.05, -0.02, -1.0])
    g_unit /= np.linalg.norm(g_unit)
    gm, rm, gf, rf = _mk(ctx, small_world, cfg_over=dict(reg_4_dof=1))
    assert_result_parity(gf.linearize(R, t, g_unit), rf.linearize(R, t, g_unit))
    # struct-default thresholds trigger the reference's degeneracy branch (H, b rebuilt as zero: F10)
    gm, rm, gf, rf = _mk(ctx, small_world, cfg_over=dict(project_on_degneneracy=1, degen_thresh_trans=1e9))
    g, r = gf.linearize(R, t), rf.linearize(R, t)
    assert not g["H_ss"].any() and not r["H_ss"].any()
    assert_result_parity(g, r, check_eigvec=False)


def test_empty_and_far_inputs(ctx, small_world):
    from mimosa_amd import capi, synth
    from oracle import ref_cpu
    cfg = small_world["cfg"]
    gm = capi.VoxelMap(ctx)
    # empty map: every point InsufficientCorresPoints
    gf = capi.ICPFactor(ctx, gm, small_world["pts"][:100], capi.make_reg_config(**cfg))
    g = gf.linearize(small_world["R"], small_world["t"])
    assert g["status_hist"][1] == 100 and g["f"] == 0.0
    # empty cloud
    gm.insert(small_world["map_xyz"])
    gf0 = capi.ICPFactor(ctx, gm, np.zeros(0, synth.POINT_DTYPE), capi.make_reg_config(**cfg))
    g0 = gf0.linearize(small_world["R"], small_world["t"])
    assert g0["status_hist"].sum() == 0 and not g0["H_ss"].any()
    # cloud transformed far outside the map
    gf2 = capi.ICPFactor(ctx, gm, small_world["pts"], capi.make_reg_config(**cfg))
    rm = ref_cpu.Map(); rm.insert(small_world["map_xyz"])
    rf2 = ref_cpu.ICP(rm, small_world["pts"], ref_cpu.make_config(**cfg))
    tf = small_world["t"] + 500.0
    assert np.array_equal(gf2.linearize(small_world["R"], tf)["status_hist"], rf2.linearize(small_world["R"], tf)["status_hist"])


def test_room_scale_parity(ctx, room_world):
    """config[0] size: 65 536-pt scan vs ~0.5 M-pt map."""
    gm, rm, gf, rf = _mk(ctx, room_world)
    R, t = room_world["R"], room_world["t"]
    r = rf.linearize(R, t)
    g = gf.linearize(R, t)
    assert_result_parity(g, r)
    assert_state_parity(gf.state(), rf.state())
    # per-point whitened residual / Jacobian rows on points valid in both (§8(d) parity check)
    e_r, J_r, valid = rf.point_rows(R, t)
    st, mean, nrm = gf.state()
    P = np.stack([room_world["pts"]["x"], room_world["pts"]["y"], room_world["pts"]["z"]], 1).astype(np.float64)
    v = (st == 8) & (valid == 1)
    q = P[v] @ R.T + t
    e = np.einsum("ij,ij->i", nrm[v], mean[v] - q)
    sigma = float(np.float32(room_world["cfg"]["lidar_point_noise_std_dev"]))
    hub = float(np.float32(room_world["cfg"]["huber_threshold"]))
    w = np.abs(e / sigma)
    sw = np.where(w > hub, np.sqrt(hub / np.maximum(w, 1e-300)), 1.0)
    assert rel(e * sw / sigma, e_r[v]) <= 1e-5


def test_deskew_bit_exact(ctx):
    from mimosa_amd import synth
    from oracle import ref_cpu
    pts, aux = synth.make_scan(32, skew=True)
    R_B_L = synth.so3_exp(np.array([0.01, 0.02, -0.03])).astype(np.float32)
    t_B_L = np.array([0.1, -0.05, 0.2], np.float32)
    got = ctx.deskew(pts, aux["unique_ns"], aux["Rt12"])
    ref = ref_cpu.deskew(pts, aux["unique_ns"], aux["Rt12"])
    assert got.tobytes() == ref.tobytes()
    got2 = ctx.deskew(pts, aux["unique_ns"], aux["Rt12"], R_B_L, t_B_L)
    ref2 = ref_cpu.transform_f32(ref, R_B_L, t_B_L)
    assert got2.tobytes() == ref2.tobytes()
    got3 = ctx.transform_f32(pts, R_B_L, t_B_L)
    assert got3.tobytes() == ref_cpu.transform_f32(pts, R_B_L, t_B_L).tobytes()
    # timestamps missing from the table are left untouched
    sub = aux["unique_ns"][::2]
    got4 = ctx.deskew(pts, sub, aux["Rt12"][::2])
    assert got4.tobytes() == ref_cpu.deskew(pts, sub, aux["Rt12"][::2]).tobytes()


def test_async_pipeline_matches_sync(ctx, small_world):
    gm, rm, gf, rf = _mk(ctx, small_world)
    R, t = small_world["R"], small_world["t"]
    sync = []
    for i in range(4):
        gf.reset()
        sync.append(gf.linearize(R, t + 0.001 * i))
    outs = []
    for i in range(4):
        gf.reset()
        outs.append(gf.linearize_async(R, t + 0.001 * i))
    gf.wait()
    for a, b in zip(sync, outs):
        b = b.as_dict()
        assert np.array_equal(a["H_ss"], b["H_ss"]) and np.array_equal(a["status_hist"], b["status_hist"])


@pytest.mark.parametrize("n", [1, 63, 511, 513, 1000])
def test_ragged_cloud_sizes(ctx, small_world, n):
    """Cloud sizes around the 512-thread workgroup / 64-lane wave boundaries."""
    from mimosa_amd import capi
    from oracle import ref_cpu
    w = dict(small_world, pts=small_world["pts"][:n])
    gm, rm, gf, rf = _mk(ctx, w)
    g, r = gf.linearize(w["R"], w["t"]), rf.linearize(w["R"], w["t"])
    assert np.array_equal(g["status_hist"], r["status_hist"]) and g["status_hist"].sum() == n
    if r["status_hist"][8] >= 6:
        assert_result_parity(g, r)
    assert_state_parity(gf.state(), rf.state())


@pytest.mark.parametrize("shift,leaf", [((1000.0, -512.0, 37.0), 0.5), ((-3.3, -7.7, -2.2), 0.5), ((0.0, 0.0, 0.0), 0.3),
                                        ((250.0, 250.0, -20.0), 0.37)])
def test_far_origin_negative_coords_and_odd_leaf(ctx, small_world, shift, leaf):
    """The coarse tier works in voxel-relative grid units, so selection must stay exact 1 km from the
    origin, across negative coordinates (floor, arithmetic block shifts, hash of negatives) and for leaf
    sizes whose inverse is not exactly representable."""
    from mimosa_amd import capi
    from oracle import ref_cpu
    sh = np.array(shift)
    m = (small_world["map_xyz"].astype(np.float64) + sh).astype(np.float32)
    cfg = dict(small_world["cfg"], target_ivox_map_leaf_size=leaf)
    gm = capi.VoxelMap(ctx, leaf=leaf)
    rm = ref_cpu.Map(leaf=leaf)
    gm.insert(m)
    rm.insert(m)
    assert gm.stats()["n_points"] == rm.num_points and gm.stats()["n_voxels"] == rm.num_voxels
    gf = capi.ICPFactor(ctx, gm, small_world["pts"], capi.make_reg_config(**cfg))
    rf = ref_cpu.ICP(rm, small_world["pts"], ref_cpu.make_config(**cfg))
    R, t = small_world["R"], small_world["t"] + sh
    g, r = gf.linearize(R, t), rf.linearize(R, t)
    assert_result_parity(g, r)
    assert_state_parity(gf.state(), rf.state(), tol=1e-9 * max(1.0, np.abs(sh).max()))
    # batched k-NN: bit-identical distances and neighbours
    rng = np.random.default_rng(9)
    q = m[rng.integers(0, len(m), 300)].astype(np.float64) + rng.normal(0, 0.1, (300, 3))
    pts, sq, found = gm.knn(q, 5)
    idx, sq_r, found_r, _ = rm.knn(q, 5)
    assert np.array_equal(found, found_r)
    for i in range(len(q)):
        assert np.array_equal(sq[i, : found[i]], sq_r[i, : found[i]])


def test_saturated_voxels_and_duplicates(ctx):
    """Voxels at the 20-point cap, exact duplicate points (rejected by the min-distance rule) and exact
    distance ties (two map points symmetric about a query): the earlier-traversed one must win."""
    from mimosa_amd import capi, synth
    from oracle import ref_cpu
    rng = np.random.default_rng(4)
    dense = rng.uniform(0, 0.5, (400, 3)).astype(np.float32)           # one voxel, far more than 20 candidates
    dup = np.repeat(np.array([[1.25, 0.25, 0.25]], np.float32), 5, 0)  # duplicates
    sym = np.array([[2.0, 0.25, 0.25], [2.5, 0.25, 0.25], [2.25, 0.0, 0.25], [2.25, 0.5 - 2**-20, 0.25],
                    [2.25, 0.25, 0.1], [2.25, 0.25, 0.4]], np.float32)  # +-0.25 pairs around (2.25, 0.25, 0.25)
    m = np.concatenate([dense, dup, sym])
    gm = capi.VoxelMap(ctx, min_dist=0.02)
    rm = ref_cpu.Map(min_dist=0.02)
    gm.insert(m)
    rm.insert(m)
    s = gm.stats()
    assert s["n_points"] == rm.num_points and s["n_voxels"] == rm.num_voxels
    _, counts, xyz = rm.export()
    assert counts.max() == 20 and np.array_equal(gm.get_cloud(), xyz)
    q = np.array([[2.25, 0.25, 0.25], [0.25, 0.25, 0.25], [1.25, 0.25, 0.25], [2.25, 0.25, 0.2500001]])
    for k in (1, 2, 5, 8):
        pts, sq, found = gm.knn(q, k)
        idx, sq_r, found_r, _ = rm.knn(q, k)
        assert np.array_equal(found, found_r)
        for i in range(len(q)):
            assert np.array_equal(sq[i, : found[i]], sq_r[i, : found[i]])
            for j in range(found[i]):
                assert np.array_equal(pts[i, j], rm.point(idx[i, j])), (k, i, j)


def test_keyframe_update_copy_then_insert_on_the_device(ctx, small_world):
    """Geometric::updateMap's copy-then-insert (geometric.cpp:494-495) with the map maintained on the device: the copy
    is device-to-device, an insert moves its own batch and nothing else, the old map (held by live factors) is
    untouched, contents stay identical to the oracle's (new voxels in old blocks, new blocks, hash growth)."""
    from mimosa_amd import capi, synth
    from oracle import ref_cpu
    m = small_world["map_xyz"]
    ga, ra = capi.VoxelMap(ctx), ref_cpu.Map()
    ga.insert(m)
    ra.insert(m)
    s0 = ga.stats()
    assert s0["uploads"] == 1 and s0["upload_bytes"] == len(m) * 12 and s0["full_uploads"] == 0
    rng = np.random.default_rng(12)
    q = m[rng.integers(0, len(m), 400)].astype(np.float64) + rng.normal(0, 0.1, (400, 3))
    # keyframe cloud: re-observes part of the room (touches existing voxels) and adds a new region
    kf = np.concatenate([m[::7] + np.float32(0.07), (m[:600] + np.array([7.5, 0.0, 0.0], np.float32))])
    gb, rb = ga.copy(), ra.copy()
    before = gb.stats()
    assert before["n_points"] == s0["n_points"] and before["upload_bytes"] == 0      # the copy moved nothing over PCIe
    gb.insert(kf)
    rb.insert(kf)
    st = gb.stats()
    assert st["upload_bytes"] - before["upload_bytes"] == len(kf) * 12
    assert st["n_points"] == rb.num_points and st["n_voxels"] == rb.num_voxels and st["n_voxels"] > s0["n_voxels"]
    q2 = np.concatenate([q, kf[-200:].astype(np.float64) + 0.03])
    for gm_, rm_ in ((gb, rb), (ga, ra)):  # the new map AND the untouched old one
        pts, sq, found = gm_.knn(q2, 5)
        idx, sq_r, found_r, _ = rm_.knn(q2, 5)
        assert np.array_equal(found, found_r)
        for i in range(len(q2)):
            assert np.array_equal(sq[i, : found[i]], sq_r[i, : found[i]])
    assert np.array_equal(gb.get_cloud(), rb.export()[2]) and np.array_equal(ga.get_cloud(), ra.export()[2])
    kf2 = m[:50] + np.float32(0.11)
    gb.insert(kf2)
    rb.insert(kf2)
    pts, sq, found = gb.knn(q2, 5)
    idx, sq_r, found_r, _ = rb.knn(q2, 5)
    assert np.array_equal(found, found_r) and all(np.array_equal(sq[i, : found[i]], sq_r[i, : found[i]]) for i in range(len(q2)))
    assert np.array_equal(gb.get_cloud(), rb.export()[2])
    # linearize on the updated map matches the oracle
    cfg = small_world["cfg"]
    gf = capi.ICPFactor(ctx, gb, small_world["pts"], capi.make_reg_config(**cfg))
    rf = ref_cpu.ICP(rb, small_world["pts"], ref_cpu.make_config(**cfg))
    assert_result_parity(gf.linearize(small_world["R"], small_world["t"]), rf.linearize(small_world["R"], small_world["t"]))


def test_copy_keeps_both_maps_writable(ctx, small_world):
    """mh_map_copy (Geometric::updateMap's copy-then-insert, geometric.cpp:494): a factor built on the source keeps its answers
    while the copy grows, and the source accepts inserts of its own afterwards."""
    from mimosa_amd import capi
    from oracle import ref_cpu

    w = small_world
    half = len(w["map_xyz"]) // 2
    gm, rm = capi.VoxelMap(ctx), ref_cpu.Map()
    gm.insert(w["map_xyz"][:half])
    rm.insert(w["map_xyz"][:half])
    cfg = capi.make_reg_config(**w["cfg"])
    f_old = capi.ICPFactor(ctx, gm, w["pts"], cfg)
    r_old = f_old.linearize(w["R"], w["t"])
    cloud_before = gm.get_cloud()
    g2, rm2 = gm.copy(), rm.copy()
    g2.insert(w["map_xyz"][half:])
    rm2.insert(w["map_xyz"][half:])
    assert np.array_equal(g2.get_cloud(), rm2.export()[2]) and g2.stats()["n_points"] == rm2.num_points
    assert np.array_equal(gm.get_cloud(), cloud_before)              # the source did not move
    f_old.reset()
    r_again = f_old.linearize(w["R"], w["t"])
    assert np.array_equal(r_again["H_ss"], r_old["H_ss"]) and r_again["f"] == r_old["f"]
    f_new = capi.ICPFactor(ctx, g2, w["pts"], cfg)
    rf = ref_cpu.ICP(rm2, w["pts"], ref_cpu.make_config(**w["cfg"]))
    assert_result_parity(f_new.linearize(w["R"], w["t"]), rf.linearize(w["R"], w["t"]))
    extra = w["map_xyz"][half:half + 500] + np.float32(0.03)         # the source is still a live map
    gm.insert(extra)
    rm.insert(extra)
    assert np.array_equal(gm.get_cloud(), rm.export()[2])

