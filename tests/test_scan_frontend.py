"""Device-resident scan front end ("next" row f-3).

CPU part: the two restatements of Manager::prepareInput (oracle/ref_cpu.hpp, oracle/numpy_ref.py) agree.
GPU part: mh_scan_* reproduces the oracle EXACTLY — same points, same order, same bits — through
prepareInput -> deskewPoints -> Geometric::preprocess -> Geometric::downsample, and the factor built from
the device cloud linearizes to the same Hessian as one built from the oracle's host cloud.
"""
import numpy as np
import pytest

from mimosa_amd import synth
from oracle import numpy_ref, ref_cpu

FIELDS = ("x", "y", "z", "intensity", "t", "idx", "range")


def _as_points(void32):
    return np.frombuffer(np.ascontiguousarray(void32).tobytes(), dtype=synth.POINT_DTYPE)


def _same_points(a, b):
    assert len(a) == len(b)
    for k in FIELDS:
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k


CONFIGS = [
    dict(),
    dict(create_full_res_pointcloud=0),
    dict(point_skip_divisor=3, ring_skip_divisor=2, range_min=2.0, range_max=30.0, intensity_min=100.0,
         intensity_max=1500.0, z_offset=-0.03618),
]


@pytest.mark.parametrize("kw", CONFIGS)
def test_prepare_input_restatements_agree(kw):
    raw, _ = synth.make_raw_scan(32, n_cols=256)
    o = ref_cpu.prepare_input(raw, ref_cpu.make_input_config(**kw))
    d = dict(range_min=0.2, range_max=100.0, intensity_min=0.0, intensity_max=1.0e10, ns_max=1.0e9, z_offset=0.0,
             create_full_res_pointcloud=1, point_skip_divisor=4, ring_skip_divisor=1)
    d.update(kw)
    d["create_full_res_pointcloud"] = bool(d["create_full_res_pointcloud"])
    p = numpy_ref.prepare_input(raw, **d)
    full = _as_points(o["points_full"])
    assert 0 < len(full) < len(raw)
    for k in FIELDS:
        assert np.array_equal(full[k].view(np.uint32), np.asarray(p["points_full"][k]).view(np.uint32)), k
    assert np.array_equal(o["geometric_idxs"], p["geometric_idxs"])
    assert np.array_equal(o["unique_ns"], p["unique_ns"])
    assert o["last_point_ns"] == p["last_point_ns"]
    assert len(o["groups"]) == len(p["groups"])
    for a, b in zip(o["groups"], p["groups"]):
        assert np.array_equal(np.sort(a), np.sort(b))  # membership: the order inside a timestamp is unspecified


def test_prepare_input_edge_cases():
    cfg = ref_cpu.make_input_config()
    raw, _ = synth.make_raw_scan(8, n_cols=64)
    o = ref_cpu.prepare_input(raw[:0], cfg)
    assert len(o["points_full"]) == 0 and len(o["unique_ns"]) == 0 and o["last_point_ns"] == 0
    bad = raw.copy()
    bad["x"] = np.nan
    o = ref_cpu.prepare_input(bad, cfg)
    assert len(o["points_full"]) == 0 and len(o["geometric_idxs"]) == 0


# ---- GPU ------------------------------------------------------------------------------------------------
def _oracle_pipeline(raw, aux, kw, R_B_L, t_B_L, leaf, min_dist):
    o = ref_cpu.prepare_input(raw, ref_cpu.make_input_config(**kw))
    full = _as_points(o["points_full"]).copy()
    # poses for the kept timestamps: make_scan's per-column table, looked up by timestamp
    col_of = {int(t): c for c, t in enumerate(aux["unique_ns"])}
    Rt12 = np.stack([aux["Rt12"][col_of[int(t)]] for t in o["unique_ns"]]) if len(o["unique_ns"]) else np.zeros((0, 12), np.float32)
    desk = ref_cpu.deskew(full, o["unique_ns"], Rt12)
    body = ref_cpu.transform_f32(desk[o["geometric_idxs"].astype(np.int64)], R_B_L, t_B_L)
    kept = ref_cpu.downsample(body, leaf, 20, min_dist)
    return o, Rt12, desk, body, kept


@pytest.mark.gpu
@pytest.mark.parametrize("kw,rows,cols", [(CONFIGS[0], 32, 256), (CONFIGS[1], 16, 512), (CONFIGS[2], 64, 256),
                                          (CONFIGS[0], 128, 1024)])
def test_scan_frontend_matches_oracle(ctx, kw, rows, cols):
    from mimosa_amd import capi
    raw, aux = synth.make_raw_scan(rows, n_cols=cols)
    R_B_L = synth.rot_z(0.3).astype(np.float32)
    t_B_L = np.array([-0.006253, 0.011775, 0.0028525], np.float32)
    leaf, min_dist = 0.5, 0.15
    o, Rt12, desk, body, kept = _oracle_pipeline(raw, aux, kw, R_B_L, t_B_L, leaf, min_dist)

    sc = capi.Scan(ctx)
    info = sc.prepare_input(raw, capi.make_input_config(**kw))
    assert info["n_in"] == len(raw)
    assert info["n_full"] == len(o["points_full"])
    assert info["n_geometric"] == len(o["geometric_idxs"])
    assert info["last_point_ns"] == o["last_point_ns"]
    assert np.array_equal(sc.unique_ns(), o["unique_ns"])
    _same_points(sc.points(capi.Scan.FULL), _as_points(o["points_full"]))
    assert np.array_equal(sc.indices(0), o["geometric_idxs"].astype(np.uint32))

    sc.deskew(Rt12)
    _same_points(sc.points(capi.Scan.FULL), desk)

    info = sc.preprocess_geometric(R_B_L, t_B_L, leaf, 20, min_dist)
    assert info["n_body"] == len(body)
    _same_points(sc.points(capi.Scan.BODY), body)
    assert info["n_downsampled"] == len(kept)
    assert np.array_equal(sc.indices(1), kept)          # same points, same (first-seen voxel) order
    _same_points(sc.points(capi.Scan.DOWNSAMPLED), body[kept])
    sc.destroy()


@pytest.mark.gpu
def test_scan_frontend_empty_and_all_rejected(ctx):
    from mimosa_amd import capi
    raw, _ = synth.make_raw_scan(8, n_cols=64)
    sc = capi.Scan(ctx)
    for cloud in (raw[:0], None):
        if cloud is None:
            cloud = raw.copy()
            cloud["intensity"] = np.nan
        info = sc.prepare_input(cloud, capi.make_input_config())
        assert info["n_full"] == 0 and info["n_geometric"] == 0 and info["n_unique_ns"] == 0
        assert len(sc.unique_ns()) == 0
        sc.deskew(np.zeros((0, 12), np.float32))
        info = sc.preprocess_geometric(np.eye(3, dtype=np.float32), np.zeros(3, np.float32))
        assert info["n_body"] == 0 and info["n_downsampled"] == 0
        assert len(sc.points(capi.Scan.DOWNSAMPLED)) == 0
    sc.destroy()


@pytest.mark.gpu
def test_scan_frontend_voxel_cap_and_duplicates(ctx):
    """More than 20 admissible points in one voxel (cap) and exact duplicates (distance 0 < min_dist)."""
    from mimosa_amd import capi
    rng = np.random.default_rng(5)
    n = 4096
    raw = np.zeros(n, dtype=synth.OUSTER_DTYPE)
    xyz = rng.uniform(2.0, 3.5, size=(n, 3)).astype(np.float32)  # 27 voxels, ~150 points each
    xyz[100:200] = xyz[:100]                                     # duplicates
    raw["x"], raw["y"], raw["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    raw["intensity"] = 10.0
    raw["t"] = (np.arange(n) // 64 * 1000).astype(np.uint32)
    kw = dict(point_skip_divisor=1)
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    o = ref_cpu.prepare_input(raw, ref_cpu.make_input_config(**kw))
    body = _as_points(o["points_full"])[o["geometric_idxs"].astype(np.int64)]
    sc = capi.Scan(ctx)
    sc.prepare_input(raw, capi.make_input_config(**kw))
    for min_dist in (0.01, 0.15):
        kept = ref_cpu.downsample(body, 0.5, 20, min_dist)
        info = sc.preprocess_geometric(I3, z3, 0.5, 20, min_dist)
        assert info["n_downsampled"] == len(kept)
        assert np.array_equal(sc.indices(1), kept)
    sc.destroy()


@pytest.mark.gpu
def test_scan_frontend_long_voxel_segments(ctx):
    """Voxels with 1 / a few / ~300 / ~6000 points: all three sorting paths of the greedy kernel (one wave's registers,
    rank sort over <= 1024 indices, wave-level radix above), interleaved in the input order."""
    from mimosa_amd import capi
    rng = np.random.default_rng(11)
    n = 16384
    xyz = rng.uniform(-20.0, 20.0, size=(n, 3)).astype(np.float32)            # mostly singletons
    big = rng.permutation(n)[:6000]
    xyz[big] = rng.uniform(4.0, 4.5, size=(6000, 3)).astype(np.float32)        # one voxel, 6000 points
    mid = rng.permutation(np.setdiff1d(np.arange(n), big))[:900]
    xyz[mid] = (np.array([-7.0, 3.0, 1.0]) + rng.uniform(0.0, 0.5, size=(900, 3)) + np.array([0.5, 0.0, 0.0]) * rng.integers(0, 3, size=(900, 1))).astype(np.float32)
    raw = np.zeros(n, dtype=synth.OUSTER_DTYPE)
    raw["x"], raw["y"], raw["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    raw["intensity"] = 10.0
    raw["t"] = (np.arange(n) % 512 * 977).astype(np.uint32)
    kw = dict(point_skip_divisor=1, range_min=0.0)
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    o = ref_cpu.prepare_input(raw, ref_cpu.make_input_config(**kw))
    body = _as_points(o["points_full"])[o["geometric_idxs"].astype(np.int64)]
    sc = capi.Scan(ctx)
    sc.prepare_input(raw, capi.make_input_config(**kw))
    for min_dist in (0.02, 0.15):
        kept = ref_cpu.downsample(body, 0.5, 20, min_dist)
        info = sc.preprocess_geometric(I3, z3, 0.5, 20, min_dist)
        assert info["n_downsampled"] == len(kept)
        assert np.array_equal(sc.indices(1), kept)
        _same_points(sc.points(capi.Scan.DOWNSAMPLED), body[kept])
    sc.destroy()


@pytest.mark.gpu
def test_scan_frontend_many_distinct_timestamps_and_device_input(ctx):
    """Every point its own timestamp (the rank sort loops over many tiles), the value 0xFFFFFFFF among them (the hash
    set's empty marker), and the same cloud handed over as a device buffer."""
    import ctypes as C
    from mimosa_amd import capi
    rng = np.random.default_rng(3)
    raw, _ = synth.make_raw_scan(16, n_cols=512)
    n = len(raw)
    raw["t"] = rng.permutation(np.arange(n, dtype=np.uint64) * 4099 % (1 << 32)).astype(np.uint32)
    raw["t"][5] = 0xFFFFFFFF
    raw["t"][77] = raw["t"][78]                      # one duplicate pair
    kw = dict(ns_max=1.0e10)
    o = ref_cpu.prepare_input(raw, ref_cpu.make_input_config(**kw))
    assert o["unique_ns"][-1] == 0xFFFFFFFF and len(o["unique_ns"]) > 5000
    sc = capi.Scan(ctx)
    info = sc.prepare_input(raw, capi.make_input_config(**kw))
    assert info["n_unique_ns"] == len(o["unique_ns"]) and info["last_point_ns"] == o["last_point_ns"] == 0xFFFFFFFF
    assert np.array_equal(sc.unique_ns(), o["unique_ns"])
    _same_points(sc.points(capi.Scan.FULL), _as_points(o["points_full"]))
    hip = C.CDLL("libamdhip64.so")
    d_raw = C.c_void_p()
    assert hip.hipMalloc(C.byref(d_raw), C.c_size_t(raw.nbytes)) == 0
    assert hip.hipMemcpy(d_raw, C.c_void_p(raw.ctypes.data), C.c_size_t(raw.nbytes), 1) == 0
    sc2 = capi.Scan(ctx)
    info2 = sc2.prepare_input_device(d_raw.value, n, capi.make_input_config(**kw))
    assert info2 == info
    assert np.array_equal(sc2.unique_ns(), o["unique_ns"])
    _same_points(sc2.points(capi.Scan.FULL), _as_points(o["points_full"]))
    assert np.array_equal(sc2.indices(0), o["geometric_idxs"].astype(np.uint32))
    sc.destroy()
    sc2.destroy()
    hip.hipFree(d_raw)


@pytest.mark.gpu
def test_factor_from_device_scan_equals_factor_from_host_cloud(ctx, small_world):
    from mimosa_amd import capi
    w = small_world
    raw, aux = synth.make_raw_scan(32, seed=1235, n_cols=256, room=(6.0, 5.0, 3.0), sensor_local=np.array([2.3, 2.6, 1.2]))
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    kw = dict(point_skip_divisor=1, range_min=0.05)
    o, Rt12, desk, body, kept = _oracle_pipeline(raw, aux, kw, I3, z3, 0.5, 0.15)
    m = capi.VoxelMap(ctx)
    m.insert(w["map_xyz"])
    sc = capi.Scan(ctx)
    sc.prepare_input(raw, capi.make_input_config(**kw))
    sc.deskew(Rt12)
    sc.preprocess_geometric(I3, z3, 0.5, 20, 0.15)
    rc = capi.make_reg_config(**w["cfg"])
    f_dev = sc.make_factor(m, rc)
    f_host = capi.ICPFactor(ctx, m, body[kept], rc)
    assert f_dev.n == f_host.n == len(kept) > 100
    a, b = f_dev.linearize(w["R"], w["t"]), f_host.linearize(w["R"], w["t"])
    for k in ("H_ss", "b_s", "f"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
    assert a["status_hist"][8] > 50  # plenty of valid correspondences: the comparison is not vacuous
    sa, sb = f_dev.state(), f_host.state()
    assert np.array_equal(sa[0], sb[0])
    sc.destroy()


@pytest.mark.gpu
def test_scan_api_misuse_is_reported_not_crashed(ctx):
    """Out-of-order calls and bad arguments come back as error codes with a message."""
    from mimosa_amd import capi
    sc = capi.Scan(ctx)
    I3, z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    with pytest.raises(capi.MhError, match="prepare_input"):
        sc.deskew(np.zeros((1, 12), np.float32))
    with pytest.raises(capi.MhError, match="prepare_input"):
        sc.preprocess_geometric(I3, z3)
    raw, aux = synth.make_raw_scan(8, n_cols=64)
    with pytest.raises(capi.MhError, match="divisors"):
        sc.prepare_input(raw, capi.make_input_config(point_skip_divisor=0))
    sc.prepare_input(raw, capi.make_input_config())
    with pytest.raises(capi.MhError, match="one pose per unique timestamp"):
        sc.deskew(np.zeros((3, 12), np.float32))
    with pytest.raises(capi.MhError, match="1..20"):
        sc.preprocess_geometric(I3, z3, 0.5, 21, 0.1)
    with pytest.raises(capi.MhError, match="leaf_size"):
        sc.preprocess_geometric(I3, z3, 0.0, 20, 0.1)
    with pytest.raises(capi.MhError, match="has not run"):
        sc.points(capi.Scan.DOWNSAMPLED)
    m = capi.VoxelMap(ctx)
    with pytest.raises(capi.MhError, match="preprocess_geometric"):
        sc.make_factor(m, capi.make_reg_config(**synth.enwide_config()))
    sc.destroy()
