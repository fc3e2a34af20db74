"""Manager::prepareInput<PointT> for every point type of the reference (include/mimosa/lidar/point.hpp:40-131):
the typed oracle restatement against the original PointOuster one (CPU), and the device path driven by a layout
descriptor against the typed oracle, bit for bit (GPU)."""
import numpy as np
import pytest

from mimosa_amd import synth
from oracle import ref_cpu

FIELDS = ("x", "y", "z", "intensity", "t", "idx", "range")
HEADER_TS = 1_727_000_000.25   # seconds, a unix time


def _as_points(void32):
    return np.frombuffer(np.ascontiguousarray(void32).tobytes(), dtype=synth.POINT_DTYPE)


def _same_points(a, b):
    assert len(a) == len(b)
    for k in FIELDS:
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k


def make_sensor_scan(kind, rows=32, cols=128, seed=5, order="row"):
    """A synthetic Ouster scan re-expressed as the given sensor's point record, with each type's own defects: Livox tags
    that fail the tag test, timestamps slightly before the header (they wrap past ns_max in the reference's
    double -> uint32 conversion), reflectivity as the intensity.  order="col": points arrive column by column (the
    unorganised Hesai JT128 stream organize_by_ring is for)."""
    from mimosa_amd import capi
    raw, _ = synth.make_raw_scan(rows, seed=synth.BASE_SEED + seed, n_cols=cols)
    n = len(raw)
    rng = np.random.default_rng(seed)
    if order == "col":
        raw = raw.reshape(rows, cols).T.reshape(-1).copy()
    out = np.zeros(n, capi.point_dtype(kind))
    for k in ("x", "y", "z"):
        out[k] = raw[k]
    names = out.dtype.names
    if "intensity" in names:
        out["intensity"] = raw["intensity"]
    if kind == "ouster_odyssey":
        out["reflectivity"] = np.nan_to_num(raw["intensity"], nan=0.0).astype(np.uint16)
        out["near_ir"] = rng.integers(0, 65535, n)
    elif "reflectivity" in names:
        out["reflectivity"] = raw["reflectivity"]
    t = raw["t"].astype(np.int64)
    early = rng.random(n) < 0.01                       # stamped before the header
    if "t" in names:
        out["t"] = raw["t"]
    elif kind in ("hesai", "rslidar"):
        out["timestamp"] = HEADER_TS + np.where(early, -1e-6, t * 1e-9)
    elif kind == "livox":
        out["timestamp"] = HEADER_TS * 1e9 + np.where(early, -1000.0, t.astype(np.float64))
    else:
        out["time"] = np.where(early, -1e-6, t * 1e-9).astype(np.float32)
    if "ring" in names:
        out["ring"] = raw["ring"].astype(out.dtype["ring"])
    if "tag" in names:
        out["tag"] = rng.choice(np.array([0x00, 0x10, 0x20, 0x30, 0x05, 0x15, 0x2a, 0xff], np.uint8), n)
        out["line"] = rng.integers(0, 6, n)
    return out


CFG = dict(point_skip_divisor=2, ring_skip_divisor=2, range_min=1.0, range_max=60.0, z_offset=-0.02)


def test_typed_oracle_is_the_ouster_oracle():
    from mimosa_amd import capi
    raw, _ = synth.make_raw_scan(32, n_cols=256)
    for kw in (dict(), CFG, dict(create_full_res_pointcloud=0)):
        cfg = ref_cpu.make_input_config(**kw)
        a = ref_cpu.prepare_input(raw, cfg)
        b = ref_cpu.prepare_input_typed("ouster", raw, cfg)
        _same_points(_as_points(a["points_full"]), _as_points(b["points_full"]))
        assert np.array_equal(a["geometric_idxs"], b["geometric_idxs"])
        assert np.array_equal(a["unique_ns"], b["unique_ns"])
        assert a["last_point_ns"] == b["last_point_ns"]


def test_point_records_have_the_reference_sizes():
    from mimosa_amd import capi
    for kind in ref_cpu.POINT_KINDS:
        assert capi.point_dtype(kind).itemsize == ref_cpu.point_sizeof(kind)
        L = capi.point_layout(kind)
        assert L.stride == capi.point_dtype(kind).itemsize


@pytest.mark.parametrize("kind", ref_cpu.POINT_KINDS)
def test_typed_oracle_branches(kind):
    """the per-type branches do what the reference's source says: tag filter, no ring filter, wrapped early timestamps"""
    from mimosa_amd import capi
    raw = make_sensor_scan(kind)
    o = ref_cpu.prepare_input_typed(kind, raw, ref_cpu.make_input_config(**CFG), header_ts=HEADER_TS)
    full = _as_points(o["points_full"])
    assert 0 < len(full) < len(raw)
    src = raw[full["idx"]]
    if "tag" in raw.dtype.names:
        assert np.all(np.isin(src["tag"] & 0x30, (0x00, 0x10)))
    geo = src[o["geometric_idxs"].astype(np.int64)]
    if capi.point_layout(kind).ring_filter:
        assert np.all(geo["ring"].astype(np.int64) % 2 == 0)
    elif "ring" in raw.dtype.names:
        assert np.any(geo["ring"].astype(np.int64) % 2 == 1)
    assert full["t"].max() <= 110_000_000          # nothing stamped before the header survived as t = 0 ... or as 2^32 - x
    assert o["last_point_ns"] == full["t"].max()


@pytest.mark.parametrize("kind", ref_cpu.POINT_KINDS)
def test_typed_oracle_against_the_numpy_twin(kind):
    """oracle/ref_cpu.hpp::prepare_input_typed (sequential, per-type structs) and oracle/numpy_ref.py::prepare_input_typed
    (vectorised, index arithmetic) are two independent restatements of manager.cpp:149-383; they must agree bit for bit on
    every point type, with and without the two re-orderings."""
    from oracle import numpy_ref
    cases = [dict(width=128, height=32), dict(organize=True)]
    if kind in ("rslidar", "velodyne_anybotics"):
        cases.append(dict(width=32, height=128, transpose=True))
    for kw in (dict(), CFG, dict(create_full_res_pointcloud=0, point_skip_divisor=3, ring_skip_divisor=3)):
        for order in cases:
            raw = make_sensor_scan(kind, order="col" if (order.get("organize") or order.get("transpose")) else "row")
            a = ref_cpu.prepare_input_typed(kind, raw, ref_cpu.make_input_config(**kw), header_ts=HEADER_TS, **order)
            cfgd = dict(range_min=0.0, range_max=100.0, intensity_min=0.0, intensity_max=1.0e10, ns_max=1.0e9, z_offset=0.0,
                        create_full_res_pointcloud=True, point_skip_divisor=1, ring_skip_divisor=1)
            ic = ref_cpu.make_input_config(**kw)
            for f in cfgd:
                cfgd[f] = type(cfgd[f])(getattr(ic, f))
            b = numpy_ref.prepare_input_typed(kind, raw, header_ts=HEADER_TS, **order, **cfgd)
            pa = _as_points(a["points_full"])
            assert len(pa) == len(b["points_full"]["x"]) > 0
            for k in FIELDS:
                assert np.array_equal(pa[k].view(np.uint32), np.ascontiguousarray(b["points_full"][k]).view(np.uint32)), (kind, order, k)
            assert np.array_equal(a["geometric_idxs"], b["geometric_idxs"].astype(np.uint64))
            assert np.array_equal(a["unique_ns"], b["unique_ns"]) and a["last_point_ns"] == b["last_point_ns"]


def _check(ctx, kind, raw, kw, **order):
    from mimosa_amd import capi
    cfg = capi.make_input_config(**kw)
    o = ref_cpu.prepare_input_typed(kind, raw, ref_cpu.make_input_config(**kw), header_ts=HEADER_TS, width=order.get("width"), height=order.get("height", 1),
                                    transpose=order.get("transpose", False), organize=order.get("organize_by_ring", False))
    sc = capi.Scan(ctx)
    info = sc.prepare_input_layout(raw, capi.point_layout(kind), cfg, header_ts=HEADER_TS, **order)
    assert info["n_in"] == len(raw)
    assert info["n_full"] == len(o["points_full"])
    assert info["n_geometric"] == len(o["geometric_idxs"])
    assert info["last_point_ns"] == o["last_point_ns"]
    assert np.array_equal(sc.unique_ns(), o["unique_ns"])
    _same_points(sc.points(capi.Scan.FULL), _as_points(o["points_full"]))
    assert np.array_equal(sc.indices(0), o["geometric_idxs"].astype(np.uint32))
    return sc, o


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ref_cpu.POINT_KINDS)
@pytest.mark.parametrize("kw", [dict(), CFG, dict(create_full_res_pointcloud=0, point_skip_divisor=3)])
def test_prepare_input_every_point_type(ctx, kind, kw):
    sc, _ = _check(ctx, kind, make_sensor_scan(kind), kw, width=128, height=32)
    sc.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ouster", "ouster_r8", "hesai", "velodyne", "velodyne_anybotics", "rslidar"])
def test_organize_by_ring(ctx, kind):
    """lidar/manager.cpp:205-241: an unorganised (height 1) cloud is bucketed by ring, order inside a ring preserved"""
    raw = make_sensor_scan(kind, rows=64, cols=96, order="col")
    sc, o = _check(ctx, kind, raw, CFG, organize_by_ring=True)
    # the re-ordered cloud is ring-major: surviving points of ring r come before those of ring r + 1
    full = _as_points(o["points_full"])
    assert np.all(np.diff(full["idx"].astype(np.int64)) > 0)
    # with height != 1 the flag does nothing (:210)
    sc2, o2 = _check(ctx, kind, raw, CFG, organize_by_ring=True, width=64, height=96)
    assert not np.array_equal(o2["geometric_idxs"], o["geometric_idxs"]) or len(o["geometric_idxs"]) == 0
    sc.destroy()
    sc2.destroy()


@pytest.mark.gpu
def test_full_size_unorganised_hesai_scan(ctx):
    """128 rings x 1024 columns arriving column by column (the JT128 case organize_by_ring was written for)"""
    raw = make_sensor_scan("hesai", rows=128, cols=1024, order="col")
    sc, o = _check(ctx, "hesai", raw, dict(point_skip_divisor=4, ring_skip_divisor=1), organize_by_ring=True)
    assert len(o["unique_ns"]) > 900
    sc.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["ouster_odyssey", "livox", "livox_custom2"])
def test_organize_by_ring_is_ignored_where_the_reference_ignores_it(ctx, kind):
    raw = make_sensor_scan(kind, rows=16, cols=64)
    a, oa = _check(ctx, kind, raw, CFG, organize_by_ring=True)
    b, ob = _check(ctx, kind, raw, CFG, organize_by_ring=False)
    assert np.array_equal(oa["geometric_idxs"], ob["geometric_idxs"])
    a.destroy()
    b.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["rslidar", "velodyne_anybotics"])
def test_transpose(ctx, kind):
    """lidar/manager.cpp:177-203: a cloud published column-major (width = rings) is transposed first"""
    raw = make_sensor_scan(kind, rows=32, cols=80, order="col")     # 80 rows of 32: width 32, height 80
    sc, o = _check(ctx, kind, raw, CFG, width=32, height=80, transpose=True)
    row_major = raw.reshape(80, 32).T.reshape(-1).copy()
    sc2, o2 = _check(ctx, kind, row_major, CFG, width=80, height=32)
    _same_points(_as_points(o["points_full"]), _as_points(o2["points_full"]))   # the transposed cloud IS the row-major one
    # transpose of a height-1 cloud, then organise (height becomes width != 1 unless width == 1): flag combination
    sc3, _ = _check(ctx, kind, raw, CFG, width=len(raw), height=1, transpose=True, organize_by_ring=True)
    for s in (sc, sc2, sc3):
        s.destroy()


@pytest.mark.gpu
def test_other_sensor_through_the_whole_front_end(ctx):
    """after prepareInput the pipeline is the same for every sensor: deskew + Geometric::preprocess on a Hesai cloud"""
    from mimosa_amd import capi
    raw = make_sensor_scan("hesai", rows=64, cols=256, order="col")
    sc, o = _check(ctx, "hesai", raw, dict(), organize_by_ring=True)
    uns = o["unique_ns"]
    rng = np.random.default_rng(3)
    Rt12 = np.zeros((len(uns), 12), np.float32)
    for g in range(len(uns)):
        Rt12[g, :9] = synth.rot_z(1e-4 * g).astype(np.float32).reshape(-1)
        Rt12[g, 9:] = rng.normal(0, 0.01, 3)
    full = _as_points(o["points_full"])
    desk = ref_cpu.deskew(full, uns, Rt12)
    sc.deskew(Rt12)
    _same_points(sc.points(capi.Scan.FULL), desk)
    R = synth.rot_z(0.1).astype(np.float32)
    t = np.array([0.01, -0.02, 0.03], np.float32)
    body = ref_cpu.transform_f32(desk[o["geometric_idxs"].astype(np.int64)], R, t)
    kept = ref_cpu.downsample(body, 0.5, 20, 0.15)
    info = sc.preprocess_geometric(R, t, 0.5, 20, 0.15)
    assert info["n_downsampled"] == len(kept)
    assert np.array_equal(sc.indices(1), kept)
    _same_points(sc.points(capi.Scan.DOWNSAMPLED), body[kept])
    sc.destroy()


@pytest.mark.gpu
def test_layout_misuse_is_reported(ctx):
    from mimosa_amd import capi
    sc = capi.Scan(ctx)
    cfg = capi.make_input_config()
    raw = make_sensor_scan("hesai", rows=8, cols=32)
    L = capi.point_layout("hesai")
    with pytest.raises(capi.MhError):
        sc.prepare_input_layout(raw, L, cfg, width=7, height=3)                       # width * height != n
    bad = capi.point_layout("hesai")
    bad.off_time = 44                                                                  # the double would end at 52 > 48
    with pytest.raises(capi.MhError):
        sc.prepare_input_layout(raw, bad, cfg)
    bad = capi.point_layout("livox")
    bad.ring_filter = 1                                                                # no ring field to filter on
    with pytest.raises(capi.MhError):
        sc.prepare_input_layout(make_sensor_scan("livox", rows=8, cols=32), bad, cfg)
    raw["ring"][5] = 200                                                               # beyond the reference's 128-entry tables
    with pytest.raises(capi.MhError):
        sc.prepare_input_layout(raw, L, cfg, organize_by_ring=True)
    info = sc.prepare_input_layout(raw, L, cfg, header_ts=HEADER_TS)                   # the scan is still usable
    assert info["n_in"] == len(raw)
    info = sc.prepare_input_layout(raw[:0], L, cfg, width=0, height=1)                 # empty cloud
    assert info["n_full"] == 0
    sc.destroy()


def _EXTRA(base):
    import os
    return [base + i for i in range(int(os.environ.get("MH_FUZZ_EXTRA", "0")))]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(24)) + _EXTRA(100))
def test_random_record_layouts(ctx, seed):
    """The layout descriptor is general: the same sensor data re-packed into a record with the fields in a random order, at
    random (also unaligned) offsets and a random stride must give the result of the reference's typed code on the original."""
    from mimosa_amd import capi
    rng = np.random.default_rng(4200 + seed)
    kind = ref_cpu.POINT_KINDS[int(rng.integers(0, len(ref_cpu.POINT_KINDS)))]
    rows, cols = int(rng.choice([8, 32, 64])), int(rng.choice([33, 128, 200]))
    order = "col" if rng.integers(0, 2) else "row"
    raw = make_sensor_scan(kind, rows=rows, cols=cols, seed=int(rng.integers(0, 1000)), order=order)
    L0, dt0 = capi.point_layout(kind), capi.point_dtype(kind)
    fields = [("x", "off_x"), ("y", "off_y"), ("z", "off_z")]
    fields.append(("reflectivity" if L0.intensity_is_u16 else "intensity", "off_intensity"))
    tname = {capi.TIME_U32_NS: "t", capi.TIME_F64_S_ABS: "timestamp", capi.TIME_F64_NS_ABS: "timestamp", capi.TIME_F32_S: "time"}[L0.time_kind]
    fields.append((tname, "off_time"))
    if L0.ring_kind != capi.RING_NONE:
        fields.append(("ring", "off_ring"))
    if L0.has_tag:
        fields.append(("tag", "off_tag"))
    # random order, random gaps
    perm = rng.permutation(len(fields))
    off, names, formats, offsets = int(rng.integers(0, 5)), [], [], []
    L = capi.point_layout(kind)
    for i in perm:
        name, attr = fields[i]
        fmt = dt0.fields[name][0]
        names.append(name), formats.append(fmt), offsets.append(off)
        setattr(L, attr, off)
        off += fmt.itemsize + int(rng.choice([0, 0, 1, 3, 4, 7]))
    stride = off + int(rng.choice([0, 1, 5, 16]))
    L.stride = stride
    dt = np.dtype({"names": names, "formats": formats, "offsets": offsets, "itemsize": stride})
    packed = np.zeros(len(raw), dt)
    junk = rng.integers(0, 256, size=(len(raw), stride), dtype=np.uint8)     # the gaps hold other fields' bytes in real messages
    packed.view(np.uint8).reshape(len(raw), stride)[:] = junk
    for name in names:
        packed[name] = raw[name]
    kw = dict(point_skip_divisor=int(rng.choice([1, 2, 3, 4])), ring_skip_divisor=int(rng.choice([1, 2, 3])),
              create_full_res_pointcloud=int(rng.integers(0, 2)), range_min=float(rng.choice([0.0, 1.0])), z_offset=float(rng.choice([0.0, -0.03])))
    can_transpose = kind in ("rslidar", "velodyne_anybotics")
    ordr = {}
    if order == "col" and can_transpose and rng.integers(0, 2):
        ordr = dict(width=rows, height=cols, transpose=True)
    elif rng.integers(0, 2):
        ordr = dict(organize_by_ring=True)
    else:
        ordr = dict(width=cols, height=rows) if order == "row" else dict(width=rows, height=cols)
    cfg = capi.make_input_config(**kw)
    o = ref_cpu.prepare_input_typed(kind, raw, ref_cpu.make_input_config(**kw), header_ts=HEADER_TS, width=ordr.get("width"),
                                    height=ordr.get("height", 1), transpose=ordr.get("transpose", False), organize=ordr.get("organize_by_ring", False))
    sc = capi.Scan(ctx)
    info = sc.prepare_input_layout(packed, L, cfg, header_ts=HEADER_TS, **ordr)
    assert info["n_full"] == len(o["points_full"]) and info["n_geometric"] == len(o["geometric_idxs"])
    assert info["last_point_ns"] == o["last_point_ns"]
    assert np.array_equal(sc.unique_ns(), o["unique_ns"])
    _same_points(sc.points(capi.Scan.FULL), _as_points(o["points_full"]))
    assert np.array_equal(sc.indices(0), o["geometric_idxs"].astype(np.uint32))
    sc.destroy()


def test_rotated_patch_locations_against_the_numpy_twin():
    """getGradientBasedLocations / snapPoint (photometric_utils.cpp:453-518): oracle/photo_ref.hpp vs oracle/numpy_photo.py
    on random gradient directions (including exact multiples of 45 degrees, where rounding ties and collisions happen) for the
    5 x 5, 8 x 8 and 3 x 3 patterns."""
    import ctypes as C
    from oracle import numpy_photo, ref_cpu as rc
    from mimosa_amd import synth_photo as sp
    L = rc.lib()
    L.refphoto_gradient_locations.argtypes = [C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(11)
    for patch in (5, 8, 3):
        pat = np.ascontiguousarray(sp.photo_config(patch=patch)["patch_offsets"], np.int32)
        dirs = [(np.cos(a), np.sin(a)) for a in np.arange(0, 2 * np.pi, np.pi / 8)] + [tuple(rng.normal(0, 1, 2)) for _ in range(60)] + [(0.0, 0.0)]
        for gx, gy in dirs:
            out = np.zeros_like(pat)
            L.refphoto_gradient_locations(C.c_float(gx), C.c_float(gy), pat.ctypes.data_as(C.c_void_p), len(pat), out.ctypes.data_as(C.c_void_p))
            tw = numpy_photo.gradient_based_locations(gx, gy, pat)
            assert np.array_equal(out, tw), (patch, gx, gy)
            if gx or gy:                                             # (a zero gradient piles everything onto 9 pixels, there as here)
                assert len({tuple(r) for r in out}) == len(out)
