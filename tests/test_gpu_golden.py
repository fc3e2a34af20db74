"""-m gpu: the HIP path against the committed golden fixtures (tests/golden/, see make_golden.py)."""
import numpy as np
import pytest

from golden_cases import LINEARIZE_CASES, cfg_of, check_linearize, check_state, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", LINEARIZE_CASES)
def test_linearize_golden(ctx, case):
    from mimosa_amd import capi, synth
    m, pts, aux = synth.small_world()
    g = load("linearize_" + case)
    gm = capi.VoxelMap(ctx, mode=int(g["mode"]))
    gm.insert(m)
    f = capi.ICPFactor(ctx, gm, pts, capi.make_reg_config(**cfg_of(g)), binary=bool(int(g["binary"])))
    kw = dict(R_tgt=g["R_tgt"], t_tgt=g["t_tgt"]) if int(g["binary"]) else {}
    r1 = f.linearize(g["R1"], g["t1"], g["g_unit"], **kw)
    check_linearize(g, "a", r1)
    check_state(g, 1, *f.state())
    r2 = f.linearize(g["R2"], g["t2"], g["g_unit"], **kw)
    check_linearize(g, "b", r2)
    check_state(g, 2, *f.state())
    assert r1["n_exact_fallback"] == 0 and r2["n_exact_fallback"] == 0
    assert r1["mean_scanned"] < r1["mean_candidates"]  # box pruning did skip voxels


def test_deskew_golden_bit_exact(ctx):
    from mimosa_amd import synth
    g = load("deskew")
    pts = np.zeros(len(g["xyz"]), synth.POINT_DTYPE)
    pts["x"], pts["y"], pts["z"], pts["t"] = g["xyz"][:, 0], g["xyz"][:, 1], g["xyz"][:, 2], g["t"]
    d = ctx.deskew(pts, g["unique_ns"], g["Rt12"])
    assert np.array_equal(synth.points_xyz(d), g["deskewed"])
    b = ctx.deskew(pts, g["unique_ns"], g["Rt12"], g["R_B_L"], g["t_B_L"])
    assert np.array_equal(synth.points_xyz(b), g["body"])


def test_map_lru_golden(ctx):
    from mimosa_amd import capi
    g = load("map_lru")
    gm = capi.VoxelMap(ctx, lru_horizon=2, lru_clear_cycle=2)
    sizes = []
    for c in np.array_split(g["chunks"], int(g["n_chunks"])):
        gm.insert(c)
        sizes.append(gm.stats()["n_points"])
    assert sizes == list(g["sizes"])
    assert np.array_equal(gm.get_cloud(), g["cloud"])
    # k-NN still consistent after the purge renumbered the voxels
    pts, sq, found = gm.knn(g["cloud"][:50].astype(np.float64), 1)
    assert (found == 1).all() and np.abs(sq[:, 0]).max() == 0.0


def test_frontend_golden_bit_exact(ctx):
    """Device scan front end vs the committed fixture: every stage, every bit, same order."""
    from mimosa_amd import capi, synth

    g = load("frontend")
    raw = np.frombuffer(np.ascontiguousarray(g["raw"]).tobytes(), dtype=synth.OUSTER_DTYPE)
    kw = {str(k): float(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    for k in ("create_full_res_pointcloud", "point_skip_divisor", "ring_skip_divisor"):
        kw[k] = int(kw[k])
    sc = capi.Scan(ctx)
    info = sc.prepare_input(raw, capi.make_input_config(**kw))
    full = sc.points(capi.Scan.FULL)
    assert info["last_point_ns"] == int(g["last_point_ns"])
    assert np.array_equal(synth.points_xyz(full).view(np.uint32), g["full_xyz"].view(np.uint32))
    assert np.array_equal(full["t"], g["full_t"]) and np.array_equal(full["idx"], g["full_idx"])
    assert np.array_equal(full["range"].view(np.uint32), g["full_range"].view(np.uint32))
    assert np.array_equal(sc.indices(0), g["geometric_idxs"].astype(np.uint32))
    assert np.array_equal(sc.unique_ns(), g["unique_ns"])
    sc.deskew(g["Rt12"])
    assert np.array_equal(synth.points_xyz(sc.points(capi.Scan.FULL)).view(np.uint32), g["deskewed"].view(np.uint32))
    sc.preprocess_geometric(g["R_B_L"], g["t_B_L"], 1.0, 3, 0.5)
    assert np.array_equal(synth.points_xyz(sc.points(capi.Scan.BODY)).view(np.uint32), g["body"].view(np.uint32))
    assert np.array_equal(sc.indices(1), g["kept"].astype(np.uint32))
    sc.destroy()
