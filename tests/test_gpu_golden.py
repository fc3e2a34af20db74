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
