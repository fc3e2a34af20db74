"""-m gpu: the wave-cooperative EXACT fallback of the k-NN (icp_kernels.hip: knn_query, "exact fallback") forced by construction.

The coarse tier ranks candidates on a 10-bit copy of the buckets (error up to ~0.9 grid units = 0.44 mm at the 0.5 m leaf) and keeps
8 (k = 5) or 12 (generic k) survivors; a proof check decides whether a non-survivor could still belong to the answer, and the lane
it does not cover re-runs KnnResult::push over the scanned voxels with the whole wave.  On ordinary worlds that is 0-1 query
in 131 072, so this test builds queries for which it MUST happen: 20 map points on a sphere shell of radius 0.23 m around each
query (the vertices of a randomly rotated dodecahedron, radii 10 um apart — far inside the coarse tier's error band — spread
over the centre voxel and its face / edge neighbours).  The coarse keys of the 20 are quantisation noise, so the survivors are
a random subset and the answer can only come out right through the fallback: parity of mh_map_knn and of linearize with the
oracle (src/lidar/incremental_voxel_map.cpp:26-32 semantics: ascending distance, ties by traversal order) fails with near
certainty if that branch is removed, and n_exact_fallback is asserted directly."""
import numpy as np
import pytest

from parity import assert_result_parity, assert_state_parity

pytestmark = pytest.mark.gpu


def _dodecahedron():
    p = (1 + 5 ** 0.5) / 2
    v = [(x, y, z) for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)]
    for a in (-1 / p, 1 / p):
        for b in (-p, p):
            v += [(0, a, b), (a, b, 0), (b, 0, a)]
    v = np.array(v, float)
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def _shell_world(small_world, n_clusters, seed):
    """small_world's map + n_clusters shells far away from it; the scan = small_world's points + one query per shell."""
    from mimosa_amd import synth
    rng = np.random.default_rng(seed)
    D = _dodecahedron()
    R, t = small_world["R"], small_world["t"]
    shells, queries = [], []
    for c in range(n_clusters):
        vox = np.array([260 + 6 * (c % 12), 6 * ((c // 12) % 12) - 30, 6 * (c // 144) - 4], float)   # voxels, 3 m apart
        q = (vox + 0.5) * 0.5 + rng.uniform(-0.1, 0.1, 3)
        p32 = (R.T @ (q - t)).astype(np.float32)               # the source point; what the kernel sees is R p + t in fp64
        q = R @ p32.astype(np.float64) + t
        Rr = synth.so3_exp(rng.normal(0, 1.0, 3))
        rad = 0.23 + 1e-5 * rng.permutation(20)
        shells.append((q + (D @ Rr.T) * rad[:, None]).astype(np.float32))
        queries.append(p32)
    m = np.concatenate([small_world["map_xyz"]] + shells)
    base = small_world["pts"]                                  # mh_point32 records
    extra = np.zeros(n_clusters, dtype=base.dtype)
    qa = np.array(queries)
    extra["x"], extra["y"], extra["z"] = qa[:, 0], qa[:, 1], qa[:, 2]
    return m, np.concatenate([base, extra]), np.array([R @ p.astype(np.float64) + t for p in queries])


@pytest.mark.parametrize("mode", [7, 19, 27])
def test_knn_with_the_proof_check_failing(ctx, small_world, mode):
    from mimosa_amd import capi
    from oracle import ref_cpu
    m, _, q = _shell_world(small_world, 150, seed=mode)
    gm, rm = capi.VoxelMap(ctx, mode=mode), ref_cpu.Map(mode=mode)
    gm.insert(m)
    rm.insert(m)
    assert np.array_equal(gm.get_cloud(), rm.export()[2])
    for k in (1, 5, 8):
        pts, sq, found = gm.knn(q, k)
        idx, sq_r, found_r, ncand = rm.knn(q, k)
        assert np.array_equal(found, found_r) and found.min() == k
        assert np.median(ncand) >= 13                             # more candidates in the band than the coarse tier keeps
        for i in range(len(q)):
            assert np.array_equal(sq[i], sq_r[i]), (k, i, sq[i] - sq_r[i])
            for j in range(k):
                assert np.array_equal(pts[i, j], rm.point(idx[i, j])), (k, i, j)
    gm.release()


@pytest.mark.parametrize("mode,k", [(19, 5), (7, 5), (27, 5), (19, 8), (27, 3)])
def test_linearize_takes_the_exact_fallback_and_agrees_with_the_oracle(ctx, small_world, mode, k):
    from mimosa_amd import capi
    from oracle import ref_cpu
    n_clusters = 200
    m, pts, _ = _shell_world(small_world, n_clusters, seed=100 + mode + k)
    cfg = dict(small_world["cfg"], num_corres_points=k)
    gm, rm = capi.VoxelMap(ctx, mode=mode), ref_cpu.Map(mode=mode)
    gm.insert(m)
    rm.insert(m)
    gf = capi.ICPFactor(ctx, gm, pts, capi.make_reg_config(**cfg))
    rf = ref_cpu.ICP(rm, pts, ref_cpu.make_config(**cfg))
    got, want = gf.linearize(small_world["R"], small_world["t"]), rf.linearize(small_world["R"], small_world["t"])
    # nearly every shell query must have gone through the exact pass (and a few ordinary ones may have)
    assert got["n_exact_fallback"] >= int(0.9 * n_clusters), got["n_exact_fallback"]
    assert_result_parity(got, want)
    assert_state_parity(gf.state(), rf.state())
    # ... and again after a pose step that re-associates part of the points (the fallback on warm calls)
    from mimosa_amd import synth
    R2, t2 = small_world["R"] @ synth.so3_exp(np.array([2e-4, -1e-4, 3e-4])), small_world["t"] + np.array([0.06, -0.04, 0.02])
    assert_result_parity(gf.linearize(R2, t2), rf.linearize(R2, t2))
    assert_state_parity(gf.state(), rf.state())
    gf.destroy()
    gm.release()
