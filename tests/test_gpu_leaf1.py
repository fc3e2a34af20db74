"""-m gpu: the parameter block most shipped configurations use — 1 m leaf, 0.2 m minimum distance
(/root/reference/mimosa/config/hornbill/params.yaml:86-95; the same in euroc, lapwing, magpie, parrot) — at FULL size: the
131 072-point OS0-128 scan against a >= 5 M-point map built with that block (20 rooms, walls sampled every 0.1 m: 98 % of the
voxels sit AT the 20-point cap, a query's 19-neighbourhood holds ~190 candidates on average and up to 380).  The regime the
box pruning and the proof check of the k-NN were not tuned on.  HIP path through the C ABI vs the CPU oracle
(geometric_factor.hpp:231-562): map point for point, cold linearize, a re-linearization across the association threshold,
the per-point state."""
import numpy as np
import pytest

from parity import assert_result_parity, assert_state_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    from mimosa_amd import synth
    rooms = [xyz for _, _, xyz in synth.make_hornbill_rooms()]
    pts, _ = synth.make_scan(128)
    R, t = synth.query_pose()
    return dict(rooms=rooms, pts=pts, R=R, t=t, cfg=synth.hornbill_config())


def test_leaf1_full_size_against_the_oracle(ctx, world):
    from mimosa_amd import capi, synth, synth_hostile as sh
    from oracle import ref_cpu
    w = world
    cfg = w["cfg"]
    kw = dict(leaf=cfg["target_ivox_map_leaf_size"], min_dist=cfg["target_ivox_map_min_dist_in_voxel"], max_pts=synth.MAX_PTS_PER_VOXEL,
              mode=synth.ENWIDE_NEIGHBOR_MODE, lru_horizon=synth.ENWIDE_LRU_HORIZON)
    gm, rm = capi.VoxelMap(ctx, **kw), ref_cpu.Map(**kw)
    for xyz in w["rooms"]:
        gm.insert(xyz)
        rm.insert(xyz)
    s = gm.stats()
    assert s["n_points"] == rm.num_points and s["n_voxels"] == rm.num_voxels and s["n_points"] >= 5_000_000
    cloud = gm.get_cloud()
    assert np.array_equal(cloud, rm.export()[2])
    fill = sh.voxel_fill_stats(cloud, cfg["target_ivox_map_leaf_size"], synth.MAX_PTS_PER_VOXEL)
    assert fill["share_at_cap"] > 0.9
    gf = capi.ICPFactor(ctx, gm, w["pts"], capi.make_reg_config(**cfg))
    rf = ref_cpu.ICP(rm, w["pts"], ref_cpu.make_config(**cfg))
    R, t = w["R"], w["t"]
    g, r = gf.linearize(R, t), rf.linearize(R, t)
    assert g["n_knn"] == 131072 and g["mean_candidates"] > 150          # ~2.3x the ENWIDE world's 83
    assert_result_parity(g, r)
    assert_state_parity(gf.state(), rf.state())
    dR = synth.so3_exp(np.array([0.0, 0.0, 2.0e-3]))                     # min_dist / 4 = 5 cm here: points beyond ~25 m re-associate
    R2, t2 = R @ dR, t + np.array([0.03, -0.02, 0.005])
    g2, r2 = gf.linearize(R2, t2), rf.linearize(R2, t2)
    assert 0 < g2["n_knn"] < 131072
    assert_result_parity(g2, r2)
    assert_state_parity(gf.state(), rf.state())
    print("leaf 1.0 world:", dict(map_points=s["n_points"], voxels=s["n_voxels"], share_at_cap=round(fill["share_at_cap"], 4), mean_candidates=round(g["mean_candidates"], 1),
                                  mean_scanned=round(g["mean_scanned"], 1), exact_fallback=g["n_exact_fallback"], status_hist=g["status_hist"].tolist()))
    gf.destroy()
    gm.release()
