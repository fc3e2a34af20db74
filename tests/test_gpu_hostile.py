"""-m gpu: the hostile world (mimosa_amd/synth_hostile.py) at FULL size against the CPU oracle: a 131 072-point OS0-128
scan of a cluttered room registered to a map that is the union of 50 past ray-cast scans (2 x 5 rooms x 5 poses, one iVox
insert each) — 1 / r^2 sampling density, saturated voxels next to the past poses, sparse far walls and shadows, thin plates
and poles.  Every branch of ICPFactor::linearize (geometric_factor.hpp:231-562) is populated: InsufficientCorresPoints,
CorresMaxDist, MinEigenValueLow, Line, CorresPlaneInvalid, MaxError, Valid."""
import os

import numpy as np
import pytest

from parity import assert_result_parity, assert_state_parity

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hostile(ctx):
    from mimosa_amd import capi, synth, synth_hostile as sh
    from oracle import ref_cpu

    gm, rm = capi.VoxelMap(ctx), ref_cpu.Map()
    n_in = 0
    for _, _, _, hits in sh.make_map_scans(2, 5, 5, workers=min(32, os.cpu_count() or 1)):
        gm.insert(hits)
        rm.insert(hits)
        n_in += len(hits)
    pts, aux = sh.make_query_scan()
    R, t = synth.query_pose(aux["R_W_L"], aux["t_W_L"])
    yield dict(gm=gm, rm=rm, pts=pts, R=R, t=t, n_in=n_in, cfg=synth.enwide_config())
    gm.release()


def test_hostile_map_identical_and_saturated(hostile):
    from mimosa_amd import synth_hostile as sh
    gm, rm = hostile["gm"], hostile["rm"]
    s = gm.stats()
    assert s["n_points"] == rm.num_points and s["n_voxels"] == rm.num_voxels
    assert hostile["n_in"] == 50 * 131072 and 1_500_000 < s["n_points"] < hostile["n_in"] // 2   # the greedy rule rejects most of a dense past scan
    cloud = gm.get_cloud()
    assert np.array_equal(cloud, rm.export()[2])
    fill = sh.voxel_fill_stats(cloud)
    assert fill["share_at_cap"] > 0.001 and fill["share_below_5"] > 0.2   # saturated voxels AND sparse ones in one map


def test_hostile_cold_and_relinearize(ctx, hostile):
    from mimosa_amd import capi, synth
    from oracle import ref_cpu

    h = hostile
    gf = capi.ICPFactor(ctx, h["gm"], h["pts"], capi.make_reg_config(**h["cfg"]))
    rf = ref_cpu.ICP(h["rm"], h["pts"], ref_cpu.make_config(**h["cfg"]))
    g, r = gf.linearize(h["R"], h["t"]), rf.linearize(h["R"], h["t"])
    assert g["n_knn"] == 131072
    hist = np.asarray(r["status_hist"])
    assert hist[1] > 500 and hist[4] > 1000 and hist[5] > 10000 and hist[6] > 100 and hist[8] > 50000   # the branches are populated
    assert_result_parity(g, r)
    assert_state_parity(gf.state(), rf.state())
    R2, t2 = h["R"] @ synth.so3_exp(np.array([0.0, 0.0, 1.5e-3])), h["t"] + np.array([0.02, -0.015, 0.004])
    g2, r2 = gf.linearize(R2, t2), rf.linearize(R2, t2)
    assert 0 < g2["n_knn"] < 131072
    assert_result_parity(g2, r2)
    assert_state_parity(gf.state(), rf.state())
    gf.destroy()
