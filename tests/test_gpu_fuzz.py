"""-m gpu: randomised configurations of the hot path against the oracle — room size, map density (sparse to
saturated voxels), leaf size, neighbour mode, k, Huber on/off, 4-DoF, pose error, and a re-linearization walk
whose steps straddle the data-association threshold (a mix of cached and re-associated points per call)."""
import numpy as np
import pytest

from mimosa_amd import synth
from parity import assert_result_parity, assert_state_parity

pytestmark = pytest.mark.gpu


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    room = np.array([rng.uniform(5, 14), rng.uniform(4, 11), rng.uniform(2.5, 5)])
    grid = float(rng.choice([0.07, 0.11, 0.16, 0.3, 0.45]))       # 0.07: saturated 20-point voxels; 0.45: ~1 point / voxel
    leaf = float(rng.choice([0.3, 0.5, 1.0]))
    m = synth.make_room(4000 + seed, 0, 0, grid=grid, room=room)
    loc = np.array([rng.uniform(1.5, room[0] - 1.5), rng.uniform(1.5, room[1] - 1.5), rng.uniform(0.8, room[2] - 0.8)])
    pts, aux = synth.make_scan(n_rows=int(rng.choice([8, 16, 32])), seed=5000 + seed, n_cols=int(rng.choice([64, 128])),
                               room=room, sensor_local=loc)
    cfg = synth.enwide_config()
    cfg.update(target_ivox_map_leaf_size=leaf, source_voxel_grid_filter_leaf_size=leaf,
               target_ivox_map_min_dist_in_voxel=float(rng.choice([0.02, 0.1, 0.15])),
               num_corres_points=int(rng.choice([5, 5, 5, 8, 3])), use_huber=int(rng.integers(0, 2)),
               reg_4_dof=int(rng.integers(0, 2)), max_corres_distance=float(rng.choice([0.3, 1.0, 2.24])),
               plane_validity_distance=float(rng.choice([0.04, 0.07, 0.2])),
               project_on_degneneracy=int(rng.integers(0, 2)), degen_thresh_rot=float(rng.choice([0.0, 10.0])),
               degen_thresh_trans=float(rng.choice([15.0, 40.0])))
    mode = int(rng.choice([7, 19, 19, 27]))
    dR = synth.so3_exp(np.deg2rad(rng.normal(0, 1.0, 3)))
    R = aux["R_W_L"] @ dR
    t = aux["t_W_L"] + aux["R_W_L"] @ rng.normal(0, 0.06, 3)
    return m, pts, cfg, mode, R, t, rng


@pytest.mark.parametrize("seed", range(14))
def test_random_configuration(ctx, seed):
    from mimosa_amd import capi
    from oracle import ref_cpu

    m, pts, cfg, mode, R, t, rng = _case(seed)
    leaf, md = cfg["target_ivox_map_leaf_size"], cfg["target_ivox_map_min_dist_in_voxel"]
    gm = capi.VoxelMap(ctx, leaf=leaf, min_dist=md, mode=mode)
    rm = ref_cpu.Map(leaf=leaf, min_dist=md, mode=mode)
    for chunk in np.array_split(m, 2):
        gm.insert(chunk)
        rm.insert(chunk)
    assert gm.stats()["n_points"] == rm.num_points
    gf = capi.ICPFactor(ctx, gm, pts, capi.make_reg_config(**cfg))
    rf = ref_cpu.ICP(rm, pts, ref_cpu.make_config(**cfg))
    g_unit = np.array([0.02, -0.01, -1.0])
    g_unit /= np.linalg.norm(g_unit)
    thr = md / 4.0
    for step in range(4):
        got, ref = gf.linearize(R, t, g_unit), rf.linearize(R, t, g_unit)
        assert_result_parity(got, ref, check_eigvec=False)
        assert_state_parity(gf.state(), rf.state())
        # next pose: translation around the DA threshold so that only part of the cloud re-associates,
        # plus a rotation that moves far points more than near ones
        t = t + rng.normal(0, 1.0, 3) * thr * rng.choice([0.3, 1.0, 3.0])
        R = R @ synth.so3_exp(rng.normal(0, 1.0, 3) * thr / 5.0)
    gf.destroy()
    gm.release()
