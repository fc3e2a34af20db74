"""-m gpu: the NATIVE map-sharded factor (mh_map_insert_shard + mh_shard_icp_*, mimosa_amd/csrc/shard_api.hip) at FULL size
against the ORACLE (not against the unsharded HIP factor): the 131 072-point OS0-128 scan vs the configs[1] map (~4.96 M
points, ten iVox inserts), the map hash-sharded over 2 and over 8 ranks.  Ranks are threads of this process over the
in-process transport (a one-GPU box cannot run RCCL with more than one rank); every rank's GLOBAL result must equal the
unsharded oracle (include/mimosa/lidar/geometric_factor.hpp:231-562) and the state of the points it holds the oracle's
state of those points — cold, and after a pose step that moves part of the points past the data-association threshold and
some of them across shard-block faces.  The window form (mh_shard_icp_linearize_batch: two whole scans per protocol round,
rounds pipelined) runs at 8 ranks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(big_world, seed_offset=0):
    from mimosa_amd import synth
    w = big_world
    if seed_offset:
        pts, _ = synth.make_scan(128, seed=synth.BASE_SEED + 1 + seed_offset)
    else:
        pts = w["pts"]
    R, t = w["R"], w["t"]
    dR = synth.so3_exp(np.array([0.0, 0.0, 1.5e-3]))  # 1.5 mrad of yaw: points beyond ~15 m pass the 3.75 cm threshold
    poses = [(R, t), (R @ dR, t + np.array([0.02, -0.015, 0.004])), (R @ dR, t + np.array([0.02, -0.015, 0.004]))]
    return dict(map_chunks=w["map_rooms"], scan=pts, cfg=w["cfg"], poses=poses, tgt=None, binary=False)


@pytest.mark.parametrize("world", [2, 8])
def test_full_size_sharded_factor_equals_the_oracle(big_world, world):
    import shard_native_common as C
    case = _case(big_world)
    assert len(case["scan"]) == 131072
    results, n_map = C.run_local_world(world, case=case)
    assert n_map > 4_900_000
    held = [r["stats"]["n_live"] for r in results]
    stored = [r["map_points"] for r in results]
    assert sum(held) == 131072
    for r in results:
        assert 0 < r["map_points"] < n_map           # a shard with its halo, not the map
        assert r["moved"][0] > 0 and r["stats"]["collective"] == 1
    # the partition the design documents: halo storage and the fullest rank
    print(f"world {world}: points held per rank {held}; stored map points per rank {stored} of {n_map} "
          f"(total {sum(stored) / n_map:.2f}x, fullest {max(stored) / n_map:.1%})")


def test_full_size_sharded_window_batch_pipelined_equals_the_oracle(big_world):
    """Two whole scans per protocol round over 8 ranks, every round of the pose sequence in flight before the one wait."""
    import shard_native_common as C
    C.run_local_world_batch(8, [_case(big_world), _case(big_world, seed_offset=1000)], pipelined=True, uneven=False)


# ---- BASELINE configs[2]: the ~50 M-point map through the NATIVE sharded path -----------------------------------------------------
def test_configs2_native_sharded_path_equals_the_oracle(huge_world):
    """The 10 x 10-room map (~50 M points) hash-sharded over 8 ranks with mh_map_insert_shard (each rank stores its blocks + the
    one-voxel halo: ~11 GB in all on the one GPU), the 131 072-point scan through mh_shard_icp_linearize: every rank's GLOBAL
    result and the state of the points it holds against the ORACLE on the whole map — cold, after a pose step that re-associates
    part of the cloud and moves points across shard-block faces, and once more at the same pose (all cache hits)."""
    import shard_native_common as C
    case = _case(huge_world)
    assert len(case["scan"]) == 131072 and len(case["map_chunks"]) == 100
    results, n_map = C.run_local_world(8, case=case)
    assert n_map > 49_000_000
    held = [r["stats"]["n_live"] for r in results]
    stored = [r["map_points"] for r in results]
    assert sum(held) == 131072
    for r in results:
        assert 0 < r["map_points"] < n_map and r["stats"]["collective"] == 1
    print(f"configs[2], world 8: points held per rank {held}; stored map points per rank {stored} of {n_map} "
          f"(total {sum(stored) / n_map:.2f}x, fullest {max(stored) / n_map:.1%})")


def test_configs2_native_window_batch_async_equals_the_oracle(huge_world):
    """The throughput form bench.py reports at N > 1 (mh_shard_icp_linearize_batch_async: two whole scans per protocol round, every
    round of the pose sequence in flight before the one wait) on the configs[2] map over 8 ranks, against the oracle."""
    import shard_native_common as C
    C.run_local_world_batch(8, [_case(huge_world), _case(huge_world, seed_offset=1000)], pipelined=True, uneven=False)
