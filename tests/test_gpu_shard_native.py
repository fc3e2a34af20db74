"""-m gpu: the native map-sharded factor (mh_shard_*, mimosa_amd/csrc/shard_api.hip) — the exchange inside the library.
World > 1 runs over the in-process transport (ranks = threads on the one GPU of the box): the whole protocol — routing,
fixed-size segments, tombstones, device-side slot counts, all-reduced sums, retries on overflow, compaction — against the
unsharded oracle.  World 1 over RCCL (the only size a one-GPU box offers to RCCL) runs in a fresh process with the full
protocol forced."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_native_sharded_equals_unsharded_oracle(world):
    import shard_native_common as C
    results, n_map = C.run_local_world(world)
    for r, res in enumerate(results):
        st = res["stats"]
        assert st["world"] == world and st["rank"] == r and st["collective"] == 1
        if world > 1:
            assert 0 < res["map_points"] < n_map        # a real shard (with halo), not the whole map
            assert res["moved"][0] > 0                   # the cold call routes points
            assert res["moved"][-1] == 0                 # the last pose repeats the one before: nothing moves
    # every call but retries is 1 all-to-all + 2 all-reduces (components on)
    assert results[0]["stats"]["collectives_last"] == 3


def test_native_sharded_uneven_shares_and_components_off():
    import shard_native_common as C
    results, _ = C.run_local_world(3, uneven=True, components_off_from=2)
    assert results[0]["stats"]["collectives_last"] == 2   # no K4, no second all-reduce


def test_native_sharded_small_blocks_and_4dof():
    import shard_native_common as C
    C.run_local_world(2, case=C.default_case(dict(reg_4_dof=1)), block_log2=2)


def test_native_sharded_binary_factor():
    import shard_native_common as C
    C.run_local_world(2, case=C.default_case(binary=True))


def test_native_sharded_segment_overflow_retries():
    """A pose jump of metres after the segments have shrunk to their minimum: far more points change owner than a segment
    holds, every rank repeats the call with larger segments, and the result is still the unsharded one."""
    import shard_native_common as C
    from mimosa_amd import synth
    case = C.default_case()
    R0, t0 = case["poses"][0]
    case["poses"] = [(R0, t0), (R0, t0), (R0, t0), (R0 @ synth.so3_exp(np.array([0, 0, 0.3])), t0 + np.array([0.4, -0.3, 0.0])), (R0, t0)]
    results, _ = C.run_local_world(2, case=case)
    assert results[0]["stats"]["retries_total"] >= 1


@pytest.mark.parametrize("world", [1, 2, 3])
def test_native_sharded_async_form_is_the_blocking_one(world):
    """mh_shard_icp_linearize_async + mh_shard_icp_wait per pose: the same protocol round, the same results."""
    import shard_native_common as C
    C.run_local_world(world, mode="async")


@pytest.mark.parametrize("world", [1, 2, 8])
def test_native_sharded_pipelined_calls_are_the_sequential_ones(world):
    """Every pose of the sequence enqueued before the one wait (6 rounds in flight): one stream, so the calls — and the
    data-association cache they share — run in order; results and final state are the unsharded oracle's."""
    import shard_native_common as C
    C.run_local_world(world, mode="pipelined")


def test_native_sharded_pipelined_overflow_is_repeated_at_the_wait():
    """Three blocking calls shrink the segments to their minimum; then a pose jump of metres and the way back are enqueued
    TOGETHER: both overflow, both are repeated (larger segments) at the wait, every rank in the same order."""
    import shard_native_common as C
    from mimosa_amd import synth
    case = C.default_case()
    R0, t0 = case["poses"][0]
    case["poses"] = [(R0, t0), (R0, t0), (R0, t0), (R0 @ synth.so3_exp(np.array([0, 0, 0.3])), t0 + np.array([0.4, -0.3, 0.0])), (R0, t0)]
    # (the k-NN counters of a call repeated BEHIND a later one are statistics: points it had counted re-associate in the repeat)
    results, _ = C.run_local_world(2, case=case, mode="pipelined", sync_first=3, check_counters=False)
    assert results[0]["stats"]["retries_total"] >= 1


def _window_cases(n, binary_at=()):
    import shard_native_common as C
    from mimosa_amd import synth
    base = C.default_case()
    cases = []
    for i in range(n):
        c = C.default_case(binary=(i in binary_at))
        room = np.array([20.0, 14.0, 3.0])
        scan, aux = synth.make_scan(n_rows=[32, 16, 32, 8, 32][i % 5], seed=200 + i, n_cols=[128, 64, 96, 128, 32][i % 5], room=room,
                                    sensor_local=np.array([9.3 - 0.7 * i, 6.6 + 0.4 * i, 1.2]))
        R, t = synth.query_pose(aux["R_W_L"], aux["t_W_L"])
        poses = [(R, t)] + [(R @ synth.so3_exp(w * (1 + 0.3 * i)), t + d * (1 + 0.2 * i)) for w, d in C.POSE_STEPS]
        if c["binary"]:
            Rt, tt = c["tgt"]
            poses = [(Rt @ Rk, Rt @ tk + tt) for Rk, tk in poses]
        cases.append(dict(c, scan=scan, poses=poses, map_chunks=base["map_chunks"]))
    return cases


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_native_sharded_window_batch_equals_unsharded_oracle(world):
    """mh_shard_icp_linearize_batch: five factors (different scans, sizes and pose walks) in ONE protocol round per pose —
    one all-to-all, one all-reduce of 5 x 168 doubles (+ one of 5 x 16) — each equal to ITS unsharded oracle."""
    import shard_native_common as C
    results = C.run_local_world_batch(world, _window_cases(5))
    for st in results[0]["stats"]:
        assert st["collectives_last"] == 3


def test_native_sharded_window_batch_mixed_kinds_and_components():
    """Unary and binary factors in one round (two kernel instantiations), the component pass off for some of them."""
    import shard_native_common as C
    C.run_local_world_batch(2, _window_cases(4, binary_at=(1, 3)), components_off=(0, 3))


def test_native_sharded_window_batch_of_ten_and_pipelined():
    """More factors than one batched launch carries (8): two launches per stage, still one round; and the rounds of the whole
    pose sequence in flight at once."""
    import shard_native_common as C
    C.run_local_world_batch(2, _window_cases(10), pipelined=True)


def test_native_world1_is_the_plain_factor(ctx, small_world):
    from mimosa_amd import capi
    w = small_world
    comm = capi.ShardComm.local(1)[0]
    vmap = capi.VoxelMap(ctx)
    capi.map_insert_shard(ctx, vmap, w["map_xyz"], 1, 0)
    rc = capi.make_reg_config(**w["cfg"])
    f = capi.ShardedICPFactor(ctx, comm, vmap, w["pts"], rc)
    g = capi.ICPFactor(ctx, vmap, w["pts"], rc)
    a, b = f.linearize(w["R"], w["t"]), g.linearize(w["R"], w["t"])
    assert np.array_equal(a["H_ss"], b["H_ss"]) and np.array_equal(a["b_s"], b["b_s"]) and a["f"] == b["f"]
    st = f.stats()
    assert st["collective"] == 0 and st["collectives_last"] == 0 and st["n_live"] == len(w["pts"])
    f.destroy()
    g.destroy()
    vmap.release()
    comm.destroy()


def test_context_shut_down_before_its_communicator(small_world):
    """mh_shutdown(ctx) while a communicator still holds exchange buffers and a publish ring on ctx: the communicator lets go
    of the context there (shard_ctx_gone), so mh_shard_comm_destroy afterwards touches nothing of the dead context — and the
    same communicator bound to a NEW context in between gives the same results."""
    from mimosa_amd import capi
    w = small_world
    rc = capi.make_reg_config(**w["cfg"])

    def one_life(comm, pipelined_tail):
        c = capi.Context(0)
        vmap = capi.VoxelMap(c)
        capi.map_insert_shard(c, vmap, w["map_xyz"], 1, 0)
        f = capi.ShardedICPFactor(c, comm, vmap, w["pts"], rc, force_collectives=True)
        out = f.linearize(w["R"], w["t"])
        if pipelined_tail:
            f.linearize_async(w["R"], w["t"])  # a round nobody waits for: dropped with the factor
        f.destroy()
        vmap.release()
        c.close()  # the communicator is still alive and was bound to c
        assert c.h is None
        return out

    comm = capi.ShardComm.local(1)[0]
    a = one_life(comm, True)
    b = one_life(comm, False)
    assert np.array_equal(a["H_ss"], b["H_ss"]) and np.array_equal(a["b_s"], b["b_s"]) and a["f"] == b["f"]
    comm.destroy()  # after both contexts are gone


def test_factor_destroyed_behind_a_drain_with_its_capacity_update_parked(small_world):
    """factor A has a round in flight; a window batch of two OTHER factors is wider than the publish ring, so
    its enqueue drains the rounds — A's round completes and its segment-capacity decision is PARKED in the communicator.  A is
    destroyed without a wait (the API allows it); the next settle must not apply the parked decision to freed memory."""
    from mimosa_amd import capi
    w = small_world
    rc = capi.make_reg_config(**w["cfg"])
    c = capi.Context(0)
    comm = capi.ShardComm.local(1)[0]
    vmap = capi.VoxelMap(c)
    capi.map_insert_shard(c, vmap, w["map_xyz"], 1, 0)
    ref = capi.ShardedICPFactor(c, comm, vmap, w["pts"], rc, force_collectives=True)
    want = ref.linearize(w["R"], w["t"])
    for rep in range(6):
        a = capi.ShardedICPFactor(c, comm, vmap, w["pts"], rc, force_collectives=True)
        b1 = capi.ShardedICPFactor(c, comm, vmap, w["pts"], rc, force_collectives=True)
        b2 = capi.ShardedICPFactor(c, comm, vmap, w["pts"][: len(w["pts"]) // 2], rc, force_collectives=True)
        a.linearize_async(w["R"], w["t"])                       # ring width 1: a round in flight
        outs = capi.sharded_linearize_batch_async([b1, b2], [w["R"]] * 2, [w["t"]] * 2)   # B = 2 > ring width: drains, parks A's update
        a.destroy()                                             # ... which now points at a destroyed factor
        junk = [np.zeros(1 << 16) for _ in range(8)]            # (churn the host heap over the freed handle)
        b1.wait()                                               # settle_rounds -> apply_cap_updates
        got = outs.results()[0]
        assert np.array_equal(got["H_ss"], want["H_ss"]) and got["f"] == want["f"]
        del junk
        b1.destroy()
        b2.destroy()
    ref.destroy()
    vmap.release()
    comm.destroy()
    c.close()


def test_sharded_factor_outlives_its_context(small_world):
    """mh_shutdown(ctx) with a sharded factor of ctx still alive.  The factor gives its device memory back in
    the shutdown (while the context exists); afterwards every entry point refuses the handle and destroy only deletes it."""
    from mimosa_amd import capi
    w = small_world
    rc = capi.make_reg_config(**w["cfg"])
    for collective in (True, False):
        c = capi.Context(0)
        comm = capi.ShardComm.local(1)[0]
        vmap = capi.VoxelMap(c)
        capi.map_insert_shard(c, vmap, w["map_xyz"], 1, 0)
        f = capi.ShardedICPFactor(c, comm, vmap, w["pts"], rc, force_collectives=collective)
        f.linearize(w["R"], w["t"])
        f.linearize_async(w["R"], w["t"])   # in flight at the shutdown
        vmap.release()
        c.L.mh_shutdown(c.h)   # the raw call: the Python binding would defer the shutdown until the factor is gone
        c.h = None
        with pytest.raises(Exception, match="context was shut down"):
            f.reset()
        with pytest.raises(Exception, match="context was shut down"):
            f.linearize(w["R"], w["t"])
        st = f.stats()
        assert st["linearize_count"] >= 1
        f.destroy()
        comm.destroy()


def test_native_world1_over_rccl_full_protocol():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shard_native_rccl_worker.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "OK" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


def _EXTRA(base):
    return [base + i for i in range(int(os.environ.get("MH_FUZZ_EXTRA", "0")))]


@pytest.mark.parametrize("seed", list(range(6)) + _EXTRA(100))
def test_native_sharded_random_configurations(seed):
    """Random room, map density, registration options (Huber, 4-DoF, degeneracy projection, gates), world size, block size,
    uneven (also empty) scan shares, unary / binary, and a pose walk from millimetres to decimetres with the component pass
    toggled: every rank's global result and the state of the points it holds equal the unsharded oracle."""
    import shard_native_common as C
    from mimosa_amd import synth

    rng = np.random.default_rng(99000 + seed)
    room = np.array([rng.uniform(8, 30), rng.uniform(6, 20), rng.uniform(2.5, 4)])
    grid = float(rng.choice([0.11, 0.16, 0.3]))
    map_xyz = synth.make_room(6000 + seed, 0, 0, grid=grid, room=room)
    loc = np.array([rng.uniform(1.5, room[0] - 1.5), rng.uniform(1.5, room[1] - 1.5), rng.uniform(0.8, room[2] - 0.8)])
    scan, aux = synth.make_scan(n_rows=int(rng.choice([16, 32])), seed=7000 + seed, n_cols=int(rng.choice([64, 128])), room=room, sensor_local=loc)
    R, t = synth.query_pose(aux["R_W_L"], aux["t_W_L"])
    binary = bool(rng.integers(0, 2))
    cfg = dict(synth.enwide_config(), use_huber=int(rng.integers(0, 2)), reg_4_dof=int(rng.integers(0, 2)) if not binary else 0,
               project_on_degneneracy=int(rng.integers(0, 2)) if not binary else 0, degen_thresh_trans=float(rng.choice([15.0, 40.0, 1e9])),
               max_corres_distance=float(rng.choice([0.5, 1.0])), plane_validity_distance=float(rng.choice([0.04, 0.07, 0.2])))
    poses = [(R, t)]
    for _ in range(4):
        scale = float(rng.choice([0.002, 0.03, 0.15, 0.6]))
        Rk, tk = poses[-1]
        poses.append((Rk @ synth.so3_exp(rng.normal(0, 1.0, 3) * scale / 5.0), tk + rng.normal(0, 1.0, 3) * scale))
    tgt = None
    if binary:
        Rt, tt = synth.so3_exp(rng.normal(0, 0.02, 3)), rng.normal(0, 0.2, 3)
        tgt = (Rt, tt)
        poses = [(Rt @ Rk, Rt @ tk + tt) for Rk, tk in poses]
    case = dict(map_chunks=np.array_split(map_xyz, int(rng.integers(1, 5))), scan=scan, cfg=cfg, poses=poses, tgt=tgt, binary=binary)
    world = int(rng.choice([2, 3, 4, 5]))
    C.run_local_world(world, case=case, block_log2=int(rng.choice([2, 3, 4])), uneven=True,
                      components_off_from=int(rng.integers(1, 5)) if rng.integers(0, 2) else None, check_eigvec=False)
