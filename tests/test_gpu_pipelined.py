"""-m gpu: pipelined calls (mh_icp_linearize_async: up to 64 calls of a factor in flight, K3 and K4 of every call on the
context's stream) against the same calls made synchronously, and against the oracle.

K4 of call i reads the record and the partial rows K3 of call i wrote; K3 of call i + 1 of the SAME factor overwrites both and
rewrites the association state in place.  The pose changes with every call, so a K4 that read another call's record or rows
would report another call's components.  (Rounds 4-5 ran the K4 work of pipelined calls on a side stream; round 6 removed that
"component server" — these tests keep the pipelined form honest.)"""
import numpy as np
import pytest

from parity import assert_result_parity

pytestmark = pytest.mark.gpu

KEYS = ("H_ss", "b_s", "f", "status_hist", "loc_trans_comp", "loc_rot_comp", "loc_trans_final", "loc_rot_final", "eigvec_rot",
        "eigvec_trans", "n_knn", "mean_candidates", "linearize_count")


def _poses(world, n):
    from mimosa_amd import synth
    out = []
    for j in range(n):
        dR = synth.so3_exp(np.array([0.0004 * j, -0.0003 * j, 0.0011 * j]))
        out.append((world["R"] @ dR, world["t"] + np.array([0.013 * j, -0.007 * j, 0.002 * (j % 3)])))
    return out


def _run(ctx, gm, world, poses, pipelined, reset_each, sync_every=0):
    from mimosa_amd import capi
    f = capi.ICPFactor(ctx, gm, world["pts"], capi.make_reg_config(**world["cfg"]))
    outs = []
    for j, (R, t) in enumerate(poses):
        if reset_each:
            f.reset()
        if not pipelined or (sync_every and j % sync_every == sync_every - 1):
            f.wait()
            outs.append(f.linearize(R, t))       # a synchronous call (between pipelined ones)
        else:
            outs.append(f.linearize_async(R, t))
    f.wait()
    res = [o if isinstance(o, dict) else o.as_dict() for o in outs]
    st = f.state()
    f.destroy()
    return res, st


@pytest.mark.parametrize("reset_each", [False, True])
def test_pipelined_calls_equal_synchronous_calls_and_the_oracle(ctx, room_world, reset_each):
    from mimosa_amd import capi
    from oracle import ref_cpu

    gm = capi.VoxelMap(ctx)
    gm.insert(room_world["map_xyz"])
    poses = _poses(room_world, 24)                     # 24 calls in flight: the record and the rows are reused 24 times
    on, st_on = _run(ctx, gm, room_world, poses, True, reset_each)
    off, st_off = _run(ctx, gm, room_world, poses, False, reset_each)
    for a, b in zip(on, off):
        for k in KEYS:
            assert np.array_equal(np.asarray(a[k], float), np.asarray(b[k], float), equal_nan=True), k
    for x, y in zip(st_on, st_off):
        assert np.array_equal(x, y, equal_nan=True)
    # the calls differ from each other (else the comparison above could not see a mixed-up record)
    assert not np.array_equal(on[3]["loc_rot_comp"], on[4]["loc_rot_comp"]) or not np.array_equal(on[3]["status_hist"], on[4]["status_hist"])
    # and the oracle agrees, call by call
    rm = ref_cpu.Map()
    rm.insert(room_world["map_xyz"])
    fr = ref_cpu.ICP(rm, room_world["pts"], ref_cpu.make_config(**room_world["cfg"]))
    for j, (R, t) in enumerate(poses[:6]):
        if reset_each:
            fr = ref_cpu.ICP(rm, room_world["pts"], ref_cpu.make_config(**room_world["cfg"]))
        want = fr.linearize(R, t)
        want["linearize_count"] = on[j]["linearize_count"]  # (a reset keeps the factor's call counter; the fresh oracle factor starts at 1)
        assert_result_parity(on[j], want)
    gm.release()


def test_synchronous_calls_between_pipelined_ones(ctx, small_world):
    """A blocking mh_icp_linearize every fourth call of a pipeline."""
    from mimosa_amd import capi

    gm = capi.VoxelMap(ctx)
    gm.insert(small_world["map_xyz"])
    poses = _poses(small_world, 16)
    on, st_on = _run(ctx, gm, small_world, poses, True, False, sync_every=4)
    off, st_off = _run(ctx, gm, small_world, poses, False, False, sync_every=4)
    for a, b in zip(on, off):
        for k in KEYS:
            assert np.array_equal(np.asarray(a[k], float), np.asarray(b[k], float), equal_nan=True), k
    for x, y in zip(st_on, st_off):
        assert np.array_equal(x, y, equal_nan=True)
    gm.release()


def test_factor_destroyed_with_calls_just_collected_and_components_toggled(ctx, small_world):
    """Records go back to the allocation cache behind the stream's last K4 (MH_ALLOC_CHECK verifies the hand-over); switching
    the component pass off between pipelined calls makes K3's last block fold the rows the previous call's K4 read."""
    from mimosa_amd import capi

    gm = capi.VoxelMap(ctx)
    gm.insert(small_world["map_xyz"])
    cfg = capi.make_reg_config(**small_world["cfg"])
    poses = _poses(small_world, 6)
    for rep in range(4):
        f = capi.ICPFactor(ctx, gm, small_world["pts"], cfg)
        g = capi.ICPFactor(ctx, gm, small_world["pts"], cfg)
        outs = [f.linearize_async(R, t) for R, t in poses[:3]]
        f.wait()                                   # (the switch is refused with calls in flight)
        f.set_components(False)
        outs += [f.linearize_async(R, t) for R, t in poses[3:]]
        f.wait()
        want = [g.linearize(R, t) for R, t in poses[:3]]
        g.set_components(False)
        want += [g.linearize(R, t) for R, t in poses[3:]]
        for a, b in zip(outs, want):
            a = a.as_dict()
            for k in ("H_ss", "b_s", "f", "loc_trans_final", "loc_rot_final"):
                assert np.array_equal(a[k], b[k]), k
            assert np.array_equal(a["loc_rot_comp"], b["loc_rot_comp"], equal_nan=True)
        f.destroy()
        g.destroy()
    gm.release()


def test_full_size_pipelined_calls_equal_synchronous_calls(ctx, big_world):
    """131 072 points (K3's 512-thread class): pipelined calls agree with synchronous calls to the bit."""
    from mimosa_amd import capi

    gm = capi.VoxelMap(ctx)
    for xyz in big_world["map_rooms"]:
        gm.insert(xyz)
    cfg = capi.make_reg_config(**big_world["cfg"])
    poses = _poses(big_world, 8)
    f, g = capi.ICPFactor(ctx, gm, big_world["pts"], cfg), capi.ICPFactor(ctx, gm, big_world["pts"], cfg)
    outs = []
    for R, t in poses:
        f.reset()
        outs.append(f.linearize_async(R, t))
    f.wait()
    for o, (R, t) in zip(outs, poses):
        g.reset()
        want, got = g.linearize(R, t), o.as_dict()
        for k in KEYS:
            if k != "linearize_count":
                assert np.array_equal(np.asarray(got[k], float), np.asarray(want[k], float), equal_nan=True), k
    for x, y in zip(f.state(), g.state()):
        assert np.array_equal(x, y, equal_nan=True)
    f.destroy()
    g.destroy()
    gm.release()


@pytest.mark.parametrize("pause_s", [0.3])
def test_pipeline_left_alone_then_resumed(ctx, small_world, pause_s):
    """Pipelined calls, nothing for a while (the stream drains, the first calls' words have long landed), more pipelined calls,
    all collected by ONE wait."""
    import time

    from mimosa_amd import capi

    gm = capi.VoxelMap(ctx)
    gm.insert(small_world["map_xyz"])
    cfg = capi.make_reg_config(**small_world["cfg"])
    poses = _poses(small_world, 12)
    f, g = capi.ICPFactor(ctx, gm, small_world["pts"], cfg), capi.ICPFactor(ctx, gm, small_world["pts"], cfg)
    outs = [f.linearize_async(R, t) for R, t in poses[:6]]
    time.sleep(pause_s)
    outs += [f.linearize_async(R, t) for R, t in poses[6:]]
    f.wait()
    want = [g.linearize(R, t) for R, t in poses]
    for a, b in zip(outs, want):
        a = a.as_dict()
        for k in KEYS:
            assert np.array_equal(np.asarray(a[k], float), np.asarray(b[k], float), equal_nan=True), k
    f.destroy()
    g.destroy()
    gm.release()


def test_two_factors_alternating_half_bursts_never_drain(ctx, small_world):
    """Half-bursts alternate between two factors of one context and each wait collects ONE factor's calls while the other's are
    in flight.  Results equal synchronous calls to the bit."""
    from mimosa_amd import capi

    gm = capi.VoxelMap(ctx)
    gm.insert(small_world["map_xyz"])
    cfg = capi.make_reg_config(**small_world["cfg"])
    poses = _poses(small_world, 40)
    pair = (capi.ICPFactor(ctx, gm, small_world["pts"], cfg), capi.ICPFactor(ctx, gm, small_world["pts"], cfg))
    g = capi.ICPFactor(ctx, gm, small_world["pts"], cfg)
    outs, cur = [], 0
    for k in range(0, len(poses), 5):
        f = pair[cur]
        for R, t in poses[k:k + 5]:
            f.reset()
            outs.append(f.linearize_async(R, t))
        cur ^= 1
        pair[cur].wait()
    pair[cur ^ 1].wait()
    pair[cur].wait()   # nothing pending: a no-op
    for o, (R, t) in zip(outs, poses):
        g.reset()
        want, got = g.linearize(R, t), o.as_dict()
        for k in KEYS:
            if k != "linearize_count":
                assert np.array_equal(np.asarray(got[k], float), np.asarray(want[k], float), equal_nan=True), k
    # a synchronous call and a destroy of one factor while the other has calls open
    o2 = [pair[0].linearize_async(R, t) for R, t in poses[:4]]
    want = g.linearize(*poses[7])
    pair[1].destroy()
    pair[0].wait()
    assert np.isfinite(want["f"]) and all(np.isfinite(o.as_dict()["f"]) for o in o2)  # (warm calls of another history: they must complete)
    pair[0].destroy()
    g.destroy()
    gm.release()


def test_sixty_four_calls_in_flight_and_the_sixty_fifth_refused(ctx, small_world):
    """The library holds up to 64 calls of a factor (kMaxPending): all 64 agree with synchronous calls; one more is refused
    (and leaves the 64 collectable)."""
    from mimosa_amd import capi

    gm = capi.VoxelMap(ctx)
    gm.insert(small_world["map_xyz"])
    cfg = capi.make_reg_config(**small_world["cfg"])
    poses = _poses(small_world, 64)
    f, g = capi.ICPFactor(ctx, gm, small_world["pts"], cfg), capi.ICPFactor(ctx, gm, small_world["pts"], cfg)
    outs = []
    for R, t in poses:
        f.reset()
        outs.append(f.linearize_async(R, t))
    with pytest.raises(Exception, match="too many calls in flight"):
        f.linearize_async(*poses[0])
    f.wait()
    for o, (R, t) in zip(outs, poses):
        g.reset()
        want, got = g.linearize(R, t), o.as_dict()
        for k in KEYS:
            if k != "linearize_count":
                assert np.array_equal(np.asarray(got[k], float), np.asarray(want[k], float), equal_nan=True), k
    f.destroy()
    g.destroy()
    gm.release()
