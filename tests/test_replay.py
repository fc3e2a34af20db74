"""Sequence replay (SURVEY.md §8 "next" row f-4, BASELINE configs[4]): front end -> IMU-propagated deskew -> geometric +
photometric factors -> fixed-lag Gauss-Newton window (all live ICP factors re-linearized per iteration) -> keyframe map update.

CPU: the loop on the oracle tracks the ground-truth trajectory (checks Jacobian / retraction / between-factor conventions end to
end, no GPU involved).  GPU: the same loop through the C ABI (batched re-linearization, device-resident front end, map and
photometric frame) produces the same trajectory as the oracle loop, scan by scan."""
import numpy as np
import pytest

from mimosa_amd import replay
from oracle.replay_backend import OracleBackend


def small_cfg(n=6, **kw):
    return replay.ReplayConfig(n_scans=n, rows=64, cols=512, room=(12.0, 10.0, 3.0), keyframe_trans_thresh=0.2,
                               keyframe_rot_thresh_deg=5.0, **kw)


def test_replay_tracks_ground_truth_on_oracle():
    cfg = small_cfg(6)
    scans = replay.make_scans(cfg)
    r = replay.run(cfg, OracleBackend(cfg), scans)
    assert r["n_keyframes"] >= 2
    # from a 3 cm / 0.3 deg first guess; the ~5 mm floor is the small room's corner bias of 5-point plane fits, not noise
    assert max(r["trans_err"]) < 0.012 and max(r["rot_err_deg"]) < 0.06
    assert r["costs"][0][-1] < r["costs"][0][0]                          # the first window really had to move
    assert len(r["photo_valid"]) == 5 and min(r["photo_valid"]) >= 20     # features are tracked from scan to scan
    # without the photometric factor the geometric window alone tracks as well
    cfg2 = small_cfg(4, photometric=False)
    r2 = replay.run(cfg2, OracleBackend(cfg2), scans[:4])
    assert max(r2["trans_err"]) < 0.012


def test_imu_propagation_matches_the_exact_twist():
    """propagate() — Manager::deskewPoints' host part — against the closed-form constant-twist motion."""
    from mimosa_amd import synth

    cfg = small_cfg(2, imu_gyro_noise=0.0, imu_acc_noise=0.0)
    scans = replay.make_scans(cfg)
    s0, s1 = scans
    uns = s1["frame"]["unique_ns"]
    T, (Rp, pp, vp) = replay.propagate(s0["R_gt"], s0["t_gt"], s0["R_gt"] @ np.array(cfg.v), s1["imu"], s1["header_ts"], uns)
    w, v = np.array(cfg.w), np.array(cfg.v)
    assert np.abs(Rp - s0["R_gt"] @ synth.so3_exp(w * cfg.dt)).max() < 1e-9
    # exact: p(dt) = p0 + R0 * int_0^dt Exp(w s) v ds; the scan generator steps first-order, 0.5 |w x v| dt^2 ~ 2 mm apart
    ss = np.linspace(0, cfg.dt, 2001)
    exact = s0["t_gt"] + s0["R_gt"] @ (np.trapezoid(np.stack([synth.so3_exp(w * x) @ v for x in ss]), ss, axis=0))
    assert np.linalg.norm(pp - exact) < 2e-5
    assert np.linalg.norm(pp - s1["t_gt"]) < 3e-3
    assert np.abs(T[-1, :9].reshape(3, 3) - Rp).max() < 1e-12 and np.abs(T[-1, 9:] - pp).max() < 1e-12


@pytest.mark.gpu
def test_replay_hip_equals_oracle(ctx):
    cfg = small_cfg(6)
    scans = replay.make_scans(cfg)
    ro = replay.run(cfg, OracleBackend(cfg), scans)
    rh = replay.run(cfg, replay.HipBackend(ctx, cfg), scans)
    assert rh["n_keyframes"] == ro["n_keyframes"] and rh["photo_valid"] == ro["photo_valid"]
    for (Ra, ta), (Rb, tb) in zip(rh["poses_est"], ro["poses_est"]):
        assert np.max(np.abs(ta - tb)) < 1e-7 and np.max(np.abs(Ra - Rb)) < 1e-8
    assert max(rh["trans_err"]) < 0.012


def test_native_replay_driver_compiles():
    """CPU-runnable: the C++ replay harness (host/mimosa_hip/replay.hpp + replay_main.cpp) builds warning-free."""
    import os
    from mimosa_amd import build
    assert os.path.exists(build.build_replay_native())


@pytest.mark.gpu
def test_native_replay_equals_python_replay(ctx, tmp_path):
    """The same sequence through the C++ host mirror (no Python between the library calls) and through replay.run on the
    C ABI binding: same keyframes, same tracked features, same trajectory (the two differ only in the order of the host-side
    floating-point operations of the 30 x 30 solve)."""
    cfg = small_cfg(7)
    scans = replay.make_scans(cfg)
    rp = replay.run(cfg, replay.HipBackend(ctx, cfg), scans)
    rn = replay.run_native(cfg, scans, str(tmp_path))
    assert rn["n_keyframes"] == rp["n_keyframes"] and rn["photo_valid"] == rp["photo_valid"]
    assert np.allclose(rn["first_costs"], rp["costs"][0], rtol=1e-9)
    for (Ra, ta), (Rb, tb) in zip(rn["poses_est"], rp["poses_est"]):
        assert np.max(np.abs(ta - tb)) < 1e-7 and np.max(np.abs(Ra - Rb)) < 1e-8


@pytest.mark.gpu
def test_replay_full_size_hip_equals_oracle(ctx, tmp_path):
    """BASELINE configs[4] at the bench's size inside the suite: 20 scans of 128 x 1024, window 5, 6 update iterations,
    photometric on — the HIP loop (Python harness over the C ABI), the native pipelined loop and the oracle loop produce the
    same keyframes, the same tracked features and the same trajectory."""
    cfg = replay.ReplayConfig(n_scans=20, rows=128)
    scans = replay.make_scans(cfg)
    ro = replay.run(cfg, OracleBackend(cfg), scans)
    rh = replay.run(cfg, replay.HipBackend(ctx, cfg), scans)
    rn = replay.run_native(cfg, scans, str(tmp_path))
    assert len(ro["poses_est"]) == 20 and ro["n_keyframes"] >= 2
    assert rh["n_keyframes"] == ro["n_keyframes"] == rn["n_keyframes"] and rh["photo_valid"] == ro["photo_valid"] == rn["photo_valid"]
    for (Ra, ta), (Rb, tb), (Rc, tc) in zip(rh["poses_est"], ro["poses_est"], rn["poses_est"]):
        assert np.max(np.abs(ta - tb)) < 1e-7 and np.max(np.abs(Ra - Rb)) < 1e-8
        assert np.max(np.abs(tc - tb)) < 1e-6 and np.max(np.abs(Rc - Rb)) < 1e-7
    assert max(rh["trans_err"]) < 0.004 and max(rh["rot_err_deg"]) < 0.02


@pytest.mark.gpu
def test_pipelined_native_replay_is_the_sequential_one_bit_for_bit(tmp_path):
    """replay_native overlaps across scans by default (next cloud staged on a copy stream by one worker thread, the photometric
    map update of scan k on another beside the geometric path of scan k + 1); `sequential` runs the calls one after the other.
    Same calls on the same handles in the same order per handle: identical trajectories, keyframes, tracked features — to the bit."""
    cfg = small_cfg(8)
    scans = replay.make_scans(cfg)
    rp = replay.run_native(cfg, scans, str(tmp_path), repeats=2)
    rs = replay.run_native(cfg, scans, str(tmp_path), repeats=2, sequential=True)
    assert rp["n_keyframes"] == rs["n_keyframes"] and rp["photo_valid"] == rs["photo_valid"]
    assert rp["first_costs"] == rs["first_costs"]
    for (Ra, ta), (Rb, tb) in zip(rp["poses_est"], rs["poses_est"]):
        assert np.array_equal(Ra, Rb) and np.array_equal(ta, tb)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_sharded_replay_equals_the_unsharded_one(tmp_path, world):
    """BASELINE configs[4] with the map SHARDED (host/mimosa_hip/sharded_replay.hpp): a 20-scan sequence, every rank (a host
    thread with its own context, in-process transport) runs the replay loop on its shard of the map — mh_map_insert_shard of the
    seed and of every keyframe's cloud, a ShardedICPFactor per scan, every smoother iteration ONE protocol round over the live
    window (mh_shard_icp_linearize_batch).  Every rank's trajectory is rank 0's, and rank 0's is the unsharded replay's up to
    the order in which the shards' rows are summed."""
    import dataclasses
    cfg = dataclasses.replace(small_cfg(20), photometric=False)
    scans = replay.make_scans(cfg)
    plain = replay.run_native(cfg, scans, str(tmp_path))
    shard = replay.run_native(cfg, scans, str(tmp_path), sharded_world=world)
    assert shard["n_ranks"] == world and shard["max_rank_deviation_m"] <= 1e-9
    assert shard["n_keyframes"] == plain["n_keyframes"] >= 3
    for (Ra, ta), (Rb, tb) in zip(shard["poses_est"], plain["poses_est"]):
        assert np.max(np.abs(ta - tb)) <= 1e-9 and np.max(np.abs(Ra - Rb)) <= 1e-9


@pytest.mark.gpu
def test_sharded_replay_with_the_photometric_path_replicated(tmp_path):
    """The same with the photometric factor on: every rank builds its own frame and features (the scan is replicated, only the
    map is sharded); 2 ranks, 6 scans."""
    cfg = small_cfg(6)
    scans = replay.make_scans(cfg)
    plain = replay.run_native(cfg, scans, str(tmp_path))
    shard = replay.run_native(cfg, scans, str(tmp_path), sharded_world=2)
    assert shard["max_rank_deviation_m"] <= 1e-9 and shard["photo_valid"] == plain["photo_valid"]
    for (Ra, ta), (Rb, tb) in zip(shard["poses_est"], plain["poses_est"]):
        assert np.max(np.abs(ta - tb)) <= 1e-8 and np.max(np.abs(Ra - Rb)) <= 1e-8


@pytest.mark.gpu
def test_native_replay_through_the_manager_mirror(ctx, tmp_path):
    """The sequence through lidar::Manager::callback (host/mimosa_hip/manager.hpp: callback -> prepareInput -> declare ->
    deskewPoints -> preprocess -> getFactors -> define -> postDefineUpdate, the reference's call order, with stand-ins behind the
    graph / IMU manager interfaces).  Reference semantics for the FIRST cloud (manager.cpp:111-121, :399-408): it initialises at
    the pose it is given, is not registered and is NOT deskewed.  Geometric path alone: the trajectory follows the ground truth
    to millimetres.  With the photometric factor the features detected on that first, undeskewed cloud (the platform moves
    12 cm / 2 deg during the sweep here; the reference expects a robot at rest) bias the next poses by ~0.2 deg, and the bias
    decays as those features are replaced."""
    import dataclasses
    cfg = small_cfg(7, prior_trans_noise=0.0, prior_rot_noise_deg=0.0)   # the initial pose is the graph's: make it the true one
    scans = replay.make_scans(cfg)

    def errors(r):
        te = [float(np.linalg.norm(t - sc["t_gt"])) for (R, t), sc in zip(r["poses_est"], scans)]
        re = [float(np.degrees(np.arccos(np.clip((np.trace(R.T @ sc["R_gt"]) - 1) / 2, -1, 1)))) for (R, t), sc in zip(r["poses_est"], scans)]
        return te, re

    geo = dataclasses.replace(cfg, photometric=False)
    rm = replay.run_native(geo, scans, str(tmp_path), through_manager=True)
    rf = replay.run_native(geo, scans, str(tmp_path))
    te, re = errors(rm)
    assert len(rm["poses_est"]) == 7 and rm["n_keyframes"] >= 2
    assert te[0] < 0.02 and re[0] < 0.5, (te, re)                        # scan 0: the given state through one sweep of noisy IMU samples
    assert max(te[1:]) < 0.006 and max(re[1:]) < 0.08, (te, re)
    for (Ra, ta), (Rb, tb) in zip(rm["poses_est"][1:], rf["poses_est"][1:]):
        assert np.max(np.abs(ta - tb)) < 8e-3                            # FixedLagReplay registers (and deskews) the first cloud too
    rp = replay.run_native(cfg, scans, str(tmp_path), through_manager=True)
    te, re = errors(rp)
    assert len(rp["photo_valid"]) >= 5 and min(rp["photo_valid"]) >= 20  # features tracked from scan to scan
    assert max(te[1:]) < 0.012 and max(re[1:]) < 0.3 and re[-1] < re[1], (te, re)


def _EXTRA(base):
    import os
    return [base + i for i in range(int(os.environ.get("MH_FUZZ_EXTRA", "0")))]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", list(range(6)) + _EXTRA(10))
def test_replay_hip_equals_oracle_random_sequences(ctx, seed):
    """Random platform twists, rooms, image sizes, window lengths and keyframe thresholds: the HIP loop and the oracle loop
    make the same keyframe decisions, track the same features and produce the same trajectory."""
    rng = np.random.default_rng(515 + seed)
    rows, cols = int(rng.choice([32, 64])), int(rng.choice([256, 512]))
    cfg = replay.ReplayConfig(n_scans=5, rows=rows, cols=cols, room=(float(rng.uniform(9, 16)), float(rng.uniform(8, 12)), float(rng.uniform(2.8, 3.5))),
                              v=(float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-0.5, 0.5)), 0.0), w=(0.0, 0.0, float(rng.uniform(-0.5, 0.5))),
                              window=int(rng.choice([2, 3, 5])), update_iters=int(rng.choice([2, 6])),
                              keyframe_trans_thresh=float(rng.choice([0.1, 0.2, 1.0])), keyframe_rot_thresh_deg=float(rng.choice([2.0, 5.0, 20.0])),
                              photometric=bool(rng.integers(0, 2)))
    scans = replay.make_scans(cfg)
    ro = replay.run(cfg, OracleBackend(cfg), scans)
    rh = replay.run(cfg, replay.HipBackend(ctx, cfg), scans)
    assert rh["n_keyframes"] == ro["n_keyframes"] and rh["photo_valid"] == ro["photo_valid"]
    for (Ra, ta), (Rb, tb) in zip(rh["poses_est"], ro["poses_est"]):
        assert np.max(np.abs(ta - tb)) < 1e-7 and np.max(np.abs(Ra - Rb)) < 1e-8


@pytest.mark.gpu
def test_concurrent_replays_on_separate_contexts(ctx):
    """Three host threads, each running a whole replay (front end, photometric path, map generations, window batches) on its
    own context at the same time: every one reproduces the single-threaded trajectory bit for bit."""
    import threading
    from mimosa_amd import capi

    cfg = small_cfg(5)
    scans = replay.make_scans(cfg)
    ref = replay.run(cfg, replay.HipBackend(ctx, cfg), scans)
    out, errors = [None] * 3, []

    def worker(i):
        try:
            out[i] = replay.run(cfg, replay.HipBackend(capi.Context(0), cfg), scans)
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(3)]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=600)
    assert not any(x.is_alive() for x in th) and not errors, errors[:2]
    for r in out:
        assert r["n_keyframes"] == ref["n_keyframes"] and r["photo_valid"] == ref["photo_valid"]
        for (Ra, ta), (Rb, tb) in zip(r["poses_est"], ref["poses_est"]):
            assert np.array_equal(ta, tb) and np.array_equal(Ra, Rb)
