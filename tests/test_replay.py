"""Sequence replay (SURVEY.md §8 "next" row f-4): front end -> factor -> Gauss-Newton -> keyframe map update.

CPU: the loop on the oracle converges to the ground-truth trajectory (checks the Jacobian / retraction
convention end to end, with no GPU involved).  GPU: the same loop through the C ABI produces the same
trajectory as the oracle loop, scan by scan."""
import numpy as np
import pytest

from mimosa_amd import replay, synth
from oracle.replay_backend import OracleBackend


def small_cfg(n=5):
    return replay.ReplayConfig(n_scans=n, rows=32, cols=256, v=(0.6, 0.2, 0.0), w=(0.0, 0.0, 0.3), start_local=(4.0, 4.0, 1.2),
                               room=(12.0, 10.0, 3.0), keyframe_trans_thresh=0.1, keyframe_rot_thresh_deg=5.0)


def test_replay_converges_on_oracle():
    cfg = small_cfg(5)
    scans = replay.make_scans(cfg)
    r = replay.run(cfg, OracleBackend(cfg.reg), scans)
    assert r["n_keyframes"] >= 2
    assert max(r["trans_err"]) < 0.01 and max(r["rot_err_deg"]) < 0.05   # from 3 cm / 0.3 deg priors
    for fs in r["costs"]:
        assert fs[-1] < fs[0]                                             # every optimisation lowered the cost


@pytest.mark.gpu
def test_replay_hip_equals_oracle(ctx):
    cfg = small_cfg(5)
    scans = replay.make_scans(cfg)
    ro = replay.run(cfg, OracleBackend(cfg.reg), scans)
    rh = replay.run(cfg, replay.HipBackend(ctx, cfg.reg), scans)
    assert rh["n_keyframes"] == ro["n_keyframes"]
    for (Ra, ta), (Rb, tb) in zip(rh["poses_est"], ro["poses_est"]):
        assert np.max(np.abs(ta - tb)) < 1e-8 and np.max(np.abs(Ra - Rb)) < 1e-9
    assert max(rh["trans_err"]) < 0.01
