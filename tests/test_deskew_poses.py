"""Manager::deskewPoints' pose part (src/lidar/manager.cpp:455-499) in the C++ host mirror vs the numpy
restatement: constant-acceleration / constant-rate extrapolation inside each IMU interval, then
T_Le_W * T_W_Bt * T_B_S.  CPU only (host-side fp64 code in the reference too)."""
import json
import os
import struct
import subprocess

import numpy as np

from mimosa_amd import synth
from oracle import numpy_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _exe():
    from mimosa_amd import build
    lib = build.build()
    exe = os.path.join(os.path.dirname(lib), "deskew_poses")
    src = os.path.join(ROOT, "tests", "cpp", "deskew_poses.cpp")
    hdr = os.path.join(ROOT, "mimosa_amd", "host", "mimosa_hip", "lidar.hpp")
    if not os.path.exists(exe) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(exe):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I", ROOT, "-I", os.path.join(ROOT, "mimosa_amd", "host", "gtsam_sig"), src, "-o", exe,
                               "-L", os.path.dirname(lib), "-lmimosa_hip", "-Wl,-rpath,$ORIGIN"])
    return exe


def test_deskew_poses_match_numpy(tmp_path):
    rng = np.random.default_rng(11)
    n_imu, header_ts = 12, 1000.25
    imu_t = header_ts - 0.004 + np.arange(n_imu) * 0.01            # 100 Hz, first sample before the scan starts
    acc = rng.normal(0, 0.5, (n_imu, 3)) + np.array([0, 0, 9.81])
    gyro = rng.normal(0, 0.2, (n_imu, 3))
    bias_a, bias_g = rng.normal(0, 0.02, 3), rng.normal(0, 0.005, 3)
    g_unit, g_norm = np.array([0.0, 0.0, -1.0]), 9.81
    # NavStates at the IMU times: any smooth trajectory (they come from the caller's preintegrator)
    nav_R, nav_p, nav_v = [], [], []
    R, p, v = synth.so3_exp(np.array([0.02, -0.01, 0.4])), np.array([3.0, -2.0, 1.0]), np.array([1.5, 0.2, -0.1])
    for j in range(n_imu):
        nav_R.append(R.copy()); nav_p.append(p.copy()); nav_v.append(v.copy())
        R = R @ synth.so3_exp((gyro[j] - bias_g) * 0.01)
        p = p + v * 0.01
        v = v + (R @ (acc[j] - bias_a) + g_unit * g_norm) * 0.01
    unique_ns = (np.arange(0, 1024, 7) * 97_656).astype(np.uint32)   # 0 .. ~0.0999 s
    T_B_S = (synth.so3_exp(np.array([0.01, 0.02, -0.03])), np.array([-0.006253, 0.011775, 0.0028525]))
    want = numpy_ref.deskew_poses(imu_t, acc, gyro, nav_R, nav_p, nav_v, bias_a, bias_g, g_unit, g_norm, unique_ns,
                                  header_ts, T_B_S)
    assert len(want) == len(unique_ns)

    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        def w(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            f.write(struct.pack("<Q", a.size))
            f.write(a.tobytes())
        w(imu_t, np.float64)
        w(np.concatenate([acc, gyro], 1), np.float64)
        w(np.concatenate([np.reshape(nav_R, (n_imu, 9)), np.array(nav_p), np.array(nav_v)], 1), np.float64)
        w(np.concatenate([bias_a, bias_g, g_unit, [g_norm, header_ts], T_B_S[0].ravel(), T_B_S[1]]), np.float64)
        w(unique_ns, np.uint32)
    out = subprocess.run([_exe(), str(inp)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    got = np.array(json.loads(out.stdout))
    assert got.shape == (len(want), 12)
    for g, T in zip(got, want):
        assert np.max(np.abs(g[:9].reshape(3, 3) - T[:3, :3])) < 1e-12
        assert np.max(np.abs(g[9:] - T[:3, 3])) < 1e-11
    # the pose of the last timestamp is close to identity only if it is the scan end; sanity: all rigid
    for g in got:
        Rm = g[:9].reshape(3, 3)
        assert np.max(np.abs(Rm @ Rm.T - np.eye(3))) < 1e-12


def test_too_few_imu_samples_is_an_error(tmp_path):
    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        def w(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            f.write(struct.pack("<Q", a.size))
            f.write(a.tobytes())
        w([1.0], np.float64)
        w(np.zeros(6), np.float64)
        w(np.concatenate([np.eye(3).ravel(), np.zeros(6)]), np.float64)
        w(np.concatenate([np.zeros(6), [0, 0, -1, 9.81, 1.0], np.eye(3).ravel(), np.zeros(3)]), np.float64)
        w([0], np.uint32)
    out = subprocess.run([_exe(), str(inp)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 1 and "less than 2 measurements" in out.stderr
