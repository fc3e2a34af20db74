"""-m gpu: the C++ host mirror (ICPFactor / Geometric / IncrementalVoxelMapPCL / deskewPoints over the
C ABI) driven through the reference's call order, compared with the oracle doing the same steps."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from parity import rel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_exe(name="host_pipeline"):
    from mimosa_amd import build
    lib = build.build()
    exe = os.path.join(os.path.dirname(lib), name)
    src = os.path.join(ROOT, "tests", "cpp", name + ".cpp")
    hdrs = [os.path.join(ROOT, "mimosa_amd", "host", "mimosa_hip", h) for h in ("lidar.hpp", "photometric.hpp", "types.hpp", "sharded.hpp", "binio.hpp", "manager.hpp")]
    sig = os.path.join(ROOT, "mimosa_amd", "host", "gtsam_sig")
    hdrs += [os.path.join(dp, f) for dp, _, fs in os.walk(sig) for f in fs]
    if not os.path.exists(exe) or max([os.path.getmtime(src), os.path.getmtime(lib)] + [os.path.getmtime(h) for h in hdrs]) > os.path.getmtime(exe):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-I", ROOT, "-I", os.path.join(ROOT, "mimosa_amd", "host", "gtsam_sig"), src, "-o", exe,
                               "-L", os.path.dirname(lib), "-lmimosa_hip", "-lpthread", "-Wl,-rpath,$ORIGIN"])
    return exe


def test_host_layer_compiles():
    """CPU-runnable: the header-only host mirror builds warning-free against the C ABI."""
    assert os.path.exists(build_exe())
    assert os.path.exists(build_exe("photo_pipeline"))
    assert os.path.exists(build_exe("point_types"))
    assert os.path.exists(build_exe("sharded_pipeline"))
    assert os.path.exists(build_exe("manager_types"))    # lidar::Manager::callback<PointT> for all nine sensor point types


def test_host_mirror_is_written_against_gtsam_headers():
    """The mirror includes <gtsam/...> and uses GTSAM's own spellings; what those headers are is an include-path matter
    (host/gtsam_sig here, a GTSAM installation in a deployment): no #ifdef, no second code path."""
    host = os.path.join(ROOT, "mimosa_amd", "host", "mimosa_hip")
    src = "".join(open(os.path.join(host, h)).read() for h in os.listdir(host))
    assert "#include <gtsam/nonlinear/NonlinearFactor.h>" in src and "#include <gtsam/linear/HessianFactor.h>" in src
    assert "c.at<Pose3>(keys()[0])" in src and "c.at<Unit3>(G(0)).unitVector()" in src      # geometric_factor.hpp:247-257
    assert "std::make_shared<HessianFactor>(keys[0], matrix6(r.H_ss), g1, r.f)" in src          # :559-560
    assert "MIMOSA_HIP_WITH_GTSAM" not in src and "atPose3" not in src and "gravityUnit" not in src


def _pose12(R, t):
    return np.concatenate([np.asarray(R, float).ravel(), np.asarray(t, float)])


@pytest.mark.gpu
def test_host_pipeline_matches_oracle(tmp_path):
    from mimosa_amd import synth
    from oracle import ref_cpu

    room = np.array([6.0, 5.0, 3.0])
    map_xyz = synth.make_room(1234, 0, 0, room=room)
    scan, aux = synth.make_scan(n_rows=32, seed=77, skew=True, n_cols=256, room=room,
                                sensor_local=np.array([2.3, 2.6, 1.2]))
    T_B_L = (synth.so3_exp(np.array([0.01, -0.02, 0.03])), np.array([0.05, 0.02, -0.1]))
    # body pose: T_W_B = T_W_L * T_B_L^-1, perturbed like the bench query pose
    R_WL, t_WL = synth.query_pose(aux["R_W_L"], aux["t_W_L"])
    R_WB = R_WL @ T_B_L[0].T
    t_WB = t_WL - R_WB @ T_B_L[1]
    R_WB2, t_WB2 = R_WB @ synth.so3_exp(np.array([0.0, 0.0, 0.004])), t_WB + np.array([0.05, 0.02, 0.0])
    poses = np.concatenate([aux["Rt12"][g].astype(np.float64) for g in range(len(aux["unique_ns"]))])

    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        def w(arr):
            arr = np.ascontiguousarray(arr)
            f.write(struct.pack("<Q", arr.size if arr.dtype.itemsize != 32 else len(arr)))
            f.write(arr.tobytes())
        w(map_xyz.astype(np.float32).ravel())
        w(scan)
        w(aux["unique_ns"].astype(np.uint32))
        w(poses)
        w(np.concatenate([_pose12(*T_B_L), _pose12(R_WB, t_WB), _pose12(R_WB2, t_WB2)]))
        # the raw Ouster cloud of the same scan, for the device-resident front end (ScanFrontEnd)
        raw, _ = synth.make_raw_scan(32, seed=77, n_cols=256, room=room, sensor_local=np.array([2.3, 2.6, 1.2]),
                                     dropouts=False)
        assert np.array_equal(raw["x"], scan["x"]) and np.array_equal(raw["t"], scan["t"])
        w(raw)
    out = subprocess.run([build_exe(), str(inp)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout)

    # ---- the same steps through the oracle --------------------------------------------------------
    cfg = synth.enwide_config()
    desk = ref_cpu.deskew(scan, aux["unique_ns"], aux["Rt12"])
    sub = desk[desk["idx"] % 4 == 0]
    Be = ref_cpu.transform_f32(sub, T_B_L[0].astype(np.float32), T_B_L[1].astype(np.float32))
    ds = Be[ref_cpu.downsample(Be, 0.5, 20, 0.15)]
    rmap = ref_cpu.Map()
    rmap.insert(map_xyz)
    fac = ref_cpu.ICP(rmap, ds, ref_cpu.make_config(**cfg))
    fac.linearize(R_WB, t_WB)
    r1 = fac.linearize(R_WB, t_WB)  # the driver relinearizes once more before dumping
    for key, ref in (("first", r1),):
        g = got[key]
        assert g["n_ds"] == len(ds)
        assert g["status_hist"] == [int(v) for v in ref["status_hist"]]
        assert rel(np.array(g["H"]).reshape(6, 6), ref["H_ss"]) <= 1e-5
        assert rel(np.array(g["g"]), -ref["b_s"]) <= 1e-5  # HessianFactor gets -J^T e
        assert abs(g["f"] - ref["f"]) <= 1e-5 * ref["f"]
        assert rel(g["loc_trans_comp"], ref["loc_trans_comp"]) <= 1e-5
        want = [float(ref["loc_rot_comp"][i] < cfg["degen_thresh_rot"]) for i in range(3)] + \
               [float(ref["loc_trans_comp"][i] < cfg["degen_thresh_trans"]) for i in range(3)]
        assert g["degen_directions"] == want
    # device-resident front end: same cloud, same order -> the same factor to the last bit
    gd = got["first_device_frontend"]
    assert gd["n_ds"] == got["first"]["n_ds"] and gd["status_hist"] == got["first"]["status_hist"]
    assert gd["H"] == got["first"]["H"] and gd["g"] == got["first"]["g"] and gd["f"] == got["first"]["f"]
    assert abs(got["corrected_ts"] - (100.0 + float(scan["t"].max()) * 1e-9)) < 1e-9
    assert got["batch_equal"] == 1
    assert got["map_points_after_device"] == got["map_points_after"]   # device-side updateMap ≡ host-cloud updateMap
    assert got["linearize_count"] == 2 and abs(got["clone_f"] - r1["f"]) <= 1e-5 * r1["f"]
    assert got["map_updated"] == 1 and got["map_updated_2"] == 0
    W = ref_cpu.transform_f32(Be, R_WB.astype(np.float32), t_WB.astype(np.float32))
    rmap2 = rmap.copy()
    rmap2.insert(np.stack([W["x"], W["y"], W["z"]], 1))
    assert got["map_points_after"] == rmap2.num_points
    fac2 = ref_cpu.ICP(rmap2, ds, ref_cpu.make_config(**cfg))
    fac2.linearize(R_WB2, t_WB2)
    r2 = fac2.linearize(R_WB2, t_WB2)
    g = got["second"]
    assert g["status_hist"] == [int(v) for v in r2["status_hist"]]
    assert rel(np.array(g["H"]).reshape(6, 6), r2["H_ss"]) <= 1e-5 and abs(g["f"] - r2["f"]) <= 1e-5 * r2["f"]


@pytest.mark.gpu
def test_photometric_host_pipeline_matches_oracle(tmp_path):
    """Photometric / PhotometricFactor of the C++ mirror through preprocess -> updateMap -> preprocess -> getFactors ->
    linearize -> clone -> updateMap, against the oracle doing the same steps."""
    from mimosa_amd import synth_photo as sp
    from oracle import photo_ref

    cfg = sp.photo_config(rows=64, cols=512)
    frames = [sp.make_frame(cfg, k) for k in range(2)]
    inp = tmp_path / "photo.bin"
    with open(inp, "wb") as f:
        def w(arr, dtype=None):
            arr = np.ascontiguousarray(arr if dtype is None else np.asarray(arr, dtype))
            f.write(struct.pack("<Q", len(arr) if arr.dtype.itemsize == 32 else arr.size))
            f.write(arr.tobytes())
        w([cfg["rows"], cfg["cols"], cfg["destagger"], cfg["erosion_buffer"], cfg["patch_size"], cfg["margin_size"], cfg["remove_lines"],
           cfg["filter_brightness"], cfg["gaussian_blur"], cfg["gaussian_blur_size"], cfg["nma_radius"], cfg["num_features_detect"],
           cfg["max_feature_life_time"], cfg["rotate_patch_to_align_with_gradient"], cfg["use_robust_cost_function"],
           cfg["robust_cost_function"], cfg["brightness_window_size"][0], cfg["brightness_window_size"][1]], np.int32)
        w([cfg["range_min"], cfg["range_max"], cfg["intensity_scale"], cfg["intensity_gamma"], cfg["gradient_threshold"],
           cfg["max_dist_from_mean"], cfg["max_dist_from_plane"], cfg["occlusion_range_diff_threshold"],
           cfg["lidar_origin_to_beam_origin_mm"], cfg["robust_cost_function_parameter"], cfg["error_scale"], cfg["max_error"],
           cfg["sigma"]], np.float64)
        w(cfg["pixel_shift_by_row"], np.int32)
        w(cfg["beam_altitude_angles"], np.float32)
        w(cfg["high_pass_fir"], np.float64)
        w(cfg["low_pass_fir"], np.float64)
        w(np.asarray(cfg["patch_offsets"], np.int32).ravel())
        w(_pose12(cfg["T_B_L_R"], cfg["T_B_L_t"]))
        w(np.asarray(sp.BIAS_DIRECTIONS, np.float64).ravel())
        for fr in frames:
            w(fr["raw"])
            w(fr["deskewed"])
            w(fr["unique_ns"].astype(np.uint32))
            w(np.asarray(fr["T_Le_Lt"], np.float64).ravel())
            w(_pose12(fr["R_W_Be"], fr["t_W_Be"]))
    out = subprocess.run([build_exe("photo_pipeline"), str(inp)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout)

    ref = photo_ref.Photo(cfg)
    d0 = ref.preprocess(frames[0]["raw"], frames[0]["deskewed"], frames[0]["unique_ns"], frames[0]["T_Le_Lt"])
    assert got["n_factors_0"] == 0 and abs(got["intensity_sum_0"] - float(d0["intensity"].astype(np.float64).sum())) < 1e-6
    ref.update_map(None, frames[0]["R_W_Be"], frames[0]["t_W_Be"], sp.BIAS_DIRECTIONS)
    feats = ref.features()
    assert got["n_features_0"] == len(feats) > 20
    assert abs(got["feature_sum_0"] - sum(ft["center"][0] + 1e-3 * ft["center"][1] + ft["life_time"] for ft in feats)) < 1e-9
    ref.preprocess(frames[1]["raw"], frames[1]["deskewed"], frames[1]["unique_ns"], frames[1]["T_Le_Lt"])
    pf = ref.make_factor()
    r = pf.linearize(frames[1]["R_W_Be"], frames[1]["t_W_Be"])
    assert got["n_factors_1"] == 1 and got["status_hist"] == [int(v) for v in r["status_hist"]]
    assert got["n_valid"] == int(r["status_hist"][8]) >= 10 and got["clone_equal"] == 1
    assert rel(np.array(got["H"]).reshape(6, 6), np.asarray(r["H_bb"]).reshape(6, 6)) <= 1e-5
    assert rel(np.array(got["g"]), -np.asarray(r["b_b"])) <= 1e-5 and abs(got["f"] - r["f"]) <= 1e-5 * abs(r["f"])
    assert rel(got["loc_trans_final"], r["loc_trans_final"]) <= 1e-5
    ref.update_map(pf, frames[1]["R_W_Be"], frames[1]["t_W_Be"], sp.BIAS_DIRECTIONS)
    feats = ref.features()
    assert got["n_features_1"] == len(feats)
    assert abs(got["feature_sum_1"] - sum(ft["center"][0] + 1e-3 * ft["center"][1] + ft["life_time"] for ft in feats)) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("kind,order", [("ouster", {}), ("ouster_odyssey", {}), ("ouster_r8", dict(organize=True)), ("hesai", dict(organize=True)),
                                        ("livox", {}), ("livox_custom2", {}), ("velodyne", {}),
                                        ("velodyne_anybotics", dict(transpose=True)), ("rslidar", dict(transpose=True))])
def test_host_mirror_prepare_input_for_every_point_type(tmp_path, kind, order):
    """ScanFrontEnd::prepareInput<PointT> with the C++ point structs and layoutOf<PointT>() == the typed oracle"""
    from mimosa_amd import capi
    from oracle import ref_cpu
    from test_point_types import CFG, HEADER_TS, make_sensor_scan
    rows, cols = 32, 96
    raw = make_sensor_scan(kind, rows=rows, cols=cols, order="col" if order else "row")
    if order.get("transpose"):
        width, height = rows, cols
    elif order.get("organize"):
        width, height = rows * cols, 1
    else:
        width, height = cols, rows
    cfg = capi.make_input_config(**CFG)
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<iIIiiIdQ", ref_cpu.POINT_KINDS.index(kind), width, height, int(order.get("transpose", False)),
                            int(order.get("organize", False)), 0, HEADER_TS, len(raw)))
        f.write(bytes(cfg) + b"\0" * (-len(bytes(cfg)) % 8))      # struct Header's tail padding
        f.write(struct.pack("<Q", raw.nbytes))
        f.write(raw.tobytes())
    out = subprocess.run([build_exe("point_types"), str(inp), str(outp)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    o = ref_cpu.prepare_input_typed(kind, raw, ref_cpu.make_input_config(**CFG), header_ts=HEADER_TS, width=width, height=height,
                                    transpose=order.get("transpose", False), organize=order.get("organize", False))
    blob = open(outp, "rb").read()
    n_full, n_geo, last_ns, n_u = struct.unpack("<4Q", blob[:32])
    assert (n_full, n_geo, last_ns, n_u) == (len(o["points_full"]), len(o["geometric_idxs"]), o["last_point_ns"], len(o["unique_ns"]))
    assert blob[32:32 + 32 * n_full] == np.ascontiguousarray(o["points_full"]).tobytes() or _same_but_padding(blob[32:32 + 32 * n_full], o["points_full"])
    assert np.array_equal(np.frombuffer(blob[32 + 32 * n_full:], np.uint32), o["unique_ns"])
    assert abs(float(out.stdout) - (HEADER_TS + last_ns * 1e-9)) < 1e-6


def _same_but_padding(got, want):
    from mimosa_amd import synth
    a = np.frombuffer(got, synth.POINT_DTYPE)
    b = np.frombuffer(np.ascontiguousarray(want).tobytes(), synth.POINT_DTYPE)
    return all(np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)) for k in ("x", "y", "z", "intensity", "t", "idx", "range"))


def _write_sharded_input(path):
    from mimosa_amd import synth
    room = np.array([20.0, 14.0, 3.0])
    map_xyz = synth.make_room(4321, 0, 0, room=room)
    scan, aux = synth.make_scan(n_rows=32, seed=99, n_cols=128, room=room, sensor_local=np.array([9.3, 6.6, 1.2]))
    R, t = synth.query_pose(aux["R_W_L"], aux["t_W_L"])
    steps = [(np.zeros(3), np.zeros(3)), (np.zeros(3), np.array([0.004, 0.003, -0.002])), (np.array([0.0, 0.0, 0.012]), np.array([0.06, -0.05, 0.02])),
             (np.array([0.004, -0.003, 0.02]), np.array([-0.09, 0.11, 0.03]))]
    poses = np.concatenate([_pose12(R @ synth.so3_exp(w), t + d) for w, d in steps])
    with open(path, "wb") as f:
        for arr, n in ((map_xyz.astype(np.float32), map_xyz.size), (scan, len(scan)), (poses, poses.size)):
            f.write(struct.pack("<Q", n))
            f.write(np.ascontiguousarray(arr).tobytes())
    return map_xyz, scan, (R, t)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [("local", "1"), ("local", "3"), ("rccl",)])
def test_sharded_host_mirror(tmp_path, mode):
    """ShardCommunicator / ShardedVoxelMap / ShardedICPFactor through gtsam::Values and gtsam::HessianFactor: every rank's
    global result equals the unsharded ICPFactor's on the full map (and the oracle's)."""
    from mimosa_amd import synth
    from oracle import ref_cpu
    exe = build_exe("sharded_pipeline")
    inp = tmp_path / "sh.bin"
    map_xyz, scan, (R, t) = _write_sharded_input(inp)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    out = subprocess.run([exe, str(inp), *mode], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout[out.stdout.index("{"):])
    assert d["worst_rel"] <= 1e-12 and d["hist_equal"] == 1 and d["ranks_equal"] == 1, d
    assert d["batch_worst_rel"] <= 1e-12 and d["async_worst_rel"] <= 1e-12, d   # ShardedICPFactor::linearizeBatch / linearizeAsync + wait
    assert d["collective"] == 1 and d["collectives_last"] == 3
    assert d["backend"] == ("rccl" if mode[0] == "rccl" else "local")
    assert sum(d["points_held"]) == len(scan)
    M = ref_cpu.Map()
    M.insert(map_xyz)
    ref = ref_cpu.ICP(M, scan, ref_cpu.make_config(**synth.enwide_config())).linearize(R, t)
    assert abs(d["H00"] - ref["H_ss"][0, 0]) <= 1e-9 * abs(ref["H_ss"][0, 0]) and abs(d["f"] - ref["f"]) <= 1e-9 * abs(ref["f"])


@pytest.mark.gpu
def test_manager_callback_for_every_point_type():
    """lidar::Manager::callback<PointT> runs prepareInput for each of the nine sensor point types (64 zero records each: all
    points filtered) and honours a FAILURE_* declaration result by skipping the message (sensor_manager_base.hpp:208-260)."""
    exe = build_exe("manager_types")
    out = subprocess.run([exe, "run"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and json.loads(out.stdout)["ok"] == 1, out.stderr[-2000:]
