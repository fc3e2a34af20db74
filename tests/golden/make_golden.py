#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ (committed data, not code).

The reference (ntnu-arl/mimosa) has no tests and cannot be built or imported here, so these vectors
come from the build's own INDEPENDENT numpy restatement (oracle/numpy_ref.py) of the reference
algorithm — PARITY UNPINNED.  They pin the C++ oracle (CPU suite) and the HIP path (GPU suite) to the
same answers and keep the three implementations from drifting together.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from mimosa_amd import synth  # noqa: E402
from oracle import numpy_ref  # noqa: E402


def linearize_case(name, cfg_over=None, mode=19, binary=False, g_unit=(0.0, 0.0, -1.0), k=None):
    m, pts, aux = synth.small_world()
    R, t = synth.query_pose(aux["R_W_L"], aux["t_W_L"])
    cfg = synth.enwide_config()
    if cfg_over:
        cfg.update(cfg_over)
    if k:
        cfg["num_corres_points"] = k
    vm = numpy_ref.VoxelMap(mode=mode, lru_horizon=synth.ENWIDE_LRU_HORIZON)
    vm.insert(m)
    xyz = synth.points_xyz(pts)
    kw = {}
    if binary:
        Rt = synth.so3_exp(np.array([0.01, -0.02, 0.015]))
        tt = np.array([0.02, -0.01, 0.03])
        kw = dict(R_tgt=Rt, t_tgt=tt)
        Rs, ts = Rt @ R, Rt @ t + tt
    else:
        Rs, ts = R, t
    r1, st = numpy_ref.linearize(vm, xyz, cfg, Rs, ts, g_unit, **kw)
    state1 = {k_: v.copy() for k_, v in st.items() if isinstance(v, np.ndarray)}
    # relinearize after a small rotation: part of the cloud re-associates, the rest hits the cache
    R3 = R @ synth.so3_exp(np.array([0.0, 0.0, 0.015]))
    Rs3 = (kw["R_tgt"] @ R3) if binary else R3
    r2, st = numpy_ref.linearize(vm, xyz, cfg, Rs3, ts, g_unit, state=st, **kw)
    out = dict(cfg_keys=np.array(sorted(cfg)), cfg_vals=np.array([float(cfg[k_]) for k_ in sorted(cfg)]),
               mode=mode, binary=int(binary), g_unit=np.array(g_unit, float), R1=Rs, t1=ts, R2=Rs3, t2=ts)
    if binary:
        out.update(R_tgt=kw["R_tgt"], t_tgt=kw["t_tgt"])
    for tag, r in (("a", r1), ("b", r2)):
        for key in ("H_ss", "H_st", "H_tt", "b_s", "b_t", "f", "status_hist", "n_knn", "mean_candidates",
                    "loc_trans_comp", "loc_rot_comp", "loc_trans_final", "loc_rot_final", "degen_rot", "degen_trans"):
            out[f"{tag}_{key}"] = np.asarray(r[key])
    out.update(status1=state1["status"], mean1=state1["mean"], normal1=state1["normal"],
               status2=st["status"], mean2=st["mean"], normal2=st["normal"])
    np.savez_compressed(os.path.join(HERE, f"linearize_{name}.npz"), **out)
    print(name, r1["status_hist"], r2["status_hist"], r2["n_knn"])


def main():
    linearize_case("enwide")
    linearize_case("mode7_k4", mode=7, k=4)
    linearize_case("mode27_k8", mode=27, k=8)
    linearize_case("binary", binary=True)
    gu = np.array([0.05, -0.02, -1.0])
    linearize_case("reg4dof", cfg_over=dict(reg_4_dof=1), g_unit=tuple(gu / np.linalg.norm(gu)))

    # deskew / body transform: f32, reference operation order, no FMA
    pts, aux = synth.make_scan(16, skew=True, n_cols=256)
    xyz = synth.points_xyz(pts)
    R_B_L = synth.so3_exp(np.array([0.01, 0.02, -0.03])).astype(np.float32)
    t_B_L = np.array([0.1, -0.05, 0.2], np.float32)
    desk = numpy_ref.deskew(xyz, pts["t"], aux["unique_ns"], aux["Rt12"])
    body = numpy_ref.transform_f32(desk, R_B_L, t_B_L)
    np.savez_compressed(os.path.join(HERE, "deskew.npz"), xyz=xyz, t=pts["t"], unique_ns=aux["unique_ns"],
                        Rt12=aux["Rt12"], R_B_L=R_B_L, t_B_L=t_B_L, deskewed=desk, body=body)

    # voxel down-sampler (order-sensitive kept-index list)
    rng = np.random.default_rng(11)
    cloud = (rng.normal(0, 1.0, (4096, 3)) * np.array([3.0, 2.0, 0.3])).astype(np.float32)
    kept = numpy_ref.downsample(cloud, 0.5, 20, 0.15)
    np.savez_compressed(os.path.join(HERE, "downsample.npz"), cloud=cloud, kept=kept)

    # scan front end: Manager::prepareInput on a raw Ouster cloud with dropouts -> deskew -> body subset ->
    # down-sampler (every stage's output in order)
    raw, raux = synth.make_raw_scan(16, seed=4242, n_cols=128)
    fcfg = dict(range_min=2.0, range_max=60.0, intensity_min=10.0, intensity_max=2000.0, ns_max=1.0e9, z_offset=-0.03618,
                create_full_res_pointcloud=True, point_skip_divisor=2, ring_skip_divisor=1)
    pi = numpy_ref.prepare_input(raw, **fcfg)
    full = pi["points_full"]
    col = {int(u): c for c, u in enumerate(raux["unique_ns"])}
    Rt12f = np.stack([raux["Rt12"][col[int(u)]] for u in pi["unique_ns"]])
    fxyz = np.stack([full["x"], full["y"], full["z"]], 1)
    fdesk = numpy_ref.deskew(fxyz, full["t"], pi["unique_ns"], Rt12f)
    fbody = numpy_ref.transform_f32(fdesk[pi["geometric_idxs"]], R_B_L, t_B_L)
    fkept = numpy_ref.downsample(fbody, 1.0, 3, 0.5)  # coarse leaf, cap 3, large min distance: every rule fires
    np.savez_compressed(os.path.join(HERE, "frontend.npz"), raw=raw.view(np.uint8).reshape(len(raw), 32),
                        cfg_keys=np.array(sorted(fcfg)), cfg_vals=np.array([float(fcfg[k_]) for k_ in sorted(fcfg)]),
                        full_xyz=fxyz, full_intensity=full["intensity"], full_t=full["t"], full_idx=full["idx"],
                        full_range=full["range"], geometric_idxs=pi["geometric_idxs"], unique_ns=pi["unique_ns"],
                        last_point_ns=pi["last_point_ns"], Rt12=Rt12f, R_B_L=R_B_L, t_B_L=t_B_L, deskewed=fdesk,
                        body=fbody, kept=fkept)
    print("frontend", len(raw), len(fxyz), len(pi["geometric_idxs"]), len(fkept))

    # map insert: kept points in voxel order after three inserts + LRU purge behaviour
    m, _, _ = synth.small_world()
    vm = numpy_ref.VoxelMap(lru_horizon=2, lru_clear_cycle=2)
    chunks = [c + np.float32(i) * np.array([4.0, 0, 0], np.float32)  # walk away: old voxels age out
              for i, c in enumerate(np.array_split(m[:3000], 6))]
    sizes = []
    for c in chunks:
        vm.insert(c)
        sizes.append(vm.num_points)
    cloud_out = np.concatenate([np.array(vm.cells[c][0]) for c in vm.order if vm.cells[c][0]]).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "map_lru.npz"), chunks=np.concatenate(chunks), n_chunks=6,
                        sizes=np.array(sizes), cloud=cloud_out)
    print("map_lru sizes", sizes)


if __name__ == "__main__":
    main()
