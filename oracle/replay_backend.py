"""ORACLE — TEST INFRASTRUCTURE ONLY.  The CPU twin of mimosa_amd.replay.HipBackend: the same replay loop
(mimosa_amd/replay.py) driven by the oracle restatements (oracle/ref_cpu.*).  Imported by tests/ and by
bench.py's cpu_baseline legs only; never by the product."""
import numpy as np

from mimosa_amd import synth
from oracle import ref_cpu


class OracleBackend:
    def __init__(self, reg: dict, mode=synth.ENWIDE_NEIGHBOR_MODE):
        self.regd = reg
        self.cfg = ref_cpu.make_config(**reg)
        self.map = ref_cpu.Map(leaf=reg["target_ivox_map_leaf_size"], min_dist=reg["target_ivox_map_min_dist_in_voxel"], mode=mode)
        self.icfg = ref_cpu.make_input_config()
        self.I3, self.z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)

    def seed_map(self, xyz):
        self.map.insert(xyz)

    def front_end(self, raw, aux):
        o = ref_cpu.prepare_input(raw, self.icfg)
        full = np.frombuffer(o["points_full"].tobytes(), dtype=synth.POINT_DTYPE).copy()
        Rt12 = aux["Rt12"][np.searchsorted(aux["unique_ns"], o["unique_ns"])]
        desk = ref_cpu.deskew(full, o["unique_ns"], Rt12)
        self.body = ref_cpu.transform_f32(desk[o["geometric_idxs"].astype(np.int64)], self.I3, self.z3)
        kept = ref_cpu.downsample(self.body, self.regd["source_voxel_grid_filter_leaf_size"], 20,
                                  self.regd["source_voxel_grid_min_dist_in_voxel"])
        self.ds = self.body[kept]
        return len(kept)

    def make_factor(self):
        return ref_cpu.ICP(self.map, self.ds, self.cfg)

    def linearize(self, f, R, t):
        r = f.linearize(R, t)
        return np.asarray(r["H_ss"]).reshape(6, 6), np.asarray(r["b_s"]), float(r["f"]), r

    def body_cloud_xyz(self):
        return self.body

    def update_map(self, body, R, t):
        W = ref_cpu.transform_f32(body, R.astype(np.float32), t.astype(np.float32))
        new = self.map.copy()
        new.insert(np.stack([W["x"], W["y"], W["z"]], 1))
        self.map = new
