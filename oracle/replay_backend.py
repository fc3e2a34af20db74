"""ORACLE — TEST INFRASTRUCTURE ONLY.  The CPU twin of mimosa_amd.replay.HipBackend: the same replay loop
(mimosa_amd/replay.py) driven by the oracle restatements (oracle/ref_cpu.*, oracle/photo_ref.*).  Imported by tests/ and by
bench.py's cpu_baseline legs only; never by the product."""
import numpy as np

from mimosa_amd import synth, synth_photo
from oracle import photo_ref, ref_cpu


class OracleBackend:
    def __init__(self, cfg, mode=synth.ENWIDE_NEIGHBOR_MODE):
        reg = cfg.reg
        self.regd = reg
        self.cfg = ref_cpu.make_config(**reg)
        self.map = ref_cpu.Map(leaf=reg["target_ivox_map_leaf_size"], min_dist=reg["target_ivox_map_min_dist_in_voxel"], mode=mode)
        self.icfg = ref_cpu.make_input_config()
        self.I3, self.z3 = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
        self.photo = photo_ref.Photo(cfg.photo) if cfg.photometric else None

    def seed_map(self, xyz):
        self.map.insert(xyz)

    def prepare(self, raw):
        self.o = ref_cpu.prepare_input(raw, self.icfg)
        self.full_raw = np.frombuffer(self.o["points_full"].tobytes(), dtype=synth.POINT_DTYPE).copy()
        return self.o["unique_ns"]

    def deskew_and_preprocess(self, T_Le_Lt):
        o = self.o
        desk = ref_cpu.deskew(self.full_raw, o["unique_ns"], T_Le_Lt.astype(np.float32))
        if self.photo is not None:
            desk = self.photo.preprocess(self.full_raw, desk, o["unique_ns"], T_Le_Lt)   # corrected intensities come back
        self.body = ref_cpu.transform_f32(desk[o["geometric_idxs"].astype(np.int64)], self.I3, self.z3)
        kept = ref_cpu.downsample(self.body, self.regd["source_voxel_grid_filter_leaf_size"], 20,
                                  self.regd["source_voxel_grid_min_dist_in_voxel"])
        self.ds = self.body[kept]
        return len(kept)

    def make_factor(self):
        return ref_cpu.ICP(self.map, self.ds, self.cfg)

    def make_photo_factor(self):
        return self.photo.make_factor() if self.photo is not None and self.photo.features() else None

    def linearize_window(self, factors, poses):
        out = []
        for f, (R, t) in zip(factors, poses):
            r = f.linearize(R, t)
            out.append((np.asarray(r["H_ss"]).reshape(6, 6), np.asarray(r["b_s"]), float(r["f"])))
        return out

    def linearize_photo(self, pf, R, t):
        r = pf.linearize(R, t)
        return np.asarray(r["H_bb"]).reshape(6, 6), np.asarray(r["b_b"]), float(r["f"]), int(r["status_hist"][8])

    def update_map(self, R, t):
        W = ref_cpu.transform_f32(self.body, R.astype(np.float32), t.astype(np.float32))
        new = self.map.copy()
        new.insert(np.stack([W["x"], W["y"], W["z"]], 1))
        self.map = new

    def photo_update_map(self, pf, R, t):
        self.photo.update_map(pf, R, t, synth_photo.BIAS_DIRECTIONS)

    def release(self, f):
        pass
