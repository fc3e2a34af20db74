#!/usr/bin/env python3
"""The halo decision as far as arithmetic on one machine allows: for shard blocks of
B = 4 / 8 / 16 voxels per axis and P = 2 / 4 / 8 ranks on the configs[1] world (2 x 5 rooms; the 10 x 10-room map of configs[2]
has the same per-room geometry, its totals scale by 10) —
  storage     points a rank stores (owned blocks + one-voxel halo) vs its fair share; total stored vs the map
  migration   scan points that change owner per pose step (Gauss-Newton-sized to decimetres), the largest per-destination count,
              and the bytes one fixed-size all-to-all of the native protocol then carries per rank
  alternative what the query-exchange design (no halo: forward every query whose 19-neighbourhood touches a foreign block, get the
              owner's top-5 back) would move per linearize
CPU only (numpy); prints a markdown table and one JSON line."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from mimosa_amd import synth

P1, P2, P3 = np.uint64(9132043225175502913), np.uint64(7277549399757405689), np.uint64(6673468629021231217)
LEAF = 0.5


def owner_hash(v, log2, world):
    """Rounds 3-4: XOR hash of the block coordinates (kept for the comparison column)."""
    b = (v >> log2).astype(np.int64).astype(np.uint64)
    with np.errstate(over="ignore"):
        h = (b[..., 0] * P1) ^ (b[..., 1] * P2) ^ (b[..., 2] * P3)
    return (h % np.uint64(world)).astype(np.int16)


def owner(v, log2, world):
    """The library's owner function (shard_kernels.hip owner_of_block): the lattice colouring of tools/lattice_table.py."""
    import lattice_table
    return lattice_table.owner_of_block(np.asarray(v, np.int64) >> log2, world)


def vox(xyz):
    return np.floor(np.asarray(xyz, np.float64) / LEAF).astype(np.int64)


OFFS19 = np.array([(i, j, k) for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1) if not (i and j and k)], np.int64)
OFFS27 = np.array([(i, j, k) for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)], np.int64)


def main():
    rooms = [xyz for _, _, xyz in synth.make_map_rooms(2, 5)]
    # the stored map: greedy 0.15 m rule keeps ~99 % of this world's points; occupancy per voxel is what matters here
    mv = vox(np.concatenate(rooms))
    n_map = len(mv)
    pts, aux = synth.make_scan(128)
    R, t = synth.query_pose(aux["R_W_L"], aux["t_W_L"])
    P_s = synth.points_xyz(pts).astype(np.float64)
    steps = [("1 cm / 0.5 mrad", np.array([0.0, 0.0, 5e-4]), np.array([0.008, 0.006, 0.0])),
             ("5 cm / 3 mrad", np.array([0.0, 0.0, 3e-3]), np.array([0.04, 0.03, 0.0])),
             ("30 cm / 20 mrad", np.array([0.0, 0.0, 2e-2]), np.array([0.25, 0.16, 0.0]))]
    rows, out = [], []
    for log2 in (2, 3, 4):
        B = 1 << log2
        for world in (2, 4, 8):
            own = owner(mv, log2, world)
            stored = np.zeros(world, np.int64)
            need = np.zeros((world, n_map), bool)
            for o in OFFS27:
                ow = owner(mv + o, log2, world)
                for r in range(world):
                    need[r] |= ow == r
            stored = need.sum(1)
            q0 = P_s @ R.T + t
            o0 = owner(vox(q0), log2, world)
            # boundary queries: some neighbour voxel of the centre voxel belongs to another rank
            c0 = vox(q0)
            foreign = np.zeros(len(c0), np.int64)
            seen = [o0]
            nf = np.zeros(len(c0), bool)
            owners_touched = np.zeros((len(c0), world), bool)
            for o in OFFS19:
                ow = owner(c0 + o, log2, world)
                owners_touched[np.arange(len(c0)), ow] = True
            owners_touched[np.arange(len(c0)), o0] = False
            n_foreign = owners_touched.sum(1)
            qc, qh = np.bincount(o0, minlength=world), np.bincount(owner_hash(vox(q0), log2, world), minlength=world)
            rec = dict(block=B, world=world, query_imbalance=float(qc.max() / qc.mean()), query_imbalance_xor_hash=float(qh.max() / qh.mean()),
                       queries_min_max=[int(qc.min()), int(qc.max())], stored_total_over_map=float(stored.sum() / n_map), stored_max_over_map=float(stored.max() / n_map),
                       fair_share=1.0 / world, boundary_query_fraction=float((n_foreign > 0).mean()), foreign_owners_per_query=float(n_foreign.mean()),
                       query_exchange_bytes_per_linearize=int(n_foreign.sum() * (16 + 80)), steps=[])
            for name, w, d in steps:
                q1 = P_s @ (R @ synth.so3_exp(w)).T + (t + d)
                o1 = owner(vox(q1), log2, world)
                mv_mask = o1 != o0
                # per (source rank, destination) counts with the scan block-partitioned over the ranks after the cold routing:
                # a point sits on o0 and leaves for o1
                pair = np.zeros((world, world), np.int64)
                np.add.at(pair, (o0[mv_mask], o1[mv_mask]), 1)
                mx = int(pair.max())
                cap = max(256, 1 << int(np.ceil(np.log2(max(4 * mx, 1)))))
                rec["steps"].append(dict(step=name, migrated_fraction=float(mv_mask.mean()), max_movers_per_pair=mx, segment_records=cap,
                                         all_to_all_bytes_per_rank=int(world * (16 + cap * 112)), payload_bytes_total=int(mv_mask.sum() * 112)))
            out.append(rec)
            s = rec["steps"]
            rows.append(f"| {B} | {world} | {rec['stored_total_over_map']:.2f}x | {rec['stored_max_over_map'] * 100:.1f} % ({100.0 / world:.1f} %) | "
                        f"{rec['query_imbalance']:.2f} x ({rec['query_imbalance_xor_hash']:.2f} x) | "
                        f"{rec['boundary_query_fraction'] * 100:.0f} % / {rec['foreign_owners_per_query']:.2f} | {rec['query_exchange_bytes_per_linearize'] / 1e6:.1f} MB | "
                        + " | ".join(f"{x['migrated_fraction'] * 100:.2f} % / {x['max_movers_per_pair']} / {x['all_to_all_bytes_per_rank'] / 1e3:.0f} KB" for x in s) + " |")
    print("| B (voxels) | P | stored total / map | fullest rank (fair share) | queries on the fullest rank / fair share (XOR hash of rounds 3-4) | boundary queries / foreign owners per query | query-exchange traffic per linearize | "
          + " | ".join(f"step {n}: migrated / max per pair / all-to-all per rank" for n, _, _ in steps) + " |")
    print("|---|---|---|---|---|---|---|" + "---|" * len(steps))
    print("\n".join(rows))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
