#!/usr/bin/env python3
"""Pin kit, step 3: turn the outputs of the REAL mimosa (mimosa_pin, tools/pin_reference/pin_main.cpp) into fixtures under
tests/golden/pinned/ — what tests/test_reference_pin.py holds the oracle and the HIP path to.
usage: import_outputs.py <dir with <case>.out>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = ["enwide", "mode7_k4", "mode27_k8", "binary", "reg4dof"]


class Reader:
    def __init__(self, path):
        self.b = open(path, "rb").read()
        self.o = 0

    def vec(self, dtype):
        n = int(np.frombuffer(self.b, np.uint64, 1, self.o)[0])
        self.o += 8
        a = np.frombuffer(self.b, dtype, n, self.o).copy()
        self.o += a.nbytes
        return a


def main(d, out_dir=None):
    out_dir = out_dir or os.path.join(ROOT, "tests", "golden", "pinned")
    os.makedirs(out_dir, exist_ok=True)
    for case in CASES:
        p = os.path.join(d, case + ".out")
        if not os.path.exists(p):
            print("missing", p)
            continue
        r = Reader(p)
        checks = r.vec(np.int32)
        if not (checks.size == 2 and checks.all()):
            sys.exit(f"{case}: an assumption of the restatement does NOT hold in the reference (abs(double), deep map copy): {checks}")
        out = {"checks": checks}
        for tag in ("a", "b"):
            H = r.vec(np.float64)
            n = int(round(np.sqrt(H.size)))
            out[f"{tag}_H"] = H.reshape(n, n)
            out[f"{tag}_g"] = r.vec(np.float64)
            out[f"{tag}_f"] = r.vec(np.float64)[0]
            out[f"{tag}_status"] = r.vec(np.int32)
            out[f"{tag}_mean"] = r.vec(np.float64).reshape(-1, 3)
            out[f"{tag}_normal"] = r.vec(np.float64).reshape(-1, 3)
            loc = r.vec(np.float64)
            out[f"{tag}_loc_trans_comp"], out[f"{tag}_loc_rot_comp"] = loc[0:3], loc[3:6]
            out[f"{tag}_loc_trans_final"], out[f"{tag}_loc_rot_final"] = loc[6:9], loc[9:12]
            eig = r.vec(np.float64)
            out[f"{tag}_eig_trans"], out[f"{tag}_eig_rot"] = eig[:9].reshape(3, 3), eig[9:].reshape(3, 3)
            dg = r.vec(np.float64)
            out[f"{tag}_degen_rot"], out[f"{tag}_degen_trans"] = dg[:3], dg[3:]
        found = r.vec(np.int32)
        sq = r.vec(np.float64)
        k = sq.size // max(found.size, 1)
        out["knn_found"], out["knn_sq"] = found, sq.reshape(-1, k)
        out["knn_points"] = r.vec(np.float64).reshape(-1, k, 3)
        np.savez_compressed(os.path.join(out_dir, case + ".npz"), **out)
        print("pinned", case, "H", out["a_H"].shape, "points", len(out["a_status"]), "knn queries", len(found))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "pin_io")
