"""-m gpu: K3's launch classes with several lanes per point (small clouds; icp_kernels.hip, "several lanes per query") against the
one-lane-per-point class on the same clouds: the per-point results — status, mean, normal of every point, cold and after a pose
step that re-associates part of them — must be IDENTICAL to the bit (the k-NN answer is exact in every class and the plane fit
sums in an order that does not depend on which lane scanned what), the k-NN counters equal, and the sums equal up to the order
in which the classes add their rows.  The class is chosen per process (environment), so each class runs in a worker."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [(24576, False), (5000, False), (333, False), (12000, True)]


def _run(tmp_path, tag, env):
    out = str(tmp_path / f"{tag}.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "lane_class_worker.py"), out] + [f"{n}:{int(b)}" for n, b in CASES],
                       env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]
    return np.load(out)


def test_lane_classes_agree_point_for_point(tmp_path):
    one = _run(tmp_path, "one", {"MH_QL2_MAX": "0", "MH_QL4_MAX": "0"})
    two = _run(tmp_path, "two", {"MH_QL2_MAX": "32768", "MH_QL4_MAX": "0"})
    four = _run(tmp_path, "four", {"MH_QL2_MAX": "32768", "MH_QL4_MAX": "32768"})
    for ci, (n_pts, binary) in enumerate(CASES):
        assert int(one[f"c{ci}_cold_n_knn"]) == n_pts
        for other in (two, four):
            for ph in ("cold", "warm"):
                p = f"c{ci}_{ph}_"
                for i in range(3):
                    assert np.array_equal(one[p + f"state{i}"], other[p + f"state{i}"], equal_nan=True), (n_pts, ph, i)
                for k in ("status_hist", "n_knn", "mean_candidates", "n_exact_fallback"):
                    assert np.array_equal(one[p + k], other[p + k]), (n_pts, ph, k)
                for k in ("H_ss", "b_s", "f", "loc_trans_comp", "loc_rot_comp") + (("H_st", "H_tt", "b_t") if binary else ()):
                    a, b = np.asarray(one[p + k], float), np.asarray(other[p + k], float)
                    assert np.linalg.norm(a - b) <= 1e-12 * max(np.linalg.norm(a), 1e-300), (n_pts, ph, k)
