"""Helpers shared by the CPU (oracle) and GPU (HIP) golden-fixture tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LINEARIZE_CASES = ["enwide", "mode7_k4", "mode27_k8", "binary", "reg4dof"]
INT_KEYS = {"num_corres_points", "use_huber", "reg_4_dof", "project_on_degneneracy"}


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def cfg_of(g):
    return {str(k): (int(v) if str(k) in INT_KEYS else float(v)) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def check_linearize(g, tag, got, tol=1e-5):
    """got: result dict of an implementation, compared with the fixture's pass `tag` ('a' | 'b')."""
    assert np.array_equal(np.asarray(got["status_hist"]), g[f"{tag}_status_hist"])
    keys = ["H_ss", "b_s", "loc_trans_comp", "loc_rot_comp", "loc_trans_final", "loc_rot_final", "degen_rot", "degen_trans"]
    if int(g["binary"]):
        keys += ["H_st", "H_tt", "b_t"]
    for k in keys:
        assert rel(got[k], g[f"{tag}_{k}"]) <= tol, k
    assert abs(float(got["f"]) - float(g[f"{tag}_f"])) <= tol * abs(float(g[f"{tag}_f"]))
    assert int(got["n_knn"]) == int(g[f"{tag}_n_knn"])
    assert abs(float(got["mean_candidates"]) - float(g[f"{tag}_mean_candidates"])) < 1e-9


def check_state(g, n, status, mean, normal, tol=1e-9):
    assert np.array_equal(status, g[f"status{n}"])
    assert np.abs(mean - g[f"mean{n}"]).max() <= tol
    assert np.abs(normal - g[f"normal{n}"]).max() <= tol
