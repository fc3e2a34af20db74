"""Shared parity assertions: HIP path vs oracle (the bar of BASELINE.json: 1e-5 relative)."""
import numpy as np

TOL = 1e-5  # BASELINE.json north_star: residuals / Jacobians within 1e-5 relative of the reference


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def eigvec_equal_mod_sign(A, B, tol=1e-7):
    """Eigenvector matrices (columns) equal up to a per-column sign."""
    A, B = np.asarray(A), np.asarray(B)
    for c in range(3):
        d = min(np.linalg.norm(A[:, c] - B[:, c]), np.linalg.norm(A[:, c] + B[:, c]))
        if d > tol:
            return False
    return True


def assert_result_parity(got, ref, binary=False, tol=TOL, check_eigvec=True):
    assert np.array_equal(got["status_hist"], ref["status_hist"]), (got["status_hist"], ref["status_hist"])
    assert rel(got["H_ss"], ref["H_ss"]) <= tol
    assert rel(got["b_s"], ref["b_s"]) <= tol
    assert abs(got["f"] - ref["f"]) <= tol * max(abs(ref["f"]), 1e-300)
    if binary:
        assert rel(got["H_st"], ref["H_st"]) <= tol
        assert rel(got["H_tt"], ref["H_tt"]) <= tol
        assert rel(got["b_t"], ref["b_t"]) <= tol
    for k in ("loc_trans_final", "loc_rot_final", "loc_trans_comp", "loc_rot_comp", "degen_rot", "degen_trans"):
        if np.all(np.isfinite(ref[k])):
            assert rel(got[k], ref[k]) <= tol, k
    if check_eigvec:
        assert eigvec_equal_mod_sign(got["eigvec_trans"], ref["eigvec_trans"])
        assert eigvec_equal_mod_sign(got["eigvec_rot"], ref["eigvec_rot"])
    assert got["n_knn"] == ref["n_knn"]
    assert abs(got["mean_candidates"] - ref["mean_candidates"]) <= 1e-9 * max(1.0, ref["mean_candidates"])
    assert got["linearize_count"] == ref["linearize_count"]


def assert_state_parity(got_state, ref_state, tol=1e-9):
    st_g, mean_g, nrm_g = got_state[:3]
    st_r, mean_r, nrm_r = ref_state[:3]
    assert np.array_equal(st_g, st_r)
    assert np.abs(mean_g - mean_r).max() <= tol
    assert np.abs(nrm_g - nrm_r).max() <= tol
