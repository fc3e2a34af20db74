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


def assert_result_parity(got, ref, binary=False, tol=TOL, check_eigvec=True, check_counters=True):
    assert np.array_equal(got["status_hist"], ref["status_hist"]), (got["status_hist"], ref["status_hist"])
    assert rel(got["H_ss"], ref["H_ss"]) <= tol
    assert rel(got["b_s"], ref["b_s"]) <= tol
    assert abs(got["f"] - ref["f"]) <= tol * max(abs(ref["f"]), 1e-300)
    if binary:
        assert rel(got["H_st"], ref["H_st"]) <= tol
        assert rel(got["H_tt"], ref["H_tt"]) <= tol
        assert rel(got["b_t"], ref["b_t"]) <= tol
    # Localizabilities are square roots of the eigenvalues of 3 x 3 blocks of H (utils.hpp:308-313).  An eigenvalue at the
    # rounding level of its block (a rank-deficient H: a handful of valid points, all on one plane) has a solver-dependent
    # sign, so its root is NaN in one implementation and 1e-7 in the other — in the reference too, whatever Eigen's rounding
    # happens to do.  The EIGENVALUES are what is comparable: squares, NaN read as 0, absolute tolerance scaled by the
    # largest.  (Found by the wide fuzz sweep, MH_FUZZ_EXTRA; the default cases are all well conditioned.)
    well_conditioned = True
    for k in ("loc_trans_final", "loc_rot_final"):
        g2, r2 = np.nan_to_num(np.asarray(got[k], float)) ** 2, np.nan_to_num(np.asarray(ref[k], float)) ** 2
        assert np.abs(g2 - r2).max() <= tol * max(r2.max(), 1e-300), k
        well_conditioned = well_conditioned and np.all(np.isfinite(ref[k])) and r2.min() > 1e-9 * r2.max()
    for k in ("loc_trans_comp", "loc_rot_comp"):
        if np.all(np.isfinite(ref[k])):
            assert rel(got[k], ref[k]) <= tol, k
    # the degeneracy info inverts the Schur complements of H (geometric_factor.hpp:413-428): meaningful, and comparable,
    # only when H is not singular to rounding
    # (fewer valid points than unknowns make H singular whatever its diagonal blocks look like; a singular Schur complement
    # shows as a 1e8 entry next to O(1) ones)
    # (six valid points on two planes: both diagonal blocks fine, H of rank 4 — wide sweep, seed 1332: 3e-4 apart)
    ev = np.linalg.eigvalsh(np.asarray(ref["H_ss"], float))
    well_conditioned = well_conditioned and int(ref["status_hist"][8]) >= 6 and ev.min() > 1e-9 * ev.max()
    for k in ("degen_rot", "degen_trans"):
        r = np.asarray(ref[k], float)
        if well_conditioned and np.all(np.isfinite(r)) and np.all(np.isfinite(got[k])) and r.max() < 1e5 * max(r.min(), 1e-300):
            assert rel(got[k], ref[k]) <= tol, k
    if check_eigvec:
        assert eigvec_equal_mod_sign(got["eigvec_trans"], ref["eigvec_trans"])
        assert eigvec_equal_mod_sign(got["eigvec_rot"], ref["eigvec_rot"])
    if check_counters:  # (statistics: a sharded call REPEATED behind later pipelined calls re-associates points it had already counted)
        assert got["n_knn"] == ref["n_knn"]
        assert abs(got["mean_candidates"] - ref["mean_candidates"]) <= 1e-9 * max(1.0, ref["mean_candidates"])
    assert got["linearize_count"] == ref["linearize_count"]


def assert_state_parity(got_state, ref_state, tol=1e-9):
    st_g, mean_g, nrm_g = got_state[:3]
    st_r, mean_r, nrm_r = ref_state[:3]
    assert np.array_equal(st_g, st_r)
    assert np.abs(mean_g - mean_r).max() <= tol
    # The cached normal is the eigenvector of the smallest eigenvalue of a k-point covariance: its conditioning is the
    # eigen-gap.  A Valid plane has w0 << w1 (every point within plane_validity_distance of it) and is held to `tol`;
    # a rejected neighbourhood (CorresPlaneInvalid = 6 and friends: w1 / w0 can be ~1.1, found by the wide fuzz sweep,
    # seed 225) amplifies the two solvers' 1e-16 differences by 1 / gap and is held to 1e-6.  Neither H nor b read it.
    valid = st_r == 8
    if valid.any():
        assert np.abs(nrm_g[valid] - nrm_r[valid]).max() <= tol
    if (~valid).any():
        assert np.abs(nrm_g[~valid] - nrm_r[~valid]).max() <= 1e-6
