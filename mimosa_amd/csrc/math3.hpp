// Small fp64 3x3 helpers shared by the HIP kernels and the host epilogue of libmimosa_hip.
// Replaces the Eigen calls on the reference hot path:
//   Eigen::SelfAdjointEigenSolver<Matrix3d>  (geometric_factor.hpp:196, include/mimosa/utils.hpp:308-313)
//   Matrix3d::inverse()                      (geometric_factor.hpp:413-422)
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#define MH_HD __host__ __device__ inline
#else
#define MH_HD inline
#endif

namespace mh
{
// Symmetric 3x3 eigen-decomposition, cyclic Jacobi in fp64.  A row-major (only the upper triangle
// is read); on return w[0] <= w[1] <= w[2] and V (row-major) holds the eigenvectors in COLUMNS.
// Branch-light and register-resident: every lane runs the same fixed sweep schedule, rotations
// whose off-diagonal element is already negligible are skipped by predication.  Converges
// quadratically; 3x3 needs <= 5 sweeps for fp64 round-off (6 are run).  The sign of each
// eigenvector is arbitrary (as it is in Eigen); the plane normal's sign is fixed afterwards by the
// "normal faces the sensor" rule (geometric_factor.hpp:217-220).
MH_HD void sym_eigen3(const double A[9], double w[3], double V[9])
{
  double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[4], a12 = A[5], a22 = A[8];
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
  // Scale-invariant convergence floor: elements below eps * trace-scale cannot change the result.
  const double scale = fabs(a00) + fabs(a11) + fabs(a22) + fabs(a01) + fabs(a02) + fabs(a12);
  const double tiny = scale * 1e-19;

#define MH_JACOBI_ROT(app, aqq, apq, apr, aqr, vp0, vq0, vp1, vq1, vp2, vq2) \
  if (fabs(apq) > tiny) {                                                     \
    const double theta = (aqq - app) / (2.0 * apq);                           \
    const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0)); \
    const double c = 1.0 / sqrt(tt * tt + 1.0);                               \
    const double s = tt * c;                                                  \
    app -= tt * apq;                                                          \
    aqq += tt * apq;                                                          \
    apq = 0.0;                                                                \
    const double pr = apr, qr = aqr;                                          \
    apr = c * pr - s * qr;                                                    \
    aqr = s * pr + c * qr;                                                    \
    double tp, tq;                                                            \
    tp = vp0; tq = vq0; vp0 = c * tp - s * tq; vq0 = s * tp + c * tq;         \
    tp = vp1; tq = vq1; vp1 = c * tp - s * tq; vq1 = s * tp + c * tq;         \
    tp = vp2; tq = vq2; vp2 = c * tp - s * tq; vq2 = s * tp + c * tq;         \
  }

#pragma unroll 1
  for (int sweep = 0; sweep < 6; ++sweep) {
    if (fabs(a01) + fabs(a02) + fabs(a12) <= tiny) break;
    // (p,q) = (0,1): r = 2
    MH_JACOBI_ROT(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21)
    // (p,q) = (0,2): r = 1   (a_pr = a01, a_qr = a21 = a12)
    MH_JACOBI_ROT(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22)
    // (p,q) = (1,2): r = 0   (a_pr = a10 = a01, a_qr = a20 = a02)
    MH_JACOBI_ROT(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22)
  }
#undef MH_JACOBI_ROT

  // ascending sort (3-element network), columns follow
  double e0 = a00, e1 = a11, e2 = a22;
#define MH_SWAPD(x, y) { const double t_ = x; x = y; y = t_; }
  if (e1 < e0) {
    MH_SWAPD(e0, e1) MH_SWAPD(v00, v01) MH_SWAPD(v10, v11) MH_SWAPD(v20, v21)
  }
  if (e2 < e1) {
    MH_SWAPD(e1, e2) MH_SWAPD(v01, v02) MH_SWAPD(v11, v12) MH_SWAPD(v21, v22)
  }
  if (e1 < e0) {
    MH_SWAPD(e0, e1) MH_SWAPD(v00, v01) MH_SWAPD(v10, v11) MH_SWAPD(v20, v21)
  }
#undef MH_SWAPD
  w[0] = e0; w[1] = e1; w[2] = e2;
  V[0] = v00; V[1] = v01; V[2] = v02;
  V[3] = v10; V[4] = v11; V[5] = v12;
  V[6] = v20; V[7] = v21; V[8] = v22;
}

// computeLocalizability, include/mimosa/utils.hpp:308-313: sqrt(eigenvalues) ascending + eigenvectors
MH_HD void compute_localizability(const double JtJ[9], double loc[3], double E[9])
{
  double w[3];
  sym_eigen3(JtJ, w, E);
  loc[0] = sqrt(w[0]);
  loc[1] = sqrt(w[1]);
  loc[2] = sqrt(w[2]);
}

MH_HD void mat3_mul(const double A[9], const double B[9], double C[9])
{
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}

// cofactor / determinant inverse (what Eigen does for fixed 3x3)
MH_HD void mat3_inv(const double A[9], double R[9])
{
  const double c00 = A[4] * A[8] - A[5] * A[7];
  const double c10 = A[5] * A[6] - A[3] * A[8];
  const double c20 = A[3] * A[7] - A[4] * A[6];
  const double inv = 1.0 / (A[0] * c00 + A[1] * c10 + A[2] * c20);
  R[0] = c00 * inv;
  R[3] = c10 * inv;
  R[6] = c20 * inv;
  R[1] = (A[2] * A[7] - A[1] * A[8]) * inv;
  R[4] = (A[0] * A[8] - A[2] * A[6]) * inv;
  R[7] = (A[1] * A[6] - A[0] * A[7]) * inv;
  R[2] = (A[1] * A[5] - A[2] * A[4]) * inv;
  R[5] = (A[2] * A[3] - A[0] * A[5]) * inv;
  R[8] = (A[0] * A[4] - A[1] * A[3]) * inv;
}

}  // namespace mh
