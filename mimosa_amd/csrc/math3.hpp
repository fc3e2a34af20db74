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
// 1 / sqrt(x): the device's reciprocal-square-root sequence (one refinement chain instead of a square root followed by a division)
MH_HD double mh_rsqrt(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return rsqrt(x);
#else
  return 1.0 / std::sqrt(x);
#endif
}

// Reciprocal without the IEEE division sequence (v_div_scale x 2, v_rcp_f64, 5 fma, v_div_fmas, v_div_fixup: ~11 dependent
// instructions): the hardware estimate (v_rcp_f64, ~2^-23 relative) and Newton refinements y <- y + y (1 - x y), each of which
// squares the error.  mh_rcp: two steps, <= 1 ulp-ish (2^-52 relative) for normal x; mh_rcp1: one step (~2^-46), for the
// self-correcting Newton iterations of the eigen solvers, whose fixed point does not depend on how accurately the step is
// divided.  x must be a normal, non-zero number (no scaling for denormal / huge operands: callers test for that).  On the
// host these are plain divisions.
MH_HD double mh_rcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(x);
  y = fma(fma(-x, y, 1.0), y, y);
  y = fma(fma(-x, y, 1.0), y, y);
  return y;
#else
  return 1.0 / x;
#endif
}
MH_HD double mh_rcp1(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, y, 1.0), y, y);
#else
  return 1.0 / x;
#endif
}
// a / b to the last bit or two: the refined reciprocal, then one residual correction of the quotient
MH_HD double mh_div(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = mh_rcp(b);
  const double q = a * y;
  return fma(fma(-b, q, a), y, q);
#else
  return a / b;
#endif
}

// Symmetric 3x3 eigen-decomposition, cyclic Jacobi in fp64.  A row-major (only the upper triangle
// is read); on return w[0] <= w[1] <= w[2] and V (row-major) holds the eigenvectors in COLUMNS.
// Branch-light and register-resident: every lane runs the same fixed sweep schedule, rotations
// whose off-diagonal element is already negligible are skipped by predication.  Converges
// quadratically; 3x3 needs <= 5 sweeps for fp64 round-off (6 are run).  The sign of each
// eigenvector is arbitrary (as it is in Eigen); the plane normal's sign is fixed afterwards by the
// "normal faces the sensor" rule (geometric_factor.hpp:217-220).
MH_HD void sym_eigen3_jacobi(const double A[9], double w[3], double V[9])
{
  double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[4], a12 = A[5], a22 = A[8];
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
  // Scale-invariant convergence floor: elements below eps * trace-scale cannot change the result.
  const double scale = fabs(a00) + fabs(a11) + fabs(a22) + fabs(a01) + fabs(a02) + fabs(a12);
  const double tiny = scale * 1e-19;

#define MH_JACOBI_ROT(app, aqq, apq, apr, aqr, vp0, vq0, vp1, vq1, vp2, vq2) \
  if (fabs(apq) > tiny) {                                                     \
    /* t = tan(rotation angle), the smaller root of t^2 + 2 theta t - 1 = 0 with theta = (aqq - app) / (2 apq), written   \
       without theta: t = 2 apq sgn(d) / (|d| + sqrt(d^2 + 4 apq^2)), d = aqq - app — one square root and one division   \
       (fp64 divisions and square roots are ~100-cycle dependent sequences on the device, and this runs on a single lane   \
       in the serial tail of K3: 3 divisions + 2 square roots per rotation cost 3.7 us per linearize) */                  \
    const double dd = aqq - app;                                              \
    const double tt = (dd >= 0 ? 2.0 : -2.0) * apq / (fabs(dd) + sqrt(dd * dd + 4.0 * apq * apq)); \
    const double c = mh_rsqrt(tt * tt + 1.0);                                 \
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

// The same decomposition without iterating, for the serial tail of K3 (one lane decomposes H_rr, another H_tt, while every
// other wave of the machine has finished: the Jacobi sweeps above are ~12 rotations of dependent fp64 division / square-root
// sequences, 3.5 us of a 38 us kernel).  Eigenvalues from the characteristic cubic of the scaled, shifted matrix (its
// trigonometric form, evaluated by a six-step Newton iteration instead of acos / cos); the eigenvector of the best-separated eigenvalue from the largest cross product of two rows of A - w I, the
// second from the 2 x 2 problem in its orthogonal complement, the third as their cross product (the construction of Eberly,
// "A robust eigensolver for 3 x 3 symmetric matrices"); one Rayleigh-quotient step per eigenvalue afterwards.  The result is
// VERIFIED — |A v - w v| against 1e-13 |A|, eigenvalue order — and anything that does not pass (clustered eigenvalues) goes
// through the Jacobi sweeps instead, so the accuracy is Jacobi's either way.
MH_HD void sym_eigen3(const double A[9], double w[3], double V[9])
{
  const double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[4], a12 = A[5], a22 = A[8];
  const double mx = fmax(fmax(fabs(a00), fabs(a11)), fmax(fmax(fabs(a22), fabs(a01)), fmax(fabs(a02), fabs(a12))));
  bool ok = mx > 0.0 && mx < 1e300;
  if (ok) {
    const double inv = mh_rcp(mx);
    const double b00 = a00 * inv, b01 = a01 * inv, b02 = a02 * inv, b11 = a11 * inv, b12 = a12 * inv, b22 = a22 * inv;
    const double norm = b01 * b01 + b02 * b02 + b12 * b12;
    const double q = (b00 + b11 + b22) * (1.0 / 3.0);
    const double c00 = b00 - q, c11 = b11 - q, c22 = b22 - q;
    const double p = sqrt((c00 * c00 + c11 * c11 + c22 * c22 + 2.0 * norm) * (1.0 / 6.0));
    ok = p > 1e-8;  // (nearly) a multiple of the identity: let the sweeps decide
    double e0 = 0, e1 = 0, e2 = 0;
    if (ok) {
      const double ip = mh_rcp(p);
      const double d00 = c00 * ip, d01 = b01 * ip, d02 = b02 * ip, d11 = c11 * ip, d12 = b12 * ip, d22 = c22 * ip;
      double half_det = 0.5 * (d00 * (d11 * d22 - d12 * d12) - d01 * (d01 * d22 - d12 * d02) + d02 * (d01 * d12 - d11 * d02));
      half_det = fmin(fmax(half_det, -1.0), 1.0);
      // The eigenvalues of the scaled, shifted matrix are the roots of g(x) = x^3 - 3 x - 2 half_det, i.e. 2 cos(theta / 3 +
      // 2 pi k / 3) with cos(theta) = half_det.  An acos and two cos are ~250 dependent fp64 instructions EACH on the device
      // (this runs on one lane, with a grid waiting: 2.5 us in round 3).  Instead: Newton on g for the root that stands
      // alone — the largest (in [sqrt 3, 2]) when half_det >= 0, started at 2, the smallest (in [-2, -sqrt 3]) otherwise,
      // started at -2: g is convex (concave) and monotone there, the iterates approach the root from outside without
      // overshoot, g' >= 6, the root is simple for every half_det, and the error goes 0.27 -> 5e-2 -> 2e-3 -> 2e-6 -> 2e-12
      // -> round-off: six steps.  The other two from the deflated quadratic x^2 + r x + (r^2 - 3).  (Accuracy does not rest
      // on this: Rayleigh step + residual check below, Jacobi sweeps as the fallback.)
      // Round 5: started on the chord sqrt 3 + (2 - sqrt 3) |half_det| instead (the root is 2 cos(acos(|half_det|) / 3): within
      // 0.014 of the chord, below it; the first step lands above the root, the rest descend): 1.4e-2 -> 1.5e-4 -> 2e-8 -> round-off,
      // four steps instead of six — every step is eight dependent fp64 operations on a single lane with the workgroup waiting.
      const double dd = 2.0 * half_det;
      double br = 1.7320508075688772 + 0.2679491924311228 * fabs(half_det);
      br = half_det >= 0.0 ? br : -br;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const double b2 = br * br;
        br -= ((b2 - 3.0) * br - dd) * mh_rcp1(3.0 * b2 - 3.0);  // (g' >= 6 on the iterates: a plain reciprocal)
      }
      const double sq = sqrt(fmax(12.0 - 3.0 * br * br, 0.0));
      const double blo = 0.5 * (-br - sq), bhi = 0.5 * (-br + sq);
      const double beta0 = half_det >= 0.0 ? blo : br, beta1 = half_det >= 0.0 ? bhi : blo, beta2 = half_det >= 0.0 ? br : bhi;
      e0 = q + p * beta0;
      e1 = q + p * beta1;
      e2 = q + p * beta2;
      // eigenvectors (of the scaled matrix b): start with the eigenvalue that stands alone
      auto evec_first = [&](double ev, double (&v)[3]) {
        const double m00 = b00 - ev, m11 = b11 - ev, m22 = b22 - ev;
        const double x0 = b01 * b12 - b02 * m11, y0 = b02 * b01 - m00 * b12, z0 = m00 * m11 - b01 * b01;
        const double x1 = b01 * m22 - b02 * b12, y1 = b02 * b02 - m00 * m22, z1 = m00 * b12 - b01 * b02;
        const double x2 = m11 * m22 - b12 * b12, y2 = b12 * b02 - b01 * m22, z2 = b01 * b12 - m11 * b02;
        const double n0 = x0 * x0 + y0 * y0 + z0 * z0, n1 = x1 * x1 + y1 * y1 + z1 * z1, n2 = x2 * x2 + y2 * y2 + z2 * z2;
        double vx = x0, vy = y0, vz = z0, nn = n0;
        if (n1 > nn) { vx = x1; vy = y1; vz = z1; nn = n1; }
        if (n2 > nn) { vx = x2; vy = y2; vz = z2; nn = n2; }
        const double r = nn > 0.0 ? mh_rsqrt(nn) : 0.0;
        v[0] = vx * r; v[1] = vy * r; v[2] = vz * r;
        return nn > 0.0;
      };
      auto evec_second = [&](const double (&v0)[3], double ev, double (&v)[3]) {
        // orthonormal complement {U, W} of v0
        double U[3], W[3];
        if (fabs(v0[0]) > fabs(v0[1])) {
          const double r = mh_rsqrt(v0[0] * v0[0] + v0[2] * v0[2]);
          U[0] = -v0[2] * r; U[1] = 0.0; U[2] = v0[0] * r;
        } else {
          const double r = mh_rsqrt(v0[1] * v0[1] + v0[2] * v0[2]);
          U[0] = 0.0; U[1] = v0[2] * r; U[2] = -v0[1] * r;
        }
        W[0] = v0[1] * U[2] - v0[2] * U[1]; W[1] = v0[2] * U[0] - v0[0] * U[2]; W[2] = v0[0] * U[1] - v0[1] * U[0];
        const double AU0 = b00 * U[0] + b01 * U[1] + b02 * U[2], AU1 = b01 * U[0] + b11 * U[1] + b12 * U[2], AU2 = b02 * U[0] + b12 * U[1] + b22 * U[2];
        const double AW0 = b00 * W[0] + b01 * W[1] + b02 * W[2], AW1 = b01 * W[0] + b11 * W[1] + b12 * W[2], AW2 = b02 * W[0] + b12 * W[1] + b22 * W[2];
        double m00 = U[0] * AU0 + U[1] * AU1 + U[2] * AU2 - ev, m01 = U[0] * AW0 + U[1] * AW1 + U[2] * AW2,
               m11 = W[0] * AW0 + W[1] * AW1 + W[2] * AW2 - ev;
        const double am00 = fabs(m00), am01 = fabs(m01), am11 = fabs(m11);
        double cu, cw;  // v = cu U + cw W in the null space of [[m00, m01], [m01, m11]]
        if (am00 >= am11) {
          if (fmax(am00, am01) > 0.0) {
            if (am00 >= am01) { const double t = mh_div(m01, m00); const double r = mh_rsqrt(1.0 + t * t); cu = t * r; cw = -r; }
            else { const double t = mh_div(m00, m01); const double r = mh_rsqrt(1.0 + t * t); cu = r; cw = -t * r; }
          } else { cu = 1.0; cw = 0.0; }
        } else {
          if (fmax(am11, am01) > 0.0) {
            if (am11 >= am01) { const double t = mh_div(m01, m11); const double r = mh_rsqrt(1.0 + t * t); cu = -r; cw = t * r; }
            else { const double t = mh_div(m11, m01); const double r = mh_rsqrt(1.0 + t * t); cu = -t * r; cw = r; }
          } else { cu = 1.0; cw = 0.0; }
        }
        v[0] = cu * U[0] + cw * W[0]; v[1] = cu * U[1] + cw * W[1]; v[2] = cu * U[2] + cw * W[2];
      };
      double va[3], vb[3], vc[3];
      // Fast path (round 5): all three eigenvectors as cross products of rows of b - e I — three INDEPENDENT chains the
      // scheduler interleaves, a third of the dependent depth of first / second-in-the-complement / third-as-cross-product.
      // Good whenever the eigenvalues are separated (the Hessian blocks of any scene with structure); accepted only if the
      // vectors come out orthogonal, and verified like the others below.  Otherwise: Eberly's construction, as before.
      bool fast = evec_first(e0, va);
      fast = evec_first(e1, vb) && fast;
      fast = evec_first(e2, vc) && fast;
      {
        const double dab = va[0] * vb[0] + va[1] * vb[1] + va[2] * vb[2], dac = va[0] * vc[0] + va[1] * vc[1] + va[2] * vc[2],
                     dbc = vb[0] * vc[0] + vb[1] * vc[1] + vb[2] * vc[2];
        fast = fast && fmax(fabs(dab), fmax(fabs(dac), fabs(dbc))) <= 1e-10;
      }
      if (fast) {
        // (nothing to do: va, vb, vc stand)
      } else if (half_det >= 0.0) {  // e2 stands alone (e0, e1 may be close)
        ok = evec_first(e2, vc);
        evec_second(vc, e1, vb);
        va[0] = vb[1] * vc[2] - vb[2] * vc[1]; va[1] = vb[2] * vc[0] - vb[0] * vc[2]; va[2] = vb[0] * vc[1] - vb[1] * vc[0];
      } else {                // e0 stands alone
        ok = evec_first(e0, va);
        evec_second(va, e1, vb);
        vc[0] = va[1] * vb[2] - va[2] * vb[1]; vc[1] = va[2] * vb[0] - va[0] * vb[2]; vc[2] = va[0] * vb[1] - va[1] * vb[0];
      }
      // Rayleigh quotients on the unscaled matrix, then the check
      auto rq = [&](const double (&v)[3], double & res) {
        const double Av0 = a00 * v[0] + a01 * v[1] + a02 * v[2], Av1 = a01 * v[0] + a11 * v[1] + a12 * v[2], Av2 = a02 * v[0] + a12 * v[1] + a22 * v[2];
        const double lam = v[0] * Av0 + v[1] * Av1 + v[2] * Av2;
        const double r0 = Av0 - lam * v[0], r1 = Av1 - lam * v[1], r2 = Av2 - lam * v[2];
        res = fmax(fabs(r0), fmax(fabs(r1), fabs(r2)));
        return lam;
      };
      double ra, rb, rc;
      const double la = rq(va, ra), lb = rq(vb, rb), lc = rq(vc, rc);
      ok = ok && fmax(ra, fmax(rb, rc)) <= 1e-13 * mx && la <= lb && lb <= lc;
      if (ok) {
        w[0] = la; w[1] = lb; w[2] = lc;
        V[0] = va[0]; V[1] = vb[0]; V[2] = vc[0];
        V[3] = va[1]; V[4] = vb[1]; V[5] = vc[1];
        V[6] = va[2]; V[7] = vb[2]; V[8] = vc[2];
      }
    }
  }
  if (!ok) sym_eigen3_jacobi(A, w, V);
}

// Eigenvectors ONLY (columns of V, eigenvalues ascending), for K4's projections: the decomposition sits on the critical path of
// the component pass (one lane per 3 x 3 block, a workgroup waiting), and K4 never looks at the eigenvalues (the host derives the
// localizabilities from the sums itself).  Works on D = (A - q I) / p (trace 0, |D|_F^2 = 6: no separate scaling), eigenvalues as in
// sym_eigen3, all three eigenvectors as cross products of rows of D - beta I; accepted only if they are orthogonal to 1e-10 and
// |D v - beta v| <= 2e-13 — otherwise false, and the caller runs sym_eigen3 (Eberly's construction, Jacobi behind it).
MH_HD bool sym_eigvec3_fast(const double A[9], double V[9])
{
  const double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[4], a12 = A[5], a22 = A[8];
  const double q = (a00 + a11 + a22) * (1.0 / 3.0);
  const double c00 = a00 - q, c11 = a11 - q, c22 = a22 - q;
  const double p2 = (c00 * c00 + c11 * c11 + c22 * c22 + 2.0 * (a01 * a01 + a02 * a02 + a12 * a12)) * (1.0 / 6.0);
  if (!(p2 > 1e-280) || !(p2 < 1e280)) return false;  // (also NaN, and a multiple of the identity)
  const double ip = mh_rsqrt(p2);
  if (!(p2 * ip * ip > 0.999999)) return false;
  const double d00 = c00 * ip, d01 = a01 * ip, d02 = a02 * ip, d11 = c11 * ip, d12 = a12 * ip, d22 = c22 * ip;
  double half_det = 0.5 * (d00 * (d11 * d22 - d12 * d12) - d01 * (d01 * d22 - d12 * d02) + d02 * (d01 * d12 - d11 * d02));
  half_det = fmin(fmax(half_det, -1.0), 1.0);
  const double dd = 2.0 * half_det;
  double br = 1.7320508075688772 + 0.2679491924311228 * fabs(half_det);
  br = half_det >= 0.0 ? br : -br;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int it = 0; it < 4; ++it) {
    const double b2 = br * br;
    br -= ((b2 - 3.0) * br - dd) * mh_rcp1(3.0 * b2 - 3.0);
  }
  const double sq = sqrt(fmax(12.0 - 3.0 * br * br, 0.0));
  const double blo = 0.5 * (-br - sq), bhi = 0.5 * (-br + sq);
  const double beta[3] = {half_det >= 0.0 ? blo : br, half_det >= 0.0 ? bhi : blo, half_det >= 0.0 ? br : bhi};
  double v[3][3];
  bool ok = true;
  double worst = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < 3; ++k) {
    const double m00 = d00 - beta[k], m11 = d11 - beta[k], m22 = d22 - beta[k];
    const double x0 = d01 * d12 - d02 * m11, y0 = d02 * d01 - m00 * d12, z0 = m00 * m11 - d01 * d01;
    const double x1 = d01 * m22 - d02 * d12, y1 = d02 * d02 - m00 * m22, z1 = m00 * d12 - d01 * d02;
    const double x2 = m11 * m22 - d12 * d12, y2 = d12 * d02 - d01 * m22, z2 = d01 * d12 - m11 * d02;
    const double n0 = x0 * x0 + y0 * y0 + z0 * z0, n1 = x1 * x1 + y1 * y1 + z1 * z1, n2 = x2 * x2 + y2 * y2 + z2 * z2;
    double vx = x0, vy = y0, vz = z0, nn = n0;
    if (n1 > nn) { vx = x1; vy = y1; vz = z1; nn = n1; }
    if (n2 > nn) { vx = x2; vy = y2; vz = z2; nn = n2; }
    ok = ok && nn > 0.0;
    const double r = nn > 0.0 ? mh_rsqrt(nn) : 0.0;
    vx *= r; vy *= r; vz *= r;
    v[k][0] = vx; v[k][1] = vy; v[k][2] = vz;
    const double r0 = m00 * vx + d01 * vy + d02 * vz, r1 = d01 * vx + m11 * vy + d12 * vz, r2 = d02 * vx + d12 * vy + m22 * vz;
    worst = fmax(worst, fmax(fabs(r0), fmax(fabs(r1), fabs(r2))));
  }
  const double dab = v[0][0] * v[1][0] + v[0][1] * v[1][1] + v[0][2] * v[1][2], dac = v[0][0] * v[2][0] + v[0][1] * v[2][1] + v[0][2] * v[2][2],
               dbc = v[1][0] * v[2][0] + v[1][1] * v[2][1] + v[1][2] * v[2][2];
  ok = ok && worst <= 2e-13 && fmax(fabs(dab), fmax(fabs(dac), fabs(dbc))) <= 1e-10;
  if (!ok) return false;
  for (int k = 0; k < 3; ++k) {
    V[k] = v[k][0];
    V[3 + k] = v[k][1];
    V[6 + k] = v[k][2];
  }
  return true;
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
