// rgc_lm.h -- the scalar pieces of LsqRegistration::step_lm (lsq_registration_impl.hpp:125-172) shared by the host
// driver and the single-lane device kernel that performs the first LM try of every outer iteration.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace rgclm {

// so3_exp (so3/so3.hpp:58-77) followed by Eigen's Quaterniond::toRotationMatrix()
__host__ __device__ inline void so3_exp_R(const double w[3], double R[9]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
  double imag, real;
  if (th2 < 1e-10) {
    const double th4 = th2 * th2;
    imag = 0.5 - 1.0 / 48.0 * th2 + 1.0 / 3840.0 * th4;
    real = 1.0 - 1.0 / 8.0 * th2 + 1.0 / 384.0 * th4;
#ifdef __HIP_DEVICE_COMPILE__
  } else if (th2 < 0.25) {
    // The device's one deciding lane: sin(th/2)/th and cos(th/2) as series in u = (th/2)^2 (an LM step of a scan-to-map registration
    // turns by well under half a radian) -- fourteen dependent fma instead of a square root, a division and libm's sin and cos with their
    // range reduction (~300 dependent fp64 instructions).  Truncation below 1e-19 relative at u = 1/16: the same double as the closed
    // form up to its own rounding.  (The host driver keeps the closed form: the two agree to 1e-15, tests/test_gpu_parity.py.)
    const double u = 0.25 * th2;
    double sc = -1.0 / 1307674368000.0;                       // sin(x)/x = sum (-u)^k / (2k+1)!
    sc = fma(sc, u, 1.0 / 6227020800.0); sc = fma(sc, u, -1.0 / 39916800.0); sc = fma(sc, u, 1.0 / 362880.0);
    sc = fma(sc, u, -1.0 / 5040.0); sc = fma(sc, u, 1.0 / 120.0); sc = fma(sc, u, -1.0 / 6.0); sc = fma(sc, u, 1.0);
    double cs = 1.0 / 87178291200.0;                          // cos(x) = sum (-u)^k / (2k)!
    cs = fma(cs, u, -1.0 / 479001600.0); cs = fma(cs, u, 1.0 / 3628800.0); cs = fma(cs, u, -1.0 / 40320.0);
    cs = fma(cs, u, 1.0 / 720.0); cs = fma(cs, u, -1.0 / 24.0); cs = fma(cs, u, 0.5); cs = fma(-cs, u, 1.0);
    imag = 0.5 * sc;
    real = cs;
#endif
  } else {
    const double th = sqrt(th2), half = 0.5 * th;
    imag = sin(half) / th;
    real = cos(half);
  }
  const double qw = real, qx = imag * w[0], qy = imag * w[1], qz = imag * w[2];
  const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
  const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx, tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

// (H + lambda I) d = -b, symmetric 6x6 (Eigen::LDLT at lsq_registration_impl.hpp:136-137).  LDL^T without pivoting:
// every index is static after unrolling, so the single device lane keeps the factor in registers (a pivoted version
// lives in scratch memory and costs ~10 us per call).  H + lambda I is positive definite whenever there are
// correspondences; a zero pivot reports failure (d = NaN upstream, like a failed Eigen solve would propagate).
__host__ __device__ inline bool solve_ldlt6(const double Ain[36], const double rhs[6], double x[6]) {
  double L[6][6], D[6], invD[6];  // one division per pivot (an fp64 division is ~40 dependent instructions on the device lane)
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 6; k++) {
    double dk = Ain[k * 6 + k];
#pragma unroll
    for (int j = 0; j < 6; j++)
      if (j < k) dk -= L[k][j] * L[k][j] * D[j];
    D[k] = dk;
    if (dk == 0.0 || !(fabs(dk) < 1.0e300)) ok = false;
#ifdef __HIP_DEVICE_COMPILE__
    {  // v_rcp_f64 and two Newton steps (five dependent instructions; the compiler's IEEE division is ~40): within an ulp or two of 1 / dk
      double r = __builtin_amdgcn_rcp(dk);
      r = fma(fma(-dk, r, 1.0), r, r);
      r = fma(fma(-dk, r, 1.0), r, r);
      invD[k] = r;
    }
#else
    invD[k] = 1.0 / dk;
#endif
#pragma unroll
    for (int i = 0; i < 6; i++) {
      if (i > k) {
        double v = Ain[i * 6 + k];
#pragma unroll
        for (int j = 0; j < 6; j++)
          if (j < k) v -= L[i][j] * L[k][j] * D[j];
        L[i][k] = v * invD[k];
      }
    }
  }
  if (!ok) return false;
  double y[6];
#pragma unroll
  for (int i = 0; i < 6; i++) {
    double s = rhs[i];
#pragma unroll
    for (int j = 0; j < 6; j++)
      if (j < i) s -= L[i][j] * y[j];
    y[i] = s;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) y[i] *= invD[i];
#pragma unroll
  for (int i = 5; i >= 0; i--) {
    double s = y[i];
#pragma unroll
    for (int j = 0; j < 6; j++)
      if (j > i) s -= L[j][i] * x[j];
    x[i] = s;
  }
  return true;
}

__host__ __device__ inline void mul4(const double A[16], const double B[16], double C[16]) {
  double t[16];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j];
      t[i * 4 + j] = s;
    }
#pragma unroll
  for (int i = 0; i < 16; i++) C[i] = t[i];
}

// one LM try: d = solve(H + lambda I, -b); delta = [so3_exp(d[0:3]) | d[3:6]]; xi = delta * x0   (:136-143)
// H is the full symmetric 6x6.  d = NaN if the solve fails (like a failed Eigen LDLT propagates).
__host__ __device__ inline void lm_try(const double H[36], const double b[6], double lambda, const double x0[16], double d[6],
                                       double delta[16], double xi[16]) {
  double A[36], nb[6];
#pragma unroll
  for (int i = 0; i < 36; i++) A[i] = H[i];
#pragma unroll
  for (int i = 0; i < 6; i++) { A[i * 7] += lambda; nb[i] = -b[i]; }
  if (!solve_ldlt6(A, nb, d)) {
#pragma unroll
    for (int i = 0; i < 6; i++) d[i] = NAN;
  }
  double R[9];
  so3_exp_R(d, R);
#pragma unroll
  for (int i = 0; i < 16; i++) delta[i] = 0.0;
#pragma unroll
  for (int a = 0; a < 3; a++) {
#pragma unroll
    for (int e = 0; e < 3; e++) delta[a * 4 + e] = R[a * 3 + e];
    delta[a * 4 + 3] = d[3 + a];
  }
  delta[15] = 1.0;
  mul4(delta, x0, xi);
}

}  // namespace rgclm
