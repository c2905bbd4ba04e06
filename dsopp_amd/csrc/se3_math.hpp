// SE3 arithmetic shared by host and device code of the HIP library (double precision).
// The reference delegates this to Sophus @593db47 through energy::motion::SE3
// (src/energy/motion/include/energy/motion/se3_motion.hpp:16-253): tangent = (translation, rotation),
// exp: R = Exp(omega), t = V(omega) upsilon; Adj = [[R, hat(t) R], [0, R]]; storage (qx, qy, qz, qw, t).
// Poses are kept as rotation matrix + translation here; conversion to/from the quaternion form happens at the C-ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

namespace dsopp_hip {

#define DSOPP_HD __host__ __device__ inline

struct Rigid {
  double R[9];  // row-major
  double t[3];
};

DSOPP_HD Rigid rigidIdentity() {
  Rigid T;
#pragma unroll
  for (int i = 0; i < 9; ++i) T.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
  T.t[0] = T.t[1] = T.t[2] = 0;
  return T;
}

DSOPP_HD Rigid rigidFromParams(const double *p) {
  Rigid T;
  const double x = p[0], y = p[1], z = p[2], w = p[3];
  T.R[0] = 1 - 2 * (y * y + z * z);
  T.R[1] = 2 * (x * y - z * w);
  T.R[2] = 2 * (x * z + y * w);
  T.R[3] = 2 * (x * y + z * w);
  T.R[4] = 1 - 2 * (x * x + z * z);
  T.R[5] = 2 * (y * z - x * w);
  T.R[6] = 2 * (x * z - y * w);
  T.R[7] = 2 * (y * z + x * w);
  T.R[8] = 1 - 2 * (x * x + y * y);
  T.t[0] = p[4];
  T.t[1] = p[5];
  T.t[2] = p[6];
  return T;
}

/** rotation matrix -> unit quaternion (x, y, z, w), w >= 0 branch-stable */
DSOPP_HD void rigidToParams(const Rigid &T, double *p) {
  const double *R = T.R;
  const double tr = R[0] + R[4] + R[8];
  double x, y, z, w;
  if (tr > 0) {
    const double s = sqrt(tr + 1.0) * 2;
    w = 0.25 * s;
    x = (R[7] - R[5]) / s;
    y = (R[2] - R[6]) / s;
    z = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
    w = (R[7] - R[5]) / s;
    x = 0.25 * s;
    y = (R[1] + R[3]) / s;
    z = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
    w = (R[2] - R[6]) / s;
    x = (R[1] + R[3]) / s;
    y = 0.25 * s;
    z = (R[5] + R[7]) / s;
  } else {
    const double s = sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
    w = (R[3] - R[1]) / s;
    x = (R[2] + R[6]) / s;
    y = (R[5] + R[7]) / s;
    z = 0.25 * s;
  }
  const double n = sqrt(x * x + y * y + z * z + w * w);
  p[0] = x / n;
  p[1] = y / n;
  p[2] = z / n;
  p[3] = w / n;
  p[4] = T.t[0];
  p[5] = T.t[1];
  p[6] = T.t[2];
}

/** exp of the twist (upsilon, omega) */
DSOPP_HD Rigid rigidExp(const double *xi) {
  Rigid T;
  const double wx = xi[3], wy = xi[4], wz = xi[5];
  const double th2 = wx * wx + wy * wy + wz * wz;
  double A, B, Cc;  // R = I + A*W + B*W^2 ; V = I + B*W + Cc*W^2
  if (th2 < 0.25) {
    // |omega| < 0.5 rad (always the case for the increments of a solve): even Taylor series in th2, truncation < 1e-17;
    // avoids the large-argument range reduction of the libm sin/cos on the device
    A = 1.0 + th2 * (-1.0 / 6 + th2 * (1.0 / 120 + th2 * (-1.0 / 5040 + th2 * (1.0 / 362880 + th2 * (-1.0 / 39916800 + th2 * (1.0 / 6227020800.0 + th2 * (-1.0 / 1307674368000.0)))))));
    B = 0.5 + th2 * (-1.0 / 24 + th2 * (1.0 / 720 + th2 * (-1.0 / 40320 + th2 * (1.0 / 3628800 + th2 * (-1.0 / 479001600 + th2 * (1.0 / 87178291200.0 + th2 * (-1.0 / 20922789888000.0)))))));
    Cc = 1.0 / 6 + th2 * (-1.0 / 120 + th2 * (1.0 / 5040 + th2 * (-1.0 / 362880 + th2 * (1.0 / 39916800 + th2 * (-1.0 / 6227020800.0 + th2 * (1.0 / 1307674368000.0 + th2 * (-1.0 / 355687428096000.0)))))));
  } else {
    const double th = sqrt(th2);
    A = sin(th) / th;
    B = (1 - cos(th)) / th2;
    Cc = (th - sin(th)) / (th2 * th);
  }
  // W = hat(w); W^2 = w w^T - th2 I
  const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
  const double W2[9] = {wx * wx - th2, wx * wy, wx * wz, wx * wy, wy * wy - th2, wy * wz, wx * wz, wy * wz, wz * wz - th2};
  double V[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const double id = (i % 4 == 0) ? 1.0 : 0.0;
    T.R[i] = id + A * W[i] + B * W2[i];
    V[i] = id + B * W[i] + Cc * W2[i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) T.t[i] = V[3 * i] * xi[0] + V[3 * i + 1] * xi[1] + V[3 * i + 2] * xi[2];
  return T;
}

/** log of a rigid transform, tangent (upsilon, omega): omega from the unit quaternion (2 atan(|v| / w) / |v| * v, series for
 *  small |v|), upsilon = V^-1 t with V^-1 = I - W/2 + (1 - th cos(th/2) / (2 sin(th/2))) / th^2 W^2 — the inverse of rigidExp
 *  (reference: SE3::log of Sophus through se3_motion.hpp, call site monocular_tracker.cpp:153) */
DSOPP_HD void rigidLog(const Rigid &T, double *xi) {
  double p[7];
  rigidToParams(T, p);
  const double kEps = 1e-10;
  const double squared_n = p[0] * p[0] + p[1] * p[1] + p[2] * p[2], w = p[3];
  double k;
  if (squared_n < kEps * kEps) {
    k = 2.0 / w - (2.0 / 3.0) * squared_n / (w * w * w);
  } else {
    const double n = sqrt(squared_n);
    k = fabs(w) < kEps ? (w > 0 ? 3.14159265358979323846 : -3.14159265358979323846) / n : 2.0 * atan(n / w) / n;
  }
  const double wx = k * p[0], wy = k * p[1], wz = k * p[2];
  const double th2 = wx * wx + wy * wy + wz * wz, th = sqrt(th2);
  const double c = th < kEps ? 1.0 / 12.0 : (1.0 - 0.5 * th * cos(0.5 * th) / sin(0.5 * th)) / th2;
  const double *t = T.t;
  const double c1x = wy * t[2] - wz * t[1], c1y = wz * t[0] - wx * t[2], c1z = wx * t[1] - wy * t[0];
  const double c2x = wy * c1z - wz * c1y, c2y = wz * c1x - wx * c1z, c2z = wx * c1y - wy * c1x;
  xi[0] = t[0] - 0.5 * c1x + c * c2x;
  xi[1] = t[1] - 0.5 * c1y + c * c2y;
  xi[2] = t[2] - 0.5 * c1z + c * c2z;
  xi[3] = wx;
  xi[4] = wy;
  xi[5] = wz;
}

DSOPP_HD Rigid rigidMul(const Rigid &a, const Rigid &b) {
  Rigid c;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) c.R[3 * i + j] = a.R[3 * i] * b.R[j] + a.R[3 * i + 1] * b.R[3 + j] + a.R[3 * i + 2] * b.R[6 + j];
    c.t[i] = a.R[3 * i] * b.t[0] + a.R[3 * i + 1] * b.t[1] + a.R[3 * i + 2] * b.t[2] + a.t[i];
  }
  return c;
}

DSOPP_HD Rigid rigidInverse(const Rigid &a) {
  Rigid c;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) c.R[3 * i + j] = a.R[3 * j + i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) c.t[i] = -(c.R[3 * i] * a.t[0] + c.R[3 * i + 1] * a.t[1] + c.R[3 * i + 2] * a.t[2]);
  return c;
}

/** re-orthonormalise a rotation that accumulated products (the reference's quaternion form renormalises on multiply) */
DSOPP_HD void rigidNormalize(Rigid &T) {
  double p[7];
  rigidToParams(T, p);
  T = rigidFromParams(p);
}

/** the rotation closest to a 3x3 matrix with positive determinant (SO3::fitToSO3 of Sophus, reached through
 *  SE3::setRotationMatrix, se3_motion.hpp:215): Newton iteration of the polar decomposition, R <- (R + R^-T) / 2 */
DSOPP_HD void fitToSO3(const double *Rin, double *R) {
  for (int i = 0; i < 9; ++i) R[i] = Rin[i];
  for (int it = 0; it < 30; ++it) {
    const double c[9] = {R[4] * R[8] - R[5] * R[7], R[5] * R[6] - R[3] * R[8], R[3] * R[7] - R[4] * R[6],
                         R[2] * R[7] - R[1] * R[8], R[0] * R[8] - R[2] * R[6], R[1] * R[6] - R[0] * R[7],
                         R[1] * R[5] - R[2] * R[4], R[2] * R[3] - R[0] * R[5], R[0] * R[4] - R[1] * R[3]};  // cofactors = det * R^-T
    const double inv_det = 1.0 / (R[0] * c[0] + R[1] * c[1] + R[2] * c[2]);
    double delta = 0;
    for (int i = 0; i < 9; ++i) {
      const double n = 0.5 * (R[i] + c[i] * inv_det);
      delta = fmax(delta, fabs(n - R[i]));
      R[i] = n;
    }
    if (delta < 1e-16) break;
  }
}

/** Adjoint, row-major 6x6 */
DSOPP_HD void rigidAdj(const Rigid &T, double *A) {
  const double *t = T.t;
  const double hx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
#pragma unroll
  for (int i = 0; i < 36; ++i) A[i] = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      A[6 * i + j] = T.R[3 * i + j];
      A[6 * (i + 3) + (j + 3)] = T.R[3 * i + j];
      A[6 * i + (j + 3)] = hx[3 * i] * T.R[j] + hx[3 * i + 1] * T.R[3 + j] + hx[3 * i + 2] * T.R[6 + j];
    }
}

}  // namespace dsopp_hip
