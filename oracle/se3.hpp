// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// SE3 restatement.  The reference's SE3 (src/energy/motion/include/energy/motion/se3_motion.hpp:16-253) is a thin
// wrapper over Sophus::SE3 @593db47500ea1a2de5f0e6579c86147991509c59 (cmake/modules/sophus.cmake:1-2), which is NOT
// under /root/reference.  PARITY UNPINNED for exp/Adj *values*: the conventions below are the published Sophus ones
//   tangent = (upsilon translation, omega rotation); exp: R = Exp(omega), t = V(omega) * upsilon;
//   Adj = [[R, hat(t) R], [0, R]]; storage = unit quaternion (x, y, z, w) followed by translation
// and are pinned by closed-form identity tests (tests/test_oracle_se3.py) — the reference's own tests at this boundary
// (test/test/energy/motion/se3_motion.cpp:51-140, test/test/energy/projector/test_reprojects.cpp:160,196-204)
// pin derivatives only.
#pragma once
#include <algorithm>
#include <cmath>

namespace oracle {

struct SE3 {
  double q[4] = {0, 0, 0, 1};  // x, y, z, w
  double t[3] = {0, 0, 0};

  static SE3 fromParams(const double *p) {
    SE3 s;
    for (int i = 0; i < 4; ++i) s.q[i] = p[i];
    for (int i = 0; i < 3; ++i) s.t[i] = p[4 + i];
    return s;
  }
  void toParams(double *p) const {
    for (int i = 0; i < 4; ++i) p[i] = q[i];
    for (int i = 0; i < 3; ++i) p[4 + i] = t[i];
  }
  /** rotation matrix, row-major 3x3 */
  void rotation(double R[9]) const {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    R[0] = 1 - 2 * (y * y + z * z);
    R[1] = 2 * (x * y - z * w);
    R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w);
    R[4] = 1 - 2 * (x * x + z * z);
    R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w);
    R[7] = 2 * (y * z + x * w);
    R[8] = 1 - 2 * (x * x + y * y);
  }
  /** Sophus::SE3::exp — tangent (upsilon, omega) */
  static SE3 exp(const double *xi) {
    SE3 s;
    const double wx = xi[3], wy = xi[4], wz = xi[5];
    const double theta_sq = wx * wx + wy * wy + wz * wz;
    const double theta = std::sqrt(theta_sq);
    double imag, real;
    const double kEps = 1e-10;
    if (theta_sq < kEps * kEps) {
      const double theta_po4 = theta_sq * theta_sq;
      imag = 0.5 - theta_sq / 48.0 + theta_po4 / 3840.0;
      real = 1.0 - theta_sq / 8.0 + theta_po4 / 384.0;
    } else {
      const double half = 0.5 * theta;
      imag = std::sin(half) / theta;
      real = std::cos(half);
    }
    s.q[0] = imag * wx;
    s.q[1] = imag * wy;
    s.q[2] = imag * wz;
    s.q[3] = real;
    // V = I + a*Omega + b*Omega^2
    double a, b;
    if (theta < kEps) {
      a = 0.5;
      b = 1.0 / 6.0;
    } else {
      a = (1 - std::cos(theta)) / theta_sq;
      b = (theta - std::sin(theta)) / (theta_sq * theta);
    }
    const double ux = xi[0], uy = xi[1], uz = xi[2];
    // Omega*u = w x u ; Omega^2*u = w x (w x u)
    const double c1x = wy * uz - wz * uy, c1y = wz * ux - wx * uz, c1z = wx * uy - wy * ux;
    const double c2x = wy * c1z - wz * c1y, c2y = wz * c1x - wx * c1z, c2z = wx * c1y - wy * c1x;
    s.t[0] = ux + a * c1x + b * c2x;
    s.t[1] = uy + a * c1y + b * c2y;
    s.t[2] = uz + a * c1z + b * c2z;
    return s;
  }
  /** SE3::log (Sophus @593db47, published algorithm): omega = SO3::log of the unit quaternion (atan-based, series below
   *  |vec| < 1e-10), upsilon = V^-1 t with V^-1 = I - Omega/2 + (1 - theta cos(theta/2) / (2 sin(theta/2))) / theta^2 Omega^2
   *  (1/12 for small theta).  Tangent order (upsilon, omega).  Call site: monocular_tracker.cpp:153. */
  void log(double xi[6]) const {
    const double kEps = 1e-10;
    const double squared_n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    const double w = q[3];
    double two_atan_nbyw_by_n;
    if (squared_n < kEps * kEps) {
      two_atan_nbyw_by_n = 2.0 / w - (2.0 / 3.0) * squared_n / (w * w * w);
    } else {
      const double n = std::sqrt(squared_n);
      if (std::abs(w) < kEps) {
        two_atan_nbyw_by_n = (w > 0 ? M_PI : -M_PI) / n;
      } else {
        two_atan_nbyw_by_n = 2.0 * std::atan(n / w) / n;
      }
    }
    const double wx = two_atan_nbyw_by_n * q[0], wy = two_atan_nbyw_by_n * q[1], wz = two_atan_nbyw_by_n * q[2];
    const double theta_sq = wx * wx + wy * wy + wz * wz, theta = std::sqrt(theta_sq);
    double c;
    if (theta < kEps) {
      c = 1.0 / 12.0;
    } else {
      const double half = 0.5 * theta;
      c = (1.0 - 0.5 * theta * std::cos(half) / std::sin(half)) / theta_sq;
    }
    // V^-1 t = t - (omega x t)/2 + c * omega x (omega x t)
    const double c1x = wy * t[2] - wz * t[1], c1y = wz * t[0] - wx * t[2], c1z = wx * t[1] - wy * t[0];
    const double c2x = wy * c1z - wz * c1y, c2y = wz * c1x - wx * c1z, c2z = wx * c1y - wy * c1x;
    xi[0] = t[0] - 0.5 * c1x + c * c2x;
    xi[1] = t[1] - 0.5 * c1y + c * c2y;
    xi[2] = t[2] - 0.5 * c1z + c * c2z;
    xi[3] = wx;
    xi[4] = wy;
    xi[5] = wz;
  }
  /** setRotationMatrix — se3_motion.hpp:215: so3() = SO3::fitToSO3(R), the rotation closest to R (Sophus: U diag(1, 1, det) V^T
   *  of the SVD; here the equivalent polar iteration R <- (R + R^-T) / 2, valid for det R > 0), stored as a unit quaternion */
  void setRotationMatrix(const double Rin[9]) {
    double R[9];
    for (int i = 0; i < 9; ++i) R[i] = Rin[i];
    for (int it = 0; it < 30; ++it) {
      const double c00 = R[4] * R[8] - R[5] * R[7], c01 = R[5] * R[6] - R[3] * R[8], c02 = R[3] * R[7] - R[4] * R[6];
      const double c10 = R[2] * R[7] - R[1] * R[8], c11 = R[0] * R[8] - R[2] * R[6], c12 = R[1] * R[6] - R[0] * R[7];
      const double c20 = R[1] * R[5] - R[2] * R[4], c21 = R[2] * R[3] - R[0] * R[5], c22 = R[0] * R[4] - R[1] * R[3];
      const double det = R[0] * c00 + R[1] * c01 + R[2] * c02;
      const double invT[9] = {c00 / det, c01 / det, c02 / det, c10 / det, c11 / det, c12 / det, c20 / det, c21 / det, c22 / det};  // R^-T
      double delta = 0;
      for (int i = 0; i < 9; ++i) {
        const double n = 0.5 * (R[i] + invT[i]);
        delta = std::max(delta, std::abs(n - R[i]));
        R[i] = n;
      }
      if (delta < 1e-16) break;
    }
    const double tr = R[0] + R[4] + R[8];
    double x, y, z, w;
    if (tr > 0) {
      const double sq = std::sqrt(tr + 1.0) * 2;
      w = 0.25 * sq, x = (R[7] - R[5]) / sq, y = (R[2] - R[6]) / sq, z = (R[3] - R[1]) / sq;
    } else if (R[0] > R[4] && R[0] > R[8]) {
      const double sq = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
      w = (R[7] - R[5]) / sq, x = 0.25 * sq, y = (R[1] + R[3]) / sq, z = (R[2] + R[6]) / sq;
    } else if (R[4] > R[8]) {
      const double sq = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
      w = (R[2] - R[6]) / sq, x = (R[1] + R[3]) / sq, y = 0.25 * sq, z = (R[5] + R[7]) / sq;
    } else {
      const double sq = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
      w = (R[3] - R[1]) / sq, x = (R[2] + R[6]) / sq, y = (R[5] + R[7]) / sq, z = 0.25 * sq;
    }
    const double n = std::sqrt(x * x + y * y + z * z + w * w);
    q[0] = x / n, q[1] = y / n, q[2] = z / n, q[3] = w / n;
  }
  SE3 inverse() const {
    SE3 s;
    s.q[0] = -q[0];
    s.q[1] = -q[1];
    s.q[2] = -q[2];
    s.q[3] = q[3];
    double R[9];
    s.rotation(R);
    for (int i = 0; i < 3; ++i) s.t[i] = -(R[3 * i] * t[0] + R[3 * i + 1] * t[1] + R[3 * i + 2] * t[2]);
    return s;
  }
  SE3 operator*(const SE3 &o) const {
    SE3 s;
    const double ax = q[0], ay = q[1], az = q[2], aw = q[3];
    const double bx = o.q[0], by = o.q[1], bz = o.q[2], bw = o.q[3];
    s.q[0] = aw * bx + ax * bw + ay * bz - az * by;
    s.q[1] = aw * by - ax * bz + ay * bw + az * bx;
    s.q[2] = aw * bz + ax * by - ay * bx + az * bw;
    s.q[3] = aw * bw - ax * bx - ay * by - az * bz;
    const double n = std::sqrt(s.q[0] * s.q[0] + s.q[1] * s.q[1] + s.q[2] * s.q[2] + s.q[3] * s.q[3]);
    for (double &v : s.q) v /= n;
    double R[9];
    rotation(R);
    for (int i = 0; i < 3; ++i) s.t[i] = R[3 * i] * o.t[0] + R[3 * i + 1] * o.t[1] + R[3 * i + 2] * o.t[2] + t[i];
    return s;
  }
  /** leftIncrement: exp(eps) * this — se3_motion.hpp:231 */
  SE3 leftIncrement(const double *eps) const { return SE3::exp(eps) * (*this); }
  /** rightIncrement: this * exp(eps) — se3_motion.hpp:239 */
  SE3 rightIncrement(const double *eps) const { return (*this) * SE3::exp(eps); }
  /** Adjoint, row-major 6x6: [[R, hat(t) R], [0, R]] — rightLogTransformer, se3_motion.hpp:245 */
  void Adj(double A[36]) const {
    double R[9];
    rotation(R);
    const double hx[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
    for (int i = 0; i < 36; ++i) A[i] = 0;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        A[6 * i + j] = R[3 * i + j];
        A[6 * (i + 3) + (j + 3)] = R[3 * i + j];
        double s = 0;
        for (int k = 0; k < 3; ++k) s += hx[3 * i + k] * R[3 * k + j];
        A[6 * i + (j + 3)] = s;
      }
  }
  /** 3x4 [R|t], row-major */
  void matrix3x4(double M[12]) const {
    double R[9];
    rotation(R);
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) M[4 * i + j] = R[3 * i + j];
      M[4 * i + 3] = t[i];
    }
  }
};

}  // namespace oracle
