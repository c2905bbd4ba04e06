// Small dense double-precision routines that stay on the host, as in the reference (they run once per solve / per
// new keyframe on K x K matrices, K <= 128): pseudo-inverses for the pose covariance
// (PROB_SRC/eigen_photometric_bundle_adjustment.cpp:31-45), Schur elimination of marginalised frames
// (NormalLinearSystem::reduce_system, PROB_SRC/normal_linear_system.cpp:19-50).  The reference uses Eigen's JacobiSVD /
// completeOrthogonalDecomposition; for the symmetric matrices involved both equal the spectral pseudo-inverse computed
// here with a cyclic Jacobi eigen-solver.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace dsopp_hip {
namespace hostla {

using Mat = std::vector<double>;  // row-major n x n

/** A = Q diag(w) Q^T for symmetric A (cyclic Jacobi rotations) */
inline void symmetricEigen(const Mat &Ain, int n, std::vector<double> &w, Mat &Q) {
  Mat A = Ain;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) A[i * n + j] = A[j * n + i] = 0.5 * (A[i * n + j] + A[j * n + i]);
  Q.assign(static_cast<size_t>(n) * n, 0.0);
  for (int i = 0; i < n; ++i) Q[i * n + i] = 1;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; ++i) {
      diag += A[i * n + i] * A[i * n + i];
      for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    }
    if (off <= 1e-60 || off <= 1e-34 * diag) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p * n + q];
        if (std::abs(apq) < 1e-300) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1));
        const double c = 1 / std::sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < n; ++k) {
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq;
          A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk;
          A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double qkp = Q[k * n + p], qkq = Q[k * n + q];
          Q[k * n + p] = c * qkp - s * qkq;
          Q[k * n + q] = s * qkp + c * qkq;
        }
      }
  }
  w.resize(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) w[static_cast<size_t>(i)] = A[i * n + i];
}

/** spectral pseudo-inverse keeping the eigenpairs selected by `keep(e)` */
template <typename Keep>
Mat spectralPinv(const std::vector<double> &w, const Mat &Q, int n, Keep keep) {
  Mat R(static_cast<size_t>(n) * n, 0.0);
  for (int e = 0; e < n; ++e) {
    if (!keep(e)) continue;
    const double inv = 1.0 / w[static_cast<size_t>(e)];
    for (int i = 0; i < n; ++i) {
      const double qi = Q[i * n + e] * inv;
      for (int j = 0; j < n; ++j) R[i * n + j] += qi * Q[j * n + e];
    }
  }
  return R;
}

/** pseudoInverse(origin, number_of_nullspaces): invert all but the smallest `nullspaces` singular values */
inline Mat pinvDropSmallest(const Mat &H, int n, int nullspaces) {
  std::vector<double> w;
  Mat Q;
  symmetricEigen(H, n, w, Q);
  std::vector<int> order(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) order[static_cast<size_t>(i)] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return std::abs(w[static_cast<size_t>(a)]) > std::abs(w[static_cast<size_t>(b)]); });
  std::vector<char> keep(static_cast<size_t>(n), 0);
  for (int s = 0; s < n - nullspaces; ++s) keep[static_cast<size_t>(order[static_cast<size_t>(s)])] = 1;
  return spectralPinv(w, Q, n, [&](int e) { return keep[static_cast<size_t>(e)] != 0; });
}

/** rank-revealing pseudo-inverse with Eigen's default threshold (epsilon * size * largest pivot) */
inline Mat pinvRankRevealing(const Mat &H, int n) {
  std::vector<double> w;
  Mat Q;
  symmetricEigen(H, n, w, Q);
  double wmax = 0;
  for (double v : w) wmax = std::max(wmax, std::abs(v));
  const double thr = std::numeric_limits<double>::epsilon() * n * wmax;
  return spectralPinv(w, Q, n, [&](int e) { return std::abs(w[static_cast<size_t>(e)]) > thr; });
}

/** NormalLinearSystem::reduce_system: eliminates `elim` (sorted) from (H, b) of size n; result has size n - |elim| */
inline void reduceSystem(Mat &H, std::vector<double> &b, int n, const std::vector<int> &elim) {
  std::vector<char> is_elim(static_cast<size_t>(n), 0);
  for (int i : elim) is_elim[static_cast<size_t>(i)] = 1;
  std::vector<int> keep;
  for (int i = 0; i < n; ++i)
    if (!is_elim[static_cast<size_t>(i)]) keep.push_back(i);
  const int nk = static_cast<int>(keep.size()), ne = static_cast<int>(elim.size());
  std::vector<double> p(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) p[static_cast<size_t>(i)] = 1.0 / std::sqrt(H[i * n + i] + 10.0);
  auto hp = [&](int i, int j) { return p[static_cast<size_t>(i)] * H[i * n + j] * p[static_cast<size_t>(j)]; };
  Mat Hee(static_cast<size_t>(ne) * ne), Hke(static_cast<size_t>(nk) * ne), Hkk(static_cast<size_t>(nk) * nk);
  for (int i = 0; i < ne; ++i)
    for (int j = 0; j < ne; ++j) Hee[i * ne + j] = hp(elim[static_cast<size_t>(i)], elim[static_cast<size_t>(j)]);
  for (int i = 0; i < nk; ++i) {
    for (int j = 0; j < ne; ++j) Hke[i * ne + j] = hp(keep[static_cast<size_t>(i)], elim[static_cast<size_t>(j)]);
    for (int j = 0; j < nk; ++j) Hkk[i * nk + j] = hp(keep[static_cast<size_t>(i)], keep[static_cast<size_t>(j)]);
  }
  const Mat Hee_inv = pinvRankRevealing(Hee, ne);
  Mat ST(static_cast<size_t>(nk) * ne, 0.0);  // schur_transform = Hke * pinv(Hee)
  for (int i = 0; i < nk; ++i)
    for (int k = 0; k < ne; ++k) {
      const double a = Hke[i * ne + k];
      if (a == 0) continue;
      for (int j = 0; j < ne; ++j) ST[i * ne + j] += a * Hee_inv[k * ne + j];
    }
  std::vector<double> bk(static_cast<size_t>(nk)), be(static_cast<size_t>(ne));
  for (int i = 0; i < nk; ++i) bk[static_cast<size_t>(i)] = p[static_cast<size_t>(keep[static_cast<size_t>(i)])] * b[static_cast<size_t>(keep[static_cast<size_t>(i)])];
  for (int i = 0; i < ne; ++i) be[static_cast<size_t>(i)] = p[static_cast<size_t>(elim[static_cast<size_t>(i)])] * b[static_cast<size_t>(elim[static_cast<size_t>(i)])];
  for (int i = 0; i < nk; ++i) {
    double sb = 0;
    for (int k = 0; k < ne; ++k) sb += ST[i * ne + k] * be[static_cast<size_t>(k)];
    bk[static_cast<size_t>(i)] -= sb;
    for (int j = 0; j < nk; ++j) {
      double s = 0;
      for (int k = 0; k < ne; ++k) s += ST[i * ne + k] * Hke[j * ne + k];
      Hkk[i * nk + j] -= s;
    }
  }
  Mat Hn(static_cast<size_t>(nk) * nk);
  std::vector<double> bn(static_cast<size_t>(nk));
  for (int i = 0; i < nk; ++i) {
    const double pi = 1.0 / p[static_cast<size_t>(keep[static_cast<size_t>(i)])];
    bn[static_cast<size_t>(i)] = pi * bk[static_cast<size_t>(i)];
    for (int j = 0; j < nk; ++j) {
      const double pj = 1.0 / p[static_cast<size_t>(keep[static_cast<size_t>(j)])];
      Hn[i * nk + j] = pi * (0.5 * (Hkk[i * nk + j] + Hkk[j * nk + i])) * pj;
    }
  }
  H.swap(Hn);
  b.swap(bn);
}

}  // namespace hostla
}  // namespace dsopp_hip
