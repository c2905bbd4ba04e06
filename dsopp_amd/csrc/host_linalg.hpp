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

/** A = Q diag(w) Q^T for symmetric A (cyclic Jacobi rotations; chosen over tridiagonal QL because the systems are graded
 *  over 16 decades — 1e16 fixed-frame prior next to O(1..1e6) photometric terms — and Jacobi keeps the small eigenvalues
 *  to relative accuracy).  Works on the upper triangle + its mirror with contiguous row updates: a rotation costs two
 *  row-pair passes (A and the eigenvector rows) instead of three strided ones. */
inline void symmetricEigen(const Mat &Ain, int n, std::vector<double> &w, Mat &Q) {
  Mat A = Ain;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) A[i * n + j] = A[j * n + i] = 0.5 * (A[i * n + j] + A[j * n + i]);
  Mat Qt(static_cast<size_t>(n) * n, 0.0);  // rows = eigenvectors
  for (int i = 0; i < n; ++i) Qt[i * n + i] = 1;
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
        const double app = A[p * n + p], aqq = A[q * n + q];
        // an off-diagonal entry below eps * sqrt(|a_pp a_qq|) no longer moves either eigenvalue in double precision
        // (de Rijk's criterion for relative accuracy on graded matrices): skip the rotation
        if (std::abs(apq) < 1e-300 || std::abs(apq) <= 1e-17 * std::sqrt(std::abs(app * aqq))) continue;
        const double theta = (aqq - app) / (2 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1));
        const double c = 1 / std::sqrt(t * t + 1), sn = t * c;
        double *rp = &A[static_cast<size_t>(p) * n], *rq = &A[static_cast<size_t>(q) * n];
        for (int k = 0; k < n; ++k) {  // rows p, q of J^T A (for k outside {p, q} these are also the final values)
          const double apk = rp[k], aqk = rq[k];
          rp[k] = c * apk - sn * aqk;
          rq[k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {  // mirror into columns p, q
          A[static_cast<size_t>(k) * n + p] = rp[k];
          A[static_cast<size_t>(k) * n + q] = rq[k];
        }
        // the 2 x 2 pivot block of J^T A J in closed form
        rp[p] = app - t * apq;
        rq[q] = aqq + t * apq;
        rp[q] = rq[p] = 0;
        double *qp = &Qt[static_cast<size_t>(p) * n], *qq = &Qt[static_cast<size_t>(q) * n];
        for (int k = 0; k < n; ++k) {
          const double a = qp[k], b = qq[k];
          qp[k] = c * a - sn * b;
          qq[k] = sn * a + c * b;
        }
      }
  }
  w.resize(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) w[static_cast<size_t>(i)] = A[i * n + i];
  Q.assign(static_cast<size_t>(n) * n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) Q[i * n + j] = Qt[j * n + i];
}

/** spectral pseudo-inverse keeping the eigenpairs selected by `keep(e)` */
template <typename Keep>
Mat spectralPinv(const std::vector<double> &w, const Mat &Q, int n, Keep keep) {
  Mat R(static_cast<size_t>(n) * n, 0.0);
  std::vector<double> col(static_cast<size_t>(n));
  for (int e = 0; e < n; ++e) {
    if (!keep(e)) continue;
    const double inv = 1.0 / w[static_cast<size_t>(e)];
    for (int j = 0; j < n; ++j) col[static_cast<size_t>(j)] = Q[j * n + e];  // eigenvector e, gathered once
    for (int i = 0; i < n; ++i) {
      const double qi = col[static_cast<size_t>(i)] * inv;
      double *row = &R[static_cast<size_t>(i) * n];
      for (int j = 0; j < n; ++j) row[j] += qi * col[static_cast<size_t>(j)];
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
