// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
//
// Dense linear algebra restated for the CPU oracle (no Eigen in this image).
// PARITY UNPINNED at matrix level: the reference calls Eigen @1f4c0311 (cmake/modules/eigen.cmake:1-2),
// which is not vendored under /root/reference; the routines here restate the published algorithms
// (LDLT with symmetric diagonal pivoting, symmetric eigen-decomposition based pseudo-inverses) and
// are checked by identity tests (tests/test_oracle_linalg.py), not against Eigen outputs.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <limits>
#include <vector>

namespace oracle {

/** row-major dense matrix of doubles */
struct Mat {
  int r = 0, c = 0;
  std::vector<double> a;
  Mat() = default;
  Mat(int rows, int cols) : r(rows), c(cols), a(static_cast<size_t>(rows) * cols, 0.0) {}
  double &operator()(int i, int j) { return a[static_cast<size_t>(i) * c + j]; }
  double operator()(int i, int j) const { return a[static_cast<size_t>(i) * c + j]; }
  void setZero() { std::fill(a.begin(), a.end(), 0.0); }
  static Mat Identity(int n) {
    Mat m(n, n);
    for (int i = 0; i < n; ++i) m(i, i) = 1;
    return m;
  }
};
using Vec = std::vector<double>;

inline Mat matmul(const Mat &A, const Mat &B) {
  Mat C(A.r, B.c);
  for (int i = 0; i < A.r; ++i)
    for (int k = 0; k < A.c; ++k) {
      const double aik = A(i, k);
      if (aik == 0) continue;
      for (int j = 0; j < B.c; ++j) C(i, j) += aik * B(k, j);
    }
  return C;
}
inline Mat transpose(const Mat &A) {
  Mat T(A.c, A.r);
  for (int i = 0; i < A.r; ++i)
    for (int j = 0; j < A.c; ++j) T(j, i) = A(i, j);
  return T;
}
inline Vec matvec(const Mat &A, const Vec &x) {
  Vec y(static_cast<size_t>(A.r), 0.0);
  for (int i = 0; i < A.r; ++i) {
    double s = 0;
    for (int j = 0; j < A.c; ++j) s += A(i, j) * x[static_cast<size_t>(j)];
    y[static_cast<size_t>(i)] = s;
  }
  return y;
}
inline double dot(const Vec &a, const Vec &b) {
  double s = 0;
  for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
  return s;
}

/**
 * Robust Cholesky decomposition with symmetric pivoting, A = P^T L D L^T P, and solve.
 * Restates Eigen::LDLT (used by NormalLinearSystem::solve, src/energy/problems/src/normal_linear_system.cpp:52-59):
 * at step k the remaining diagonal entry of largest magnitude is moved to position k.
 */
inline Vec ldltSolve(Mat A, Vec b) {
  const int n = A.r;
  std::vector<int> transpositions(static_cast<size_t>(n));
  Vec temp(static_cast<size_t>(n));
  for (int k = 0; k < n; ++k) {
    int piv = k;
    double biggest = std::abs(A(k, k));
    for (int i = k + 1; i < n; ++i)
      if (std::abs(A(i, i)) > biggest) {
        biggest = std::abs(A(i, i));
        piv = i;
      }
    transpositions[static_cast<size_t>(k)] = piv;
    if (piv != k) {
      // symmetric swap on the lower triangle
      const int s = n - piv - 1;
      for (int j = 0; j < k; ++j) std::swap(A(k, j), A(piv, j));
      for (int i = 0; i < s; ++i) std::swap(A(piv + 1 + i, k), A(piv + 1 + i, piv));
      std::swap(A(k, k), A(piv, piv));
      for (int i = k + 1; i < piv; ++i) std::swap(A(i, k), A(piv, i));
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; ++j) temp[static_cast<size_t>(j)] = A(j, j) * A(k, j);
      double s = 0;
      for (int j = 0; j < k; ++j) s += A(k, j) * temp[static_cast<size_t>(j)];
      A(k, k) -= s;
      for (int i = 0; i < rs; ++i) {
        double t = 0;
        for (int j = 0; j < k; ++j) t += A(k + 1 + i, j) * temp[static_cast<size_t>(j)];
        A(k + 1 + i, k) -= t;
      }
    }
    const double akk = A(k, k);
    if (rs > 0 && std::abs(akk) > 0)
      for (int i = 0; i < rs; ++i) A(k + 1 + i, k) /= akk;
  }
  // solve: x = P^T L^-T D^-1 L^-1 P b
  for (int k = 0; k < n; ++k) std::swap(b[static_cast<size_t>(k)], b[static_cast<size_t>(transpositions[static_cast<size_t>(k)])]);
  for (int i = 0; i < n; ++i) {
    double s = b[static_cast<size_t>(i)];
    for (int j = 0; j < i; ++j) s -= A(i, j) * b[static_cast<size_t>(j)];
    b[static_cast<size_t>(i)] = s;
  }
  const double tol = std::numeric_limits<double>::min();
  for (int i = 0; i < n; ++i) {
    if (std::abs(A(i, i)) > tol)
      b[static_cast<size_t>(i)] /= A(i, i);
    else
      b[static_cast<size_t>(i)] = 0;
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[static_cast<size_t>(i)];
    for (int j = i + 1; j < n; ++j) s -= A(j, i) * b[static_cast<size_t>(j)];
    b[static_cast<size_t>(i)] = s;
  }
  for (int k = n - 1; k >= 0; --k) std::swap(b[static_cast<size_t>(k)], b[static_cast<size_t>(transpositions[static_cast<size_t>(k)])]);
  return b;
}

/**
 * Cyclic Jacobi eigen-decomposition of a symmetric matrix: A = Q diag(w) Q^T.
 * Used to restate the pseudo-inverses the reference obtains from Eigen::JacobiSVD
 * (eigen_photometric_bundle_adjustment.cpp:31-45) and completeOrthogonalDecomposition
 * (normal_linear_system.cpp:35-37, eigen_pose_alignment.cpp:320-323) — for the symmetric
 * matrices they are applied to, both equal Q diag(1/w_i) Q^T over the retained spectrum.
 */
inline void symmetricEigen(const Mat &Ain, Vec &w, Mat &Q) {
  const int n = Ain.r;
  Mat A = Ain;
  // symmetrise defensively
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      const double m = 0.5 * (A(i, j) + A(j, i));
      A(i, j) = A(j, i) = m;
    }
  Q = Mat::Identity(n);
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; ++i) {
      diag += A(i, i) * A(i, i);
      for (int j = i + 1; j < n; ++j) off += A(i, j) * A(i, j);
    }
    if (off <= 1e-60 || off <= 1e-34 * diag) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A(p, q);
        if (apq == 0) continue;
        const double app = A(p, p), aqq = A(q, q);
        if (std::abs(apq) < 1e-300) continue;
        const double theta = (aqq - app) / (2 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::abs(theta) + std::sqrt(theta * theta + 1));
        const double cs = 1 / std::sqrt(t * t + 1), sn = t * cs;
        for (int k = 0; k < n; ++k) {
          const double akp = A(k, p), akq = A(k, q);
          A(k, p) = cs * akp - sn * akq;
          A(k, q) = sn * akp + cs * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A(p, k), aqk = A(q, k);
          A(p, k) = cs * apk - sn * aqk;
          A(q, k) = sn * apk + cs * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double qkp = Q(k, p), qkq = Q(k, q);
          Q(k, p) = cs * qkp - sn * qkq;
          Q(k, q) = sn * qkp + cs * qkq;
        }
      }
  }
  w.resize(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) w[static_cast<size_t>(i)] = A(i, i);
}

/**
 * pseudoInverse(origin, number_of_nullspaces) — eigen_photometric_bundle_adjustment.cpp:31-45:
 * JacobiSVD, invert all but the `number_of_nullspaces` smallest singular values.
 */
inline Mat pseudoInverseDropSmallest(const Mat &H, int number_of_nullspaces) {
  const int n = H.r;
  Vec w;
  Mat Q;
  symmetricEigen(H, w, Q);
  std::vector<int> order(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) order[static_cast<size_t>(i)] = i;
  std::sort(order.begin(), order.end(),
            [&](int a, int b) { return std::abs(w[static_cast<size_t>(a)]) > std::abs(w[static_cast<size_t>(b)]); });
  Mat R(n, n);
  for (int s = 0; s < n - number_of_nullspaces; ++s) {
    const int e = order[static_cast<size_t>(s)];
    const double inv = 1.0 / w[static_cast<size_t>(e)];
    for (int i = 0; i < n; ++i) {
      const double qi = Q(i, e) * inv;
      for (int j = 0; j < n; ++j) R(i, j) += qi * Q(j, e);
    }
  }
  return R;
}

/**
 * completeOrthogonalDecomposition().pseudoInverse() of a symmetric matrix with Eigen's default
 * rank threshold (epsilon * size relative to the largest pivot).
 */
inline Mat pseudoInverseCOD(const Mat &H) {
  const int n = H.r;
  Vec w;
  Mat Q;
  symmetricEigen(H, w, Q);
  double wmax = 0;
  for (double v : w) wmax = std::max(wmax, std::abs(v));
  const double thr = std::numeric_limits<double>::epsilon() * n * wmax;
  Mat R(n, n);
  for (int e = 0; e < n; ++e) {
    if (std::abs(w[static_cast<size_t>(e)]) <= thr) continue;
    const double inv = 1.0 / w[static_cast<size_t>(e)];
    for (int i = 0; i < n; ++i) {
      const double qi = Q(i, e) * inv;
      for (int j = 0; j < n; ++j) R(i, j) += qi * Q(j, e);
    }
  }
  return R;
}

/** jacobiPreconditioner — normal_linear_system.cpp:10-16 */
inline Vec jacobiPreconditioner(const Mat &H) {
  const double kPreconditionerMinValue = 10;
  Vec p(static_cast<size_t>(H.r));
  for (int i = 0; i < H.r; ++i) p[static_cast<size_t>(i)] = 1.0 / std::sqrt(H(i, i) + kPreconditionerMinValue);
  return p;
}

/** NormalLinearSystem — src/energy/problems/include/energy/normal_linear_system.hpp:16-145 */
struct NormalLinearSystem {
  Mat H;
  Vec b;
  NormalLinearSystem() = default;
  explicit NormalLinearSystem(int n) : H(n, n), b(static_cast<size_t>(n), 0.0) {}
  int size() const { return H.r; }
  void setZero() {
    H.setZero();
    std::fill(b.begin(), b.end(), 0.0);
  }
  /** conservativeResize semantics (normal_linear_system.hpp:57-62); new entries zeroed by the caller in the reference,
   *  zeroed here directly (eigen_photometric_bundle_adjustment.cpp:137-140) */
  void resizeKeep(int n) {
    Mat Hn(n, n);
    Vec bn(static_cast<size_t>(n), 0.0);
    const int m = std::min(n, H.r);
    for (int i = 0; i < m; ++i) {
      bn[static_cast<size_t>(i)] = b[static_cast<size_t>(i)];
      for (int j = 0; j < m; ++j) Hn(i, j) = H(i, j);
    }
    H = Hn;
    b = bn;
  }
  /** NormalLinearSystem::solve — normal_linear_system.cpp:52-59 */
  Vec solve() const {
    const int n = H.r;
    Vec p = jacobiPreconditioner(H);
    Mat Hp(n, n);
    Vec bp(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
      bp[static_cast<size_t>(i)] = p[static_cast<size_t>(i)] * b[static_cast<size_t>(i)];
      for (int j = 0; j < n; ++j) Hp(i, j) = p[static_cast<size_t>(i)] * H(i, j) * p[static_cast<size_t>(j)];
    }
    Vec x = ldltSolve(Hp, bp);
    for (int i = 0; i < n; ++i) x[static_cast<size_t>(i)] *= p[static_cast<size_t>(i)];
    return x;
  }
  /** NormalLinearSystem::reduce_system — normal_linear_system.cpp:19-50 */
  void reduceSystem(const std::vector<int> &eliminate) {
    const int n = H.r;
    std::vector<char> is_elim(static_cast<size_t>(n), 0);
    for (int i : eliminate) is_elim[static_cast<size_t>(i)] = 1;
    std::vector<int> keep;
    for (int i = 0; i < n; ++i)
      if (!is_elim[static_cast<size_t>(i)]) keep.push_back(i);
    Vec p = jacobiPreconditioner(H);
    const int nk = static_cast<int>(keep.size()), ne = static_cast<int>(eliminate.size());
    auto hp = [&](int i, int j) { return p[static_cast<size_t>(i)] * H(i, j) * p[static_cast<size_t>(j)]; };
    Mat Hee(ne, ne), Hke(nk, ne), Hkk(nk, nk);
    for (int i = 0; i < ne; ++i)
      for (int j = 0; j < ne; ++j) Hee(i, j) = hp(eliminate[static_cast<size_t>(i)], eliminate[static_cast<size_t>(j)]);
    for (int i = 0; i < nk; ++i) {
      for (int j = 0; j < ne; ++j) Hke(i, j) = hp(keep[static_cast<size_t>(i)], eliminate[static_cast<size_t>(j)]);
      for (int j = 0; j < nk; ++j) Hkk(i, j) = hp(keep[static_cast<size_t>(i)], keep[static_cast<size_t>(j)]);
    }
    Mat schur_transform = matmul(Hke, pseudoInverseCOD(Hee));
    Mat red = matmul(schur_transform, transpose(Hke));
    Vec bk(static_cast<size_t>(nk)), be(static_cast<size_t>(ne));
    for (int i = 0; i < nk; ++i) bk[static_cast<size_t>(i)] = p[static_cast<size_t>(keep[static_cast<size_t>(i)])] * b[static_cast<size_t>(keep[static_cast<size_t>(i)])];
    for (int i = 0; i < ne; ++i) be[static_cast<size_t>(i)] = p[static_cast<size_t>(eliminate[static_cast<size_t>(i)])] * b[static_cast<size_t>(eliminate[static_cast<size_t>(i)])];
    Vec sb = matvec(schur_transform, be);
    Mat Hn(nk, nk);
    Vec bn(static_cast<size_t>(nk));
    for (int i = 0; i < nk; ++i)
      for (int j = 0; j < nk; ++j) Hkk(i, j) -= red(i, j);
    for (int i = 0; i < nk; ++i) {
      const double pi = 1.0 / p[static_cast<size_t>(keep[static_cast<size_t>(i)])];
      bn[static_cast<size_t>(i)] = pi * (bk[static_cast<size_t>(i)] - sb[static_cast<size_t>(i)]);
      for (int j = 0; j < nk; ++j) {
        const double pj = 1.0 / p[static_cast<size_t>(keep[static_cast<size_t>(j)])];
        Hn(i, j) = pi * (0.5 * (Hkk(i, j) + Hkk(j, i))) * pj;
      }
    }
    H = Hn;
    b = bn;
  }
};

}  // namespace oracle
