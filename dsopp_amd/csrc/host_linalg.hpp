// Small dense double-precision routines that stay on the host, as in the reference (they run once per solve / per
// new keyframe on K x K matrices, K <= 128): pseudo-inverses for the pose covariance
// (PROB_SRC/eigen_photometric_bundle_adjustment.cpp:31-45), Schur elimination of marginalised frames
// (NormalLinearSystem::reduce_system, PROB_SRC/normal_linear_system.cpp:19-50).  The reference uses Eigen's JacobiSVD /
// completeOrthogonalDecomposition; for the symmetric matrices involved both equal the spectral pseudo-inverse computed
// here with a cyclic Jacobi eigen-solver.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
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

/**
 * pinv of a symmetric positive definite matrix with exactly its smallest eigen-direction dropped, without a full
 * eigendecomposition:  sum_{i != min} v_i v_i^T / l_i  =  H^-1 - v_min v_min^T / l_min.
 * H^-1 through a Cholesky factorisation of the Jacobi-scaled matrix (entries span 1e16: the fixed-frame prior), the smallest
 * eigenpair by inverse iteration with that inverse.  Returns false (caller falls back to the Jacobi eigen-solver) when the
 * matrix is not safely positive definite, the iteration does not converge (no spectral gap), or the subtraction would cancel
 * more than ~7 digits (l_2 / l_min too large).  ~10x cheaper than the cyclic Jacobi sweep for the 56 x 56 system of a window.
 */
inline bool pinvDropSmallestSpd(const Mat &H, int n, Mat &out, int *why = nullptr, double *info = nullptr) {
  auto no = [&](int code, double v) {
    if (why) *why = code;
    if (info) *info = v;
    return false;
  };
  const size_t N = static_cast<size_t>(n);
  std::vector<double> d(N);
  for (int i = 0; i < n; ++i) {
    const double a = H[N * static_cast<size_t>(i) + static_cast<size_t>(i)];
    if (!(a > 0) || !std::isfinite(a)) return no(1, a);
    d[static_cast<size_t>(i)] = 1.0 / std::sqrt(a);
  }
  Mat L(N * N, 0.0);  // Cholesky factor of S = D H D (unit diagonal), lower triangle
  for (int j = 0; j < n; ++j) {
    for (int i = j; i < n; ++i) {
      double s = H[N * static_cast<size_t>(i) + static_cast<size_t>(j)] * d[static_cast<size_t>(i)] * d[static_cast<size_t>(j)];
      const double *li = &L[N * static_cast<size_t>(i)], *lj = &L[N * static_cast<size_t>(j)];
      for (int k = 0; k < j; ++k) s -= li[k] * lj[k];
      if (i == j) {
        if (!(s > 1e-13)) return no(2, s);
        L[N * static_cast<size_t>(j) + static_cast<size_t>(j)] = std::sqrt(s);
      } else {
        L[N * static_cast<size_t>(i) + static_cast<size_t>(j)] = s / L[N * static_cast<size_t>(j) + static_cast<size_t>(j)];
      }
    }
  }
  Mat Li(N * N, 0.0);  // L^-1 (lower triangular), row by row
  for (int i = 0; i < n; ++i) {
    double *ri = &Li[N * static_cast<size_t>(i)];
    const double *li = &L[N * static_cast<size_t>(i)];
    const double inv = 1.0 / li[i];
    for (int j = 0; j <= i; ++j) {
      double s = i == j ? 1.0 : 0.0;
      for (int k = j; k < i; ++k) s -= li[k] * Li[N * static_cast<size_t>(k) + static_cast<size_t>(j)];
      ri[j] = s * inv;
    }
  }
  Mat Ainv(N * N, 0.0);  // H^-1 = D (L^-T L^-1) D
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0;
      for (int k = i; k < n; ++k) s += Li[N * static_cast<size_t>(k) + static_cast<size_t>(i)] * Li[N * static_cast<size_t>(k) + static_cast<size_t>(j)];
      s *= d[static_cast<size_t>(i)] * d[static_cast<size_t>(j)];
      Ainv[N * static_cast<size_t>(i) + static_cast<size_t>(j)] = Ainv[N * static_cast<size_t>(j) + static_cast<size_t>(i)] = s;
    }
  // inverse iteration for the smallest eigenpair of H = dominant eigenpair of H^-1
  std::vector<double> x(N), y(N);
  for (int i = 0; i < n; ++i) x[static_cast<size_t>(i)] = 1.0 + 0.37 * std::sin(1.0 + 2.3 * i);  // fixed, generic start vector
  double mu = 0;
  bool converged = false;
  for (int it = 0; it < 200 && !converged; ++it) {
    double nrm = 0;
    for (int i = 0; i < n; ++i) {
      double s = 0;
      const double *row = &Ainv[N * static_cast<size_t>(i)];
      for (int k = 0; k < n; ++k) s += row[k] * x[static_cast<size_t>(k)];
      y[static_cast<size_t>(i)] = s;
      nrm += s * s;
    }
    nrm = std::sqrt(nrm);
    if (!(nrm > 0) || !std::isfinite(nrm)) return no(3, nrm);
    double diff = 0, dot = 0;
    for (int i = 0; i < n; ++i) dot += y[static_cast<size_t>(i)] * x[static_cast<size_t>(i)];
    const double sgn = dot < 0 ? -1.0 : 1.0;
    for (int i = 0; i < n; ++i) {
      const double v = sgn * y[static_cast<size_t>(i)] / nrm;
      diff = std::max(diff, std::abs(v - x[static_cast<size_t>(i)]));
      x[static_cast<size_t>(i)] = v;
    }
    if (it > 0 && diff < 1e-15) {
      converged = true;
      mu = nrm;  // |H^-1 x| for the unit eigenvector = 1 / l_min
    }
    if (it == 0) {
      double xn = 0;
      for (double v : x) xn += v * v;
      (void)xn;
    }
  }
  if (!converged) return no(4, mu);
  // refine mu as the Rayleigh quotient of H^-1 and bound the cancellation of the subtraction below
  {
    double q = 0;
    for (int i = 0; i < n; ++i) {
      double s = 0;
      const double *row = &Ainv[N * static_cast<size_t>(i)];
      for (int k = 0; k < n; ++k) s += row[k] * x[static_cast<size_t>(k)];
      q += s * x[static_cast<size_t>(i)];
    }
    mu = q;
  }
  out.assign(N * N, 0.0);
  double rest = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      const double v = Ainv[N * static_cast<size_t>(i) + static_cast<size_t>(j)] - mu * x[static_cast<size_t>(i)] * x[static_cast<size_t>(j)];
      out[N * static_cast<size_t>(i) + static_cast<size_t>(j)] = v;
      rest = std::max(rest, std::abs(v));
    }
  if (!(rest > 0) || mu > 1e7 * rest) return no(5, mu / rest);  // l_2 / l_min beyond 1e7: the difference keeps fewer than ~9 digits
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j) {
      const double v = 0.5 * (out[N * static_cast<size_t>(i) + static_cast<size_t>(j)] + out[N * static_cast<size_t>(j) + static_cast<size_t>(i)]);
      out[N * static_cast<size_t>(i) + static_cast<size_t>(j)] = out[N * static_cast<size_t>(j) + static_cast<size_t>(i)] = v;
    }
  return true;
}

/** lower Cholesky factor of a dense symmetric matrix (row-major n x n, lower triangle read); false if a pivot <= floor.
 *  Right-looking with the finished column kept in a contiguous buffer: the trailing update is a row-wise axpy the compiler vectorises
 *  (the dot-product form is a floating-point reduction, which it may not reorder) — this routine and the two below are on the critical
 *  path of every production solve(): the host inverts the K x K system while the device is already idle (DSOPP_HIP_HOST_TIMES). */
// (cores on raw pointers with __restrict__: the inner loops are contiguous axpy updates the compiler vectorises.  A second build of them for
// AVX2 + FMA picked at load time — target_clones — was tried and brought nothing over the SSE2 baseline at K = 64: 97 against 96 us for the
// whole pseudo-inverse; what did was walking L by rows of its transpose: 171 -> 96 us)
inline bool choleskyLowerCore(double *__restrict__ L, double *__restrict__ c, int n, double floor) {
  const size_t N = static_cast<size_t>(n);
  for (int j = 0; j < n; ++j) {
    const double d = L[N * static_cast<size_t>(j) + static_cast<size_t>(j)];
    if (!(d > floor)) return false;
    const double r = std::sqrt(d), inv = 1.0 / r;
    L[N * static_cast<size_t>(j) + static_cast<size_t>(j)] = r;
    for (int i = j + 1; i < n; ++i) {
      const double l = L[N * static_cast<size_t>(i) + static_cast<size_t>(j)] * inv;
      L[N * static_cast<size_t>(i) + static_cast<size_t>(j)] = l;
      c[i] = l;
    }
    for (int i = j + 1; i < n; ++i) {
      const double l = c[i];
      double *__restrict__ row = L + N * static_cast<size_t>(i);
      for (int k = j + 1; k <= i; ++k) row[k] -= l * c[k];
    }
  }
  return true;
}
inline bool choleskyLower(const Mat &A, int n, double floor, Mat &L) {
  const size_t N = static_cast<size_t>(n);
  L.assign(N * N, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) L[N * static_cast<size_t>(i) + static_cast<size_t>(j)] = A[N * static_cast<size_t>(i) + static_cast<size_t>(j)];
  std::vector<double> col(N);
  return choleskyLowerCore(L.data(), col.data(), n, floor);
}
/** x <- (L L^T)^-1 x */
inline void choleskySolveCore(const double *__restrict__ L, int n, double *__restrict__ x) {
  const size_t N = static_cast<size_t>(n);
  for (int i = 0; i < n; ++i) {
    double s = x[i];
    const double *li = L + N * static_cast<size_t>(i);
    for (int k = 0; k < i; ++k) s -= li[k] * x[k];
    x[i] = s / li[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = x[i];
    for (int k = i + 1; k < n; ++k) s -= L[N * static_cast<size_t>(k) + static_cast<size_t>(i)] * x[k];
    x[i] = s / L[N * static_cast<size_t>(i) + static_cast<size_t>(i)];
  }
}
inline void choleskySolveInPlace(const Mat &L, int n, std::vector<double> &x) { choleskySolveCore(L.data(), n, x.data()); }
/** (L L^T)^-1 as a dense symmetric matrix.  Lt, Xt, X: n x n work arrays, zero-filled; out: zero-filled */
inline void choleskyInverseCore(const double *__restrict__ L, int n, double *__restrict__ Lt, double *__restrict__ Xt, double *__restrict__ X,
                                                 double *__restrict__ out) {
  const size_t N = static_cast<size_t>(n);
  // L^T row-major: column i of L (what a forward substitution walks) is its row i — contiguous
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) Lt[N * static_cast<size_t>(j) + static_cast<size_t>(i)] = L[N * static_cast<size_t>(i) + static_cast<size_t>(j)];
  // X = L^-1 column by column, kept TRANSPOSED (Xt[j] = column j of L^-1, entries i >= j): forward substitution as row-wise axpy
  for (int j = 0; j < n; ++j) {
    double *__restrict__ x = Xt + N * static_cast<size_t>(j);
    x[j] = 1.0;
    for (int i = j; i < n; ++i) {
      const double *__restrict__ li = Lt + N * static_cast<size_t>(i);
      const double xi = x[i] / li[i];
      x[i] = xi;
      for (int k = i + 1; k < n; ++k) x[k] -= li[k] * xi;  // x[k] -= L[k][i] x[i] for k > i
    }
  }
  // out = X^T X: out[i][j] = sum_k X[k][i] X[k][j]; as axpy over the rows of X (= columns of Xt) it needs X itself row-major: build it once
  for (int j = 0; j < n; ++j)
    for (int i = j; i < n; ++i) X[N * static_cast<size_t>(i) + static_cast<size_t>(j)] = Xt[N * static_cast<size_t>(j) + static_cast<size_t>(i)];
  for (int k = 0; k < n; ++k) {
    const double *__restrict__ xk = X + N * static_cast<size_t>(k);  // row k of L^-1: entries 0 .. k
    for (int i = 0; i <= k; ++i) {
      const double a = xk[i];
      if (a == 0) continue;
      double *__restrict__ orow = out + N * static_cast<size_t>(i);
      for (int j = 0; j <= i; ++j) orow[j] += a * xk[j];
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j) out[N * static_cast<size_t>(j) + static_cast<size_t>(i)] = out[N * static_cast<size_t>(i) + static_cast<size_t>(j)];
}
inline Mat choleskyInverse(const Mat &L, int n) {
  const size_t N = static_cast<size_t>(n);
  Mat work(3 * N * N, 0.0), out(N * N, 0.0);
  choleskyInverseCore(L.data(), n, work.data(), work.data() + N * N, work.data() + 2 * N * N, out.data());
  return out;
}

/** y = M x (four running sums per row, fixed order) */
inline void matVecCore(const double *__restrict__ M, const double *__restrict__ x, int n, double *__restrict__ y) {
  const size_t N = static_cast<size_t>(n);
  for (int i = 0; i < n; ++i) {
    const double *__restrict__ row = M + N * static_cast<size_t>(i);
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int k = 0;
    for (; k + 3 < n; k += 4) {
      s0 += row[k] * x[k];
      s1 += row[k + 1] * x[k + 1];
      s2 += row[k + 2] * x[k + 2];
      s3 += row[k + 3] * x[k + 3];
    }
    for (; k < n; ++k) s0 += row[k] * x[k];
    y[i] = (s0 + s1) + (s2 + s3);
  }
}
/** out[i][j] = in[i][j] * d[i] * d[j] (+ c * e[i] * e[j] when e is given) — the element-wise passes of pinvDropNullDirection */
inline void scaleSymmetricCore(const double *__restrict__ in, const double *__restrict__ d, int n, double *__restrict__ out, double c,
                                                const double *__restrict__ e) {
  const size_t N = static_cast<size_t>(n);
  for (int i = 0; i < n; ++i) {
    const double di = d[i], ce = e ? c * e[i] : 0.0;
    const double *__restrict__ r = in + N * static_cast<size_t>(i);
    double *__restrict__ o = out + N * static_cast<size_t>(i);
    if (e)
      for (int j = 0; j < n; ++j) o[j] = r[j] * di * d[j] + ce * e[j];
    else
      for (int j = 0; j < n; ++j) o[j] = r[j] * di * d[j];
  }
}
/** out = M - Mv v^T - v Mv^T + vMv v v^T; returns the squared Frobenius norm of out */
inline double projectOutCore(const double *__restrict__ M, const double *__restrict__ Mv, const double *__restrict__ v, double vMv, int n,
                                              double *__restrict__ out) {
  const size_t N = static_cast<size_t>(n);
  double fro = 0;
  for (int i = 0; i < n; ++i) {
    const double mvi = Mv[i], vi = v[i], a = vMv * vi;
    const double *__restrict__ r = M + N * static_cast<size_t>(i);
    double *__restrict__ o = out + N * static_cast<size_t>(i);
    double f = 0;
    for (int j = 0; j < n; ++j) {
      const double w = r[j] - mvi * v[j] - vi * Mv[j] + a * v[j];
      o[j] = w;
      f += w * w;
    }
    fro += f;
  }
  return fro;
}

/**
 * pinv of a symmetric positive SEMI-definite matrix whose smallest singular direction is a numerical null space (the scale
 * gauge of a monocular window: l_min / l_max ~ 1e-14 after Jacobi scaling) with exactly that direction dropped:
 *   v  = null vector: inverse iteration on the Jacobi-scaled matrix S = D H D shifted by tau (z), mapped back v = D z / |D z|;
 *   pinv = P (H + c v v^T)^-1 P,  P = I - v v^T  (for an exact null vector (H + c v v^T)^-1 = pinv + v v^T / c).
 * Verified before it is returned (else the caller falls back to the Jacobi eigen-solver): the dropped direction's singular
 * value |H v| is 1e-6 below every kept one, and H pinv H = H to 1e-7 in the scaled metric.
 */
inline bool pinvDropNullDirection(const Mat &H, int n, Mat &out, int *why = nullptr, double *info = nullptr) {
  auto no = [&](int code, double v) {
    if (why) *why = code;
    if (info) *info = v;
    return false;
  };
  const size_t N = static_cast<size_t>(n);
  std::vector<double> d(N);
  for (int i = 0; i < n; ++i) {
    const double a = H[N * static_cast<size_t>(i) + static_cast<size_t>(i)];
    if (!(a > 0) || !std::isfinite(a)) return no(11, a);
    d[static_cast<size_t>(i)] = 1.0 / std::sqrt(a);
  }
  Mat S(N * N);
  scaleSymmetricCore(H.data(), d.data(), n, S.data(), 0.0, nullptr);
  Mat St = S, L;
  const double tau = 1e-9;
  for (int i = 0; i < n; ++i) St[N * static_cast<size_t>(i) + static_cast<size_t>(i)] += tau;
  if (!choleskyLower(St, n, 1e-12, L)) return no(12, 0);
  std::vector<double> z(N), prev(N);
  for (int i = 0; i < n; ++i) z[static_cast<size_t>(i)] = 1.0 + 0.37 * std::sin(1.0 + 2.3 * i);
  bool converged = false;
  for (int it = 0; it < 100 && !converged; ++it) {
    prev = z;
    choleskySolveInPlace(L, n, z);
    double nrm = 0, dot = 0;
    for (int i = 0; i < n; ++i) nrm += z[static_cast<size_t>(i)] * z[static_cast<size_t>(i)];
    nrm = std::sqrt(nrm);
    if (!(nrm > 0) || !std::isfinite(nrm)) return no(13, nrm);
    for (int i = 0; i < n; ++i) dot += z[static_cast<size_t>(i)] * prev[static_cast<size_t>(i)];
    const double sg = dot < 0 ? -1.0 / nrm : 1.0 / nrm;
    double diff = 0;
    for (int i = 0; i < n; ++i) {
      z[static_cast<size_t>(i)] *= sg;
      diff = std::max(diff, std::abs(z[static_cast<size_t>(i)] - prev[static_cast<size_t>(i)]));
    }
    converged = it > 0 && diff < 1e-13;
  }
  if (!converged) return no(14, 0);
  std::vector<double> v(N), Dv(N);
  double vn = 0;
  for (int i = 0; i < n; ++i) {
    v[static_cast<size_t>(i)] = d[static_cast<size_t>(i)] * z[static_cast<size_t>(i)];
    vn += v[static_cast<size_t>(i)] * v[static_cast<size_t>(i)];
  }
  vn = std::sqrt(vn);
  double dv2 = 0, sigma = 0;
  for (int i = 0; i < n; ++i) {
    v[static_cast<size_t>(i)] /= vn;
    Dv[static_cast<size_t>(i)] = d[static_cast<size_t>(i)] * v[static_cast<size_t>(i)];
    dv2 += Dv[static_cast<size_t>(i)] * Dv[static_cast<size_t>(i)];
  }
  {
    std::vector<double> Hv(N);
    matVecCore(H.data(), v.data(), n, Hv.data());
    for (int i = 0; i < n; ++i) sigma += Hv[static_cast<size_t>(i)] * Hv[static_cast<size_t>(i)];
  }
  sigma = std::sqrt(sigma);  // singular value of the dropped direction
  // S_M = D (H + c v v^T) D = S + c (Dv)(Dv)^T with c |Dv|^2 = 1
  const double c = 1.0 / dv2;
  Mat SM(N * N);
  scaleSymmetricCore(H.data(), d.data(), n, SM.data(), c, Dv.data());
  if (!choleskyLower(SM, n, 1e-11, L)) return no(15, 0);
  const Mat Minv_scaled = choleskyInverse(L, n);
  Mat Minv(N * N);
  scaleSymmetricCore(Minv_scaled.data(), d.data(), n, Minv.data(), 0.0, nullptr);
  // P Minv P
  std::vector<double> Mv(N, 0.0);
  double vMv = 0;
  matVecCore(Minv.data(), v.data(), n, Mv.data());
  for (int i = 0; i < n; ++i) vMv += Mv[static_cast<size_t>(i)] * v[static_cast<size_t>(i)];
  out.assign(N * N, 0.0);
  const double fro = std::sqrt(projectOutCore(Minv.data(), Mv.data(), v.data(), vMv, n, out.data()));
  if (!(sigma * fro < 1e-6)) return no(16, sigma * fro);  // the dropped direction is not (numerically) a null space
  // H pinv H = H in the scaled metric, tested on three fixed probe vectors instead of entry by entry: with E = D (H pinv H - H) D the
  // entrywise bound max |E_ij| < 1e-7 implies |E x|_inf < 1e-7 |x|_1, and a pinv that is wrong in ANY direction fails the probes unless that
  // direction is orthogonal to all three (the two full K x K products of the entrywise test were half of this routine's time, which sits on
  // the critical path of every production solve(): the device is idle by the time the host gets here)
  double worst = 0;
  {
    std::vector<double> x(N), y(N), z(N), u(N);
    for (int probe = 0; probe < 3; ++probe) {
      double x1 = 0;
      for (int i = 0; i < n; ++i) {
        const double xt = probe == 0 ? 1.0 : (probe == 1 ? ((i & 1) ? -1.0 : 1.0) : std::sin(1.0 + 2.3 * i));
        x1 += std::abs(xt);
        x[static_cast<size_t>(i)] = d[static_cast<size_t>(i)] * xt;  // x = D x~
      }
      auto mul = [&](const Mat &M, const std::vector<double> &in, std::vector<double> &res) { matVecCore(M.data(), in.data(), n, res.data()); };
      mul(H, x, y);    // H x
      mul(out, y, z);  // pinv H x
      mul(H, z, u);    // H pinv H x
      for (int i = 0; i < n; ++i) worst = std::max(worst, std::abs(u[static_cast<size_t>(i)] - y[static_cast<size_t>(i)]) * d[static_cast<size_t>(i)] / x1);
    }
  }
  if (!(worst < 1e-7)) return no(17, worst);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j) {
      const double w = 0.5 * (out[N * static_cast<size_t>(i) + static_cast<size_t>(j)] + out[N * static_cast<size_t>(j) + static_cast<size_t>(i)]);
      out[N * static_cast<size_t>(i) + static_cast<size_t>(j)] = out[N * static_cast<size_t>(j) + static_cast<size_t>(i)] = w;
    }
  return true;
}

/** pseudoInverse(origin, number_of_nullspaces) through the cyclic Jacobi eigen-solver (relative accuracy on graded matrices) */
inline Mat pinvDropSmallestJacobi(const Mat &H, int n, int nullspaces) {
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

/** pseudoInverse(origin, number_of_nullspaces): invert all but the smallest `nullspaces` singular values */
inline Mat pinvDropSmallest(const Mat &H, int n, int nullspaces) {
  if (nullspaces == 1) {
    Mat fast;
    int why = 0;
    double info = 0;
    if (pinvDropNullDirection(H, n, fast, &why, &info)) return fast;
    const int why_null = why;
    const double info_null = info;
    if (pinvDropSmallestSpd(H, n, fast, &why, &info)) return fast;
    if (std::getenv("DSOPP_HIP_TRACE"))
      std::fprintf(stderr, "[dsopp_hip] pinv fast paths declined: null-space path %d (%g), positive-definite path %d (%g)\n", why_null, info_null,
                   why, info);
  }
  return pinvDropSmallestJacobi(H, n, nullspaces);
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
