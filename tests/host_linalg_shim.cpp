// Test shim: exposes the host-side dense routines of the HIP library (dsopp_amd/csrc/host_linalg.hpp, plain C++) to
// tests/test_host_linalg.py.  Built with g++ by the test; not part of the product.
#include "host_linalg.hpp"

extern "C" {
/* path: 0 Jacobi eigen-solver, 1 null-direction path, 2 positive-definite path */
int shim_pinv_drop_smallest(const double *H, int n, double *out, double *out_jacobi) {
  dsopp_hip::hostla::Mat A(H, H + static_cast<size_t>(n) * n), fast;
  int path = 0;
  if (dsopp_hip::hostla::pinvDropNullDirection(A, n, fast)) path = 1;
  else if (dsopp_hip::hostla::pinvDropSmallestSpd(A, n, fast)) path = 2;
  const dsopp_hip::hostla::Mat P = dsopp_hip::hostla::pinvDropSmallest(A, n, 1);
  const dsopp_hip::hostla::Mat J = dsopp_hip::hostla::pinvDropSmallestJacobi(A, n, 1);
  for (size_t i = 0; i < P.size(); ++i) {
    out[i] = P[i];
    out_jacobi[i] = J[i];
  }
  return path;
}
}
