// Counter calibration aid (NOT part of the C-ABI in include/dsopp_hip.h nor of libdsopp_hip.so: libdsopp_hip_tools.so, used by
// scripts/pmc_target.py only): a gather with the
// access shape of the sweeps' bilinear sampling — every lane reads whole 32-byte texels at data-dependent positions — over a
// buffer of known geometry, so that FETCH_SIZE of the TCC counters can be converted to bytes for THIS access pattern
// (MI355X_MICROARCH.md §HBM: the counter is calibrated for wide streaming reads only).
#include "pyramid.hpp"

namespace dsopp_hip {
namespace {

/** lane i sums `per_lane` consecutive texels starting at idx[i] (per_lane = 1: isolated texel; 2: one row of a bilinear footprint) */
__global__ void gatherCalibrationKernel(const Texel<double> *__restrict__ tex, const uint32_t *__restrict__ idx, size_t n, int per_lane,
                                        size_t row_stride, double *__restrict__ out) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Texel<double> *p = tex + idx[i];
  double s = 0;
  for (int k = 0; k < per_lane; ++k) {
    const Texel<double> t = p[k];
    s += t.I + t.mask + t.Ix + t.Iy;
    if (row_stride) {  // the second row of the footprint
      const Texel<double> u = p[row_stride + k];
      s += u.I + u.mask + u.Ix + u.Iy;
    }
  }
  if (s == 12345.678) out[0] = s;  // never true for the calibration data: keeps the loads alive without a store stream
}

}  // namespace
}  // namespace dsopp_hip

using namespace dsopp_hip;

extern "C" int dsopp_hip_debug_gather_calibration(const void *texels, const uint32_t *indices, size_t n, int per_lane, size_t row_stride, void *scratch,
                                                  void *stream) {
  // (self-contained: this file is linked into libdsopp_hip_tools.so, without the product library's error plumbing)
  if (!texels || !indices || !scratch || per_lane < 1) return -1;
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  gatherCalibrationKernel<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(static_cast<const Texel<double> *>(texels), indices, n, per_lane, row_stride,
                                                                            static_cast<double *>(scratch));
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
