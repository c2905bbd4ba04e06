// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// CPU restatement of the per-frame image pyramid construction (hot loop G):
//   PixelDataFrame ctor          — src/features/src/pixel_data_frame.cpp:12-31
//   photometricallyCorrectedImage — src/features/src/photometrically_corrected_image.cpp:9-29
//   downscaleImage               — src/features/internal/features/camera/downscale_image.hpp:16-33
//   calculate_pixelinfo (scalar) — src/features/src/calculate_pixelinfo.cpp:340-374
// The reference's AVX2 variant is asserted bit-exact to the scalar definition by its own test
// (test/test/features/test_dxdy_accelerated.cpp:11-85), so the scalar definition is the spec.
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

namespace oracle {

/** level-0 plane: LUT[u8] * (vmax / (vignette + 1)) */
inline std::vector<double> photometricallyCorrectedImage(const uint8_t *image, int width, int height, const double *lut256,
                                                         const uint8_t *vignetting) {
  const size_t n = static_cast<size_t>(width) * height;
  std::vector<double> result(n);
  double max_value_vignetting = 0;
  if (vignetting) max_value_vignetting = static_cast<double>(*std::max_element(vignetting, vignetting + n));
  for (size_t i = 0; i < n; ++i) {
    double v = lut256 ? lut256[image[i]] : static_cast<double>(image[i]);
    if (vignetting) v *= max_value_vignetting / (static_cast<double>(vignetting[i]) + 1);
    result[i] = v;
  }
  return result;
}

/** 2x box filter of the scalar plane */
inline std::vector<double> downscaleImage(const std::vector<double> &image, int height, int width) {
  const int h2 = height / 2, w2 = width / 2;
  std::vector<double> out(static_cast<size_t>(h2) * w2);
  for (int y = 0; y < h2; ++y)
    for (int x = 0; x < w2; ++x) {
      const size_t i00 = static_cast<size_t>(2 * y) * width + 2 * x;
      out[static_cast<size_t>(y) * w2 + x] =
          0.25 * (image[i00] + image[i00 + width + 1] + image[i00 + 1] + image[i00 + width]);
    }
  return out;
}

/** (I, dx, dy) AoS with central differences, one-sided (x1.0) at the borders */
inline std::vector<double> calculatePixelInfo(const std::vector<double> &plane, int width, int height) {
  std::vector<double> out(static_cast<size_t>(width) * height * 3);
  for (int y = 0; y < height; ++y) {
    const bool first_last_row = (y == 0) || (y == height - 1);
    const double *cur = plane.data() + static_cast<size_t>(width) * y;
    const double *up = (y == 0) ? cur : cur - width;
    const double *bot = (y == height - 1) ? cur : cur + width;
    double *o = out.data() + static_cast<size_t>(width) * y * 3;
    for (int i = 0; i < width; ++i) {
      o[3 * i] = cur[i];
      if (i == 0)
        o[3 * i + 1] = 1.0 * (cur[i + 1] - cur[i]);
      else if (i == width - 1)
        o[3 * i + 1] = 1.0 * (cur[i] - cur[i - 1]);
      else
        o[3 * i + 1] = 0.5 * (cur[i + 1] - cur[i - 1]);
      o[3 * i + 2] = (first_last_row ? 1.0 : 0.5) * (bot[i] - up[i]);
    }
  }
  return out;
}

struct Pyramid {
  std::vector<std::vector<double>> planes;     // scalar plane per level
  std::vector<std::vector<double>> pixelinfo;  // (I,dx,dy) AoS per level
  std::vector<int> widths, heights;
};

inline Pyramid buildPyramid(const uint8_t *image, int width, int height, const double *lut256, const uint8_t *vignetting,
                            int levels) {
  const int kMaxPyramidDepth = 5;  // pixel_data_frame.hpp:26
  levels = std::min(levels, kMaxPyramidDepth);
  Pyramid p;
  p.planes.push_back(photometricallyCorrectedImage(image, width, height, lut256, vignetting));
  p.widths.push_back(width);
  p.heights.push_back(height);
  for (int level = 1; level < levels; ++level) {
    p.planes.push_back(downscaleImage(p.planes.back(), height, width));
    width /= 2;
    height /= 2;
    p.widths.push_back(width);
    p.heights.push_back(height);
  }
  for (size_t l = 0; l < p.planes.size(); ++l) p.pixelinfo.push_back(calculatePixelInfo(p.planes[l], p.widths[l], p.heights[l]));
  return p;
}

}  // namespace oracle
