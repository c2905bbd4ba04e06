// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README in oracle.h).  PARITY UNPINNED: the reference cannot be built or
// imported here and ships no vectors for this function; this file is a CPU restatement of
//   createReferenceDepthMaps — src/tracker/tracker/src/create_depth_maps.cpp:18-147
// (fillFineDepthMap :18-59, fillCoarseDepthMaps :70-88, dilateDepthMaps :90-122), over plain arrays instead of
// track::ActiveKeyframe, and of calculateMeanSquareOpticalFlow — src/tracker/tracker/src/monocular_tracker.cpp:104-134.  The product path never links or calls it.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "geometry.hpp"
#include "se3.hpp"

namespace oracle {

/** energy::problem::DepthMap: per pixel {idepth (weighted sum), weight}; stored row-major H x W here
 *  (the reference indexes its Eigen::Array as map(x, y)) */
struct DepthMapLevel {
  int width = 0, height = 0;
  std::vector<double> idepth, weight;
  DepthMapLevel(int w, int h) : width(w), height(h), idepth(static_cast<size_t>(w) * h, 0.0), weight(static_cast<size_t>(w) * h, 0.0) {}
  size_t at(int x, int y) const { return static_cast<size_t>(y) * width + x; }
};

/** the landmarks of one older keyframe as createReferenceDepthMaps reads them from track::ActiveKeyframe */
struct DepthMapSource {
  SE3 t_world_agent;
  int n = 0;
  const double *uv = nullptr;         // projection(), 2 per landmark
  const double *idepth = nullptr;     // idepth()
  const double *variance = nullptr;   // idepthVariance()
  const uint8_t *skip = nullptr;      // isOutlier() || isMarginalized()
  const uint8_t *status = nullptr;    // referenceReprojectionStatuses towards the newest keyframe
};

/** fillFineDepthMap — create_depth_maps.cpp:18-59 */
inline void fillFineDepthMap(const std::vector<DepthMapSource> &sources, const SE3 &t_world_newest, const PinholeModel &model,
                             DepthMapLevel &fine) {
  const double kEps = 1e-12;
  const double kVariationScale = 1e-3;
  for (const DepthMapSource &frame : sources) {
    const SE3 t_t_r = t_world_newest.inverse() * frame.t_world_agent;  // :28
    const ArrayReprojector<true> reprojector(model, model, t_t_r);      // :31 (kCheckSuccess defaults to true, camera_reproject.hpp:13)
    double M[12];
    t_t_r.matrix3x4(M);
    for (int i = 0; i < frame.n; ++i) {
      if (frame.status[i] != kOk) continue;  // :36
      if (frame.skip[i]) continue;           // :38
      const double u = frame.uv[2 * i], v = frame.uv[2 * i + 1];
      double tu, tv;
      if (!reprojector.reprojectPattern<1>(&u, &v, frame.idepth[i], &tu, &tv)) continue;  // :42-44
      const int ix = static_cast<int>(std::round(tu)), iy = static_cast<int>(std::round(tv));  // :46
      // getDepthScale — camera_model_base.hpp:102-107: z of T_t_r * [direction; idepth], direction = unproject(projection)
      // = ((u - cx)/fx, (v - cy)/fy, 1) (build_features.hpp:24-27, pinhole_camera.hpp:137-139)
      const double dx = (u - model.cx) * (1.0 / model.fx), dy = (v - model.cy) * (1.0 / model.fy);
      const double depth_scale = M[8] * dx + M[9] * dy + M[10] * 1.0 + M[11] * frame.idepth[i];
      const double weight = std::sqrt(kVariationScale / (frame.variance[i] + kEps));  // :51
      fine.idepth[fine.at(ix, iy)] += frame.idepth[i] / depth_scale * weight;        // :52
      fine.weight[fine.at(ix, iy)] += weight;                                        // :53
    }
  }
}

/** fillCoarseDepthMaps — create_depth_maps.cpp:70-88 (2 x 2 sum of both fields) */
inline void fillCoarseDepthMaps(std::vector<DepthMapLevel> &maps) {
  for (size_t lvl = 1; lvl < maps.size(); ++lvl) {
    DepthMapLevel &m = maps[lvl];
    const DepthMapLevel &up = maps[lvl - 1];
    for (int y = 0; y < m.height; ++y)
      for (int x = 0; x < m.width; ++x) {
        m.idepth[m.at(x, y)] = up.idepth[up.at(2 * x, 2 * y)] + up.idepth[up.at(2 * x + 1, 2 * y)] + up.idepth[up.at(2 * x, 2 * y + 1)] +
                               up.idepth[up.at(2 * x + 1, 2 * y + 1)];
        m.weight[m.at(x, y)] = up.weight[up.at(2 * x, 2 * y)] + up.weight[up.at(2 * x + 1, 2 * y)] + up.weight[up.at(2 * x, 2 * y + 1)] +
                               up.weight[up.at(2 * x + 1, 2 * y + 1)];
      }
  }
}

/** dilateDepthMaps — create_depth_maps.cpp:90-122: empty interior cells take the mean of their valid neighbours
 *  (diagonal neighbours on levels 0 and 1, 4-neighbours above) */
inline void dilateDepthMaps(std::vector<DepthMapLevel> &maps) {
  for (size_t lvl = 0; lvl < maps.size(); ++lvl) {
    DepthMapLevel &m = maps[lvl];
    const std::vector<double> backup = m.weight;
    const int off_axis[4][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}};
    const int off_diag[4][2] = {{1, 1}, {-1, -1}, {1, -1}, {-1, 1}};
    for (int y = 1; y < m.height - 1; ++y)
      for (int x = 1; x < m.width - 1; ++x) {
        if (backup[m.at(x, y)] > 0) continue;
        double sum = 0, num = 0, numn = 0;
        for (int k = 0; k < 4; ++k) {
          const int cx = x + (lvl > 1 ? off_axis[k][0] : off_diag[k][0]);
          const int cy = y + (lvl > 1 ? off_axis[k][1] : off_diag[k][1]);
          if (backup[m.at(cx, cy)] > 0) {
            sum += m.idepth[m.at(cx, cy)];
            num += backup[m.at(cx, cy)];
            numn += 1;
          }
        }
        if (numn > 0) {
          m.idepth[m.at(x, y)] = sum / numn;
          m.weight[m.at(x, y)] = num / numn;
        }
      }
  }
}

/** createReferenceDepthMaps — create_depth_maps.cpp:124-147 for one sensor; level sizes = the newest frame's pyramid */
inline std::vector<DepthMapLevel> createReferenceDepthMaps(const std::vector<DepthMapSource> &sources, const SE3 &t_world_newest,
                                                           const PinholeModel &model, int levels) {
  std::vector<DepthMapLevel> maps;
  int w = static_cast<int>(model.width), h = static_cast<int>(model.height);
  for (int l = 0; l < levels; ++l) {
    maps.emplace_back(w, h);  // initDepthMaps :62-68
    w /= 2;                   // downscale_image.hpp: level sizes halve (floor)
    h /= 2;
  }
  fillFineDepthMap(sources, t_world_newest, model, maps[0]);
  fillCoarseDepthMaps(maps);
  dilateDepthMaps(maps);
  return maps;
}

/** calculateMeanSquareOpticalFlow — monocular_tracker.cpp:104-134: RMS distance between the bearing of every depth-map pixel
 *  and the bearing of its reprojection under t_t_r (the keyframe strategy's parallax measure) */
inline double calculateMeanSquareOpticalFlow(const DepthMapLevel &m, const SE3 &t_t_r, const PinholeModel &model) {
  const int kBorderSize = 4;
  const double kMinIdepth = 1e-6;
  double square_optical_flow = 0;
  size_t n = 0;
  const ArrayReprojector<true> reprojector(model, model, t_t_r);
  for (int y = kBorderSize; y < m.height - kBorderSize; y++)
    for (int x = kBorderSize; x < m.width - kBorderSize; x++) {
      if (!(m.weight[m.at(x, y)] > 0)) continue;
      const double idepth = m.idepth[m.at(x, y)] / m.weight[m.at(x, y)];
      if (idepth < kMinIdepth) continue;
      const double u = x, v = y;
      double tu, tv;
      if (!reprojector.reprojectPattern<1>(&u, &v, idepth, &tu, &tv)) continue;
      // PinholeCamera::unproject — pinhole_camera.hpp:129-141: (p - c) * (1 / f), z = 1
      const double ax = (u - model.cx) * (1 / model.fx), ay = (v - model.cy) * (1 / model.fy);
      const double bx = (tu - model.cx) * (1 / model.fx), by = (tv - model.cy) * (1 / model.fy);
      square_optical_flow += (ax - bx) * (ax - bx) + (ay - by) * (ay - by);
      ++n;
    }
  return std::sqrt(square_optical_flow / static_cast<double>(n));
}

}  // namespace oracle
