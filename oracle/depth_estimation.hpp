// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (the reference cannot be built or imported here and ships no vectors
// for this path).  CPU restatement of the depth estimator for immature landmarks (row f-1 of SURVEY.md §8):
//   DepthEstimation::estimate / estimateLandmark / findBest / refine / DepthEstimationProblem
//       — src/tracker/depth_estimators/src/depth_estimation.cpp:26-381
//   EpipolarLine                       — src/energy/epipolar_geometry/src/epipolar_line.cpp:1-82
//   EpipolarLineBuilder<Pinhole, SE3>  — .../epipolar_geometry/epipolar_line_builder_pinhole_se3.hpp:22-390
//   EpipolarLineTriangulatorSE3        — .../epipolar_geometry/src/se3_epipolar_line_triangulator.cpp:7-39
//   ImmatureTrackingLandmark           — src/track/landmarks/src/immature_tracking_landmark.cpp:9-62
// The product path never links or calls it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <utility>
#include <vector>

#include "geometry.hpp"
#include "se3.hpp"

namespace oracle {

/** track::landmarks::ImmatureStatus — immature_tracking_landmark.hpp:14-22 */
enum ImmatureStatus : uint8_t { kGood = 0, kOutOfBoundary, kImmatureOutlier, kSkipped, kIllConditioned, kUninitialized, kDelete };

/** ImmatureTrackingLandmark (the fields the estimator reads and writes) — immature_tracking_landmark.hpp:93-106 */
struct ImmatureLandmark {
  double projection[2] = {0, 0};
  double direction[3] = {0, 0, 1};  // model.unproject(projection), build_features.hpp:24-27
  double patch[kPatternSize] = {0};
  double gradient[2] = {0, 0};
  double idepth_min = 0;
  double idepth_max = 1. / 0.001;
  double uniqueness = std::numeric_limits<double>::max();
  double search_pixel_interval = std::numeric_limits<double>::max();
  uint8_t status = kUninitialized;
  bool traced = false;

  void setStatus(uint8_t s) {  // immature_tracking_landmark.cpp:28-33
    if (s == kGood) traced = true;
    status = s;
  }
  void setUniqueness(double u, bool force_set) {  // :46-50
    if (force_set || u < uniqueness) uniqueness = u;
  }
};

/** EpipolarLineTriangulatorSE3 — se3_epipolar_line_triangulator.cpp:7-39 */
struct EpipolarLineTriangulatorSE3 {
  double Kt[3], bearing[3];
  bool use_x_direction;
  double kMaxIdepth;
  static constexpr double kZeroIdepthEps = 1e-5;  // se3_epipolar_line_triangulator.hpp:38

  EpipolarLineTriangulatorSE3(const SE3 &t_t_r, const PinholeModel &m, const double point_reference[2], double max_idepth)
      : kMaxIdepth(max_idepth) {
    double R[9];
    t_t_r.rotation(R);
    const double K[9] = {m.fx, 0, m.cx, 0, m.fy, m.cy, 0, 0, 1};
    const double Kinv[9] = {1 / m.fx, 0, -m.cx / m.fx, 0, 1 / m.fy, -m.cy / m.fy, 0, 0, 1};
    for (int i = 0; i < 3; ++i) Kt[i] = K[3 * i] * t_t_r.t[0] + K[3 * i + 1] * t_t_r.t[1] + K[3 * i + 2] * t_t_r.t[2];
    // K * R * K^-1 * p.homogeneous(), evaluated left to right as Eigen does
    double KR[9], M[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) KR[3 * i + j] = K[3 * i] * R[j] + K[3 * i + 1] * R[3 + j] + K[3 * i + 2] * R[6 + j];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) M[3 * i + j] = KR[3 * i] * Kinv[j] + KR[3 * i + 1] * Kinv[3 + j] + KR[3 * i + 2] * Kinv[6 + j];
    for (int i = 0; i < 3; ++i) bearing[i] = M[3 * i] * point_reference[0] + M[3 * i + 1] * point_reference[1] + M[3 * i + 2];
    const double a[3] = {bearing[0] + max_idepth * Kt[0], bearing[1] + max_idepth * Kt[1], bearing[2] + max_idepth * Kt[2]};
    const double b[3] = {bearing[0] + kZeroIdepthEps * Kt[0], bearing[1] + kZeroIdepthEps * Kt[1], bearing[2] + kZeroIdepthEps * Kt[2]};
    const double dx = a[0] / a[2] - b[0] / b[2], dy = a[1] / a[2] - b[1] / b[2];
    use_x_direction = dx * dx > dy * dy;
  }

  double getInverseDepth(const double p[2]) const {
    double idepth;
    const double x_divider = Kt[0] - Kt[2] * p[0];
    const double y_divider = Kt[1] - Kt[2] * p[1];
    if (use_x_direction || std::abs(y_divider) < kZeroIdepthEps)
      idepth = (bearing[2] * p[0] - bearing[0]) / x_divider;
    else
      idepth = (bearing[2] * p[1] - bearing[1]) / y_divider;
    if (std::abs(idepth - kMaxIdepth) < kZeroIdepthEps) idepth = kMaxIdepth;
    if (std::abs(idepth) < kZeroIdepthEps) idepth = 0;
    return idepth;
  }
};

/** EpipolarLine — epipolar_line.cpp:1-82 */
struct EpipolarLine {
  struct Point {
    double projection[2];
    double reference_idepth, target_idepth;
  };
  std::vector<Point> points;
  double length_ = -1;

  bool empty() const { return points.empty(); }
  void addPoint(const double p[2], double idepth, double target_idepth) { points.push_back(Point{{p[0], p[1]}, idepth, target_idepth}); }
  static double dist(const Point &a, const Point &b) { return std::hypot(a.projection[0] - b.projection[0], a.projection[1] - b.projection[1]); }
  double length() {
    if (length_ < 0) {
      length_ = 0;
      for (size_t i = 1; i < points.size(); i++) length_ += dist(points[i], points[i - 1]);
    }
    return length_;
  }
  void tangent(size_t idx, double out[2]) const {
    const int left = std::max(0, static_cast<int>(idx) - 1);
    const int right = std::min(static_cast<int>(points.size()) - 1, static_cast<int>(idx) + 1);
    out[0] = points[static_cast<size_t>(right)].projection[0] - points[static_cast<size_t>(left)].projection[0];
    out[1] = points[static_cast<size_t>(right)].projection[1] - points[static_cast<size_t>(left)].projection[1];
  }
  static Point lerp(const Point &p0, double alpha, const Point &a, const Point &b) {  // p0 + alpha * (a - b)
    return Point{{p0.projection[0] + alpha * (a.projection[0] - b.projection[0]), p0.projection[1] + alpha * (a.projection[1] - b.projection[1])},
                 p0.reference_idepth + alpha * (a.reference_idepth - b.reference_idepth),
                 p0.target_idepth + alpha * (a.target_idepth - b.target_idepth)};
  }
  Point shift(size_t idx, double step) const {  // :18-57
    if (idx == 0)
      step /= dist(points[idx], points[idx + 1]);
    else if (idx == points.size() - 1)
      step /= dist(points[idx - 1], points[idx]);
    else
      step /= dist(points[idx - 1], points[idx + 1]) / 2;  // ((p[i-1] - p[i+1]) / 2).norm()
    const int idx_step = static_cast<int>(std::round(step));
    double subpixel_shift = step - static_cast<double>(idx_step);
    const int idx_signed = static_cast<int>(idx) + idx_step;
    if (idx_signed <= 0) {
      const double alpha = step + static_cast<double>(idx);
      return lerp(points[0], alpha, points[1], points[0]);
    }
    if (idx_signed >= static_cast<int>(points.size()) - 1) {
      const double alpha = step + static_cast<double>(idx) - (static_cast<double>(points.size()) - 1);
      return lerp(points.back(), alpha, points.back(), points[points.size() - 2]);
    }
    const int neighbour = subpixel_shift > 0 ? 1 : -1;
    subpixel_shift = std::abs(subpixel_shift);
    const Point &pt1 = points[static_cast<size_t>(idx_signed)], &pt2 = points[static_cast<size_t>(idx_signed + neighbour)];
    const double a1 = 1 - subpixel_shift, a2 = subpixel_shift;
    return Point{{a1 * pt1.projection[0] + a2 * pt2.projection[0], a1 * pt1.projection[1] + a2 * pt2.projection[1]},
                 a1 * pt1.reference_idepth + a2 * pt2.reference_idepth, a1 * pt1.target_idepth + a2 * pt2.target_idepth};
  }
};

/** EpipolarLineBuilder<PinholeCamera, SE3> — epipolar_line_builder_pinhole_se3.hpp:22-390 */
struct EpipolarLineBuilder {
  const PinholeModel &model;
  const SE3 &t_t_r;
  static constexpr double kMaxIdepth = 1000;  // epipolar_line_builder.hpp:36
  EpipolarLineBuilder(const PinholeModel &m, const SE3 &t) : model(m), t_t_r(t) {}

  static bool createGeneralLine(double k, double s, double W, double H, double ps[2], double pe[2], double border) {  // :44-85
    if (k == 0) {
      ps[0] = border;
      ps[1] = s;
      pe[0] = W - 1 - border;
      pe[1] = s;
      return s >= border && s <= (H - 1 - border);
    }
    const double x_min = border, y_min = border, x_max = W - 1 - border, y_max = H - 1 - border;
    auto y_x = [=](double x) { return k * x + s; };
    auto x_y = [=](double y) { return y / k - s / k; };
    double y_start = std::clamp(y_x(x_min), y_min, y_max), y_end = std::clamp(y_x(x_max), y_min, y_max);
    const double x_start = std::clamp(x_y(y_min), x_min, x_max), x_end = std::clamp(x_y(y_max), x_min, x_max);
    if (k < 0) std::swap(y_start, y_end);
    ps[0] = x_start;
    ps[1] = y_start;
    pe[0] = x_end;
    pe[1] = y_end;
    if (y_x(x_min) < border && y_x(x_max) < border) return false;
    if (y_x(x_min) > (H - 1 - border) && y_x(x_max) > (H - 1 - border)) return false;
    if (x_y(y_min) < border && x_y(y_max) < border) return false;
    if (x_y(y_min) > (W - 1 - border) && x_y(y_max) > (W - 1 - border)) return false;
    return true;
  }
  static bool intersectImageBorders(const double p1[2], const double p2[2], double W, double H, double left[2], double right[2], double border) {  // :87-108
    const double a = p1[1] - p2[1], b = p2[0] - p1[0], c = p1[0] * p2[1] - p2[0] * p1[1];
    if (a == 0 && b == 0) return p1[0] >= border && p1[1] >= border && p1[0] <= (W - 1 - border) && p1[1] <= (H - 1 - border);
    if (b == 0) {  // createVerticalLine :34-40
      left[0] = p1[0];
      left[1] = border;
      right[0] = p1[0];
      right[1] = H - 1 - border;
      return p1[0] >= border && p1[0] <= (W - 1 - border);
    }
    return createGeneralLine(-(a / b), -(c / b), W, H, left, right, border);
  }
  bool valid(const double p[2], double idepth_1) const {  // :118-138
    const double kDepthEps = 1e-4;
    if (idepth_1 < 0 || idepth_1 > (1 / PinholeModel::kMinDepth + kDepthEps)) return false;
    const double d[3] = {(p[0] - model.cx) * (1 / model.fx), (p[1] - model.cy) * (1 / model.fy), 1};
    double R[9];
    t_t_r.rotation(R);
    const double z = R[6] * d[0] + R[7] * d[1] + R[8] * d[2] + t_t_r.t[2] * idepth_1;
    const double idepth_2 = 1 / z;
    return idepth_2 >= 0 && idepth_1 <= (1 / PinholeModel::kMinDepth + kDepthEps);
  }
  bool reproject(const ArrayReprojector<true> &rp, const double p[2], double idepth, double out[2]) const {
    return rp.reprojectPattern<1>(&p[0], &p[1], idepth, &out[0], &out[1]);
  }

  EpipolarLine buildSegment(const double observed[2], double idepthmin, double idepthmax) const {  // :296-372
    EpipolarLine line;
    if (std::sqrt(t_t_r.t[0] * t_t_r.t[0] + t_t_r.t[1] * t_t_r.t[1] + t_t_r.t[2] * t_t_r.t[2]) < 1. / kMaxIdepth) return line;
    double start[2], end[2], left[2] = {0, 0}, right[2] = {0, 0};
    double lim0 = kMaxIdepth, lim1 = 0;  // idepth_limits
    const ArrayReprojector<true> rp(model, model, t_t_r);
    const bool zero_depth_reprojected = reproject(rp, observed, lim0, start);
    const bool inf_depth_reprojected = reproject(rp, observed, lim1, end);
    const bool zero_valid = valid(observed, lim0), inf_valid = valid(observed, lim1);
    const bool limits_diff_validity = zero_valid != inf_valid;
    const EpipolarLineTriangulatorSE3 tri(t_t_r, model, observed, kMaxIdepth);
    const bool intersect = intersectImageBorders(start, end, model.width, model.height, left, right, PinholeModel::kBorderSize);
    const double border0 = tri.getInverseDepth(left), border1 = tri.getInverseDepth(right);
    const bool left_valid = valid(observed, border0), right_valid = valid(observed, border1);
    const bool borders_diff_validity = left_valid != right_valid;
    // one point epipolar line — isApprox: |a - b|^2 <= eps^2 * min(|a|^2, |b|^2), eps = 1e-12 for double
    {
      const double d2 = (start[0] - end[0]) * (start[0] - end[0]) + (start[1] - end[1]) * (start[1] - end[1]);
      const double na = start[0] * start[0] + start[1] * start[1], nb = end[0] * end[0] + end[1] * end[1];
      if (d2 <= 1e-12 * 1e-12 * std::min(na, nb)) {
        if (left_valid) line.points.push_back(EpipolarLine::Point{{start[0], start[1]}, 0, 0});
        return line;
      }
    }
    // findLineBorders :140-201
    auto set2 = [](double d[2], const double s[2]) {
      d[0] = s[0];
      d[1] = s[1];
    };
    auto dir_dot = [&]() { return (end[0] - start[0]) * (right[0] - left[0]) + (end[1] - start[1]) * (right[1] - left[1]); };
    if (limits_diff_validity) {
      if (zero_depth_reprojected && !inf_depth_reprojected) {
        if (dir_dot() > 0 && border0 > 0) {
          set2(end, left);
          lim1 = border0;
        } else {
          set2(end, right);
          lim1 = border1;
        }
      }
      if (inf_depth_reprojected && !zero_depth_reprojected) {
        if (dir_dot() > 0 && border1 > 0) {
          set2(start, right);
          lim0 = border1;
        } else {
          set2(start, left);
          lim0 = border0;
        }
      }
    } else {
      if (zero_depth_reprojected && !inf_depth_reprojected) {
        if (dir_dot() > 0 && border1 > 0) {
          set2(end, right);
          lim1 = border1;
        } else {
          set2(end, left);
          lim1 = border0;
        }
      }
      if (inf_depth_reprojected && !zero_depth_reprojected) {
        if (dir_dot() > 0 && border0 > 0) {
          set2(start, left);
          lim0 = border0;
        } else {
          set2(start, right);
          lim0 = border1;
        }
      }
    }
    if (!zero_depth_reprojected && !inf_depth_reprojected) {
      set2(start, left);
      set2(end, right);
      lim0 = border0;
      lim1 = border1;
    }
    if (lim1 > lim0) {  // make it always rising
      std::swap(lim0, lim1);
      std::swap(start[0], end[0]);
      std::swap(start[1], end[1]);
    }
    // epipolarLineNotExists :203-234
    if (!intersect) return line;
    if (limits_diff_validity && !zero_depth_reprojected && !inf_depth_reprojected && borders_diff_validity) return line;
    if (!zero_depth_reprojected && !inf_depth_reprojected && !left_valid && !right_valid) return line;
    if (lim1 > idepthmax || lim0 < idepthmin) return line;
    if (idepthmax < idepthmin) return line;
    if (lim0 > idepthmax) {
      lim0 = idepthmax;
      reproject(rp, observed, lim0, start);
    }
    if (lim1 < idepthmin) {
      lim1 = idepthmin;
      reproject(rp, observed, lim1, end);
    }
    // getSize :110-116, makePinholeCameraEpipolarLine(size, point_end_depth, point_start_depth, ...) :22-32
    const size_t size = std::max<size_t>(1, static_cast<size_t>(std::hypot(end[0] - start[0], end[1] - start[1]) * 1.0));
    const double step[2] = {(start[0] - end[0]) / static_cast<double>(size), (start[1] - end[1]) / static_cast<double>(size)};
    double p[2] = {end[0] - step[0], end[1] - step[1]};
    for (size_t pos = 0; pos <= size; ++pos) {
      p[0] += step[0];
      p[1] += step[1];
      line.addPoint(p, tri.getInverseDepth(p), 0);
    }
    return line;
  }
};

/** what DepthEstimation::estimate gets besides the landmarks */
struct DepthEstimationFrame {
  PixelMapView target;  // level 0 of the new frame
  MaskView mask;
  PinholeModel model;
  SE3 t_t_r;
  double reference_exposure_time = 1, target_exposure_time = 1;
  double reference_affine[2] = {0, 0}, target_affine[2] = {0, 0};
  double sigma_huber_loss = 20;
};

/** estimateLandmark — depth_estimation.cpp:223-357 (findBest :36-76, refine :184-221, DepthEstimationProblem :80-181,
 *  the LM loop of levenberg_marquardt_algorithm.hpp:77-128 with the stop() test of this problem) */
inline void estimateLandmark(const DepthEstimationFrame &f, ImmatureLandmark &lm) {
  const double kMinEpilineSize = 2, kMinDepthScale = 0.75, kMaxDepthScale = 1.5, kMaxError = 10, kErrorStep = 10;
  const size_t kUniquenessRadius = 2, kMinEpilineSizeForUniqueness = 10;
  const double kMaxEnergyForInliers = kPatternSize * 12.0 * 12.0, kEps = 1e-10;
  const double kMaxPixSearch = (f.model.width + f.model.height) * 0.027;
  if (lm.status == kOutOfBoundary || lm.status == kDelete || lm.status == kImmatureOutlier) return;
  const double *coords = lm.projection;
  const EpipolarLineBuilder builder(f.model, f.t_t_r);
  EpipolarLine epiline = builder.buildSegment(coords, lm.idepth_min, lm.idepth_max);
  if (epiline.empty()) {
    lm.search_pixel_interval = 0;
    lm.setStatus(kOutOfBoundary);
    return;
  }
  const double search_distance = epiline.length();
  if (search_distance < kMinEpilineSize) {
    lm.search_pixel_interval = search_distance;
    lm.setStatus(kSkipped);
    return;
  }
  double M[12];
  f.t_t_r.matrix3x4(M);
  const double depth_scale = M[8] * lm.direction[0] + M[9] * lm.direction[1] + M[10] * lm.direction[2] + M[11] * epiline.points.front().reference_idepth;
  if (lm.idepth_min >= 0 && (depth_scale < kMinDepthScale || depth_scale > kMaxDepthScale)) {
    lm.search_pixel_interval = 0;
    lm.setStatus(kOutOfBoundary);
    return;
  }
  const size_t n = epiline.points.size();
  size_t optimum = 0;
  double best_energy = 1e6;
  std::vector<double> energies(n, std::numeric_limits<double>::max());
  size_t distance = n;
  if (!lm.traced) distance = std::min(distance, static_cast<size_t>(kMaxPixSearch / search_distance * static_cast<double>(n)));
  // findBest :36-76
  const ArrayReprojector<true> rp(f.model, f.model, f.t_t_r);
  double ref_u[kPatternSize], ref_v[kPatternSize], precalc[kPatternSize];
  const double scale = (f.target_exposure_time / f.reference_exposure_time) * std::exp(f.target_affine[0] - f.reference_affine[0]);
  for (int k = 0; k < kPatternSize; ++k) {
    ref_u[k] = coords[0] + kPatternData[2 * k];
    ref_v[k] = coords[1] + kPatternData[2 * k + 1];
    precalc[k] = scale * (lm.patch[k] - f.reference_affine[1]);
  }
  for (size_t idx = 0; idx < distance; ++idx) {
    const EpipolarLine::Point &pt = epiline.points[idx];
    double tu[kPatternSize], tv[kPatternSize];
    bool success = rp.reprojectPattern<kPatternSize>(ref_u, ref_v, pt.reference_idepth, tu, tv);
    success = success && f.mask.valid(pt.projection[0], pt.projection[1]);
    if (success) {
      double energy = 0;
      for (int k = 0; k < kPatternSize; ++k) {
        const double r = (interpolateLinear1(f.target, tu[k], tv[k]) - f.target_affine[1]) - precalc[k];
        energy += r * r;
      }
      energies[idx] = energy;
      if (energy < best_energy) {
        best_energy = energy;
        optimum = idx;
      }
    }
  }
  double second_best = std::numeric_limits<double>::max();
  for (size_t idx = 0; idx < n; ++idx)
    if ((idx + kUniquenessRadius < optimum || idx > optimum + kUniquenessRadius) && energies[idx] < second_best) second_best = energies[idx];
  lm.setUniqueness(second_best / best_energy, search_distance > static_cast<double>(kMinEpilineSizeForUniqueness));
  double epiline_vector[2];
  epiline.tangent(optimum, epiline_vector);
  // stableNormalized
  double tangent[2];
  {
    const double w = std::max(std::abs(epiline_vector[0]), std::abs(epiline_vector[1]));
    if (w > 0) {
      const double a = epiline_vector[0] / w, b = epiline_vector[1] / w, nn = std::sqrt(a * a + b * b);
      tangent[0] = a / nn;
      tangent[1] = b / nn;
    } else {
      tangent[0] = epiline_vector[0];
      tangent[1] = epiline_vector[1];
    }
  }
  // refine :184-221
  double pat_u[kPatternSize], pat_v[kPatternSize];
  if (!rp.reprojectPattern<kPatternSize>(ref_u, ref_v, epiline.points[optimum].reference_idepth, pat_u, pat_v)) {
    lm.search_pixel_interval = 0;
    lm.setStatus(kOutOfBoundary);
    return;
  }
  {
    const double sigma = f.sigma_huber_loss;
    double old_u[kPatternSize], old_v[kPatternSize];
    double hessian = 0, b = 0, step = 0;
    bool stop = false;
    auto calculateEnergy = [&]() {
      double e = 0;
      for (int k = 0; k < kPatternSize; ++k) {
        const double r = (interpolateLinear1(f.target, pat_u[k], pat_v[k]) - f.target_affine[1]) - precalc[k];
        e += std::max(std::min(r, sigma), -sigma) * r;
      }
      return e;
    };
    auto linearize = [&]() {
      hessian = 0;
      b = 0;
      for (int k = 0; k < kPatternSize; ++k) {
        double s3[3];
        interpolateLinear3(f.target, pat_u[k], pat_v[k], s3);
        const double r = (s3[0] - f.target_affine[1]) - precalc[k];
        const double w = sigma * (1.0 / std::max(std::abs(r), sigma));
        const double d = tangent[0] * s3[1] + tangent[1] * s3[2];
        hessian += w * (d * d);
        b += w * (r * d);
      }
    };
    auto calculateStep = [&](double lambda) {
      step = b / (hessian + hessian * lambda);
      step = std::clamp(step, -0.3, 0.3);
      for (int k = 0; k < kPatternSize; ++k) {
        old_u[k] = pat_u[k];
        old_v[k] = pat_v[k];
        pat_u[k] -= step * tangent[0];
        pat_v[k] -= step * tangent[1];
      }
      if (!f.model.insideCameraROI(pat_u, pat_v, kPatternSize)) {
        for (int k = 0; k < kPatternSize; ++k) {
          pat_u[k] = old_u[k];
          pat_v[k] = old_v[k];
        }
        stop = true;
      }
    };
    auto rejectStep = [&]() {
      for (int k = 0; k < kPatternSize; ++k) {
        pat_u[k] = old_u[k];
        pat_v[k] = old_v[k];
      }
    };
    for (int k = 0; k < kPatternSize; ++k) {
      old_u[k] = pat_u[k];
      old_v[k] = pat_v[k];
    }
    // levenberg_marquardt_algorithm::solve with options {lambda0 2, ftol 0, ptol 1e-1, 3 iterations, /2, x2} (:191-197)
    double lambda = 2.0, energy = calculateEnergy();
    bool converged = false, linear_system_valid = false;
    for (int it = 0; it < 3 && !converged; ++it) {
      if (!linear_system_valid) linearize();
      calculateStep(lambda);
      const double next = calculateEnergy();
      if (stop) {
        rejectStep();
        break;
      }
      converged |= std::abs(energy - next) / energy < 0.0;
      if (next < energy) {
        converged |= step * step < 1e-1 * (0 + 1e-1);
        energy = next;
        lambda /= 2.0;
        linear_system_valid = false;
      } else {
        rejectStep();
        lambda *= 2.0;
        linear_system_valid = true;
      }
    }
    best_energy = energy;
  }
  const double subpixel_optimum[2] = {pat_u[4], pat_v[4]};  // Pattern::kCenter = 4
  const double shift_vec[2] = {subpixel_optimum[0] - epiline.points[optimum].projection[0], subpixel_optimum[1] - epiline.points[optimum].projection[1]};
  double shift = std::hypot(shift_vec[0], shift_vec[1]);
  if (shift_vec[0] * epiline_vector[0] + shift_vec[1] * epiline_vector[1] < 0) shift *= -1;
  if (best_energy > kMaxEnergyForInliers) {
    lm.search_pixel_interval = 0;
    lm.setStatus(kImmatureOutlier);
    return;
  }
  // calculateError :26-33
  double error;
  {
    const double ox = epiline_vector[1], oy = -epiline_vector[0];
    const double a = std::pow(epiline_vector[0] * lm.gradient[0] + epiline_vector[1] * lm.gradient[1], 2.0);
    const double bb = std::pow(ox * lm.gradient[0] + oy * lm.gradient[1], 2.0);
    error = 0.2 + 0.2 * (a + bb) / a;
  }
  if (error > search_distance / 2 && lm.traced) {
    lm.search_pixel_interval = search_distance;
    lm.setStatus(kIllConditioned);
    return;
  }
  error = std::min(error, kMaxError);
  double idepth_min = -1, idepth_max = -1;
  const EpipolarLineTriangulatorSE3 tri(f.t_t_r, f.model, coords, EpipolarLineBuilder::kMaxIdepth);
  const double error_step = error / kErrorStep;
  while ((!f.model.validIdepth(idepth_min) || !f.model.validIdepth(idepth_max)) && error > -kEps) {
    const EpipolarLine::Point right = epiline.shift(optimum, -error + shift), left = epiline.shift(optimum, error + shift);
    idepth_min = tri.getInverseDepth(right.projection);
    idepth_max = tri.getInverseDepth(left.projection);
    error -= error_step;
  }
  if (!f.model.validIdepth(idepth_min) || !f.model.validIdepth(idepth_max)) {
    lm.search_pixel_interval = 0;
    lm.setStatus(kOutOfBoundary);
    return;
  }
  if (idepth_min > idepth_max) std::swap(idepth_min, idepth_max);
  lm.idepth_min = idepth_min;
  lm.idepth_max = idepth_max;
  lm.search_pixel_interval = 2 * error_step * kErrorStep;
  lm.setStatus(kGood);
}

}  // namespace oracle
