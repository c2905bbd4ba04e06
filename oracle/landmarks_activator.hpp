// ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (the reference cannot be built or imported here and ships no vectors
// for this path).  CPU restatement of the activation of immature landmarks (row f-3 of SURVEY.md §8):
//   LandmarksActivator::activate, reprojectActivePoints, activationStatus, haveNoNeighbors, recalculateMinDistanceToNeighbor,
//   LandmarkActivationProblem, optimizeImmatureLandmark(s)
//       — src/tracker/landmarks_activator/src/landmarks_activator.cpp:29-391
//   ImmatureTrackingLandmark::readyForActivation — src/track/landmarks/src/immature_tracking_landmark.cpp:46-52
//   ActiveKeyframe::applyImmatureLandmarkActivationStatuses — src/track/frames/src/active_keyframe.cpp:209-239
// The product path never links or calls it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

#include "depth_estimation.hpp"
#include "geometry.hpp"
#include "pba.hpp"  // lmSolve
#include "se3.hpp"

namespace oracle {

/** ActiveKeyframe::ImmatureLandmarkActivationStatus — active_keyframe.hpp:40-44 */
enum ActivationStatus : uint8_t { kActivate = 0, kSkipActivation = 1, kDeleteLandmark = 2 };

/** one keyframe of track.activeFrames() as the activator reads it */
struct ActivationKeyframe {
  SE3 t_world_agent;
  double exposure_time = 1;
  double affine_brightness[2] = {0, 0};
  PixelMapView level0;   // getLevel(sensor, 0)
  MaskView mask0;        // getMask(sensor, 0)
  MaskView mask_sparsity;  // getMask(sensor, kLevelToMaintainSparsity) (read for the newest keyframe only)
  // activeLandmarks(sensor)
  int n_active = 0;
  const double *active_uv = nullptr;
  const double *active_idepth = nullptr;
  const uint8_t *active_skip = nullptr;  // isOutlier() || isMarginalized()
  // immatureLandmarks(sensor)
  std::vector<ImmatureLandmark> immature;
};

/** ImmatureTrackingLandmark::readyForActivation — immature_tracking_landmark.cpp:46-52 */
inline bool readyForActivation(const ImmatureLandmark &l) {
  const double kMaxSearchPixelInterval = 8;
  const double kMinUniqueness = 3;
  const double idepth = l.idepth_max * 0.5 + l.idepth_min * 0.5;  // idepth(), :24
  return (l.status == kGood || l.status == kSkipped || l.status == kIllConditioned || l.status == kOutOfBoundary) &&
         (l.search_pixel_interval < kMaxSearchPixelInterval) && (l.uniqueness > kMinUniqueness) && (idepth > 0);
}

/** recalculateMinDistanceToNeighbor — landmarks_activator.cpp:29-39 */
inline void recalculateMinDistanceToNeighbor(size_t number_of_active_points, size_t number_of_desired_points, double &min_distance_to_neighbor) {
  const double kPRegulatorCoefficient = 0.001;
  min_distance_to_neighbor += (static_cast<double>(number_of_active_points) - static_cast<double>(number_of_desired_points)) * kPRegulatorCoefficient;
  min_distance_to_neighbor = std::clamp(min_distance_to_neighbor, 0.0, 10.0);
}

struct Point2 {
  double x, y;
};

/** haveNoNeighbors — :41-49 */
inline bool haveNoNeighbors(const Point2 &p, const std::vector<Point2> &reprojected, double distance) {
  for (const Point2 &q : reprojected) {
    const double dx = q.x - p.x, dy = q.y - p.y;
    if (std::sqrt(dx * dx + dy * dy) < distance) return false;
  }
  return true;
}

/** reprojectActivePoints — :51-87 */
inline size_t reprojectActivePoints(const std::vector<ActivationKeyframe> &frames, const PinholeModel &model, std::vector<Point2> &reprojected,
                                    int level_to_maintain_sparsity) {
  size_t number_of_active_points = 0;
  const ActivationKeyframe &last = frames.back();
  const double scale = static_cast<double>(1 << level_to_maintain_sparsity);
  for (size_t f = 0; f + 1 < frames.size(); ++f) {
    const ActivationKeyframe &frame = frames[f];
    const SE3 t_t_r = last.t_world_agent.inverse() * frame.t_world_agent;
    const ArrayReprojector<true> reprojector(model, model, t_t_r);
    for (int i = 0; i < frame.n_active; ++i) {
      if (frame.active_skip[i]) continue;
      number_of_active_points++;
      const double u = frame.active_uv[2 * i] / scale, v = frame.active_uv[2 * i + 1] / scale;
      double tu, tv;
      if (!reprojector.reprojectPattern<1>(&u, &v, frame.active_idepth[i], &tu, &tv)) continue;
      if (!last.mask_sparsity.valid(tu, tv)) continue;
      reprojected.push_back({tu, tv});
    }
  }
  return number_of_active_points;
}

/** activationStatus — :89-126 */
inline uint8_t activationStatus(const ImmatureLandmark &landmark, const ArrayReprojector<true> &reprojector, std::vector<Point2> &reprojected,
                                double min_distance_to_neighbor, const MaskView &target_mask, int level_to_maintain_sparsity) {
  if (landmark.status == kDelete) return kDeleteLandmark;
  if (!landmark.traced || landmark.status == kImmatureOutlier) return kDeleteLandmark;
  if (!readyForActivation(landmark)) return landmark.status == kOutOfBoundary ? kDeleteLandmark : kSkipActivation;
  const double scale = static_cast<double>(1 << level_to_maintain_sparsity);
  const double u = landmark.projection[0] / scale, v = landmark.projection[1] / scale;
  const double idepth = landmark.idepth_max * 0.5 + landmark.idepth_min * 0.5;
  double tu, tv;
  if (!reprojector.reprojectPattern<1>(&u, &v, idepth, &tu, &tv)) return kDeleteLandmark;
  if (!target_mask.valid(tu, tv)) return kDeleteLandmark;
  if (haveNoNeighbors({tu, tv}, reprojected, min_distance_to_neighbor)) {
    reprojected.push_back({tu, tv});
    return kActivate;
  }
  return kSkipActivation;
}

/** LandmarkActivationProblem — :128-277 */
struct LandmarkActivationProblem {
  static constexpr double kMaxEnergyForInliers = kPatternSize * 12 * 12;  // :130
  const ImmatureLandmark &landmark;
  size_t reference_idx;
  const std::vector<ActivationKeyframe> &frames;
  const PinholeModel &model;
  double sigma_huber_loss;
  double ref_u[kPatternSize], ref_v[kPatternSize];
  double idepth, old_idepth;
  double hessian = 0, b = 0, step = 0;
  bool stop_ = false;

  LandmarkActivationProblem(const ImmatureLandmark &l, size_t reference, const std::vector<ActivationKeyframe> &f, const PinholeModel &m,
                            double sigma, double idepth0)
      : landmark(l), reference_idx(reference), frames(f), model(m), sigma_huber_loss(sigma), idepth(idepth0), old_idepth(idepth0) {
    for (int k = 0; k < kPatternSize; ++k) {  // PatternPatch::shiftPattern
      ref_u[k] = l.projection[0] + kPatternData[2 * k];
      ref_v[k] = l.projection[1] + kPatternData[2 * k + 1];
    }
  }

  double brightnessChangeScale(const ActivationKeyframe &target) const {  // :166-167
    const ActivationKeyframe &reference = frames[reference_idx];
    return (target.exposure_time / reference.exposure_time) * std::exp(target.affine_brightness[0] - reference.affine_brightness[0]);
  }

  /** :150-202 */
  std::pair<double, int> calculateEnergy() {
    double energy = 0;
    int number_of_valid_residuals = 0;
    if (stop_) {
      idepth = -1;
      return {energy, number_of_valid_residuals};
    }
    const ActivationKeyframe &reference = frames[reference_idx];
    for (size_t t = 0; t < frames.size(); ++t) {
      if (t == reference_idx) continue;
      const ActivationKeyframe &target = frames[t];
      const double scale = brightnessChangeScale(target);
      const ArrayReprojector<true> reprojector(model, model, target.t_world_agent.inverse() * reference.t_world_agent);
      double tu[kPatternSize], tv[kPatternSize];
      bool success = reprojector.reprojectPattern<kPatternSize>(ref_u, ref_v, idepth, tu, tv);
      success = success && target.mask0.valid(tu, tv, kPatternSize);
      if (!success) continue;
      double sq = 0;
      for (int k = 0; k < kPatternSize; ++k) {
        const double r = (interpolateLinear1(target.level0, tu[k], tv[k]) - target.affine_brightness[1]) -
                         scale * (landmark.patch[k] - reference.affine_brightness[1]);
        sq += r * r;
      }
      const double norm = std::sqrt(sq);
      const double huber_weight = norm > sigma_huber_loss ? sigma_huber_loss / norm : 1;
      if (sq < kMaxEnergyForInliers) {
        energy += huber_weight * sq;
        number_of_valid_residuals++;
      } else {
        energy += kMaxEnergyForInliers;
      }
    }
    if (number_of_valid_residuals == 0) {
      idepth = -1;
      stop_ = true;
    }
    return {energy, number_of_valid_residuals};
  }

  /** :204-256 */
  void linearize() {
    hessian = 0;
    b = 0;
    const ActivationKeyframe &reference = frames[reference_idx];
    for (size_t t = 0; t < frames.size(); ++t) {
      if (t == reference_idx) continue;
      const ActivationKeyframe &target = frames[t];
      const double scale = brightnessChangeScale(target);
      const ArrayReprojector<true> reprojector(model, model, target.t_world_agent.inverse() * reference.t_world_agent);
      double tu[kPatternSize], tv[kPatternSize], dui[kPatternSize], dvi[kPatternSize], duT[6 * kPatternSize], dvT[6 * kPatternSize];
      bool success = reprojector.reprojectPattern<kPatternSize>(ref_u, ref_v, idepth, tu, tv, dui, dvi, duT, dvT);
      success = success && target.mask0.valid(tu, tv, kPatternSize);
      if (!success) continue;
      double r[kPatternSize], d[kPatternSize], sq = 0;
      for (int k = 0; k < kPatternSize; ++k) {
        double s[3];
        interpolateLinear3(target.level0, tu[k], tv[k], s);
        r[k] = (s[0] - target.affine_brightness[1]) - scale * (landmark.patch[k] - reference.affine_brightness[1]);
        sq += r[k] * r[k];
        d[k] = s[1] * dui[k] + s[2] * dvi[k];
      }
      const double norm = std::sqrt(sq);
      const double huber_weight = norm > sigma_huber_loss ? sigma_huber_loss / norm : 1;
      double dd = 0, dr = 0;
      for (int k = 0; k < kPatternSize; ++k) {
        dd += (huber_weight * d[k]) * d[k];
        dr += (huber_weight * d[k]) * r[k];
      }
      hessian += dd;
      b += dr;
    }
    if (hessian == 0) stop_ = true;
  }
  /** :258-262 */
  void calculateStep(double levenberg_marquardt_regularizer) {
    step = b / (hessian + hessian * levenberg_marquardt_regularizer);
    old_idepth = idepth;
    idepth -= step;
  }
  std::pair<double, double> acceptStep() { return {idepth * idepth, step * step}; }  // :264
  void rejectStep() { idepth = old_idepth; }                                           // :266
};

/** optimizeImmatureLandmark — :279-311 */
inline uint8_t optimizeImmatureLandmark(ImmatureLandmark &landmark, size_t reference_idx, const std::vector<ActivationKeyframe> &frames,
                                        const PinholeModel &model, int minimum_inliers, double sigma_huber_loss) {
  LmOptions options;
  options.initial_levenberg_marquardt_regularizer = 1. / 10;
  options.function_tolerance = 0;
  options.parameter_tolerance = 1e-8;
  options.max_num_iterations = 3;
  options.levenberg_marquardt_regularizer_decrease_on_accept = 2.;
  options.levenberg_marquardt_regularizer_increase_on_reject = 5.;
  LandmarkActivationProblem problem(landmark, reference_idx, frames, model, sigma_huber_loss, landmark.idepth_max * 0.5 + landmark.idepth_min * 0.5);
  const LmResult result = lmSolve(problem, options);
  if (result.number_of_valid_residuals < minimum_inliers || problem.idepth < 0) return kDeleteLandmark;
  landmark.idepth_min = problem.idepth;
  landmark.idepth_max = problem.idepth;
  return kActivate;
}

struct ActivationResult {
  size_t number_of_active_points = 0;
  std::vector<std::vector<uint8_t>> statuses;  // one vector per keyframe but the newest
};

/**
 * LandmarksActivator<SE3, PinholeCamera, PixelMap, 1, REFINE>::activate — :351-391, followed by
 * ActiveKeyframe::applyImmatureLandmarkActivationStatuses for the part of it that touches the immature landmarks
 * (status := kDelete for activated and deleted landmarks, active_keyframe.cpp:232-235).
 * `frames` = track.activeFrames(), oldest first, the newest keyframe last; `model` = level-0 camera.
 */
inline ActivationResult activateLandmarks(std::vector<ActivationKeyframe> &frames, const PinholeModel &model, double sigma_huber_loss,
                                          size_t number_of_desired_points, double &min_distance_to_neighbor, bool refine) {
  const int kLevelToMaintainSparsity = 1;
  const int kMinimumInliers = 1;
  ActivationResult out;
  const PinholeModel sparsity_model = model.scaled(kLevelToMaintainSparsity);
  std::vector<Point2> reprojected;
  out.number_of_active_points = reprojectActivePoints(frames, sparsity_model, reprojected, kLevelToMaintainSparsity);
  recalculateMinDistanceToNeighbor(out.number_of_active_points, number_of_desired_points, min_distance_to_neighbor);
  const ActivationKeyframe &last = frames.back();
  for (size_t f = 0; f + 1 < frames.size(); ++f) {
    const ArrayReprojector<true> reprojector(sparsity_model, sparsity_model, last.t_world_agent.inverse() * frames[f].t_world_agent);
    std::vector<uint8_t> st;
    for (const ImmatureLandmark &l : frames[f].immature)
      st.push_back(activationStatus(l, reprojector, reprojected, min_distance_to_neighbor, last.mask_sparsity, kLevelToMaintainSparsity));
    out.statuses.push_back(std::move(st));
  }
  if (refine) {  // optimizeImmatureLandmarks — :313-338
    const int minimum_inliers = std::min(kMinimumInliers, static_cast<int>(frames.size()) - 1);
    for (size_t f = 0; f + 1 < frames.size(); ++f)
      for (size_t i = 0; i < frames[f].immature.size(); ++i) {
        uint8_t &status = out.statuses[f][i];
        if (status != kActivate) continue;
        status = optimizeImmatureLandmark(frames[f].immature[i], f, frames, model, minimum_inliers, sigma_huber_loss);
      }
  }
  for (size_t f = 0; f + 1 < frames.size(); ++f)  // applyImmatureLandmarkActivationStatuses
    for (size_t i = 0; i < frames[f].immature.size(); ++i)
      if (out.statuses[f][i] == kActivate || out.statuses[f][i] == kDeleteLandmark) frames[f].immature[i].setStatus(kDelete);
  return out;
}

}  // namespace oracle
