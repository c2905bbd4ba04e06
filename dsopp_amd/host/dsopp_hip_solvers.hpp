// Host-side C++ mirror of the reference's solver interfaces over the C-ABI (include/dsopp_hip.h).
//
// The reference selects its solvers through two abstract class templates:
//   energy::problem::PhotometricBundleAdjustment<Precision, SE3, PinholeCamera, 8, PixelMap, true, true, true, 1>
//       virtuals  pushFrame(const ActiveKeyframe&, size_t level, const Model&, FrameParameterization)      PBA_INC/photometric_bundle_adjustment.hpp:55-56
//                 updateLocalFrame(const ActiveKeyframe&)                                                   :127
//                 solve(size_t number_of_threads) -> Precision                                              :154
//       services  updateFrame(ActiveKeyframe&), getPose(time), getAffineBrightness(time)                   PROB_SRC/photometric_bundle_adjustment.cpp:156-264
//   energy::problem::PoseAlignment<SE3, PinholeCamera, 1, PixelMap, 1>                                      PA_INC/pose_alignment.hpp:24-63
//       virtuals  solve -> rmse | kZeroCost, reset(), pushKnownPose(time, Motion), setRotationPrior(Matrix3)
// Those headers drag in Eigen / Sophus / glog / Ceres / OpenCV, none of which exist in this build environment, so this
// file states the SAME interface (names, argument meaning, error behaviour) over small value types that carry exactly
// what the solvers read from `track::ActiveKeyframe` / `sensors::calibration::CameraCalibration`.  INTEGRATION.md shows
// the ~60-line adapter that derives from the reference's real base classes and forwards to these.
//
// Ownership mirrors the reference: the solver copies what it needs at pushFrame, image pyramids are borrowed (they must
// outlive the frame's stay in the solver, as the keyframe's PixelMap does there), single-threaded use per object.
#pragma once
#include <array>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include <limits>
#include <functional>

#include "../../include/dsopp_hip.h"

namespace dsopp_hip_host {

using time_point = int64_t;                 // dsopp::time (time_point of the keyframe), as integer ticks
using Motion = std::array<double, 7>;       // energy::motion::SE3<Precision>: Sophus storage (qx, qy, qz, qw, tx, ty, tz)
using Vector2 = std::array<double, 2>;
using Matrix6 = std::array<double, 36>;

enum class FrameParameterization { kFree = 0, kFixed = 1 };  // PBA_INC/frame_parameterization.hpp:9-12

/** energy::model::PinholeCamera<Precision> at a pyramid level (focal lengths, principal point), pinhole_camera.hpp:21-200 */
struct PinholeModel {
  double fx, fy, cx, cy;
};

struct SolverError : std::runtime_error {
  int code;
  SolverError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
  // the reference aborts through glog CHECK on contract violations; a library cannot, so violations throw
  if (rc != DSOPP_HIP_OK) throw SolverError(rc, dsopp_hip_last_error());
}

/** device-resident pyramid of one frame = features::PixelDataFrame / ActiveKeyframe::pyramids() + masks */
class DevicePyramid {
 public:
  DevicePyramid(int width, int height, int levels, int device = 0, void *stream = nullptr, int dtype = DSOPP_HIP_F64) {
    check(dsopp_hip_pyramid_create(device, stream, width, height, levels, dtype, &p_));
  }
  ~DevicePyramid() { dsopp_hip_pyramid_destroy(p_); }
  DevicePyramid(const DevicePyramid &) = delete;
  DevicePyramid &operator=(const DevicePyramid &) = delete;
  /** PixelDataFrame(image, photometric_calibration, vignetting, levels) */
  void build(const uint8_t *image, const double *photometric_calibration256 = nullptr, const uint8_t *vignetting = nullptr) {
    check(dsopp_hip_pyramid_build(p_, image, photometric_calibration256, vignetting));
  }
  /** adopt a PixelMap<1> level built on the host by the reference */
  void setLevel(int level, const double *pixelinfo) { check(dsopp_hip_pyramid_set_level(p_, level, pixelinfo)); }
  void setMask(int level, const uint8_t *mask) { check(dsopp_hip_pyramid_set_mask(p_, level, mask)); }
  dsopp_hip_pyramid *handle() const { return p_; }

 private:
  dsopp_hip_pyramid *p_ = nullptr;
};

/** the same image on every device of a multi-device solver (HipPhotometricBundleAdjustment with a `devices` list): one
 *  device-resident copy per distinct device, each built on its own device */
class DevicePyramidGroup {
 public:
  DevicePyramidGroup(dsopp_hip_window_group *group, int width, int height, int levels) {
    check(dsopp_hip_pyramid_group_create(group, width, height, levels, &p_));
  }
  ~DevicePyramidGroup() { dsopp_hip_pyramid_group_destroy(p_); }
  DevicePyramidGroup(const DevicePyramidGroup &) = delete;
  DevicePyramidGroup &operator=(const DevicePyramidGroup &) = delete;
  void build(const uint8_t *image, const double *photometric_calibration256 = nullptr, const uint8_t *vignetting = nullptr) {
    check(dsopp_hip_pyramid_group_build(p_, image, photometric_calibration256, vignetting));
  }
  void setLevel(int level, const double *pixelinfo) { check(dsopp_hip_pyramid_group_set_level(p_, level, pixelinfo)); }
  void setMask(int level, const uint8_t *mask) { check(dsopp_hip_pyramid_group_set_mask(p_, level, mask)); }
  dsopp_hip_pyramid_group *handle() const { return p_; }

 private:
  dsopp_hip_pyramid_group *p_ = nullptr;
};

/** what the solvers read from track::landmarks::ActiveTrackingLandmark */
struct LandmarkView {
  Vector2 projection;
  double idepth;
  std::array<double, DSOPP_HIP_PATTERN_SIZE> patch;
  bool is_marginalized;
  bool is_outlier;
};

/** what the solvers read from / write to track::ActiveKeyframe<Motion> (frames/include/track/frames/active_keyframe.hpp:36-274) */
struct KeyframeView {
  int32_t keyframe_id;
  time_point timestamp;
  Motion t_world_agent;
  double exposure_time;
  Vector2 affine_brightness;
  bool is_marginalized;
  const DevicePyramid *pyramids;
  const DevicePyramidGroup *pyramid_group = nullptr;  // multi-device bundle adjustment: the image on every device of the solver
  std::vector<LandmarkView> active_landmarks;
  /** connections: other keyframe id -> statuses of THIS frame's landmarks reprojected into the other frame
   *  (FrameConnection::referenceReprojectionStatuses / targetReprojectionStatuses, whichever side this frame is) */
  std::map<int32_t, std::vector<uint8_t>> reprojection_statuses;
  // --- written back by updateFrame ---
  std::vector<double> idepth_variance;
  std::vector<int32_t> inlier_residuals;
  std::vector<double> relative_baseline;
  std::map<int32_t, Matrix6> covariances;
};

/** TrustRegionPhotometricBundleAdjustmentOptions<Precision> — trust_region_photometric_bundle_adjustment_options.hpp:14-52 */
struct TrustRegionOptions {
  size_t max_iterations;
  double initial_trust_region_radius, function_tolerance, parameter_tolerance;
  Vector2 affine_brightness_regularizer;
  double fixed_state_regularizer;
  double sigma_huber_loss = 5;
};

/** mirror of EigenPhotometricBundleAdjustment<SE3, PinholeCamera, 8, PixelMap, true, true, true, 1> backed by HIP kernels */
/** track::landmarks::ImmatureTrackingLandmark as the depth estimator reads and writes it
 *  (src/track/landmarks/include/track/landmarks/immature_tracking_landmark.hpp:26-106) */
struct ImmatureLandmarkView {
  Vector2 projection{};
  std::array<double, 3> direction{};
  std::array<double, 8> patch{};
  Vector2 gradient{};
  double idepth_min = 0, idepth_max = 1. / 0.001;
  double uniqueness = std::numeric_limits<double>::max(), search_pixel_interval = std::numeric_limits<double>::max();
  uint8_t status = 5;  // ImmatureStatus::kUninitialized
  bool traced = false;
};

/** tracker::DepthEstimation (src/tracker/depth_estimators/include/tracker/depth_estimators/depth_estimation.hpp:24-58): same
 *  static entry point, the target frame given as its device pyramid (level 0 is used, as monocular_tracker.cpp:98-100 does) */
struct HipDepthEstimation {
  static void estimate(const DevicePyramid &target_frame, std::vector<std::reference_wrapper<ImmatureLandmarkView>> &reference_landmarks,
                       const Motion &t_t_r, double reference_exposure_time, const Vector2 &reference_affine_brightness,
                       double target_exposure_time, const Vector2 &target_affine_brightness, const PinholeModel &model,
                       double sigma_huber_loss) {
    const size_t n = reference_landmarks.size();
    std::vector<double> proj(2 * n), dir(3 * n), patch(8 * n), grad(2 * n), imin(n), imax(n), uniq(n), spi(n);
    std::vector<uint8_t> status(n), traced(n);
    for (size_t i = 0; i < n; ++i) {
      const ImmatureLandmarkView &l = reference_landmarks[i];
      for (int k = 0; k < 2; ++k) {
        proj[2 * i + k] = l.projection[static_cast<size_t>(k)];
        grad[2 * i + k] = l.gradient[static_cast<size_t>(k)];
      }
      for (int k = 0; k < 3; ++k) dir[3 * i + k] = l.direction[static_cast<size_t>(k)];
      for (int k = 0; k < 8; ++k) patch[8 * i + k] = l.patch[static_cast<size_t>(k)];
      imin[i] = l.idepth_min;
      imax[i] = l.idepth_max;
      uniq[i] = l.uniqueness;
      spi[i] = l.search_pixel_interval;
      status[i] = l.status;
      traced[i] = l.traced ? 1 : 0;
    }
    const double intr[4] = {model.fx, model.fy, model.cx, model.cy};
    check(dsopp_hip_estimate_depths(target_frame.handle(), 0, intr, t_t_r.data(), reference_exposure_time, reference_affine_brightness.data(),
                                    target_exposure_time, target_affine_brightness.data(), sigma_huber_loss, static_cast<int32_t>(n), proj.data(),
                                    dir.data(), patch.data(), grad.data(), imin.data(), imax.data(), uniq.data(), spi.data(), status.data(),
                                    traced.data()));
    for (size_t i = 0; i < n; ++i) {
      ImmatureLandmarkView &l = reference_landmarks[i];
      l.idepth_min = imin[i];
      l.idepth_max = imax[i];
      l.uniqueness = uniq[i];
      l.search_pixel_interval = spi[i];
      l.status = status[i];
      l.traced = traced[i] != 0;
    }
  }
};

/** Reference depth maps of the newest keyframe, resident on the device: what tracker::createReferenceDepthMaps
 *  (src/tracker/tracker/src/create_depth_maps.cpp:124-147) returns as std::vector<energy::problem::DepthMap>. */
class DeviceDepthMaps {
 public:
  explicit DeviceDepthMaps(dsopp_hip_depth_maps *m = nullptr) : m_(m) {}
  ~DeviceDepthMaps() { dsopp_hip_depth_maps_destroy(m_); }
  DeviceDepthMaps(const DeviceDepthMaps &) = delete;
  DeviceDepthMaps &operator=(const DeviceDepthMaps &) = delete;
  DeviceDepthMaps(DeviceDepthMaps &&o) noexcept : m_(o.m_) { o.m_ = nullptr; }
  const dsopp_hip_depth_maps *handle() const { return m_; }
  dsopp_hip_depth_maps *mutableHandle() { return m_; }
  /** calculateMeanSquareOpticalFlow(reference_frame_depth_map[level], t_t_r, model) — monocular_tracker.cpp:104-134 — for up
   *  to four relative poses in one pass over the device-resident map (the tracker needs t_t_r and t_t_r without rotation) */
  std::vector<double> meanSquareOpticalFlow(int level, const std::vector<Motion> &t_t_r, const PinholeModel &model) const {
    const double intr[4] = {model.fx, model.fy, model.cx, model.cy};
    std::vector<double> T(7 * t_t_r.size()), flow(t_t_r.size());
    for (size_t i = 0; i < t_t_r.size(); ++i) std::copy(t_t_r[i].begin(), t_t_r[i].end(), T.begin() + 7 * static_cast<long>(i));
    check(dsopp_hip_depth_maps_mean_square_optical_flow(m_, level, intr, static_cast<int32_t>(t_t_r.size()), T.data(), flow.data()));
    return flow;
  }
  /** host copy of one level: two row-major H x W planes (DepthMap::map(x, y).idepth / .weight) */
  void level(int level, std::vector<double> &idepth_sum, std::vector<double> &weight, int &width, int &height) const {
    int32_t w = 0, h = 0;
    check(dsopp_hip_depth_maps_level_size(m_, level, &w, &h));
    idepth_sum.assign(static_cast<size_t>(w) * h, 0.0);
    weight.assign(static_cast<size_t>(w) * h, 0.0);
    check(dsopp_hip_depth_maps_get_level(m_, level, idepth_sum.data(), weight.data()));
    width = w;
    height = h;
  }

 private:
  dsopp_hip_depth_maps *m_;
};

class HipPhotometricBundleAdjustment {
 public:
  /** EigenPhotometricBundleAdjustment(options, estimate_uncertainty, force_accept) — eigen_photometric_bundle_adjustment.cpp:47-57 */
  HipPhotometricBundleAdjustment(const TrustRegionOptions &o, bool estimate_uncertainty = true, bool force_accept = true, int device = 0)
      : HipPhotometricBundleAdjustment(o, estimate_uncertainty, force_accept, std::vector<int>{device}) {}
  /** the single-process multi-GPU form (fabric_hip.patch: `devices: "0 1 2 3"`): the landmarks of every keyframe are sharded over
   *  `devices`, frames / images replicated, one all-reduce of the reduced system per Gauss-Newton iteration (dsopp_hip_window_group).
   *  `transport`: DSOPP_HIP_TRANSPORT_AUTO = RCCL between distinct devices */
  HipPhotometricBundleAdjustment(const TrustRegionOptions &o, bool estimate_uncertainty, bool force_accept, const std::vector<int> &devices,
                                 int transport = DSOPP_HIP_TRANSPORT_AUTO)
      : estimate_uncertainty_(estimate_uncertainty) {
    dsopp_hip_options c;
    dsopp_hip_default_pba_options(&c);
    c.max_iterations = static_cast<int32_t>(o.max_iterations);
    c.initial_trust_region_radius = o.initial_trust_region_radius;
    c.function_tolerance = o.function_tolerance;
    c.parameter_tolerance = o.parameter_tolerance;
    c.affine_brightness_regularizer[0] = o.affine_brightness_regularizer[0];
    c.affine_brightness_regularizer[1] = o.affine_brightness_regularizer[1];
    c.fixed_state_regularizer = o.fixed_state_regularizer;
    c.sigma_huber_loss = o.sigma_huber_loss;
    c.estimate_uncertainty = estimate_uncertainty ? 1 : 0;
    c.force_accept = force_accept ? 1 : 0;
    std::vector<int32_t> ids(devices.begin(), devices.end());
    check(dsopp_hip_window_group_create(&c, ids.data(), static_cast<int32_t>(ids.size()), transport, &g_));
    n_shards_ = static_cast<int>(ids.size());
    if (n_shards_ == 1) check(dsopp_hip_window_group_shard(g_, 0, &w_, nullptr));  // a group of one IS this window
  }
  ~HipPhotometricBundleAdjustment() { dsopp_hip_window_group_destroy(g_); }
  HipPhotometricBundleAdjustment(const HipPhotometricBundleAdjustment &) = delete;
  HipPhotometricBundleAdjustment &operator=(const HipPhotometricBundleAdjustment &) = delete;

  /** pushFrame(frame, level, model, frame_parameterization): local copy of the keyframe + residual lists between the new
   *  frame and all previous ones (photometric_bundle_adjustment.cpp:98-124); folds pending marginalisations first
   *  (eigen_photometric_bundle_adjustment.cpp:119-141) */
  void pushFrame(const KeyframeView &frame, size_t level, const PinholeModel &model,
                 FrameParameterization frame_parameterization = FrameParameterization::kFree) {
    const double intr[4] = {model.fx, model.fy, model.cx, model.cy};
    const int fixed = frame_parameterization == FrameParameterization::kFixed ? 1 : 0;
    if (frame.pyramid_group) {
      check(dsopp_hip_window_group_push_frame(g_, frame.keyframe_id, frame.timestamp, frame.pyramid_group->handle(), static_cast<int>(level), intr,
                                              frame.t_world_agent.data(), frame.exposure_time, frame.affine_brightness.data(), fixed,
                                              frame.is_marginalized ? 1 : 0));
    } else {
      if (!w_) throw SolverError(DSOPP_HIP_ERR_INVALID_ARGUMENT, "a multi-device solver needs KeyframeView::pyramid_group (the image on every device)");
      check(dsopp_hip_window_push_frame(w_, frame.keyframe_id, frame.timestamp, frame.pyramids->handle(), static_cast<int>(level), intr,
                                        frame.t_world_agent.data(), frame.exposure_time, frame.affine_brightness.data(), fixed,
                                        frame.is_marginalized ? 1 : 0));
    }
    times_[frame.timestamp] = frame.keyframe_id;
    uploadLandmarks(frame);
    uploadConnections(frame);
  }
  /** updateLocalFrame(frame): freshly matured landmarks, marginalisation flags, new connections
   *  (eigen_photometric_bundle_adjustment.cpp:103-113, local_frame.hpp:484-521) */
  void updateLocalFrame(const KeyframeView &frame) {
    uploadLandmarks(frame);
    uploadConnections(frame);
    if (frame.is_marginalized) check(dsopp_hip_window_group_mark_frame_marginalized(g_, frame.keyframe_id));
  }
  /** tracker::createReferenceDepthMaps(active frames, calibration) for the window this solver holds, without leaving the
   *  device (monocular_tracker.cpp:465,509 call it right after the bundle adjustment, on the same keyframes) */
  DeviceDepthMaps createReferenceDepthMaps(int levels) {
    dsopp_hip_depth_maps *m = nullptr;
    check(dsopp_hip_window_group_create_reference_depth_maps(g_, levels, &m));
    return DeviceDepthMaps(m);
  }
  /** reference_frame_depth_map_ = createReferenceDepthMaps(...) into the object the tracker already holds (no allocation) */
  void createReferenceDepthMaps(DeviceDepthMaps &maps) { check(dsopp_hip_window_group_refill_reference_depth_maps(g_, maps.mutableHandle())); }
  /** solve(number_of_threads) -> final energy; number_of_threads is accepted and ignored, as in the Eigen backend */
  double solve(const size_t number_of_threads = 1) {
    (void)number_of_threads;
    double e = 0;
    int32_t it = 0, nv = 0;
    check(dsopp_hip_window_group_solve(g_, &e, &it, &nv));
    return e;
  }
  /** updateFrame(frame): poses, affine brightness, idepths, variances, inlier counts, baselines, statuses, covariances
   *  (photometric_bundle_adjustment.cpp:182-264) */
  void updateFrame(KeyframeView &frame) {
    check(dsopp_hip_window_group_get_pose(g_, frame.keyframe_id, frame.t_world_agent.data(), frame.affine_brightness.data()));
    int32_t n = 0;
    check(dsopp_hip_window_group_num_landmarks(g_, frame.keyframe_id, &n));
    std::vector<double> &idepth = scratch_idepth_, &inv_h = scratch_uv_, &baseline = scratch_patch_;  // (the solver's scratch: see uploadLandmarks)
    std::vector<int32_t> &inliers = scratch_inliers_;
    std::vector<uint8_t> &flags = scratch_flags_, &statuses = scratch_statuses_;
    idepth.resize(static_cast<size_t>(n));
    inv_h.resize(static_cast<size_t>(n));
    baseline.resize(static_cast<size_t>(n));
    inliers.resize(static_cast<size_t>(n));
    flags.resize(static_cast<size_t>(n));
    // one packed transfer for the landmark arrays and the statuses towards every connected frame
    std::vector<int32_t> &targets = scratch_targets_;
    targets.clear();
    for (const auto &kv : frame.reprojection_statuses)
      if (static_cast<int32_t>(kv.second.size()) == n && n > 0) targets.push_back(kv.first);
    statuses.resize(targets.size() * static_cast<size_t>(n));
    check(dsopp_hip_window_group_get_frame_update(g_, frame.keyframe_id, idepth.data(), inv_h.data(), baseline.data(), inliers.data(), flags.data(),
                                            static_cast<int32_t>(targets.size()), targets.data(), statuses.data()));
    // entries of landmarks the solver has not written yet start at the reference's defaults; entries of marginalised
    // landmarks are left as they are (photometric_bundle_adjustment.cpp:232-262 `continue`s before touching them)
    frame.idepth_variance.resize(n, 1e-5);
    frame.inlier_residuals.resize(n, 0);
    frame.relative_baseline.resize(n, 0.0);
    const double kIdepthEps = 1e-8;
    for (int32_t i = 0; i < n && i < static_cast<int32_t>(frame.active_landmarks.size()); ++i) {
      LandmarkView &lm = frame.active_landmarks[static_cast<size_t>(i)];
      if (flags[i] & 2) lm.is_outlier = true;
      if (flags[i] & 1) continue;  // marginalised landmarks keep their values
      if (std::abs(idepth[i]) < kIdepthEps)
        lm.idepth = 0;
      else if (idepth[i] < 0)
        lm.is_outlier = true;
      else
        lm.idepth = idepth[i];
      frame.idepth_variance[i] = estimate_uncertainty_ ? inv_h[i] : 1e-5;  // :252-254
      frame.inlier_residuals[i] = inliers[i];
      if (baseline[i] > frame.relative_baseline[i]) frame.relative_baseline[i] = baseline[i];
    }
    for (size_t k = 0; k < targets.size(); ++k)
      frame.reprojection_statuses[targets[k]].assign(statuses.begin() + static_cast<long>(k * static_cast<size_t>(n)),
                                                     statuses.begin() + static_cast<long>((k + 1) * static_cast<size_t>(n)));
    for (auto &kv : frame.reprojection_statuses) {
      if (kv.second.empty()) continue;
      Matrix6 cov;
      if (estimate_uncertainty_ && dsopp_hip_window_group_get_covariance(g_, frame.keyframe_id, kv.first, cov.data()) == DSOPP_HIP_OK)
        frame.covariances[kv.first] = cov;
    }
  }
  Motion getPose(time_point timestamp) const {
    Motion T;
    check(dsopp_hip_window_group_get_pose(g_, idOf(timestamp), T.data(), nullptr));
    return T;
  }
  Vector2 getAffineBrightness(time_point timestamp) const {
    Vector2 ab{0, 0};
    auto it = times_.find(timestamp);
    if (it == times_.end()) return ab;  // the reference returns zero for an unknown frame
    check(dsopp_hip_window_group_get_pose(g_, it->second, nullptr, ab.data()));
    return ab;
  }
  /** the one device window of a single-device solver (nullptr for a multi-device one): what the device-resident landmark
   *  activation reads, which needs ALL active landmarks of the window on one device */
  dsopp_hip_window *handle() const { return w_; }
  dsopp_hip_window_group *group() const { return g_; }
  int numDevices() const { return n_shards_; }

 private:
  int32_t idOf(time_point t) const {
    auto it = times_.find(t);
    if (it == times_.end()) throw SolverError(DSOPP_HIP_ERR_NOT_FOUND, "no local copy of this frame in the solver");
    return it->second;
  }
  void uploadLandmarks(const KeyframeView &frame) {
    const size_t n = frame.active_landmarks.size();
    // flags of every landmark, coordinates / inverse depth / patch only of those the device does not hold yet (the C-ABI reads these
    // arrays from its own landmark count on)
    int32_t held = 0;
    check(dsopp_hip_window_group_num_landmarks(g_, frame.keyframe_id, &held));
    // (scratch kept with the solver: updateLocalFrame runs for every keyframe of the window three times per keyframe step, and four fresh
    // zero-filled arrays of all its landmarks per call were a measurable part of that step's host time)
    std::vector<double> &uv = scratch_uv_, &idepth = scratch_idepth_, &patch = scratch_patch_;
    std::vector<uint8_t> &flags = scratch_flags_;
    uv.resize(2 * n);
    idepth.resize(n);
    patch.resize(DSOPP_HIP_PATTERN_SIZE * n);
    flags.resize(n);
    for (size_t i = 0; i < n; ++i) {
      const LandmarkView &lm = frame.active_landmarks[i];
      flags[i] = static_cast<uint8_t>((lm.is_marginalized ? 1 : 0) | (lm.is_outlier ? 2 : 0));
      if (i < static_cast<size_t>(held)) continue;
      uv[2 * i] = lm.projection[0];
      uv[2 * i + 1] = lm.projection[1];
      idepth[i] = lm.idepth;
      for (int k = 0; k < DSOPP_HIP_PATTERN_SIZE; ++k) patch[DSOPP_HIP_PATTERN_SIZE * i + static_cast<size_t>(k)] = lm.patch[static_cast<size_t>(k)];
    }
    check(dsopp_hip_window_group_set_landmarks(g_, frame.keyframe_id, static_cast<int32_t>(n), uv.data(), idepth.data(), patch.data(), flags.data()));
  }
  void uploadConnections(const KeyframeView &frame) {
    int32_t nf = 0;
    check(dsopp_hip_window_group_num_frames(g_, &nf));
    for (const auto &kv : frame.reprojection_statuses) {
      if (kv.second.empty()) continue;
      const int rc = dsopp_hip_window_group_set_connection(g_, frame.keyframe_id, kv.first, static_cast<int32_t>(kv.second.size()), kv.second.data());
      if (rc != DSOPP_HIP_OK && rc != DSOPP_HIP_ERR_NOT_FOUND) check(rc);
    }
  }
  std::vector<double> scratch_uv_, scratch_idepth_, scratch_patch_;
  std::vector<uint8_t> scratch_flags_, scratch_statuses_;
  std::vector<int32_t> scratch_inliers_, scratch_targets_;
  dsopp_hip_window_group *g_ = nullptr;
  dsopp_hip_window *w_ = nullptr;  // shard 0's window when the group has one shard
  int n_shards_ = 1;
  bool estimate_uncertainty_;
  std::map<time_point, int32_t> times_;
};

/** what MonocularTracker::estimatePose leaves behind (monocular_tracker.cpp:179-245): the new frame's pose and affine brightness,
 *  whether an initialisation passed the per-level rmse gates, how many were tried and the LM iterations spent */
struct PoseEstimate {
  Motion t_world_target;
  Vector2 affine_brightness;
  bool success;
  int tries, lm_iterations;
};

/** mirror of EigenPoseAlignment<SE3, PinholeCamera, 1, PixelMap, 1, true> backed by HIP kernels */
class HipPoseAlignment {
 public:
  static constexpr double kZeroCost = -1;  // PA_INC/pose_alignment.hpp
  explicit HipPoseAlignment(const TrustRegionOptions &o, int device = 0, void *stream = nullptr) {
    dsopp_hip_options c;
    dsopp_hip_default_align_options(&c);
    c.max_iterations = static_cast<int32_t>(o.max_iterations);
    c.initial_trust_region_radius = o.initial_trust_region_radius;
    c.function_tolerance = o.function_tolerance;
    c.parameter_tolerance = o.parameter_tolerance;
    c.affine_brightness_regularizer[0] = o.affine_brightness_regularizer[0];
    c.affine_brightness_regularizer[1] = o.affine_brightness_regularizer[1];
    c.sigma_huber_loss = o.sigma_huber_loss;
    check(dsopp_hip_aligner_create(&c, device, stream, &a_));
  }
  ~HipPoseAlignment() { dsopp_hip_aligner_destroy(a_); }
  HipPoseAlignment(const HipPoseAlignment &) = delete;
  HipPoseAlignment &operator=(const HipPoseAlignment &) = delete;

  void reset() { check(dsopp_hip_aligner_reset(a_)); }
  /** pushFrame(timestamp, t_world_agent, pyramids, masks, depths_maps, exposure, affine, level, model, kFixed)
   *  — photometric_bundle_adjustment.cpp:58-74; depth maps are H_l x W_l {idepth sum, weight} (energy/problems/depth_map.hpp) */
  void pushFrame(time_point timestamp, const Motion &t_world_agent, const DevicePyramid &pyramids, const double *depth_idepth_sum,
                 const double *depth_weight, double exposure_time, const Vector2 &affine_brightness, size_t level, const PinholeModel &model) {
    const double intr[4] = {model.fx, model.fy, model.cx, model.cy};
    check(dsopp_hip_aligner_push_reference_depth_map(a_, timestamp, t_world_agent.data(), pyramids.handle(), static_cast<int>(level), intr,
                                                     depth_idepth_sum, depth_weight, exposure_time, affine_brightness.data()));
  }
  /** the same overload fed from device-resident maps (HipPhotometricBundleAdjustment::createReferenceDepthMaps) */
  void pushFrame(time_point timestamp, const Motion &t_world_agent, const DevicePyramid &pyramids, const DeviceDepthMaps &depth_maps,
                 double exposure_time, const Vector2 &affine_brightness, size_t level, const PinholeModel &model) {
    const double intr[4] = {model.fx, model.fy, model.cx, model.cy};
    check(dsopp_hip_aligner_push_reference_depth_maps(a_, timestamp, t_world_agent.data(), pyramids.handle(), static_cast<int>(level), intr,
                                                      depth_maps.handle(), exposure_time, affine_brightness.data()));
  }
  /** pushFrame(timestamp, t_world_agent_init, pyramids, masks, exposure, affine, level, model, kFree) — :131-154 */
  void pushFrame(time_point timestamp, const Motion &t_world_agent_init, const DevicePyramid &pyramids, double exposure_time,
                 const Vector2 &affine_brightness, size_t level, const PinholeModel &model) {
    const double intr[4] = {model.fx, model.fy, model.cx, model.cy};
    check(dsopp_hip_aligner_push_target(a_, timestamp, t_world_agent_init.data(), pyramids.handle(), static_cast<int>(level), intr,
                                        exposure_time, affine_brightness.data()));
    target_time_ = timestamp;
  }
  /** setRotationPrior(r_t_r) — pose_alignment.hpp:39, eigen_pose_alignment.cpp:254-257 (row-major 3x3) */
  void setRotationPrior(const std::array<double, 9> &r_t_r) { check(dsopp_hip_aligner_set_rotation_prior(a_, r_t_r.data())); }
  void pushKnownPose(time_point timestamp, const Motion &t_w_agent) { check(dsopp_hip_aligner_push_known_pose(a_, timestamp, t_w_agent.data())); }
  /** solve(number_of_threads) -> rmse, or kZeroCost when the pose was known */
  double solve(const size_t number_of_threads = 1) {
    (void)number_of_threads;
    check(dsopp_hip_aligner_solve(a_, &last_));
    return last_.rmse;
  }
  /** estimatePose of the tracker (monocular_tracker.cpp:179-245) as ONE call: for every initialisation in turn, coarse to fine over the
   *  levels of the target pyramid { reset; pushFrame(keyframe, depth map of the level); pushFrame(target, current estimate); solve; gate on
   *  2.5 x rmse_last_pose_estimation[level] } — the loop the tracker runs over this object, executed by one persistent launch per frame
   *  (up to 8 initialisations per launch once a first one has failed).  `rmse_last_pose_estimation` (one entry per level) is updated
   *  as the reference updates it. */
  PoseEstimate estimatePose(time_point reference_time, const Motion &t_world_reference, const DevicePyramid &reference_pyramids,
                            const DeviceDepthMaps &reference_depth_maps, double reference_exposure_time, const Vector2 &reference_affine_brightness,
                            time_point target_time, const DevicePyramid &target_pyramids, double target_exposure_time, const PinholeModel &model,
                            const std::vector<Motion> &initializations, const Vector2 &affine_brightness_init,
                            std::vector<double> &rmse_last_pose_estimation) {
    const double intr[4] = {model.fx, model.fy, model.cx, model.cy};
    std::vector<double> inits(7 * initializations.size());
    for (size_t i = 0; i < initializations.size(); ++i) std::copy(initializations[i].begin(), initializations[i].end(), inits.begin() + 7 * static_cast<long>(i));
    PoseEstimate out{};
    int32_t success = 0, tries = 0, its = 0;
    check(dsopp_hip_aligner_estimate_pose(a_, reference_time, t_world_reference.data(), reference_pyramids.handle(), reference_depth_maps.handle(),
                                          reference_exposure_time, reference_affine_brightness.data(), target_time, target_pyramids.handle(),
                                          target_exposure_time, intr, static_cast<int32_t>(initializations.size()), inits.data(),
                                          affine_brightness_init.data(), rmse_last_pose_estimation.data(), out.t_world_target.data(),
                                          out.affine_brightness.data(), &success, &tries, &its));
    out.success = success != 0;
    out.tries = tries;
    out.lm_iterations = its;
    return out;
  }
  Motion getPose(time_point) const {
    Motion T;
    for (int i = 0; i < 7; ++i) T[static_cast<size_t>(i)] = last_.T_world_target[i];
    return T;
  }
  Vector2 getAffineBrightness(time_point) const { return {last_.affine_brightness[0], last_.affine_brightness[1]}; }
  Matrix6 tTargetReferenceCovariance() const {
    Matrix6 c;
    for (int i = 0; i < 36; ++i) c[static_cast<size_t>(i)] = last_.covariance[i];
    return c;
  }

 private:
  dsopp_hip_aligner *a_ = nullptr;
  dsopp_hip_align_result last_{};
  time_point target_time_ = 0;
};

/** The immature landmarks of one keyframe kept on the device for their lifetime (ActiveKeyframe::immature_landmarks_:
 *  created by pushImmatureLandmarks, traced every frame by the depth estimator, consumed by the activator). */
class DeviceImmatureSet {
 public:
  explicit DeviceImmatureSet(const std::vector<ImmatureLandmarkView> &landmarks, int device = 0, void *stream = nullptr)
      : n_(landmarks.size()) {
    std::vector<double> proj(2 * n_), dir(3 * n_), patch(8 * n_), grad(2 * n_);
    for (size_t i = 0; i < n_; ++i) {
      const ImmatureLandmarkView &l = landmarks[i];
      for (size_t k = 0; k < 2; ++k) {
        proj[2 * i + k] = l.projection[k];
        grad[2 * i + k] = l.gradient[k];
      }
      for (size_t k = 0; k < 3; ++k) dir[3 * i + k] = l.direction[k];
      for (size_t k = 0; k < 8; ++k) patch[8 * i + k] = l.patch[k];
    }
    check(dsopp_hip_immature_set_create(device, stream, static_cast<int32_t>(n_), proj.data(), dir.data(), patch.data(), grad.data(), &s_));
    upload(landmarks);
  }
  ~DeviceImmatureSet() { dsopp_hip_immature_set_destroy(s_); }
  DeviceImmatureSet(const DeviceImmatureSet &) = delete;
  DeviceImmatureSet &operator=(const DeviceImmatureSet &) = delete;
  dsopp_hip_immature_set *handle() const { return s_; }
  size_t size() const { return n_; }
  /** DepthEstimator::estimate on the resident set (asynchronous) */
  void estimate(const DevicePyramid &target_frame, const Motion &t_t_r, double reference_exposure_time, const Vector2 &reference_affine_brightness,
                double target_exposure_time, const Vector2 &target_affine_brightness, const PinholeModel &model, double sigma_huber_loss) {
    const double intr[4] = {model.fx, model.fy, model.cx, model.cy};
    check(dsopp_hip_immature_set_estimate(s_, target_frame.handle(), 0, intr, t_t_r.data(), reference_exposure_time, reference_affine_brightness.data(),
                                          target_exposure_time, target_affine_brightness.data(), sigma_huber_loss));
  }
  /** wait for the estimator launches enqueued on the set's stream (a download with no outputs) */
  void synchronize() const { check(dsopp_hip_immature_set_download_state(s_, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr)); }
  void upload(const std::vector<ImmatureLandmarkView> &landmarks) {
    std::vector<double> imin(n_), imax(n_), uniq(n_), spi(n_);
    std::vector<uint8_t> status(n_), traced(n_);
    for (size_t i = 0; i < n_; ++i) {
      const ImmatureLandmarkView &l = landmarks[i];
      imin[i] = l.idepth_min;
      imax[i] = l.idepth_max;
      uniq[i] = l.uniqueness;
      spi[i] = l.search_pixel_interval;
      status[i] = l.status;
      traced[i] = l.traced ? 1 : 0;
    }
    check(dsopp_hip_immature_set_upload_state(s_, imin.data(), imax.data(), uniq.data(), spi.data(), status.data(), traced.data()));
  }
  void download(std::vector<ImmatureLandmarkView> &landmarks) const {
    std::vector<double> imin(n_), imax(n_), uniq(n_), spi(n_);
    std::vector<uint8_t> status(n_), traced(n_);
    check(dsopp_hip_immature_set_download_state(s_, imin.data(), imax.data(), uniq.data(), spi.data(), status.data(), traced.data()));
    for (size_t i = 0; i < n_; ++i) {
      ImmatureLandmarkView &l = landmarks[i];
      l.idepth_min = imin[i];
      l.idepth_max = imax[i];
      l.uniqueness = uniq[i];
      l.search_pixel_interval = spi[i];
      l.status = status[i];
      l.traced = traced[i] != 0;
    }
  }

 private:
  size_t n_;
  dsopp_hip_immature_set *s_ = nullptr;
};

/** initializationPoses(track) — src/tracker/tracker/src/monocular_tracker.cpp:136-176: the hypotheses estimatePose tries in
 *  turn (113 for a track with at least two frames, else the identity).  Pass nullptr for a shorter track. */
inline std::vector<Motion> initializationPoses(const Motion *t_world_previous, const Motion *t_world_last, const Motion *t_world_keyframe) {
  std::vector<double> buf(7 * 128);
  int32_t n = 0;
  const bool two = t_world_previous && t_world_last && t_world_keyframe;
  check(dsopp_hip_initialization_poses(two ? t_world_previous->data() : nullptr, two ? t_world_last->data() : nullptr,
                                       two ? t_world_keyframe->data() : nullptr, 128, buf.data(), &n));
  std::vector<Motion> out(static_cast<size_t>(n));
  for (int32_t i = 0; i < n; ++i) std::copy(buf.begin() + 7 * i, buf.begin() + 7 * (i + 1), out[static_cast<size_t>(i)].begin());
  return out;
}

/** estimateDepths of the tracker (monocular_tracker.cpp:74-102) for all keyframes of the window in one launch:
 *  DepthEstimator::estimate(target_frame, keyframe k's immature landmarks, t_t_r[k], ...) for every k */
inline void estimateDepthsOfWindow(const DevicePyramid &target_frame, const std::vector<DeviceImmatureSet *> &keyframe_sets,
                                   const std::vector<Motion> &t_target_keyframe, const std::vector<double> &keyframe_exposure_times,
                                   const std::vector<Vector2> &keyframe_affine_brightness, double target_exposure_time,
                                   const Vector2 &target_affine_brightness, const PinholeModel &model, double sigma_huber_loss) {
  const size_t n = keyframe_sets.size();
  std::vector<dsopp_hip_immature_set *> sets(n);
  std::vector<double> T(7 * n), ab(2 * n);
  for (size_t k = 0; k < n; ++k) {
    sets[k] = keyframe_sets[k]->handle();
    std::copy(t_target_keyframe[k].begin(), t_target_keyframe[k].end(), T.begin() + 7 * static_cast<long>(k));
    ab[2 * k] = keyframe_affine_brightness[k][0];
    ab[2 * k + 1] = keyframe_affine_brightness[k][1];
  }
  const double intr[4] = {model.fx, model.fy, model.cx, model.cy};
  check(dsopp_hip_immature_sets_estimate(static_cast<int32_t>(n), sets.data(), target_frame.handle(), 0, intr, T.data(), keyframe_exposure_times.data(),
                                         ab.data(), target_exposure_time, target_affine_brightness.data(), sigma_huber_loss));
}

/** ActiveKeyframe::ImmatureLandmarkActivationStatus — src/track/frames/include/track/frames/active_keyframe.hpp:40-44 */
enum class ImmatureLandmarkActivationStatus : uint8_t { kActivate = 0, kSkip = 1, kDelete = 2 };

/**
 * tracker::LandmarksActivator<SE3, PinholeCamera, PixelMap, 1, REFINE>
 * (src/tracker/landmarks_activator/include/tracker/landmarks_activator/landmarks_activator.hpp:31-57): same constructor
 * arguments and the same persistent min_distance_to_neighbor_.  The track is what the HIP backend holds of it: the window
 * solver (poses, affine brightness, images and active landmarks of the older keyframes), their device-resident immature
 * sets, and the new keyframe that pushNewKeyframe just created (monocular_tracker.cpp:491-497 call order).
 */
template <bool REFINE = false>
class HipLandmarksActivator {
 public:
  struct Track {
    HipPhotometricBundleAdjustment *solver;          // holds every keyframe of activeFrames() but the newest
    std::vector<int32_t> keyframe_ids;               // oldest first
    std::vector<DeviceImmatureSet *> immature;       // per keyframe, nullptr = no immature landmarks
    const KeyframeView *newest;                      // pose, exposure, affine brightness, pyramid (>= 2 levels)
  };
  explicit HipLandmarksActivator(double sigma_huber_loss = 9, size_t number_of_desired_points = 2000)
      : sigma_huber_loss_(sigma_huber_loss), number_of_desired_points_(number_of_desired_points) {}
  /** activate(track): returns the statuses applyImmatureLandmarkActivationStatuses consumes, per keyframe; `idepths` (optional)
   *  receives landmark.idepth() after the refinement — what the new ActiveTrackingLandmark is constructed with */
  std::vector<std::vector<ImmatureLandmarkActivationStatus>> activate(const Track &track, std::vector<std::vector<double>> *idepths = nullptr) {
    const size_t n = track.keyframe_ids.size();
    std::vector<dsopp_hip_immature_set *> sets(n, nullptr);
    std::vector<std::vector<ImmatureLandmarkActivationStatus>> statuses(n);
    std::vector<uint8_t *> st_ptr(n, nullptr);
    std::vector<double *> id_ptr(n, nullptr);
    if (idepths) idepths->assign(n, {});
    for (size_t k = 0; k < n; ++k) {
      if (!track.immature[k]) continue;
      sets[k] = track.immature[k]->handle();
      statuses[k].resize(track.immature[k]->size());
      st_ptr[k] = reinterpret_cast<uint8_t *>(statuses[k].data());
      if (idepths) {
        (*idepths)[k].resize(track.immature[k]->size());
        id_ptr[k] = (*idepths)[k].data();
      }
    }
    const KeyframeView &nk = *track.newest;
    check(dsopp_hip_window_activate_landmarks(track.solver->handle(), static_cast<int32_t>(n), track.keyframe_ids.data(), sets.data(),
                                              nk.pyramids->handle(), nk.t_world_agent.data(), nk.exposure_time, nk.affine_brightness.data(),
                                              static_cast<int32_t>(number_of_desired_points_), &min_distance_to_neighbor_, REFINE ? 1 : 0,
                                              sigma_huber_loss_, st_ptr.data(), id_ptr.data(), &last_));
    return statuses;
  }
  double minDistanceToNeighbor() const { return min_distance_to_neighbor_; }
  const dsopp_hip_activation_result &lastResult() const { return last_; }

 private:
  double min_distance_to_neighbor_ = 2;  // landmarks_activator.hpp:51
  const double sigma_huber_loss_;
  const size_t number_of_desired_points_;
  dsopp_hip_activation_result last_{};
};

}  // namespace dsopp_hip_host
