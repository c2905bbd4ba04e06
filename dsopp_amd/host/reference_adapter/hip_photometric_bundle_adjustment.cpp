// Implementation of HipPhotometricBundleAdjustment (see the header for the design).  Goes to
//   src/energy/problems/src/hip_photometric_bundle_adjustment.cpp
// of the reference tree; needs `dsopp_hip.h` on the include path and links against libdsopp_hip.so (reference_adapter/README.md).
#include "energy/problems/photometric_bundle_adjustment/hip_photometric_bundle_adjustment.hpp"

#include <dsopp_hip.h>
#include <glog/logging.h>

#include <algorithm>
#include <array>

#include "energy/camera_model/pinhole/pinhole_camera.hpp"
#include "energy/problems/photometric_bundle_adjustment/local_frame.hpp"
#include "track/connections/frame_connection.hpp"
#include "track/frames/active_keyframe.hpp"
#include "track/landmarks/active_tracking_landmark.hpp"

namespace dsopp {
namespace energy {
namespace problem {
namespace {

#define DSOPP_HIP_CHECKED(call) CHECK_EQ((call), DSOPP_HIP_OK) << #call << ": " << dsopp_hip_last_error()

/** Sophus storage order of the C-ABI: (qx, qy, qz, qw, tx, ty, tz) */
template <typename MotionT>
std::array<double, 7> toParameters(const MotionT &motion) {
  const auto q = motion.unit_quaternion();
  const auto t = motion.translation();
  return {static_cast<double>(q.x()), static_cast<double>(q.y()), static_cast<double>(q.z()), static_cast<double>(q.w()),
          static_cast<double>(t.x()), static_cast<double>(t.y()), static_cast<double>(t.z())};
}

template <typename MotionT>
MotionT fromParameters(const double *p) {
  using S = typename MotionT::Scalar;
  return MotionT(Eigen::Quaternion<S>(static_cast<S>(p[3]), static_cast<S>(p[0]), static_cast<S>(p[1]), static_cast<S>(p[2])),
                 Eigen::Vector<S, 3>(static_cast<S>(p[4]), static_cast<S>(p[5]), static_cast<S>(p[6])));
}

int64_t ticks(time timestamp) { return static_cast<int64_t>(timestamp.time_since_epoch().count()); }

/** a one-level device image of a PixelMap level: the (I, dI/dx, dI/dy) triplets the reference computed on the host are adopted
 *  as they are (PixelInfo<1>::data_ is three contiguous scalars per pixel, row-major: features/src/pixel_map.cpp:105-110), the
 *  CameraMask of the level rides in the texel's spare lane */
dsopp_hip_pyramid_group *makeDeviceLevel(dsopp_hip_window_group *group, const features::PixelMap<1> &level,
                                         const sensors::calibration::CameraMask &mask) {
  const int width = static_cast<int>(level.width()), height = static_cast<int>(level.height());
  dsopp_hip_pyramid_group *pyramid = nullptr;  // one device-resident copy per device of the group
  DSOPP_HIP_CHECKED(dsopp_hip_pyramid_group_create(group, width, height, 1, &pyramid));
  const size_t n = static_cast<size_t>(width) * static_cast<size_t>(height);
  if constexpr (std::is_same_v<Precision, double>) {
    DSOPP_HIP_CHECKED(dsopp_hip_pyramid_group_set_level(pyramid, 0, reinterpret_cast<const double *>(&level(size_t(0)))));
  } else {
    std::vector<double> pixelinfo(3 * n);
    const Precision *source = reinterpret_cast<const Precision *>(&level(size_t(0)));
    std::copy(source, source + 3 * n, pixelinfo.begin());
    DSOPP_HIP_CHECKED(dsopp_hip_pyramid_group_set_level(pyramid, 0, pixelinfo.data()));
  }
  const cv::Mat &mask_image = mask.data();
  if (!mask_image.empty()) {
    CHECK(mask_image.rows == height && mask_image.cols == width) << "mask and image level differ in size";
    const cv::Mat continuous = mask_image.isContinuous() ? mask_image : mask_image.clone();
    DSOPP_HIP_CHECKED(dsopp_hip_pyramid_group_set_mask(pyramid, 0, continuous.ptr<uint8_t>()));
  }
  return pyramid;
}

}  // namespace

template <energy::motion::Motion Motion, model::Model Model>
HipPhotometricBundleAdjustment<Motion, Model>::HipPhotometricBundleAdjustment(
    const TrustRegionPhotometricBundleAdjustmentOptions<Precision> &trust_region_options, bool estimate_uncertainty, bool force_accept,
    const std::vector<int> &devices)
    : Base(estimate_uncertainty), devices_(devices) {
  CHECK(!devices_.empty()) << "the hip photometric bundle adjustment needs at least one device";
  dsopp_hip_options options;
  dsopp_hip_default_pba_options(&options);
  options.max_iterations = static_cast<int32_t>(trust_region_options.max_iterations);
  options.initial_trust_region_radius = static_cast<double>(trust_region_options.initial_trust_region_radius);
  options.function_tolerance = static_cast<double>(trust_region_options.function_tolerance);
  options.parameter_tolerance = static_cast<double>(trust_region_options.parameter_tolerance);
  options.affine_brightness_regularizer[0] = static_cast<double>(trust_region_options.affine_brightness_regularizer[0]);
  options.affine_brightness_regularizer[1] = static_cast<double>(trust_region_options.affine_brightness_regularizer[1]);
  options.fixed_state_regularizer = static_cast<double>(trust_region_options.fixed_state_regularizer);
  options.sigma_huber_loss = static_cast<double>(trust_region_options.sigma_huber_loss);
  options.estimate_uncertainty = estimate_uncertainty ? 1 : 0;
  options.force_accept = force_accept ? 1 : 0;
  options.first_estimate_jacobians = 1;
  options.optimize_idepths = 1;
  options.dtype = DSOPP_HIP_F64;
  // one shard of the window per device; distinct devices exchange over RCCL / xGMI (DSOPP_HIP_TRANSPORT_AUTO)
  std::vector<int32_t> ids(devices_.begin(), devices_.end());
  DSOPP_HIP_CHECKED(dsopp_hip_window_group_create(&options, ids.data(), static_cast<int32_t>(ids.size()), DSOPP_HIP_TRANSPORT_AUTO, &group_));
}

template <energy::motion::Motion Motion, model::Model Model>
dsopp_hip_window *HipPhotometricBundleAdjustment<Motion, Model>::window() const {
  if (devices_.size() != 1) return nullptr;
  dsopp_hip_window *window = nullptr;
  DSOPP_HIP_CHECKED(dsopp_hip_window_group_shard(group_, 0, &window, nullptr));
  return window;
}

template <energy::motion::Motion Motion, model::Model Model>
HipPhotometricBundleAdjustment<Motion, Model>::~HipPhotometricBundleAdjustment() {
  dsopp_hip_window_group_destroy(group_);  // before the pyramids its windows borrow
  for (auto &[id, pyramid] : pyramids_) dsopp_hip_pyramid_group_destroy(pyramid);
}

template <energy::motion::Motion Motion, model::Model Model>
void HipPhotometricBundleAdjustment<Motion, Model>::pushFrame(const track::ActiveKeyframe<Motion> &frame, size_t level, const Model &model,
                                                              FrameParameterization frame_parameterization) {
  CHECK(this->frames_.empty() or this->frames_.back()->timestamp < frame.timestamp())
      << "Frames must be processed in ascending order of time";
  CHECK_EQ(frame.sensors().size(), 1u);
  const size_t sensor = frame.sensors()[0];
  const int id = static_cast<int>(frame.keyframeId());

  // the device image of this keyframe's level (the base keeps a raw pointer to the same PixelMap)
  CHECK(pyramids_.find(id) == pyramids_.end());
  dsopp_hip_pyramid_group *pyramid = makeDeviceLevel(group_, frame.getLevel(sensor, level), frame.getMask(sensor, level));
  pyramids_[id] = pyramid;

  // Device window(s): when it already holds more than one frame, dsopp_hip_window_group_push_frame first folds the landmarks / frames
  // flagged for marginalisation into the marginal prior and erases those frames (updateMarginalizedLinearSystem), exactly where
  // EigenPhotometricBundleAdjustment::pushFrame does (eigen_photometric_bundle_adjustment.cpp:122-131).
  const bool folds = this->frames_.size() > 1;
  const auto intrinsics = model.intrinsicsParameters();  // fx, fy, cx, cy of that level
  const double intr[4] = {static_cast<double>(intrinsics[0]), static_cast<double>(intrinsics[1]), static_cast<double>(intrinsics[2]),
                          static_cast<double>(intrinsics[3])};
  const auto pose = toParameters(frame.tWorldAgent());
  const double affine[2] = {static_cast<double>(frame.affineBrightness()[0]), static_cast<double>(frame.affineBrightness()[1])};
  DSOPP_HIP_CHECKED(dsopp_hip_window_group_push_frame(group_, id, ticks(frame.timestamp()), pyramid, 0, intr, pose.data(),
                                                static_cast<double>(frame.exposureTime()), affine,
                                                frame_parameterization == FrameParameterization::kFixed ? 1 : 0,
                                                frame.isMarginalized() ? 1 : 0));
  if (folds) {
    // what updateMarginalizedLinearSystem leaves behind in `frames_` (eigen_photometric_bundle_adjustment_problem.hpp:170-202)
    for (auto &local_frame : this->frames_)
      for (auto &[sensor_id, landmarks] : local_frame->active_landmarks)
        for (auto &landmark : landmarks) landmark.to_marginalize = false;
    this->frames_.erase(std::remove_if(this->frames_.begin(), this->frames_.end(), [](auto &local_frame) { return local_frame->to_marginalize; }),
                        this->frames_.end());
    releaseUnusedPyramids();
  }

  // the base creates the LocalFrame of the new keyframe and the residual lists between it and every earlier frame
  Base::pushFrame(frame, level, model, frame_parameterization);

  // ... and the device receives exactly those: landmarks first (a connection may not be longer than its frame's landmark list)
  for (const auto &local_frame : this->frames_) uploadLandmarks(*local_frame);
  for (const auto &local_frame : this->frames_) uploadConnections(*local_frame);
}

template <energy::motion::Motion Motion, model::Model Model>
void HipPhotometricBundleAdjustment<Motion, Model>::updateLocalFrame(const track::ActiveKeyframe<Motion> &frame) {
  auto local_frame = this->getLocalFrame(frame.timestamp());
  CHECK(local_frame) << "Cannot update frame, there is no local copy in the solver";
  // eigen_photometric_bundle_adjustment.cpp:106-113
  local_frame->update(frame, frame.connections());
  local_frame->to_marginalize = frame.isMarginalized() && !local_frame->is_marginalized;
  local_frame->is_marginalized = frame.isMarginalized();
  // device: refreshed landmark flags + freshly matured landmarks, appended connection statuses, the frame flags
  uploadLandmarks(*local_frame);
  uploadConnections(*local_frame);
  if (frame.isMarginalized()) DSOPP_HIP_CHECKED(dsopp_hip_window_group_mark_frame_marginalized(group_, local_frame->id));
}

template <energy::motion::Motion Motion, model::Model Model>
Precision HipPhotometricBundleAdjustment<Motion, Model>::solve(const size_t number_of_threads) {
  (void)number_of_threads;
  CHECK(!this->frames_.empty());
  CHECK(this->frames_[0]->sensors().size() == 1);
  double energy = 0;
  int32_t iterations = 0, number_of_valid_residuals = 0;
  DSOPP_HIP_CHECKED(dsopp_hip_window_group_solve(group_, &energy, &iterations, &number_of_valid_residuals));
  writeBack();
  return static_cast<Precision>(energy);
}

template <energy::motion::Motion Motion, model::Model Model>
void HipPhotometricBundleAdjustment<Motion, Model>::uploadLandmarks(const Local &local_frame) {
  for (const auto &[sensor, landmarks] : local_frame.active_landmarks) {
    const size_t n = landmarks.size();
    // What travels: the flag byte of EVERY landmark (LocalFrame::update refreshes is_marginalized of the existing ones,
    // local_frame.hpp:489-497) but coordinates, inverse depth and patch of the landmarks the device does not hold yet only — the
    // C-ABI reads those arrays from index `held` on (include/dsopp_hip.h: dsopp_hip_window_set_landmarks), so the head of the
    // vectors below stays unwritten.  A keyframe update with no new landmarks marshals n bytes instead of 11 n doubles.
    int32_t held = 0;
    DSOPP_HIP_CHECKED(dsopp_hip_window_group_num_landmarks(group_, local_frame.id, &held));
    CHECK_LE(static_cast<size_t>(held), n) << "the device holds more landmarks than the local frame";
    std::vector<double> projection(2 * n), idepth(n), patch(static_cast<size_t>(Pattern::kSize) * n);
    std::vector<uint8_t> flags(n);
    for (size_t i = 0; i < n; ++i) {
      const auto &landmark = landmarks[i];
      flags[i] = static_cast<uint8_t>((landmark.is_marginalized ? 1 : 0) | (landmark.is_outlier ? 2 : 0));
      if (i < static_cast<size_t>(held)) continue;
      projection[2 * i] = static_cast<double>(landmark.projection[0]);
      projection[2 * i + 1] = static_cast<double>(landmark.projection[1]);
      idepth[i] = static_cast<double>(landmark.idepth);
      for (int k = 0; k < Pattern::kSize; ++k) patch[static_cast<size_t>(Pattern::kSize) * i + static_cast<size_t>(k)] = static_cast<double>(landmark.patch(k, 0));
    }
    // existing landmarks only have their flags refreshed (to_marginalize = newly marginalised && !outlier, the rule of
    // LocalFrame::update, local_frame.hpp:489-497), landmarks beyond the device's count are appended
    DSOPP_HIP_CHECKED(dsopp_hip_window_group_set_landmarks(group_, local_frame.id, static_cast<int32_t>(n), projection.data(), idepth.data(),
                                                           patch.data(), flags.data()));
  }
}

template <energy::motion::Motion Motion, model::Model Model>
void HipPhotometricBundleAdjustment<Motion, Model>::uploadConnections(const Local &local_frame) {
  for (const auto &[sensors, by_target] : local_frame.residuals) {
    for (const auto &[target_id, point_residuals] : by_target) {
      if (target_id == Local::kFrontendReferenceFrameId || target_id == Local::kFrontendTargetFrameId || point_residuals.empty()) continue;
      if (!this->getLocalFrame(target_id)) continue;  // the target has left the window
      std::vector<uint8_t> statuses(point_residuals.size());
      for (size_t i = 0; i < point_residuals.size(); ++i) statuses[i] = static_cast<uint8_t>(point_residuals[i].connection_status);
      // entries the device already holds are ignored, the tail is appended (photometric_bundle_adjustment.cpp:109-123,
      // local_frame.hpp:507-519)
      DSOPP_HIP_CHECKED(dsopp_hip_window_group_set_connection(group_, local_frame.id, target_id, static_cast<int32_t>(statuses.size()), statuses.data()));
    }
  }
}

template <energy::motion::Motion Motion, model::Model Model>
void HipPhotometricBundleAdjustment<Motion, Model>::writeBack() {
  using Scalar = Precision;
  for (auto &local_frame : this->frames_) {
    const int id = local_frame->id;
    // ---- frame state: after relinearizeSystem the newest frame's linearisation point has moved, so all three travel
    double T0[7], ab0[2], eps[DSOPP_HIP_BLOCK_SIZE], step[DSOPP_HIP_BLOCK_SIZE];
    DSOPP_HIP_CHECKED(dsopp_hip_window_group_get_frame_state(group_, id, T0, ab0, eps, step));
    local_frame->T_w_agent_linearization_point = fromParameters<typename Motion::template CastT<Scalar>>(T0);
    local_frame->affine_brightness0 = Eigen::Vector2<Scalar>(static_cast<Scalar>(ab0[0]), static_cast<Scalar>(ab0[1]));
    for (int i = 0; i < DSOPP_HIP_BLOCK_SIZE; ++i) {
      local_frame->state_eps[i] = static_cast<Scalar>(eps[i]);
      local_frame->state_eps_step[i] = 0;
    }
    // ---- landmarks + statuses of every connection in one packed transfer
    for (auto &[sensor, landmarks] : local_frame->active_landmarks) {
      const size_t n = landmarks.size();
      auto &by_target = local_frame->residuals[{sensor, sensor}];
      std::vector<int32_t> target_ids, short_target_ids;  // connections that cover every landmark ride in the packed transfer
      for (const auto &[target_id, point_residuals] : by_target) {
        if (target_id < 0 || point_residuals.empty() || !this->getLocalFrame(target_id)) continue;
        (point_residuals.size() == n ? target_ids : short_target_ids).push_back(target_id);
      }
      std::vector<double> idepth(n), inv_hessian(n), baseline(n);
      std::vector<int32_t> inliers(n);
      std::vector<uint8_t> flags(n), statuses(target_ids.size() * n);
      DSOPP_HIP_CHECKED(dsopp_hip_window_group_get_frame_update(group_, id, idepth.data(), inv_hessian.data(), baseline.data(), inliers.data(), flags.data(),
                                                          static_cast<int32_t>(target_ids.size()), target_ids.data(), statuses.data()));
      for (size_t i = 0; i < n; ++i) {
        auto &landmark = landmarks[i];
        landmark.is_outlier = (flags[i] & 2) != 0;
        landmark.ill_conditioned = (flags[i] & 8) != 0;
        if (landmark.is_marginalized) continue;  // the solver does not move marginalised landmarks
        landmark.idepth = static_cast<Scalar>(idepth[i]);
        landmark.idepth_step = 0;
        landmark.inv_hessian_idepth_idepth = static_cast<Scalar>(inv_hessian[i]);
        landmark.relative_baseline = static_cast<Scalar>(baseline[i]);
        landmark.number_of_inlier_residuals = static_cast<size_t>(inliers[i]);
      }
      for (size_t k = 0; k < target_ids.size(); ++k) {
        auto &point_residuals = by_target.at(target_ids[k]);
        for (size_t i = 0; i < n; ++i) {
          const auto status = static_cast<track::PointConnectionStatus>(statuses[k * n + i]);
          point_residuals[i].connection_status = status;
          point_residuals[i].connection_status_candidate = status;
        }
      }
      for (const int32_t target_id : short_target_ids) {  // (a connection shorter than the landmark list: one small read-back each)
        auto &point_residuals = by_target.at(target_id);
        std::vector<uint8_t> row(point_residuals.size());
        DSOPP_HIP_CHECKED(dsopp_hip_window_group_get_residuals(group_, id, target_id, static_cast<int32_t>(row.size()), row.data(), nullptr, nullptr));
        for (size_t i = 0; i < row.size(); ++i) {
          const auto status = static_cast<track::PointConnectionStatus>(row[i]);
          point_residuals[i].connection_status = status;
          point_residuals[i].connection_status_candidate = status;
        }
      }
    }
    // ---- covariances of the relative poses (updateFrame reads covariance_matrices.at(target) for every connected frame in
    // the window when estimate_uncertainty_, photometric_bundle_adjustment.cpp:206-208)
    if (this->estimate_uncertainty_) {
      for (const auto &other : this->frames_) {
        if (other->id == id) continue;
        double covariance[36];
        Eigen::Matrix<Scalar, Motion::Product::DoF, Motion::Product::DoF> matrix;
        if (dsopp_hip_window_group_get_covariance(group_, id, other->id, covariance) == DSOPP_HIP_OK) {
          matrix = Eigen::Map<const Eigen::Matrix<double, 6, 6, Eigen::RowMajor>>(covariance).template cast<Scalar>();
        } else {
          matrix.setZero();
        }
        local_frame->covariance_matrices[other->id] = matrix;
      }
    }
  }
}

template <energy::motion::Motion Motion, model::Model Model>
void HipPhotometricBundleAdjustment<Motion, Model>::releaseUnusedPyramids() {
  int32_t ids[DSOPP_HIP_MAX_FRAMES], n = 0;
  DSOPP_HIP_CHECKED(dsopp_hip_window_group_frame_ids(group_, DSOPP_HIP_MAX_FRAMES, ids, &n));
  for (auto it = pyramids_.begin(); it != pyramids_.end();) {
    if (std::find(ids, ids + n, it->first) == ids + n) {
      dsopp_hip_pyramid_group_destroy(it->second);
      it = pyramids_.erase(it);
    } else {
      ++it;
    }
  }
}

#undef DSOPP_HIP_CHECKED

template class HipPhotometricBundleAdjustment<energy::motion::SE3<Precision>, model::PinholeCamera<Precision>>;

}  // namespace problem
}  // namespace energy
}  // namespace dsopp
