// Implementation of HipPoseAlignment (see the header).  Reference tree location: src/energy/problems/src/hip_pose_alignment.cpp
#include "energy/problems/pose_alignment/hip_pose_alignment.hpp"

#include <dsopp_hip.h>
#include <glog/logging.h>

#include <array>
#include <vector>

#include "energy/camera_model/pinhole/pinhole_camera.hpp"
#include "energy/problems/photometric_bundle_adjustment/local_frame.hpp"

namespace dsopp {
namespace energy {
namespace problem {
namespace {

#define DSOPP_HIP_CHECKED(call) CHECK_EQ((call), DSOPP_HIP_OK) << #call << ": " << dsopp_hip_last_error()

template <typename MotionT>
std::array<double, 7> toParameters(const MotionT &motion) {  // Sophus storage order (qx, qy, qz, qw, tx, ty, tz)
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

constexpr size_t kCachedLevels = 16;  // two frames x up to PixelDataFrame::kMaxPyramidDepth levels, and some slack

}  // namespace

template <energy::motion::Motion Motion, model::Model Model>
HipPoseAlignment<Motion, Model>::HipPoseAlignment(const TrustRegionPhotometricBundleAdjustmentOptions<Precision> &trust_region_options, int device)
    : device_(device) {
  dsopp_hip_options options;
  dsopp_hip_default_align_options(&options);
  options.max_iterations = static_cast<int32_t>(trust_region_options.max_iterations);
  options.initial_trust_region_radius = static_cast<double>(trust_region_options.initial_trust_region_radius);
  options.function_tolerance = static_cast<double>(trust_region_options.function_tolerance);
  options.parameter_tolerance = static_cast<double>(trust_region_options.parameter_tolerance);
  options.affine_brightness_regularizer[0] = static_cast<double>(trust_region_options.affine_brightness_regularizer[0]);
  options.affine_brightness_regularizer[1] = static_cast<double>(trust_region_options.affine_brightness_regularizer[1]);
  options.fixed_state_regularizer = static_cast<double>(trust_region_options.fixed_state_regularizer);
  options.sigma_huber_loss = static_cast<double>(trust_region_options.sigma_huber_loss);
  options.dtype = DSOPP_HIP_F64;
  DSOPP_HIP_CHECKED(dsopp_hip_aligner_create(&options, device_, nullptr, &aligner_));
}

template <energy::motion::Motion Motion, model::Model Model>
HipPoseAlignment<Motion, Model>::~HipPoseAlignment() {
  dsopp_hip_aligner_destroy(aligner_);
  for (auto &entry : cache_) dsopp_hip_pyramid_destroy(entry.pyramid);
}

template <energy::motion::Motion Motion, model::Model Model>
void HipPoseAlignment<Motion, Model>::setRotationPrior(const Eigen::Matrix3<Precision> &r_t_r) {
  prior_rotation_t_r_ = r_t_r;
}

template <energy::motion::Motion Motion, model::Model Model>
void HipPoseAlignment<Motion, Model>::reset() {
  prior_rotation_t_r_ = std::nullopt;
  this->frames_.clear();
  DSOPP_HIP_CHECKED(dsopp_hip_aligner_reset(aligner_));
}

template <energy::motion::Motion Motion, model::Model Model>
void HipPoseAlignment<Motion, Model>::pushKnownPose(time timestamp, const Motion &t_w_agent) {
  this->timestamp_t_w_agents_.pushData(timestamp, t_w_agent);
}

template <energy::motion::Motion Motion, model::Model Model>
dsopp_hip_pyramid *HipPoseAlignment<Motion, Model>::deviceLevel(time timestamp, const features::PixelMap<1> *level,
                                                                const sensors::calibration::CameraMask *mask) {
  for (auto it = cache_.begin(); it != cache_.end(); ++it) {
    if (it->timestamp == timestamp && it->level == level && it->width == level->width() && it->masked == (mask != nullptr)) {
      cache_.splice(cache_.begin(), cache_, it);
      return cache_.front().pyramid;
    }
  }
  const int width = static_cast<int>(level->width()), height = static_cast<int>(level->height());
  dsopp_hip_pyramid *pyramid = nullptr;
  if (cache_.size() >= kCachedLevels) {  // recycle the least recently used entry of the same size, else drop it
    auto &victim = cache_.back();
    int w = 0, h = 0;
    DSOPP_HIP_CHECKED(dsopp_hip_pyramid_level_size(victim.pyramid, 0, &w, &h));
    if (w == width && h == height)
      pyramid = victim.pyramid;
    else
      dsopp_hip_pyramid_destroy(victim.pyramid);
    cache_.pop_back();
  }
  if (!pyramid) DSOPP_HIP_CHECKED(dsopp_hip_pyramid_create(device_, nullptr, width, height, 1, DSOPP_HIP_F64, &pyramid));
  const size_t n = static_cast<size_t>(width) * static_cast<size_t>(height);
  // PixelInfo<1>::data_ = (I, dI/dx, dI/dy), three contiguous scalars per pixel, row-major (features/src/pixel_map.cpp:105-110)
  if constexpr (std::is_same_v<Precision, double>) {
    DSOPP_HIP_CHECKED(dsopp_hip_pyramid_set_level(pyramid, 0, reinterpret_cast<const double *>(&(*level)(size_t(0)))));
  } else {
    std::vector<double> pixelinfo(3 * n);
    const Precision *source = reinterpret_cast<const Precision *>(&(*level)(size_t(0)));
    std::copy(source, source + 3 * n, pixelinfo.begin());
    DSOPP_HIP_CHECKED(dsopp_hip_pyramid_set_level(pyramid, 0, pixelinfo.data()));
  }
  const uint8_t *mask_bytes = nullptr;
  cv::Mat continuous;
  if (mask && !mask->data().empty()) {
    CHECK(mask->data().rows == height && mask->data().cols == width) << "mask and image level differ in size";
    continuous = mask->data().isContinuous() ? mask->data() : mask->data().clone();
    mask_bytes = continuous.ptr<uint8_t>();
  }
  DSOPP_HIP_CHECKED(dsopp_hip_pyramid_set_mask(pyramid, 0, mask_bytes));  // (NULL = all valid: also clears a recycled entry's mask)
  cache_.push_front(CachedLevel{timestamp, level, level->width(), mask != nullptr, pyramid});
  return pyramid;
}

template <energy::motion::Motion Motion, model::Model Model>
Precision HipPoseAlignment<Motion, Model>::solve(const size_t number_of_threads) {
  (void)number_of_threads;
  CHECK(!this->frames_.empty());
  auto &reference_frame = *this->frames_[0];
  auto &target_frame = *this->frames_.back();

  // eigen_pose_alignment.cpp:282-289: a pose pushed with pushKnownPose wins
  auto target_found = this->timestamp_t_w_agents_.getData(target_frame.timestamp);
  if (target_found) {
    target_frame.T_w_agent_linearization_point = *target_found;
    target_frame.state_eps.setZero();
    return Base::kZeroCost;
  }
  CHECK_EQ(this->frames_.size(), 2u);
  CHECK(reference_frame.frame_parameterization == FrameParameterization::kFixed);
  CHECK(target_frame.frame_parameterization == FrameParameterization::kFree);
  CHECK_EQ(reference_frame.sensors().size(), 1u);
  const size_t sensor_id = reference_frame.sensors()[0];

  // ---- inputs, read from the LocalFrames the base built
  const auto &landmarks = reference_frame.active_landmarks.at(sensor_id);
  const size_t n = landmarks.size();
  std::vector<double> u(n), v(n), idepth(n);
  for (size_t i = 0; i < n; ++i) {
    u[i] = static_cast<double>(landmarks[i].projection[0]);
    v[i] = static_cast<double>(landmarks[i].projection[1]);
    idepth[i] = static_cast<double>(landmarks[i].idepth);
  }
  dsopp_hip_pyramid *reference_image = deviceLevel(reference_frame.timestamp, reference_frame.grids.at(sensor_id), nullptr);
  dsopp_hip_pyramid *target_image = deviceLevel(target_frame.timestamp, target_frame.grids.at(sensor_id), &target_frame.masks.at(sensor_id));
  auto intrinsicsOf = [](const auto &frame) {
    return std::array<double, 4>{static_cast<double>(frame.intrinsic_parameters[0]), static_cast<double>(frame.intrinsic_parameters[1]),
                                 static_cast<double>(frame.intrinsic_parameters[2]), static_cast<double>(frame.intrinsic_parameters[3])};
  };
  const auto reference_intrinsics = intrinsicsOf(reference_frame), target_intrinsics = intrinsicsOf(target_frame);
  const auto reference_pose = toParameters(reference_frame.T_w_agent_linearization_point);
  const auto target_pose = toParameters(target_frame.T_w_agent_linearization_point);
  const double reference_affine[2] = {static_cast<double>(reference_frame.affine_brightness0[0]), static_cast<double>(reference_frame.affine_brightness0[1])};
  const double target_affine[2] = {static_cast<double>(target_frame.affine_brightness0[0]), static_cast<double>(target_frame.affine_brightness0[1])};

  // ---- device solve.  reset() also clears the rotation prior on the device, so it is (re-)set here, before the frames
  DSOPP_HIP_CHECKED(dsopp_hip_aligner_reset(aligner_));
  if (prior_rotation_t_r_) {
    double rotation[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) rotation[3 * r + c] = static_cast<double>((*prior_rotation_t_r_)(r, c));
    DSOPP_HIP_CHECKED(dsopp_hip_aligner_set_rotation_prior(aligner_, rotation));
  }
  // the reference points keep the order of the LocalFrame's depth-map scan; their intensities are sampled on the device at
  // the same integer pixels PatternPatch::getIntensities read on the host (a bilinear sample at an integer position is that pixel)
  DSOPP_HIP_CHECKED(dsopp_hip_aligner_push_reference_points(aligner_, ticks(reference_frame.timestamp), reference_pose.data(), reference_image, 0,
                                                            reference_intrinsics.data(), static_cast<int32_t>(n), u.data(), v.data(), idepth.data(),
                                                            static_cast<double>(reference_frame.exposure_time), reference_affine));
  DSOPP_HIP_CHECKED(dsopp_hip_aligner_push_target(aligner_, ticks(target_frame.timestamp), target_pose.data(), target_image, 0, target_intrinsics.data(),
                                                  static_cast<double>(target_frame.exposure_time), target_affine));
  dsopp_hip_align_result result;
  DSOPP_HIP_CHECKED(dsopp_hip_aligner_solve(aligner_, &result));

  // ---- write-back (eigen_pose_alignment.cpp:320-328)
  covariance_t_t_r_ = Eigen::Map<const Eigen::Matrix<double, 6, 6, Eigen::RowMajor>>(result.covariance).template cast<Precision>();
  target_frame.T_w_agent_linearization_point = fromParameters<std::decay_t<decltype(target_frame.T_w_agent_linearization_point)>>(result.T_world_target);
  target_frame.affine_brightness0 = Eigen::Vector2<Precision>(static_cast<Precision>(result.affine_brightness[0]), static_cast<Precision>(result.affine_brightness[1]));
  target_frame.state_eps.setZero();
  return static_cast<Precision>(result.rmse);
}

#undef DSOPP_HIP_CHECKED

template class HipPoseAlignment<energy::motion::SE3<Precision>, model::PinholeCamera<Precision>>;

}  // namespace problem
}  // namespace energy
}  // namespace dsopp
