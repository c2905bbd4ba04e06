// HipPhotometricBundleAdjustment — the MI355X backend behind DSOPP's own bundle-adjustment interface.
//
// Lives in the reference tree as
//   src/energy/problems/include/energy/problems/photometric_bundle_adjustment/hip_photometric_bundle_adjustment.hpp
// (implementation: hip_photometric_bundle_adjustment.cpp next to eigen_photometric_bundle_adjustment.cpp) and is selected by
// `photometric_bundle_adjustment: {solver: hip}` through the factory branch of fabric_hip.patch.  It is compiled only where the
// reference and its dependencies exist (Eigen, Sophus, glog — none of them are in this repository's build image, so this file
// has never been compiled here).
//
// How it stays a drop-in for the UNCHANGED reference.  The tracker reaches the solver through a pointer to the base class
// and calls, besides the virtual pushFrame / updateLocalFrame / solve, three NON-virtual services of that base:
//   updateFrame(ActiveKeyframe&), getPose(time), getAffineBrightness(time)
//     (photometric_bundle_adjustment.hpp:121,134,141; called from monocular_tracker.cpp:215-216,252-255)
// which read the base's own `frames_` (a deque of LocalFrame, :181).  Shadowing them in a derived class is never reached.  So
// this class keeps `frames_` exactly as EigenPhotometricBundleAdjustment keeps it — every LocalFrame, landmark and residual
// list is created by the base's own code (Base::pushFrame, LocalFrame::update) — mirrors each change into the device window
// through the C-ABI, and after every solve() writes the device results back into those LocalFrames: poses and affine
// brightness (linearisation point + state_eps), inverse depths, H_dd^-1, relative baselines, inlier counts, outlier flags,
// connection statuses and the relative-pose covariances.  The three services then work unchanged, on current data.
//
// More than one GPU.  The tracker owns ONE solver object in ONE process (fabric.cpp:58-121, dsopp_main.cpp:114-119), so the
// multi-GPU form lives behind this same class: `devices` (YAML `photometric_bundle_adjustment: {solver: hip, devices: "0 1 2 3"}`)
// lists the HIP devices, the device side is a dsopp_hip_window_group — landmarks dealt round-robin over the devices, frames and
// images replicated, one RCCL all-reduce of the reduced system per Gauss-Newton iteration.  A single device (the default) is a
// group of one, i.e. a plain window on the calling thread.
#ifndef DSOPP_HIP_PHOTOMETRIC_BUNDLE_ADJUSTMENT_HPP
#define DSOPP_HIP_PHOTOMETRIC_BUNDLE_ADJUSTMENT_HPP

#include <map>
#include <vector>

#include "common/pattern/pattern.hpp"
#include "energy/motion/motion.hpp"
#include "energy/problems/photometric_bundle_adjustment/photometric_bundle_adjustment.hpp"
#include "energy/problems/photometric_bundle_adjustment/trust_region_photometric_bundle_adjustment_options.hpp"

struct dsopp_hip_window;
struct dsopp_hip_window_group;
struct dsopp_hip_pyramid_group;

namespace dsopp {
namespace energy {
namespace problem {

/** \brief Photometric bundle adjustment of the keyframe window on an MI355X (hand-written HIP kernels behind dsopp_hip.h).
 *
 * Same template switches as the production EigenPhotometricBundleAdjustment instance of fabric.cpp:58-100:
 * PatternSize = Pattern::kSize, Grid2D = PixelMap, OPTIMIZE_POSES, OPTIMIZE_IDEPTHS, FIRST_ESTIMATE_JACOBIANS, C = 1.
 */
template <energy::motion::Motion Motion, model::Model Model>
class HipPhotometricBundleAdjustment
    : public PhotometricBundleAdjustment<Precision, Motion, Model, Pattern::kSize, features::PixelMap, true, true, true, 1> {
 public:
  /** the base class this backend plugs into */
  using Base = PhotometricBundleAdjustment<Precision, Motion, Model, Pattern::kSize, features::PixelMap, true, true, true, 1>;
  /** local copy of a keyframe as the base keeps it */
  using Local = LocalFrame<Precision, Motion, Model, Pattern::kSize, features::PixelMap, 1>;

  /**
   * @param trust_region_options the options EigenPhotometricBundleAdjustment takes
   * @param estimate_uncertainty estimate pose / idepth uncertainty after the optimisation
   * @param force_accept accept every LM iteration
   * @param devices HIP device indices the landmarks are sharded over (one entry = single-GPU)
   */
  HipPhotometricBundleAdjustment(const TrustRegionPhotometricBundleAdjustmentOptions<Precision> &trust_region_options,
                                 bool estimate_uncertainty = false, bool force_accept = false, const std::vector<int> &devices = {0});
  ~HipPhotometricBundleAdjustment() override;

  /** photometric_bundle_adjustment.hpp:55-56; as eigen_photometric_bundle_adjustment.cpp:119-141 it first folds the frames and
   *  landmarks flagged for marginalisation into the prior */
  void pushFrame(const track::ActiveKeyframe<Motion> &frame, size_t level, const Model &model,
                 FrameParameterization frame_parameterization = FrameParameterization::kFree) override;
  /** photometric_bundle_adjustment.hpp:127; eigen_photometric_bundle_adjustment.cpp:106-113 */
  void updateLocalFrame(const track::ActiveKeyframe<Motion> &frame) override;
  /** photometric_bundle_adjustment.hpp:154: LM loop, relinearisation, covariances, point statuses — on the device; then the
   *  write-back into `frames_`.  number_of_threads is accepted and ignored, as in the Eigen backend */
  Precision solve(const size_t number_of_threads) override;

  /** the device side: all calls of the window go through the group (a group of one shard is a plain window) */
  dsopp_hip_window_group *group() const { return group_; }
  /** single-device set-ups only (nullptr otherwise): the one window, for the device-resident landmark activation of
   *  INTEGRATION.md §2d, which reads ALL active landmarks of the window on one device */
  dsopp_hip_window *window() const;

 private:
  /** landmark arrays and connection statuses of one local frame -> device (only what the device does not hold yet travels) */
  void uploadLandmarks(const Local &local_frame);
  void uploadConnections(const Local &local_frame);
  /** device results -> the LocalFrames of `frames_` */
  void writeBack();
  /** drops the device pyramids of frames the window has erased */
  void releaseUnusedPyramids();

  dsopp_hip_window_group *group_ = nullptr;
  std::vector<int> devices_;
  /** device texel images of the keyframes in the window (one copy per device of the group), by keyframe id; their lifetime is the
   *  frame's stay in the window (the reference keeps raw pointers into the keyframe's PixelMap for the same span,
   *  local_frame.hpp:323-325) */
  std::map<int, dsopp_hip_pyramid_group *> pyramids_;
};

}  // namespace problem
}  // namespace energy
}  // namespace dsopp

#endif  // DSOPP_HIP_PHOTOMETRIC_BUNDLE_ADJUSTMENT_HPP
