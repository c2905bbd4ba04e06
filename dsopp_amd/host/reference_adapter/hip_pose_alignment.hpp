// HipPoseAlignment — the MI355X backend behind DSOPP's PoseAlignment interface (coarse-to-fine tracker, one pyramid level per
// solve).  Reference tree location:
//   src/energy/problems/include/energy/problems/pose_alignment/hip_pose_alignment.hpp   (+ src/hip_pose_alignment.cpp)
// selected by `pose_alignment: {solver: hip}` (fabric_hip.patch).  Compiled only where the reference and its dependencies exist.
//
// estimatePose (monocular_tracker.cpp:179-245) drives the solver through a PoseAlignment* with
//   reset(); pushFrame(keyframe time, pose, pyramids, masks, DEPTH MAPS, ...); pushFrame(new frame time, pose, pyramids, masks, ...);
//   solve(); getPose(time); getAffineBrightness(time)
// of which only reset / solve / pushKnownPose / setRotationPrior are virtual: both pushFrame overloads and both getters are
// non-virtual members of the PhotometricBundleAdjustment base and work on its `frames_`
// (photometric_bundle_adjustment.hpp:71-115,134,141).  So this class lets the base build the two LocalFrames exactly as it
// does for EigenPoseAlignment (the depth-map constructor scans the map into reference points, local_frame.hpp:350-393), and
// its solve() reads its input FROM those LocalFrames — reference points, both image levels, the target mask, poses, photometric
// parameters — runs the Levenberg-Marquardt alignment on the device and writes the result back into the target LocalFrame the
// way eigen_pose_alignment.cpp:325-327 does.  getPose / getAffineBrightness then return it, unchanged.
//
// The device images are cached per (frame timestamp, PixelMap level): the keyframe's levels are uploaded once per keyframe, the
// new frame's once per frame, however many of the 113 initialisations estimatePose tries.
#ifndef DSOPP_HIP_POSE_ALIGNMENT_HPP
#define DSOPP_HIP_POSE_ALIGNMENT_HPP

#include <list>
#include <optional>

#include "energy/problems/photometric_bundle_adjustment/trust_region_photometric_bundle_adjustment_options.hpp"
#include "energy/problems/pose_alignment/pose_alignment.hpp"

struct dsopp_hip_aligner;
struct dsopp_hip_pyramid;

namespace dsopp {
namespace energy {
namespace problem {

/** \brief Two-frame direct image alignment on an MI355X: EigenPoseAlignment<Motion, Model, 1, PixelMap, 1, true>'s drop-in. */
template <energy::motion::Motion Motion, model::Model Model>
class HipPoseAlignment : public PoseAlignment<Motion, Model, 1, features::PixelMap, 1> {
 public:
  /** the interface this backend plugs into */
  using Base = PoseAlignment<Motion, Model, 1, features::PixelMap, 1>;
  /**
   * @param trust_region_options the options EigenPoseAlignment takes (fabric.cpp:127-142)
   * @param device HIP device index
   */
  explicit HipPoseAlignment(const TrustRegionPhotometricBundleAdjustmentOptions<Precision> &trust_region_options, int device = 0);
  ~HipPoseAlignment() override;

  /** pose_alignment.hpp:34: root mean square error in energy per pixel, or kZeroCost for a known pose */
  Precision solve(const size_t number_of_threads) override;
  /** eigen_pose_alignment.cpp:254-257 */
  void setRotationPrior(const Eigen::Matrix3<Precision> &r_t_r) override;
  /** eigen_pose_alignment.cpp:261-264 */
  void reset() override;
  /** eigen_pose_alignment.cpp:268-271 */
  void pushKnownPose(time timestamp, const Motion &t_w_agent) override;
  /** @return covariance of the relative pose t_t_r of the last solve (EigenPoseAlignment::tTargetReferenceCovariance) */
  Eigen::Matrix<Precision, Motion::DoF, Motion::DoF> tTargetReferenceCovariance() const { return covariance_t_t_r_; }

 private:
  /** device image of one PixelMap level of the frame captured at `timestamp` (uploaded on first use) */
  dsopp_hip_pyramid *deviceLevel(time timestamp, const features::PixelMap<1> *level, const sensors::calibration::CameraMask *mask);

  struct CachedLevel {
    time timestamp;
    const features::PixelMap<1> *level;
    long width;
    bool masked;
    dsopp_hip_pyramid *pyramid;
  };
  std::list<CachedLevel> cache_;  // most recently used first
  dsopp_hip_aligner *aligner_ = nullptr;
  int device_ = 0;
  std::optional<Eigen::Matrix3<Precision>> prior_rotation_t_r_;
  Eigen::Matrix<Precision, Motion::DoF, Motion::DoF> covariance_t_t_r_ = Eigen::Matrix<Precision, Motion::DoF, Motion::DoF>::Zero();
};

}  // namespace problem
}  // namespace energy
}  // namespace dsopp

#endif  // DSOPP_HIP_POSE_ALIGNMENT_HPP
