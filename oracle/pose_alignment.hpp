// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// CPU restatement of the two-frame direct image alignment used per pyramid level by the coarse tracker:
// EigenPoseAlignment<SE3, Pinhole, 1, PixelMap, 1, true> — src/energy/problems/src/eigen_pose_alignment.cpp:26-329.
// PARITY UNPINNED (see pba.hpp header): pinned by identity tests on synthetic frames only.
#pragma once
#include <optional>
#include <vector>

#include "pba.hpp"

namespace oracle {

/** reference point of the aligner: LocalFrame::Landmark with PatternSize = 1 (local_frame.hpp:264-266) */
struct AlignPoint {
  double u, v, idepth, intensity;
};

/** LocalFrame depth-map constructor — PBA_INT/local_frame.hpp:350-393: every pixel of the reference depth map with
 *  weight > 0 and idepth >= 1e-6 inside a 4-px border becomes a point; its patch is the bilinear sample of the
 *  reference image at the (integer) pixel.  depth maps are W x H arrays {idepth sum, weight} (row-major y, x here). */
inline std::vector<AlignPoint> pointsFromDepthMap(const PixelMapView &image, const double *idepth_sum,
                                                  const double *weight) {
  const int kBorderSize = 4;
  const double kMinIdepth = 1e-6;
  std::vector<AlignPoint> pts;
  for (int y = kBorderSize; y < image.height - kBorderSize; ++y)
    for (int x = kBorderSize; x < image.width - kBorderSize; ++x) {
      const size_t i = static_cast<size_t>(y) * image.width + x;
      if (weight[i] > 0) {
        const double idepth = idepth_sum[i] / weight[i];
        if (idepth < kMinIdepth) continue;
        pts.push_back({static_cast<double>(x), static_cast<double>(y), idepth,
                       interpolateLinear1(image, static_cast<double>(x), static_cast<double>(y))});
      }
    }
  return pts;
}

struct AlignFrame {
  SE3 T_w_agent;
  double exposure_time = 1;
  double affine_brightness0[2] = {0, 0};
  PinholeModel model;
  PixelMapView grid;
  MaskView mask;
};

/** PoseAlignerProblem<SE3, Pinhole, 1, PixelMap, 1, true> — eigen_pose_alignment.cpp:26-241 */
struct PoseAlignerProblem {
  const AlignFrame &reference_frame;
  const AlignFrame &target_frame;
  const std::vector<AlignPoint> &points;
  double sigma_huber_loss;
  double affine_brightness_regularizer[2];
  SE3 &t_t_r;
  double *affine_brightness_eps;
  SE3 old_t_t_r;
  double old_affine_brightness_eps[2];
  NormalLinearSystem system{8};
  Vec step = Vec(8, 0.0);
  std::vector<char> success_statuses;
  std::vector<double> target_patches, d_intensity_u, d_intensity_v;

  PoseAlignerProblem(const AlignFrame &r, const AlignFrame &t, const std::vector<AlignPoint> &p, double sigma,
                     const double reg[2], SE3 &ttr, double *ab_eps)
      : reference_frame(r), target_frame(t), points(p), sigma_huber_loss(sigma), t_t_r(ttr), affine_brightness_eps(ab_eps),
        success_statuses(p.size(), 0), target_patches(p.size()), d_intensity_u(p.size()), d_intensity_v(p.size()) {
    affine_brightness_regularizer[0] = reg[0];
    affine_brightness_regularizer[1] = reg[1];
    old_t_t_r = t_t_r;
    old_affine_brightness_eps[0] = ab_eps[0];
    old_affine_brightness_eps[1] = ab_eps[1];
  }

  /** calculateEnergy — :55-108 */
  std::pair<double, int> calculateEnergy() {
    const double kSigmaHuberSqr = sigma_huber_loss * sigma_huber_loss;
    double energy = 0;
    int number_of_valid_residuals = 0;
    const ArrayReprojector<true> reprojector(reference_frame.model, target_frame.model, t_t_r);
    const double *rab = reference_frame.affine_brightness0;
    const double tab[2] = {target_frame.affine_brightness0[0] + affine_brightness_eps[0],
                           target_frame.affine_brightness0[1] + affine_brightness_eps[1]};
    const double brightness_change_scale =
        (target_frame.exposure_time / reference_frame.exposure_time) * std::exp(tab[0] - rab[0]);
    for (size_t i = 0; i < points.size(); ++i) {
      const AlignPoint &lm = points[i];
      double tu, tv;
      bool success = reprojector.reprojectPattern<1>(&lm.u, &lm.v, lm.idepth, &tu, &tv);
      success = success && target_frame.mask.valid(&tu, &tv, 1);
      success_statuses[i] = success;
      if (success) {
        double v3[3];
        interpolateLinear3(target_frame.grid, tu, tv, v3);
        target_patches[i] = v3[0];
        d_intensity_u[i] = v3[1];
        d_intensity_v[i] = v3[2];
        const double residual = (v3[0] - tab[1]) - brightness_change_scale * (lm.intensity - rab[1]);
        const double residuals_norm = std::abs(residual);
        const double residuals_squared_norm = residuals_norm * residuals_norm;
        if (residuals_squared_norm > kSigmaHuberSqr)
          energy += sigma_huber_loss * residuals_norm - kSigmaHuberSqr / 2;
        else
          energy += residuals_squared_norm / 2;
        number_of_valid_residuals++;
      }
    }
    energy += (tab[0] * affine_brightness_regularizer[0] * tab[0] + tab[1] * affine_brightness_regularizer[1] * tab[1]) / 2;
    return {energy, number_of_valid_residuals};
  }

  /** linearize — :110-192 (reuses the samples cached by the preceding calculateEnergy) */
  void linearize() {
    const double kSigmaHuberSqr = sigma_huber_loss * sigma_huber_loss;
    system.setZero();
    const ArrayReprojector<false> reprojector(reference_frame.model, target_frame.model, t_t_r);
    const double *rab = reference_frame.affine_brightness0;
    const double tab[2] = {target_frame.affine_brightness0[0] + affine_brightness_eps[0],
                           target_frame.affine_brightness0[1] + affine_brightness_eps[1]};
    const double brightness_change_scale =
        (target_frame.exposure_time / reference_frame.exposure_time) * std::exp(tab[0] - rab[0]);
    for (size_t i = 0; i < points.size(); ++i) {
      const AlignPoint &lm = points[i];
      double tu, tv, dui, dvi, duT[6], dvT[6];
      reprojector.reprojectPattern<1>(&lm.u, &lm.v, lm.idepth, &tu, &tv, &dui, &dvi, duT, dvT);
      if (!success_statuses[i]) continue;
      const double residuals_right = brightness_change_scale * (lm.intensity - rab[1]);
      const double residual = (target_patches[i] - tab[1]) - residuals_right;
      const bool huber_linear = residual * residual > kSigmaHuberSqr;
      const double huber_weight = huber_linear ? sigma_huber_loss / std::abs(residual) : 1;
      double d_state[8];
      for (int c = 0; c < 6; ++c) d_state[c] = -(d_intensity_u[i] * duT[c] + d_intensity_v[i] * dvT[c]);
      d_state[6] = -residuals_right;
      d_state[7] = -1;
      for (int a = 0; a < 8; ++a) {
        for (int b = 0; b < 8; ++b) system.H(a, b) += huber_weight * (d_state[a] * d_state[b]);
        system.b[static_cast<size_t>(a)] += huber_weight * (d_state[a] * residual);
      }
    }
    for (int a = 0; a < 2; ++a) {
      system.H(6 + a, 6 + a) += affine_brightness_regularizer[a];
      system.b[static_cast<size_t>(6 + a)] += affine_brightness_regularizer[a] * tab[a];
    }
  }

  /** calculateStep — :194-206 */
  void calculateStep(double lambda) {
    NormalLinearSystem reg = system;
    for (int a = 0; a < 8; ++a) reg.H(a, a) += system.H(a, a) * lambda;
    step = reg.solve();
    old_t_t_r = t_t_r;
    t_t_r = t_t_r.leftIncrement(step.data());
    old_affine_brightness_eps[0] = affine_brightness_eps[0];
    old_affine_brightness_eps[1] = affine_brightness_eps[1];
    affine_brightness_eps[0] -= step[6];
    affine_brightness_eps[1] -= step[7];
  }
  /** acceptStep — :208-212 */
  std::pair<double, double> acceptStep() {
    const double a0 = target_frame.affine_brightness0[0] + old_affine_brightness_eps[0];
    const double a1 = target_frame.affine_brightness0[1] + old_affine_brightness_eps[1];
    return {a0 * a0 + a1 * a1, dot(step, step)};
  }
  /** rejectStep — :214-217 */
  void rejectStep() {
    t_t_r = old_t_t_r;
    affine_brightness_eps[0] = old_affine_brightness_eps[0];
    affine_brightness_eps[1] = old_affine_brightness_eps[1];
  }
};

struct AlignResult {
  double rmse = 0;
  SE3 T_w_target;
  double affine_brightness[2] = {0, 0};
  double covariance[36] = {0};
  double H[64] = {0};
  LmResult lm;
};

/** EigenPoseAlignment::solve — eigen_pose_alignment.cpp:275-329; production options tracker/src/fabric.cpp:127-134 */
inline AlignResult alignSolve(const AlignFrame &reference_frame, AlignFrame &target_frame,
                              const std::vector<AlignPoint> &points, const PbaOptions &opt,
                              const double *prior_rotation_t_r /* 3x3 row-major or nullptr */) {
  LmOptions options;
  options.initial_levenberg_marquardt_regularizer = 1. / opt.initial_trust_region_radius;
  options.function_tolerance = opt.function_tolerance;
  options.parameter_tolerance = opt.parameter_tolerance;
  options.max_num_iterations = static_cast<size_t>(opt.max_iterations);
  options.levenberg_marquardt_regularizer_decrease_on_accept = 2.;
  options.levenberg_marquardt_regularizer_increase_on_reject = 2.;
  SE3 t_t_r = target_frame.T_w_agent.inverse() * reference_frame.T_w_agent;
  if (prior_rotation_t_r) t_t_r.setRotationMatrix(prior_rotation_t_r);  // eigen_pose_alignment.cpp:309-311
  double affine_brightness_eps[2] = {0, 0};
  PoseAlignerProblem problem(reference_frame, target_frame, points, opt.sigma_huber_loss,
                             opt.affine_brightness_regularizer, t_t_r, affine_brightness_eps);
  AlignResult out;
  out.lm = lmSolve(problem, options);
  Mat pinv = pseudoInverseCOD(problem.system.H);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) out.covariance[6 * i + j] = pinv(i, j);
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j) out.H[8 * i + j] = problem.system.H(i, j);
  target_frame.T_w_agent = reference_frame.T_w_agent * t_t_r.inverse();
  target_frame.affine_brightness0[0] += affine_brightness_eps[0];
  target_frame.affine_brightness0[1] += affine_brightness_eps[1];
  out.T_w_target = target_frame.T_w_agent;
  out.affine_brightness[0] = target_frame.affine_brightness0[0];
  out.affine_brightness[1] = target_frame.affine_brightness0[1];
  out.rmse = std::sqrt(out.lm.energy / static_cast<double>(out.lm.number_of_valid_residuals) / 1);
  return out;
}

/** initializationPoses — src/tracker/tracker/src/monocular_tracker.cpp:136-176: the pose hypotheses estimatePose tries in
 *  turn.  have_two_frames == false (track.frames().size() < 2) -> {identity}. */
inline std::vector<SE3> initializationPoses(bool have_two_frames, const SE3 &t_w_previous /* getFrame(-2) */, const SE3 &t_w_last /* getFrame(-1) */,
                                            const SE3 &t_w_keyframe /* lastKeyframe() */) {
  if (!have_two_frames) return {SE3()};
  const double kMinAnglePerturbation = 1. * M_PI / 180.;
  const double kMaxAnglePerturbation = 3. * M_PI / 180.;
  const double kAnglePerturbationsStep = 0.5 * M_PI / 180.;
  std::vector<SE3> initializations;
  const SE3 t_r_t_prev = t_w_previous.inverse() * t_w_last;
  initializations.push_back(t_w_last * t_r_t_prev);               // previous motion
  initializations.push_back(t_w_last * t_r_t_prev * t_r_t_prev);  // double previous motion (frame skipped)
  double xi[6];
  t_r_t_prev.log(xi);
  for (double &v : xi) v *= 0.5;
  initializations.push_back(t_w_last * SE3::exp(xi));             // half motion
  initializations.push_back(t_w_last);                            // zero motion
  initializations.push_back(t_w_keyframe);                        // zero motion from keyframe
  for (double delta = kMinAnglePerturbation; delta < kMaxAnglePerturbation; delta += kAnglePerturbationsStep)
    for (double rx : {0.0, delta, -delta})
      for (double ry : {0.0, delta, -delta})
        for (double rz : {0.0, delta, -delta}) {
          const double rot[6] = {0, 0, 0, rx, ry, rz};
          initializations.push_back(initializations[0] * SE3::exp(rot));  // SE3(SO3::exp(.), 0)
        }
  return initializations;
}

}  // namespace oracle
