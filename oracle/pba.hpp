// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// CPU restatement of the reference's sliding-window photometric bundle adjustment (Eigen backend), stage by stage,
// with the reference's data layout (materialised per-(landmark,target) ResidualPoint AoS, Kahan accumulators) so it
// can double as the "port" CPU baseline.  PARITY UNPINNED: the reference cannot be compiled in this image (no
// Eigen/Sophus/TBB/glog/Ceres/OpenCV) and ships no golden vectors for this path (its solver tests need the un-shipped
// track30seconds data) — the oracle is pinned by identity tests restating the reference's own test
// assertions on synthetic windows (tests/test_oracle_*.py) and by an independently written NumPy spec (oracle/spec.py).
//
// Path shorthands: PBA_INT = src/energy/problems/internal/energy/problems/photometric_bundle_adjustment,
//                  PROB_SRC = src/energy/problems/src.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include "geometry.hpp"
#include "linalg.hpp"
#include "parallel.hpp"

namespace oracle {

/** ResidualPoint<double, SE3, 8, 1> — PBA_INT/local_frame.hpp:173-220 */
struct ResidualPoint {
  uint8_t connection_status = kOk;
  uint8_t connection_status_candidate = kOk;
  double residuals[kPatternSize] = {0};
  double d_u_idepth[kPatternSize];
  double d_v_idepth[kPatternSize];
  double d_u_tReferenceTarget[kPatternSize * kDoF];  // 8 x 6 row-major
  double d_v_tReferenceTarget[kPatternSize * kDoF];
  bool reprojection_jacobians_valid = false;
  bool was_estimated = false;
  double d_reference_state_eps[kPatternSize * kBlockSize] = {0};  // 8 x 8 row-major
  double d_target_state_eps[kPatternSize * kBlockSize] = {0};
  double d_idepth[kPatternSize] = {0};
  double huber_weight = 1;
  double energy = 0;
  double brightness_change_scale = 0;
  explicit ResidualPoint(uint8_t status = kOk) : connection_status(status), connection_status_candidate(status) {}
};

/** LocalFrame::Landmark — PBA_INT/local_frame.hpp:244-300 */
struct Landmark {
  double projection[2];
  double idepth;
  double idepth_step = 0;
  double patch[kPatternSize];
  bool is_marginalized;
  bool to_marginalize = false;
  bool is_outlier;
  double ref_u[kPatternSize], ref_v[kPatternSize];  // reference_pattern = projection + offsets (:261-266)
  double corrected_intensities[kPatternSize] = {0};
  double relative_baseline = 0;
  double inv_hessian_idepth_idepth = 0;
  bool ill_conditioned = false;
  double b_idepth_block = 0;
  std::vector<double> hessian_poses_idepth_block;
  size_t number_of_inlier_residuals = 0;

  Landmark(const double uv[2], double _idepth, const double *_patch, bool _is_marginalized, bool _is_outlier)
      : idepth(_idepth), is_marginalized(_is_marginalized), is_outlier(_is_outlier) {
    projection[0] = uv[0];
    projection[1] = uv[1];
    for (int i = 0; i < kPatternSize; ++i) {
      patch[i] = _patch[i];
      ref_u[i] = kPatternData[2 * i] + uv[0];
      ref_v[i] = kPatternData[2 * i + 1] + uv[1];
    }
  }
};

/** LocalFrame — PBA_INT/local_frame.hpp:232-584 (single sensor) */
struct LocalFrame {
  int id = 0;
  int64_t timestamp = 0;
  SE3 T_w_agent_linearization_point;
  double exposure_time = 1;
  double affine_brightness0[2] = {0, 0};
  PinholeModel model;
  double state_eps[kBlockSize] = {0};
  double state_eps_step[kBlockSize] = {0};
  PixelMapView grid;
  MaskView mask;
  bool is_marginalized = false;
  bool to_marginalize = false;
  bool fixed = false;  // FrameParameterization::kFixed
  std::vector<Landmark> active_landmarks;
  std::map<int, std::vector<ResidualPoint>> residuals;  // [target frame id][reference point id]
  std::map<int, std::array<double, 36>> covariance_matrices;

  /** tWorldAgent — local_frame.hpp:525-527 */
  SE3 tWorldAgent() const { return T_w_agent_linearization_point.rightIncrement(state_eps); }
  void affineBrightness(double ab[2]) const {
    ab[0] = affine_brightness0[0] + state_eps[6];
    ab[1] = affine_brightness0[1] + state_eps[7];
  }
};

using Frames = std::deque<std::unique_ptr<LocalFrame>>;

/** MatrixAccumulator (Kahan) — src/energy/problems/internal/matrix_accumulator.hpp:16-62 */
template <int N>
struct KahanFixed {
  double sum[N], comp[N];
  KahanFixed() {
    for (int i = 0; i < N; ++i) sum[i] = comp[i] = 0;
  }
  void add(const double *summand) {
    for (int i = 0; i < N; ++i) {
      const double y = summand[i] - comp[i];
      const double t = sum[i] + y;
      comp[i] = (t - sum[i]) - y;
      sum[i] = t;
    }
  }
};
struct KahanDynamic {
  std::vector<double> sum, comp, y, t;
  explicit KahanDynamic(size_t n) : sum(n, 0.0), comp(n, 0.0), y(n), t(n) {}
  template <typename F>
  void add(F summand) {
    const size_t n = sum.size();
    for (size_t i = 0; i < n; ++i) y[i] = summand(i) - comp[i];
    for (size_t i = 0; i < n; ++i) t[i] = sum[i] + y[i];
    for (size_t i = 0; i < n; ++i) comp[i] = (t[i] - sum[i]) - y[i];
    for (size_t i = 0; i < n; ++i) sum[i] = t[i];
  }
};

/** firstEstimateJacobians_ — PBA_INT/first_estimate_jacobians.hpp:14-71 */
inline void firstEstimateJacobians(Frames &frames) {
  // flattened (reference frame, target frame, landmark chunk) task list; the reference nests two parallel_for
  for (auto &reference_frame_ptr : frames) {
    LocalFrame &reference_frame = *reference_frame_ptr;
    for (auto &target_frame_ptr : frames) {
      LocalFrame &target_frame = *target_frame_ptr;
      if (reference_frame.id == target_frame.id) continue;
      auto it = reference_frame.residuals.find(target_frame.id);
      if (it == reference_frame.residuals.end()) continue;
      const SE3 t_t_r0 =
          target_frame.T_w_agent_linearization_point.inverse() * reference_frame.T_w_agent_linearization_point;
      const ArrayReprojector<true> reprojector(reference_frame.model, target_frame.model, t_t_r0);
      const double brightness_change_scale = (target_frame.exposure_time / reference_frame.exposure_time) *
                                             std::exp(target_frame.affine_brightness0[0] - reference_frame.affine_brightness0[0]);
      auto &residuals = it->second;
      auto &landmarks = reference_frame.active_landmarks;
      parallelFor(landmarks.size(), 64, [&](size_t begin, size_t end) {
        for (size_t li = begin; li < end; ++li) {
          Landmark &landmark = landmarks[li];
          if (landmark.is_marginalized && !landmark.to_marginalize) continue;
          ResidualPoint &residual = residuals[li];
          double tu[kPatternSize], tv[kPatternSize];
          residual.reprojection_jacobians_valid = reprojector.reprojectPattern<kPatternSize>(
              landmark.ref_u, landmark.ref_v, landmark.idepth, tu, tv, residual.d_u_idepth, residual.d_v_idepth,
              residual.d_u_tReferenceTarget, residual.d_v_tReferenceTarget);
          for (int k = 0; k < kPatternSize; ++k)
            landmark.corrected_intensities[k] =
                brightness_change_scale * (landmark.patch[k] - reference_frame.affine_brightness0[1]);
          residual.brightness_change_scale = brightness_change_scale;
        }
      });
    }
  }
}

/**
 * evaluateJacobians<double, SE3, Pinhole, 8, PixelMap, 1, FEJ, OPT_IDEPTHS, EVAL_J, NEW_PT, HUBER>
 * — PBA_INT/evaluate_jacobians.hpp:20-202
 */
template <bool FIRST_ESTIMATE_JACOBIANS, bool OPTIMIZE_IDEPTHS, bool EVALUATE_JACOBIANS, bool NEW_EVALUATION_POINT,
          bool APPLY_HUBER_LOSS>
void evaluateJacobians(Frames &frames, const double sigma_huber_loss = 0) {
  const double kSigmaHuberSqr = sigma_huber_loss * sigma_huber_loss;
  for (auto &reference_frame_ptr : frames) {
    LocalFrame &reference_frame = *reference_frame_ptr;
    for (auto &target_frame_ptr : frames) {
      LocalFrame &target_frame = *target_frame_ptr;
      if (reference_frame.id == target_frame.id) continue;
      auto it = reference_frame.residuals.find(target_frame.id);
      if (it == reference_frame.residuals.end()) continue;

      double t_log_w_r_eps[6], minus_t_log_w_t_eps[6];
      for (int i = 0; i < 6; ++i) {
        t_log_w_r_eps[i] = reference_frame.state_eps[i] + reference_frame.state_eps_step[i];
        minus_t_log_w_t_eps[i] = -(target_frame.state_eps[i] + target_frame.state_eps_step[i]);
      }
      const double reference_affine_brightness[2] = {
          reference_frame.affine_brightness0[0] + reference_frame.state_eps[6] + reference_frame.state_eps_step[6],
          reference_frame.affine_brightness0[1] + reference_frame.state_eps[7] + reference_frame.state_eps_step[7]};
      const double target_affine_brightness[2] = {
          target_frame.affine_brightness0[0] + target_frame.state_eps[6] + target_frame.state_eps_step[6],
          target_frame.affine_brightness0[1] + target_frame.state_eps[7] + target_frame.state_eps_step[7]};

      const SE3 t_t_r0 =
          target_frame.T_w_agent_linearization_point.inverse() * reference_frame.T_w_agent_linearization_point;
      const SE3 t_t_r = t_t_r0.rightIncrement(t_log_w_r_eps).leftIncrement(minus_t_log_w_t_eps);
      const double brightness_change_scale = (target_frame.exposure_time / reference_frame.exposure_time) *
                                             std::exp(target_affine_brightness[0] - reference_affine_brightness[0]);
      const ArrayReprojector<true> reprojector(reference_frame.model, target_frame.model, t_t_r);
      double rightLogTransformer[36];
      (FIRST_ESTIMATE_JACOBIANS ? t_t_r0 : t_t_r).Adj(rightLogTransformer);
      // leftLogTransformer is the identity for SE3 (se3_motion.hpp:252)

      const MaskView &target_mask = target_frame.mask;
      const auto &landmarks = reference_frame.active_landmarks;
      auto &residuals = it->second;
      const PixelMapView &grid = target_frame.grid;

      parallelFor(landmarks.size(), 32, [&](size_t begin, size_t end) {
        double d_reference_affineBrightnessShift = 0;
        for (size_t landmark_i = begin; landmark_i < end; ++landmark_i) {
          const Landmark &landmark = landmarks[landmark_i];
          if (landmark.is_marginalized && !landmark.to_marginalize) continue;
          ResidualPoint &residual = residuals[landmark_i];
          double corrected_reference_intensities[kPatternSize];
          residual.was_estimated = true;
          double tu[kPatternSize], tv[kPatternSize];
          bool success = false;
          if (FIRST_ESTIMATE_JACOBIANS || !EVALUATE_JACOBIANS) {
            success = reprojector.reprojectPattern<kPatternSize>(landmark.ref_u, landmark.ref_v,
                                                                 landmark.idepth + landmark.idepth_step, tu, tv);
            success = success && (!FIRST_ESTIMATE_JACOBIANS || residual.reprojection_jacobians_valid);
            d_reference_affineBrightnessShift = residual.brightness_change_scale;
            for (int k = 0; k < kPatternSize; ++k) corrected_reference_intensities[k] = landmark.corrected_intensities[k];
          } else {
            success = residual.reprojection_jacobians_valid = reprojector.reprojectPattern<kPatternSize>(
                landmark.ref_u, landmark.ref_v, landmark.idepth + landmark.idepth_step, tu, tv, residual.d_u_idepth,
                residual.d_v_idepth, residual.d_u_tReferenceTarget, residual.d_v_tReferenceTarget);
            for (int k = 0; k < kPatternSize; ++k)
              corrected_reference_intensities[k] =
                  brightness_change_scale * (landmark.patch[k] - reference_affine_brightness[1]);
            d_reference_affineBrightnessShift = brightness_change_scale;
          }
          success = success && target_mask.valid(tu, tv, kPatternSize);
          if (!success) residual.connection_status_candidate = kOOB;
          if (success && residual.connection_status == kOk) {
            residual.connection_status_candidate = kOk;
            double target_patch[kPatternSize], d_intensity_u_diag[kPatternSize], d_intensity_v_diag[kPatternSize];
            if (EVALUATE_JACOBIANS) {
              for (int k = 0; k < kPatternSize; ++k) {
                double v3[3];
                interpolateLinear3(grid, tu[k], tv[k], v3);
                target_patch[k] = v3[0];
                d_intensity_u_diag[k] = v3[1];
                d_intensity_v_diag[k] = v3[2];
              }
            } else {
              for (int k = 0; k < kPatternSize; ++k) target_patch[k] = interpolateLinear1(grid, tu[k], tv[k]);
            }
            if (NEW_EVALUATION_POINT) {
              double residuals_squared_norm = 0;
              for (int k = 0; k < kPatternSize; ++k) {
                const double residuals_left = target_patch[k] - target_affine_brightness[1];
                const double residuals_right =
                    brightness_change_scale * (landmark.patch[k] - reference_affine_brightness[1]);
                residual.residuals[k] = residuals_left - residuals_right;
                residuals_squared_norm += residual.residuals[k] * residual.residuals[k];
              }
              residual.energy = residuals_squared_norm * 0.5;
              residual.huber_weight = 1;
              if (APPLY_HUBER_LOSS) {
                if (residuals_squared_norm > kSigmaHuberSqr) {
                  const double residuals_norm = std::sqrt(residuals_squared_norm);
                  residual.huber_weight = sigma_huber_loss / residuals_norm;
                  residual.energy = sigma_huber_loss * residuals_norm - kSigmaHuberSqr * 0.5;
                }
              }
            }
            if (EVALUATE_JACOBIANS) {
              double d_target_reference_state[kPatternSize * kDoF];
              for (int i = 0; i < kPatternSize; ++i)
                for (int c = 0; c < kDoF; ++c)
                  d_target_reference_state[6 * i + c] =
                      d_intensity_v_diag[i] * residual.d_v_tReferenceTarget[6 * i + c] +
                      d_intensity_u_diag[i] * residual.d_u_tReferenceTarget[6 * i + c];
              for (int i = 0; i < kPatternSize; ++i) {
                for (int c = 0; c < kDoF; ++c) {
                  residual.d_target_state_eps[8 * i + c] = -d_target_reference_state[6 * i + c];
                  double s = 0;
                  for (int k = 0; k < kDoF; ++k) s += d_target_reference_state[6 * i + k] * rightLogTransformer[6 * k + c];
                  residual.d_reference_state_eps[8 * i + c] = s;
                }
              }
              if (OPTIMIZE_IDEPTHS) {
                for (int i = 0; i < kPatternSize; ++i)
                  residual.d_idepth[i] =
                      d_intensity_u_diag[i] * residual.d_u_idepth[i] + d_intensity_v_diag[i] * residual.d_v_idepth[i];
              }
              for (int i = 0; i < kPatternSize; ++i) {
                residual.d_reference_state_eps[8 * i + 6] = corrected_reference_intensities[i];
                residual.d_reference_state_eps[8 * i + 7] = d_reference_affineBrightnessShift;
                residual.d_target_state_eps[8 * i + 6] = -corrected_reference_intensities[i];
                residual.d_target_state_eps[8 * i + 7] = -1;
              }
            }
          } else {
            if (NEW_EVALUATION_POINT) {
              for (int k = 0; k < kPatternSize; ++k) residual.residuals[k] = 0;
              residual.energy = 0;
            }
            if (EVALUATE_JACOBIANS) {
              for (double &v : residual.d_reference_state_eps) v = 0;
              for (double &v : residual.d_target_state_eps) v = 0;
              for (double &v : residual.d_idepth) v = 0;
            }
          }
        }
      });
    }
  }
}

/** evaluateLinearSystemPosePoseBlock — PBA_INT/hessian_block_evaluation.hpp:38-90 */
template <bool FOR_MARGINALIZED>
bool evaluateLinearSystemPosePoseBlock(const LocalFrame &reference_frame, const LocalFrame &target_frame, double *H_rr,
                                       double *H_rt, double *H_tt, double *b_r, double *b_t) {
  auto it = reference_frame.residuals.find(target_frame.id);
  if (it == reference_frame.residuals.end()) return false;  // outputs left untouched, as in the reference (:55-58)
  KahanFixed<64> acc_rr, acc_rt, acc_tt;
  KahanFixed<8> acc_br, acc_bt;
  const auto &residuals = it->second;
  const auto &landmarks = reference_frame.active_landmarks;
  double s_rr[64], s_rt[64], s_tt[64], s_br[8], s_bt[8];
  for (size_t li = 0; li < landmarks.size(); ++li) {
    const Landmark &landmark = landmarks[li];
    if (FOR_MARGINALIZED) {
      if (!landmark.to_marginalize) continue;
    } else {
      if (landmark.is_marginalized) continue;
    }
    const ResidualPoint &residual = residuals[li];
    const double w = residual.huber_weight;
    const double *Jr = residual.d_reference_state_eps, *Jt = residual.d_target_state_eps;
    for (int a = 0; a < 8; ++a) {
      double br = 0, bt = 0;
      for (int k = 0; k < 8; ++k) {
        br += Jr[8 * k + a] * residual.residuals[k];
        bt += Jt[8 * k + a] * residual.residuals[k];
      }
      s_br[a] = w * br;
      s_bt[a] = w * bt;
      for (int b = 0; b < 8; ++b) {
        double rr = 0, rt = 0, tt = 0;
        for (int k = 0; k < 8; ++k) {
          rr += Jr[8 * k + a] * Jr[8 * k + b];
          rt += Jr[8 * k + a] * Jt[8 * k + b];
          tt += Jt[8 * k + a] * Jt[8 * k + b];
        }
        s_rr[8 * a + b] = w * rr;
        s_rt[8 * a + b] = w * rt;
        s_tt[8 * a + b] = w * tt;
      }
    }
    acc_rr.add(s_rr);
    acc_tt.add(s_tt);
    acc_rt.add(s_rt);
    acc_br.add(s_br);
    acc_bt.add(s_bt);
  }
  for (int i = 0; i < 64; ++i) {
    H_rr[i] = acc_rr.sum[i];
    H_rt[i] = acc_rt.sum[i];
    H_tt[i] = acc_tt.sum[i];
  }
  for (int i = 0; i < 8; ++i) {
    b_r[i] = acc_br.sum[i];
    b_t[i] = acc_bt.sum[i];
  }
  return true;
}

/** evaluateLinearSystemPosePose — PBA_INT/hessian_block_evaluation.hpp:96-164 */
template <bool FOR_MARGINALIZED>
void evaluateLinearSystemPosePose(const Frames &frames, NormalLinearSystem &system_pose) {
  const int F = static_cast<int>(frames.size());
  std::mutex mutex;
  parallelFor(static_cast<size_t>(F), 1, [&](size_t begin, size_t end) {
    double H_rr[64] = {0}, H_rt[64] = {0}, H_tt[64] = {0}, b_r[8] = {0}, b_t[8] = {0};
    for (size_t r = begin; r < end; ++r) {
      const LocalFrame &reference_frame = *frames[r];
      for (int t = 0; t < F; ++t) {
        if (static_cast<int>(r) == t) continue;
        const LocalFrame &target_frame = *frames[static_cast<size_t>(t)];
        evaluateLinearSystemPosePoseBlock<FOR_MARGINALIZED>(reference_frame, target_frame, H_rr, H_rt, H_tt, b_r, b_t);
        std::lock_guard<std::mutex> guard(mutex);
        const int rb = 8 * static_cast<int>(r), tb = 8 * t;
        for (int a = 0; a < 8; ++a) {
          for (int b = 0; b < 8; ++b) {
            system_pose.H(rb + a, rb + b) += H_rr[8 * a + b];
            system_pose.H(rb + a, tb + b) = H_rt[8 * a + b];
            system_pose.H(tb + a, tb + b) += H_tt[8 * a + b];
          }
          system_pose.b[static_cast<size_t>(rb + a)] += b_r[a];
          system_pose.b[static_cast<size_t>(tb + a)] += b_t[a];
        }
      }
    }
  });
  // symmetrisation pass (:147-163)
  for (int r = 0; r < F; ++r) {
    const int rb = 8 * r;
    for (int a = 0; a < 8; ++a)
      for (int b = a + 1; b < 8; ++b) system_pose.H(rb + a, rb + b) = system_pose.H(rb + b, rb + a);
    for (int t = r + 1; t < F; ++t) {
      const int tb = 8 * t;
      for (int a = 0; a < 8; ++a)
        for (int b = 0; b < 8; ++b) system_pose.H(rb + a, tb + b) += system_pose.H(tb + b, rb + a);
      for (int a = 0; a < 8; ++a)
        for (int b = 0; b < 8; ++b) system_pose.H(tb + a, rb + b) = system_pose.H(rb + b, tb + a);
    }
  }
}

/** evaluateLinearSystemPoseDepthSchurComplement — PBA_INT/hessian_block_evaluation.hpp:169-236 */
template <bool FOR_MARGINALIZED>
void evaluateLinearSystemPoseDepthSchurComplement(Frames &frames, NormalLinearSystem &system_schur) {
  const int F = static_cast<int>(frames.size());
  const int K = 8 * F;
  const double kScaleNullspaceRegularizer = 1e8;
  system_schur.setZero();
  std::mutex mutex;
  parallelFor(static_cast<size_t>(F), 1, [&](size_t begin, size_t end) {
    KahanDynamic hessian(static_cast<size_t>(K) * K);
    KahanDynamic b(static_cast<size_t>(K));
    std::vector<double> hpib(static_cast<size_t>(K));
    for (size_t r = begin; r < end; ++r) {
      LocalFrame &reference_frame = *frames[r];
      auto &landmarks = reference_frame.active_landmarks;
      for (size_t li = 0; li < landmarks.size(); ++li) {
        Landmark &landmark = landmarks[li];
        if (FOR_MARGINALIZED) {
          if (!landmark.to_marginalize) continue;
        } else {
          if (landmark.is_marginalized) continue;
        }
        std::fill(hpib.begin(), hpib.end(), 0.0);
        double b_idepth_block = 0, hessian_idepth_idepth = 0;
        for (int t = 0; t < F; ++t) {
          if (static_cast<int>(r) == t) continue;
          const LocalFrame &target_frame = *frames[static_cast<size_t>(t)];
          const ResidualPoint &residual = reference_frame.residuals.at(target_frame.id)[li];
          const double w = residual.huber_weight;
          double dd = 0, bd = 0;
          for (int a = 0; a < 8; ++a) {
            double sr = 0, st = 0;
            for (int k = 0; k < 8; ++k) {
              sr += residual.d_reference_state_eps[8 * k + a] * residual.d_idepth[k];
              st += residual.d_target_state_eps[8 * k + a] * residual.d_idepth[k];
            }
            hpib[static_cast<size_t>(8 * static_cast<int>(r) + a)] += w * sr;
            hpib[static_cast<size_t>(8 * t + a)] += w * st;
          }
          for (int k = 0; k < 8; ++k) {
            dd += residual.d_idepth[k] * residual.d_idepth[k];
            bd += residual.d_idepth[k] * residual.residuals[k];
          }
          hessian_idepth_idepth += w * dd;
          b_idepth_block += w * bd;
        }
        landmark.b_idepth_block = b_idepth_block;
        landmark.hessian_poses_idepth_block = hpib;
        const double kIdepthNullSpaceThreshold = 1e-15;
        if (hessian_idepth_idepth > kIdepthNullSpaceThreshold) {
          if (FOR_MARGINALIZED && reference_frame.fixed) hessian_idepth_idepth += kScaleNullspaceRegularizer;
          landmark.inv_hessian_idepth_idepth = 1.0 / hessian_idepth_idepth;
          landmark.ill_conditioned = false;
          const double inv = landmark.inv_hessian_idepth_idepth;
          const double ib = inv * b_idepth_block;
          b.add([&](size_t i) { return ib * hpib[i]; });
          hessian.add([&](size_t idx) {
            const size_t i = idx / static_cast<size_t>(K), j = idx % static_cast<size_t>(K);
            return inv * (hpib[i] * hpib[j]);
          });
        } else {
          landmark.ill_conditioned = true;
        }
      }
    }
    std::lock_guard<std::mutex> guard(mutex);
    for (size_t i = 0; i < static_cast<size_t>(K) * K; ++i) system_schur.H.a[i] += hessian.sum[i];
    for (size_t i = 0; i < static_cast<size_t>(K); ++i) system_schur.b[i] += b.sum[i];
  });
}

/** calculateIdepths — PBA_INT/hessian_block_evaluation.hpp:238-263 */
inline void calculateIdepths(Frames &frames, const Vec &step_poses, double levenberg_marquardt_lambda) {
  const double kLevenbergMarquardtLambdaInversed = 1.0 / (1.0 + levenberg_marquardt_lambda);
  for (auto &frame : frames) {
    auto &landmarks = frame->active_landmarks;
    parallelFor(landmarks.size(), 128, [&](size_t begin, size_t end) {
      for (size_t li = begin; li < end; ++li) {
        Landmark &landmark = landmarks[li];
        if (landmark.is_marginalized) continue;
        if (!landmark.ill_conditioned) {
          double d = 0;
          for (size_t i = 0; i < step_poses.size(); ++i) d += landmark.hessian_poses_idepth_block[i] * step_poses[i];
          const double step = (landmark.b_idepth_block - d) * kLevenbergMarquardtLambdaInversed *
                              landmark.inv_hessian_idepth_idepth;
          landmark.idepth_step = -step;
        }
      }
    });
  }
}

/** changeResidualStatuses — PBA_INT/eigen_photometric_bundle_adjustment_problem.hpp:20-35 */
inline void changeResidualStatuses(Frames &frames, bool accept = true) {
  for (auto &f : frames)
    for (auto &kv : f->residuals)
      for (auto &residual : kv.second) {
        if (accept)
          residual.connection_status = residual.connection_status_candidate;
        else
          residual.connection_status_candidate = residual.connection_status;
      }
}

/** evaluateLinearSystemPrior — problem.hpp:37-77 (MotionPrior<SE3> is zero, state_priors.hpp:30-59) */
inline void evaluateLinearSystemPrior(Frames &frames, NormalLinearSystem &system_prior,
                                      const double affine_brightness_regularizer[2], double fixed_state_regularizer,
                                      bool for_marginalized = false) {
  for (size_t fi = 0; fi < frames.size(); ++fi) {
    LocalFrame &frame = *frames[fi];
    if (frame.to_marginalize != for_marginalized) continue;
    const int fb = 8 * static_cast<int>(fi);
    if (frame.fixed) {
      for (int a = 0; a < 8; ++a) {
        system_prior.H(fb + a, fb + a) += fixed_state_regularizer;
        system_prior.b[static_cast<size_t>(fb + a)] += fixed_state_regularizer * frame.state_eps[a];
      }
    } else {
      // AffineBrightnessPrior::priorSystem — state_priors.hpp:100-107
      for (int a = 0; a < 2; ++a) {
        const double ab = frame.affine_brightness0[a] + frame.state_eps[6 + a];
        system_prior.H(fb + 6 + a, fb + 6 + a) += affine_brightness_regularizer[a];
        system_prior.b[static_cast<size_t>(fb + 6 + a)] += affine_brightness_regularizer[a] * ab;
      }
    }
  }
}

/** stateEpsStacked — problem.hpp:79-92 */
inline Vec stateEpsStacked(const Frames &frames, bool with_step = false) {
  Vec state(frames.size() * 8);
  for (size_t i = 0; i < frames.size(); ++i)
    for (int a = 0; a < 8; ++a)
      state[8 * i + static_cast<size_t>(a)] = frames[i]->state_eps[a] + (with_step ? frames[i]->state_eps_step[a] : 0.0);
  return state;
}

/** calculateLandmarksEnergy — problem.hpp:93-144 */
template <bool FOR_MARGINALIZED>
std::pair<double, int> calculateLandmarksEnergy(const Frames &frames) {
  double energy = 0;
  int number_of_valid_residuals = 0;
  for (auto &reference_frame : frames) {
    for (auto &target_frame : frames) {
      if (reference_frame->id == target_frame->id) continue;
      auto it = reference_frame->residuals.find(target_frame->id);
      if (it == reference_frame->residuals.end()) continue;
      const auto &residuals = it->second;
      const auto &landmarks = reference_frame->active_landmarks;
      for (size_t li = 0; li < residuals.size(); ++li) {
        const Landmark &landmark = landmarks[li];
        if (FOR_MARGINALIZED) {
          if (!landmark.to_marginalize) continue;
        } else {
          if (landmark.is_marginalized) continue;
        }
        energy += residuals[li].energy;
        if (residuals[li].energy > 0) number_of_valid_residuals++;
      }
    }
  }
  return {energy, number_of_valid_residuals};
}

/** TrustRegionPhotometricBundleAdjustmentOptions + EigenPBA ctor flags —
 *  trust_region_photometric_bundle_adjustment_options.hpp:14-52; production values tracker/src/fabric.cpp:63-79 */
struct PbaOptions {
  int max_iterations = 7;
  double initial_trust_region_radius = 1e5;
  double function_tolerance = 1e-8;
  double parameter_tolerance = 1e-8;
  double affine_brightness_regularizer[2] = {1e12, 1e8};
  double fixed_state_regularizer = 1e16;
  double sigma_huber_loss = 20;
  bool estimate_uncertainty = true;
  bool force_accept = true;
  bool first_estimate_jacobians = true;
  bool optimize_idepths = true;
};

/** levenberg_marquardt_algorithm::Options/Result — levenberg_marquardt_algorithm.hpp:38-68 */
struct LmOptions {
  size_t max_num_iterations = 50;
  double initial_levenberg_marquardt_regularizer = 1e-5;
  double function_tolerance = 1e-8;
  double parameter_tolerance = 1e-8;
  bool force_accept = false;
  size_t min_num_iterations = 0;
  double levenberg_marquardt_regularizer_decrease_on_accept = 2;
  double levenberg_marquardt_regularizer_increase_on_reject = 10;
};
struct LmResult {
  double energy = std::numeric_limits<double>::max();
  int number_of_valid_residuals = 0;
  bool converged = false;
  int iterations = 0;  // oracle-only bookkeeping: number of loop bodies executed
};

/** levenberg_marquardt_algorithm::solve — levenberg_marquardt_algorithm.hpp:77-128 */
template <typename Problem>
LmResult lmSolve(Problem &problem, const LmOptions &options) {
  LmResult result;
  double lambda = options.initial_levenberg_marquardt_regularizer;
  std::tie(result.energy, result.number_of_valid_residuals) = problem.calculateEnergy();
  bool linear_system_valid = false;
  for (size_t iteration = 0;
       iteration < options.max_num_iterations && !result.converged && result.number_of_valid_residuals > 0; ++iteration) {
    result.iterations++;
    if (!linear_system_valid) problem.linearize();
    problem.calculateStep(lambda);
    auto [next_energy, number_of_valid_residuals] = problem.calculateEnergy();
    if (number_of_valid_residuals == 0) {
      problem.rejectStep();
      break;
    }
    const bool function_tolerance_reached =
        std::abs(result.energy - next_energy) / result.energy < options.function_tolerance;
    result.converged |= function_tolerance_reached;
    if (next_energy < result.energy || (options.force_accept && iteration < options.min_num_iterations)) {
      auto [state_squared_norm, step_squared_norm] = problem.acceptStep();
      const bool parameter_tolerance_reached =
          step_squared_norm < options.parameter_tolerance * (state_squared_norm + options.parameter_tolerance);
      result.converged |= parameter_tolerance_reached;
      result.energy = next_energy;
      result.number_of_valid_residuals = number_of_valid_residuals;
      lambda /= options.levenberg_marquardt_regularizer_decrease_on_accept;
      linear_system_valid = false;
    } else {
      problem.rejectStep();
      if (options.force_accept) {
        problem.calculateEnergy();
        return result;
      }
      lambda *= options.levenberg_marquardt_regularizer_increase_on_reject;
      linear_system_valid = true;
    }
  }
  problem.calculateEnergy();
  return result;
}

/** PhotometricBundleAdjustmentProblem — problem.hpp:255-429, runtime-flag version of the template */
struct PbaProblem {
  Frames &frames;
  const PbaOptions &opt;
  NormalLinearSystem system_marginalized;
  double energy_marginalized;
  NormalLinearSystem system_pose, system_schur;
  Vec last_step;

  PbaProblem(Frames &f, const PbaOptions &o, const NormalLinearSystem &marg, double e_marg)
      : frames(f), opt(o), system_marginalized(marg), energy_marginalized(e_marg),
        system_pose(8 * static_cast<int>(f.size())), system_schur(8 * static_cast<int>(f.size())) {}

  void evaluate(bool evaluate_jacobians) {
    const double s = opt.sigma_huber_loss;
    if (opt.first_estimate_jacobians) {
      if (evaluate_jacobians)
        evaluateJacobians<true, true, true, true, true>(frames, s);
      else
        evaluateJacobians<true, true, false, true, true>(frames, s);
    } else {
      if (evaluate_jacobians) {
        if (opt.optimize_idepths)
          evaluateJacobians<false, true, true, true, true>(frames, s);
        else
          evaluateJacobians<false, false, true, true, true>(frames, s);
      } else
        evaluateJacobians<false, true, false, true, true>(frames, s);
    }
  }

  /** calculateEnergy — problem.hpp:290-317 */
  std::pair<double, int> calculateEnergy() {
    evaluate(false);
    double energy = energy_marginalized;
    Vec state = stateEpsStacked(frames, true);
    energy += dot(system_marginalized.b, state) + dot(state, matvec(system_marginalized.H, state)) / 2;
    for (auto &frame : frames) {
      // AffineBrightnessPrior::energyTerm — state_priors.hpp:87-90
      double e = 0;
      for (int a = 0; a < 2; ++a) {
        const double ab = frame->affine_brightness0[a] + frame->state_eps[6 + a] + frame->state_eps_step[6 + a];
        e += ab * opt.affine_brightness_regularizer[a] * ab;
      }
      energy += e / 2;
    }
    auto le = calculateLandmarksEnergy<false>(frames);
    return {energy + le.first, le.second};
  }
  /** linearize — problem.hpp:322-336 */
  void linearize() {
    evaluate(true);
    system_pose.setZero();
    system_schur.setZero();
    evaluateLinearSystemPosePose<false>(frames, system_pose);
    evaluateLinearSystemPrior(frames, system_pose, opt.affine_brightness_regularizer, opt.fixed_state_regularizer);
    if (opt.optimize_idepths) evaluateLinearSystemPoseDepthSchurComplement<false>(frames, system_schur);
  }
  /** calculateStep — problem.hpp:342-361 */
  void calculateStep(double lambda) {
    const int K = system_pose.size();
    Vec state = stateEpsStacked(frames);
    NormalLinearSystem full(K);
    const double sc = -1.0 / (1.0 + lambda);
    for (int i = 0; i < K; ++i) {
      for (int j = 0; j < K; ++j) full.H(i, j) = system_pose.H(i, j) + system_marginalized.H(i, j);
      full.b[static_cast<size_t>(i)] = system_pose.b[static_cast<size_t>(i)] + system_marginalized.b[static_cast<size_t>(i)];
    }
    for (int i = 0; i < K; ++i) full.H(i, i) += system_pose.H(i, i) * lambda;
    for (int i = 0; i < K; ++i) {
      for (int j = 0; j < K; ++j) full.H(i, j) += system_schur.H(i, j) * sc;
      full.b[static_cast<size_t>(i)] += system_schur.b[static_cast<size_t>(i)] * sc;
    }
    Vec Hm_state = matvec(system_marginalized.H, state);
    for (int i = 0; i < K; ++i) full.b[static_cast<size_t>(i)] += Hm_state[static_cast<size_t>(i)];
    Vec step = full.solve();
    last_step = step;
    for (size_t fi = 0; fi < frames.size(); ++fi)
      for (int a = 0; a < 8; ++a) frames[fi]->state_eps_step[a] = -step[8 * fi + static_cast<size_t>(a)];
    if (opt.optimize_idepths) calculateIdepths(frames, step, lambda);
  }
  /** acceptStep — problem.hpp:366-388 */
  std::pair<double, double> acceptStep() {
    double state_squared_norm = 0, step_squared_norm = 0;
    for (auto &frame : frames) {
      for (int a = 0; a < 8; ++a) state_squared_norm += frame->state_eps[a] * frame->state_eps[a];
      state_squared_norm += frame->affine_brightness0[0] * frame->affine_brightness0[0] +
                            frame->affine_brightness0[1] * frame->affine_brightness0[1];
      for (int a = 0; a < 8; ++a) {
        frame->state_eps[a] += frame->state_eps_step[a];
        step_squared_norm += frame->state_eps_step[a] * frame->state_eps_step[a];
        frame->state_eps_step[a] = 0;
      }
      if (opt.optimize_idepths) {
        for (auto &landmark : frame->active_landmarks) {
          state_squared_norm += landmark.idepth * landmark.idepth;
          landmark.idepth += landmark.idepth_step;
          step_squared_norm += landmark.idepth_step * landmark.idepth_step;
          landmark.idepth_step = 0;
        }
      }
    }
    changeResidualStatuses(frames);
    return {state_squared_norm, step_squared_norm};
  }
  /** rejectStep — problem.hpp:392-402 */
  void rejectStep() {
    for (auto &frame : frames) {
      for (int a = 0; a < 8; ++a) frame->state_eps_step[a] = 0;
      if (opt.optimize_idepths)
        for (auto &landmark : frame->active_landmarks) landmark.idepth_step = 0;
    }
    changeResidualStatuses(frames, false);
  }
};

/** updateMarginalizedLinearSystem — problem.hpp:146-203 */
inline void updateMarginalizedLinearSystem(Frames &frames, NormalLinearSystem &system_marginalized,
                                           double &energy_marginalized, const double affine_brightness_regularizer[2],
                                           double fixed_state_regularizer) {
  const int K = 8 * static_cast<int>(frames.size());
  NormalLinearSystem schur(K), pose(K);
  evaluateLinearSystemPoseDepthSchurComplement<true>(frames, schur);
  evaluateLinearSystemPosePose<true>(frames, pose);
  NormalLinearSystem pts(K);
  for (size_t i = 0; i < pts.H.a.size(); ++i) pts.H.a[i] = pose.H.a[i] - schur.H.a[i];
  for (int i = 0; i < K; ++i) pts.b[static_cast<size_t>(i)] = pose.b[static_cast<size_t>(i)] - schur.b[static_cast<size_t>(i)];
  Vec state = stateEpsStacked(frames);
  Vec Hs = matvec(pts.H, state);
  energy_marginalized += calculateLandmarksEnergy<true>(frames).first + dot(state, Hs) - dot(state, pts.b);
  for (int i = 0; i < K; ++i) pts.b[static_cast<size_t>(i)] -= Hs[static_cast<size_t>(i)];
  for (size_t i = 0; i < pts.H.a.size(); ++i) system_marginalized.H.a[i] += pts.H.a[i];
  for (int i = 0; i < K; ++i) system_marginalized.b[static_cast<size_t>(i)] += pts.b[static_cast<size_t>(i)];
  for (auto &frame : frames)
    for (auto &landmark : frame->active_landmarks) landmark.to_marginalize = false;
  std::vector<int> marginalized_part;
  for (size_t i = 0; i < frames.size(); ++i)
    if (frames[i]->to_marginalize)
      for (int p = 0; p < 8; ++p) marginalized_part.push_back(static_cast<int>(i) * 8 + p);
  if (marginalized_part.empty()) return;
  NormalLinearSystem prior(K);
  evaluateLinearSystemPrior(frames, prior, affine_brightness_regularizer, fixed_state_regularizer, true);
  Vec Hps = matvec(prior.H, state);
  for (int i = 0; i < K; ++i) prior.b[static_cast<size_t>(i)] -= Hps[static_cast<size_t>(i)];
  for (size_t i = 0; i < prior.H.a.size(); ++i) system_marginalized.H.a[i] += prior.H.a[i];
  for (int i = 0; i < K; ++i) system_marginalized.b[static_cast<size_t>(i)] += prior.b[static_cast<size_t>(i)];
  system_marginalized.reduceSystem(marginalized_part);
  frames.erase(std::remove_if(frames.begin(), frames.end(), [](auto &f) { return f->to_marginalize; }), frames.end());
}

/** covarianceMatrixPosePose — problem.hpp:204-242 */
inline Mat covarianceMatrixPosePose(Frames &frames, const PbaOptions &opt, const NormalLinearSystem &system_marginalized) {
  const int K = 8 * static_cast<int>(frames.size());
  if (opt.first_estimate_jacobians) {
    if (opt.optimize_idepths)
      evaluateJacobians<true, true, true, true, false>(frames);
    else
      evaluateJacobians<true, false, true, true, false>(frames);
  } else {
    if (opt.optimize_idepths)
      evaluateJacobians<false, true, true, true, false>(frames);
    else
      evaluateJacobians<false, false, true, true, false>(frames);
  }
  NormalLinearSystem pose(K), schur(K);
  evaluateLinearSystemPosePose<false>(frames, pose);
  evaluateLinearSystemPrior(frames, pose, opt.affine_brightness_regularizer, opt.fixed_state_regularizer);
  if (opt.optimize_idepths) evaluateLinearSystemPoseDepthSchurComplement<false>(frames, schur);
  Mat full(K, K);
  for (size_t i = 0; i < full.a.size(); ++i) full.a[i] = pose.H.a[i] - schur.H.a[i] + system_marginalized.H.a[i];
  return pseudoInverseDropSmallest(full, opt.optimize_idepths ? 1 : 0);
}

/** relativeTransformationUncertainty — se3_motion.hpp:151-158 */
inline void relativeTransformationUncertainty(const SE3 &t_w_1, const SE3 &t_w_2, const double *s11, const double *s22,
                                              const double *s12, double *out) {
  double adj[36];
  (t_w_2.inverse() * t_w_1).Adj(adj);
  // adj*s11*adj^T - s12^T*adj^T - adj*s12 + s22
  double a11[36], a12[36];
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      double x = 0, y = 0;
      for (int k = 0; k < 6; ++k) {
        x += adj[6 * i + k] * s11[6 * k + j];
        y += adj[6 * i + k] * s12[6 * k + j];
      }
      a11[6 * i + j] = x;
      a12[6 * i + j] = y;
    }
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) {
      double x = 0, y = 0;
      for (int k = 0; k < 6; ++k) {
        x += a11[6 * i + k] * adj[6 * j + k];
        y += s12[6 * k + i] * adj[6 * j + k];
      }
      out[6 * i + j] = x - y - a12[6 * i + j] + s22[6 * i + j];
    }
}

/** covarianceMatricesOfRelativePoses — PBA_INT/covariance_matrices_of_relative_poses.hpp:23-62 */
inline void covarianceMatricesOfRelativePoses(Frames &frames, const Mat &cov) {
  for (size_t r = 0; r < frames.size(); ++r)
    for (size_t t = 0; t < frames.size(); ++t) {
      if (frames[r]->id == frames[t]->id) continue;
      double s_rr[36], s_tt[36], s_rt[36];
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
          s_rr[6 * i + j] = cov(static_cast<int>(8 * r) + i, static_cast<int>(8 * r) + j);
          s_tt[6 * i + j] = cov(static_cast<int>(8 * t) + i, static_cast<int>(8 * t) + j);
          s_rt[6 * i + j] = cov(static_cast<int>(8 * r) + i, static_cast<int>(8 * t) + j);
        }
      std::array<double, 36> out;
      relativeTransformationUncertainty(frames[r]->tWorldAgent(), frames[t]->tWorldAgent(), s_rr, s_tt, s_rt, out.data());
      frames[r]->covariance_matrices[frames[t]->id] = out;
    }
}

/** updatePointStatuses — PROB_SRC/photometric_bundle_adjustment.cpp:321-406 */
inline void updatePointStatuses(Frames &frames, size_t minimum_valid_reprojections_num, double sigma_huber_loss) {
  std::vector<double> energies;
  energies.reserve(12000);
  for (auto &reference_frame : frames) {
    auto &landmarks = reference_frame->active_landmarks;
    for (size_t li = 0; li < landmarks.size(); ++li) {
      if (landmarks[li].is_marginalized) continue;
      for (auto &target_frame : frames) {
        if (target_frame->is_marginalized || reference_frame->id == target_frame->id) continue;
        auto it = reference_frame->residuals.find(target_frame->id);
        if (it == reference_frame->residuals.end()) continue;
        if (li >= it->second.size()) continue;
        const ResidualPoint &residual = it->second[li];
        if (residual.connection_status == kOk) energies.push_back(residual.energy);
      }
    }
  }
  const size_t third_quartile = static_cast<size_t>(static_cast<double>(energies.size()) * 0.75);
  double energy_threshold = 0;
  if (!energies.empty()) {
    std::nth_element(energies.begin(), energies.begin() + static_cast<long>(third_quartile), energies.end());
    energy_threshold = energies.at(third_quartile) + sigma_huber_loss * sigma_huber_loss / 2;
  }
  for (auto &reference_frame : frames) {
    auto &landmarks = reference_frame->active_landmarks;
    for (size_t li = 0; li < landmarks.size(); ++li) {
      size_t valid_reprojections = 0;
      Landmark &landmark = landmarks[li];
      if (landmark.is_marginalized) continue;
      landmark.number_of_inlier_residuals = 0;
      for (auto &target_frame : frames) {
        if (target_frame->is_marginalized || reference_frame->id == target_frame->id) continue;
        auto it = reference_frame->residuals.find(target_frame->id);
        if (it == reference_frame->residuals.end()) continue;
        const SE3 Tr = reference_frame->tWorldAgent(), Tt = target_frame->tWorldAgent();
        const double dx = Tr.t[0] - Tt.t[0], dy = Tr.t[1] - Tt.t[1], dz = Tr.t[2] - Tt.t[2];
        const double distance = std::sqrt(dx * dx + dy * dy + dz * dz);
        if (li >= it->second.size()) continue;
        ResidualPoint &residual = it->second[li];
        if (residual.energy > energy_threshold) residual = ResidualPoint(kOutlier);
        if (residual.connection_status == kOk) {
          landmark.relative_baseline = std::max(landmark.relative_baseline, landmark.idepth * distance);
          valid_reprojections++;
          landmark.number_of_inlier_residuals++;
        }
      }
      if (valid_reprojections < minimum_valid_reprojections_num) landmark.is_outlier = true;
    }
  }
}

/**
 * EigenPhotometricBundleAdjustment<SE3, Pinhole, 8, PixelMap, true, OPT_IDEPTHS, FEJ, 1>
 * — PROB_SRC/eigen_photometric_bundle_adjustment.cpp:47-141 + base class PROB_SRC/photometric_bundle_adjustment.cpp
 */
struct PbaWindow {
  PbaOptions opt;
  Frames frames;
  NormalLinearSystem system_marginalized{0};
  double energy_marginalized = 0;
  LmResult last_result;
  // snapshot of the last linearised system (for stage-level parity checks)
  NormalLinearSystem last_system_pose{0}, last_system_schur{0};

  LocalFrame *getLocalFrame(int id) {
    for (auto &f : frames)
      if (f->id == id) return f.get();
    return nullptr;
  }
  int frameIndex(int id) const {
    for (size_t i = 0; i < frames.size(); ++i)
      if (frames[i]->id == id) return static_cast<int>(i);
    return -1;
  }

  /** pushFrame — eigen_photometric_bundle_adjustment.cpp:119-141 (+ photometric_bundle_adjustment.cpp:98-124;
   *  residual creation from FrameConnection statuses is done by setConnection) */
  int pushFrame(std::unique_ptr<LocalFrame> frame) {
    if (!frames.empty() && !(frames.back()->timestamp < frame->timestamp)) return -3;
    if (frames.size() > 1) {
      firstEstimateJacobians(frames);
      if (opt.first_estimate_jacobians)
        evaluateJacobians<true, true, true, true, true>(frames, opt.sigma_huber_loss);
      else
        evaluateJacobians<false, true, true, true, true>(frames, opt.sigma_huber_loss);
      changeResidualStatuses(frames);
      updateMarginalizedLinearSystem(frames, system_marginalized, energy_marginalized,
                                     opt.affine_brightness_regularizer, opt.fixed_state_regularizer);
    }
    frames.push_back(std::move(frame));
    system_marginalized.resizeKeep(8 * static_cast<int>(frames.size()));
    return 0;
  }

  /** the LM part of solve() (eigen_photometric_bundle_adjustment.cpp:67-86), without post-processing: used to time
   *  Gauss-Newton iterations on the CPU beside the GPU numbers */
  double optimize() {
    LmOptions options;
    options.initial_levenberg_marquardt_regularizer = 1.0 / opt.initial_trust_region_radius;
    options.function_tolerance = opt.function_tolerance;
    options.parameter_tolerance = opt.parameter_tolerance;
    options.max_num_iterations = static_cast<size_t>(opt.max_iterations);
    options.min_num_iterations = 3;
    options.force_accept = opt.force_accept;
    options.levenberg_marquardt_regularizer_decrease_on_accept = 1.;
    options.levenberg_marquardt_regularizer_increase_on_reject = 1.;
    PbaProblem problem(frames, opt, system_marginalized, energy_marginalized);
    if (opt.first_estimate_jacobians) firstEstimateJacobians(frames);
    last_result = lmSolve(problem, options);
    return last_result.energy;
  }

  /** solve — eigen_photometric_bundle_adjustment.cpp:61-101 */
  double solve() {
    LmOptions options;
    options.initial_levenberg_marquardt_regularizer = 1.0 / opt.initial_trust_region_radius;
    options.function_tolerance = opt.function_tolerance;
    options.parameter_tolerance = opt.parameter_tolerance;
    options.max_num_iterations = static_cast<size_t>(opt.max_iterations);
    options.min_num_iterations = 3;
    options.force_accept = opt.force_accept;
    options.levenberg_marquardt_regularizer_decrease_on_accept = 1.;
    options.levenberg_marquardt_regularizer_increase_on_reject = 1.;
    PbaProblem problem(frames, opt, system_marginalized, energy_marginalized);
    if (opt.first_estimate_jacobians) firstEstimateJacobians(frames);
    last_result = lmSolve(problem, options);
    last_system_pose = problem.system_pose;
    last_system_schur = problem.system_schur;
    relinearizeSystem();
    if (opt.estimate_uncertainty) {
      firstEstimateJacobians(frames);
      covarianceMatricesOfRelativePoses(frames, covarianceMatrixPosePose(frames, opt, system_marginalized));
    }
    updatePointStatuses(frames, 1, opt.sigma_huber_loss);
    return last_result.energy;
  }
  /** relinearizeSystem — photometric_bundle_adjustment.cpp:310-316 */
  void relinearizeSystem() {
    LocalFrame &last = *frames.back();
    last.T_w_agent_linearization_point = last.tWorldAgent();
    double ab[2];
    last.affineBrightness(ab);
    last.affine_brightness0[0] = ab[0];
    last.affine_brightness0[1] = ab[1];
    for (double &v : last.state_eps) v = 0;
  }
};

}  // namespace oracle
