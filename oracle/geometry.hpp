// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// Geometry / photometry primitives of the hot path, restated from the reference (each function cites the lines
// it follows).  Double precision throughout (reference default Precision = double,
// src/common/include/common/settings.hpp:10-14).
#pragma once
#include <cmath>
#include <cstdint>

#include "se3.hpp"

namespace oracle {

constexpr int kPatternSize = 8;  // Pattern::kSize — src/common/pattern/include/common/pattern/pattern.hpp:17
constexpr int kPatternCenter = 4;  // pattern.hpp:19
constexpr int kDoF = 6;
constexpr int kBlockSize = 8;  // Motion::DoF + 2

/** pattern offsets (x_i, y_i) — pattern.hpp:21-32 */
static const double kPatternData[2 * kPatternSize] = {0, 2, -1, 1, 1, 1, -2, 0, 0, 0, 2, 0, -1, -1, 0, -2};

/** track::PointConnectionStatus — src/track/connections/include/track/connections/frame_connection.hpp:19-25 */
enum PointConnectionStatus : uint8_t { kOk = 0, kOutlier = 1, kOccluded = 2, kOOB = 3, kUnknown = 4 };

/** PinholeCamera + CameraModelBase (pinhole_camera.hpp:21-200, camera_model_base.hpp:30-119), already level-scaled */
struct PinholeModel {
  double width = 0, height = 0, fx = 0, fy = 0, cx = 0, cy = 0;
  static constexpr double kBorderSize = 4;  // camera_model_base.hpp:34
  static constexpr double kMinDepth = 0.001;  // camera_model_base.hpp:36

  /** insideCameraROI — camera_model_base.hpp:52-60 */
  bool insideCameraROI(const double *u, const double *v, int n) const {
    bool ok = true;
    for (int i = 0; i < n; ++i) {
      ok = ok && (u[i] >= kBorderSize) && (v[i] >= kBorderSize) && (u[i] <= width - kBorderSize - 1) &&
           (v[i] <= height - kBorderSize - 1);
    }
    return ok;
  }
  /** validIdepth — camera_model_base.hpp:67-74 */
  bool validIdepth(double idepth) const {
    const double kMinIdepth = -1e-4;
    const double kMaxIdepth = 1 / kMinDepth + 1e1;
    return idepth > kMinIdepth && idepth < kMaxIdepth;
  }
  /** CameraCalibration::cameraModel(level) scaling — camera_calibration.cpp:66-70, pinhole_camera.hpp:38-43,
   *  camera_model_base.cpp:4-6 */
  PinholeModel scaled(int level) const {
    const double s = static_cast<double>(1 << level);
    PinholeModel m;
    m.width = width / s;
    m.height = height / s;
    m.fx = fx / s;
    m.fy = fy / s;
    m.cx = cx / s;
    m.cy = cy / s;
    return m;
  }
};

/** view of a PixelMap<1> level: H x W x (I, dI/dx, dI/dy) AoS — pixel_map.hpp:79-132,141-332 */
struct PixelMapView {
  const double *data = nullptr;
  int width = 0, height = 0;
  const double *texel(int ix, int iy) const { return data + 3 * (static_cast<size_t>(iy) * width + ix); }
};

/** interpolateLinear<true,1> — pixel_map.hpp:20-40: bilinear blend of the stored (I, Ix, Iy) triplets */
inline void interpolateLinear3(const PixelMapView &map, double x, double y, double out[3]) {
  const int ix = static_cast<int>(x);
  const int iy = static_cast<int>(y);
  const double dx = x - static_cast<double>(ix);
  const double dy = y - static_cast<double>(iy);
  const double dxdy = dx * dy;
  const double *p11 = map.texel(ix + 1, iy + 1), *p01 = map.texel(ix, iy + 1), *p10 = map.texel(ix + 1, iy),
               *p00 = map.texel(ix, iy);
  const double w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
  for (int c = 0; c < 3; ++c) out[c] = w11 * p11[c] + w01 * p01[c] + w10 * p10[c] + w00 * p00[c];
}
/** interpolateLinear<false,1> — pixel_map.hpp:36-39 (intensity channel only) */
inline double interpolateLinear1(const PixelMapView &map, double x, double y) {
  const int ix = static_cast<int>(x);
  const int iy = static_cast<int>(y);
  const double dx = x - static_cast<double>(ix);
  const double dy = y - static_cast<double>(iy);
  const double dxdy = dx * dy;
  return dxdy * map.texel(ix + 1, iy + 1)[0] + (dy - dxdy) * map.texel(ix, iy + 1)[0] +
         (dx - dxdy) * map.texel(ix + 1, iy)[0] + (1 - dx - dy + dxdy) * map.texel(ix, iy)[0];
}

/** CameraMask — sensors/camera_calibration/mask/camera_mask.hpp:48-89. data == nullptr means an all-valid mask. */
struct MaskView {
  const uint8_t *data = nullptr;
  int width = 0, height = 0;
  /** valid(int,int) with CHECK_BORDERS (always true on this path, camera_mask.hpp:64-66) */
  bool valid(int x, int y) const {
    if (x < 0 || x >= width) return false;
    if (y < 0 || y >= height) return false;
    return data ? data[static_cast<size_t>(y) * width + x] != 0 : true;
  }
  bool valid(double x, double y) const { return valid(static_cast<int>(std::round(x)), static_cast<int>(std::round(y))); }
  bool valid(const double *u, const double *v, int n) const {
    for (int i = 0; i < n; ++i)
      if (!valid(u[i], v[i])) return false;
    return true;
  }
};

/**
 * ArrayReprojector<double, PinholeCamera, SE3, kCheckSuccess> — camera_reproject.hpp:194-382.
 * N-point pattern reprojection with analytic derivatives w.r.t. idepth and the LEFT perturbation exp(eps)*T_t_r.
 */
template <bool kCheckSuccess = true>
struct ArrayReprojector {
  double reproject_[12];            // K_t [R|t] Kinv_r — camera_reproject.hpp:256
  double project_fx, project_fy, project_cx, project_cy;  // K_t — :257
  double transform_unproject_[12];  // [R|t] Kinv_r — :258
  double translation_[3];           // :259
  PinholeModel reference_model_, target_model_;

  ArrayReprojector(const PinholeModel &reference_model, const PinholeModel &target_model, const SE3 &t_target_reference)
      : reference_model_(reference_model), target_model_(target_model) {
    double M[12];
    t_target_reference.matrix3x4(M);
    const double ifx = 1 / reference_model.fx, ify = 1 / reference_model.fy;
    const double k02 = -reference_model.cx / reference_model.fx, k12 = -reference_model.cy / reference_model.fy;
    // [R|t] * Kinv (4x4 with Kinv(2,2) = Kinv(3,3) = 1) — camera_reproject.hpp:250-258
    for (int i = 0; i < 3; ++i) {
      transform_unproject_[4 * i + 0] = M[4 * i + 0] * ifx;
      transform_unproject_[4 * i + 1] = M[4 * i + 1] * ify;
      transform_unproject_[4 * i + 2] = M[4 * i + 0] * k02 + M[4 * i + 1] * k12 + M[4 * i + 2];
      transform_unproject_[4 * i + 3] = M[4 * i + 3];
    }
    project_fx = target_model.fx;
    project_fy = target_model.fy;
    project_cx = target_model.cx;
    project_cy = target_model.cy;
    // K * ([R|t] Kinv)
    for (int j = 0; j < 4; ++j) {
      reproject_[0 + j] = project_fx * transform_unproject_[0 + j] + project_cx * transform_unproject_[8 + j];
      reproject_[4 + j] = project_fy * transform_unproject_[4 + j] + project_cy * transform_unproject_[8 + j];
      reproject_[8 + j] = transform_unproject_[8 + j];
    }
    for (int i = 0; i < 3; ++i) translation_[i] = t_target_reference.t[i];
  }

  /** reproject without Jacobians — camera_reproject.hpp:270-293 */
  template <int N>
  bool reprojectPattern(const double *ref_u, const double *ref_v, double idepth, double *tgt_u, double *tgt_v) const {
    bool success = true;
    if (kCheckSuccess) success = reference_model_.validIdepth(idepth) && reference_model_.insideCameraROI(ref_u, ref_v, N);
    bool z_positive = true;
    for (int i = 0; i < N; ++i) {
      const double x = reproject_[0] * ref_u[i] + reproject_[1] * ref_v[i] + (reproject_[2] + reproject_[3] * idepth);
      const double y = reproject_[4] * ref_u[i] + reproject_[5] * ref_v[i] + (reproject_[6] + reproject_[7] * idepth);
      const double z = reproject_[8] * ref_u[i] + reproject_[9] * ref_v[i] + (reproject_[10] + reproject_[11] * idepth);
      tgt_u[i] = x / z;
      tgt_v[i] = y / z;
      z_positive = z_positive && (z > 0);
    }
    if (kCheckSuccess) {
      success = success && z_positive;
      success = success && target_model_.insideCameraROI(tgt_u, tgt_v, N);
    }
    return success;
  }

  /** reproject with Jacobians — camera_reproject.hpp:305-367.  d_u_T / d_v_T are N x 6, row-major here. */
  template <int N>
  bool reprojectPattern(const double *ref_u, const double *ref_v, double idepth, double *tgt_u, double *tgt_v,
                        double *d_u_idepth, double *d_v_idepth, double *d_u_T, double *d_v_T) const {
    bool success = true;
    if (kCheckSuccess) success = reference_model_.validIdepth(idepth) && reference_model_.insideCameraROI(ref_u, ref_v, N);
    bool z_positive = true;
    const double *U = transform_unproject_;
    for (int i = 0; i < N; ++i) {
      const double X = U[0] * ref_u[i] + U[1] * ref_v[i] + (U[2] + U[3] * idepth);
      const double Y = U[4] * ref_u[i] + U[5] * ref_v[i] + (U[6] + U[7] * idepth);
      const double Z = U[8] * ref_u[i] + U[9] * ref_v[i] + (U[10] + U[11] * idepth);
      z_positive = z_positive && (Z > 0);
      const double px = project_fx * X + project_cx * Z;
      const double py = project_fy * Y + project_cy * Z;
      tgt_u[i] = px / Z;
      tgt_v[i] = py / Z;
      const double rescaling = 1 / Z;
      const double b0 = X * rescaling, b1 = Y * rescaling;
      d_u_idepth[i] = project_fx * (translation_[0] * rescaling - translation_[2] * (rescaling * b0));
      d_v_idepth[i] = project_fy * (translation_[1] * rescaling - translation_[2] * (rescaling * b1));
      const double new_idepth = idepth * rescaling;
      double *du = d_u_T + 6 * i, *dv = d_v_T + 6 * i;
      dv[0] = 0;
      dv[1] = new_idepth;
      dv[2] = -new_idepth * b1;
      dv[3] = -(b1 * b1 + 1);
      dv[4] = b0 * b1;
      dv[5] = b0;
      du[0] = new_idepth;
      du[1] = 0;
      du[2] = -new_idepth * b0;
      du[3] = -dv[4];
      du[4] = b0 * b0 + 1;
      du[5] = -b1;
      for (int k = 0; k < 6; ++k) {
        du[k] *= project_fx;
        dv[k] *= project_fy;
      }
    }
    if (kCheckSuccess) {
      success = success && z_positive;
      success = success && target_model_.insideCameraROI(tgt_u, tgt_v, N);
    }
    return success;
  }
};

}  // namespace oracle
