// Depth estimation of immature landmarks on the device — row f-1 of SURVEY.md §8:
//   DepthEstimation::estimate / estimateLandmark / findBest / refine — src/tracker/depth_estimators/src/depth_estimation.cpp:26-381
//   EpipolarLineBuilder<Pinhole, SE3>::buildSegment  — src/energy/epipolar_geometry/.../epipolar_line_builder_pinhole_se3.hpp:296-372
//   EpipolarLineTriangulatorSE3                      — .../se3_epipolar_line_triangulator.cpp:7-39
//   EpipolarLine::{length, tangent, shift}           — .../epipolar_line.cpp:18-80
//
// One wavefront per landmark.  The epipolar segment is never materialised: point i is start + i * step (the reference
// accumulates the step i times; the difference is O(1e-13) px), its inverse depth comes from the triangulator on demand.
// findBest: one lane per epipolar point (64 candidates per pass, each lane walks the 8 pattern pixels), energies kept in
// LDS for the uniqueness test, lexicographic (energy, index) wave minimum = the reference's first strict minimum.
// refine (3 LM iterations on the epipolar tangent): lane & 7 owns a pattern pixel, 8-lane DPP sums, so every lane holds the
// same scalars and the control flow stays wave-uniform.  Everything that is scalar per landmark (segment construction,
// error model, re-triangulation of the interval) is executed redundantly by all lanes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <cstring>
#include <limits>
#include <memory>
#include <vector>

#include "common.hpp"
#include "device_geom.hpp"
#include "immature_set.hpp"
#include <chrono>
#include "pyramid.hpp"
#include "se3_math.hpp"

namespace dsopp_hip {
namespace {

// Energies of the epipolar points are kept in LDS for the uniqueness test: one per pixel step of the segment, i.e. at most the
// image diagonal.  The LDS buffer is sized to that per launch (640x480: 6.6 KB instead of a fixed 32 KB), which is what
// bounds the occupancy of this one-wavefront-per-landmark kernel (5 -> 24 workgroups per CU).
inline int depthMaxLine(int W, int H) { return ((static_cast<int>(std::ceil(std::hypot(static_cast<double>(W), static_cast<double>(H)))) + 16 + 63) / 64) * 64; }
enum : uint8_t { kImGood = 0, kImOutOfBoundary = 1, kImOutlier = 2, kImSkipped = 3, kImIllConditioned = 4, kImUninitialized = 5, kImDelete = 6 };

struct DepthFrame {
  const hbm_void *texels;
  int width, height;
  double fx, fy, cx, cy;
  double R[9], t[3];          // T_target_reference
  double M[12];               // K [R|t] K^-1 (ArrayReprojector::reproject_, camera_reproject.hpp:256)
  double Kt[3], KRKi[9];      // triangulator: K t and K R K^-1
  double scale;               // (e_t / e_r) exp(a_t - a_r)
  double b_r, b_t;
  double sigma;
  int n;
  int max_line;  // capacity of the LDS energy buffer (depthMaxLine of the target image)
};

struct DepthLandmarks {
  const hbm_f64 *projection, *direction, *patch, *gradient;
  hbm_f64 *idepth_min, *idepth_max, *uniqueness, *search_pixel_interval;
  hbm_u8 *status, *traced;
};

template <int CTRL>
__device__ __forceinline__ double dppMoveD(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
/** sum over the 8 lanes of an aligned lane group (quad_perm swaps + row_half_mirror); every lane gets the total */
__device__ __forceinline__ double sum8d(double v) {
  v += dppMoveD<0xB1>(v);
  v += dppMoveD<0x4E>(v);
  v += dppMoveD<0x141>(v);
  return v;
}

struct Tri {  // EpipolarLineTriangulatorSE3
  double Kt[3], bearing[3];
  bool use_x;
};
__device__ inline Tri makeTriangulator(const DepthFrame &f, double u, double v) {
  Tri t;
  for (int i = 0; i < 3; ++i) {
    t.Kt[i] = f.Kt[i];
    t.bearing[i] = f.KRKi[3 * i] * u + f.KRKi[3 * i + 1] * v + f.KRKi[3 * i + 2];
  }
  const double kMax = 1000.0, kEps = 1e-5;
  const double a2 = t.bearing[2] + kMax * t.Kt[2], b2 = t.bearing[2] + kEps * t.Kt[2];
  const double dx = (t.bearing[0] + kMax * t.Kt[0]) / a2 - (t.bearing[0] + kEps * t.Kt[0]) / b2;
  const double dy = (t.bearing[1] + kMax * t.Kt[1]) / a2 - (t.bearing[1] + kEps * t.Kt[1]) / b2;
  t.use_x = dx * dx > dy * dy;
  return t;
}
__device__ inline double triInverseDepth(const Tri &t, double px, double py) {
  const double kEps = 1e-5, kMax = 1000.0;
  const double xd = t.Kt[0] - t.Kt[2] * px, yd = t.Kt[1] - t.Kt[2] * py;
  double idepth;
  if (t.use_x || fabs(yd) < kEps)
    idepth = (t.bearing[2] * px - t.bearing[0]) / xd;
  else
    idepth = (t.bearing[2] * py - t.bearing[1]) / yd;
  if (fabs(idepth - kMax) < kEps) idepth = kMax;
  if (fabs(idepth) < kEps) idepth = 0;
  return idepth;
}

/** 1 / x for the projective divisions of the discrete search: v_rcp_f64 + two Newton steps (~1 ulp, 5 instructions instead of ~13 per
 *  IEEE division; as the window's sweeps, pba_kernels.hpp: fastRcp).  x = 0 -> inf -> NaN, rejected by the z > 0 / ROI tests. */
__device__ __forceinline__ double depthRcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

/** checked single-point reprojection (ArrayReprojector<..., true>::reproject, camera_reproject.hpp:270-293) */
__device__ inline bool reproject1(const DepthFrame &f, double u, double v, double idepth, double &tu, double &tv) {
  const double W = f.width, H = f.height;
  bool ok = validIdepth(idepth) && insideROI(u, v, W, H);
  const double x = f.M[0] * u + f.M[1] * v + (f.M[2] + f.M[3] * idepth);
  const double y = f.M[4] * u + f.M[5] * v + (f.M[6] + f.M[7] * idepth);
  const double z = f.M[8] * u + f.M[9] * v + (f.M[10] + f.M[11] * idepth);
  tu = x / z;
  tv = y / z;
  return ok && (z > 0) && insideROI(tu, tv, W, H);
}

/** valid() of the builder — epipolar_line_builder_pinhole_se3.hpp:118-138 */
__device__ inline bool builderValid(const DepthFrame &f, double u, double v, double idepth_1) {
  const double lim = 1 / 0.001 + 1e-4;
  if (idepth_1 < 0 || idepth_1 > lim) return false;
  const double dx = (u - f.cx) * (1 / f.fx), dy = (v - f.cy) * (1 / f.fy);
  const double z = f.R[6] * dx + f.R[7] * dy + f.R[8] + f.t[2] * idepth_1;
  return (1 / z) >= 0 && idepth_1 <= lim;
}

__device__ inline double clampd(double x, double lo, double hi) { return x < lo ? lo : (hi < x ? hi : x); }  // std::clamp

__device__ inline bool generalLine(double k, double s, double W, double H, double *ps, double *pe, double border) {  // :44-85
  if (k == 0) {
    ps[0] = border;
    ps[1] = s;
    pe[0] = W - 1 - border;
    pe[1] = s;
    return s >= border && s <= (H - 1 - border);
  }
  const double x_min = border, y_min = border, x_max = W - 1 - border, y_max = H - 1 - border;
  const double yx0 = k * x_min + s, yx1 = k * x_max + s, xy0 = y_min / k - s / k, xy1 = y_max / k - s / k;
  double y_start = clampd(yx0, y_min, y_max), y_end = clampd(yx1, y_min, y_max);
  const double x_start = clampd(xy0, x_min, x_max), x_end = clampd(xy1, x_min, x_max);
  if (k < 0) {
    const double tmp = y_start;
    y_start = y_end;
    y_end = tmp;
  }
  ps[0] = x_start;
  ps[1] = y_start;
  pe[0] = x_end;
  pe[1] = y_end;
  if (yx0 < border && yx1 < border) return false;
  if (yx0 > (H - 1 - border) && yx1 > (H - 1 - border)) return false;
  if (xy0 < border && xy1 < border) return false;
  if (xy0 > (W - 1 - border) && xy1 > (W - 1 - border)) return false;
  return true;
}

struct Segment {
  int n;            // points (0 = no epipolar line)
  double p0[2];     // point 0 (= point_end_depth)
  double step[2];   // point i = p0 + i * step
  bool single;      // the one-point line (its inverse depth is 0 by construction)
};

/** EpipolarLineBuilder::buildSegment — epipolar_line_builder_pinhole_se3.hpp:296-372 */
__device__ inline Segment buildSegment(const DepthFrame &f, const Tri &tri, double u, double v, double idepthmin, double idepthmax) {
  Segment seg;
  seg.n = 0;
  seg.single = false;
  seg.p0[0] = seg.p0[1] = seg.step[0] = seg.step[1] = 0;
  const double kMaxIdepth = 1000.0, border = 4.0;
  if (sqrt(f.t[0] * f.t[0] + f.t[1] * f.t[1] + f.t[2] * f.t[2]) < 1. / kMaxIdepth) return seg;
  double start[2], end[2], left[2] = {0, 0}, right[2] = {0, 0};
  double lim0 = kMaxIdepth, lim1 = 0;
  const bool zero_rep = reproject1(f, u, v, lim0, start[0], start[1]);
  const bool inf_rep = reproject1(f, u, v, lim1, end[0], end[1]);
  const bool limits_diff = builderValid(f, u, v, lim0) != builderValid(f, u, v, lim1);
  bool intersect;
  {  // intersectImageBorders :87-108
    const double W = f.width, H = f.height;
    const double a = start[1] - end[1], b = end[0] - start[0], c = start[0] * end[1] - end[0] * start[1];
    if (a == 0 && b == 0) {
      intersect = start[0] >= border && start[1] >= border && start[0] <= (W - 1 - border) && start[1] <= (H - 1 - border);
    } else if (b == 0) {
      left[0] = start[0];
      left[1] = border;
      right[0] = start[0];
      right[1] = H - 1 - border;
      intersect = start[0] >= border && start[0] <= (W - 1 - border);
    } else {
      intersect = generalLine(-(a / b), -(c / b), W, H, left, right, border);
    }
  }
  const double border0 = triInverseDepth(tri, left[0], left[1]), border1 = triInverseDepth(tri, right[0], right[1]);
  const bool left_valid = builderValid(f, u, v, border0), right_valid = builderValid(f, u, v, border1);
  const bool borders_diff = left_valid != right_valid;
  {  // one-point line: isApprox with Eigen's dummy precision 1e-12
    const double d2 = (start[0] - end[0]) * (start[0] - end[0]) + (start[1] - end[1]) * (start[1] - end[1]);
    const double na = start[0] * start[0] + start[1] * start[1], nb = end[0] * end[0] + end[1] * end[1];
    if (d2 <= 1e-24 * fmin(na, nb)) {
      if (left_valid) {
        seg.n = 1;
        seg.single = true;
        seg.p0[0] = start[0];
        seg.p0[1] = start[1];
      }
      return seg;
    }
  }
  // findLineBorders :140-201
  const double dir_dot = (end[0] - start[0]) * (right[0] - left[0]) + (end[1] - start[1]) * (right[1] - left[1]);
#define DSOPP_SET2(d, s) \
  do {                   \
    d[0] = s[0];         \
    d[1] = s[1];         \
  } while (0)
  if (limits_diff) {
    if (zero_rep && !inf_rep) {
      if (dir_dot > 0 && border0 > 0) {
        DSOPP_SET2(end, left);
        lim1 = border0;
      } else {
        DSOPP_SET2(end, right);
        lim1 = border1;
      }
    }
    if (inf_rep && !zero_rep) {
      if (dir_dot > 0 && border1 > 0) {
        DSOPP_SET2(start, right);
        lim0 = border1;
      } else {
        DSOPP_SET2(start, left);
        lim0 = border0;
      }
    }
  } else {
    if (zero_rep && !inf_rep) {
      if (dir_dot > 0 && border1 > 0) {
        DSOPP_SET2(end, right);
        lim1 = border1;
      } else {
        DSOPP_SET2(end, left);
        lim1 = border0;
      }
    }
    if (inf_rep && !zero_rep) {
      if (dir_dot > 0 && border0 > 0) {
        DSOPP_SET2(start, left);
        lim0 = border0;
      } else {
        DSOPP_SET2(start, right);
        lim0 = border1;
      }
    }
  }
  if (!zero_rep && !inf_rep) {
    DSOPP_SET2(start, left);
    DSOPP_SET2(end, right);
    lim0 = border0;
    lim1 = border1;
  }
#undef DSOPP_SET2
  if (lim1 > lim0) {
    double tmp = lim0;
    lim0 = lim1;
    lim1 = tmp;
    tmp = start[0];
    start[0] = end[0];
    end[0] = tmp;
    tmp = start[1];
    start[1] = end[1];
    end[1] = tmp;
  }
  // epipolarLineNotExists :203-234
  if (!intersect) return seg;
  if (limits_diff && !zero_rep && !inf_rep && borders_diff) return seg;
  if (!zero_rep && !inf_rep && !left_valid && !right_valid) return seg;
  if (lim1 > idepthmax || lim0 < idepthmin) return seg;
  if (idepthmax < idepthmin) return seg;
  if (lim0 > idepthmax) {
    lim0 = idepthmax;
    reproject1(f, u, v, lim0, start[0], start[1]);
  }
  if (lim1 < idepthmin) {
    lim1 = idepthmin;
    reproject1(f, u, v, lim1, end[0], end[1]);
  }
  const double len = sqrt((end[0] - start[0]) * (end[0] - start[0]) + (end[1] - start[1]) * (end[1] - start[1]));
  unsigned long long size = static_cast<unsigned long long>(len);  // getSize :110-116
  if (size < 1) size = 1;
  seg.n = static_cast<int>(size) + 1;
  seg.p0[0] = end[0];
  seg.p0[1] = end[1];
  seg.step[0] = (start[0] - end[0]) / static_cast<double>(size);
  seg.step[1] = (start[1] - end[1]) / static_cast<double>(size);
  return seg;
}

struct LinePoint {
  double x, y, idepth;
};
__device__ inline LinePoint linePoint(const Segment &s, const Tri &tri, int i) {
  LinePoint p;
  p.x = s.p0[0] + static_cast<double>(i) * s.step[0];
  p.y = s.p0[1] + static_cast<double>(i) * s.step[1];
  p.idepth = s.single ? 0.0 : triInverseDepth(tri, p.x, p.y);
  return p;
}

/** EpipolarLine::shift — epipolar_line.cpp:18-57 (projection only: the interval is re-triangulated from it) */
__device__ inline void lineShift(const Segment &s, const Tri &tri, int idx, double step, double &ox, double &oy) {
  const double seglen = sqrt(s.step[0] * s.step[0] + s.step[1] * s.step[1]);
  step /= seglen;  // |p[i-1] - p[i+1]| / 2 == |p[i] - p[i+1]| == |step| on a straight segment
  const int idx_step = static_cast<int>(round(step));
  double sub = step - static_cast<double>(idx_step);
  const int idx_signed = idx + idx_step;
  if (idx_signed <= 0) {
    const double alpha = step + static_cast<double>(idx);
    const LinePoint p0 = linePoint(s, tri, 0), p1 = linePoint(s, tri, 1);
    ox = p0.x + alpha * (p1.x - p0.x);
    oy = p0.y + alpha * (p1.y - p0.y);
    return;
  }
  if (idx_signed >= s.n - 1) {
    const double alpha = step + static_cast<double>(idx) - (static_cast<double>(s.n) - 1);
    const LinePoint p0 = linePoint(s, tri, s.n - 1), p1 = linePoint(s, tri, s.n - 2);
    ox = p0.x + alpha * (p0.x - p1.x);
    oy = p0.y + alpha * (p0.y - p1.y);
    return;
  }
  const int nb = sub > 0 ? 1 : -1;
  sub = fabs(sub);
  const LinePoint a = linePoint(s, tri, idx_signed), b = linePoint(s, tri, idx_signed + nb);
  ox = (1 - sub) * a.x + sub * b.x;
  oy = (1 - sub) * a.y + sub * b.y;
}

template <typename S>
__device__ inline double sampleI(const Texel<S> *img, int W, double x, double y) {  // interpolateLinear<false,1>, pixel_map.hpp:36-39
  const int ix = static_cast<int>(x), iy = static_cast<int>(y);
  const double dx = x - ix, dy = y - iy, dxdy = dx * dy;
  const Texel<S> *p = img + static_cast<size_t>(iy) * W + ix;
  return dxdy * static_cast<double>(p[W + 1].I) + (dy - dxdy) * static_cast<double>(p[W].I) + (dx - dxdy) * static_cast<double>(p[1].I) +
         (1 - dx - dy + dxdy) * static_cast<double>(p[0].I);
}

template <typename S>
__device__ __forceinline__ void estimateDepthsBody(const DepthFrame &f, const DepthLandmarks &L, const int li, double *energies /* LDS [f.max_line] */) {
  const int lane = threadIdx.x;
  uint8_t status = L.status[li];
  if (status == kImOutOfBoundary || status == kImDelete || status == kImOutlier) return;  // depth_estimation.cpp:246-250
  const Texel<S> *img = static_cast<const Texel<S> *>(f.texels);
  const int W = f.width, H = f.height;
  const double cu = L.projection[2 * li], cv = L.projection[2 * li + 1];
  const double idmin_in = L.idepth_min[li], idmax_in = L.idepth_max[li];
  const bool traced = L.traced[li] != 0;
  const Tri tri = makeTriangulator(f, cu, cv);
  const Segment seg = buildSegment(f, tri, cu, cv, idmin_in, idmax_in);
  auto finish = [&](double interval, uint8_t st) {
    if (lane == 0) {
      L.search_pixel_interval[li] = interval;
      L.status[li] = st;
      if (st == kImGood) L.traced[li] = 1;
    }
  };
  if (seg.n == 0) {  // :256-260
    finish(0, kImOutOfBoundary);
    return;
  }
  const double seglen = sqrt(seg.step[0] * seg.step[0] + seg.step[1] * seg.step[1]);
  const double search_distance = seg.n > 1 ? static_cast<double>(seg.n - 1) * seglen : 0.0;  // EpipolarLine::length
  if (search_distance < 2) {  // kMinEpilineSize :263-267
    finish(search_distance, kImSkipped);
    return;
  }
  const LinePoint front = linePoint(seg, tri, 0);
  const double depth_scale = f.R[6] * L.direction[3 * li] + f.R[7] * L.direction[3 * li + 1] + f.R[8] * L.direction[3 * li + 2] + f.t[2] * front.idepth;
  if (idmin_in >= 0 && (depth_scale < 0.75 || depth_scale > 1.5)) {  // :269-275
    finish(0, kImOutOfBoundary);
    return;
  }
  const double kMaxPixSearch = (static_cast<double>(W) + static_cast<double>(H)) * 0.027;
  int distance = seg.n;
  if (!traced) {
    const unsigned long long lim = static_cast<unsigned long long>(kMaxPixSearch / search_distance * static_cast<double>(seg.n));
    if (lim < static_cast<unsigned long long>(distance)) distance = static_cast<int>(lim);
  }
  if (distance > f.max_line) distance = f.max_line;  // (a segment has one point per pixel step: never longer than the diagonal)
  // ---- findBest :36-76: lane = epipolar point
  double precalc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) precalc[k] = f.scale * (L.patch[8 * li + k] - f.b_r);
  const double kBig = 1.7976931348623157e308;
  double my_best = 1e6;
  int my_idx = 0x7fffffff;
  for (int base = 0; base < distance; base += 64) {
    const int idx = base + lane;
    double energy = kBig;
    if (idx < distance) {
      const LinePoint pt = linePoint(seg, tri, idx);
      // reprojectPattern (checked) of the 8 pattern pixels at this point's inverse depth
      bool ok = validIdepth(pt.idepth);
      double e = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int ox = static_cast<int>((0x21420312u >> (4 * k)) & 0xFu) - 2, oy = static_cast<int>((0x01222334u >> (4 * k)) & 0xFu) - 2;
        const double u = cu + ox, v = cv + oy;
        ok = ok && insideROI(u, v, static_cast<double>(W), static_cast<double>(H));
        const double x = f.M[0] * u + f.M[1] * v + (f.M[2] + f.M[3] * pt.idepth);
        const double y = f.M[4] * u + f.M[5] * v + (f.M[6] + f.M[7] * pt.idepth);
        const double z = f.M[8] * u + f.M[9] * v + (f.M[10] + f.M[11] * pt.idepth);
        const double iz = depthRcp(z);  // (two divisions per pattern pixel were a quarter of the discrete search's instructions)
        const double tu = x * iz, tv = y * iz;
        ok = ok && (z > 0) && insideROI(tu, tv, static_cast<double>(W), static_cast<double>(H));
        if (ok) {
          const double r = (sampleI(img, W, tu, tv) - f.b_t) - precalc[k];
          e += r * r;
        }
      }
      if (ok) {  // camera_mask.valid<false>(point.projection): rounded position, border-checked (camera_mask.hpp:64-66)
        const int mx = static_cast<int>(round(pt.x)), my = static_cast<int>(round(pt.y));
        ok = mx >= 0 && mx < W && my >= 0 && my < H && img[static_cast<size_t>(my) * W + mx].mask != S(0);
      }
      if (ok) energy = e;
      energies[idx] = energy;
      if (energy < my_best) {
        my_best = energy;
        my_idx = idx;
      }
    }
  }
  // lexicographic (energy, index) minimum over the wave = the first strict improvement of the sequential scan
  for (int o = 32; o > 0; o >>= 1) {
    const double oe = __shfl_xor(my_best, o);
    const int oi = __shfl_xor(my_idx, o);
    if (oe < my_best || (oe == my_best && oi < my_idx)) {
      my_best = oe;
      my_idx = oi;
    }
  }
  double best_energy = my_best;
  const int optimum = my_idx == 0x7fffffff ? 0 : my_idx;
  __syncthreads();
  double second = kBig;
  for (int idx = lane; idx < distance; idx += 64)
    if ((idx + 2 < optimum || idx > optimum + 2) && energies[idx] < second) second = energies[idx];
  for (int o = 32; o > 0; o >>= 1) second = fmin(second, __shfl_xor(second, o));
  {  // setUniqueness(second / best, search_distance > kMinEpilineSizeForUniqueness) :299, immature_tracking_landmark.cpp:46-50
    const double uq = second / best_energy;
    if (lane == 0 && (search_distance > 10.0 || uq < L.uniqueness[li])) L.uniqueness[li] = uq;
  }
  // tangent at the optimum — epipolar_line.cpp:59-63, stableNormalized
  const int il = optimum - 1 < 0 ? 0 : optimum - 1, ir = optimum + 1 > seg.n - 1 ? seg.n - 1 : optimum + 1;
  const double ev0 = static_cast<double>(ir - il) * seg.step[0], ev1 = static_cast<double>(ir - il) * seg.step[1];
  double tx = ev0, ty = ev1;
  {
    const double wmax = fmax(fabs(ev0), fabs(ev1));
    if (wmax > 0) {
      const double a = ev0 / wmax, b = ev1 / wmax, nn = sqrt(a * a + b * b);
      tx = a / nn;
      ty = b / nn;
    }
  }
  // ---- refine :184-221: lane & 7 = pattern pixel
  const LinePoint opt = linePoint(seg, tri, optimum);
  const int k = lane & 7;
  const int pox = static_cast<int>((0x21420312u >> (4 * k)) & 0xFu) - 2, poy = static_cast<int>((0x01222334u >> (4 * k)) & 0xFu) - 2;
  double pu, pv;
  {
    const bool ok = reproject1(f, cu + pox, cv + poy, opt.idepth, pu, pv);
    if (__ballot(ok) != ~0ull) {  // all 8 pixels (every group of 8 lanes holds the same 8)
      finish(0, kImOutOfBoundary);
      return;
    }
  }
  const double my_precalc = f.scale * (L.patch[8 * li + k] - f.b_r);
  auto energyAt = [&](double x, double y) {
    const double r = (sampleI(img, W, x, y) - f.b_t) - my_precalc;
    const double rc = fmax(fmin(r, f.sigma), -f.sigma);
    return sum8d(rc * r);
  };
  {
    double lambda = 2.0, energy = energyAt(pu, pv), hessian = 0, b = 0, step = 0;
    double old_u = pu, old_v = pv;
    bool converged = false, linear_system_valid = false;
    for (int it = 0; it < 3 && !converged; ++it) {
      if (!linear_system_valid) {  // linearize :119-142
        const int ix = static_cast<int>(pu), iy = static_cast<int>(pv);
        const double dx = pu - ix, dy = pv - iy, dxdy = dx * dy;
        const Texel<S> *p = img + static_cast<size_t>(iy) * W + ix;
        const double w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
        const Texel<S> t00 = p[0], t10 = p[1], t01 = p[W], t11 = p[W + 1];
        const double sI = w11 * static_cast<double>(t11.I) + w01 * static_cast<double>(t01.I) + w10 * static_cast<double>(t10.I) + w00 * static_cast<double>(t00.I);
        const double sIx = w11 * static_cast<double>(t11.Ix) + w01 * static_cast<double>(t01.Ix) + w10 * static_cast<double>(t10.Ix) + w00 * static_cast<double>(t00.Ix);
        const double sIy = w11 * static_cast<double>(t11.Iy) + w01 * static_cast<double>(t01.Iy) + w10 * static_cast<double>(t10.Iy) + w00 * static_cast<double>(t00.Iy);
        const double r = (sI - f.b_t) - my_precalc;
        const double w = f.sigma * (1.0 / fmax(fabs(r), f.sigma));
        const double d = tx * sIx + ty * sIy;
        hessian = sum8d(w * (d * d));
        b = sum8d(w * (r * d));
      }
      // calculateStep :144-157
      step = b / (hessian + hessian * lambda);
      step = clampd(step, -0.3, 0.3);
      old_u = pu;
      old_v = pv;
      pu -= step * tx;
      pv -= step * ty;
      bool stop = false;
      if (__ballot(insideROI(pu, pv, static_cast<double>(W), static_cast<double>(H))) != ~0ull) {
        pu = old_u;
        pv = old_v;
        stop = true;
      }
      const double next = energyAt(pu, pv);
      if (stop) {  // LM: problem.stop() -> rejectStep, break
        pu = old_u;
        pv = old_v;
        break;
      }
      if (next < energy) {
        if (step * step < 1e-1 * (0 + 1e-1)) converged = true;
        energy = next;
        lambda /= 2.0;
        linear_system_valid = false;
      } else {
        pu = old_u;
        pv = old_v;
        lambda *= 2.0;
        linear_system_valid = true;
      }
    }
    best_energy = energy;
  }
  // centre pixel (pattern index 4) of the refined pattern: lane 4 of every group of 8
  const double sub_u = __shfl(pu, 4), sub_v = __shfl(pv, 4);
  const double sh0 = sub_u - opt.x, sh1 = sub_v - opt.y;
  double shift = sqrt(sh0 * sh0 + sh1 * sh1);
  if (sh0 * ev0 + sh1 * ev1 < 0) shift = -shift;
  if (best_energy > 8 * 144.0) {  // kMaxEnergyForInliers :325-329
    finish(0, kImOutlier);
    return;
  }
  double error;
  {  // calculateError :26-33
    const double g0 = L.gradient[2 * li], g1 = L.gradient[2 * li + 1];
    const double a = pow(ev0 * g0 + ev1 * g1, 2.0), bb = pow(ev1 * g0 - ev0 * g1, 2.0);
    error = 0.2 + 0.2 * (a + bb) / a;
  }
  if (error > search_distance / 2 && traced) {  // :332-336
    finish(search_distance, kImIllConditioned);
    return;
  }
  error = fmin(error, 10.0);
  double idepth_min = -1, idepth_max = -1;
  const double error_step = error / 10.0;
  while ((!validIdepth(idepth_min) || !validIdepth(idepth_max)) && error > -1e-10) {  // :343-349
    double rx, ry, lx, ly;
    lineShift(seg, tri, optimum, -error + shift, rx, ry);
    lineShift(seg, tri, optimum, error + shift, lx, ly);
    idepth_min = triInverseDepth(tri, rx, ry);
    idepth_max = triInverseDepth(tri, lx, ly);
    error -= error_step;
  }
  if (!validIdepth(idepth_min) || !validIdepth(idepth_max)) {
    finish(0, kImOutOfBoundary);
    return;
  }
  if (idepth_min > idepth_max) {
    const double tmp = idepth_min;
    idepth_min = idepth_max;
    idepth_max = tmp;
  }
  if (lane == 0) {
    L.idepth_min[li] = idepth_min;
    L.idepth_max[li] = idepth_max;
  }
  finish(2 * error_step * 10.0, kImGood);
}

#ifndef DSOPP_DEPTH_WAVES
#define DSOPP_DEPTH_WAVES 4  // 141 -> 128 registers: 4 waves per SIMD, 90.9 -> 88.9 us at 14 000 landmarks (5 and 6 spill: 105 / 130 us)
#endif
#if DSOPP_DEPTH_WAVES > 0
#define DSOPP_DEPTH_BOUNDS __launch_bounds__(64, DSOPP_DEPTH_WAVES)
#else
#define DSOPP_DEPTH_BOUNDS __launch_bounds__(64)
#endif
template <typename S>
__global__ void DSOPP_DEPTH_BOUNDS estimateDepthsKernel(DepthFrame f, DepthLandmarks L) {
  extern __shared__ double energies[];  // [f.max_line]
  estimateDepthsBody<S>(f, L, blockIdx.x, energies);
}

/** the same over several keyframes' sets in one launch (blockIdx.y = set): the estimator runs for every keyframe of the
 *  window on every frame (monocular_tracker.cpp:74-102), so one dispatch fills the machine instead of seven partial ones */
template <typename S>
__global__ void DSOPP_DEPTH_BOUNDS estimateDepthsBatchKernel(const DepthFrame *__restrict__ frames, const DepthLandmarks *__restrict__ landmarks) {
  extern __shared__ double energies[];
  const DepthFrame f = frames[blockIdx.y];
  if (static_cast<int>(blockIdx.x) >= f.n) return;
  const DepthLandmarks L = landmarks[blockIdx.y];
  estimateDepthsBody<S>(f, L, blockIdx.x, energies);
}

/** the same launch with the descriptor tables of up to kArgSets keyframes in the kernel arguments (3.6 KB of the 4 KB a dispatch takes): no
 *  table upload in front of the launch — a pinned copy, its copy kernel and the gap behind it were 15 us of the tracker's per-frame call */
constexpr int kArgSets = 8;
struct ArgTables {
  DepthFrame f[kArgSets];
  DepthLandmarks l[kArgSets];
};
static_assert(sizeof(ArgTables) <= 3840, "the tables must fit the kernel-argument segment next to nothing else");
template <typename S>
__global__ void DSOPP_DEPTH_BOUNDS estimateDepthsBatchArgKernel(ArgTables t) {
  extern __shared__ double energies[];
  const DepthFrame f = t.f[blockIdx.y];
  if (static_cast<int>(blockIdx.x) >= f.n) return;
  const DepthLandmarks L = t.l[blockIdx.y];
  estimateDepthsBody<S>(f, L, blockIdx.x, energies);
}

}  // namespace
}  // namespace dsopp_hip

using namespace dsopp_hip;

namespace {

DepthFrame makeDepthFrame(const dsopp_hip_pyramid *target_pyramid, int level, const double intrinsics[4], const double T_target_reference[7],
                          double reference_exposure, const double reference_affine[2], double target_exposure, const double target_affine[2],
                          double sigma_huber_loss, int n) {
  const LevelView lv = target_pyramid->view(level);
  DepthFrame f;
  f.texels = hbm(lv.texels);
  f.width = lv.width;
  f.height = lv.height;
  f.max_line = depthMaxLine(lv.width, lv.height);
  f.fx = intrinsics[0];
  f.fy = intrinsics[1];
  f.cx = intrinsics[2];
  f.cy = intrinsics[3];
  const Rigid T = rigidFromParams(T_target_reference);
  for (int i = 0; i < 9; ++i) f.R[i] = T.R[i];
  for (int i = 0; i < 3; ++i) f.t[i] = T.t[i];
  // reproject_ = K [R|t] K^-1 (camera_reproject.hpp:250-258)
  const double ifx = 1.0 / f.fx, ify = 1.0 / f.fy, k02 = -f.cx / f.fx, k12 = -f.cy / f.fy;
  double U[12];
  for (int i = 0; i < 3; ++i) {
    U[4 * i + 0] = T.R[3 * i + 0] * ifx;
    U[4 * i + 1] = T.R[3 * i + 1] * ify;
    U[4 * i + 2] = T.R[3 * i + 0] * k02 + T.R[3 * i + 1] * k12 + T.R[3 * i + 2];
    U[4 * i + 3] = T.t[i];
  }
  for (int j = 0; j < 4; ++j) {
    f.M[0 + j] = f.fx * U[0 + j] + f.cx * U[8 + j];
    f.M[4 + j] = f.fy * U[4 + j] + f.cy * U[8 + j];
    f.M[8 + j] = U[8 + j];
  }
  const double K[9] = {f.fx, 0, f.cx, 0, f.fy, f.cy, 0, 0, 1};
  const double Kinv[9] = {1 / f.fx, 0, -f.cx / f.fx, 0, 1 / f.fy, -f.cy / f.fy, 0, 0, 1};
  double KR[9];
  for (int i = 0; i < 3; ++i) {
    f.Kt[i] = K[3 * i] * T.t[0] + K[3 * i + 1] * T.t[1] + K[3 * i + 2] * T.t[2];
    for (int j = 0; j < 3; ++j) KR[3 * i + j] = K[3 * i] * T.R[j] + K[3 * i + 1] * T.R[3 + j] + K[3 * i + 2] * T.R[6 + j];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) f.KRKi[3 * i + j] = KR[3 * i] * Kinv[j] + KR[3 * i + 1] * Kinv[3 + j] + KR[3 * i + 2] * Kinv[6 + j];
  f.scale = (target_exposure / reference_exposure) * std::exp(target_affine[0] - reference_affine[0]);
  f.b_r = reference_affine[1];
  f.b_t = target_affine[1];
  f.sigma = sigma_huber_loss;
  f.n = n;
  return f;
}

/** per-launch tables of the batched estimate (one DepthFrame / DepthLandmarks per keyframe set) */
struct Tables {
  DepthFrame f[DSOPP_HIP_MAX_FRAMES];
  DepthLandmarks l[DSOPP_HIP_MAX_FRAMES];
};

DepthLandmarks landmarkPointers(dsopp_hip_immature_set *s) {
  const size_t N = static_cast<size_t>(s->n);
  DepthLandmarks L;
  L.projection = hbm(s->d_in.ptr);
  L.direction = hbm(s->d_in.ptr + 2 * N);
  L.patch = hbm(s->d_in.ptr + 5 * N);
  L.gradient = hbm(s->d_in.ptr + 13 * N);
  L.idepth_min = hbm(s->d_io.ptr);
  L.idepth_max = hbm(s->d_io.ptr + N);
  L.uniqueness = hbm(s->d_io.ptr + 2 * N);
  L.search_pixel_interval = hbm(s->d_io.ptr + 3 * N);
  L.status = hbm(s->d_flags.ptr);
  L.traced = hbm(s->d_flags.ptr + N);
  return L;
}

}  // namespace

extern "C" {

int dsopp_hip_immature_set_create(int device, void *stream, int32_t n, const double *projection, const double *direction, const double *patch,
                                  const double *gradient, dsopp_hip_immature_set **out) {
  return guarded([&] {
    if (!out || n < 0 || (n && (!projection || !direction || !patch || !gradient))) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    auto s = std::make_unique<dsopp_hip_immature_set>();
    s->sr.init(device, stream);
    s->n = n;
    hipStream_t st = s->sr.stream;
    const size_t N = static_cast<size_t>(n);
    s->d_in.reserve(std::max<size_t>(1, N * 15), 0, st);
    s->d_io.reserve(std::max<size_t>(1, N * 4), 0, st);
    s->d_flags.reserve(std::max<size_t>(1, N * 2), 0, st);
    s->d_in.upload(projection, 2 * N, 0, st);
    s->d_in.upload(direction, 3 * N, 2 * N, st);
    s->d_in.upload(patch, 8 * N, 5 * N, st);
    s->d_in.upload(gradient, 2 * N, 13 * N, st);
    // constructor defaults of ImmatureTrackingLandmark (immature_tracking_landmark.hpp:93-106)
    std::vector<double> io(4 * N);
    std::vector<uint8_t> fl(2 * N, 0);
    for (size_t i = 0; i < N; ++i) {
      io[i] = 0;
      io[N + i] = 1. / 0.001;
      io[2 * N + i] = io[3 * N + i] = std::numeric_limits<double>::max();
      fl[i] = kImUninitialized;
    }
    s->d_io.upload(io.data(), 4 * N, 0, st);
    s->d_flags.upload(fl.data(), 2 * N, 0, st);
    // the launch tables of the batched per-frame estimate (this set may be the one that leads a batch): allocated here, at
    // keyframe time, so that the per-frame call allocates nothing (a pinned allocation in its first call cost 0.1 ms)
    HIP_CHECK(hipHostMalloc(&s->h_tables, sizeof(Tables), hipHostMallocDefault));
    HIP_CHECK(hipEventCreateWithFlags(&s->tables_copied, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(s->tables_copied, st));
    s->d_tables.reserve(sizeof(Tables), 0, st);
    s->sr.sync();
    *out = s.release();
  });
}

void dsopp_hip_immature_set_destroy(dsopp_hip_immature_set *s) {
  if (!s) return;
  (void)hipSetDevice(s->sr.device);
  if (s->sr.stream) (void)hipStreamSynchronize(s->sr.stream);
  if (s->h_stage) (void)hipHostFree(s->h_stage);
  if (s->h_tables) (void)hipHostFree(s->h_tables);
  if (s->tables_copied) (void)hipEventDestroy(s->tables_copied);
  StreamRef sr = s->sr;
  delete s;
  sr.destroy();
}

int dsopp_hip_immature_set_upload_state(dsopp_hip_immature_set *s, const double *idepth_min, const double *idepth_max, const double *uniqueness,
                                        const double *search_pixel_interval, const uint8_t *status, const uint8_t *traced) {
  return guarded([&] {
    if (!s) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null set");
    s->sr.use();
    hipStream_t st = s->sr.stream;
    const size_t N = static_cast<size_t>(s->n);
    if (idepth_min) s->d_io.upload(idepth_min, N, 0, st);
    if (idepth_max) s->d_io.upload(idepth_max, N, N, st);
    if (uniqueness) s->d_io.upload(uniqueness, N, 2 * N, st);
    if (search_pixel_interval) s->d_io.upload(search_pixel_interval, N, 3 * N, st);
    if (status) s->d_flags.upload(status, N, 0, st);
    if (traced) s->d_flags.upload(traced, N, N, st);
    s->sr.sync();
  });
}

int dsopp_hip_immature_set_download_state(dsopp_hip_immature_set *s, double *idepth_min, double *idepth_max, double *uniqueness,
                                          double *search_pixel_interval, uint8_t *status, uint8_t *traced) {
  return guarded([&] {
    if (!s) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null set");
    s->sr.use();
    hipStream_t st = s->sr.stream;
    const size_t N = static_cast<size_t>(s->n);
    if (!idepth_min && !idepth_max && !uniqueness && !search_pixel_interval && !status && !traced) {
      s->sr.sync();  // no output requested: a plain wait for the set's stream (the estimator calls are asynchronous)
      return;
    }
    if (N == 0) return;
    // two contiguous copies into pinned staging (the state planes are adjacent on the device), then a host-side scatter:
    // six pageable copies cost six staged transfers
    const size_t bytes = 4 * N * sizeof(double) + 2 * N;
    if (s->h_stage_bytes < bytes) {
      if (s->h_stage) (void)hipHostFree(s->h_stage);
      s->h_stage = nullptr;
      HIP_CHECK(hipHostMalloc(&s->h_stage, bytes, hipHostMallocDefault));
      s->h_stage_bytes = bytes;
    }
    double *hd = static_cast<double *>(s->h_stage);
    uint8_t *hf = reinterpret_cast<uint8_t *>(hd + 4 * N);
    s->d_io.download(hd, 4 * N, 0, st);
    s->d_flags.download(hf, 2 * N, 0, st);
    s->sr.sync();
    if (idepth_min) std::memcpy(idepth_min, hd, N * sizeof(double));
    if (idepth_max) std::memcpy(idepth_max, hd + N, N * sizeof(double));
    if (uniqueness) std::memcpy(uniqueness, hd + 2 * N, N * sizeof(double));
    if (search_pixel_interval) std::memcpy(search_pixel_interval, hd + 3 * N, N * sizeof(double));
    if (status) std::memcpy(status, hf, N);
    if (traced) std::memcpy(traced, hf + N, N);
  });
}

int dsopp_hip_immature_set_estimate(dsopp_hip_immature_set *s, const dsopp_hip_pyramid *target_pyramid, int level, const double intrinsics[4],
                                    const double T_target_reference[7], double reference_exposure, const double reference_affine[2],
                                    double target_exposure, const double target_affine[2], double sigma_huber_loss) {
  return guarded([&] {
    if (!s || !target_pyramid || !intrinsics || !T_target_reference || !reference_affine || !target_affine)
      fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (level < 0 || level >= target_pyramid->levels) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "level out of range");
    if (target_pyramid->sr.device != s->sr.device) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "pyramid lives on another device");
    if (s->n == 0) return;
    s->sr.use();
    hipStream_t st = s->sr.stream;
    target_pyramid->waitReady(st);  // the pyramid build (enqueued on the pyramid's stream) must have finished
    const DepthFrame f = makeDepthFrame(target_pyramid, level, intrinsics, T_target_reference, reference_exposure, reference_affine, target_exposure,
                                        target_affine, sigma_huber_loss, s->n);
    const DepthLandmarks L = landmarkPointers(s);
    if (target_pyramid->dtype == DSOPP_HIP_F64)
      estimateDepthsKernel<double><<<s->n, 64, static_cast<size_t>(f.max_line) * sizeof(double), st>>>(f, L);
    else
      estimateDepthsKernel<float><<<s->n, 64, static_cast<size_t>(f.max_line) * sizeof(double), st>>>(f, L);
    HIP_CHECK(hipGetLastError());
  });
}

int dsopp_hip_immature_sets_estimate(int32_t n_sets, dsopp_hip_immature_set *const *sets, const dsopp_hip_pyramid *target_pyramid, int level,
                                     const double intrinsics[4], const double *T_target_reference, const double *reference_exposure,
                                     const double *reference_affine, double target_exposure, const double target_affine[2],
                                     double sigma_huber_loss) {
  return guarded([&] {
    if (!sets || !target_pyramid || !intrinsics || !T_target_reference || !reference_exposure || !reference_affine || !target_affine)
      fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (n_sets < 1 || n_sets > DSOPP_HIP_MAX_FRAMES) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "n_sets must be in [1, %d]", DSOPP_HIP_MAX_FRAMES);
    if (level < 0 || level >= target_pyramid->levels) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "level out of range");
    dsopp_hip_immature_set *lead = nullptr;
    int max_n = 0;
    for (int k = 0; k < n_sets; ++k) {
      if (!sets[k]) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null set %d", k);
      if (target_pyramid->sr.device != sets[k]->sr.device) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "pyramid lives on another device");
      if (!lead) lead = sets[k];
      max_n = std::max(max_n, sets[k]->n);
    }
    if (max_n == 0) return;
    static const bool trace = std::getenv("DSOPP_HIP_TRACE") != nullptr;  // tuning aid: host-side time of this call's parts
    const auto tr0 = std::chrono::steady_clock::now();
    lead->sr.use();
    hipStream_t st = lead->sr.stream;  // the launch runs on the first set's stream, ordered against the others' below
    target_pyramid->waitReady(st);
    const dim3 grid(static_cast<unsigned>(max_n), static_cast<unsigned>(n_sets));
    if (n_sets <= kArgSets) {
      ArgTables t;
      for (int k = 0; k < n_sets; ++k) {
        dsopp_hip_immature_set *s = sets[k];
        if (s->sr.stream != st) HIP_CHECK(hipStreamSynchronize(s->sr.stream));  // earlier work on that set's own stream
        t.f[k] = makeDepthFrame(target_pyramid, level, intrinsics, T_target_reference + 7 * k, reference_exposure[k], reference_affine + 2 * k,
                                target_exposure, target_affine, sigma_huber_loss, s->n);
        t.l[k] = landmarkPointers(s);
      }
      const size_t smem = static_cast<size_t>(t.f[0].max_line) * sizeof(double);
      if (target_pyramid->dtype == DSOPP_HIP_F64)
        estimateDepthsBatchArgKernel<double><<<grid, 64, smem, st>>>(t);
      else
        estimateDepthsBatchArgKernel<float><<<grid, 64, smem, st>>>(t);
    } else {
      if (!lead->h_tables) {  // (sets created before the tables moved into dsopp_hip_immature_set_create)
        HIP_CHECK(hipHostMalloc(&lead->h_tables, sizeof(Tables), hipHostMallocDefault));
        HIP_CHECK(hipEventCreateWithFlags(&lead->tables_copied, hipEventDisableTiming));
      } else {
        HIP_CHECK(hipEventSynchronize(lead->tables_copied));  // the pinned table is about to be rewritten
      }
      lead->d_tables.reserve(sizeof(Tables), 0, st);
      Tables &t = *static_cast<Tables *>(lead->h_tables);
      for (int k = 0; k < n_sets; ++k) {
        dsopp_hip_immature_set *s = sets[k];
        if (s->sr.stream != st) HIP_CHECK(hipStreamSynchronize(s->sr.stream));  // earlier work on that set's own stream
        t.f[k] = makeDepthFrame(target_pyramid, level, intrinsics, T_target_reference + 7 * k, reference_exposure[k], reference_affine + 2 * k,
                                target_exposure, target_affine, sigma_huber_loss, s->n);
        t.l[k] = landmarkPointers(s);
      }
      HIP_CHECK(hipMemcpyAsync(lead->d_tables.ptr, &t, sizeof(Tables), hipMemcpyHostToDevice, st));
      HIP_CHECK(hipEventRecord(lead->tables_copied, st));
      const DepthFrame *df = reinterpret_cast<const DepthFrame *>(lead->d_tables.ptr);
      const DepthLandmarks *dl = reinterpret_cast<const DepthLandmarks *>(lead->d_tables.ptr + offsetof(Tables, l));
      const size_t smem = static_cast<size_t>(t.f[0].max_line) * sizeof(double);
      if (target_pyramid->dtype == DSOPP_HIP_F64)
        estimateDepthsBatchKernel<double><<<grid, 64, smem, st>>>(df, dl);
      else
        estimateDepthsBatchKernel<float><<<grid, 64, smem, st>>>(df, dl);
    }
    HIP_CHECK(hipGetLastError());
    if (trace)
      std::fprintf(stderr, "[dsopp_hip] immature_sets_estimate: host side of the call %.1f us (tables + upload + launch enqueued)\n",
                   std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - tr0).count());
    // later per-set calls run on the sets' own streams: they must see this launch finished.  (Ordering the other streams behind an event
    // instead — hipStreamWaitEvent on each — was built and measured: six barrier submissions cost more host time, 20-30 us, than this wait
    // saves a caller who synchronises right behind the call anyway, as the tracker does; running the estimator under the optical-flow pass
    // of the same frame lost as well: the two kernels share the device and the flow waits behind the estimator's workgroups.)
    bool other_streams = false;
    for (int k = 0; k < n_sets; ++k) other_streams = other_streams || sets[k]->sr.stream != st;
    if (other_streams) HIP_CHECK(hipStreamSynchronize(st));
  });
}

int dsopp_hip_estimate_depths(const dsopp_hip_pyramid *target_pyramid, int level, const double intrinsics[4], const double T_target_reference[7],
                              double reference_exposure, const double reference_affine[2], double target_exposure, const double target_affine[2],
                              double sigma_huber_loss, int32_t n, const double *projection, const double *direction, const double *patch,
                              const double *gradient, double *idepth_min, double *idepth_max, double *uniqueness, double *search_pixel_interval,
                              uint8_t *status, uint8_t *traced) {
  if (!target_pyramid) return guarded([] { fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null pyramid"); });
  if (n && (!idepth_min || !idepth_max || !uniqueness || !search_pixel_interval || !status || !traced))
    return guarded([] { fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null landmark array"); });
  if (n == 0) return DSOPP_HIP_OK;
  // one-shot form over host arrays: a temporary device-resident set
  dsopp_hip_immature_set *s = nullptr;
  int rc = dsopp_hip_immature_set_create(target_pyramid->sr.device, nullptr, n, projection, direction, patch, gradient, &s);
  if (rc == DSOPP_HIP_OK) rc = dsopp_hip_immature_set_upload_state(s, idepth_min, idepth_max, uniqueness, search_pixel_interval, status, traced);
  if (rc == DSOPP_HIP_OK)
    rc = dsopp_hip_immature_set_estimate(s, target_pyramid, level, intrinsics, T_target_reference, reference_exposure, reference_affine, target_exposure,
                                         target_affine, sigma_huber_loss);
  if (rc == DSOPP_HIP_OK) rc = dsopp_hip_immature_set_download_state(s, idepth_min, idepth_max, uniqueness, search_pixel_interval, status, traced);
  dsopp_hip_immature_set_destroy(s);
  return rc;
}

}  // extern "C"
