// Activation of immature landmarks on the device — row f-3 of SURVEY.md §8:
//   LandmarksActivator::activate, reprojectActivePoints, activationStatus, haveNoNeighbors, recalculateMinDistanceToNeighbor,
//   LandmarkActivationProblem, optimizeImmatureLandmark   — src/tracker/landmarks_activator/src/landmarks_activator.cpp:29-391
//   ImmatureTrackingLandmark::readyForActivation           — src/track/landmarks/src/immature_tracking_landmark.cpp:46-52
//
// Seven launches, no host round trip in between:
//   1. activationProjectKernel — one thread per active / immature landmark: reprojection into the newest keyframe at the
//      sparsity level (pyramid level 1), the per-landmark part of activationStatus, number_of_active_points.
//   2. the selection — P-regulator for min_distance_to_neighbor (recomputed by every kernel from the device-side count), a
//      uniform grid over all reprojected points (cell >= distance, 3 x 3 neighbourhood; counting sort: activationCellCount /
//      CellScan / CellFill), activationNeighboursKernel (one thread per candidate: blocked by an active landmark? which earlier
//      candidates are within the distance?) and activationResolveKernel: the reference's sequential greedy selection
//      ("activate if no earlier accepted point is closer than the distance") resolved in parallel rounds by one workgroup
//      over those short lists — a candidate is decided as soon as every earlier candidate within the distance is decided;
//      the lowest undecided index always is, so the loop terminates with exactly the sequential result (the reference's
//      haveNoNeighbors is O(n^2)).  [First version: everything in one workgroup — 545 us of the call's 670.]
//   3. activationRefineKernel  — one wavefront per accepted candidate: the 3-iteration Levenberg-Marquardt on the inverse
//      depth over all other keyframes; 8 lanes per target keyframe (one per pattern pixel), DPP sums inside a target,
//      target sums accumulated in keyframe order like the reference's loop.
#pragma once
#include <hip/hip_runtime.h>

#include "device_geom.hpp"
#include "pba_types.hpp"
#include "pyramid.hpp"

namespace dsopp_hip {

enum : uint8_t { kActActivate = 0, kActSkip = 1, kActDelete = 2 };                 // ImmatureLandmarkActivationStatus, active_keyframe.hpp:40-44
enum : uint8_t { kImmGood = 0, kImmOutOfBoundary = 1, kImmOutlier = 2, kImmSkipped = 3, kImmIllConditioned = 4, kImmDelete = 6 };
enum : int { kCandNone = -1, kCandUndecided = 0, kCandAccepted = 1, kCandBlocked = 2 };
constexpr int kActSelectThreads = 1024;
constexpr double kActMinCell = 4.0;  // grid cell edge = max(distance, kActMinCell) pixels of the sparsity level

/** one older keyframe (track.activeFrames() without the newest) */
struct ActKeyframe {
  double M[12];  // reproject_ = K1 [R|t] K1^-1 of T_newest^-1 T_keyframe at the sparsity level (camera_reproject.hpp:256)
  int n_active, n_immature, immature_offset, frame_slot;
  const hbm_f64 *active_uv, *active_idepth;
  const hbm_u8 *active_flags;
  const hbm_f64 *projection, *patch, *uniqueness, *search_pixel_interval;
  hbm_f64 *idepth_min, *idepth_max;
  hbm_u8 *status;
  const hbm_u8 *traced;
};

/** reference keyframe r -> target keyframe t at level 0 (the refinement) */
struct ActPair {
  double M[12];  // K [R|t] K^-1   (reproject_, energy evaluation)
  double U[12];  // [R|t] K^-1     (transform_unproject_, linearisation)
  double t[3];
  double scale, b_t, b_r;  // (e_t / e_r) exp(a_t - a_r), affine offsets
};

struct ActArgs {
  const ActKeyframe *keyframes;  // n_keyframes
  const ActPair *pairs;          // [n_frames][n_frames], n_frames = n_keyframes + 1 (the newest keyframe last)
  const void *const *texels0;    // level-0 texel image per frame slot
  const void *newest_sparsity;   // texel image of the newest keyframe at the sparsity level (mask lookup)
  int n_keyframes, n_frames, n_immature, active_cap;
  int width, height;             // level 0
  int sw, sh;                    // sparsity level image size (mask), swd/shd = the camera model's size (may be fractional)
  double swd, shd;
  double fx, fy, cx, cy;         // level 0
  double sigma;
  int desired, minimum_inliers, refine, grid_cap;
  int work_cap;  // undecided candidates the greedy rounds may list in LDS (<= kActWorkCap; DSOPP_HIP_ACT_WORK_CAP: the test of the scan-all fallback)
  // work
  double *px, *py;               // [0, active_cap): reprojected active landmarks; active_cap + g: immature landmark g
  int *counters;                 // 0 number_of_active_points, 1 reprojected active points, 2 accepted, 3 rounds, 4 grid_w, 5 grid_h
  double distance_in;            // LandmarksActivator::min_distance_to_neighbor_ before the regulator
  double *distance;              // [0] out: regulated value
  int *state;                    // per immature landmark
  int *cell_start, *cell_cursor;  // counting sort of the reprojected points into grid cells
  double *sx, *sy;               // the points in cell order ...
  int *sid;                      // ... and who they are: -1 - k active landmark slot k, g >= 0 immature landmark g
  int *nbr, *nbr_count;          // per candidate: the earlier candidates within the distance (kActNbrCap kept)
  int *accepted;
  uint8_t *act_status;           // per immature landmark
  double *idepth_out;            // per immature landmark: idepth() after the call
};

/** CameraMask::valid(point) on the mask lane of a texel image: rounded position, border-checked (camera_mask.hpp:64-66) */
template <typename S>
__device__ __forceinline__ bool maskValid(const Texel<S> *img, int W, int H, double x, double y) {
  const int mx = static_cast<int>(round(x)), my = static_cast<int>(round(y));
  return mx >= 0 && mx < W && my >= 0 && my < H && img[static_cast<size_t>(my) * W + mx].mask != S(0);
}

/** ArrayReprojector<.., true>::reproject of one point — camera_reproject.hpp:270-293 */
__device__ __forceinline__ bool actReproject(const double *M, double u, double v, double idepth, double W, double H, double &tu, double &tv) {
  const bool in = validIdepth(idepth) && insideROI(u, v, W, H);
  const double x = M[0] * u + M[1] * v + (M[2] + M[3] * idepth);
  const double y = M[4] * u + M[5] * v + (M[6] + M[7] * idepth);
  const double z = M[8] * u + M[9] * v + (M[10] + M[11] * idepth);
  tu = x / z;
  tv = y / z;
  return in && (z > 0) && insideROI(tu, tv, W, H);
}

template <typename S>
__global__ void __launch_bounds__(256) activationProjectKernel(ActArgs a) {
  const ActKeyframe &kf = a.keyframes[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const Texel<S> *mask = static_cast<const Texel<S> *>(a.newest_sparsity);
  // ---- reprojectActivePoints :51-87
  {
    bool counted = false, keep = false;
    double tu = 0, tv = 0;
    if (i < kf.n_active) {
      const uint8_t flg = kf.active_flags[i];
      double idepth = kf.active_idepth[i];
      // what updateFrame hands to the keyframe (PROB_SRC/photometric_bundle_adjustment.cpp:233-239): |idepth| < 1e-8 -> 0,
      // idepth < 0 -> outlier
      bool outlier = (flg & kFlagOutlier) != 0;
      if (fabs(idepth) < 1e-8) idepth = 0;
      else if (idepth < 0) outlier = true;
      if (!outlier && !(flg & kFlagMarginalized)) {  // :65
        counted = true;
        keep = actReproject(kf.M, kf.active_uv[2 * i] / 2.0, kf.active_uv[2 * i + 1] / 2.0, idepth, a.swd, a.shd, tu, tv) &&
               maskValid(mask, a.sw, a.sh, tu, tv);
      }
    }
    const unsigned long long cm = __ballot(counted), km = __ballot(keep);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) {
      if (cm) atomicAdd(&a.counters[0], __popcll(cm));
      if (km) base = atomicAdd(&a.counters[1], __popcll(km));
    }
    base = __shfl(base, 0);
    if (keep) {
      const int slot = base + __popcll(km & ((1ull << lane) - 1));
      a.px[slot] = tu;
      a.py[slot] = tv;
    }
  }
  // ---- activationStatus :89-126 without the neighbour test
  if (i < kf.n_immature) {
    const int g = kf.immature_offset + i;
    const uint8_t st = kf.status[i];
    const double idepth = kf.idepth_max[i] * 0.5 + kf.idepth_min[i] * 0.5;  // ImmatureTrackingLandmark::idepth()
    int state = kCandNone;
    uint8_t act;
    if (st == kImmDelete || !kf.traced[i] || st == kImmOutlier) {
      act = kActDelete;
    } else {
      const bool ready = (st == kImmGood || st == kImmSkipped || st == kImmIllConditioned || st == kImmOutOfBoundary) &&
                         (kf.search_pixel_interval[i] < 8.0) && (kf.uniqueness[i] > 3.0) && (idepth > 0);  // readyForActivation
      if (!ready) {
        act = st == kImmOutOfBoundary ? kActDelete : kActSkip;
      } else {
        double tu, tv;
        if (actReproject(kf.M, kf.projection[2 * i] / 2.0, kf.projection[2 * i + 1] / 2.0, idepth, a.swd, a.shd, tu, tv) &&
            maskValid(mask, a.sw, a.sh, tu, tv)) {
          a.px[a.active_cap + g] = tu;
          a.py[a.active_cap + g] = tv;
          state = kCandUndecided;
          act = kActSkip;  // until the selection accepts it
        } else {
          act = kActDelete;
        }
      }
    }
    a.state[g] = state;
    a.act_status[g] = act;
    a.idepth_out[g] = idepth;
    if (act == kActDelete) kf.status[i] = kImmDelete;  // applyImmatureLandmarkActivationStatuses, active_keyframe.cpp:233-235
  }
}

__device__ __forceinline__ int actLoad(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void actStore(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

constexpr int kActNbrCap = 12;  // earlier candidates within the distance kept per candidate (more: that candidate rescans its cells)

/** regulated distance (recalculateMinDistanceToNeighbor :29-39) and the uniform grid derived from it; every kernel of the
 *  selection recomputes it from the same inputs instead of reading it back from a prologue launch */
struct ActGrid {
  double distance, inv_cell;
  int gw, gh, ncell;
};
__device__ __forceinline__ ActGrid actGrid(const ActArgs &a) {
  ActGrid g;
  double d = a.distance_in + (static_cast<double>(a.counters[0]) - static_cast<double>(a.desired)) * 0.001;
  d = d < 0.0 ? 0.0 : (d > 10.0 ? 10.0 : d);
  g.distance = d;
  const double cell = d > kActMinCell ? d : kActMinCell;
  g.inv_cell = 1.0 / cell;
  g.gw = static_cast<int>(a.swd * g.inv_cell) + 1;
  g.gh = static_cast<int>(a.shd * g.inv_cell) + 1;
  g.ncell = g.gw * g.gh;  // <= grid_cap by construction of the host (cell >= kActMinCell)
  return g;
}
__device__ __forceinline__ int actCellOf(const ActGrid &g, double x, double y) {
  int cx = static_cast<int>(x * g.inv_cell), cy = static_cast<int>(y * g.inv_cell);
  cx = cx < 0 ? 0 : (cx >= g.gw ? g.gw - 1 : cx);
  cy = cy < 0 ? 0 : (cy >= g.gh ? g.gh - 1 : cy);
  return cy * g.gw + cx;
}
/** slot of point k in px / py (k < active_cap: reprojected active landmark, else immature landmark k - active_cap) and
 *  whether it takes part in the selection */
__device__ __forceinline__ bool actPointLive(const ActArgs &a, int k) {
  return k < a.active_cap ? k < a.counters[1] : a.state[k - a.active_cap] == kCandUndecided;
}

/** counting sort, step 1: points per cell (cell_start zeroed by the host) */
__global__ void __launch_bounds__(256) activationCellCountKernel(ActArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.active_cap + a.n_immature || !actPointLive(a, k)) return;
  const ActGrid g = actGrid(a);
  atomicAdd(&a.cell_start[actCellOf(g, a.px[k], a.py[k])], 1);
}

/** step 2: exclusive scan of the cell counts (one workgroup: the grid has a few thousand cells) */
__global__ void __launch_bounds__(kActSelectThreads) activationCellScanKernel(ActArgs a) {
  __shared__ int s_scan[kActSelectThreads];
  const int tid = threadIdx.x;
  const ActGrid g = actGrid(a);
  const int ncell = g.ncell;
  const int chunk = (ncell + kActSelectThreads - 1) / kActSelectThreads;
  const int c0 = tid * chunk, c1 = c0 + chunk < ncell ? c0 + chunk : ncell;
  int sum = 0;
  for (int c = c0; c < c1; ++c) sum += a.cell_start[c];
  s_scan[tid] = sum;
  __syncthreads();
  for (int off = 1; off < kActSelectThreads; off <<= 1) {
    const int v = tid >= off ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  int run = s_scan[tid] - sum;
  for (int c = c0; c < c1; ++c) {
    const int n = a.cell_start[c];
    a.cell_start[c] = run;
    a.cell_cursor[c] = run;
    run += n;
  }
  if (tid == kActSelectThreads - 1) a.cell_start[ncell] = s_scan[tid];
}

/** step 3: points into cell order (coordinates and identity side by side: a cell scan reads three contiguous arrays) */
__global__ void __launch_bounds__(256) activationCellFillKernel(ActArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.active_cap + a.n_immature || !actPointLive(a, k)) return;
  const ActGrid g = actGrid(a);
  const double x = a.px[k], y = a.py[k];
  const int pos = atomicAdd(&a.cell_cursor[actCellOf(g, x, y)], 1);
  a.sx[pos] = x;
  a.sy[pos] = y;
  a.sid[pos] = k < a.active_cap ? -1 - k : k - a.active_cap;
}

/** scan of the 3 x 3 cell neighbourhood of candidate g; calls f(id) for every point closer than the distance (id < 0: an
 *  active landmark) until f returns false */
template <typename F>
__device__ __forceinline__ void actForNeighbours(const ActArgs &a, const ActGrid &g, double x, double y, F f) {
  const int c = actCellOf(g, x, y), cx = c % g.gw, cy = c / g.gw;
  for (int yy = (cy > 0 ? cy - 1 : 0); yy <= (cy + 1 < g.gh ? cy + 1 : g.gh - 1); ++yy) {
    const int lo = a.cell_start[yy * g.gw + (cx > 0 ? cx - 1 : 0)], hi = a.cell_start[yy * g.gw + (cx + 1 < g.gw ? cx + 1 : g.gw - 1) + 1];
    for (int it = lo; it < hi; ++it) {  // the three cells of a grid row are contiguous in cell order
      const double dx = a.sx[it] - x, dy = a.sy[it] - y;
      if (sqrt(dx * dx + dy * dy) < g.distance)
        if (!f(a.sid[it])) return;
    }
  }
}

/** haveNoNeighbors (:41-49), the order-independent part: a candidate within the distance of a reprojected ACTIVE landmark
 *  is blocked whatever the order; the earlier candidates within the distance are remembered for the greedy rounds */
__global__ void __launch_bounds__(256) activationNeighboursKernel(ActArgs a) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= a.n_immature || a.state[g] != kCandUndecided) return;
  const ActGrid grid = actGrid(a);
  const double x = a.px[a.active_cap + g], y = a.py[a.active_cap + g];
  bool blocked = false;
  int cnt = 0;
  int *list = a.nbr + static_cast<size_t>(g) * kActNbrCap;
  actForNeighbours(a, grid, x, y, [&](int id) {
    if (id < 0) {
      blocked = true;
      return false;
    }
    if (id < g) {
      if (cnt < kActNbrCap) list[cnt] = id;
      ++cnt;
    }
    return true;
  });
  // (no earlier candidate within the distance: accepted whatever the others become — decided here, on all CUs, instead of in the one
  // workgroup of the greedy rounds; in the tracker's scenes that is the majority of the candidates)
  if (blocked) a.state[g] = kCandBlocked;
  else if (cnt == 0) a.state[g] = kCandAccepted;
  a.nbr_count[g] = cnt;
}

/** the reference's sequential greedy selection ("activate if no earlier accepted point is closer than the distance")
 *  resolved in parallel rounds by one workgroup: a candidate is decided as soon as every earlier candidate within the distance
 *  is decided; the lowest undecided index always is, so the loop terminates with exactly the sequential result */
constexpr int kActWorkCap = 8192;  // undecided candidates the greedy rounds keep as a list in LDS (more: every round scans all candidates)
__global__ void __launch_bounds__(kActSelectThreads) activationResolveKernel(ActArgs a) {
  __shared__ int s_work[2][kActWorkCap];
  __shared__ int s_n[2];
  const int tid = threadIdx.x;
  const ActGrid grid = actGrid(a);
  const int nI = a.n_immature;
  const int cap = a.work_cap < kActWorkCap ? a.work_cap : kActWorkCap;
  // one candidate's step of a round: blocked by an accepted earlier neighbour, accepted when all of them are decided, else it waits
  auto step = [&](int g) -> bool {
    bool blocked = false, wait = false;
    const int cnt = a.nbr_count[g];
    if (cnt <= kActNbrCap) {
      const int *list = a.nbr + static_cast<size_t>(g) * kActNbrCap;
      for (int k = 0; k < cnt; ++k) {
        const int sj = actLoad(&a.state[list[k]]);
        blocked = blocked || sj == kCandAccepted;
        wait = wait || sj == kCandUndecided;
      }
    } else {  // more neighbours than the list holds: rescan the cells
      actForNeighbours(a, grid, a.px[a.active_cap + g], a.py[a.active_cap + g], [&](int id) {
        if (id >= 0 && id < g) {
          const int sj = actLoad(&a.state[id]);
          blocked = blocked || sj == kCandAccepted;
          wait = wait || sj == kCandUndecided;
        }
        return true;
      });
    }
    if (blocked) actStore(&a.state[g], kCandBlocked);
    else if (!wait) actStore(&a.state[g], kCandAccepted);
    return !blocked && wait;  // still undecided
  };
  // the candidates the neighbour pass left undecided (those with an earlier candidate within the distance), as a list: a round touches
  // only them (every round walked all candidates before: 9 dependent load chains per thread and round for 9000 candidates)
  if (tid < 2) s_n[tid] = 0;
  __syncthreads();
  for (int g = tid; g < nI; g += kActSelectThreads)
    if (actLoad(&a.state[g]) == kCandUndecided) {
      const int pos = atomicAdd(&s_n[0], 1);
      if (pos < cap) s_work[0][pos] = g;
    }
  __syncthreads();
  int rounds = 0;
  if (s_n[0] <= cap) {
    int cur = 0;
    for (;;) {
      const int n = s_n[cur];
      if (n == 0) {
        if (rounds == 0) rounds = 1;  // (nothing was left to decide: the neighbour pass was the round)
        break;
      }
      for (int i = tid; i < n; i += kActSelectThreads) {
        const int g = s_work[cur][i];
        if (step(g)) s_work[cur ^ 1][atomicAdd(&s_n[cur ^ 1], 1)] = g;
      }
      ++rounds;
      __syncthreads();
      if (tid == 0) s_n[cur] = 0;
      cur ^= 1;
      __syncthreads();
      if (rounds > nI + 1) break;  // (the bound is the longest possible dependency chain)
    }
  } else {
    for (;;) {
      int undecided = 0;
      for (int g = tid; g < nI; g += kActSelectThreads) {
        if (actLoad(&a.state[g]) != kCandUndecided) continue;
        if (step(g)) undecided = 1;
      }
      ++rounds;
      if (!__syncthreads_or(undecided) || rounds > nI + 1) break;
    }
  }
  for (int g = tid; g < nI; g += kActSelectThreads)
    if (a.state[g] == kCandAccepted) {
      a.act_status[g] = kActActivate;
      a.accepted[atomicAdd(&a.counters[2], 1)] = g;
    }
  if (tid == 0) {
    a.counters[3] = rounds;
    a.counters[4] = grid.gw;
    a.counters[5] = grid.gh;
    a.distance[0] = grid.distance;
  }
}

template <int CTRL>
__device__ __forceinline__ double actDppMove(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
/** sum over the 8 lanes of an aligned lane group; every lane of the group gets the total */
__device__ __forceinline__ double actSum8(double v) {
  v += actDppMove<0xB1>(v);
  v += actDppMove<0x4E>(v);
  v += actDppMove<0x141>(v);
  return v;
}
__device__ __forceinline__ double actReadLane(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}

/** optimizeImmatureLandmark :279-311 — one wavefront per accepted candidate */
template <typename S>
__global__ void __launch_bounds__(64) activationRefineKernel(ActArgs a) {
  const int w = blockIdx.x;
  if (w >= a.counters[2]) return;
  const int g = a.accepted[w], lane = threadIdx.x, slot = lane >> 3, k = lane & 7;
  int kfi = 0;
  while (kfi + 1 < a.n_keyframes && g >= a.keyframes[kfi + 1].immature_offset) ++kfi;
  const ActKeyframe &kf = a.keyframes[kfi];
  const int i = g - kf.immature_offset, r = kf.frame_slot, F = a.n_frames;
  const int pox = static_cast<int>((0x21420312u >> (4 * k)) & 0xFu) - 2, poy = static_cast<int>((0x01222334u >> (4 * k)) & 0xFu) - 2;  // pattern.hpp:21-32
  const double ru = kf.projection[2 * i] + pox, rv = kf.projection[2 * i + 1] + poy;  // PatternPatch::shiftPattern
  const double patch = kf.patch[8 * i + k];
  const double W = a.width, H = a.height;
  const bool ref_inside = __ballot(insideROI(ru, rv, W, H)) == ~0ull;  // the same 8 pattern pixels in every lane group
  const double kMaxEnergyForInliers = 8 * 12 * 12;                      // :130
  const int passes = (F - 1 + 7) / 8;

  double idepth = kf.idepth_max[i] * 0.5 + kf.idepth_min[i] * 0.5, old_idepth = idepth;
  double hessian = 0, b = 0, step = 0;
  bool stop = false;

  // calculateEnergy :150-202 — returns the energy, n_valid through the reference
  auto calculateEnergy = [&](int &n_valid) -> double {
    double energy = 0;
    n_valid = 0;
    if (stop) {
      idepth = -1;
      return energy;
    }
    for (int p = 0; p < passes; ++p) {
      const int tt = p * 8 + slot, t = tt < r ? tt : tt + 1;
      const bool has = tt < F - 1;
      bool ok = false;
      double tu = 0, tv = 0;
      const ActPair *pc = a.pairs + static_cast<size_t>(r) * F + (has ? t : 0);
      const Texel<S> DSOPP_HBM *img = (const Texel<S> DSOPP_HBM *)a.texels0[has ? t : 0];
      if (has) {
        const double *M = pc->M;
        const double x = M[0] * ru + M[1] * rv + (M[2] + M[3] * idepth);
        const double y = M[4] * ru + M[5] * rv + (M[6] + M[7] * idepth);
        const double z = M[8] * ru + M[9] * rv + (M[10] + M[11] * idepth);
        tu = x / z;
        tv = y / z;
        ok = validIdepth(idepth) && ref_inside && (z > 0) && insideROI(tu, tv, W, H) && maskValid(img, a.width, a.height, tu, tv);
      }
      const unsigned long long okm = __ballot(ok);
      const bool gok = ((okm >> (8 * slot)) & 0xFFull) == 0xFFull;
      double rr = 0;
      if (gok) {
        const int ix = static_cast<int>(tu), iy = static_cast<int>(tv);
        const double dx = tu - ix, dy = tv - iy, dxdy = dx * dy;
        const Texel<S> DSOPP_HBM *q = img + static_cast<size_t>(iy) * a.width + ix;
        const double sI = dxdy * static_cast<double>(q[a.width + 1].I) + (dy - dxdy) * static_cast<double>(q[a.width].I) +
                          (dx - dxdy) * static_cast<double>(q[1].I) + (1 - dx - dy + dxdy) * static_cast<double>(q[0].I);
        rr = (sI - pc->b_t) - pc->scale * (patch - pc->b_r);
      }
      const double sq = actSum8(rr * rr);
      const double norm = sqrt(sq);
      const double hw = norm > a.sigma ? a.sigma / norm : 1.0;
      const bool inlier = sq < kMaxEnergyForInliers;
      const double et = inlier ? hw * sq : kMaxEnergyForInliers;
#pragma unroll
      for (int s = 0; s < 8; ++s) {  // keyframe order, like the reference's loop over frames_
        const double es = actReadLane(et, 8 * s);
        const unsigned long long in_s = __ballot(inlier) >> (8 * s);
        if (((okm >> (8 * s)) & 0xFFull) == 0xFFull) {
          energy += es;
          n_valid += static_cast<int>(in_s & 1ull);
        }
      }
    }
    if (n_valid == 0) {
      idepth = -1;
      stop = true;
    }
    return energy;
  };
  // linearize :204-256
  auto linearize = [&]() {
    hessian = 0;
    b = 0;
    for (int p = 0; p < passes; ++p) {
      const int tt = p * 8 + slot, t = tt < r ? tt : tt + 1;
      const bool has = tt < F - 1;
      bool ok = false;
      double tu = 0, tv = 0, dui = 0, dvi = 0;
      const ActPair *pc = a.pairs + static_cast<size_t>(r) * F + (has ? t : 0);
      const Texel<S> DSOPP_HBM *img = (const Texel<S> DSOPP_HBM *)a.texels0[has ? t : 0];
      if (has) {  // reproject with Jacobians — camera_reproject.hpp:305-367
        const double *U = pc->U;
        const double X = U[0] * ru + U[1] * rv + (U[2] + U[3] * idepth);
        const double Y = U[4] * ru + U[5] * rv + (U[6] + U[7] * idepth);
        const double Z = U[8] * ru + U[9] * rv + (U[10] + U[11] * idepth);
        tu = (a.fx * X + a.cx * Z) / Z;
        tv = (a.fy * Y + a.cy * Z) / Z;
        const double rescaling = 1 / Z, b0 = X * rescaling, b1 = Y * rescaling;
        dui = a.fx * (pc->t[0] * rescaling - pc->t[2] * (rescaling * b0));
        dvi = a.fy * (pc->t[1] * rescaling - pc->t[2] * (rescaling * b1));
        ok = validIdepth(idepth) && ref_inside && (Z > 0) && insideROI(tu, tv, W, H) && maskValid(img, a.width, a.height, tu, tv);
      }
      const unsigned long long okm = __ballot(ok);
      const bool gok = ((okm >> (8 * slot)) & 0xFFull) == 0xFFull;
      double rr = 0, d = 0;
      if (gok) {
        const int ix = static_cast<int>(tu), iy = static_cast<int>(tv);
        const double dx = tu - ix, dy = tv - iy, dxdy = dx * dy;
        const Texel<S> DSOPP_HBM *q = img + static_cast<size_t>(iy) * a.width + ix;
        const double w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
        const Texel<S> t00 = loadTexel(q), t10 = loadTexel(q + 1), t01 = loadTexel(q + a.width), t11 = loadTexel(q + a.width + 1);
        const double sI = w11 * static_cast<double>(t11.I) + w01 * static_cast<double>(t01.I) + w10 * static_cast<double>(t10.I) + w00 * static_cast<double>(t00.I);
        const double sIx = w11 * static_cast<double>(t11.Ix) + w01 * static_cast<double>(t01.Ix) + w10 * static_cast<double>(t10.Ix) + w00 * static_cast<double>(t00.Ix);
        const double sIy = w11 * static_cast<double>(t11.Iy) + w01 * static_cast<double>(t01.Iy) + w10 * static_cast<double>(t10.Iy) + w00 * static_cast<double>(t00.Iy);
        rr = (sI - pc->b_t) - pc->scale * (patch - pc->b_r);
        d = sIx * dui + sIy * dvi;
      }
      const double sq = actSum8(rr * rr);
      const double norm = sqrt(sq);
      const double hw = norm > a.sigma ? a.sigma / norm : 1.0;
      const double dd = actSum8((hw * d) * d), dr = actSum8((hw * d) * rr);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const double dds = actReadLane(dd, 8 * s), drs = actReadLane(dr, 8 * s);
        if (((okm >> (8 * s)) & 0xFFull) == 0xFFull) {
          hessian += dds;
          b += drs;
        }
      }
    }
    if (hessian == 0) stop = true;
  };

  // levenberg_marquardt_algorithm::solve — levenberg_marquardt_algorithm.hpp:77-128 with the options of :286-292
  // (max 3 iterations, lambda 0.1, /2 on accept, x5 on reject, function tolerance 0, parameter tolerance 1e-8)
  int result_valid = 0;
  double result_energy = 0;
  if (a.refine) {
    double lambda = 1.0 / 10;
    result_energy = calculateEnergy(result_valid);
    bool linear_system_valid = false, converged = false;
    for (int it = 0; it < 3 && !converged && result_valid > 0; ++it) {
      if (!linear_system_valid) linearize();
      step = b / (hessian + hessian * lambda);  // calculateStep :258-262
      old_idepth = idepth;
      idepth -= step;
      int n_valid;
      const double next_energy = calculateEnergy(n_valid);
      if (n_valid == 0) {
        idepth = old_idepth;  // rejectStep
        break;
      }
      // function tolerance 0: |dE| / E < 0 never holds
      if (next_energy < result_energy) {
        const double state_sq = idepth * idepth, step_sq = step * step;  // acceptStep :264
        converged = converged || (step_sq < 1e-8 * (state_sq + 1e-8));
        result_energy = next_energy;
        result_valid = n_valid;
        lambda /= 2.0;
        linear_system_valid = false;
      } else {
        idepth = old_idepth;
        lambda *= 5.0;
        linear_system_valid = true;
      }
    }
    int unused;
    calculateEnergy(unused);  // :126 (its side effect on idepth_ is what matters)
  }
  if (lane == 0) {
    if (a.refine && (result_valid < a.minimum_inliers || idepth < 0)) {  // :305-306
      a.act_status[g] = kActDelete;
    } else if (a.refine) {
      kf.idepth_min[i] = idepth;  // :308-309
      kf.idepth_max[i] = idepth;
      a.idepth_out[g] = idepth;
    }
    kf.status[i] = kImmDelete;  // activated or deleted: active_keyframe.cpp:232-235
  }
}

}  // namespace dsopp_hip
