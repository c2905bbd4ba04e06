// Kernels of createReferenceDepthMaps (src/tracker/tracker/src/create_depth_maps.cpp:18-147).  Tiny, HBM-streaming work:
// a splat of every older keyframe's landmarks (fp64 atomics into the fine map), the 2x2 sum-pools of all coarser levels from
// level-0 tiles, the dilation of all levels (separate input / output planes: the reference reads neighbours through a backup of the
// weights, :94-99).
#pragma once
#include <hip/hip_runtime.h>

#include "device_geom.hpp"
#include "pba_types.hpp"

namespace dsopp_hip {

struct SplatArgs {
  double M[12];        // reproject_ = K [R|t] K^-1 of T_newest^-1 T_source (camera_reproject.hpp:256)
  double Tz[4];        // row 2 of [R|t]: getDepthScale = Tz . [direction; idepth] (camera_model_base.hpp:102-107)
  double fx, fy, cx, cy;
  int width, height;
  int n;
  int use_variance;    // 0: every landmark has variance 1e-5 (estimate_uncertainty off, photometric_bundle_adjustment.cpp:254)
  const double *uv, *idepth, *inv_hdd;
  const uint8_t *flags, *status;
};

// One launch per step (until round 6: one splat per older keyframe, one pool per coarser level, one dilation per level — 17 launches per
// keyframe whose launch gaps were half of the call's 0.15 ms)
constexpr int kSplatMaxSources = kMaxFrames - 1;
struct SplatBatch {
  SplatArgs src[kSplatMaxSources];  // (3.3 KB of kernel arguments: below the 4 KB the dispatch packet takes)
};
/** fillFineDepthMap — create_depth_maps.cpp:18-59 — for every older keyframe: blockIdx.y = source, one thread per landmark */
__global__ void __launch_bounds__(256) splatDepthMapsKernel(SplatBatch b, double *__restrict__ idsum, double *__restrict__ wsum) {
  const SplatArgs &a = b.src[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  if (a.status[i] != DSOPP_HIP_STATUS_OK) return;                 // :36
  const uint8_t flg = a.flags[i];
  if (flg & (kFlagMarginalized | kFlagOutlier)) return;           // :38
  double idepth = a.idepth[i];
  // what updateFrame hands to the keyframe (PROB_SRC/photometric_bundle_adjustment.cpp:233-239): |idepth| < 1e-8 -> 0,
  // idepth < 0 -> outlier
  if (fabs(idepth) < 1e-8) idepth = 0;
  else if (idepth < 0) return;
  const double u = a.uv[2 * i], v = a.uv[2 * i + 1];
  const double W = a.width, H = a.height;
  // ArrayReprojector<..., kCheckSuccess = true>::reproject — camera_reproject.hpp:270-293
  if (!(validIdepth(idepth) && insideROI(u, v, W, H))) return;
  const double x = a.M[0] * u + a.M[1] * v + (a.M[2] + a.M[3] * idepth);
  const double y = a.M[4] * u + a.M[5] * v + (a.M[6] + a.M[7] * idepth);
  const double z = a.M[8] * u + a.M[9] * v + (a.M[10] + a.M[11] * idepth);
  const double tu = x / z, tv = y / z;
  if (!(z > 0) || !insideROI(tu, tv, W, H)) return;
  const int ix = static_cast<int>(round(tu)), iy = static_cast<int>(round(tv));  // :46
  const double dx = (u - a.cx) * (1.0 / a.fx), dy = (v - a.cy) * (1.0 / a.fy);
  const double depth_scale = a.Tz[0] * dx + a.Tz[1] * dy + a.Tz[2] * 1.0 + a.Tz[3] * idepth;  // :49
  const double variance = a.use_variance ? a.inv_hdd[i] : 1e-5;
  const double weight = sqrt(1e-3 / (variance + 1e-12));  // :51
  const size_t cell = static_cast<size_t>(iy) * a.width + ix;
  atomicAdd(&idsum[cell], idepth / depth_scale * weight);  // :52
  atomicAdd(&wsum[cell], weight);                          // :53
}

constexpr int kDepthMapLevels = 5;  // dsopp_hip_window_create_reference_depth_maps: levels in [1, 5]
struct DepthMapLevels {
  double *id[kDepthMapLevels], *w[kDepthMapLevels];          // undilated planes (pool: level 0 read, coarser written; dilation: read)
  double *out_id[kDepthMapLevels], *out_w[kDepthMapLevels];  // dilated planes (dilation only)
  int width[kDepthMapLevels], height[kDepthMapLevels];
  int first_row[kDepthMapLevels + 1];                        // dilation: blockIdx.y in [first_row[l], first_row[l + 1]) works on level l
  int levels;
};
/** fillCoarseDepthMaps — create_depth_maps.cpp:70-88 — for all coarser levels: a workgroup owns a 16 x 16 tile of level 0 and pools it down in LDS — every coarse pixel is the
 *  sum of its four children in one fixed order (a, a + 1, row below: b, b + 1).  Level sizes halve with floor: the
 *  children of an existing pixel exist. */
__global__ void __launch_bounds__(256) poolDepthMapsKernel(DepthMapLevels L) {
  __shared__ double s_id[2][16 * 16], s_w[2][16 * 16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int x0 = blockIdx.x * 16 + tx, y0 = blockIdx.y * 16 + ty;
  const bool in0 = x0 < L.width[0] && y0 < L.height[0];
  s_id[0][threadIdx.x] = in0 ? L.id[0][static_cast<size_t>(y0) * L.width[0] + x0] : 0.0;
  s_w[0][threadIdx.x] = in0 ? L.w[0][static_cast<size_t>(y0) * L.width[0] + x0] : 0.0;
  __syncthreads();
  int side = 16, cur = 0;
  for (int l = 1; l < L.levels; ++l) {
    const int half = side >> 1;
    if (half == 0) break;
    if (tx < half && ty < half) {
      const int a = (2 * ty) * side + 2 * tx, b = a + side;
      const double vi = s_id[cur][a] + s_id[cur][a + 1] + s_id[cur][b] + s_id[cur][b + 1];
      const double vw = s_w[cur][a] + s_w[cur][a + 1] + s_w[cur][b] + s_w[cur][b + 1];
      s_id[cur ^ 1][ty * half + tx] = vi;
      s_w[cur ^ 1][ty * half + tx] = vw;
      const int x = blockIdx.x * half + tx, y = blockIdx.y * half + ty;
      if (x < L.width[l] && y < L.height[l]) {
        L.id[l][static_cast<size_t>(y) * L.width[l] + x] = vi;
        L.w[l][static_cast<size_t>(y) * L.width[l] + x] = vw;
      }
    }
    __syncthreads();
    side = half;
    cur ^= 1;
  }
}

/** dilateDepthMaps — create_depth_maps.cpp:90-122 — for all levels: blockIdx.y walks the rows of level 0, then of level 1, ...; in -> out planes */
__global__ void __launch_bounds__(128) dilateDepthMapsKernel(DepthMapLevels L) {
  int l = 0;
  while (l + 1 < L.levels && static_cast<int>(blockIdx.y) >= L.first_row[l + 1]) ++l;
  const int width = L.width[l], height = L.height[l];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = static_cast<int>(blockIdx.y) - L.first_row[l];
  if (x >= width || y >= height) return;
  const double *__restrict__ in_id = L.id[l], *__restrict__ in_w = L.w[l];
  const int diagonal = l > 1 ? 0 : 1;
  const size_t c = static_cast<size_t>(y) * width + x;
  double id = in_id[c], w = in_w[c];
  if (!(w > 0) && x >= 1 && y >= 1 && x < width - 1 && y < height - 1) {
    double sum = 0, num = 0, numn = 0;
    // offsets in the reference's order (:105-109): level > 1: (1,0) (-1,0) (0,1) (0,-1); else (1,1) (-1,-1) (1,-1) (-1,1)
    constexpr int kAxisX[4] = {1, -1, 0, 0}, kAxisY[4] = {0, 0, 1, -1};
    constexpr int kDiagX[4] = {1, -1, 1, -1}, kDiagY[4] = {1, -1, -1, 1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ox = diagonal ? kDiagX[k] : kAxisX[k], oy = diagonal ? kDiagY[k] : kAxisY[k];
      const size_t n = static_cast<size_t>(y + oy) * width + (x + ox);
      const double nw = in_w[n];
      if (nw > 0) {
        sum += in_id[n];
        num += nw;
        numn += 1;
      }
    }
    if (numn > 0) {
      id = sum / numn;
      w = num / numn;
    }
  }
  L.out_id[l][c] = id;
  L.out_w[l][c] = w;
}

/** calculateMeanSquareOpticalFlow — src/tracker/tracker/src/monocular_tracker.cpp:104-134 — for up to kMaxFlowTransforms
 *  relative poses in one pass over a depth-map level (the tracker asks for the flow of t_t_r and of t_t_r without its rotation
 *  on every frame, :474-479).  One thread per pixel, per-workgroup partial sums {sum, count} per transform; the second
 *  kernel adds the partials in index order (deterministic) and takes sqrt(sum / n). */
constexpr int kMaxFlowTransforms = 4;
struct FlowArgs {
  double M[kMaxFlowTransforms][12];  // reproject_ = K [R|t] K^-1
  double cx, cy, ifx, ify;
  int width, height, n_transforms;
};
constexpr int kFlowRows = 16;
__global__ void __launch_bounds__(256) opticalFlowPartialsKernel(const double *__restrict__ idsum, const double *__restrict__ wgt, FlowArgs a,
                                                                 double *__restrict__ partials) {
  __shared__ double lds[4 * 2 * kMaxFlowTransforms];
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  double sum[kMaxFlowTransforms], cnt[kMaxFlowTransforms];
#pragma unroll
  for (int t = 0; t < kMaxFlowTransforms; ++t) sum[t] = cnt[t] = 0;
  // a workgroup takes kFlowRows rows of its 256-column strip (rows in order, one running sum per thread): 16 x fewer partial
  // blocks for the closing kernel to add up
  for (int ry = 0; ry < kFlowRows; ++ry) {
  const int y = blockIdx.y * kFlowRows + ry;
  if (x >= 4 && x < a.width - 4 && y >= 4 && y < a.height - 4) {
    const size_t c = static_cast<size_t>(y) * a.width + x;
    const double w = wgt[c];
    if (w > 0) {
      const double idepth = idsum[c] / w;
      if (!(idepth < 1e-6)) {
        const double u = x, v = y, W = a.width, H = a.height;
        const double ax = (u - a.cx) * a.ifx, ay = (v - a.cy) * a.ify;  // PinholeCamera::unproject, pinhole_camera.hpp:129-141
#pragma unroll
        for (int t = 0; t < kMaxFlowTransforms; ++t) {
          if (t >= a.n_transforms) break;
          const double *M = a.M[t];
          const double px = M[0] * u + M[1] * v + (M[2] + M[3] * idepth);
          const double py = M[4] * u + M[5] * v + (M[6] + M[7] * idepth);
          const double pz = M[8] * u + M[9] * v + (M[10] + M[11] * idepth);
          const double tu = px / pz, tv = py / pz;
          if (validIdepth(idepth) && insideROI(u, v, W, H) && (pz > 0) && insideROI(tu, tv, W, H)) {  // camera_reproject.hpp:270-293
            const double bx = (tu - a.cx) * a.ifx, by = (tv - a.cy) * a.ify;
            sum[t] += (ax - bx) * (ax - bx) + (ay - by) * (ay - by);
            cnt[t] += 1;
          }
        }
      }
    }
  }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < kMaxFlowTransforms; ++t) {
    double s = sum[t], n = cnt[t];
    for (int off = 32; off > 0; off >>= 1) {
      s += __shfl_down(s, off);
      n += __shfl_down(n, off);
    }
    if (lane == 0) {
      lds[(wave * kMaxFlowTransforms + t) * 2] = s;
      lds[(wave * kMaxFlowTransforms + t) * 2 + 1] = n;
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * kMaxFlowTransforms) {
    double s = 0;
    for (int wv = 0; wv < 4; ++wv) s += lds[wv * 2 * kMaxFlowTransforms + threadIdx.x];
    partials[(static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * 2 * kMaxFlowTransforms + threadIdx.x] = s;
  }
}
/** closing sum over the partial blocks, fixed order: thread j adds the blocks j, j + 256, ... (all loads of a thread in flight
 *  together), then wave sums by shuffles and the four wave sums in wave order.  (One wave walking 5 120 single-row blocks
 *  with a load per step made this the longest part of the call: 0.2 ms for a 1280 x 1024 map.) */
__global__ void __launch_bounds__(256) opticalFlowFinishKernel(const double *__restrict__ partials, int n_blocks, int n_transforms, double *__restrict__ out) {
  __shared__ double lds[4][2 * kMaxFlowTransforms];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double s[kMaxFlowTransforms], n[kMaxFlowTransforms];
#pragma unroll
  for (int t = 0; t < kMaxFlowTransforms; ++t) s[t] = n[t] = 0;
  for (int b = tid; b < n_blocks; b += 256) {
#pragma unroll
    for (int t = 0; t < kMaxFlowTransforms; ++t) {
      s[t] += partials[(static_cast<size_t>(b) * kMaxFlowTransforms + t) * 2];
      n[t] += partials[(static_cast<size_t>(b) * kMaxFlowTransforms + t) * 2 + 1];
    }
  }
#pragma unroll
  for (int t = 0; t < kMaxFlowTransforms; ++t) {
    double ss = s[t], nn = n[t];
    for (int off = 32; off > 0; off >>= 1) {
      ss += __shfl_down(ss, off);
      nn += __shfl_down(nn, off);
    }
    if (lane == 0) {
      lds[wave][2 * t] = ss;
      lds[wave][2 * t + 1] = nn;
    }
  }
  __syncthreads();
  if (tid < n_transforms) {
    double ss = 0, nn = 0;
    for (int w = 0; w < 4; ++w) {
      ss += lds[w][2 * tid];
      nn += lds[w][2 * tid + 1];
    }
    out[tid] = sqrt(ss / nn);  // 0 / 0 = NaN for an empty map, as in the reference
  }
}


/** The same measure over the level's REFERENCE POINTS when the tracker has extracted them (depth_maps.hpp: LevelPoints — the pixels inside the
 *  4-px border with positive weight and idepth >= 1e-6, in row-major order: exactly the pixels the dense pass above keeps).  A level-0 map of
 *  1280 x 1024 holds ~12 000 of them among 1.3 M pixels: one thread per point, per-workgroup partial sums, and the workgroup that takes the
 *  last ticket adds the partials in index order (fixed summation order whatever the arrival order) — ONE launch instead of a 21 MB pass and
 *  a closing launch (19 + 6 us).  scratch: partials (2 kMaxFlowTransforms per workgroup); ticket: one unsigned the last workgroup re-arms. */
constexpr int kFlowPointThreads = 256;
__global__ void __launch_bounds__(kFlowPointThreads) opticalFlowPointsKernel(const double *__restrict__ pu, const double *__restrict__ pv,
                                                                            const double *__restrict__ pid, int n, FlowArgs a, double *partials,
                                                                            unsigned *ticket, double *__restrict__ out) {
  __shared__ double lds[kFlowPointThreads / 64][2 * kMaxFlowTransforms];
  __shared__ unsigned s_last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double sum[kMaxFlowTransforms], cnt[kMaxFlowTransforms];
#pragma unroll
  for (int t = 0; t < kMaxFlowTransforms; ++t) sum[t] = cnt[t] = 0;
  const double W = a.width, H = a.height;
  const int i = static_cast<int>(blockIdx.x) * kFlowPointThreads + tid;
  if (i < n) {
    const double u = pu[i], v = pv[i], idepth = pid[i];
    const double ax = (u - a.cx) * a.ifx, ay = (v - a.cy) * a.ify;
#pragma unroll
    for (int t = 0; t < kMaxFlowTransforms; ++t) {
      if (t >= a.n_transforms) break;
      const double *M = a.M[t];
      const double px = M[0] * u + M[1] * v + (M[2] + M[3] * idepth);
      const double py = M[4] * u + M[5] * v + (M[6] + M[7] * idepth);
      const double pz = M[8] * u + M[9] * v + (M[10] + M[11] * idepth);
      const double tu = px / pz, tv = py / pz;
      if (validIdepth(idepth) && insideROI(u, v, W, H) && (pz > 0) && insideROI(tu, tv, W, H)) {
        const double bx = (tu - a.cx) * a.ifx, by = (tv - a.cy) * a.ify;
        sum[t] = (ax - bx) * (ax - bx) + (ay - by) * (ay - by);
        cnt[t] = 1;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < kMaxFlowTransforms; ++t) {
    double s = sum[t], c = cnt[t];
    for (int off = 32; off > 0; off >>= 1) {
      s += __shfl_down(s, off);
      c += __shfl_down(c, off);
    }
    if (lane == 0) {
      lds[wave][2 * t] = s;
      lds[wave][2 * t + 1] = c;
    }
  }
  __syncthreads();
  if (tid < 2 * kMaxFlowTransforms) {
    double s = 0;
    for (int w = 0; w < kFlowPointThreads / 64; ++w) s += lds[w][tid];
    __hip_atomic_store(&partials[static_cast<size_t>(blockIdx.x) * 2 * kMaxFlowTransforms + tid], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (tid == 0) {
    __atomic_thread_fence(__ATOMIC_RELEASE);  // (device scope: the partials above are visible before the ticket)
    s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;
  // the last workgroup: all partials are in memory; thread j adds workgroups j, j + 256, ... in order, then shuffles and wave order
  double s[kMaxFlowTransforms], c[kMaxFlowTransforms];
#pragma unroll
  for (int t = 0; t < kMaxFlowTransforms; ++t) s[t] = c[t] = 0;
  for (unsigned b = tid; b < gridDim.x; b += kFlowPointThreads) {
#pragma unroll
    for (int t = 0; t < kMaxFlowTransforms; ++t) {
      s[t] += __hip_atomic_load(&partials[(static_cast<size_t>(b) * kMaxFlowTransforms + t) * 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      c[t] += __hip_atomic_load(&partials[(static_cast<size_t>(b) * kMaxFlowTransforms + t) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();  // (lds is reused)
#pragma unroll
  for (int t = 0; t < kMaxFlowTransforms; ++t) {
    double ss = s[t], cc = c[t];
    for (int off = 32; off > 0; off >>= 1) {
      ss += __shfl_down(ss, off);
      cc += __shfl_down(cc, off);
    }
    if (lane == 0) {
      lds[wave][2 * t] = ss;
      lds[wave][2 * t + 1] = cc;
    }
  }
  __syncthreads();
  if (tid < a.n_transforms) {
    double ss = 0, cc = 0;
    for (int w = 0; w < kFlowPointThreads / 64; ++w) {
      ss += lds[w][2 * tid];
      cc += lds[w][2 * tid + 1];
    }
    out[tid] = sqrt(ss / cc);  // 0 / 0 = NaN for an empty map, as in the reference
  }
  if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // armed for the next launch
}

}  // namespace dsopp_hip
