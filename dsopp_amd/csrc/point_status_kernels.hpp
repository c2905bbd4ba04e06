// updatePointStatuses on the device (PROB_SRC/photometric_bundle_adjustment.cpp:321-406) and relinearizeSystem (:310-316).
// The 3rd-quartile threshold is an exact order statistic: an 8-pass most-significant-byte radix select over the IEEE-754
// bit patterns of the (non-negative) residual energies, then one pass per landmark applies it.
#pragma once
#include <hip/hip_runtime.h>

#include "pba_kernels.hpp"
#include "pba_types.hpp"
#include "se3_math.hpp"

namespace dsopp_hip {

struct SelectState {
  unsigned long long prefix;  // bits fixed so far (from the most significant byte down)
  unsigned long long mask;    // which bits of `prefix` are fixed
  unsigned int rank;          // rank of the wanted element among the keys matching the prefix
  unsigned int n_ok;
  double threshold;           // result: selected energy + sigma^2 / 2 (0 when there is no kOk residual)
  unsigned int hist[256];
};

__global__ void selectInitKernel(SelectState *s) {
  if (threadIdx.x == 0) {
    s->prefix = 0;
    s->mask = 0;
    s->rank = 0;
    s->n_ok = 0;
    s->threshold = 0;
  }
  s->hist[threadIdx.x] = 0;
}

/** eligible = residual status kOk of a non-marginalised landmark towards a non-marginalised target frame (:340-352) */
__device__ inline bool eligibleEnergy(const SweepBlock &be, const FrameDev *frames, int i, unsigned long long &key) {
  if (i >= be.n_res) return false;
  if (frames[be.t].is_marginalized) return false;
  if (be.flags[i] & kFlagMarginalized) return false;
  if (be.status[i] != DSOPP_HIP_STATUS_OK) return false;
  key = static_cast<unsigned long long>(__double_as_longlong(be.energy[i]));
  return true;
}

/** one histogram pass of the radix select; byte index `pass` (7 = most significant) */
__global__ void __launch_bounds__(64) selectHistKernel(const FrameDev *__restrict__ frames, const SweepBlock *__restrict__ table, int n_entries,
                                                       SelectState *s, int pass) {
  const int entry = blockIdx.x * (64 / kItemsPerBlock) + (threadIdx.x / kItemsPerBlock);
  if (entry >= n_entries) return;
  const SweepBlock be = table[entry];
  const int i = be.offset + threadIdx.x % kItemsPerBlock;
  unsigned long long key;
  if (!eligibleEnergy(be, frames, i, key)) return;
  if ((key & s->mask) != s->prefix) return;
  atomicAdd(&s->hist[(key >> (8 * pass)) & 0xFFull], 1u);
}

/** picks the bucket that holds the wanted rank, narrows the prefix, clears the histogram (one wave) */
__global__ void selectScanKernel(SelectState *s, int pass, double half_sigma_sq) {
  __shared__ unsigned int h[256];
  const int t = threadIdx.x;
  for (int k = t; k < 256; k += 64) h[k] = s->hist[k];
  __syncthreads();
  if (t == 0) {
    unsigned int rank = s->rank;
    if (pass == 7) {
      unsigned int n = 0;
      for (int k = 0; k < 256; ++k) n += h[k];
      s->n_ok = n;
      rank = static_cast<unsigned int>(static_cast<double>(n) * 0.75);  // third_quartile index, :358
      if (n == 0) {
        s->threshold = 0;
        s->mask = ~0ull;     // nothing matches any more
        s->prefix = 1;
      }
    }
    if (s->n_ok > 0) {
      unsigned int cum = 0;
      int bucket = 255;
      for (int k = 0; k < 256; ++k) {
        if (rank < cum + h[k]) {
          bucket = k;
          break;
        }
        cum += h[k];
      }
      s->rank = rank - cum;
      s->prefix |= static_cast<unsigned long long>(bucket) << (8 * pass);
      s->mask |= 0xFFull << (8 * pass);
      if (pass == 0) s->threshold = __longlong_as_double(static_cast<long long>(s->prefix)) + half_sigma_sq;  // :360
    }
  }
  __syncthreads();
  for (int k = t; k < 256; k += 64) s->hist[k] = 0;
}

/** current camera-centre distances between all frame pairs: |t_r - t_t| of T = T0 exp(eps) (:380-382) */
__global__ void pairDistanceKernel(const WindowState *st, int F, double *dist /* [kMaxFrames][kMaxFrames] */) {
  __shared__ double c[kMaxFrames][3];
  const int f = threadIdx.x;
  if (f < F) {
    Rigid T0;
#pragma unroll
    for (int i = 0; i < 9; ++i) T0.R[i] = st->T0_R[f][i];
#pragma unroll
    for (int i = 0; i < 3; ++i) T0.t[i] = st->T0_t[f][i];
    double xi[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) xi[i] = st->eps[f][i];
    const Rigid T = rigidMul(T0, rigidExp(xi));
    c[f][0] = T.t[0];
    c[f][1] = T.t[1];
    c[f][2] = T.t[2];
  }
  __syncthreads();
  for (int p = threadIdx.x; p < F * F; p += blockDim.x) {
    const int r = p / F, t = p % F;
    const double dx = c[r][0] - c[t][0], dy = c[r][1] - c[t][1], dz = c[r][2] - c[t][2];
    dist[r * kMaxFrames + t] = sqrt(dx * dx + dy * dy + dz * dz);
  }
}

/** the per-landmark pass (:362-405): residuals above the threshold become kOutlier (energy 0), inlier counts and relative
 *  baselines are refreshed, landmarks without a valid reprojection become outliers.  One thread per landmark. */
__global__ void __launch_bounds__(kSchurLandmarks) applyPointStatusesKernel(const FrameDev *__restrict__ frames, const SchurBlock *__restrict__ table, int F,
                                                                            const SelectState *s, const double *__restrict__ dist) {
  const SchurBlock &be = table[blockIdx.x];
  const FrameDev &fr = frames[be.r];
  const int i = be.offset + threadIdx.x;
  if (i >= be.n) return;
  uint8_t flg = be.flags[i];
  if (flg & kFlagMarginalized) return;
  const double threshold = s->threshold;
  const double idepth = be.idepth[i];
  double bl = fr.relative_baseline[i];
  int valid = 0;
  for (int t = 0; t < F; ++t) {
    if (t == be.r || be.status[t] == nullptr || i >= be.n_res[t] || frames[t].is_marginalized) continue;
    uint8_t st = be.status[t][i];
    if (fr.energy[t][i] > threshold) {  // residual = {kOutlier}: a fresh ResidualPoint, energy 0 (:366-368)
      st = DSOPP_HIP_STATUS_OUTLIER;
      be.status[t][i] = st;
      be.cand[t][i] = st;
      fr.energy[t][i] = 0;
    }
    if (st == DSOPP_HIP_STATUS_OK) {
      bl = fmax(bl, idepth * dist[be.r * kMaxFrames + t]);
      ++valid;
    }
  }
  fr.relative_baseline[i] = bl;
  fr.n_inliers[i] = valid;
  if (valid < 1) be.flags[i] = flg | kFlagOutlier;  // minimum_valid_reprojections_num = 1 (:395-399)
}

/** what updateFrame reads back for one keyframe (PROB_SRC/photometric_bundle_adjustment.cpp:182-264), packed into one
 *  contiguous buffer so that it costs one transfer: idepth n | H_dd^-1 n | relative baseline n | inlier counts n (as
 *  doubles) | then bytes: flags n | statuses of target 0 n | target 1 n | ...  (targets = the frame's connections in the
 *  order given by the caller) */
struct FrameExportArgs {
  const double *idepth, *inv_hdd, *relative_baseline;
  const int32_t *n_inliers;
  const uint8_t *flags;
  const uint8_t *status[kMaxFrames];
  int n, n_targets;
  double *out_d;   // 4 n doubles
  uint8_t *out_b;  // (1 + n_targets) n bytes
};
__global__ void exportFrameKernel(FrameExportArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  a.out_d[i] = a.idepth[i];
  a.out_d[a.n + i] = a.inv_hdd[i];
  a.out_d[2 * a.n + i] = a.relative_baseline[i];
  a.out_d[3 * a.n + i] = static_cast<double>(a.n_inliers[i]);
  a.out_b[i] = a.flags[i];
  for (int t = 0; t < a.n_targets; ++t) a.out_b[static_cast<size_t>(1 + t) * a.n + i] = a.status[t] ? a.status[t][i] : DSOPP_HIP_STATUS_UNKNOWN;
}

/** relinearizeSystem — :310-316: the newest frame's linearisation point moves to its current estimate */
__global__ void relinearizeKernel(WindowState *st, int f) {
  if (threadIdx.x != 0) return;
  Rigid T0;
  for (int i = 0; i < 9; ++i) T0.R[i] = st->T0_R[f][i];
  for (int i = 0; i < 3; ++i) T0.t[i] = st->T0_t[f][i];
  double xi[6];
  for (int i = 0; i < 6; ++i) xi[i] = st->eps[f][i];
  Rigid T = rigidMul(T0, rigidExp(xi));
  rigidNormalize(T);
  for (int i = 0; i < 9; ++i) st->T0_R[f][i] = T.R[i];
  for (int i = 0; i < 3; ++i) st->T0_t[f][i] = T.t[i];
  st->ab0[f][0] += st->eps[f][6];
  st->ab0[f][1] += st->eps[f][7];
  for (int a = 0; a < kBlk; ++a) st->eps[f][a] = 0;
}

}  // namespace dsopp_hip
