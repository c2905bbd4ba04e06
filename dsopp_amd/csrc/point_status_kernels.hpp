// updatePointStatuses on the device (PROB_SRC/photometric_bundle_adjustment.cpp:321-406) and relinearizeSystem (:310-316).
// The 3rd-quartile threshold is an exact order statistic: a 5-pass most-significant-digit radix select (13-bit digits) over the IEEE-754
// bit patterns of the (non-negative) residual energies, then one pass per landmark applies it.
#pragma once
#include <hip/hip_runtime.h>

#include "pba_kernels.hpp"
#include "pba_types.hpp"
#include "se3_math.hpp"

namespace dsopp_hip {

// Digit width of the radix select: 13 bits -> 5 passes over the 64-bit keys (8-bit digits were 8 launches of ~4.5 us each with as
// much host enqueue time between them; the histogram of a pass is 8192 words: 32 KB of LDS per workgroup, 96 KB of state in HBM).
constexpr int kSelectBits = 13;
constexpr int kSelectBins = 1 << kSelectBits;
constexpr int kSelectPasses = (64 + kSelectBits - 1) / kSelectBits;  // 5; the top digit holds the 12 bits 52 .. 63

struct SelectState {
  // the state after the scan of pass k lives in slot k & 1: a histogram kernel reads one slot and writes the other, so a
  // workgroup that starts late never sees a half-advanced state
  unsigned long long prefix[2];  // bits fixed so far (from the most significant digit down)
  unsigned long long mask[2];    // which bits of `prefix` are fixed
  unsigned int rank[2];          // rank of the wanted element among the keys matching the prefix
  unsigned int n_ok[2];
  double threshold;           // result: selected energy + sigma^2 / 2 (0 when there is no kOk residual)
  double pad_;                // (the histograms start at byte 64: 16-byte vector loads)
  // three histograms in rotation: the kernel of pass p accumulates into hist[p % 3], reads the finished hist[(p + 1) % 3] of the
  // previous pass (to advance prefix / rank itself: no separate scan launch) and clears hist[(p + 2) % 3] for the next one
  unsigned int hist[3][kSelectBins];
};
static_assert(offsetof(SelectState, hist) % 16 == 0, "histogram alignment");

__global__ void selectInitKernel(SelectState *s) {
  if (threadIdx.x < 2) {
    s->prefix[threadIdx.x] = 0;
    s->mask[threadIdx.x] = 0;
    s->rank[threadIdx.x] = 0;
    s->n_ok[threadIdx.x] = 0;
    s->threshold = 0;
  }
  for (int k = threadIdx.x; k < 3 * kSelectBins; k += blockDim.x) (&s->hist[0][0])[k] = 0;
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

/** what the scan step of pass `done` does to the select state: picks the bucket of hist that holds the wanted rank and
 *  narrows prefix / mask / rank (the top pass also fixes n_ok and the third-quartile rank, :358).  Every workgroup of the next
 *  pass computes it for itself from the stored state + the finished histogram. */
struct SelectLocal {
  unsigned long long prefix, mask;
  unsigned int rank, n_ok;
};
/** cooperative: called by ALL THREADS threads of a workgroup; `scan` is THREADS words of LDS.  A thread owns kSelectBins / THREADS
 *  consecutive bins: their sum enters a workgroup scan, the thread whose range holds the rank walks its own bins.  (A single lane
 *  walking the histogram in global memory paid one dependent round trip per bin: 10 - 15 us per pass.) */
template <int THREADS>
__device__ inline SelectLocal selectAdvance(const SelectState *s, int done, unsigned int *scan) {
  constexpr int PER = kSelectBins / THREADS;
  static_assert(PER * THREADS == kSelectBins && PER % 4 == 0, "bins per thread");
  const int t = threadIdx.x;
  const int in = (done + 1) & 1;  // S_{done + 1}; for the top pass that is the initial all-zero state
  SelectLocal l{s->prefix[in], s->mask[in], s->rank[in], s->n_ok[in]};
  unsigned int mine[PER];
  unsigned int sum = 0;
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(&s->hist[done % 3][t * PER]);
#pragma unroll
    for (int k = 0; k < PER / 4; ++k) {
      const uint4 v = src[k];
      mine[4 * k + 0] = v.x;
      mine[4 * k + 1] = v.y;
      mine[4 * k + 2] = v.z;
      mine[4 * k + 3] = v.w;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) sum += mine[k];
  }
  scan[t] = sum;
  __syncthreads();
  for (int off = 1; off < THREADS; off <<= 1) {  // inclusive scan of the per-thread sums
    const unsigned int v = t >= off ? scan[t - off] : 0u;
    __syncthreads();
    scan[t] += v;
    __syncthreads();
  }
  const unsigned int incl = scan[t], excl = incl - sum;
  if (done == kSelectPasses - 1) {
    l = SelectLocal{0, 0, 0, 0};
    const unsigned int n = scan[THREADS - 1];
    l.n_ok = n;
    l.rank = static_cast<unsigned int>(static_cast<double>(n) * 0.75);  // third_quartile index, :358
    if (n == 0) {
      l.mask = ~0ull;  // nothing matches any more
      l.prefix = 1;
    }
  }
  __shared__ unsigned int s_bucket, s_cum;
  if (t == 0) {
    s_bucket = kSelectBins - 1;
    s_cum = 0;
  }
  __syncthreads();
  if (l.n_ok > 0 && l.rank >= excl && l.rank < incl) {
    // the first bin whose inclusive count exceeds the rank lies in this thread's range
    unsigned int c = excl;
    int found = PER - 1;
    unsigned int cum = excl;
    bool done_walk = false;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (!done_walk && l.rank < c + mine[k]) {
        found = k;
        cum = c;
        done_walk = true;
      }
      c += mine[k];
    }
    s_bucket = static_cast<unsigned int>(t * PER + found);
    s_cum = cum;
  }
  __syncthreads();
  if (l.n_ok > 0) {
    l.rank -= s_cum;
    l.prefix |= static_cast<unsigned long long>(s_bucket) << (kSelectBits * done);
    l.mask |= static_cast<unsigned long long>(kSelectBins - 1) << (kSelectBits * done);
  }
  __syncthreads();
  return l;
}

/** one histogram pass of the radix select; digit index `pass` (kSelectPasses - 1 = most significant).  Below the top pass the
 *  workgroup first advances the select state by the previous pass's histogram (workgroup 0 also stores it and clears the histogram
 *  of the next pass): one launch per digit. */
constexpr int kSelectThreads = 1024;
__global__ void __launch_bounds__(kSelectThreads) selectHistKernel(const FrameDev *__restrict__ frames, const SweepBlock *__restrict__ table, int n_entries,
                                                                  SelectState *s, int pass) {
  __shared__ unsigned int lh[kSelectBins];  // workgroup-private histogram: one global atomic per non-empty bin and workgroup
  __shared__ unsigned int scan[kSelectThreads];
  const SelectLocal l = pass == kSelectPasses - 1 ? SelectLocal{0, 0, 0, 0} : selectAdvance<kSelectThreads>(s, pass + 1, scan);
  for (int k = threadIdx.x; k < kSelectBins; k += kSelectThreads) lh[k] = 0;
  __syncthreads();
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0 && pass < kSelectPasses - 1) {  // S_{pass + 1}
      const int out = (pass + 1) & 1;
      s->prefix[out] = l.prefix;
      s->mask[out] = l.mask;
      s->rank[out] = l.rank;
      s->n_ok[out] = l.n_ok;
    }
    for (int k = threadIdx.x; k < kSelectBins; k += kSelectThreads) s->hist[(pass + 2) % 3][k] = 0;
  }
  const int entry = blockIdx.x * (kSelectThreads / kItemsPerBlock) + (threadIdx.x / kItemsPerBlock);
  const bool in_range = entry < n_entries;
  const SweepBlock be = table[in_range ? entry : 0];
  const int i = be.offset + threadIdx.x % kItemsPerBlock;
  unsigned long long key = 0;
  const bool counts = in_range && eligibleEnergy(be, frames, i, key) && (key & l.mask) == l.prefix;
  // The upper bits of the energies are nearly constant (same sign / exponent range), so almost every key of a pass lands in
  // ONE bucket: per-lane atomics on one address serialise (12 000 of them took 109 us; one per wavefront from 190 wavefronts
  // still 50 us across the XCDs).  A wavefront adds the lanes sharing the bucket of its first counting lane with one LDS
  // atomic, the rest per lane, and the workgroup flushes its non-empty bins once.
  const unsigned int bucket = static_cast<unsigned int>((key >> (kSelectBits * pass)) & static_cast<unsigned long long>(kSelectBins - 1));
  const unsigned long long active = __ballot(counts);
  if (active != 0) {
    const int leader = __ffsll(static_cast<long long>(active)) - 1;
    const unsigned int common = static_cast<unsigned int>(__shfl(static_cast<int>(bucket), leader));
    const unsigned long long same = __ballot(counts && bucket == common);
    if (static_cast<int>(threadIdx.x & 63) == leader) atomicAdd(&lh[common], static_cast<unsigned int>(__popcll(same)));
    if (counts && bucket != common) atomicAdd(&lh[bucket], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < kSelectBins; k += kSelectThreads)
    if (lh[k]) atomicAdd(&s->hist[pass % 3][k], lh[k]);
}

/** closing step of the select (after the pass-0 histogram): the selected key is the threshold energy (:360) */
template <int THREADS>
__device__ inline void selectFinish(SelectState *s, double half_sigma_sq, unsigned int *scan) {
  const SelectLocal l = selectAdvance<THREADS>(s, 0, scan);
  if (threadIdx.x == 0) {
    s->prefix[0] = l.prefix;
    s->mask[0] = l.mask;
    s->rank[0] = l.rank;
    s->n_ok[0] = l.n_ok;
    s->threshold = l.n_ok > 0 ? __longlong_as_double(static_cast<long long>(l.prefix)) + half_sigma_sq : 0.0;
  }
  // leave the state as the next select needs it: only the top pass's histogram has to be zero when that pass starts (every other
  // one is cleared by the pass before it), and it still holds this select's second-last pass — no initialisation launch per select
  for (int k = threadIdx.x; k < kSelectBins; k += THREADS) s->hist[(kSelectPasses - 1) % 3][k] = 0;
}

/** current camera-centre distances between all frame pairs: |t_r - t_t| of T = T0 exp(eps) (:380-382) */
__global__ void pairDistanceKernel(const WindowState *st, int F, double *dist /* [kMaxFrames][kMaxFrames] */, SelectState *select,
                                   double half_sigma_sq) {
  __shared__ double c[kMaxFrames][3];
  const int f = threadIdx.x;
  __shared__ unsigned int scan[256];
  if (select) selectFinish<256>(select, half_sigma_sq, scan);  // the single-workgroup step between select and apply (256 threads)
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
  const int *to_internal;  // nullable: caller index -> device index of the frame's landmarks (the export is in the caller's order)
  int n, n_targets;
  double *out_d;   // 4 n doubles
  uint8_t *out_b;  // (1 + n_targets) n bytes
};
__device__ __forceinline__ void exportLandmark(const FrameExportArgs &a, int i) {
  const int p = a.to_internal ? a.to_internal[i] : i;
  a.out_d[i] = a.idepth[p];
  a.out_d[a.n + i] = a.inv_hdd[p];
  a.out_d[2 * a.n + i] = a.relative_baseline[p];
  a.out_d[3 * a.n + i] = static_cast<double>(a.n_inliers[p]);
  a.out_b[i] = a.flags[p];
  for (int t = 0; t < a.n_targets; ++t) a.out_b[static_cast<size_t>(1 + t) * a.n + i] = a.status[t] ? a.status[t][p] : DSOPP_HIP_STATUS_UNKNOWN;
}
__global__ void exportFrameKernel(FrameExportArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  exportLandmark(a, i);
}

/** all keyframes of the window in one launch (blockIdx.y = entry): seven 3 us kernels were bound by the host's enqueue rate */
struct FrameExportBatch {
  FrameExportArgs f[kMaxFrames];
  int n_frames;
};
static_assert(sizeof(FrameExportBatch) <= 4096, "kernel argument block");
__global__ void exportFramesKernel(FrameExportBatch b) {
  const FrameExportArgs &a = b.f[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  exportLandmark(a, i);
}

/** relinearizeSystem — :310-316: the newest frame's linearisation point moves to its current estimate */
__device__ inline void relinearizeFrame(WindowState *st, int f) {
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

/** relinearizeSystem and, in the same launch, the pair constants at the new linearisation point (the covariance linearisation of
 *  solve() — or the next solve — needs them next: one launch instead of two).  One workgroup of kMaxFrames^2 threads. */
__global__ void __launch_bounds__(kMaxFrames *kMaxFrames) relinearizeKernel(const FrameDev *frames, WindowState *st, PairConst *pc, int F, int fej, int f) {
  if (threadIdx.x == 0) relinearizeFrame(st, f);
  __syncthreads();  // (one workgroup, one L1: the new state is visible to its other waves)
  const int idx = threadIdx.x;
  if (idx >= F * F) return;
  computePairConst(frames, st, pc, idx / F, idx % F, F, fej != 0);
}

}  // namespace dsopp_hip
