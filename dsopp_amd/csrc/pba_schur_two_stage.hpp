// Two-stage, order-deterministic build of the combined system (pba_solve_kernels.hpp: ReduceSchurArgs::comb) for large windows.
//
// reduceSchurKernel accumulates H_schur with fp64 atomics: every 64-landmark workgroup adds its K x K contribution to the same
// ~K^2/2 addresses.  At 7 frames / 2000 landmarks that is 35 workgroups and the cheapest way to sum; at 12 frames / 50 000
// landmarks it is 782 workgroups x 4.7 k atomics onto 4.7 k addresses (114 us per iteration, the second largest kernel of that
// window) and the result depends on the arrival order.  Here:
//   stage 1  schurTwoStageKernel: a workgroup takes SEVERAL 64-landmark chunks, keeps its MFMA tiles in registers across them and
//            writes ONE partial system, without atomics; the frame-pair workgroups write their derived blocks to per-pair slots;
//   stage 2  combineSystemKernel: one thread per entry of the combined system sums the partial systems and the pair blocks in a
//            fixed order, applies the damping and writes the entry once.
// Same arithmetic per landmark as reduceSchurKernel (hessian_block_evaluation.hpp:96-164,169-236).  Inside the fused loop the launch
// carries kScalarGroups more workgroups that pre-sum the sweep's energy scalars, and the LM decision taken from them is the prologue
// of the solve launch behind stage 2 (pba_solve_combined.hpp).  Selected when a window has more 64-landmark chunks than
// kTwoStageMinChunks, or always with dsopp_hip_window_set_deterministic.
#pragma once
#include "pba_solve_kernels.hpp"

namespace dsopp_hip {

constexpr int kPairOut = 208;  // per ordered frame pair: T^T G T [64] | G T [64] | T^T q [8] | G [64] | q [8]
constexpr int kTwoStageMinChunks = 192;  // 7 frames: 59.4 (atomics) against 67.2 us at 125 chunks, equal at 188, 96.6 against 90.5 at 313 (scripts/threshold_sweep.py)
constexpr int kMaxTilesPerWave = 5;  // K <= 128: 36 upper-triangular 16 x 16 tiles over 8 waves

/** doubles of one partial Schur system: the upper-triangular 16 x 16 tiles in MFMA layout + b_schur */
__host__ __device__ inline int twoStagePartialCount(int F) {
  const int K = kBlk * F, nt = ((K + 15) & ~15) >> 4;
  return nt * (nt + 1) / 2 * 256 + K;
}

struct TwoStageArgs {
  const FrameDev *frames;
  const PairConst *pc;
  const SchurBlock *schur_table;  // one entry per 64-landmark chunk
  const double *partials;         // the sweep's per-workgroup sums (G, q per pair block)
  const int *pair_first_block, *pair_num_blocks;
  const LmControl *ctrl;          // nullable; the launch is a no-op when the loop has ended
  double *schur_partials;         // [n_schur_wgs][twoStagePartialCount(F)]: tiles in MFMA layout | b
  double *pair_out;               // [kMaxFrames * kMaxFrames][kPairOut]
  int F;
  int n_chunks, chunks_per_wg, n_schur_wgs;
  int ublk_parity;
  long long *dbg;  // nullable tuning aid (-DDSOPP_HIP_STAMPS): phase stamps of workgroup 1, first chunk
  // fused loop: kScalarGroups further workgroups sum the sweep's four energy scalars into fixed groups
  // (group_sums[g][4]) for the decision, which is the prologue of the SOLVE launch — nothing in this launch waits for it, and the
  // separate scalar-groups launch in front of it is gone
  double *group_sums = nullptr;
  int n_sweep_blocks = 0;
};
#define TS_STAMP(i) do { if (kStamps && a.dbg && threadIdx.x == 0 && blockIdx.x == 1 && chunk == first_chunk) a.dbg[i] = wall_clock64(); } while (0)

/** tiles a wave accumulates: ceil(upper-triangular tiles / 8 waves) */
__host__ __device__ inline int twoStageTilesPerWave(int F) {
  const int nt = ((kBlk * F + 15) & ~15) >> 4;
  return (nt * (nt + 1) / 2 + kSchurThreads / 64 - 1) / (kSchurThreads / 64);
}

/** LDS row stride of the two-stage kernel: a compile-time constant per tiles-per-wave class (every stride is >= Kp and = 16 mod 32,
 *  the conflict-free condition of schurRowStride).  With a run-time stride every one of the 16 x 3 operand reads of a tile needs its
 *  own address register (the compiler hoists them out of the chunk loop and spills them); with a constant they are immediate offsets. */
__host__ __device__ constexpr int twoStageRowStride(int tpw) { return tpw <= 2 ? 80 : (tpw <= 4 ? 112 : 144); }
inline size_t twoStageSmemBytes(int F) {
  return (static_cast<size_t>(kSchurLandmarks) * twoStageRowStride(twoStageTilesPerWave(F)) + 2 * kSchurLandmarks + 40 * static_cast<size_t>(kMaxFrames)) * sizeof(double);
}

// TPW = tiles per wave, a template argument so that the accumulators are a fixed set of registers (with the generic bound of 5 tiles
// the 12-frame window carried 16 unused accumulator registers and a select chain per tile: 23 spilled VGPRs under the 128-register
// cap that two resident workgroups per compute unit need)
template <int TPW>
__global__ void __launch_bounds__(kSchurThreads, 4) schurTwoStageKernel(TwoStageArgs a) {  // 2 workgroups of 8 waves per compute unit
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (a.ctrl && !a.ctrl->active) return;
  const int F = a.F, K = kBlk * F;
  if (static_cast<int>(blockIdx.x) >= a.n_schur_wgs + F * F) {
    // ---- scalar group: energy, n_valid, |idepth step|^2, idepth . step of a fixed share of the sweep's workgroups (fixed order)
    const int g = blockIdx.x - a.n_schur_wgs - F * F;
    const int per = (a.n_sweep_blocks + kScalarGroups - 1) / kScalarGroups;
    const int b0 = g * per, b1 = min(b0 + per, a.n_sweep_blocks);
    double v[4] = {0, 0, 0, 0};
    for (int b = b0 + threadIdx.x; b < b1; b += kSchurThreads) {
      const double *p = a.partials + static_cast<size_t>(b) * kPartial;
      v[0] += p[44];
      v[1] += p[45];
      v[2] += p[46];
      v[3] += p[47];
    }
    blockSum<4, kSchurThreads>(v, reinterpret_cast<double *>(smem_raw));
    if (threadIdx.x == 0) {
      a.group_sums[4 * g + 0] = v[0];
      a.group_sums[4 * g + 1] = v[1];
      a.group_sums[4 * g + 2] = v[2];
      a.group_sums[4 * g + 3] = v[3];
    }
    return;
  }
  if (static_cast<int>(blockIdx.x) >= a.n_schur_wgs) {
    // ---- frame pair: deterministic sum of the sweep's partials, derived blocks, one slot per pair (no atomics)
    const int p = blockIdx.x - a.n_schur_wgs;
    const int r = p / F, t = p % F, pi = r * kMaxFrames + t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const PairConst &P = a.pc[pi];
    const int valid = P.valid;
    double *lds = reinterpret_cast<double *>(smem_raw);  // [48] G/q + [64] scratch + [kPairBlk] derived + [8][48] wave sums
    double *wsum = lds + 48 + 64 + kPairBlk;
    {
      double sw = 0;
      if (valid && lane < 44) {
        const int first = a.pair_first_block[pi], cnt = a.pair_num_blocks[pi];
        const double *src = a.partials + static_cast<size_t>(first) * kPartial + lane;
        int b = wave;
        for (; b + 8 < cnt; b += 16) sw += src[static_cast<size_t>(b) * kPartial] + src[static_cast<size_t>(b + 8) * kPartial];
        if (b < cnt) sw += src[static_cast<size_t>(b) * kPartial];
      }
      if (lane < 48) wsum[wave * 48 + lane] = sw;
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    if (!valid) {
      // no connection r -> t: the slot is zeroed, so that the ordered sums of stage 2 can add every slot unconditionally
      if (r != t) {
        double *out = a.pair_out + static_cast<size_t>(pi) * kPairOut;
        for (int k = lane; k < kPairOut; k += 64) out[k] = 0;
      }
      return;
    }
    double s = 0;
    if (lane < 44) {
#pragma unroll
      for (int w = 0; w < kSchurThreads / 64; ++w) s += wsum[w * 48 + lane];
    }
    if (lane < 48) lds[lane] = s;
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    double *drv = lds + 48 + 64;
    derivePairBlocks(lds, P, drv, lds + 48, lane);
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    double *out = a.pair_out + static_cast<size_t>(pi) * kPairOut;
    out[lane] = drv[lane];
    out[64 + lane] = drv[64 + lane];
    out[136 + lane] = lds[symIdx(lane >> 3, lane & 7)];
    if (lane < 8) {
      out[128 + lane] = drv[128 + lane];
      out[200 + lane] = lds[36 + lane];
    }
    return;
  }

  // ---- Schur workgroup: chunks [first, last) of 64 landmarks each, tiles kept in registers across them
  const int Kp = (K + 15) & ~15;
  constexpr int stride = twoStageRowStride(TPW);
  double *hrow = reinterpret_cast<double *>(smem_raw);  // [kSchurLandmarks][stride]
  double *wgt = hrow + kSchurLandmarks * stride;        // inv per landmark (0 = excluded)
  double *wbd = wgt + kSchurLandmarks;                  // inv * bd
  double *Tm = wbd + kSchurLandmarks;                   // [F][40]: Adj (36), s0 of the chunk's frame towards every target
  const bool bd_in_pad = K < Kp;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nt = Kp >> 4, n_tiles = nt * (nt + 1) / 2;
  f64x4 acc[TPW];
  // LDS offsets of this wave's tiles (row block ti, column block tj), decoded once; wave-uniform, kept in scalar registers
  int off_a[TPW], off_b[TPW];
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    acc[q] = f64x4{0, 0, 0, 0};
    const int tile = __builtin_amdgcn_readfirstlane(wave) + q * (kSchurThreads / 64);
    int ti = 0, rem = tile < n_tiles ? tile : 0;
    while (rem >= nt - ti) {
      rem -= nt - ti;
      ++ti;
    }
    off_a[q] = __builtin_amdgcn_readfirstlane(16 * ti);
    off_b[q] = __builtin_amdgcn_readfirstlane(16 * (ti + rem));
  }
  double bs_acc = 0;  // b_schur entry threadIdx.x when the tiles have no spare column
  // chunks are dealt out evenly: with chunks_per_wg = 0 workgroup b takes n / W chunks, the first n mod W workgroups one more
  int first_chunk, last_chunk;
  if (a.chunks_per_wg > 0) {
    first_chunk = blockIdx.x * a.chunks_per_wg;
    last_chunk = min(first_chunk + a.chunks_per_wg, a.n_chunks);
  } else {
    const int base = a.n_chunks / a.n_schur_wgs, rem = a.n_chunks - base * a.n_schur_wgs, b = blockIdx.x;
    first_chunk = b * base + min(b, rem);
    last_chunk = first_chunk + base + (b < rem ? 1 : 0);
  }
  int staged_r = -1;
  for (int chunk = first_chunk; chunk < last_chunk; ++chunk) {
    const SchurBlock &be = a.schur_table[chunk];
    const int r = be.r;
    // the thread index is opaque per chunk: what derives from it (row pointers, operand offsets) is recomputed with a few integer
    // instructions instead of being kept across the loop — next to 4 / 5 resident accumulators those loop invariants went to scratch
    int tid = static_cast<int>(threadIdx.x);
    asm volatile("" : "+v"(tid));
    const int l = tid >> 3, sub = tid & 7;
    const int lane_off = ((tid & 63) >> 4) * stride + (tid & 15);
    const int lk = (tid & 63) >> 4;
    const int i = be.offset + l;
    const unsigned conn = be.conn_mask & ~(1u << r);
    const size_t plane = ublkPlane(be.cap);
    const double *ubase = be.ublk + static_cast<size_t>(a.ublk_parity) * kMaxFrames * plane + static_cast<size_t>(i) * kUblk;
    TS_STAMP(0);
    __syncthreads();  // the previous chunk's tiles have been read
    // (no clearing pass over the 57 KB of rows: phase 1 writes every column below Kp of every row — values, or zeros where a
    // landmark has no residual towards a frame / is not taken — which saves 1.3 us and one barrier per chunk)
    if (r != staged_r) {  // (workgroup-uniform)
      for (int e = threadIdx.x; e < F * 37; e += kSchurThreads) {
        const int t = e / 37, c = e - 37 * t;
        const PairConst &P = a.pc[r * kMaxFrames + t];
        Tm[t * 40 + c] = c < 36 ? P.Adj[c] : P.s0;
      }
      staged_r = r;
      __syncthreads();
    }
    bool take = false;
    uint8_t flg = 0;
    if (i < be.n) {
      flg = be.flags[i];
      take = (flg & kFlagMarginalized) == 0;
    }
    TS_STAMP(1);
    {
      // phase 1 (as reduceSchurKernel): 8 threads per landmark, thread `sub` owns the targets t = sub, sub + 8, ...
      // (Measured and dropped: staging every slot's 64 rows as one contiguous 5 KB run with 16-byte loads by consecutive lanes,
      // finalisation from LDS — 16-26 us for the staging alone against 10 us for this whole phase; the register file cannot hold
      // the staged words next to the resident MFMA tiles at two workgroups per compute unit.)
      double hr[kBlk];
#pragma unroll
      for (int c = 0; c < kBlk; ++c) hr[c] = 0;
      double hdd = 0, bd = 0;
      double *row = hrow + l * stride;
      // (not unrolled: two iterations in flight — windows of more than 8 keyframes — hold 20 row words each next to the resident MFMA
      // accumulators, and the 4- / 5-tile variants spilled two of those accumulators around this phase)
#pragma unroll 1
      for (int t = sub; t < F; t += 8) {
        if (!(take && ((conn >> t) & 1u))) {
          if (t != r) {  // (the chunk's own frame block is written by sub 0 below)
#pragma unroll
            for (int c = 0; c < kBlk; ++c) row[kBlk * t + c] = 0;
          }
          continue;
        }
        {
          double ht[kUblk];
          const double *src = ubase + t * plane;
#pragma unroll
          for (int c = 0; c < kUblk; ++c) ht[c] = src[c];
          const double *Tt = Tm + t * 40;
#pragma unroll
          for (int c = 0; c < kBlk; ++c) row[kBlk * t + c] = ht[c];
          hdd += ht[8];
          bd += ht[9];
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) s += Tt[6 * k + c] * ht[k];
            hr[c] -= s;
            // (4 / 5 resident accumulators per wave: without this the 36 LDS reads of the adjoint are all in flight at once — 72
            // registers — and two accumulators went to scratch around this phase)
            if (TPW >= 4) __builtin_amdgcn_sched_barrier(0);
          }
          hr[6] -= ht[6];
          hr[7] -= Tt[36] * ht[7];
        }
      }
#pragma unroll
      for (int c = 0; c < kBlk; ++c) hr[c] = sum8(hr[c]);
      hdd = sum8(hdd);
      bd = sum8(bd);
      if (sub == 0) {
        double inv = 0, ibd = 0;
        if (take) {
          double *dst = const_cast<double *>(ubase) + r * plane;
#pragma unroll
          for (int c = 0; c < kBlk; ++c) {
            row[kBlk * r + c] = hr[c];
            dst[c] = hr[c];
          }
          be.b_d[i] = bd;
          const double kIdepthNullSpaceThreshold = 1e-15;
          if (hdd > kIdepthNullSpaceThreshold) {
            inv = 1.0 / hdd;
            be.inv_hdd[i] = inv;
            flg &= static_cast<uint8_t>(~kFlagIllConditioned);
            ibd = inv * bd;
          } else {
            flg |= kFlagIllConditioned;
          }
          be.flags[i] = flg;
        }
        wgt[l] = inv;
        wbd[l] = ibd;
        if (!take) {
#pragma unroll
          for (int c = 0; c < kBlk; ++c) row[kBlk * r + c] = 0;
        }
        for (int c = K; c < Kp; ++c) row[c] = (c == K && take) ? bd : 0.0;  // pad columns of the last tile: b_d rides in the first
      }
    }
    TS_STAMP(2);
    __syncthreads();
    TS_STAMP(3);
    // phase 2: this wave's tiles += A^T W A over the chunk (v_mfma_f64_16x16x4_f64), operands of a tile requested up front
#pragma unroll
    for (int q = 0; q < TPW; ++q) {
      if (wave + q * (kSchurThreads / 64) >= n_tiles) break;
      const double *pa = hrow + lane_off + off_a[q];
      const double *pb = hrow + lane_off + off_b[q];
      // operands in two batches of 8 steps (24 words in flight): with all 16 steps preloaded the kernel needs 168 registers and
      // only ONE workgroup fits per compute unit; at <= 128 two fit, and the second hides the first one's memory round trips
      // (windows of 13 .. 16 keyframes: 4 and 5 resident accumulators per wave — the batches shrink to 4 steps, 12 words in flight, so
      // that the kernel keeps its 128 registers without scratch: it spilled 11 / 47 registers with batches of 8)
      constexpr int kBatches = TPW >= 4 ? 4 : 2;
      constexpr int kHalf = kSchurLandmarks / 4 / kBatches;
      f64x4 c4 = acc[q];
#pragma unroll
      for (int half = 0; half < kBatches; ++half) {
        double av[kHalf], bv[kHalf], wv[kHalf];
#pragma unroll
        for (int s4 = 0; s4 < kHalf; ++s4) {
          const int st4 = 4 * (half * kHalf + s4);
          wv[s4] = wgt[st4 + lk];
          av[s4] = pa[st4 * stride];
          bv[s4] = pb[st4 * stride];
        }
#pragma unroll
        for (int s4 = 0; s4 < kHalf; ++s4) c4 = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[s4] * av[s4], bv[s4], c4, 0, 0, 0);
      }
      acc[q] = c4;
      __builtin_amdgcn_sched_barrier(0);  // one tile's operands at a time: hoisting the next tile's LDS reads above costs the registers
    }
    TS_STAMP(4);
    if (!bd_in_pad && tid < K) {
      double s = 0, s1 = 0, s2 = 0, s3 = 0;  // four partial sums, 16 LDS reads in flight per batch
#pragma unroll 4
      for (int ll = 0; ll < kSchurLandmarks; ll += 4) {
        s += wbd[ll] * hrow[ll * stride + tid];
        s1 += wbd[ll + 1] * hrow[(ll + 1) * stride + tid];
        s2 += wbd[ll + 2] * hrow[(ll + 2) * stride + tid];
        s3 += wbd[ll + 3] * hrow[(ll + 3) * stride + tid];
      }
      bs_acc += (s + s1) + (s2 + s3);
    }
    TS_STAMP(5);
  }
  if (kStamps && a.dbg && threadIdx.x == 0 && blockIdx.x == 1) a.dbg[6] = wall_clock64();
  // ---- this workgroup's partial system, written once, in the MFMA's own layout: [tile][reg][lane] (coalesced 512-byte stores
  // per wave), then K entries of b_schur when the tiles have no spare column
  double *out = a.schur_partials + static_cast<size_t>(blockIdx.x) * twoStagePartialCount(F);
#pragma unroll
  for (int q = 0; q < TPW; ++q) {
    const int tile = wave + q * (kSchurThreads / 64);
    if (tile >= n_tiles) break;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) out[static_cast<size_t>(tile) * 256 + reg * 64 + lane] = acc[q][reg];
  }
  if (!bd_in_pad && static_cast<int>(threadIdx.x) < K) out[static_cast<size_t>(n_tiles) * 256 + threadIdx.x] = bs_acc;
  if (kStamps && a.dbg && threadIdx.x == 0 && blockIdx.x == 1) a.dbg[7] = wall_clock64();
}

struct CombineArgs {
  const PairConst *pc;
  const double *schur_partials, *pair_out;
  const LmControl *ctrl;  // nullable
  double lambda;          // when ctrl is null
  double *comb;
  // nullable: write the four dense arrays of the stage API instead (H_pp K x K | b_pp K | H_schur K x K | b_schur K, undamped,
  // both triangles) — the deterministic build of dsopp_hip_window_linearize under dsopp_hip_window_set_deterministic
  double *dense;
  int F, n_schur_wgs;
};

constexpr int kCombineEntries = 32, kCombineSlices = 32;  // a workgroup of 1024 threads: 32 entries x 32 slices of the partial systems (512 partial systems = ONE round of 16 loads per thread; with 16 slices two)

/** pair-block part of entry (row R, col C), R >= C, of the combined system (without the damping of the diagonal).  Slots of
 *  unconnected pairs hold zeros (stage 1), so every slot is added, in frame order, with all loads in flight together. */
__device__ inline double pairEntry(const CombineArgs &a, int R, int C) {
  const int F = a.F, bi = R >> 3, bj = C >> 3, i = R & 7, j = C & 7;
  if (bi == bj) {
    const int f = bi;
    double v[2 * kMaxFrames];
#pragma unroll
    for (int t = 0; t < kMaxFrames; ++t) {
      const int tc = t < F ? t : 0;
      v[t] = a.pair_out[static_cast<size_t>(f * kMaxFrames + tc) * kPairOut + 8 * i + j];                     // f as reference: T^T G T
      v[kMaxFrames + t] = a.pair_out[static_cast<size_t>(tc * kMaxFrames + f) * kPairOut + 136 + 8 * i + j];  // f as target: G
    }
    double pair = 0;
#pragma unroll
    for (int t = 0; t < kMaxFrames; ++t) pair += (t < F && t != f) ? v[t] : 0.0;
#pragma unroll
    for (int t = 0; t < kMaxFrames; ++t) pair += (t < F && t != f) ? v[kMaxFrames + t] : 0.0;
    return pair;
  }
  // block (a, b), a > b: H_rt = -(G T)^T of the pair (r = a, t = b), H_tr = -G T of the pair (r = b, t = a)
  const double p0 = a.pair_out[static_cast<size_t>(bi * kMaxFrames + bj) * kPairOut + 64 + 8 * j + i];
  const double p1 = a.pair_out[static_cast<size_t>(bj * kMaxFrames + bi) * kPairOut + 64 + 8 * i + j];
  return -p0 - p1;
}
__device__ inline double pairRhs(const CombineArgs &a, int c) {
  const int F = a.F, f = c >> 3, i = c & 7;
  double v[2 * kMaxFrames];
#pragma unroll
  for (int t = 0; t < kMaxFrames; ++t) {
    const int tc = t < F ? t : 0;
    v[t] = a.pair_out[static_cast<size_t>(f * kMaxFrames + tc) * kPairOut + 128 + i];
    v[kMaxFrames + t] = a.pair_out[static_cast<size_t>(tc * kMaxFrames + f) * kPairOut + 200 + i];
  }
  double pair = 0;
#pragma unroll
  for (int t = 0; t < kMaxFrames; ++t) pair += (t < F && t != f) ? v[t] : 0.0;
#pragma unroll
  for (int t = 0; t < kMaxFrames; ++t) pair -= (t < F && t != f) ? v[kMaxFrames + t] : 0.0;
  return pair;
}

/** stage 2: one thread group per word of the partial systems (coalesced reads in the MFMA tile layout): 32 words per workgroup,
 *  16 threads per word — thread `slice` adds the partial systems slice, slice + 16, ... with sixteen loads in flight (one thread
 *  walking all partial systems of a word is a chain of L2 round trips), the slice sums are then added in slice order: a
 *  fixed order for a given window, hence bit-reproducible.  The word's owner maps it to its entry (row >= col) of the combined
 *  system, adds the pair blocks in fixed order and the damping, and writes the entry — every entry exactly once. */
__global__ void __launch_bounds__(kCombineEntries * kCombineSlices) combineSystemKernel(CombineArgs a) {
  __shared__ double part[kCombineSlices][kCombineEntries];
  if (a.ctrl && !a.ctrl->active) return;
  const int F = a.F, K = kBlk * F, Kp = (K + 15) & ~15, nt = Kp >> 4, n_tiles = nt * (nt + 1) / 2;
  const bool bd_in_pad = K < Kp;
  const int count = twoStagePartialCount(F);
  const int le = threadIdx.x & (kCombineEntries - 1), slice = threadIdx.x / kCombineEntries;
  const int e = blockIdx.x * kCombineEntries + le;
  // which entry of the system is word e?  tile words: row = 16 ti + (lane >> 4) + 4 reg, col = 16 tj + (lane & 15), kept for col >= row
  int row = -1, col = -1;  // (row <= col: the symmetric entry (col, row) of the lower-packed system); col == K: b_schur[row]
  if (e < n_tiles * 256) {
    int ti = 0, rem = e >> 8;
    while (rem >= nt - ti) {
      rem -= nt - ti;
      ++ti;
    }
    const int tj = ti + rem, reg = (e >> 6) & 3, lane = e & 63;
    const int r0 = 16 * ti + (lane >> 4) + 4 * reg, c0 = 16 * tj + (lane & 15);
    if (r0 < K && ((c0 < K && c0 >= r0) || (bd_in_pad && c0 == K))) {
      row = r0;
      col = c0;
    }
  } else if (e < count && !bd_in_pad) {
    row = e - n_tiles * 256;
    col = K;
  }
  // everything that does not depend on the partial sums is requested first (slice 0 owns the entry): damping, pair blocks
  const bool owner = slice == 0 && row >= 0;
  double lam = 0, pair = 0;
  if (owner) {
    lam = a.ctrl ? a.ctrl->lambda : a.lambda;
    pair = col == K ? pairRhs(a, row) : pairEntry(a, col, row);
  }
  double s = 0;
  if (row >= 0) {
    // every round has kDepth loads in flight per thread, the last one included (indices beyond the last partial system are clamped
    // and their values dropped): 313 partial systems are 2 rounds — as a batched loop with a scalar tail they were 4 + 7 round trips
    const double *src = a.schur_partials + e;
    constexpr int kDepth = 16;
    const int last = a.n_schur_wgs - 1;
    for (int w = slice; w <= last; w += kDepth * kCombineSlices) {
      double v[kDepth];
#pragma unroll
      for (int q = 0; q < kDepth; ++q) v[q] = src[static_cast<size_t>(min(w + q * kCombineSlices, last)) * count];
#pragma unroll
      for (int q = 0; q < kDepth; ++q) s += (w + q * kCombineSlices <= last) ? v[q] : 0.0;
    }
  }
  part[slice][le] = s;
  __syncthreads();
  if (!owner) return;
  double schur = 0;
#pragma unroll
  for (int q = 0; q < kCombineSlices; ++q) schur += part[q][le];
  if (a.dense) {
    double *Hpp = a.dense, *bpp = Hpp + static_cast<size_t>(K) * K, *Hs = bpp + K, *bs = Hs + static_cast<size_t>(K) * K;
    if (col == K) {
      bpp[row] = pair;
      bs[row] = schur;
    } else {
      Hpp[static_cast<size_t>(col) * K + row] = pair;
      Hpp[static_cast<size_t>(row) * K + col] = pair;
      Hs[static_cast<size_t>(col) * K + row] = schur;
      Hs[static_cast<size_t>(row) * K + col] = schur;
    }
    return;
  }
  const double sc = -1.0 / (1.0 + lam);
  if (col == K) {
    a.comb[combBlockCount(F) * 64 + row] = pair + sc * schur;
  } else {
    if (row == col) pair *= 1.0 + lam;
    a.comb[combIndex(col, row)] = pair + sc * schur;
  }
}

}  // namespace dsopp_hip
