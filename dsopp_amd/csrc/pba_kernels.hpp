// HIP kernels of the sliding-window photometric bundle adjustment (hot loops A, B, C, D of SURVEY.md §3.2).
//
// Design: the reference materialises a 2.1 KB ResidualPoint per (landmark, target) in
// evaluateJacobians and re-reads it in evaluateLinearSystemPosePose and ...SchurComplement.  Here one sweep kernel
// evaluates residual + Jacobian rows in registers and reduces them on the fly:
//   * every Jacobian row has the form  J_t = -g,  J_r = g * T  with  g = [D(6), c, 1]  and the per-pair constant
//     T = blockdiag(Adj, 1, s0)  (evaluate_jacobians.hpp:149-182), so the three 8x8 Gram blocks and two 8-vectors the
//     reference accumulates per pair (hessian_block_evaluation.hpp:74-83) all follow from ONE symmetric 8x8 G = sum w g^T g and
//     one q = sum w g^T r per pair: H_tt = G, H_rr = T^T G T, H_rt = -T^T G, b_t = -q, b_r = T^T q;
//   * the per-landmark Schur quantities (hessian_block_evaluation.hpp:198-212) need only u = w sum_k g_k Jd_k (8), hdd, bd per
//     (landmark, target): h_p[t-block] = -u, h_p[r-block] = sum_t T^T u.
// Work is laid out so that a wavefront shares one frame pair: pair constants are wave-uniform (scalar registers),
// landmark SoA reads are coalesced, the image is a gather of 2 x 64 B segments per pattern pixel.
#pragma once
#include <type_traits>
#include "device_geom.hpp"
#include "pba_types.hpp"

namespace dsopp_hip {

// pattern offsets (x_i, y_i) — src/common/pattern/include/common/pattern/pattern.hpp:21-32
__device__ __constant__ const int kPatX[kPat] = {0, -1, 1, -2, 0, 2, -1, 0};
__device__ __constant__ const int kPatY[kPat] = {2, 1, 1, 0, 0, 0, -1, -2};

/** index of (i, j), i <= j, in the packed upper triangle of a symmetric 8x8 */
__host__ __device__ constexpr int triIdx(int i, int j) { return i * kBlk - i * (i - 1) / 2 + (j - i); }
__host__ __device__ inline int symIdx(int i, int j) { return i <= j ? triIdx(i, j) : triIdx(j, i); }

// ---------------------------------------------------------------------------------------------------------------
// pair constants
// ---------------------------------------------------------------------------------------------------------------
__device__ inline void buildProjectionMatrices(const Rigid &T, const FrameDev &fr, const FrameDev &ft, double *U, double *M) {
  // ArrayReprojector ctor — camera_reproject.hpp:235-260
  const double ifx = 1.0 / fr.fx, ify = 1.0 / fr.fy;
  const double k02 = -fr.cx / fr.fx, k12 = -fr.cy / fr.fy;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    U[4 * i + 0] = T.R[3 * i + 0] * ifx;
    U[4 * i + 1] = T.R[3 * i + 1] * ify;
    U[4 * i + 2] = T.R[3 * i + 0] * k02 + T.R[3 * i + 1] * k12 + T.R[3 * i + 2];
    U[4 * i + 3] = T.t[i];
  }
  if (M) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      M[0 + j] = ft.fx * U[0 + j] + ft.cx * U[8 + j];
      M[4 + j] = ft.fy * U[4 + j] + ft.cy * U[8 + j];
      M[8 + j] = U[8 + j];
    }
  }
}

// (the frame table / window state pointers of the functions below are templates: callers hand them over as address_space(1)
// pointers — glb() — so that every access is a global load instead of a FLAT one)
template <typename FramesPtr, typename StatePtr>
__device__ __forceinline__ double linearisationScale(FramesPtr frames, StatePtr st, int r, int t) {
  return (frames[t].exposure / frames[r].exposure) * exp(st->ab0[t][0] - st->ab0[r][0]);
}

/** exp of the twist eps_f + step_f (sign = +1) or of its negative (sign = -1) */
template <typename StatePtr>
__device__ __forceinline__ Rigid frameIncrement(StatePtr st, int f, double sign) {
  double xi[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) xi[i] = sign * (st->eps[f][i] + st->step[f][i]);
  return rigidExp(xi);
}

/** evaluate_jacobians.hpp:36-66 (per-pair prologue) + first_estimate_jacobians.hpp:22-31.
 *  Er = exp(eps_r + step_r), Emt = exp(-(eps_t + step_t)) may be supplied by the caller (shared through LDS). */
/** (forced inline: as an out-of-line function its pointer arguments were generic — 134 FLAT accesses — and the Rigid temporaries
 *  lived in scratch; it runs on F^2 lanes of ONE workgroup at the head of every solve).
 *  PART selects what is computed and stored: 0 everything; 1 the members that move with the state (M, s, b_t, b_r) and `valid`;
 *  2 the members of the linearisation point (T0rel, U, tl, Adj, camera, s0, b_r0, sigma_r).  The head of a solve runs the two
 *  halves of a pair on two different waves: the one workgroup is bound by its instruction stream (~2100 instructions per lane),
 *  and two SIMDs run the two halves side by side. */
template <int PART = 0>
__device__ __forceinline__ void computePairConst(const FrameDev *frames_generic, const WindowState *st_generic, PairConst *pc_generic, int r, int t,
                                                 int F, bool fej, const Rigid *Er = nullptr, const Rigid *Emt = nullptr) {
  const auto frames = glb(frames_generic);
  const auto st = glb(st_generic);
  auto &P = glb(pc_generic)[r * kMaxFrames + t];
  const auto &fr = frames[r];
  const auto &ft = frames[t];
  const bool valid = (r != t) && fr.status[t] != nullptr;
  if (PART != 2) P.valid = valid ? 1 : 0;
  if (!valid) return;
  Rigid Tr0, Tt0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    Tr0.R[i] = st->T0_R[r][i];
    Tt0.R[i] = st->T0_R[t][i];
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Tr0.t[i] = st->T0_t[r][i];
    Tt0.t[i] = st->T0_t[t][i];
  }
  const Rigid T_tr0 = rigidMul(rigidInverse(Tt0), Tr0);
  if (PART != 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) P.T0rel[4 * i + j] = T_tr0.R[3 * i + j];
      P.T0rel[4 * i + 3] = T_tr0.t[i];
    }
  }
  // (projection matrices / adjoint are built in registers and stored: the helpers take generic pointers)
  FrameDev ir{}, it{};
  ir.fx = fr.fx;
  ir.fy = fr.fy;
  ir.cx = fr.cx;
  ir.cy = fr.cy;
  it.fx = ft.fx;
  it.fy = ft.fy;
  it.cx = ft.cx;
  it.cy = ft.cy;
  const bool need_current = PART != 2 || !fej;  // without first-estimate Jacobians the linearisation point IS the current state
  Rigid T_tr = T_tr0;
  double b_r = 0, s_cur = 0;
  if (need_current) {
    T_tr = rigidMul(Emt ? *Emt : frameIncrement(st, t, -1.0), rigidMul(T_tr0, Er ? *Er : frameIncrement(st, r, 1.0)));
    const double a_r = st->ab0[r][0] + st->eps[r][6] + st->step[r][6];
    const double a_t = st->ab0[t][0] + st->eps[t][6] + st->step[t][6];
    b_r = st->ab0[r][1] + st->eps[r][7] + st->step[r][7];
    s_cur = (ft.exposure / fr.exposure) * exp(a_t - a_r);
  }
  if (PART != 2) {
    const double b_t = st->ab0[t][1] + st->eps[t][7] + st->step[t][7];
    P.s = s_cur;
    P.b_t = b_t;
    P.b_r = b_r;
    double Ucur[12], Mcur[12];
    buildProjectionMatrices(T_tr, ir, it, Ucur, Mcur);
#pragma unroll
    for (int i = 0; i < 12; ++i) P.M[i] = Mcur[i];
  }
  if (PART != 1) {
    const Rigid Tlin = fej ? T_tr0 : T_tr;
    double Ulin[12], adj[36];
    buildProjectionMatrices(Tlin, ir, it, Ulin, nullptr);
    rigidAdj(Tlin, adj);
#pragma unroll
    for (int i = 0; i < 12; ++i) P.U[i] = Ulin[i];
#pragma unroll
    for (int i = 0; i < 36; ++i) P.Adj[i] = adj[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) P.tl[i] = Tlin.t[i];
    P.fxt = ft.fx;
    P.fyt = ft.fy;
    P.cxt = ft.cx;
    P.cyt = ft.cy;
    if (fej) {
      P.s0 = linearisationScale(frames, st, r, t);
      P.b_r0 = st->ab0[r][1];
      int last = -1;
      for (int k = 0; k < F; ++k)
        if (k != r && fr.status[k] != nullptr) last = k;
      P.sigma_r = linearisationScale(frames, st, r, last);
    } else {
      P.s0 = s_cur;
      P.b_r0 = b_r;
      P.sigma_r = s_cur;
    }
  }
}

/** FEJ fast path of the solve kernel: only the state-dependent members (M, s, b_t, b_r) move with eps + step */
__device__ inline void refreshPairCurrent(const FrameDev *frames, const WindowState *st, PairConst *pc, int r, int t, const Rigid &Er,
                                          const Rigid &Emt) {
  PairConst &P = pc[r * kMaxFrames + t];
  if (!P.valid) return;
  Rigid T0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) T0.R[3 * i + j] = P.T0rel[4 * i + j];
    T0.t[i] = P.T0rel[4 * i + 3];
  }
  const Rigid T_tr = rigidMul(Emt, rigidMul(T0, Er));
  const FrameDev &fr = frames[r];
  const FrameDev &ft = frames[t];
  double U[12];
  buildProjectionMatrices(T_tr, fr, ft, U, P.M);
  const double a_r = st->ab0[r][0] + st->eps[r][6] + st->step[r][6];
  const double a_t = st->ab0[t][0] + st->eps[t][6] + st->step[t][6];
  P.s = (ft.exposure / fr.exposure) * exp(a_t - a_r);
  P.b_t = st->ab0[t][1] + st->eps[t][7] + st->step[t][7];
  P.b_r = st->ab0[r][1] + st->eps[r][7] + st->step[r][7];
}

__global__ void __launch_bounds__(kMaxFrames *kMaxFrames) pairSetupKernel(const FrameDev *frames, const WindowState *st, PairConst *pc, int F, int fej,
                                                                          const int *run_flag) {
  if (run_flag && !*run_flag) return;
  const int idx = threadIdx.x;
  if (idx >= F * F) return;
  computePairConst(frames, st, pc, idx / F, idx % F, F, fej != 0);
}

// ---------------------------------------------------------------------------------------------------------------
// block reductions
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double waveSum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

/** wave sum without LDS traffic: DPP within the rows of 16 lanes (xor 1, xor 2, half mirror, mirror), then the four row totals
 *  read with v_readlane and added in row order.  Every lane receives the same, order-fixed total.  (__shfl_xor is six dependent
 *  ds_bpermute round trips.) */
__device__ __forceinline__ double waveSumDpp(double v) {
  auto mov = [](double x, auto ctrl_tag) {
    constexpr int CTRL = decltype(ctrl_tag)::value;
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
  };
  v += mov(v, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
  v += mov(v, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
  v += mov(v, std::integral_constant<int, 0x141>{});  // row_half_mirror
  v += mov(v, std::integral_constant<int, 0x140>{});  // row_mirror
  auto row = [&](int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
  };
  return ((row(0) + row(16)) + row(32)) + row(48);
}

/** sums `N` per-thread doubles over the block; result valid in thread 0's `vals` */
template <int N, int THREADS>
__device__ __forceinline__ void blockSum(double (&vals)[N], double *lds /* [THREADS/64][N] */) {
  constexpr int kWaves = THREADS / 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < N; ++i) vals[i] = waveSum(vals[i]);
  if (kWaves > 1) {
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < N; ++i) lds[wave * N + i] = vals[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        double s = lds[i];
        for (int w = 1; w < kWaves; ++w) s += lds[w * N + i];
        vals[i] = s;
      }
    }
  }
}

constexpr int kItemsPerBlock = kSweepThreads / kPat;  // 8 lanes (one per pattern pixel) per (landmark, target) item

/** firstEstimateJacobians_ (first_estimate_jacobians.hpp:14-71): validity of the reprojection at the linearisation point
 *  + idepth snapshot.  The geometric Jacobians themselves are NOT stored (they are recomputed from the snapshot). */
template <typename S>
__global__ void __launch_bounds__(64) fejKernel(const FrameDev *__restrict__ frames, const PairConst *__restrict__ pc,
                                                const SweepBlock *__restrict__ table, int n_entries) {
  // one wave = 4 table entries of kItemsPerBlock (16) landmarks each
  const int entry = blockIdx.x * (64 / kItemsPerBlock) + (threadIdx.x / kItemsPerBlock);
  if (entry >= n_entries) return;
  const SweepBlock be = table[entry];
  const FrameDev &fr = frames[be.r];
  const FrameDev &ft = frames[be.t];
  const PairConst &P = pc[be.r * kMaxFrames + be.t];
  const int i = be.offset + threadIdx.x % kItemsPerBlock;
  if (i >= fr.n_res[be.t]) return;
  const uint8_t flg = fr.flags[i];
  if ((flg & kFlagMarginalized) && !(flg & kFlagToMarginalize)) return;
  const S u = static_cast<S>(fr.uv[2 * i]), v = static_cast<S>(fr.uv[2 * i + 1]);
  const double idepth_d = fr.idepth[i];
  const S idepth = static_cast<S>(idepth_d);
  bool ok = validIdepth(idepth) && insideROI(u - S(2), v - S(2), S(fr.width), S(fr.height)) &&
            insideROI(u + S(2), v + S(2), S(fr.width), S(fr.height));
  const S U0 = S(P.U[0]), U1 = S(P.U[1]), U4 = S(P.U[4]), U5 = S(P.U[5]), U8 = S(P.U[8]), U9 = S(P.U[9]);
  const S cX = S(P.U[2]) + S(P.U[3]) * idepth, cY = S(P.U[6]) + S(P.U[7]) * idepth, cZ = S(P.U[10]) + S(P.U[11]) * idepth;
#pragma unroll
  for (int k = 0; k < kPat; ++k) {
    const S pu = u + S(kPatX[k]), pv = v + S(kPatY[k]);
    const S X = U0 * pu + U1 * pv + cX, Y = U4 * pu + U5 * pv + cY, Z = U8 * pu + U9 * pv + cZ;
    const S tu = (S(P.fxt) * X + S(P.cxt) * Z) / Z, tv = (S(P.fyt) * Y + S(P.cyt) * Z) / Z;
    ok = ok && (Z > S(0)) && insideROI(tu, tv, S(ft.width), S(ft.height));
  }
  fr.fej_valid[be.t][i] = ok ? 1 : 0;
  fr.idepth_fej[i] = idepth_d;  // every connected target writes the same value
}

// ---------------------------------------------------------------------------------------------------------------
// the sweep: residuals (+ Jacobian rows) over all (landmark, target) items
// ---------------------------------------------------------------------------------------------------------------
struct SweepParams {
  double sigma_huber;
  int for_marginalized;  // accumulate only landmarks flagged to_marginalize (FOR_MARGINALIZED of the reference)
  int use_fej_flag;      // FIRST_ESTIMATE_JACOBIANS: success requires reprojection_jacobians_valid
  const LmControl *ctrl;   // nullable: device-driven LM loop (active / linear_system_valid / pending / lambda are read here)
  double *clear_buf;       // LIN: buffer the following reduction kernel accumulates into with atomics; zeroed here
  int clear_count;
  const double *step;      // BACKSUB: pose step of calculateStep (K doubles)
  double lambda;             // BACKSUB: damping when ctrl == nullptr
  int F;
  int ublk_read, ublk_write;  // parity of the double-buffered Schur rows read by BACKSUB / written by LIN (0, 0 outside the fused loop)
  int gate_on_pending;        // fused loop: BACKSUB only when a candidate step is pending (LmControl::pending)
  const int *run_flag;        // nullable: launch is a no-op unless *run_flag != 0 (closing evaluation after a rejected step)
  int external_backsub;       // large windows: calculateIdepths ran as its own kernel (once per landmark instead of once per
                              // (landmark, target) item); this sweep still records the step norms
  long long *dbg;  // nullable tuning aid: per-phase wall_clock64 stamps of workgroup 0 / max end stamp
};
// Phase stamps (wall_clock64) are a tuning aid: compiled in only with -DDSOPP_HIP_STAMPS (scripts/dbg_*.py); the release
// kernels carry none of it.
#ifdef DSOPP_HIP_STAMPS
constexpr bool kStamps = true;
#else
constexpr bool kStamps = false;
#endif
#define SWEEP_STAMP(i) do { if (kStamps && prm.dbg && threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) prm.dbg[i] = wall_clock64(); } while (0)


/** Workgroup barrier that only orders LDS traffic: unlike __syncthreads() it does not drain outstanding global stores
 *  (vmcnt), which costs 1-2 us when a phase ends with scattered stores nobody in the workgroup reads back. */
__device__ __forceinline__ void ldsBarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


/** v of the lane selected by a DPP control word (quad_perm / row_half_mirror / ...): pure VALU, no LDS round trip */
template <int CTRL>
__device__ __forceinline__ double dppMove(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

/** sum over the 8 lanes of an item (lanes differ in bits 0..2); every lane receives the total.
 *  quad_perm [1,0,3,2] (0xB1), quad_perm [2,3,0,1] (0x4E), row_half_mirror (0x141: lane i <-> 7 - i) */
__device__ __forceinline__ double sum8(double v) {
  v += dppMove<0xB1>(v);
  v += dppMove<0x4E>(v);
  v += dppMove<0x141>(v);
  return v;
}

/** 1 / x for the projective divisions: v_rcp_f64 (2^-23 relative) + two Newton steps, ~1 ulp; 5 instructions instead of the
 *  ~13 of an IEEE division.  x = 0 gives inf -> NaN, which every caller rejects through its ROI / z > 0 test. */
__device__ __forceinline__ double fastRcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float fastRcp(float x) { return 1.0f / x; }

/** 1 / sqrt(x), x > 0: v_rsq_f64 + two Newton steps (~1 ulp) */
__device__ __forceinline__ double fastRsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = fma(y, fma(-h * y, y, 0.5), y);
  y = fma(y, fma(-h * y, y, 0.5), y);
  return y;
}

using f64x4 = __attribute__((ext_vector_type(4))) double;

// ---- the per-pair Gram G = sum w g^T g, q = sum w g^T r on the f64 matrix cores -------------------------------------------
// Until round 4 every lane kept the 36 + 8 running sums of its own pattern pixel in registers (96 VGPRs: the sweep ran at 2 waves per
// SIMD, VALU busy 37 %, bound by its own dependent latencies) and the workgroup summed 48 x 128 values through an LDS transpose.  Now a
// lane hands the two 8-vectors of its pixel
//     a = w [g0 .. g6, r]      b = [g0 .. g6, 1]                               (g7 = 1: evaluate_jacobians.hpp:181)
// to the wave through LDS and v_mfma_f64_16x16x4_f64 contracts them over the pixels: E = sum_p a(p) b(p)^T holds
//     E[i][j] = G[i][j] (i, j < 7),  E[i][7] = G[i][7],  E[7][j] = q[j],  E[7][7] = q[7];
// only G[7][7] = sum w is left to a scalar.  One instruction contracts 4 values of k; its 16 rows / columns carry TWO pixel sets
// (rows and columns 0..7: pixels 0..3 of an item, 8..15: pixels 4..7), whose products sit in the two diagonal 8 x 8 blocks of the
// result (the off-diagonal blocks are cross terms nobody reads): 8 instructions per 64 pixels, 8 accumulator registers per lane
// instead of 96, and the sum over lanes is the instruction's own.  (hessian_block_evaluation.hpp:74-83 is what this accumulates.)
constexpr int kGramStride = 18;  // doubles per pixel: a[8] | b[8] | pad 2 — 144 B: the two 8-double runs a 16-lane read touches
                                 // (pixels 4 apart) fall on disjoint halves of the 32 banks, b128 stores of 8 adjacent lanes cover all
constexpr int kSweepScalars = 5; // energy, n_valid, |idepth step|^2, idepth . step, sum of weights (= G[7][7])

/** hands (a, b) of this lane's pixel to the wave: its row of the wave's exchange buffer */
__device__ __forceinline__ void gramPublish(const double (&a)[kBlk], const double (&b)[kBlk], double *wave_rows /* [64][kGramStride] */) {
  const int lane = threadIdx.x & 63;
  double2 *row = reinterpret_cast<double2 *>(wave_rows + lane * kGramStride);
#pragma unroll
  for (int c = 0; c < kBlk / 2; ++c) {
    row[c] = double2{a[2 * c], a[2 * c + 1]};
    row[kBlk / 2 + c] = double2{b[2 * c], b[2 * c + 1]};
  }
}

/** adds the products of the 64 published pixels to acc.  The sweep calls this for group g - 1 right behind the texel requests of
 *  group g: the eight dependent matrix instructions (64 cycles each) and the LDS round trip in front of them run in the shadow of
 *  the gather's memory latency instead of on the group's own critical path.  (LDS instructions of one wave execute in order: these
 *  reads see the wave's stores, and the next group's stores come after them — no barrier, only the compiler is kept in line.) */
__device__ __forceinline__ void gramContract(const double *wave_rows /* [64][kGramStride] */, f64x4 &acc) {
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // operand lane l = (m = l & 15, k = l >> 4) of step s reads value m & 7 of pixel 8 s + 4 (m >> 3) + k
  const double *src = wave_rows + (4 * ((lane & 15) >> 3) + (lane >> 4)) * kGramStride + (lane & 7);
  double av[8], bv[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    av[s] = src[8 * s * kGramStride];
    bv[s] = src[8 * s * kGramStride + kBlk];
  }
  asm volatile("" ::: "memory");  // (all sixteen operands are requested before the first instruction waits for one)
#pragma unroll
  for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

/** end of a sweep workgroup: the waves' Gram accumulators and scalars -> partials[kPartial] of the workgroup (fixed order) */
template <bool LIN>
__device__ __forceinline__ void gramScalarsStore(const f64x4 &acc, const double (&sc)[kSweepScalars], double *lds /* [waves][64 + kSweepScalars] */,
                                                 double *__restrict__ out) {
  constexpr int kWaves = kSweepThreads / 64, kSlot = 64 + kSweepScalars;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // result layout of the instruction: register v of lane l = entry (row (l >> 4) + 4 v, column l & 15).  Lanes with column < 8 hold
  // the first set's block in registers 0, 1; the second set's block (rows / columns 8..15) sits in registers 2, 3 of the lanes 8 to
  // the right in the same row of 16 (row_ror:8)
  const double e0 = acc[0] + dppMove<0x128>(acc[2]), e1 = acc[1] + dppMove<0x128>(acc[3]);
  double s[kSweepScalars];
#pragma unroll
  for (int i = 0; i < kSweepScalars; ++i) s[i] = waveSumDpp(sc[i]);
  const int col = lane & 15, r4 = lane >> 4;
  if (LIN && col < 8) {
    lds[wave * kSlot + 8 * r4 + col] = e0;             // E[r4][col]
    lds[wave * kSlot + 8 * (r4 + 4) + col] = e1;       // E[r4 + 4][col]
  }
  if (lane < kSweepScalars) {
    double v = s[0];
#pragma unroll
    for (int i = 1; i < kSweepScalars; ++i) v = lane == i ? s[i] : v;
    lds[wave * kSlot + 64 + lane] = v;
  }
  ldsBarrier();
  if (threadIdx.x < kSlot && (LIN || threadIdx.x >= 64)) {  // (residual-only sweeps: the four scalars alone)
    const int e = threadIdx.x;
    double v = lds[e];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) v += lds[w * kSlot + e];
    // where entry e goes: E[i][j] -> G upper triangle (i <= j < 7, and column 7), q (row 7); scalars -> 44..47, sum w -> G[7][7]
    int idx = -1;
    if (e < 64) {
      const int i = e >> 3, j = e & 7;
      if (i < 7) idx = j >= i ? triIdx(i, j) : -1;
      else idx = 36 + j;
    } else {
      idx = e - 64 < 4 ? 44 + (e - 64) : (LIN ? triIdx(7, 7) : -1);
    }
    if (idx >= 0) out[idx] = v;
  }
}

/**
 * evaluateJacobians<S, SE3, Pinhole, 8, PixelMap, 1, FEJ, OPT_IDEPTHS, LIN, true, HUBER> fused with
 * evaluateLinearSystemPosePoseBlock and the per-(landmark,target) part of ...SchurComplement
 * (PBA_INT/evaluate_jacobians.hpp:20-202, hessian_block_evaluation.hpp:38-90,198-212).
 *   LIN = false: residual-only sweep (calculateEnergy), optionally with calculateIdepths fused in (BACKSUB).
 *   LIN = true : linearisation sweep.
 *   WFEJ = true: the opening linearisation of a fused solve — the sweep IS firstEstimateJacobians_ for its items: it takes the
 *                inverse-depth snapshot and the reprojection validity itself (and uses them) instead of a kernel in front.
 * Thread mapping: one lane per pattern pixel, 8 adjacent lanes per (landmark, target) item, a workgroup = 16 items of one
 * ordered frame pair.  The dependent chain of a lane is: landmark words (coalesced, broadcast within the item) ->
 * 1 reprojection -> 4 texel loads (2 x 64 B segments) -> 1 Jacobian row -> reductions, so a C1-sized sweep is one
 * memory round trip deep per stage instead of eight.
 */
template <typename S, bool LIN, bool FEJ, bool HUBER, bool BACKSUB = false, bool WFEJ = false>
#ifdef DSOPP_HIP_EXPERIMENT_SWEEP_WAVES  // experiment: force an occupancy (waves per SIMD) on the sweeps, whatever it spills
__global__ void __launch_bounds__(kSweepThreads, DSOPP_HIP_EXPERIMENT_SWEEP_WAVES) sweepKernel(
#else
__global__ void __launch_bounds__(kSweepThreads) sweepKernel(
#endif
const FrameDev *__restrict__ frames, const PairConst *__restrict__ pc,
                                                             const SweepBlock *__restrict__ table, double *__restrict__ partials,
                                                             // (= prm.ctrl, prm.run_flag once more, as pointer arguments of their own: the
                                                             // first 16 argument words are preloaded into scalar registers by the dispatcher
                                                             // — build.sh: -amdgpu-kernarg-preload-count — and members of a by-value struct
                                                             // are not; with them the prologue's burst below starts at the wave's first cycle)
                                                             const LmControl *ctrl_arg, const int *run_flag_arg,
                                                             SweepParams prm) {
  // LIN: the exchange buffer of the Gram accumulation (gramAccumulate: one row of kGramStride doubles per pattern pixel);
  // the four energy scalars (+ sum of weights) cross the workgroup through kSweepScalars doubles per wave
  __shared__ __attribute__((aligned(16))) double gram_lds[LIN ? kSweepThreads * kGramStride : 2];
  __shared__ double scalar_lds[(kSweepThreads / 64) * (64 + kSweepScalars)];
  // ---- round trip 1: block descriptor + LM control block (scalar loads, all requested before any of them is tested)
  // The whole descriptor, the control block's words and the run flag leave in ONE burst of scalar loads with one wait behind it: the
  // control block and the flag are read through a pointer that is valid either way (the table itself stands in for an absent one) and
  // every word is pinned below.  Until round 5 the compiler's own placement was descriptor head -> wait -> control block -> wait ->
  // run flag -> wait -> rest of the descriptor + remaining arguments -> wait: four scalar round trips in front of the first item word.
  const SweepBlock be = table[blockIdx.x];
  // (the stand-in has to cover every word read through it)
  static_assert(sizeof(SweepBlock) >= sizeof(LmControl) && sizeof(SweepBlock) >= sizeof(int), "the sweep table stands in for an absent control block / run flag");
  const LmControl DSOPP_CONSTANT *ctrl_or_any =
      ctrl_arg ? (const LmControl DSOPP_CONSTANT *)ctrl_arg : (const LmControl DSOPP_CONSTANT *)(const void DSOPP_CONSTANT *)table;
  const int DSOPP_CONSTANT *flag_or_any = run_flag_arg ? (const int DSOPP_CONSTANT *)run_flag_arg : (const int DSOPP_CONSTANT *)(const void DSOPP_CONSTANT *)table;
  int c_active = ctrl_or_any->active, c_lsv = ctrl_or_any->linear_system_valid, c_pending = ctrl_or_any->pending, run = *flag_or_any;
  double lam = ctrl_or_any->lambda;
  asm volatile("" ::"s"(be.r), "s"(be.t), "s"(be.offset), "s"(be.n_res), "s"(be.cap), "s"(be.owns_landmark_sums), "s"(be.width_r), "s"(be.height_r),
               "s"(be.width_t), "s"(be.height_t), "s"(be.conn_mask), "s"(be.n_groups), "s"(be.uv), "s"(be.idepth), "s"(be.patch), "s"(be.idepth_fej),
               "s"(be.b_d), "s"(be.inv_hdd), "s"(be.idepth_step), "s"(be.ublk), "s"(be.energy), "s"(be.flags), "s"(be.status), "s"(be.fej_valid),
               "s"(be.cand), "s"(be.texels_t), "s"(be.iplane_t), "s"(be.itiles_t), "s"(be.partial_row));
  asm volatile("" : "+s"(c_active), "+s"(c_lsv), "+s"(c_pending), "+s"(run), "+s"(lam));
  // (the argument words beyond the preloaded ones ride in the same burst instead of being fetched one by one where they are first used)
  asm volatile("" ::"s"(prm.sigma_huber), "s"(prm.for_marginalized), "s"(prm.use_fej_flag), "s"(prm.clear_buf), "s"(prm.clear_count), "s"(prm.step),
               "s"(prm.lambda), "s"(prm.F), "s"(prm.ublk_read), "s"(prm.ublk_write), "s"(prm.gate_on_pending), "s"(prm.external_backsub), "s"(prm.dbg));
  if (!ctrl_arg) {
    c_active = 1;
    c_lsv = 0;
    c_pending = 1;
    lam = prm.lambda;
  }
  if (!run_flag_arg) run = 1;
  // device-driven LM: skip when the loop has ended (or, for the linearisation, when the last step was rejected and the
  // linear system is still valid — levenberg_marquardt_algorithm.hpp:88-90)
  if (!run || !c_active || (LIN && c_lsv)) return;
  if (LIN && prm.clear_buf) {
    for (int k = blockIdx.x * kSweepThreads + threadIdx.x; k < prm.clear_count; k += gridDim.x * kSweepThreads) prm.clear_buf[k] = 0;
  }
  SWEEP_STAMP(0);
  // every pointer of the descriptor addresses HBM (see glb())
  const auto g_flags = glb(be.flags), g_status = glb(be.status), g_fej_valid = glb(be.fej_valid);
  const auto g_cand = glb(be.cand);
  const auto g_uv = glb(be.uv), g_idepth = glb(be.idepth), g_patch = glb(be.patch), g_idepth_fej = glb(be.idepth_fej);
  const auto g_b_d = glb(be.b_d), g_inv_hdd = glb(be.inv_hdd);
  const auto g_idepth_step = glb(be.idepth_step), g_ublk = glb(be.ublk), g_energy = glb(be.energy);
  const auto g_step = glb(prm.step);
  const PairConst DSOPP_CONSTANT &P = *(const PairConst DSOPP_CONSTANT *)(pc + (be.r * kMaxFrames + be.t));
  const int k = threadIdx.x & 7;                  // pattern pixel of this lane
  // pattern offsets (x_i, y_i) — src/common/pattern/include/common/pattern/pattern.hpp:21-32, +2 packed in nibbles
  const int ox = static_cast<int>((0x21420312u >> (4 * k)) & 0xFu) - 2;
  const int oy = static_cast<int>((0x01222334u >> (4 * k)) & 0xFu) - 2;

  double sc[kSweepScalars];  // energy, n_valid, |idepth step|^2, idepth . step, sum of weights
#pragma unroll
  for (int e = 0; e < kSweepScalars; ++e) sc[e] = 0;
  f64x4 gram = {0, 0, 0, 0};

  // Large windows (hundreds of thousands of items) give a workgroup several groups of 16 items of its frame pair
  // (SweepBlock::n_groups, chosen by the host): the 48 workgroup sums are reduced once per workgroup instead of once per 16
  // items, and the chip dispatches a quarter of the workgroups.  Small windows keep one group per workgroup (parallelism).
  // Several groups per workgroup: the per-item words of group g + 1 are requested while group g waits for its texels (below), so a
  // group's dependent chain starts at the reprojection instead of at a memory round trip.  (counters at 12 KF / 50 k: VALU busy
  // 37 %, memory unit stalled 0.3 %: with 2 waves per SIMD the sweep is bound by its own dependent latencies)
  struct ItemWords {
    uint8_t flg, status, cand, fej_bit;
    S u, v, patch_k;
    double idepth_d, idepth_step_d, idepth_fej_d;
  };
  auto loadItemWords = [&](int item, bool in_bounds) {
    // lanes past the pair's last residual read the block's first item instead (a valid index): they are inactive below and nothing
    // of what they load is used — no exec-masked region, no default values to materialise
    const int j = in_bounds ? item : be.offset;
    ItemWords w;
    w.flg = g_flags[j];
    w.u = static_cast<S>(g_uv[2 * j]);
    w.v = static_cast<S>(g_uv[2 * j + 1]);
    w.idepth_d = g_idepth[j];
#ifdef DSOPP_HIP_EXPERIMENT_FEW_ITEM_WORDS  // timing only (wrong results): 5 item-word loads per group instead of 10
    w.idepth_step_d = w.idepth_d * 1e-9;
    w.status = g_status[j];
    w.cand = w.status;
    w.patch_k = static_cast<S>(g_patch[kPat * j + k]);
    w.fej_bit = 1;
    w.idepth_fej_d = w.idepth_d;
    return w;
#endif
    w.idepth_step_d = g_idepth_step[j];
    w.status = g_status[j];
    w.cand = g_cand[j];
    w.patch_k = static_cast<S>(g_patch[kPat * j + k]);
    w.fej_bit = 1;
    w.idepth_fej_d = 0;
    if (FEJ) {
      constexpr bool fresh_fej = LIN && WFEJ;
      if (prm.use_fej_flag && !fresh_fej) w.fej_bit = g_fej_valid[j];  // evaluate_jacobians.hpp:94
      if (LIN) w.idepth_fej_d = fresh_fej ? w.idepth_d : g_idepth_fej[j];
    }
    return w;
  };
  ItemWords nxt;
  {
    const int i0 = be.offset + (threadIdx.x >> 3);
    nxt = loadItemWords(i0, i0 < be.n_res);
  }
  for (int grp = 0; grp < be.n_groups; ++grp) {
  const int i = be.offset + grp * kItemsPerBlock + (threadIdx.x >> 3);   // landmark
  // The pair constants are read through pointers that are opaque to the compiler at two points of every group (here, and
  // after the reprojection): each group re-fetches the ~20 doubles it needs with scalar loads next to the vector loads it has
  // to wait for anyway, instead of holding 50+ scalar registers across the loop (they were spilled into vector lanes).
  const PairConst DSOPP_CONSTANT *Pm = &P;
  asm volatile("" : "+s"(Pm) : "v"(i));

  // ---- round trip 2: per-item words (identical addresses within the 8 lanes of an item: one request).  Every load is
  // gated by the index bound only, never by a loaded value, so they all go out together.
  const bool inb = i < be.n_res;
  uint8_t flg = 0;
  S u = S(0), v = S(0);
  double idepth_d = 0, idepth_step_d = 0, idepth_fej_d = 0, bd_d = 0, inv_hdd_d = 0;
  uint8_t status = DSOPP_HIP_STATUS_OOB, cand = DSOPP_HIP_STATUS_OOB;
  uint8_t fej_bit = 1;
  S patch_k = S(0);
  double hrow[kBlk], srow[kBlk];  // BACKSUB: Schur row block / pose step block of this lane's frame slot tt = k
#pragma unroll
  for (int c = 0; c < kBlk; ++c) hrow[c] = srow[c] = 0;
  const size_t plane = ublkPlane(be.cap);
  {
    const ItemWords cur = nxt;
    flg = cur.flg;
    u = cur.u;
    v = cur.v;
    idepth_d = cur.idepth_d;
    idepth_step_d = cur.idepth_step_d;
    status = cur.status;
    cand = cur.cand;
    patch_k = cur.patch_k;
    fej_bit = cur.fej_bit;
    idepth_fej_d = cur.idepth_fej_d;
  }
  if (inb) {
    if (BACKSUB) {
      bd_d = g_b_d[i];
      inv_hdd_d = g_inv_hdd[i];
      if (k < prm.F && (k == be.r || ((be.conn_mask >> k) & 1u))) {
        const auto src = g_ublk + (static_cast<size_t>(prm.ublk_read) * kMaxFrames + k) * plane + static_cast<size_t>(i) * kUblk;
#pragma unroll
        for (int c = 0; c < kBlk; ++c) {
          hrow[c] = src[c];
          srow[c] = g_step[kBlk * k + c];
        }
      }
    }
  }
  bool active = inb && !((flg & kFlagMarginalized) && !(flg & kFlagToMarginalize));  // evaluate_jacobians.hpp:83-85
  const bool accumulate = active && (prm.for_marginalized ? (flg & kFlagToMarginalize) != 0 : (flg & kFlagMarginalized) == 0);
  bool fej_ok = fej_bit != 0;
  if (!active) {
    status = DSOPP_HIP_STATUS_OOB;
    cand = DSOPP_HIP_STATUS_OOB;
  }
  if (BACKSUB) {
    // calculateIdepths — hessian_block_evaluation.hpp:238-263: lane k of the item takes the frame blocks k, k+8, ... of
    // h_p^T step; the 8 partial dot products are summed across the item's lanes
    double d = 0;
    const bool upd = active && !(flg & (kFlagMarginalized | kFlagIllConditioned)) && (!prm.gate_on_pending || c_pending);
#pragma unroll
    for (int c = 0; c < kBlk; ++c) d += hrow[c] * srow[c];
    if (upd) {
      for (int tt = k + kPat; tt < prm.F; tt += kPat) {  // windows of more than 8 frames
        if (tt != be.r && !((be.conn_mask >> tt) & 1u)) continue;
        const auto src = g_ublk + (static_cast<size_t>(prm.ublk_read) * kMaxFrames + tt) * plane + static_cast<size_t>(i) * kUblk;
#pragma unroll
        for (int c = 0; c < kBlk; ++c) d += src[c] * g_step[kBlk * tt + c];
      }
    }
    d = sum8(d);
    if (upd) {
      idepth_step_d = -((bd_d - d) * (1.0 / (1.0 + lam)) * inv_hdd_d);
      if (k == 0 && be.owns_landmark_sums) g_idepth_step[i] = idepth_step_d;
    }
  }
  SWEEP_STAMP(1);
  if (kStamps && prm.dbg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SWEEP_STAMP(2);
  const S idepth = static_cast<S>(idepth_d + idepth_step_d);
  const S Wr = S(be.width_r), Hr = S(be.height_r), Wt = S(be.width_t), Ht = S(be.height_t);

  // ---- reprojection of this lane's pattern pixel
  const S pu = u + S(ox), pv = v + S(oy);
  S tu, tv;
  bool ok = active && validIdepth(idepth) && insideROI(pu, pv, Wr, Hr);
  if (!LIN || FEJ) {
    // reproject without Jacobians — camera_reproject.hpp:270-293
    const S x = S(Pm->M[0]) * pu + S(Pm->M[1]) * pv + (S(Pm->M[2]) + S(Pm->M[3]) * idepth);
    const S y = S(Pm->M[4]) * pu + S(Pm->M[5]) * pv + (S(Pm->M[6]) + S(Pm->M[7]) * idepth);
    const S z = S(Pm->M[8]) * pu + S(Pm->M[9]) * pv + (S(Pm->M[10]) + S(Pm->M[11]) * idepth);
    const S iz = fastRcp(z);
    tu = x * iz;
    tv = y * iz;
    ok = ok && (z > S(0));
  } else {
    // non-FEJ linearisation: positions come from the Jacobian path — camera_reproject.hpp:323-333
    const S X = S(Pm->U[0]) * pu + S(Pm->U[1]) * pv + (S(Pm->U[2]) + S(Pm->U[3]) * idepth);
    const S Y = S(Pm->U[4]) * pu + S(Pm->U[5]) * pv + (S(Pm->U[6]) + S(Pm->U[7]) * idepth);
    const S Z = S(Pm->U[8]) * pu + S(Pm->U[9]) * pv + (S(Pm->U[10]) + S(Pm->U[11]) * idepth);
    tu = (S(Pm->fxt) * X + S(Pm->cxt) * Z) / Z;
    tv = (S(Pm->fyt) * Y + S(Pm->cyt) * Z) / Z;
    ok = ok && (Z > S(0));
  }
  ok = ok && insideROI(tu, tv, Wt, Ht);

  // ---- the next group's item words go out in front of this group's texel gather (in-order return: they land with it)
  if (grp + 1 < be.n_groups) {  // (wave-uniform)
    const int in = i + kItemsPerBlock;
    nxt = loadItemWords(in, in < be.n_res);
    asm volatile("" ::: "memory");  // (keeps the compiler from sinking these loads to their uses in the next iteration)
  }
  // ---- bilinear gather of the stored (I, Ix, Iy) triplet + mask lookup at the rounded position
  // (pixel_map.hpp:20-40, camera_mask.hpp:64-66); a lane only touches the image when its own pixel is inside the ROI
  S sI = S(0), sIx = S(0), sIy = S(0);
  double *const gram_rows = gram_lds + (LIN ? (threadIdx.x >> 6) * 64 * kGramStride : 0);
  if (!LIN && be.iplane_t != nullptr) {
    if (ok) {
      // residual-only sweep: one word per pixel from the tiled intensity plane (mask in the lowest mantissa bit) instead of
      // four texels of which only {I, mask} are used
      const int ix = static_cast<int>(tu), iy = static_cast<int>(tv);
      const S dx = tu - static_cast<S>(ix), dy = tv - static_cast<S>(iy);
      const S dxdy = dx * dy;
      const S w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = S(1) - dx - dy + dxdy;
      const int rx = static_cast<int>(floor(tu + S(0.5))) - ix, ry = static_cast<int>(floor(tv + S(0.5))) - iy;
      const int tiles = be.itiles_t;
      if constexpr (std::is_same<S, double>::value) {
        // 8-byte words, 4 x 2 pixels per 64-byte tile
        const auto ip = (GlobalPtr<const unsigned long long>)be.iplane_t;
        auto at = [&](int x, int y) { return ip[(static_cast<size_t>(y >> 1) * tiles + (x >> 2)) * 8 + ((y & 1) << 2) + (x & 3)]; };
        const unsigned long long b00 = at(ix, iy), b10 = at(ix + 1, iy), b01 = at(ix, iy + 1), b11 = at(ix + 1, iy + 1);
        const unsigned long long bm = ry ? (rx ? b11 : b01) : (rx ? b10 : b00);
        ok = (bm & 1ull) != 0;
        auto val = [](unsigned long long b) { return __longlong_as_double(static_cast<long long>(b & ~1ull)); };
        sI = w11 * val(b11) + w01 * val(b01) + w10 * val(b10) + w00 * val(b00);
      } else {
        // 4-byte words, 4 x 4 pixels per 64-byte tile (f32 storage: the texel gather moved 3.45 x the algorithmic bytes of this sweep)
        const auto ip = (GlobalPtr<const unsigned>)be.iplane_t;
        auto at = [&](int x, int y) { return ip[(static_cast<size_t>(y >> 2) * tiles + (x >> 2)) * 16 + ((y & 3) << 2) + (x & 3)]; };
        const unsigned b00 = at(ix, iy), b10 = at(ix + 1, iy), b01 = at(ix, iy + 1), b11 = at(ix + 1, iy + 1);
        const unsigned bm = ry ? (rx ? b11 : b01) : (rx ? b10 : b00);
        ok = (bm & 1u) != 0;
        auto val = [](unsigned b) { return __uint_as_float(b & ~1u); };
        sI = w11 * val(b11) + w01 * val(b01) + w10 * val(b10) + w00 * val(b00);
      }
    }
  } else {
    // the four texels are requested ...
    Texel<S> t00{}, t10{}, t01{}, t11{};
    const int ix = static_cast<int>(tu), iy = static_cast<int>(tv);
    if (ok) {
      const Texel<S> *__restrict__ img = static_cast<const Texel<S> *>(be.texels_t);
      const int W = be.width_t;
      const Texel<S> *p = img + static_cast<size_t>(iy) * W + ix;
      t00 = loadTexel(p);
      t10 = loadTexel(p + 1);
      t01 = loadTexel(p + W);
      t11 = loadTexel(p + W + 1);
    }
    // ... and while they travel the wave contracts the previous group's Gram rows (gramContract)
    if (LIN && grp > 0) {
      __builtin_amdgcn_sched_barrier(0);
      gramContract(gram_rows, gram);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ok) {
      const S dx = tu - static_cast<S>(ix), dy = tv - static_cast<S>(iy);
      const S dxdy = dx * dy;
      const S w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = S(1) - dx - dy + dxdy;
      const int rx = static_cast<int>(floor(tu + S(0.5))) - ix, ry = static_cast<int>(floor(tv + S(0.5))) - iy;
      const S m = ry ? (rx ? t11.mask : t01.mask) : (rx ? t10.mask : t00.mask);
      ok = (m != S(0));
      sI = w11 * t11.I + w01 * t01.I + w10 * t10.I + w00 * t00.I;
      if (LIN) {
        sIx = w11 * t11.Ix + w01 * t01.Ix + w10 * t10.Ix + w00 * t00.Ix;
        sIy = w11 * t11.Iy + w01 * t01.Iy + w10 * t10.Iy + w00 * t00.Iy;
      }
    }
  }
  SWEEP_STAMP(3);
  const PairConst DSOPP_CONSTANT *Pj = &P;  // constants of the residual and of the Jacobian path: fetched from here on
  asm volatile("" : "+s"(Pj) : "v"(tu));
  const int shift = (threadIdx.x & 63) & ~7;
  if (FEJ && LIN && WFEJ) {
    // firstEstimateJacobians_ for this item (first_estimate_jacobians.hpp:30-62, as fejKernel): all 8 pattern pixels reproject
    // with valid Jacobians at the snapshot inverse depth, the pattern's extent lies inside the reference image
    const S idf = static_cast<S>(idepth_d);
    const S X = S(Pj->U[0]) * pu + S(Pj->U[1]) * pv + (S(Pj->U[2]) + S(Pj->U[3]) * idf);
    const S Y = S(Pj->U[4]) * pu + S(Pj->U[5]) * pv + (S(Pj->U[6]) + S(Pj->U[7]) * idf);
    const S Z = S(Pj->U[8]) * pu + S(Pj->U[9]) * pv + (S(Pj->U[10]) + S(Pj->U[11]) * idf);
    const S tuf = (S(Pj->fxt) * X + S(Pj->cxt) * Z) / Z, tvf = (S(Pj->fyt) * Y + S(Pj->cyt) * Z) / Z;
    const bool okf = (Z > S(0)) && insideROI(tuf, tvf, Wt, Ht);
    const unsigned long long fmask = __ballot(okf);
    const bool item_ok = (((fmask >> shift) & 0xFFull) == 0xFFull) && validIdepth(idf) && insideROI(u - S(2), v - S(2), Wr, Hr) &&
                         insideROI(u + S(2), v + S(2), Wr, Hr);
    if (prm.use_fej_flag) fej_ok = item_ok;
    if (active && k == 0) {
      glb(const_cast<uint8_t *>(static_cast<const uint8_t *>(be.fej_valid)))[i] = item_ok ? 1 : 0;
      glb(const_cast<double *>(static_cast<const double *>(be.idepth_fej)))[i] = idepth_d;  // every connected target writes the same value
    }
  }
  // success of the item = all 8 pixels fine (and the FEJ validity bit)
  const unsigned long long okmask = __ballot(ok);
  const bool success = active && fej_ok && (((okmask >> shift) & 0xFFull) == 0xFFull);

  if (active && !success) cand = DSOPP_HIP_STATUS_OOB;  // evaluate_jacobians.hpp:111-113
  const bool evaluate = success && status == DSOPP_HIP_STATUS_OK;
  if (evaluate) cand = DSOPP_HIP_STATUS_OK;

  // ---- residual, Huber on the norm of the 8-vector — evaluate_jacobians.hpp:124-146
  const S res = evaluate ? (sI - S(Pj->b_t)) - S(Pj->s) * (patch_k - S(Pj->b_r)) : S(0);
  const double r2 = sum8(static_cast<double>(res * res));
  double wgt = 1.0, energy = 0.5 * r2;
  if (HUBER) {
    const double sig = prm.sigma_huber;
    if (r2 > sig * sig) {
      const double inv_nrm = fastRsqrt(r2);
      const double nrm = r2 * inv_nrm;
      wgt = sig * inv_nrm;
      energy = sig * nrm - 0.5 * sig * sig;
    }
  }
  if (!evaluate) energy = 0;
  SWEEP_STAMP(10);

  if (LIN) {
    double gj[kBlk + 2];  // w * g (8), then hdd, bd contributions
    double ga[kBlk], gb[kBlk];  // this pixel's rows of the pair's Gram accumulation (gramAccumulate): zero unless evaluated
#pragma unroll
    for (int a = 0; a < kBlk + 2; ++a) gj[a] = 0;
#pragma unroll
    for (int a = 0; a < kBlk; ++a) ga[a] = gb[a] = 0;
    if (evaluate) {
      // geometric Jacobians at the linearisation point (FEJ: idepth snapshot) — camera_reproject.hpp:339-365
      const S idj = FEJ ? static_cast<S>(idepth_fej_d) : idepth;
      const S X = S(Pj->U[0]) * pu + S(Pj->U[1]) * pv + (S(Pj->U[2]) + S(Pj->U[3]) * idj);
      const S Y = S(Pj->U[4]) * pu + S(Pj->U[5]) * pv + (S(Pj->U[6]) + S(Pj->U[7]) * idj);
      const S Z = S(Pj->U[8]) * pu + S(Pj->U[9]) * pv + (S(Pj->U[10]) + S(Pj->U[11]) * idj);
      const S fxt = S(Pj->fxt), fyt = S(Pj->fyt);
      const S rho = fastRcp(Z);
      const S b0 = X * rho, b1 = Y * rho;
      const S du_id = fxt * (S(Pj->tl[0]) * rho - S(Pj->tl[2]) * (rho * b0));
      const S dv_id = fyt * (S(Pj->tl[1]) * rho - S(Pj->tl[2]) * (rho * b1));
      const S nid = idj * rho;
      const S Iu = sIx, Iv = sIy;
      // g = [ Iv * d_v_T + Iu * d_u_T (6) , c , 1 ] — evaluate_jacobians.hpp:149-157,176-182
      S g[kBlk];
      const S b0b1 = b0 * b1;
      g[0] = Iu * (fxt * nid);
      g[1] = Iv * (fyt * nid);
      g[2] = Iv * (fyt * (-nid * b1)) + Iu * (fxt * (-nid * b0));
      g[3] = Iv * (fyt * (-(b1 * b1 + S(1)))) + Iu * (fxt * (-b0b1));
      g[4] = Iv * (fyt * b0b1) + Iu * (fxt * (b0 * b0 + S(1)));
      g[5] = Iv * (fyt * b0) + Iu * (fxt * (-b1));
      g[6] = S(Pj->sigma_r) * (patch_k - S(Pj->b_r0));
      g[7] = S(1);
      const double jdd = static_cast<double>(Iu * du_id + Iv * dv_id);  // evaluate_jacobians.hpp:165-174
      const double rk = static_cast<double>(res);
#pragma unroll
      for (int a = 0; a < kBlk; ++a) {
        const double wga = wgt * static_cast<double>(g[a]);
        gj[a] = wga * jdd;
        if (a < kBlk - 1) {
          ga[a] = accumulate ? wga : 0.0;
          gb[a] = static_cast<double>(g[a]);
        }
      }
      ga[kBlk - 1] = accumulate ? wgt * rk : 0.0;
      gb[kBlk - 1] = 1.0;
      if (accumulate) sc[4] += wgt;
      gj[8] = wgt * jdd * jdd;
      gj[9] = wgt * jdd * rk;
    }
    SWEEP_STAMP(11);
    gramPublish(ga, gb, gram_rows);
    SWEEP_STAMP(12);
    // per-item Schur quantities: sum over the 8 pixels; h_p block of target t is w * J_t^T J_d = -u
    // (hessian_block_evaluation.hpp:207-208), zero for invalid residuals (:190-192).  Transposed butterfly: in every step a lane
    // keeps one value of a pair and hands the other to its partner, so the 8 totals end up one per lane (lane k holds entry
    // 4 b1 + 2 b0 + b2 of u, k = b2 b1 b0) after 7 exchanges instead of 24, and the 8 lanes store the row without a select chain.
    {
      const bool b0 = (k & 1) != 0, b1 = (k & 2) != 0, b2 = (k & 4) != 0;
      double ra[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {  // partner 7 - k (row_half_mirror)
        const double keep = b2 ? gj[2 * j + 1] : gj[2 * j], send = b2 ? gj[2 * j] : gj[2 * j + 1];
        ra[j] = keep + dppMove<0x141>(send);
      }
      double rb[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) {  // partner k ^ 1
        const double keep = b0 ? ra[2 * m + 1] : ra[2 * m], send = b0 ? ra[2 * m] : ra[2 * m + 1];
        rb[m] = keep + dppMove<0xB1>(send);
      }
      const double u_tot = (b1 ? rb[1] : rb[0]) + dppMove<0x4E>(b1 ? rb[0] : rb[1]);  // partner k ^ 2
      // (H_dd, b_d) partials: lanes 0..3 end with the first, lanes 4..7 with the second
      double hb = (b2 ? gj[9] : gj[8]) + dppMove<0x141>(b2 ? gj[8] : gj[9]);
      hb += dppMove<0xB1>(hb);
      hb += dppMove<0x4E>(hb);
      if (active) {
        const auto dst = g_ublk + (static_cast<size_t>(prm.ublk_write) * kMaxFrames + be.t) * plane + static_cast<size_t>(i) * kUblk;
        dst[((k & 3) << 1) | (k >> 2)] = -u_tot;
        if ((k & 3) == 0) dst[8 + (k >> 2)] = hb;
      }
    }
  }
  SWEEP_STAMP(13);
  // NEW_EVALUATION_POINT bookkeeping (lane 0 of the item)
  if (active && k == 0) {
    g_energy[i] = energy;
    g_cand[i] = cand;
    if (accumulate) {
      sc[0] += energy;
      sc[1] += energy > 0 ? 1.0 : 0.0;
    }
    if ((!LIN || BACKSUB || prm.external_backsub) && be.owns_landmark_sums) {
      // per-landmark norms of acceptStep (problem.hpp:379-381), counted once per landmark
      // (the opening round of the fused loop has no step yet: slot 46 carries sum idepth^2, the initial state norm)
      sc[2] += (prm.gate_on_pending && !c_pending) ? idepth_d * idepth_d : idepth_step_d * idepth_step_d;
      sc[3] += idepth_d * idepth_step_d;
    }
  }
  }  // groups
  // the last group's rows (a padding entry of the XCD-banded launch order has no group: nothing was published, its sums stay zero)
  SWEEP_STAMP(14);
  if (LIN && be.n_groups > 0) gramContract(gram_lds + (threadIdx.x >> 6) * 64 * kGramStride, gram);
  SWEEP_STAMP(4);
  double *out = partials + static_cast<size_t>(be.partial_row) * kPartial;
  gramScalarsStore<LIN>(gram, sc, scalar_lds, out);
  SWEEP_STAMP(5);
  SWEEP_STAMP(6);
  if (kStamps && prm.dbg && threadIdx.x == 0) {
    atomicMin(reinterpret_cast<unsigned long long *>(prm.dbg + 8), static_cast<unsigned long long>(wall_clock64()));
    atomicMax(reinterpret_cast<unsigned long long *>(prm.dbg + 9), static_cast<unsigned long long>(wall_clock64()));
  }
}

}  // namespace dsopp_hip
