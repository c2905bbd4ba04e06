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
  for (int i = 0; i < 3; ++i) {
    U[4 * i + 0] = T.R[3 * i + 0] * ifx;
    U[4 * i + 1] = T.R[3 * i + 1] * ify;
    U[4 * i + 2] = T.R[3 * i + 0] * k02 + T.R[3 * i + 1] * k12 + T.R[3 * i + 2];
    U[4 * i + 3] = T.t[i];
  }
  if (M) {
    for (int j = 0; j < 4; ++j) {
      M[0 + j] = ft.fx * U[0 + j] + ft.cx * U[8 + j];
      M[4 + j] = ft.fy * U[4 + j] + ft.cy * U[8 + j];
      M[8 + j] = U[8 + j];
    }
  }
}

__device__ inline double linearisationScale(const FrameDev *frames, const WindowState *st, int r, int t) {
  return (frames[t].exposure / frames[r].exposure) * exp(st->ab0[t][0] - st->ab0[r][0]);
}

/** evaluate_jacobians.hpp:36-66 (per-pair prologue) + first_estimate_jacobians.hpp:22-31 */
__device__ inline void computePairConst(const FrameDev *frames, const WindowState *st, PairConst *pc, int r, int t, int F, bool fej) {
  PairConst &P = pc[r * kMaxFrames + t];
  const FrameDev &fr = frames[r];
  const FrameDev &ft = frames[t];
  const bool valid = (r != t) && fr.status[t] != nullptr;
  P.valid = valid ? 1 : 0;
  if (!valid) return;
  Rigid Tr0, Tt0;
  for (int i = 0; i < 9; ++i) {
    Tr0.R[i] = st->T0_R[r][i];
    Tt0.R[i] = st->T0_R[t][i];
  }
  for (int i = 0; i < 3; ++i) {
    Tr0.t[i] = st->T0_t[r][i];
    Tt0.t[i] = st->T0_t[t][i];
  }
  const Rigid T_tr0 = rigidMul(rigidInverse(Tt0), Tr0);
  double xr[6], mxt[6];
  for (int i = 0; i < 6; ++i) {
    xr[i] = st->eps[r][i] + st->step[r][i];
    mxt[i] = -(st->eps[t][i] + st->step[t][i]);
  }
  const Rigid T_tr = rigidMul(rigidExp(mxt), rigidMul(T_tr0, rigidExp(xr)));
  const double a_r = st->ab0[r][0] + st->eps[r][6] + st->step[r][6];
  const double b_r = st->ab0[r][1] + st->eps[r][7] + st->step[r][7];
  const double a_t = st->ab0[t][0] + st->eps[t][6] + st->step[t][6];
  const double b_t = st->ab0[t][1] + st->eps[t][7] + st->step[t][7];
  P.s = (ft.exposure / fr.exposure) * exp(a_t - a_r);
  P.b_t = b_t;
  P.b_r = b_r;
  double Ucur[12];
  buildProjectionMatrices(T_tr, fr, ft, Ucur, P.M);
  const Rigid &Tlin = fej ? T_tr0 : T_tr;
  buildProjectionMatrices(Tlin, fr, ft, P.U, nullptr);
  for (int i = 0; i < 3; ++i) P.tl[i] = Tlin.t[i];
  rigidAdj(Tlin, P.Adj);
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
    P.s0 = P.s;
    P.b_r0 = b_r;
    P.sigma_r = P.s;
  }
}

__global__ void pairSetupKernel(const FrameDev *frames, const WindowState *st, PairConst *pc, int F, int fej) {
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

// ---------------------------------------------------------------------------------------------------------------
// geometry helpers (templated on the evaluation scalar S)
// ---------------------------------------------------------------------------------------------------------------
template <typename S>
__device__ __forceinline__ bool insideROI(S u, S v, S width, S height) {
  // CameraModelBase::insideCameraROI — camera_model_base.hpp:52-60 (border 4)
  return (u >= S(4)) && (v >= S(4)) && (u <= width - S(5)) && (v <= height - S(5));
}
template <typename S>
__device__ __forceinline__ bool validIdepth(S idepth) {
  // CameraModelBase::validIdepth — camera_model_base.hpp:67-74
  return idepth > S(-1e-4) && idepth < S(1.0 / 0.001 + 1e1);
}

/** firstEstimateJacobians_ (first_estimate_jacobians.hpp:14-71): validity of the reprojection at the linearisation point
 *  + idepth snapshot.  The geometric Jacobians themselves are NOT stored (they are recomputed from the snapshot). */
template <typename S>
__global__ void __launch_bounds__(kSweepThreads) fejKernel(const FrameDev *__restrict__ frames, const PairConst *__restrict__ pc,
                                                           const SweepBlock *__restrict__ table) {
  const SweepBlock be = table[blockIdx.x];
  const FrameDev &fr = frames[be.r];
  const FrameDev &ft = frames[be.t];
  const PairConst &P = pc[be.r * kMaxFrames + be.t];
  const int i = be.offset + threadIdx.x;
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
};

/**
 * evaluateJacobians<S, SE3, Pinhole, 8, PixelMap, 1, FEJ, OPT_IDEPTHS, LIN, true, HUBER> fused with
 * evaluateLinearSystemPosePoseBlock and the per-(landmark,target) part of ...SchurComplement
 * (PBA_INT/evaluate_jacobians.hpp:20-202, hessian_block_evaluation.hpp:38-90,198-212).
 *   LIN = false: residual-only sweep (calculateEnergy).   LIN = true: linearisation sweep.
 */
template <typename S, bool LIN, bool FEJ, bool HUBER>
__global__ void __launch_bounds__(kSweepThreads) sweepKernel(const FrameDev *__restrict__ frames, const PairConst *__restrict__ pc,
                                                             const SweepBlock *__restrict__ table, double *__restrict__ partials,
                                                             SweepParams prm) {
  __shared__ double red_lds[(kSweepThreads / 64) * kPartial];
  const SweepBlock be = table[blockIdx.x];
  const FrameDev &fr = frames[be.r];
  const FrameDev &ft = frames[be.t];
  const PairConst &P = pc[be.r * kMaxFrames + be.t];
  const int i = be.offset + threadIdx.x;

  double acc[kPartial];
#pragma unroll
  for (int k = 0; k < kPartial; ++k) acc[k] = 0;

  bool active = i < fr.n_res[be.t];
  uint8_t flg = 0;
  if (active) {
    flg = fr.flags[i];
    active = !((flg & kFlagMarginalized) && !(flg & kFlagToMarginalize));  // evaluate_jacobians.hpp:83-85
  }
  if (active) {
    const bool accumulate = prm.for_marginalized ? (flg & kFlagToMarginalize) != 0 : (flg & kFlagMarginalized) == 0;
    const S u = static_cast<S>(fr.uv[2 * i]), v = static_cast<S>(fr.uv[2 * i + 1]);
    const S idepth = static_cast<S>(fr.idepth[i] + fr.idepth_step[i]);
    const uint8_t status = fr.status[be.t][i];
    const S Wr = S(fr.width), Hr = S(fr.height), Wt = S(ft.width), Ht = S(ft.height);

    bool success = validIdepth(idepth) && insideROI(u - S(2), v - S(2), Wr, Hr) && insideROI(u + S(2), v + S(2), Wr, Hr);
    S tu[kPat], tv[kPat];
    if (!LIN || FEJ) {
      // reproject without Jacobians — camera_reproject.hpp:270-293
      const S M0 = S(P.M[0]), M1 = S(P.M[1]), M4 = S(P.M[4]), M5 = S(P.M[5]), M8 = S(P.M[8]), M9 = S(P.M[9]);
      const S cx = S(P.M[2]) + S(P.M[3]) * idepth, cy = S(P.M[6]) + S(P.M[7]) * idepth, cz = S(P.M[10]) + S(P.M[11]) * idepth;
#pragma unroll
      for (int k = 0; k < kPat; ++k) {
        const S pu = u + S(kPatX[k]), pv = v + S(kPatY[k]);
        const S x = M0 * pu + M1 * pv + cx, y = M4 * pu + M5 * pv + cy, z = M8 * pu + M9 * pv + cz;
        tu[k] = x / z;
        tv[k] = y / z;
        success = success && (z > S(0));
      }
    } else {
      // non-FEJ linearisation: positions come from the Jacobian path — camera_reproject.hpp:323-333
      const S U0 = S(P.U[0]), U1 = S(P.U[1]), U4 = S(P.U[4]), U5 = S(P.U[5]), U8 = S(P.U[8]), U9 = S(P.U[9]);
      const S cX = S(P.U[2]) + S(P.U[3]) * idepth, cY = S(P.U[6]) + S(P.U[7]) * idepth, cZ = S(P.U[10]) + S(P.U[11]) * idepth;
#pragma unroll
      for (int k = 0; k < kPat; ++k) {
        const S pu = u + S(kPatX[k]), pv = v + S(kPatY[k]);
        const S X = U0 * pu + U1 * pv + cX, Y = U4 * pu + U5 * pv + cY, Z = U8 * pu + U9 * pv + cZ;
        tu[k] = (S(P.fxt) * X + S(P.cxt) * Z) / Z;
        tv[k] = (S(P.fyt) * Y + S(P.cyt) * Z) / Z;
        success = success && (Z > S(0));
      }
    }
#pragma unroll
    for (int k = 0; k < kPat; ++k) success = success && insideROI(tu[k], tv[k], Wt, Ht);
    if (FEJ && prm.use_fej_flag) success = success && (fr.fej_valid[be.t][i] != 0);  // evaluate_jacobians.hpp:94

    // bilinear gather of the stored (I, Ix, Iy) triplets + mask lookup at the rounded position
    // (pixel_map.hpp:20-40, camera_mask.hpp:64-66).  Only issued for geometrically valid patterns: the ROI test
    // guarantees the 2x2 footprints are inside the image.
    S sI[kPat], sIx[kPat], sIy[kPat];
    if (success) {
      const Texel<S> *__restrict__ img = static_cast<const Texel<S> *>(ft.texels);
      const int W = ft.width;
      bool mask_ok = true;
#pragma unroll
      for (int k = 0; k < kPat; ++k) {
        const int ix = static_cast<int>(tu[k]), iy = static_cast<int>(tv[k]);
        const S dx = tu[k] - static_cast<S>(ix), dy = tv[k] - static_cast<S>(iy);
        const S dxdy = dx * dy;
        const Texel<S> *p = img + static_cast<size_t>(iy) * W + ix;
        const S w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = S(1) - dx - dy + dxdy;
        const int rx = static_cast<int>(floor(tu[k] + S(0.5))) - ix, ry = static_cast<int>(floor(tv[k] + S(0.5))) - iy;
        const S m = ry ? (rx ? p[W + 1].mask : p[W].mask) : (rx ? p[1].mask : p[0].mask);
        mask_ok = mask_ok && (m != S(0));
        sI[k] = w11 * p[W + 1].I + w01 * p[W].I + w10 * p[1].I + w00 * p[0].I;
        if (LIN) {
          sIx[k] = w11 * p[W + 1].Ix + w01 * p[W].Ix + w10 * p[1].Ix + w00 * p[0].Ix;
          sIy[k] = w11 * p[W + 1].Iy + w01 * p[W].Iy + w10 * p[1].Iy + w00 * p[0].Iy;
        }
      }
      success = success && mask_ok;
    }

    uint8_t cand = fr.cand[be.t][i];
    if (!success) cand = DSOPP_HIP_STATUS_OOB;  // evaluate_jacobians.hpp:111-113
    double energy = 0;
    double uvec[kBlk];
    double hdd = 0, bd = 0;
#pragma unroll
    for (int a = 0; a < kBlk; ++a) uvec[a] = 0;

    if (success && status == DSOPP_HIP_STATUS_OK) {
      cand = DSOPP_HIP_STATUS_OK;
      const S s = S(P.s), b_t = S(P.b_t), b_r = S(P.b_r);
      S res[kPat];
      S patch[kPat];
      double r2 = 0;
#pragma unroll
      for (int k = 0; k < kPat; ++k) {
        patch[k] = static_cast<S>(fr.patch[kPat * i + k]);
        res[k] = (sI[k] - b_t) - s * (patch[k] - b_r);  // evaluate_jacobians.hpp:124-135
        r2 += static_cast<double>(res[k] * res[k]);
      }
      // Huber on the norm of the 8-vector — evaluate_jacobians.hpp:136-146
      double wgt = 1.0;
      energy = 0.5 * r2;
      if (HUBER) {
        const double sig = prm.sigma_huber;
        if (r2 > sig * sig) {
          const double nrm = sqrt(r2);
          wgt = sig / nrm;
          energy = sig * nrm - 0.5 * sig * sig;
        }
      }
      if (LIN) {
        // geometric Jacobians at the linearisation point (FEJ: idepth snapshot) — camera_reproject.hpp:339-365
        const S idj = FEJ ? static_cast<S>(fr.idepth_fej[i]) : idepth;
        const S U0 = S(P.U[0]), U1 = S(P.U[1]), U4 = S(P.U[4]), U5 = S(P.U[5]), U8 = S(P.U[8]), U9 = S(P.U[9]);
        const S cX = S(P.U[2]) + S(P.U[3]) * idj, cY = S(P.U[6]) + S(P.U[7]) * idj, cZ = S(P.U[10]) + S(P.U[11]) * idj;
        const S fxt = S(P.fxt), fyt = S(P.fyt), t0 = S(P.tl[0]), t1 = S(P.tl[1]), t2 = S(P.tl[2]);
        const S csc = S(P.sigma_r), b_r0 = S(P.b_r0);
        double G[36];
#pragma unroll
        for (int e = 0; e < 36; ++e) G[e] = 0;
        double q[kBlk];
#pragma unroll
        for (int a = 0; a < kBlk; ++a) q[a] = 0;
#pragma unroll
        for (int k = 0; k < kPat; ++k) {
          const S pu = u + S(kPatX[k]), pv = v + S(kPatY[k]);
          const S X = U0 * pu + U1 * pv + cX, Y = U4 * pu + U5 * pv + cY, Z = U8 * pu + U9 * pv + cZ;
          const S rho = S(1) / Z;
          const S b0 = X * rho, b1 = Y * rho;
          const S du_id = fxt * (t0 * rho - t2 * (rho * b0));
          const S dv_id = fyt * (t1 * rho - t2 * (rho * b1));
          const S nid = idj * rho;
          const S Iu = sIx[k], Iv = sIy[k];
          // g = [ Iv * d_v_T + Iu * d_u_T (6) , c , 1 ] — evaluate_jacobians.hpp:149-157,176-182
          S g[kBlk];
          const S b0b1 = b0 * b1;
          g[0] = Iu * (fxt * nid);
          g[1] = Iv * (fyt * nid);
          g[2] = Iv * (fyt * (-nid * b1)) + Iu * (fxt * (-nid * b0));
          g[3] = Iv * (fyt * (-(b1 * b1 + S(1)))) + Iu * (fxt * (-b0b1));
          g[4] = Iv * (fyt * b0b1) + Iu * (fxt * (b0 * b0 + S(1)));
          g[5] = Iv * (fyt * b0) + Iu * (fxt * (-b1));
          g[6] = csc * (patch[k] - b_r0);
          g[7] = S(1);
          const S jd = Iu * du_id + Iv * dv_id;  // evaluate_jacobians.hpp:165-174
          const double rk = static_cast<double>(res[k]), jdd = static_cast<double>(jd);
          int e = 0;
#pragma unroll
          for (int a = 0; a < kBlk; ++a) {
            const double ga = static_cast<double>(g[a]);
#pragma unroll
            for (int b = a; b < kBlk; ++b) G[e++] += ga * static_cast<double>(g[b]);
            q[a] += ga * rk;
            uvec[a] += ga * jdd;
          }
          hdd += jdd * jdd;
          bd += jdd * rk;
        }
#pragma unroll
        for (int a = 0; a < kBlk; ++a) uvec[a] *= wgt;
        hdd *= wgt;
        bd *= wgt;
        if (accumulate) {
#pragma unroll
          for (int e = 0; e < 36; ++e) acc[e] = wgt * G[e];
#pragma unroll
          for (int a = 0; a < kBlk; ++a) acc[36 + a] = wgt * q[a];
        }
      }
    }
    // NEW_EVALUATION_POINT bookkeeping
    fr.energy[be.t][i] = energy;
    fr.cand[be.t][i] = cand;
    if (accumulate) {
      acc[44] = energy;
      acc[45] = energy > 0 ? 1.0 : 0.0;
    }
    if (LIN) {
      // h_p block of target t is w * J_t^T J_d = -u (hessian_block_evaluation.hpp:207-208); zero for invalid residuals (:190-192)
      double *dst = fr.ublk + (static_cast<size_t>(be.t) * fr.cap + i) * kUblk;
#pragma unroll
      for (int a = 0; a < kBlk; ++a) dst[a] = -uvec[a];
      dst[8] = hdd;
      dst[9] = bd;
    }
  }
  blockSum<kPartial, kSweepThreads>(acc, red_lds);
  if (threadIdx.x == 0) {
    double *out = partials + static_cast<size_t>(blockIdx.x) * kPartial;
#pragma unroll
    for (int k = 0; k < kPartial; ++k) out[k] = acc[k];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Schur complement: per-landmark finalisation + K x K rank-1 accumulation
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSchurLandmarks = 64;
constexpr int kSchurThreads = 256;

/**
 * evaluateLinearSystemPoseDepthSchurComplement — hessian_block_evaluation.hpp:169-236.
 * Phase 1 (one thread per landmark): gather the h_p blocks written by the sweep, form the reference-frame block
 * sum_t T^T u, H_dd, b_d, invert, store the landmark caches (b_idepth_block, inv_hessian_idepth_idepth, ill_conditioned).
 * Phase 2 (whole block): H_schur += sum_l inv_l h_l h_l^T (upper triangle), b_schur += sum_l inv_l bd_l h_l.
 */
__global__ void __launch_bounds__(kSchurThreads) schurKernel(const FrameDev *__restrict__ frames, const PairConst *__restrict__ pc,
                                                             const SchurBlock *__restrict__ table, double *__restrict__ Hsc,
                                                             double *__restrict__ bsc, int F, int for_marginalized) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *hrow = reinterpret_cast<double *>(smem_raw);  // [kSchurLandmarks][K]
  const int K = kBlk * F;
  double *wgt = hrow + kSchurLandmarks * K;  // inv per landmark (0 = excluded)
  double *wbd = wgt + kSchurLandmarks;       // inv * bd
  const SchurBlock be = table[blockIdx.x];
  const FrameDev &fr = frames[be.r];
  const int r = be.r;
  const int l = threadIdx.x;
  if (l < kSchurLandmarks) {
    const int i = be.offset + l;
    double inv = 0, ibd = 0;
    double *row = hrow + l * K;
    for (int k = 0; k < K; ++k) row[k] = 0;
    if (i < fr.n) {
      uint8_t flg = fr.flags[i];
      const bool take = for_marginalized ? (flg & kFlagToMarginalize) != 0 : (flg & kFlagMarginalized) == 0;
      if (take) {
        double hr[kBlk];
#pragma unroll
        for (int a = 0; a < kBlk; ++a) hr[a] = 0;
        double hdd = 0, bd = 0;
        for (int t = 0; t < F; ++t) {
          if (t == r || fr.status[t] == nullptr) continue;
          const PairConst &P = pc[r * kMaxFrames + t];
          const double *src = fr.ublk + (static_cast<size_t>(t) * fr.cap + i) * kUblk;
          double ht[kBlk];
#pragma unroll
          for (int a = 0; a < kBlk; ++a) {
            ht[a] = src[a];
            row[kBlk * t + a] = ht[a];
          }
          hdd += src[8];
          bd += src[9];
          // reference block += T^T u with u = -ht, T = blockdiag(Adj, 1, s0)
#pragma unroll
          for (int a = 0; a < 6; ++a) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 6; ++k) s += P.Adj[6 * k + a] * ht[k];
            hr[a] -= s;
          }
          hr[6] -= ht[6];
          hr[7] -= P.s0 * ht[7];
        }
        double *dst = fr.ublk + (static_cast<size_t>(r) * fr.cap + i) * kUblk;
#pragma unroll
        for (int a = 0; a < kBlk; ++a) {
          row[kBlk * r + a] = hr[a];
          dst[a] = hr[a];
        }
        fr.b_d[i] = bd;
        const double kIdepthNullSpaceThreshold = 1e-15;
        if (hdd > kIdepthNullSpaceThreshold) {
          if (for_marginalized && fr.fixed) hdd += 1e8;  // kScaleNullspaceRegularizer
          inv = 1.0 / hdd;
          fr.inv_hdd[i] = inv;
          flg &= static_cast<uint8_t>(~kFlagIllConditioned);
          ibd = inv * bd;
        } else {
          flg |= kFlagIllConditioned;
        }
        fr.flags[i] = flg;
      }
    }
    wgt[l] = inv;
    wbd[l] = ibd;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < K * K; idx += kSchurThreads) {
    const int a = idx / K, b = idx % K;
    if (b < a) continue;
    double s = 0;
    for (int ll = 0; ll < kSchurLandmarks; ++ll) s += (wgt[ll] * hrow[ll * K + a]) * hrow[ll * K + b];
    if (s != 0) atomicAdd(&Hsc[a * K + b], s);
  }
  for (int a = threadIdx.x; a < K; a += kSchurThreads) {
    double s = 0;
    for (int ll = 0; ll < kSchurLandmarks; ++ll) s += wbd[ll] * hrow[ll * K + a];
    if (s != 0) atomicAdd(&bsc[a], s);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// assemble + solve (single workgroup)
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSolveThreads = 256;

struct SolveBuffers {
  const double *partials;       // [n_sweep_blocks][kPartial]
  const int *pair_first_block;  // [kMaxFrames*kMaxFrames] first sweep block of the pair (-1 = none)
  const int *pair_num_blocks;
  double *Gpair;                // [F*F][48] (compact, pair index r*F+t): G (36 upper) + q (8)
  double *GT;                   // scratch [F*F][64]: G*T
  double *TGT;                  // scratch [F*F][64]: T^T*G*T
  double *Hpp, *bpp;            // out: system_pose (with priors), K x K / K
  double *Hsc, *bsc;            // in: schur (upper triangle) ; symmetrised in place
  const double *Hm, *bm;        // marginal prior (K x K, K) or nullptr
  double *step;                 // out: K
  double *energy_out;           // out [4]: {sum of landmark energies, n_valid, -, -}
};

/** sums the per-block G/q partials of every pair (deterministic order) */
__device__ inline void reducePairPartials(const double *__restrict__ partials, const int *__restrict__ pair_first_block,
                                          const int *__restrict__ pair_num_blocks, double *__restrict__ Gpair, int F, int tid, int nthreads) {
  for (int idx = tid; idx < F * F * 48; idx += nthreads) {
    const int p = idx / 48, e = idx % 48;
    const int r = p / F, t = p % F;
    const int pi = r * kMaxFrames + t;
    double s = 0;
    if (e < 44) {
      const int first = pair_first_block[pi], cnt = pair_num_blocks[pi];
      for (int b = 0; b < cnt; ++b) s += partials[static_cast<size_t>(first + b) * kPartial + e];
    }
    Gpair[p * 48 + e] = s;
  }
}

/** stand-alone version used when the per-pair sums must be all-reduced across GPUs before the assembly */
__global__ void pairReduceKernel(const double *__restrict__ partials, const int *__restrict__ pair_first_block,
                                 const int *__restrict__ pair_num_blocks, double *__restrict__ Gpair, int F) {
  reducePairPartials(partials, pair_first_block, pair_num_blocks, Gpair, F, blockIdx.x * blockDim.x + threadIdx.x,
                     gridDim.x * blockDim.x);
}

/**
 * evaluateLinearSystemPosePose (hessian_block_evaluation.hpp:96-164) from the per-pair G/q, evaluateLinearSystemPrior
 * (problem.hpp:37-77), calculateStep (problem.hpp:342-361) and NormalLinearSystem::solve
 * (normal_linear_system.cpp:10-16,52-59: Jacobi preconditioner + LDL^T) in one workgroup; also rebuilds the pair
 * constants for the candidate state so the following energy sweep needs no extra launch.
 */
__global__ void __launch_bounds__(kSolveThreads) assembleSolveKernel(const FrameDev *__restrict__ frames, WindowState *st, PairConst *pc,
                                                                     SolveBuffers B, SolveParams prm, int fej, int do_solve,
                                                                     int skip_pair_reduce) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int F = prm.F, K = kBlk * F;
  double *A = reinterpret_cast<double *>(smem_raw);  // K x K
  double *bv = A + K * K;                            // K
  double *pv = bv + K;                               // K preconditioner
  double *col = pv + K;                              // K scratch
  const int tid = threadIdx.x;

  if (!skip_pair_reduce) reducePairPartials(B.partials, B.pair_first_block, B.pair_num_blocks, B.Gpair, F, tid, kSolveThreads);
  __syncthreads();
  // per pair: GT = G*T, TGT = T^T*GT with T = blockdiag(Adj, 1, s0)
  for (int idx = tid; idx < F * F * 64; idx += kSolveThreads) {
    const int p = idx / 64, e = idx % 64, i = e / 8, j = e % 8;
    const int r = p / F, t = p % F, pi = r * kMaxFrames + t;
    const PairConst &P = pc[pi];
    if (!P.valid) continue;
    const double *G = B.Gpair + p * 48;
    double s;
    if (j < 6) {
      s = 0;
      for (int k = 0; k < 6; ++k) s += G[symIdx(i, k)] * P.Adj[6 * k + j];
    } else if (j == 6) {
      s = G[symIdx(i, 6)];
    } else {
      s = G[symIdx(i, 7)] * P.s0;
    }
    B.GT[p * 64 + e] = s;
  }
  __syncthreads();
  for (int idx = tid; idx < F * F * 64; idx += kSolveThreads) {
    const int p = idx / 64, e = idx % 64, i = e / 8, j = e % 8;
    const int r = p / F, t = p % F, pi = r * kMaxFrames + t;
    const PairConst &P = pc[pi];
    if (!P.valid) continue;
    const double *GT = B.GT + p * 64;
    double s;
    if (i < 6) {
      s = 0;
      for (int k = 0; k < 6; ++k) s += P.Adj[6 * k + i] * GT[8 * k + j];
    } else if (i == 6) {
      s = GT[8 * 6 + j];
    } else {
      s = P.s0 * GT[8 * 7 + j];
    }
    B.TGT[p * 64 + e] = s;
  }
  __syncthreads();
  // H_pp, b_pp
  for (int idx = tid; idx < K * K; idx += kSolveThreads) {
    const int a = idx / K, b = idx % K;
    const int fa = a / 8, fb = b / 8, i = a % 8, j = b % 8;
    double s = 0;
    if (fa == fb) {
      for (int t = 0; t < F; ++t) {
        if (t == fa) continue;
        if (pc[fa * kMaxFrames + t].valid) s += B.TGT[(fa * F + t) * 64 + 8 * i + j];
        if (pc[t * kMaxFrames + fa].valid) s += B.Gpair[(t * F + fa) * 48 + symIdx(i, j)];
      }
    } else {
      // H[r,t] = -T^T G (pair fa->fb) ; H[t,r] of pair (fb->fa) transposed = -(G T)
      if (pc[fa * kMaxFrames + fb].valid) s -= B.GT[(fa * F + fb) * 64 + 8 * j + i];  // (T^T G)_{ij} = (G T)_{ji}
      if (pc[fb * kMaxFrames + fa].valid) s -= B.GT[(fb * F + fa) * 64 + 8 * i + j];
    }
    A[idx] = s;
  }
  for (int a = tid; a < K; a += kSolveThreads) {
    const int fa = a / 8, i = a % 8;
    double s = 0;
    for (int t = 0; t < F; ++t) {
      if (t == fa) continue;
      const int prt = fa * kMaxFrames + t, ptr_ = t * kMaxFrames + fa;
      if (pc[prt].valid) {
        // b_r = T^T q
        const double *q = B.Gpair + (fa * F + t) * 48 + 36;
        if (i < 6) {
          for (int k = 0; k < 6; ++k) s += pc[prt].Adj[6 * k + i] * q[k];
        } else if (i == 6) {
          s += q[6];
        } else {
          s += pc[prt].s0 * q[7];
        }
      }
      if (pc[ptr_].valid) s -= B.Gpair[(t * F + fa) * 48 + 36 + i];  // b_t = -q
    }
    bv[a] = s;
  }
  __syncthreads();
  // priors — problem.hpp:39-62
  for (int a = tid; a < K; a += kSolveThreads) {
    const int f = a / 8, i = a % 8;
    if (frames[f].to_marginalize) continue;  // for_marginalized == false
    if (frames[f].fixed) {
      A[a * K + a] += prm.fixed_reg;
      bv[a] += prm.fixed_reg * st->eps[f][i];
    } else if (i >= 6) {
      const double ab = st->ab0[f][i - 6] + st->eps[f][i];
      A[a * K + a] += prm.affine_reg[i - 6];
      bv[a] += prm.affine_reg[i - 6] * ab;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < K * K; idx += kSolveThreads) B.Hpp[idx] = A[idx];
  for (int a = tid; a < K; a += kSolveThreads) B.bpp[a] = bv[a];
  // symmetrise the Schur accumulation (upper triangle was accumulated)
  for (int idx = tid; idx < K * K; idx += kSolveThreads) {
    const int a = idx / K, b = idx % K;
    if (b < a) B.Hsc[idx] = B.Hsc[b * K + a];
  }
  if (!do_solve) return;
  __syncthreads();
  // calculateStep — problem.hpp:347-351
  const double lam = prm.lambda, sc = -1.0 / (1.0 + lam);
  for (int idx = tid; idx < K * K; idx += kSolveThreads) {
    const int a = idx / K, b = idx % K;
    double v = A[idx];
    if (a == b) v += A[idx] * lam;
    if (prm.use_marginal) v += B.Hm[idx];
    v += sc * B.Hsc[idx];
    A[idx] = v;
  }
  __syncthreads();
  for (int a = tid; a < K; a += kSolveThreads) {
    double v = bv[a] + sc * B.bsc[a];
    if (prm.use_marginal) {
      v += B.bm[a];
      double s = 0;
      for (int k = 0; k < K; ++k) s += B.Hm[a * K + k] * st->eps[k / 8][k % 8];
      v += s;
    }
    bv[a] = v;
    pv[a] = 1.0 / sqrt(A[a * K + a] + 10.0);  // jacobiPreconditioner — normal_linear_system.cpp:10-16
  }
  __syncthreads();
  for (int idx = tid; idx < K * K; idx += kSolveThreads) {
    const int a = idx / K, b = idx % K;
    A[idx] = pv[a] * A[idx] * pv[b];
  }
  for (int a = tid; a < K; a += kSolveThreads) bv[a] *= pv[a];
  __syncthreads();
  // LDL^T (right-looking, lower triangle), forward substitution fused; zero pivots are skipped as Eigen::LDLT does
  for (int k = 0; k < K; ++k) {
    const double d = A[k * K + k];
    const bool okp = fabs(d) > 1e-300;
    const double dinv = okp ? 1.0 / d : 0.0;
    for (int i = k + 1 + tid; i < K; i += kSolveThreads) {
      const double aik = A[i * K + k];
      col[i] = aik;             // A(i,k) before scaling
      A[i * K + k] = aik * dinv;  // L(i,k)
    }
    __syncthreads();
    const int rem = K - k - 1;
    for (int idx = tid; idx < rem * rem; idx += kSolveThreads) {
      const int i = k + 1 + idx / rem, j = k + 1 + idx % rem;
      if (j <= i) A[i * K + j] -= A[i * K + k] * col[j];
    }
    // forward: y_i -= L(i,k) * y_k
    const double yk = bv[k];
    __syncthreads();
    for (int i = k + 1 + tid; i < K; i += kSolveThreads) bv[i] -= A[i * K + k] * yk;
    __syncthreads();
  }
  for (int a = tid; a < K; a += kSolveThreads) {
    const double d = A[a * K + a];
    bv[a] = fabs(d) > 1e-300 ? bv[a] / d : 0.0;
  }
  __syncthreads();
  // backward: x = L^-T y
  for (int k = K - 1; k >= 0; --k) {
    const double xk = bv[k];
    __syncthreads();
    for (int i = tid; i < k; i += kSolveThreads) bv[i] -= A[k * K + i] * xk;
    __syncthreads();
  }
  for (int a = tid; a < K; a += kSolveThreads) {
    const double x = pv[a] * bv[a];
    B.step[a] = x;
    st->step[a / 8][a % 8] = -x;  // problem.hpp:353-357
  }
  __syncthreads();
  if (tid < F * F) computePairConst(frames, st, pc, tid / F, tid % F, F, fej != 0);
}

// ---------------------------------------------------------------------------------------------------------------
// back-substitution, accept / reject, energy reduction
// ---------------------------------------------------------------------------------------------------------------
/** calculateIdepths — hessian_block_evaluation.hpp:238-263 */
__global__ void backsubKernel(const FrameDev *__restrict__ frames, const SchurBlock *__restrict__ table, const double *__restrict__ step,
                              double lambda, int F) {
  const SchurBlock be = table[blockIdx.x];
  const FrameDev &fr = frames[be.r];
  const int i = be.offset + threadIdx.x;
  if (threadIdx.x >= kSchurLandmarks || i >= fr.n) return;
  const uint8_t flg = fr.flags[i];
  if (flg & kFlagMarginalized) return;
  if (flg & kFlagIllConditioned) return;
  double d = 0;
  for (int t = 0; t < F; ++t) {
    if (t != be.r && fr.status[t] == nullptr) continue;
    const double *src = fr.ublk + (static_cast<size_t>(t) * fr.cap + i) * kUblk;
#pragma unroll
    for (int a = 0; a < kBlk; ++a) d += src[a] * step[kBlk * t + a];
  }
  const double s = (fr.b_d[i] - d) * (1.0 / (1.0 + lambda)) * fr.inv_hdd[i];
  fr.idepth_step[i] = -s;
}

/** sums the (energy, n_valid) partials of a residual-only sweep; also usable after a linearisation sweep */
__global__ void energyReduceKernel(const double *__restrict__ partials, int n_blocks, double *out) {
  __shared__ double lds[(256 / 64) * 2];
  double v[2] = {0, 0};
  for (int b = threadIdx.x; b < n_blocks; b += blockDim.x) {
    v[0] += partials[static_cast<size_t>(b) * kPartial + 44];
    v[1] += partials[static_cast<size_t>(b) * kPartial + 45];
  }
  blockSum<2, 256>(v, lds);
  if (threadIdx.x == 0) {
    out[0] = v[0];
    out[1] = v[1];
  }
}

/** acceptStep / rejectStep for landmarks and residual statuses — problem.hpp:366-402 + changeResidualStatuses :20-35.
 *  grid: one SchurBlock chunk per block.  norms[0] += sum idepth^2 (before), norms[1] += sum idepth_step^2. */
__global__ void acceptLandmarksKernel(const FrameDev *__restrict__ frames, const SchurBlock *__restrict__ table, int F, int accept,
                                      double *norms) {
  __shared__ double lds[(kSchurThreads / 64) * 2];
  const SchurBlock be = table[blockIdx.x];
  const FrameDev &fr = frames[be.r];
  const int i = be.offset + threadIdx.x;
  double v[2] = {0, 0};
  if (threadIdx.x < kSchurLandmarks && i < fr.n) {
    if (accept) {
      const double id = fr.idepth[i], st = fr.idepth_step[i];
      v[0] = id * id;
      v[1] = st * st;
      fr.idepth[i] = id + st;
    }
    fr.idepth_step[i] = 0;
    for (int t = 0; t < F; ++t) {
      if (fr.status[t] == nullptr || i >= fr.n_res[t]) continue;
      if (accept)
        fr.status[t][i] = fr.cand[t][i];
      else
        fr.cand[t][i] = fr.status[t][i];
    }
  }
  blockSum<2, kSchurThreads>(v, lds);
  if (threadIdx.x == 0 && accept) {
    atomicAdd(&norms[0], v[0]);
    atomicAdd(&norms[1], v[1]);
  }
}

__global__ void acceptFramesKernel(WindowState *st, int F, int accept, double *norms /* [2] frame part */) {
  if (threadIdx.x != 0) return;
  double state_sq = 0, step_sq = 0;
  for (int f = 0; f < F; ++f) {
    for (int a = 0; a < kBlk; ++a) {
      if (accept) {
        state_sq += st->eps[f][a] * st->eps[f][a];
        step_sq += st->step[f][a] * st->step[f][a];
        st->eps[f][a] += st->step[f][a];
      }
      st->step[f][a] = 0;
    }
    if (accept) state_sq += st->ab0[f][0] * st->ab0[f][0] + st->ab0[f][1] * st->ab0[f][1];
  }
  if (accept) {
    norms[0] = state_sq;
    norms[1] = step_sq;
  }
}

}  // namespace dsopp_hip
