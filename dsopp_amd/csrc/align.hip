// dsopp_hip_aligner_*: two-frame direct image alignment of one pyramid level on the GPU
// (hot loops E, F of SURVEY.md §3.3; replaces EigenPoseAlignment<SE3, PinholeCamera, 1, PixelMap, 1, true>,
// PROB_SRC/eigen_pose_alignment.cpp:26-329).
//
// The reference runs this serially: calculateEnergy (reproject + sample every reference point, cache the samples),
// linearize (8x8 H, b from the cached samples), 8x8 solve, calculateEnergy at the candidate, accept/reject — up to 50 times
// per level.  With n = 2 000..10 000 points (<= 0.7 MB of traffic) an iteration is pure launch latency on a GPU, so the
// whole LM iteration is ONE kernel launch here:
//   * the sweep evaluates energy AND the normal equations at the candidate state in one pass (the reference's linearize
//     reuses exactly the samples its preceding calculateEnergy cached, so evaluating both at the same state is identical);
//   * the Levenberg-Marquardt control (compare, accept/reject, lambda update, 8x8 Jacobi-preconditioned solve, next
//     candidate) runs in the prologue of the next launch, redundantly in every workgroup from the previous launch's
//     per-workgroup partial sums (deterministic), workgroup 0 publishing the new control block for the launch after;
//   * the host enqueues launches in small batches and reads the control block back once per batch.
#include <algorithm>
#include <map>
#include <memory>

#include "depth_maps.hpp"
#include "device_geom.hpp"
#include "host_linalg.hpp"
#include "pyramid.hpp"
#include "se3_math.hpp"

namespace dsopp_hip {
namespace {

constexpr int kAlignThreads = 256;
constexpr int kAlignPartial = 48;  // H (36 upper) + b (8) + energy + n_valid + pad

struct AlignFrameDev {
  const void *texels;
  int width, height;
  double fx, fy, cx, cy;
  double exposure;
  double ab0[2];
};

/** LM state of the aligner, double-buffered by launch parity */
struct AlignControl {
  double T_tr[12];       // accepted state: [R | t] rows of T_target_reference
  double ab_eps[2];
  double cand_T[12];     // candidate evaluated by the launch that reads this block
  double cand_ab[2];
  double H[64], b[8];    // system_ of the accepted state (last linearize)
  double H_used[64];     // system_ used by the most recent calculateStep (== problem.hessian())
  double step[8];
  double lambda;
  double energy;         // result.energy
  int n_valid;
  int converged;
  int active;
  int iteration;
  int have_candidate;    // 0: the evaluated state IS the accepted state (first launch)
  int linear_system_valid;
  int pad0, pad1;
};

struct AlignParams {
  double sigma_huber;
  double affine_reg[2];
  double function_tolerance, parameter_tolerance;
  double decrease_on_accept, increase_on_reject;
  int max_iterations;
  int n_points;
  int n_blocks;
};

__device__ inline void mat34Compose(const double *A, const double *B, double *C) {
  // C = A * B for rigid [R|t] 3x4
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) C[4 * i + j] = A[4 * i] * B[j] + A[4 * i + 1] * B[4 + j] + A[4 * i + 2] * B[8 + j];
    C[4 * i + 3] = A[4 * i] * B[3] + A[4 * i + 1] * B[7] + A[4 * i + 2] * B[11] + A[4 * i + 3];
  }
}

/** 1 / x for the projective divisions of the sweep: v_rcp_f64 (2^-23 relative) + two Newton steps, ~1 ulp — 5 instructions on the
 *  path to the texel addresses instead of the ~13 of an IEEE division (as the window's sweeps, pba_kernels.hpp: fastRcp) */
__device__ __forceinline__ double alignRcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float alignRcp(float x) { return 1.0f / x; }

/** 1/sqrt(x) for x > 0: hardware estimate + two Newton steps (~1 ulp); a libm sqrt + division pair costs ~180 cycles on
 *  the single thread that runs the LM control, this a dozen instructions */
__device__ inline double rsqrtNewton(double x) {
  const double h = 0.5 * x;
  double r = __builtin_amdgcn_rsq(x);  // v_rsq_f64, 2^-23 relative: no f32 round trip on the dependent chain
  r = fma(r, fma(-h * r, r, 0.5), r);
  r = fma(r, fma(-h * r, r, 0.5), r);
  return r;
}

/** 8x8 NormalLinearSystem::solve, single thread.  The reference solves the Jacobi-scaled system p H p, p = 1 / sqrt(diag + 10)
 *  (normal_linear_system.cpp:10-16,52-59) with a pivoted LDL^T; a Cholesky factorisation is invariant under symmetric diagonal
 *  scaling, so the scaling itself is not carried out (as in the window's K x K solve, pba_solve_combined.hpp) — only the zero-pivot
 *  guard refers to the scaled pivot d / (diag + 10).  The control step runs on one wave at one instruction per ~4.7 cycles: the
 *  160 instructions of the scaling were 0.35 us of every LM pass. */
template <typename GetH, typename GetB>
__device__ __forceinline__ void solve8Impl(GetH getH /* (i, j), j <= i */, GetB getB, double lambda, double *x) {
  // the system is H + lambda * diag(H) (calculateStep, eigen_pose_alignment.cpp:194-198), formed on the fly
  double A[36], y[8], linv[8], guard[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < i; ++j) A[i * (i + 1) / 2 + j] = getH(i, j);
    const double hii = getH(i, i);
    const double dg = hii + hii * lambda;
    A[i * (i + 1) / 2 + i] = dg;
    guard[i] = 1e-300 * (dg + 10.0);
    y[i] = getB(i);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double d = A[k * (k + 1) / 2 + k];
    const bool ok = d > guard[k];
    const double inv = ok ? rsqrtNewton(ok ? d : 1.0) : 0.0;
    linv[k] = inv;
    A[k * (k + 1) / 2 + k] = ok ? d * inv : 0.0;
#pragma unroll
    for (int i = k + 1; i < 8; ++i) A[i * (i + 1) / 2 + k] *= inv;
#pragma unroll
    for (int j = k + 1; j < 8; ++j)
#pragma unroll
      for (int i = j; i < 8; ++i) A[i * (i + 1) / 2 + j] -= A[i * (i + 1) / 2 + k] * A[j * (j + 1) / 2 + k];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    double s = y[i];
#pragma unroll
    for (int j = 0; j < i; ++j) s -= A[i * (i + 1) / 2 + j] * y[j];
    y[i] = s * linv[i];  // 0 for a zero pivot
  }
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    double s = y[i];
#pragma unroll
    for (int j = i + 1; j < 8; ++j) s -= A[j * (j + 1) / 2 + i] * y[j];
    y[i] = s * linv[i];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = y[i];
}

/** from the stored 8 x 8 system (row-major H, b) */
__device__ inline void solve8(const double *Hin, double lambda, const double *bin, double *x) {
  solve8Impl([&](int i, int j) { return Hin[8 * i + j]; }, [&](int i) { return bin[i]; }, lambda, x);
}

/** straight from a pass's sums (packed upper triangle 36 | b 8) plus the affine prior (eigen_pose_alignment.cpp:174-190): what the
 *  control step solves when it has just taken a new system — without the round trip of writing the expanded system to LDS and
 *  reading it back */
__device__ inline void solve8FromSums(const double *red, const double *affine_reg, double tab0, double tab1, double lambda, double *x) {
  solve8Impl(
      [&](int i, int j) {
        const double v = red[j * 8 - j * (j - 1) / 2 + (i - j)];
        return (i == j && i >= 6) ? v + affine_reg[i - 6] : v;
      },
      [&](int i) {
        const double v = red[36 + i];
        return i == 6 ? v + affine_reg[0] * tab0 : (i == 7 ? v + affine_reg[1] * tab1 : v);
      },
      lambda, x);
}

/** sample the reference intensities of the points: PatternPatch::getIntensities with PatternSize 1 (local_frame.hpp:384-388) */
template <typename S>
__global__ void sampleReferenceKernel(const Texel<S> *__restrict__ img, int W, const double *__restrict__ u, const double *__restrict__ v,
                                      double *__restrict__ intensity, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const S x = static_cast<S>(u[i]), y = static_cast<S>(v[i]);
  const int ix = static_cast<int>(x), iy = static_cast<int>(y);
  const S dx = x - static_cast<S>(ix), dy = y - static_cast<S>(iy), dxdy = dx * dy;
  const Texel<S> *p = img + static_cast<size_t>(iy) * W + ix;
  intensity[i] = static_cast<double>(dxdy * p[W + 1].I + (dy - dxdy) * p[W].I + (dx - dxdy) * p[1].I + (S(1) - dx - dy + dxdy) * p[0].I);
}

/** The LM control step of one iteration (levenberg_marquardt_algorithm.hpp:77-128 unrolled over launches): `red` holds the
 *  sums of the pass that evaluated the candidate in `c` (H upper 36 | b 8 | energy | n_valid); decides accept / reject,
 *  keeps or replaces the linear system, solves for the next step and stores the next candidate.  Single thread. */
__device__ inline void alignDecide(AlignControl &c, const double *red, const AlignFrameDev &tgt, const AlignParams &prm) {
    // evaluated state = candidate (or the initial state): energy with the affine prior (eigen_pose_alignment.cpp:101-104)
    const double tab0 = tgt.ab0[0] + c.cand_ab[0], tab1 = tgt.ab0[1] + c.cand_ab[1];
    const double e_eval = red[44] + 0.5 * (tab0 * prm.affine_reg[0] * tab0 + tab1 * prm.affine_reg[1] * tab1);
    const int n_eval = static_cast<int>(red[45] + 0.5);
    bool take_system = false;
    if (!c.have_candidate) {
      // result = problem.calculateEnergy() before the loop (levenberg_marquardt_algorithm.hpp:82)
      c.energy = e_eval;
      c.n_valid = n_eval;
      c.active = (prm.max_iterations > 0 && n_eval > 0) ? 1 : 0;
      take_system = true;
    } else {
      c.iteration += 1;
      if (n_eval == 0) {
        c.active = 0;  // rejectStep(); break;
      } else {
        if (fabs(c.energy - e_eval) / c.energy < prm.function_tolerance) c.converged = 1;
        if (e_eval < c.energy) {
          // acceptStep (eigen_pose_alignment.cpp:208-212)
          const double a0 = tgt.ab0[0] + c.ab_eps[0], a1 = tgt.ab0[1] + c.ab_eps[1];
          double step_sq = 0;
          for (int a = 0; a < 8; ++a) step_sq += c.step[a] * c.step[a];
          if (step_sq < prm.parameter_tolerance * ((a0 * a0 + a1 * a1) + prm.parameter_tolerance)) c.converged = 1;
          for (int a = 0; a < 12; ++a) c.T_tr[a] = c.cand_T[a];
          c.ab_eps[0] = c.cand_ab[0];
          c.ab_eps[1] = c.cand_ab[1];
          c.energy = e_eval;
          c.n_valid = n_eval;
          c.lambda /= prm.decrease_on_accept;
          take_system = true;
        } else {
          c.lambda *= prm.increase_on_reject;  // rejectStep: the accepted state and its system stay
        }
        if (c.converged || c.iteration >= prm.max_iterations) c.active = 0;
      }
    }
    if (take_system) {
      // system of the evaluated state: symmetric expansion of the 36 sums + affine prior block (eigen_pose_alignment.cpp:183-187)
      int e = 0;
      for (int a = 0; a < 8; ++a)
        for (int b2 = a; b2 < 8; ++b2) {
          c.H[8 * a + b2] = c.H[8 * b2 + a] = red[e];
          ++e;
        }
      for (int a = 0; a < 8; ++a) c.b[a] = red[36 + a];
      c.H[8 * 6 + 6] += prm.affine_reg[0];
      c.H[8 * 7 + 7] += prm.affine_reg[1];
      c.b[6] += prm.affine_reg[0] * tab0;
      c.b[7] += prm.affine_reg[1] * tab1;
    }
    if (c.active) {
      // calculateStep (eigen_pose_alignment.cpp:194-206): H + lambda * diag(H), leftIncrement, ab_eps -= step[6:8]
      for (int a = 0; a < 64; ++a) c.H_used[a] = c.H[a];
      double step[8];
      solve8(c.H, c.lambda, c.b, step);
#pragma unroll
      for (int a = 0; a < 8; ++a) c.step[a] = step[a];
      const Rigid E = rigidExp(step);
      double Em[12];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) Em[4 * i + j] = E.R[3 * i + j];
        Em[4 * i + 3] = E.t[i];
      }
      mat34Compose(Em, c.T_tr, c.cand_T);
      c.cand_ab[0] = c.ab_eps[0] - step[6];
      c.cand_ab[1] = c.ab_eps[1] - step[7];
      c.have_candidate = 1;
    }
}

/**
 * alignDecide executed by ONE WAVE (all 64 lanes of wave 0, uniform control flow) on register copies.  The single-lane form
 * above works in place on the LDS control block: every access is an LDS instruction whose address the compiler cannot prove
 * distinct from the previous store, so the ~300 loads and stores of a control step serialise at one LDS latency each
 * (measured: 4.7 us per step, more than the sweep, the workgroup reduction and the cross-workgroup exchange together).
 * Here everything is loaded first (one LDS latency), the 8 x 8 system is expanded / copied one entry per lane, the decision
 * and the solve run in registers (redundantly in every lane: no cross-lane traffic), and the results leave in one batch of
 * stores.  Same arithmetic, operation for operation, as alignDecide.
 */
__device__ inline void alignDecideWave(AlignControl &c, const double *red, const AlignFrameDev &tgt, const AlignParams &prm) {
  const int lane = threadIdx.x & 63;
  const int hi = lane >> 3, hj = lane & 7;
  // ---- loads
  const double cand_ab0 = c.cand_ab[0], cand_ab1 = c.cand_ab[1];
  double ab_eps0 = c.ab_eps[0], ab_eps1 = c.ab_eps[1];
  double energy = c.energy, lambda = c.lambda;
  int n_valid = c.n_valid, converged = c.converged, active = c.active, iteration = c.iteration;
  const int have_candidate = c.have_candidate;
  double stepv[8], Ttr[12], candT[12];
#pragma unroll
  for (int a = 0; a < 8; ++a) stepv[a] = c.step[a];
#pragma unroll
  for (int a = 0; a < 12; ++a) {
    Ttr[a] = c.T_tr[a];
    candT[a] = c.cand_T[a];
  }
  const double red_energy = red[44], red_n = red[45];
  const int lo8 = hi < hj ? hi : hj, hi8 = hi < hj ? hj : hi;
  const double h_new = red[lo8 * 8 - lo8 * (lo8 - 1) / 2 + (hi8 - lo8)], h_old = c.H[lane];  // packed upper triangle, as alignDecide expands it
  const double b_new = red[36 + hj], b_old = c.b[hj];
  // ---- decision (uniform)
  const double tab0 = tgt.ab0[0] + cand_ab0, tab1 = tgt.ab0[1] + cand_ab1;
  const double e_eval = red_energy + 0.5 * (tab0 * prm.affine_reg[0] * tab0 + tab1 * prm.affine_reg[1] * tab1);
  const int n_eval = static_cast<int>(red_n + 0.5);
  bool take_system = false;
  if (!have_candidate) {
    energy = e_eval;
    n_valid = n_eval;
    active = (prm.max_iterations > 0 && n_eval > 0) ? 1 : 0;
    take_system = true;
  } else {
    iteration += 1;
    if (n_eval == 0) {
      active = 0;
    } else {
      if (fabs(energy - e_eval) / energy < prm.function_tolerance) converged = 1;
      if (e_eval < energy) {
        const double a0 = tgt.ab0[0] + ab_eps0, a1 = tgt.ab0[1] + ab_eps1;
        double step_sq = 0;
#pragma unroll
        for (int a = 0; a < 8; ++a) step_sq += stepv[a] * stepv[a];
        if (step_sq < prm.parameter_tolerance * ((a0 * a0 + a1 * a1) + prm.parameter_tolerance)) converged = 1;
#pragma unroll
        for (int a = 0; a < 12; ++a) Ttr[a] = candT[a];
        ab_eps0 = cand_ab0;
        ab_eps1 = cand_ab1;
        energy = e_eval;
        n_valid = n_eval;
        lambda /= prm.decrease_on_accept;
        take_system = true;
      } else {
        lambda *= prm.increase_on_reject;
      }
      if (converged || iteration >= prm.max_iterations) active = 0;
    }
  }
  // ---- system of the accepted state: one entry per lane
  double h = h_old, b = b_old;
  if (take_system) {
    h = h_new;
    b = b_new;
    if (lane == 8 * 6 + 6) h += prm.affine_reg[0];
    if (lane == 8 * 7 + 7) h += prm.affine_reg[1];
    if (hj == 6) b += prm.affine_reg[0] * tab0;
    if (hj == 7) b += prm.affine_reg[1] * tab1;
    c.H[lane] = h;
    if (lane < 8) c.b[lane] = b;
  }
  double stepn[8], candTn[12], cand_abn0 = cand_ab0, cand_abn1 = cand_ab1;
  int have_candidate_n = have_candidate;
  if (active) {
    c.H_used[lane] = h;
    if (take_system) {
      // the new system is solved from the sums it came from (same values, entry for entry, as the expansion stored above)
      solve8FromSums(red, prm.affine_reg, tab0, tab1, lambda, stepn);
    } else {
      // (kept system: nothing was stored to c.H / c.b in this step)
      solve8(c.H, lambda, c.b, stepn);
    }
    const Rigid E = rigidExp(stepn);
    double Em[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) Em[4 * i + j] = E.R[3 * i + j];
      Em[4 * i + 3] = E.t[i];
    }
    mat34Compose(Em, Ttr, candTn);
    cand_abn0 = ab_eps0 - stepn[6];
    cand_abn1 = ab_eps1 - stepn[7];
    have_candidate_n = 1;
  } else {
#pragma unroll
    for (int a = 0; a < 8; ++a) stepn[a] = stepv[a];
#pragma unroll
    for (int a = 0; a < 12; ++a) candTn[a] = candT[a];
  }
  // ---- one batch of stores (lane 0)
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 12; ++a) {
      c.T_tr[a] = Ttr[a];
      c.cand_T[a] = candTn[a];
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) c.step[a] = stepn[a];
    c.ab_eps[0] = ab_eps0;
    c.ab_eps[1] = ab_eps1;
    c.cand_ab[0] = cand_abn0;
    c.cand_ab[1] = cand_abn1;
    c.lambda = lambda;
    c.energy = energy;
    c.n_valid = n_valid;
    c.converged = converged;
    c.active = active;
    c.iteration = iteration;
    c.have_candidate = have_candidate_n;
  }
}

/** per-pass constants of the sweep: reprojection matrices of the candidate pose, photometric parameters */
template <typename S>
struct AlignSweepCtx {
  S U[12], M[12];
  S b_t, b_r, Wr, Hr, Wt, Ht, fxt, fyt;
  double s_ratio, s_arg;  // brightness scale = s_ratio * exp(s_arg): evaluated by the point code behind its texel loads
  const Texel<S> *img;
  int W;
  double sigma;
};

/** per-pass constants from the candidate pose and the reference camera's reciprocal intrinsics (per-level constants of the persistent
 *  kernel: four f64 divisions less per pass) */
template <typename S>
__device__ __forceinline__ void alignSweepSetup(AlignSweepCtx<S> &x, const AlignFrameDev &ref, const AlignFrameDev &tgt, const double *T, double cand_ab0,
                                                double cand_ab1, double sigma, double ifx, double ify, double k02, double k12, double s_ratio) {
  double Ud[12];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    Ud[4 * i] = T[4 * i] * ifx;
    Ud[4 * i + 1] = T[4 * i + 1] * ify;
    Ud[4 * i + 2] = T[4 * i] * k02 + T[4 * i + 1] * k12 + T[4 * i + 2];
    Ud[4 * i + 3] = T[4 * i + 3];
  }
#pragma unroll
  for (int i = 0; i < 12; ++i) x.U[i] = S(Ud[i]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    x.M[j] = S(tgt.fx * Ud[j] + tgt.cx * Ud[8 + j]);
    x.M[4 + j] = S(tgt.fy * Ud[4 + j] + tgt.cy * Ud[8 + j]);
    x.M[8 + j] = S(Ud[8 + j]);
  }
  const double tab0 = tgt.ab0[0] + cand_ab0, tab1 = tgt.ab0[1] + cand_ab1;
  x.s_ratio = s_ratio;
  x.s_arg = tab0 - ref.ab0[0];
  x.b_t = S(tab1);
  x.b_r = S(ref.ab0[1]);
  x.Wr = S(ref.width);
  x.Hr = S(ref.height);
  x.Wt = S(tgt.width);
  x.Ht = S(tgt.height);
  x.fxt = S(tgt.fx);
  x.fyt = S(tgt.fy);
  x.img = static_cast<const Texel<S> *>(tgt.texels);
  x.W = tgt.width;
  x.sigma = sigma;
}

template <typename S>
__device__ __forceinline__ void alignSweepSetup(AlignSweepCtx<S> &x, const AlignFrameDev &ref, const AlignFrameDev &tgt, const AlignControl &sc,
                                                const AlignParams &prm) {
  const double *T = sc.cand_T;
  // ArrayReprojector ctor — camera_reproject.hpp:235-260
  const double ifx = 1.0 / ref.fx, ify = 1.0 / ref.fy, k02 = -ref.cx / ref.fx, k12 = -ref.cy / ref.fy;
  alignSweepSetup<S>(x, ref, tgt, T, sc.cand_ab[0], sc.cand_ab[1], prm.sigma_huber, ifx, ify, k02, k12, tgt.exposure / ref.exposure);
}

/** one reference point: residual r, Huber weight, Jacobian row d[8], energy term and validity (1 / 0).
 *  Everything is predicated instead of branching: an invalid point reads a safe texel and comes back with weight zero. */
template <typename S>
__device__ __forceinline__ void alignPointEval(const AlignSweepCtx<S> &x, S u, S v, S idepth, S iref, bool present, double (&d)[8], double &r_out,
                                               double &wgt_out, double &energy_out, double &valid_out) {
  const S *M = x.M, *U = x.U;
  // reproject (checked) — camera_reproject.hpp:270-293
  bool good = present && validIdepth(idepth) && insideROI(u, v, x.Wr, x.Hr);
  const S px = M[0] * u + M[1] * v + (M[2] + M[3] * idepth);
  const S py = M[4] * u + M[5] * v + (M[6] + M[7] * idepth);
  const S pz = M[8] * u + M[9] * v + (M[10] + M[11] * idepth);
  const S ipz = alignRcp(pz);
  const S ta = px * ipz, tb = py * ipz;
  good = good && (pz > S(0)) && insideROI(ta, tb, x.Wt, x.Ht);
  const S tu = good ? ta : S(4), tv = good ? tb : S(4);
  const int ix = static_cast<int>(tu), iy = static_cast<int>(tv);
  const Texel<S> *p = x.img + static_cast<size_t>(iy) * x.W + ix;
  const Texel<S> t00 = loadTexel(p), t10 = loadTexel(p + 1), t01 = loadTexel(p + x.W), t11 = loadTexel(p + x.W + 1);
  const S s_scale = S(x.s_ratio * exp(x.s_arg));  // (behind the loads: ~50 instructions that do not depend on them)
  const S dx = tu - static_cast<S>(ix), dy = tv - static_cast<S>(iy), dxdy = dx * dy;
  const int rx = static_cast<int>(floor(tu + S(0.5))) - ix, ry = static_cast<int>(floor(tv + S(0.5))) - iy;
  const S m = ry ? (rx ? t11.mask : t01.mask) : (rx ? t10.mask : t00.mask);
  const bool valid = good && (m != S(0));  // mask_.valid(target_pattern) — eigen_pose_alignment.cpp:78
  const S w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = S(1) - dx - dy + dxdy;
  const S sI = w11 * t11.I + w01 * t01.I + w10 * t10.I + w00 * t00.I;
  const S sIx = w11 * t11.Ix + w01 * t01.Ix + w10 * t10.Ix + w00 * t00.Ix;
  const S sIy = w11 * t11.Iy + w01 * t01.Iy + w10 * t10.Iy + w00 * t00.Iy;
  const S right = s_scale * (iref - x.b_r);
  const double r = static_cast<double>((sI - x.b_t) - right);
  const double r2 = r * r, sig = x.sigma;
  const bool lin = r2 > sig * sig;
  const double nrm = fabs(r);
  r_out = r;
  wgt_out = valid ? (lin ? sig / nrm : 1.0) : 0.0;
  energy_out = valid ? (lin ? sig * nrm - 0.5 * sig * sig : 0.5 * r2) : 0.0;
  valid_out = valid ? 1.0 : 0.0;
  // Jacobian row at the same state (non-checking reprojector, camera_reproject.hpp:339-365; eigen_pose_alignment.cpp:158-172)
  const S X = U[0] * u + U[1] * v + (U[2] + U[3] * idepth);
  const S Y = U[4] * u + U[5] * v + (U[6] + U[7] * idepth);
  const S Z = U[8] * u + U[9] * v + (U[10] + U[11] * idepth);
  const S rho = valid ? alignRcp(Z) : S(0), b0 = X * rho, b1 = Y * rho, nid = idepth * rho;
  const S fxt = x.fxt, fyt = x.fyt, b0b1 = b0 * b1;
  d[0] = -static_cast<double>(sIx * (fxt * nid));
  d[1] = -static_cast<double>(sIy * (fyt * nid));
  d[2] = -static_cast<double>(sIx * (fxt * (-nid * b0)) + sIy * (fyt * (-nid * b1)));
  d[3] = -static_cast<double>(sIx * (fxt * (-b0b1)) + sIy * (fyt * (-(b1 * b1 + S(1)))));
  d[4] = -static_cast<double>(sIx * (fxt * (b0 * b0 + S(1))) + sIy * (fyt * b0b1));
  d[5] = -static_cast<double>(sIx * (fxt * (-b1)) + sIy * (fyt * b0));
  d[6] = -static_cast<double>(right);
  d[7] = -1.0;
}

/** ... and its contribution to this thread's H (36 upper) | b (8) | energy | n_valid (the launch-per-iteration and single-workgroup
 *  kernels: per-thread accumulators, workgroup reduction through an LDS transpose) */
template <typename S>
__device__ __forceinline__ void alignPoint(const AlignSweepCtx<S> &x, S u, S v, S idepth, S iref, bool present, double (&acc)[kAlignPartial]) {
  double d[8], r, wgt, en, vl;
  alignPointEval<S>(x, u, v, idepth, iref, present, d, r, wgt, en, vl);
  acc[44] += en;
  acc[45] += vl;
  int e = 0;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const double wa = wgt * d[a];
#pragma unroll
    for (int b2 = a; b2 < 8; ++b2) acc[e++] += wa * d[b2];
    acc[36 + a] += wa * r;
  }
}

/** PoseAlignerProblem::calculateEnergy + linearize (eigen_pose_alignment.cpp:55-192) at the candidate state of `sc` for the
 *  points first, first + stride, ...: fills acc with this thread's share of H (36 upper) | b (8) | energy | n_valid */
template <typename S>
__device__ __forceinline__ void alignSweep(const AlignFrameDev &ref, const AlignFrameDev &tgt, const double *__restrict__ pu, const double *__restrict__ pv,
                                           const double *__restrict__ pid, const double *__restrict__ pint, const AlignControl &sc,
                                           const AlignParams &prm, int first, int stride, double (&acc)[kAlignPartial]) {
#pragma unroll
  for (int e = 0; e < kAlignPartial; ++e) acc[e] = 0;
  AlignSweepCtx<S> x;
  alignSweepSetup<S>(x, ref, tgt, sc, prm);
  // (4 points in flight per thread was tried: the 48 f64 accumulators + 16 texels per lane spill)
  for (int i = first; i < prm.n_points; i += stride)
    alignPoint<S>(x, static_cast<S>(pu[i]), static_cast<S>(pv[i]), static_cast<S>(pid[i]), static_cast<S>(pint[i]), true, acc);
}

/**
 * One LM iteration = one launch: LM control from the previous launch's partial sums (prologue, every workgroup),
 * then PoseAlignerProblem::calculateEnergy + linearize (eigen_pose_alignment.cpp:55-192) at the new candidate.
 */
template <typename S>
__global__ void __launch_bounds__(kAlignThreads) alignIterationKernel(AlignFrameDev ref, AlignFrameDev tgt, const double *__restrict__ pu,
                                                                       const double *__restrict__ pv, const double *__restrict__ pid,
                                                                       const double *__restrict__ pint, const AlignControl *__restrict__ cin,
                                                                       AlignControl *__restrict__ cout, const double *__restrict__ prev_partials,
                                                                       double *__restrict__ partials, AlignParams prm, int launch_index) {
  __shared__ __attribute__((aligned(16))) double red[kAlignPartial * (kAlignThreads + 2)];
  __shared__ AlignControl sc;
  const int tid = threadIdx.x;
  // ---------------- prologue: LM control ----------------
  if (launch_index == 0) {
    if (tid == 0) sc = *cin;  // initial block prepared by the host: candidate == accepted state
    __syncthreads();
  } else {
    // the incoming control block is requested first so that its round trip overlaps the partial sums below: all threads
    // fetch one 8-byte word each into LDS
    constexpr int kCtrlWords = static_cast<int>(sizeof(AlignControl) / 8);
    static_assert(sizeof(AlignControl) % 8 == 0 && kCtrlWords <= kAlignThreads, "control block copy assumes one word per thread");
    double ctrl_word = 0;
    if (tid < kCtrlWords) ctrl_word = reinterpret_cast<const double *>(cin)[tid];
    // deterministic sum of the previous launch's partials: thread e < 48 adds column e over all workgroups
    // (5 thread groups of 48 take the workgroups g, g + 5, ... with four independent loads in flight each — a single
    // dependent load-add chain over all workgroups costs one memory round trip per workgroup — then a fixed-order combine)
    {
      constexpr int kGroups = kAlignThreads / kAlignPartial;  // 5
      const int e = tid % kAlignPartial, grp = tid / kAlignPartial;
      double p0 = 0, p1 = 0, p2 = 0, p3 = 0;
      if (grp < kGroups) {
        const double *src = prev_partials + e;
        // the first 8 workgroups of this group in ONE round trip: clamped indices + a 0 / 1 factor instead of predicated loads
        // (a level of the tracker has <= 40 workgroups: this is the whole sum)
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int b = grp + j * kGroups;
          const int bc = b < prm.n_blocks ? b : prm.n_blocks - 1;
          v[j] = src[static_cast<size_t>(bc) * kAlignPartial];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (grp + j * kGroups < prm.n_blocks) ? v[j] : 0.0;
        p0 = v[0] + v[4];
        p1 = v[1] + v[5];
        p2 = v[2] + v[6];
        p3 = v[3] + v[7];
        for (int b = grp + 8 * kGroups; b < prm.n_blocks; b += kGroups) p0 += src[static_cast<size_t>(b) * kAlignPartial];
        red[kAlignPartial + grp * kAlignPartial + e] = (p0 + p1) + (p2 + p3);
      }
      __syncthreads();
      if (tid < kAlignPartial) {
        double tot = 0;
#pragma unroll
        for (int g2 = 0; g2 < kGroups; ++g2) tot += red[kAlignPartial + g2 * kAlignPartial + tid];
        red[tid] = tot;
      }
    }
    __syncthreads();
    if (tid < kCtrlWords) reinterpret_cast<double *>(&sc)[tid] = ctrl_word;
    __syncthreads();
    if (!sc.active) {
      if (blockIdx.x == 0 && tid < kCtrlWords) reinterpret_cast<double *>(cout)[tid] = ctrl_word;
      return;
    }
    if (tid < 64) alignDecideWave(sc, red, tgt, prm);  // wave 0, register copies (see alignDecideWave)
    __syncthreads();
    if (blockIdx.x == 0 && tid < kCtrlWords) reinterpret_cast<double *>(cout)[tid] = reinterpret_cast<const double *>(&sc)[tid];
    if (!sc.active) return;
    __syncthreads();
  }
  if (launch_index == 0 && blockIdx.x == 0 && tid == 0) *cout = sc;

  // ---------------- sweep at the candidate state ----------------
  double acc[kAlignPartial];
  alignSweep<S>(ref, tgt, pu, pv, pid, pint, sc, prm, blockIdx.x * kAlignThreads + tid, gridDim.x * kAlignThreads, acc);
  // workgroup reduction through an LDS transpose (see blockReduceStore in pba_kernels.hpp)
  {
    constexpr int RS = kAlignThreads + 2;
#pragma unroll
    for (int e = 0; e < kAlignPartial; ++e) red[e * RS + tid] = acc[e];
    __syncthreads();
    // four adjacent lanes per row (each sums a quarter, fixed order), combined with two DPP swaps: 192 lanes read 32 entries
    // each instead of 48 lanes walking 128
    static_assert(4 * kAlignPartial <= kAlignThreads, "four lanes per row");
    const int row_idx = tid >> 2, quarter = tid & 3;
    double s = 0;
    if (row_idx < kAlignPartial) {
      const double2 *row = reinterpret_cast<const double2 *>(red + row_idx * RS) + quarter * (kAlignThreads / 8);
      double s0 = 0, s1 = 0;
#pragma unroll 8
      for (int j = 0; j < kAlignThreads / 8; ++j) {
        const double2 q = row[j];
        s0 += q.x;
        s1 += q.y;
      }
      s = s0 + s1;
    }
    auto dpp = [](double v, auto ctrl) {
      int lo = __double2loint(v), hi = __double2hiint(v);
      lo = __builtin_amdgcn_mov_dpp(lo, decltype(ctrl)::value, 0xF, 0xF, true);
      hi = __builtin_amdgcn_mov_dpp(hi, decltype(ctrl)::value, 0xF, 0xF, true);
      return __hiloint2double(hi, lo);
    };
    s += dpp(s, std::integral_constant<int, 0xB1>{});  // quad_perm [1,0,3,2]
    s += dpp(s, std::integral_constant<int, 0x4E>{});  // quad_perm [2,3,0,1]
    if (row_idx < kAlignPartial && quarter == 0) partials[static_cast<size_t>(blockIdx.x) * kAlignPartial + row_idx] = s;
  }
}

/**
 * The whole LM loop of one alignment in ONE launch of ONE workgroup (1024 threads): up to a few thousand reference points
 * are a latency problem, not a throughput one — a launch per iteration costs more than the iteration.  Per iteration:
 * control step (thread 0, alignDecide) -> sweep (<= a dozen points per thread) -> deterministic reduction: two DPP steps
 * fold 4 lanes, an LDS transpose of the 256 remaining columns, 16 lanes per row.  Same state machine and arithmetic as
 * alignIterationKernel (the multi-workgroup path stays for large point sets).
 */
constexpr int kLoopThreads = 512;   // 2 waves per SIMD: the sweep's 48 accumulators + temporaries fit in registers
constexpr int kAlignLoopMaxPoints = 1024;  // <= 2 points per lane: beyond that one launch per iteration on many workgroups is faster (measured)
constexpr int kLoopCols = kLoopThreads / 2;  // columns left after the DPP fold of lane pairs

template <int CTRL>
__device__ __forceinline__ double alignDpp(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

template <typename S>
__global__ void __launch_bounds__(kLoopThreads) alignLoopKernel(AlignFrameDev ref, AlignFrameDev tgt, const double *__restrict__ pu,
                                                                const double *__restrict__ pv, const double *__restrict__ pid,
                                                                const double *__restrict__ pint, AlignControl *ctrl_io, AlignParams prm) {
  extern __shared__ __attribute__((aligned(16))) char loop_smem[];
  double *red = reinterpret_cast<double *>(loop_smem);  // [kAlignPartial][kLoopCols + 2]
  double *tot = red + kAlignPartial * (kLoopCols + 2);  // [kAlignPartial]
  __shared__ AlignControl sc;
  const int tid = threadIdx.x;
  constexpr int kCtrlWords = static_cast<int>(sizeof(AlignControl) / 8);
  if (tid < kCtrlWords) reinterpret_cast<double *>(&sc)[tid] = reinterpret_cast<const double *>(ctrl_io)[tid];
  __syncthreads();
  const int total_passes = prm.max_iterations + 2;  // initial evaluation + one per iteration + the closing control step
  for (int pass = 0; pass < total_passes; ++pass) {
    if (pass > 0) {
      if (tid < 64) alignDecideWave(sc, tot, tgt, prm);
      __syncthreads();
      if (!sc.active) break;
    }
    double acc[kAlignPartial];
    alignSweep<S>(ref, tgt, pu, pv, pid, pint, sc, prm, tid, kLoopThreads, acc);
    // fold lane pairs (quad_perm swap), the sum stays in the even lane: fixed order => deterministic
#pragma unroll
    for (int e = 0; e < kAlignPartial; ++e) acc[e] += alignDpp<0xB1>(acc[e]);
    constexpr int RS = kLoopCols + 2;
    if ((tid & 1) == 0) {
#pragma unroll
      for (int e = 0; e < kAlignPartial; ++e) red[e * RS + (tid >> 1)] = acc[e];
    }
    __syncthreads();
    {
      // row e: 8 lanes, each sums 32 columns (16 x double2), then the 8-lane DPP tree
      const int row = tid >> 3, part = tid & 7;
      double sacc = 0;
      if (row < kAlignPartial) {
        const double2 *src = reinterpret_cast<const double2 *>(red + row * RS) + part * (kLoopCols / 16);
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int j = 0; j < kLoopCols / 16; ++j) {
          const double2 q = src[j];
          s0 += q.x;
          s1 += q.y;
        }
        sacc = s0 + s1;
      }
      sacc += alignDpp<0xB1>(sacc);
      sacc += alignDpp<0x4E>(sacc);
      sacc += alignDpp<0x141>(sacc);
      if (row < kAlignPartial && part == 0) tot[row] = sacc;
    }
    __syncthreads();
  }
  __syncthreads();
  if (tid < kCtrlWords) reinterpret_cast<double *>(ctrl_io + 1)[tid] = reinterpret_cast<const double *>(&sc)[tid];
}


// ---------------------------------------------------------------------------------------------------------------
// estimatePose's coarse-to-fine loop of ONE initialisation as ONE launch (monocular_tracker.cpp:202-226): all pyramid levels,
// every Levenberg-Marquardt iteration of each.  The launch-per-iteration path spends 8.5 us + a ~2.5 us kernel boundary per
// iteration, of which the sweep itself is a fraction: the rest is the start-up of a kernel (control block and point words from
// memory, again every iteration).  Here a few dozen workgroups stay resident, keep the LM state in LDS and meet once per
// iteration at a device-scope arrival counter:
//     sweep own points at the candidate -> workgroup sums (LDS) -> 48 partial sums written write-through (sc1) -> arrive ->
//     every workgroup reads all partial sums back (sc1 loads, fixed order: deterministic and identical in every workgroup) ->
//     every workgroup takes the same LM decision (alignDecide) -> next sweep.
// Hand-off per cdna_hip_programming.md, guideline 16 (R1: sc1 payload stores, every storing wave drains vmcnt, ONE relaxed
// agent-scope arrival, relaxed polls with s_sleep, sc1 loads on the consumer side — no fences).  The partial sums are double
// buffered by pass parity: a workgroup can only be one barrier ahead of the slowest one.  Every spin is bounded; a time-out
// sets `failed` and the host repeats the frame on the launch-per-iteration path.
// Level transitions (reset; push the keyframe's points of the next finer level; push the target with the current estimate;
// accept the level when rmse < 2.5 * rmse_last[level]) run on the device as well: identical in every workgroup.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kPyramidMaxWorkgroups = 32;  // participants: one per CU (100 KB of LDS each); an XCD has 32 CUs
// Cross-workgroup exchange of the persistent kernel: every partial sum is its own ready flag.  The host fills the three rotating
// partial buffers with a NaN bit pattern no sum can take (both 32-bit halves equal, so one 32-bit fill does it); a workgroup
// re-arms its slots of the buffer after next before it publishes into the current one; consumers poll the values themselves.
// One memory round trip per pass (store -> load) instead of three (arrive atomic, counter poll, partial loads).
constexpr unsigned kPyramidSentinelWord = 0x7FF85A5Au;
constexpr unsigned long long kPyramidSentinel = (static_cast<unsigned long long>(kPyramidSentinelWord) << 32) | kPyramidSentinelWord;
constexpr int kPyramidBuffers = 3;
constexpr unsigned kPyramidFailed = 1u;
constexpr int kPyramidHypotheses = 8;  // initialisations per launch: one per XCD
constexpr size_t kPyramidSetDoubles = static_cast<size_t>(kPyramidBuffers) * kPyramidMaxWorkgroups * 48 + 1;  // one hypothesis: exchange buffers + failed flag

struct AlignLevelDev {
  AlignFrameDev ref, tgt;
  const double *pu, *pv, *pid, *pint;
  int n_points;
  int pad;
  double rmse_limit;  // kEnergyRatioThreshold * rmse_last_pose_estimation[level]
};

struct AlignPyramidResult {
  int levels_done;   // levels whose solve ran (counted from the coarsest)
  int success;       // every level passed its energy test
  int failed;        // a bounded spin timed out: nothing in here is valid
  int lm_iterations; // sum over the levels
  int same_xcd;      // all participants ran on one XCD: passes after the first exchanged through that XCD's L2
  int end_phase;     // buffer phase (pass counter mod 3) the NEXT launch has to start with: the slots stay armed across launches
  double rmse[DSOPP_HIP_MAX_LEVELS];
  int iterations[DSOPP_HIP_MAX_LEVELS];
  int n_valid[DSOPP_HIP_MAX_LEVELS];
  double T_tr[12];   // T_target_reference after the last accepted level ([R | t] rows)
  double ab[2];      // target affine brightness after the last accepted level
  long long stamps[16];  // -DDSOPP_HIP_STAMPS: wall_clock64 at the phase boundaries of one pass (level 0, third pass) of workgroup 0
};
#ifdef DSOPP_HIP_STAMPS
#define AP_STAMP(i) do { if (blk == 0 && tid == 0 && lvl == 0 && pass == 2) h_out->stamps[i] = wall_clock64(); } while (0)
#elif defined(DSOPP_HIP_MARKS)
// ISA reading aid (scripts/isa_phase_count.py): a comment line in the assembly at every phase boundary
#define AP_STAMP(i) asm volatile("; ##AP_MARK " #i)
#else
#define AP_STAMP(i) do { } while (0)
#endif

struct AlignPyramidArgs {
  AlignLevelDev level[DSOPP_HIP_MAX_LEVELS];
  int n_levels;
  int max_iterations;
  double sigma_huber, affine_reg[2], function_tolerance, parameter_tolerance, decrease_on_accept, increase_on_reject, lambda0;
  double inv_decrease;  // 1 / decrease_on_accept, exact (a power of two): the control step multiplies
  // Up to kPyramidHypotheses initialisations of estimatePose run in ONE launch, one per XCD: the workgroups with blockIdx % spread == h
  // are the participants of hypothesis h (the dispatcher deals workgroups round-robin over the 8 XCDs), every hypothesis with its own
  // exchange buffers, failed flag, buffer phase and result slot.  n_hyp == 1 is the single-initialisation launch (only every spread-th
  // workgroup takes part).  The reference tries its initialisations one after the other (monocular_tracker.cpp:193-243); they do not
  // depend on each other, so the host takes the lowest-index success of a batch and gets the sequential loop's result.
  int n_hyp;
  double T_tr0[kPyramidHypotheses][12];
  double ab0[2];
  double *partials;        // [n_hyp][kPyramidSetDoubles]: per hypothesis [kPyramidBuffers][kPyramidMaxWorkgroups][kAlignPartial] + the failed flag
  AlignPyramidResult *out; // [n_hyp]
  int start_phase[kPyramidHypotheses];  // pass counter (mod 3) the hypothesis' buffers continue from
  int debug_poison_lds;    // -DDSOPP_HIP_STAMPS only: fill the LDS block with NaN at kernel start (a read of uninitialised LDS then shows in every call)
  double *debug_sums;      // -DDSOPP_HIP_STAMPS only: [pass][workgroup][2] totals (energy, H00) every workgroup derived, hypothesis 0 (exchange check)
  int spread;              // launch = spread x participants per hypothesis (8: one XCD each, 1: no placement attempt — single hypothesis only)
};

using gu32 = __attribute__((address_space(1))) unsigned;
using gu64 = __attribute__((address_space(1))) unsigned long long;

// Workgroup sums of the persistent kernel: the f64 matrix cores add up the points of a wave.  Every lane hands the row of its point to
// the wave through LDS — A row [w d0 .. w d7 | energy | valid], B row [d0 .. d7 | r | 1] — and sixteen v_mfma_f64_16x16x4_f64
// (four points each) leave  D[i][j] = sum w d_i d_j (i, j < 8),  D[i][8] = sum w d_i r,  D[8][9] = sum energy,  D[9][9] = n_valid  in
// 4 registers per lane: no 48 f64 accumulators per lane (96 VGPRs: the kernel spilled and the workgroup reduction moved 2 x 98 KB
// through LDS per pass, 1.2 us), and the wave's sum needs no shuffle tree.
constexpr int kRowStride = 18;                       // doubles per published row (10 used; 144 B: b128 stores of 64 lanes spread over all banks)
constexpr int kWaveRows = 2 * 64 * kRowStride;       // A rows, then B rows of one wave
constexpr int kAlignWaves = kAlignThreads / 64;
using af64x4 = __attribute__((ext_vector_type(4))) double;

__device__ __forceinline__ void alignPublishRow(double *wave_rows, const double (&d)[8], double r, double wgt, double en, double vl) {
  const int lane = threadIdx.x & 63;
  double2 *ra = reinterpret_cast<double2 *>(wave_rows + lane * kRowStride);
  double2 *rb = reinterpret_cast<double2 *>(wave_rows + 64 * kRowStride + lane * kRowStride);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    ra[c] = double2{wgt * d[2 * c], wgt * d[2 * c + 1]};
    rb[c] = double2{d[2 * c], d[2 * c + 1]};
  }
  ra[4] = double2{en, vl};
  rb[4] = double2{r, 1.0};
}

/** acc0 / acc1 += the products of the 64 published points (even / odd steps: two independent accumulation chains) */
__device__ __forceinline__ void alignContract(const double *wave_rows, af64x4 &acc0, af64x4 &acc1) {
  const int lane = threadIdx.x & 63;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // operand lane l = (m = l & 15, k = l >> 4) of step t reads entry m of point 4 t + k (entries 10 .. 15 feed result entries nobody reads)
  const double *src = wave_rows + (lane >> 4) * kRowStride + (lane & 15);
  double av[16], bv[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    av[t] = src[4 * t * kRowStride];
    bv[t] = src[64 * kRowStride + 4 * t * kRowStride];
  }
  asm volatile("" ::: "memory");
  // (pinning all 32 operands in registers before the first matrix instruction was measured and dropped: the phase is bound by the sixteen
  // 64-cycle matrix instructions themselves, 0.68 against 0.64 us — profiles/r06/tracker_stamps.txt)
#pragma unroll
  for (int t = 0; t < 16; t += 2) {
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[t], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t + 1], bv[t + 1], acc1, 0, 0, 0);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

/** where result register v of this lane goes in the packed sums (H upper 36 | b 8 | energy | n_valid), -1: nowhere.
 *  Result layout of the instruction: register v of lane l = entry (row (l >> 4) + 4 v, column l & 15). */
__device__ __forceinline__ int alignPackedIndex(int lane, int v) {
  const int i = (lane >> 4) + 4 * v, j = lane & 15;
  if (i < 8) {
    if (j < 8) return j >= i ? i * 8 - i * (i - 1) / 2 + (j - i) : -1;
    return j == 8 ? 36 + i : -1;
  }
  if (j == 9) return i == 8 ? 44 : (i == 9 ? 45 : -1);
  return -1;
}

#ifdef DSOPP_HIP_MARKS
#define PD_MARK(i) asm volatile("; ##PD_MARK " #i)
#else
#define PD_MARK(i) do { } while (0)
#endif

/**
 * 8 x 8 NormalLinearSystem::solve for the persistent kernel's control step: (H + lambda diag(H)) x = b by an LDL^T factorisation, written for
 * LATENCY.  One wave runs this alone, in order, at ~8 cycles per dependent f64 operation: what it costs is the length of its dependency
 * chain, not its instruction count.  Hence
 *   * right-looking: as soon as a pivot's reciprocal is known the whole trailing matrix is updated by independent FMAs, which the
 *     scheduler interleaves with the NEXT pivot's reciprocal chain (the left-looking form the compiler made of the Cholesky above put a
 *     dot-product chain in front of every pivot);
 *   * the right-hand side is carried as a ninth column, so the forward substitution costs no extra chain;
 *   * LDL^T: a reciprocal (v_rcp_f64 + 2 Newton steps, 5 dependent operations) per pivot instead of a reciprocal square root (8), unit
 *     triangles in the substitutions;
 *   * no branches: a basic-block boundary is a wall for the scheduler.  A pivot that fails the guard (the reference's zero-pivot test on
 *     the Jacobi-scaled pivot d / (diag + 10), normal_linear_system.cpp:10-16,52-59) gets reciprocal 0: its column, its y and its x vanish.
 * Chain: 8 x (reciprocal 5 + scale 1 + update 1) + back substitution 8 = ~64 dependent operations (the Cholesky form: ~170).
 */
template <typename GetH, typename GetB>
__device__ __forceinline__ void solve8Ldl(GetH getH /* (i, j), j <= i */, GetB getB, double lambda, double *x) {
  double A[36], y[8], dinv[8], guard[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
#pragma unroll
    for (int j = 0; j < i; ++j) A[i * (i + 1) / 2 + j] = getH(i, j);
    const double hii = getH(i, i);
    const double dg = hii + hii * lambda;  // H + lambda * diag(H) (calculateStep, eigen_pose_alignment.cpp:194-198)
    A[i * (i + 1) / 2 + i] = dg;
    guard[i] = 1e-300 * (dg + 10.0);
    y[i] = getB(i);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const double d = A[k * (k + 1) / 2 + k];
    const bool ok = d > guard[k];
    double r = __builtin_amdgcn_rcp(d);  // (a pivot that fails the guard: whatever this becomes is dropped by the select below)
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double inv = ok ? r : 0.0;
    dinv[k] = inv;
    // column k of L (unit diagonal): l_ik = a_ik / d_k, kept beside the unscaled a_ik the update needs
    double l[8];
#pragma unroll
    for (int i = k + 1; i < 8; ++i) l[i] = A[i * (i + 1) / 2 + k] * inv;
#pragma unroll
    for (int j = k + 1; j < 8; ++j)
#pragma unroll
      for (int i = j; i < 8; ++i) A[i * (i + 1) / 2 + j] = fma(-l[i], A[j * (j + 1) / 2 + k], A[i * (i + 1) / 2 + j]);
#pragma unroll
    for (int i = k + 1; i < 8; ++i) {
      y[i] = fma(-l[i], y[k], y[i]);
      A[i * (i + 1) / 2 + k] = l[i];
    }
  }
  // D z = y, L^T x = z
#pragma unroll
  for (int i = 7; i >= 0; --i) {
    double sacc = y[i] * dinv[i];
#pragma unroll
    for (int j = i + 1; j < 8; ++j) sacc = fma(-A[j * (j + 1) / 2 + i], y[j], sacc);
    y[i] = sacc;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = y[i];
}

/**
 * The LM control step of the persistent kernel: alignDecideWave's state machine and arithmetic (levenberg_marquardt_algorithm.hpp:77-128,
 * eigen_pose_alignment.cpp:101-104,174-212) in the form one wave executes fastest — one instruction per ~4.7 cycles whatever it does, so
 * the step costs what its instruction count says:
 *   * `tot` arrives with the affine prior already folded into its four entries, and the accepted state's system is kept in the same packed
 *     form (`acc_sys`): the solve reads ONE of the two through a selected pointer — one instance of the factorisation instead of two, no
 *     expanded 8 x 8 copies (H / H_used are not results of estimatePose);
 *   * decisions are selects, not branches (the branchy form spent a third of its instructions on register copies at the joins);
 *   * no f64 division: lambda / decrease_on_accept is a multiplication by the exact reciprocal of a power of two (checked by the host), and
 *     |e - e'| / e < tol is decided by |e - e'| < tol e outside a 1e-12 band around the threshold (inside it: the division).
 * All 64 lanes run it redundantly; lane 0 stores.
 */
__device__ __forceinline__ void pyramidDecide(AlignControl &c, const double *tot, double *acc_sys, double tgt_ab0, double tgt_ab1, double reg0, double reg1,
                                              double function_tolerance, double parameter_tolerance, double inv_decrease, double increase, int max_iterations) {
  const int lane = threadIdx.x & 63;
  // ---- loads
  const double cand_ab0 = c.cand_ab[0], cand_ab1 = c.cand_ab[1];
  double ab_eps0 = c.ab_eps[0], ab_eps1 = c.ab_eps[1];
  double energy = c.energy, lambda = c.lambda;
  int n_valid = c.n_valid, converged = c.converged, iteration = c.iteration;
  const bool first = c.have_candidate == 0;
  double stepv[8], Ttr[12], candT[12];
#pragma unroll
  for (int a = 0; a < 8; ++a) stepv[a] = c.step[a];
#pragma unroll
  for (int a = 0; a < 12; ++a) {
    Ttr[a] = c.T_tr[a];
    candT[a] = c.cand_T[a];
  }
  const double red_energy = tot[44], red_n = tot[45];
  const double sys_new = lane < 44 ? tot[lane] : 0.0;
  // ---- decision (uniform)
  const double tab0 = tgt_ab0 + cand_ab0, tab1 = tgt_ab1 + cand_ab1;
  const double e_eval = red_energy + 0.5 * (tab0 * reg0 * tab0 + tab1 * reg1 * tab1);
  const int n_eval = static_cast<int>(red_n + 0.5);
  const bool has = n_eval != 0, better = e_eval < energy;
  const double diff = fabs(energy - e_eval), thr = function_tolerance * energy;
  bool conv_f = diff < thr * (1.0 - 1e-12);
  if (!conv_f && !(diff > thr * (1.0 + 1e-12))) conv_f = diff / energy < function_tolerance;  // on the threshold (or not a number): the reference's own expression
  const double a0 = tgt_ab0 + ab_eps0, a1 = tgt_ab1 + ab_eps1;
  double step_sq = 0;
#pragma unroll
  for (int a = 0; a < 8; ++a) step_sq += stepv[a] * stepv[a];
  const bool conv_p = step_sq < parameter_tolerance * ((a0 * a0 + a1 * a1) + parameter_tolerance);
  PD_MARK(1);
  const bool accept = !first && has && better;   // acceptStep (eigen_pose_alignment.cpp:208-212)
  const bool take = first || accept;             // the evaluated state's system becomes the accepted state's
  iteration += first ? 0 : 1;
  converged = (!first && has && (conv_f || (better && conv_p))) ? 1 : converged;
  lambda = accept ? lambda * inv_decrease : ((!first && has) ? lambda * increase : lambda);  // rejectStep: the accepted state and its system stay
  energy = take ? e_eval : energy;
  n_valid = take ? n_eval : n_valid;
  int active = first ? ((max_iterations > 0 && n_eval > 0) ? 1 : 0) : ((has && !converged && iteration < max_iterations) ? 1 : 0);
#pragma unroll
  for (int a = 0; a < 12; ++a) Ttr[a] = accept ? candT[a] : Ttr[a];
  ab_eps0 = accept ? cand_ab0 : ab_eps0;
  ab_eps1 = accept ? cand_ab1 : ab_eps1;
  // ---- calculateStep (eigen_pose_alignment.cpp:194-206): (H + lambda diag(H)) step = b from the packed sums of the accepted state
  PD_MARK(2);
  // (a pointer the compiler cannot see through: left to itself it reads BOTH systems and selects entry by entry)
  using LdsDouble = const __attribute__((address_space(3))) double;
  LdsDouble *src = take ? (LdsDouble *)tot : (LdsDouble *)acc_sys;
  asm volatile("" : "+v"(src));
  double stepn[8];
  solve8Ldl([&](int i, int j) { return src[j * 8 - j * (j - 1) / 2 + (i - j)]; }, [&](int i) { return src[36 + i]; }, lambda, stepn);
  PD_MARK(3);
  if (take && lane < 44) acc_sys[lane] = sys_new;  // (behind the solve's reads: nothing waits for it)
  const Rigid E = rigidExp(stepn);
  PD_MARK(4);
  double Em[12], candTn[12];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) Em[4 * i + j] = E.R[3 * i + j];
    Em[4 * i + 3] = E.t[i];
  }
  mat34Compose(Em, Ttr, candTn);
  PD_MARK(5);
  // ---- one batch of stores (lane 0)
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 12; ++a) {
      c.T_tr[a] = Ttr[a];
      c.cand_T[a] = candTn[a];
    }
#pragma unroll
    for (int a = 0; a < 8; ++a) c.step[a] = stepn[a];
    c.ab_eps[0] = ab_eps0;
    c.ab_eps[1] = ab_eps1;
    c.cand_ab[0] = ab_eps0 - stepn[6];
    c.cand_ab[1] = ab_eps1 - stepn[7];
    c.lambda = lambda;
    c.energy = energy;
    c.n_valid = n_valid;
    c.converged = converged;
    c.active = active;
    c.iteration = iteration;
    c.have_candidate = 1;
  }
}

template <typename S>
__global__ void __launch_bounds__(kAlignThreads) alignPyramidKernel(AlignPyramidArgs a) {
  // one LDS block, the small hot arrays first: their addresses fit the 16-bit offset field of the LDS instructions (an address beyond
  // 64 KB costs a v_mov per access — the control step makes ~120 of them), the 72 KB of point rows behind
  struct PyramidLds {
    AlignControl sc;                         // (H, H_used, b: not maintained by this kernel)
    double tot[kAlignPartial];               // sums of the pass, affine prior folded in
    double acc_sys[kAlignPartial];           // system of the accepted state, same packed form
    double psum[kAlignWaves][kAlignPartial]; // per wave: its quarter of the participants' sums
    double wsum[kAlignWaves][kAlignPartial]; // per wave: packed sums of its points
    double T_cur[12], ab_cur[2];             // current estimate (T_target_reference rows, affine brightness)
    int failed, same_xcd, pad[2];
    double rows[kAlignWaves * kWaveRows];    // per wave: published point rows
  };
  __shared__ __attribute__((aligned(16))) PyramidLds lds;
  AlignControl &sc = lds.sc;
  double(&tot)[kAlignPartial] = lds.tot;
  double(&acc_sys)[kAlignPartial] = lds.acc_sys;
  double(&psum)[kAlignWaves][kAlignPartial] = lds.psum;
  double(&wsum)[kAlignWaves][kAlignPartial] = lds.wsum;
  double(&s_T_cur)[12] = lds.T_cur;
  double(&s_ab_cur)[2] = lds.ab_cur;
  int &s_failed = lds.failed, &s_same_xcd = lds.same_xcd;
  double *const rows = lds.rows;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD co-location (speed only): the host launches `spread` x G workgroups and every spread-th one takes part — the dispatcher
  // deals workgroups round-robin over the 8 XCDs, so with spread = 8 the participants tend to share ONE XCD and its L2.  Whether
  // they really do is measured, not assumed: every participant publishes its XCC id in the first pass (agent-scope stores, valid
  // for any placement) and only when all ids agree do the later passes publish with plain stores, which stay in that L2 where the
  // L1-bypassing polls of the others find them without a trip over the fabric.
  const int hyp = static_cast<int>(blockIdx.x) % a.spread;
  if (hyp >= a.n_hyp) return;
  const int blk = static_cast<int>(blockIdx.x) / a.spread;
  const int G = static_cast<int>(gridDim.x) / a.spread;
  // this hypothesis' exchange buffers, failed flag, result slot and buffer phase (locals: the kernel argument struct is never written)
  double *const h_partials = a.partials + static_cast<size_t>(hyp) * kPyramidSetDoubles;
  unsigned *const h_failed = reinterpret_cast<unsigned *>(h_partials + (kPyramidSetDoubles - 1));
  AlignPyramidResult *const h_out = a.out + hyp;
  const int h_start_phase = a.start_phase[hyp];
  unsigned xcc_id = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
  xcc_id &= 0xFu;
#ifdef DSOPP_HIP_STAMPS
  if (a.debug_poison_lds) {
    for (size_t i = tid; i < sizeof(PyramidLds) / 8; i += kAlignThreads) reinterpret_cast<unsigned long long *>(&lds)[i] = 0x7FF8DEADDEADBEEFull;
    __syncthreads();
  }
#endif
  if (tid == 0) s_same_xcd = 0;
  unsigned pass_global = 0;  // barriers passed so far (identical in every workgroup)
  // current estimate (T_target_reference rows, affine brightness): in LDS, not in every thread's registers
  if (tid < 12) s_T_cur[tid] = a.T_tr0[hyp][tid];
  if (tid < 2) s_ab_cur[tid] = a.ab0[tid];
  if (tid == 0) s_failed = 0;
  // where this lane's four result registers go in the packed sums (lane constants)
  int pk[4];
#pragma unroll
  for (int v = 0; v < 4; ++v) pk[v] = alignPackedIndex(lane, v);
  double *const my_rows = rows + wave * kWaveRows;
  // affine prior of the packed sums (eigen_pose_alignment.cpp:183-190) as lane constants: entry e gets pc0 + pc1 * tab0 + pc2 * tab1
  const double pc0 = lane == 33 ? a.affine_reg[0] : (lane == 35 ? a.affine_reg[1] : 0.0);
  const double pc1 = lane == 42 ? a.affine_reg[0] : 0.0, pc2 = lane == 43 ? a.affine_reg[1] : 0.0;
  int levels_done = 0, success = 1, lm_iterations = 0;
  for (int lvl = a.n_levels - 1; lvl >= 0; --lvl) {
    const AlignLevelDev &L = a.level[lvl];
    __syncthreads();  // s_T_cur / s_ab_cur of the previous level (or the start) are in place
    AlignFrameDev tgt = L.tgt;
    tgt.ab0[0] = s_ab_cur[0];
    tgt.ab0[1] = s_ab_cur[1];
    // per-level constants of the sweep set-up: ArrayReprojector ctor — camera_reproject.hpp:235-260
    const double ifx = 1.0 / L.ref.fx, ify = 1.0 / L.ref.fy, k02 = -L.ref.cx / L.ref.fx, k12 = -L.ref.cy / L.ref.fy;
    const double s_ratio = tgt.exposure / L.ref.exposure;
    __syncthreads();  // previous level's reads of sc are done
    // reset(); pushFrame(reference); pushFrame(target, current estimate): the control block dsopp_hip_aligner_solve prepares, written
    // straight into LDS one entry per thread (as a private AlignControl filled by one lane it lived in 1.4 KB of scratch)
    if (tid < 48) {
      acc_sys[tid] = 0;
    } else if (tid < 64) {
      // (nothing)
    } else if (tid < 64 + 12) {
      sc.T_tr[tid - 64] = s_T_cur[tid - 64];
    } else if (tid < 64 + 24) {
      sc.cand_T[tid - 76] = s_T_cur[tid - 76];
    } else if (tid < 64 + 32) {
      sc.step[tid - 88] = 0;
    } else if (tid == 96) {
      sc.ab_eps[0] = sc.ab_eps[1] = sc.cand_ab[0] = sc.cand_ab[1] = 0;
      sc.lambda = a.lambda0;
      sc.energy = 0;
      sc.n_valid = 0;
      sc.converged = 0;
      sc.active = 1;
      sc.iteration = 0;
      sc.have_candidate = 0;
      sc.linear_system_valid = 0;
      sc.pad0 = sc.pad1 = 0;
    }
    __syncthreads();
    // up to kPreload points per thread are read once per level (all of them when the level has <= kPreload * G * 256 points)
    constexpr int kPreload = 2;
    const bool preloaded = L.n_points <= kPreload * G * kAlignThreads;
    S ru[kPreload], rv[kPreload], rid[kPreload], rint[kPreload];
    if (preloaded) {
#pragma unroll
      for (int q = 0; q < kPreload; ++q) {
        const int i = blk * kAlignThreads + tid + q * G * kAlignThreads;
        const int ic = i < L.n_points ? i : (L.n_points > 0 ? L.n_points - 1 : 0);
        const bool have = L.n_points > 0;
        ru[q] = have ? static_cast<S>(L.pu[ic]) : S(0);
        rv[q] = have ? static_cast<S>(L.pv[ic]) : S(0);
        rid[q] = have ? static_cast<S>(L.pid[ic]) : S(0);
        rint[q] = have ? static_cast<S>(L.pint[ic]) : S(0);
      }
    }
    const int total_passes = a.max_iterations + 1;  // the initial evaluation + one per iteration, each followed by its control step
    int pass = 0;
    for (; pass < total_passes; ++pass) {
      AP_STAMP(0);
      // re-arm this workgroup's slots of the buffer the NEXT pass publishes into (its last readers finished two passes ago);
      // the stores complete behind the sweep and are waited for before this pass's sums go out
      const bool local_xcd = s_same_xcd != 0;  // (written once, behind barriers, after the first pass)
      auto publish = [&](double *addr, unsigned long long bits) {
        if (local_xcd)
          *(gu64 *)addr = bits;  // plain store: the line stays in this XCD's L2, which all participants share
        else
          __hip_atomic_store((gu64 *)addr, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      };
      // (the slot layout does not depend on G and the pass counter continues from the previous launch, so the buffers stay armed from
      // launch to launch and the host fills them only once: a workgroup re-arms its own slot and the slots blk + G, blk + 2 G, ...
      // no participant of THIS launch owns — a previous launch with more participants may have left sums there)
      const unsigned buf_pass = static_cast<unsigned>(h_start_phase) + pass_global;
      if (tid < kAlignPartial)
        for (int slot = blk; slot < kPyramidMaxWorkgroups; slot += G)
          publish(h_partials + (static_cast<size_t>((buf_pass + 1u) % kPyramidBuffers) * kPyramidMaxWorkgroups + slot) * kAlignPartial + tid, kPyramidSentinel);
      // ---- sweep of this workgroup's points at the candidate state; the wave's sums come out of the matrix cores
      const int first = blk * kAlignThreads + tid;
      const bool wg_has_points = blk * kAlignThreads < L.n_points;
      if (wg_has_points) {
        AlignSweepCtx<S> x;
        alignSweepSetup<S>(x, L.ref, tgt, sc.cand_T, sc.cand_ab[0], sc.cand_ab[1], a.sigma_huber, ifx, ify, k02, k12, s_ratio);
        AP_STAMP(1);
        af64x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        double d[8], r, wgt, en, vl;
        if (preloaded) {
          // this thread's points live in registers for the whole level: a pass starts with the texel gather.
          // point q of a thread exists on this level only when the level has more than q * G * 256 points: a grid-uniform test
#pragma unroll
          for (int q = 0; q < kPreload; ++q) {
            if (q > 0 && L.n_points <= q * G * kAlignThreads) break;
            alignPointEval<S>(x, ru[q], rv[q], rid[q], rint[q], first + q * G * kAlignThreads < L.n_points, d, r, wgt, en, vl);
            alignPublishRow(my_rows, d, r, wgt, en, vl);
            AP_STAMP(2);
            alignContract(my_rows, acc0, acc1);
          }
        } else {
          // (a wave-uniform trip count: every lane of a wave publishes a row — weight zero behind the end — until the wave's first lane runs out)
          const int wave_first = first - lane;
          for (int base = 0; wave_first + base < L.n_points; base += G * kAlignThreads) {
            const int i = first + base;
            const int ic = i < L.n_points ? i : L.n_points - 1;
            alignPointEval<S>(x, static_cast<S>(L.pu[ic]), static_cast<S>(L.pv[ic]), static_cast<S>(L.pid[ic]), static_cast<S>(L.pint[ic]),
                              i < L.n_points, d, r, wgt, en, vl);
            alignPublishRow(my_rows, d, r, wgt, en, vl);
            alignContract(my_rows, acc0, acc1);
          }
        }
        AP_STAMP(3);
#pragma unroll
        for (int v = 0; v < 4; ++v)
          if (pk[v] >= 0) wsum[wave][pk[v]] = acc0[v] + acc1[v];
      }
      __syncthreads();  // B1: the waves' sums are in LDS
      AP_STAMP(4);
      // ---- wave 0: the workgroup's sums go out
      if (wave == 0 && lane < kAlignPartial) {
        double *dst = h_partials + (static_cast<size_t>(buf_pass % kPyramidBuffers) * kPyramidMaxWorkgroups + blk) * kAlignPartial;
        double sv = 0;
        if (wg_has_points && lane < 46) sv = (wsum[0][lane] + wsum[1][lane]) + (wsum[2][lane] + wsum[3][lane]);
        // slots 46 / 47 (unused by the sums) carry x = XCC id + 1 and x^2: all ids are equal iff G * sum x^2 == (sum x)^2
        if (lane == 46) sv = static_cast<double>(xcc_id + 1u);
        if (lane == 47) sv = static_cast<double>((xcc_id + 1u) * (xcc_id + 1u));
        // (the re-arming stores of the previous pass target this address from the same lane of the same wave: in order)
        publish(dst + lane, static_cast<unsigned long long>(__double_as_longlong(sv)));
#ifdef DSOPP_HIP_STAMPS
        if (a.debug_sums && hyp == 0 && pass_global < 256u && (lane == 44 || lane == 0))
          a.debug_sums[(static_cast<size_t>(pass_global) * kPyramidMaxWorkgroups + blk) * 4 + (lane == 0 ? 3 : 2)] = lane == 0 ? sc.cand_T[3] : sv;
#endif
      }
      AP_STAMP(5);
      // ---- every wave polls a quarter of the participants (lane e < 48: entry e of the participants 8 w .. 8 w + 7) and adds them in a
      // fixed order: a value that is not the sentinel IS the published sum.  All loads of a poll are issued before the first test
      // (clamped indices + a 0 / 1 factor instead of predicated loads): one round trip per poll
      {
        constexpr int kPer = kPyramidMaxWorkgroups / kAlignWaves;  // 8
        bool failed = false;
        if (lane < kAlignPartial) {
          const double *src = h_partials + static_cast<size_t>(buf_pass % kPyramidBuffers) * kPyramidMaxWorkgroups * kAlignPartial + lane;
          unsigned long long w[kPer];
          unsigned spins = 0;
          for (;;) {
            bool ready = true;
#pragma unroll
            for (int j = 0; j < kPer; ++j) {
              const int bi = wave * kPer + j;
              const int bc = bi < G ? bi : G - 1;
              w[j] = __hip_atomic_load((const gu64 *)(src + static_cast<size_t>(bc) * kAlignPartial), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int j = 0; j < kPer; ++j) ready = ready && (w[j] != kPyramidSentinel);
            if (ready) break;
            if (++spins > (1u << 20) || __hip_atomic_load((gu32 *)h_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kPyramidFailed) {
              __hip_atomic_store((gu32 *)h_failed, kPyramidFailed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              failed = true;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
          double v[kPer];
#pragma unroll
          for (int j = 0; j < kPer; ++j) v[j] = (wave * kPer + j < G) ? __longlong_as_double(static_cast<long long>(w[j])) : 0.0;
          psum[wave][lane] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        if (__any(failed) && lane == 0) s_failed = 1;  // a workgroup never showed up (GPU shared with other work): the host falls back to launch-per-iteration
      }
      AP_STAMP(6);
      __syncthreads();  // B2: the four quarters are in LDS
      // ---- wave 0: total (affine prior folded in, eigen_pose_alignment.cpp:183-190) and the LM control step; the others wait at B3
      if (wave == 0 && !s_failed) {
        const double reg0 = a.affine_reg[0], reg1 = a.affine_reg[1];
        double total = 0;
        if (lane < kAlignPartial) {
          total = (psum[0][lane] + psum[1][lane]) + (psum[2][lane] + psum[3][lane]);
          const double tab0 = tgt.ab0[0] + sc.cand_ab[0], tab1 = tgt.ab0[1] + sc.cand_ab[1];
          tot[lane] = total + fma(pc2, tab1, fma(pc1, tab0, pc0));  // (+ 0 for the entries without a prior; one product, one sum for the others)
        }
#ifdef DSOPP_HIP_STAMPS
        if (a.debug_sums && hyp == 0 && pass_global < 256u && (lane == 44 || lane == 0))
          a.debug_sums[(static_cast<size_t>(pass_global) * kPyramidMaxWorkgroups + blk) * 4 + (lane == 0 ? 1 : 0)] = total;
#endif
        // first pass of the launch: do all participants sit on one XCD?  (identical sums in every workgroup -> identical verdict)
        if (pass_global == 0 && a.spread > 1) {
          const double sx = __shfl(total, 46), sxx = __shfl(total, 47);
          if (lane == 0) s_same_xcd = (static_cast<double>(G) * sxx == sx * sx) ? 1 : 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        AP_STAMP(7);
        pyramidDecide(sc, tot, acc_sys, tgt.ab0[0], tgt.ab0[1], reg0, reg1, a.function_tolerance, a.parameter_tolerance, a.inv_decrease,
                      a.increase_on_reject, a.max_iterations);
        AP_STAMP(8);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the publishing wave drains its stores (sums and re-arming)
      __syncthreads();                                   // B3: control block of the next pass (or the level's result) is in place
      AP_STAMP(9);
      ++pass_global;
      if (s_failed) {
        if (blk == 0 && tid == 0) h_out->failed = 1;
        return;
      }
      if (!sc.active) break;
    }
    // ---- the level's verdict (identical in every workgroup) — monocular_tracker.cpp:218-226, eigen_pose_alignment.cpp:320-328
    ++levels_done;
    const double rmse = sqrt(sc.energy / static_cast<double>(sc.n_valid));  // NaN without a valid residual: fails the test below
    lm_iterations += sc.iteration;
    if (blk == 0 && tid == 0) {
      h_out->rmse[lvl] = rmse;
      h_out->iterations[lvl] = sc.iteration;
      h_out->n_valid[lvl] = sc.n_valid;
    }
    if (!(rmse < L.rmse_limit)) {
      success = 0;
      break;
    }
    // accepted: the next finer level starts from this estimate
    __syncthreads();  // every thread has read the verdict's inputs from sc
    if (tid == 0) {
      Rigid Tfin;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) Tfin.R[3 * i + j] = sc.T_tr[4 * i + j];
        Tfin.t[i] = sc.T_tr[4 * i + 3];
      }
      rigidNormalize(Tfin);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) s_T_cur[4 * i + j] = Tfin.R[3 * i + j];
        s_T_cur[4 * i + 3] = Tfin.t[i];
      }
      s_ab_cur[0] += sc.ab_eps[0];
      s_ab_cur[1] += sc.ab_eps[1];
    }
  }
  __syncthreads();
  if (blk == 0 && tid == 0) {
    h_out->same_xcd = s_same_xcd;
    h_out->end_phase = static_cast<int>((static_cast<unsigned>(h_start_phase) + pass_global) % kPyramidBuffers);
    h_out->levels_done = levels_done;
    h_out->success = success;
    h_out->failed = 0;
    h_out->lm_iterations = lm_iterations;
    for (int i = 0; i < 12; ++i) h_out->T_tr[i] = s_T_cur[i];
    h_out->ab[0] = s_ab_cur[0];
    h_out->ab[1] = s_ab_cur[1];
  }
}

}  // namespace
}  // namespace dsopp_hip

using namespace dsopp_hip;

struct dsopp_hip_aligner {
  StreamRef sr;
  dsopp_hip_options opt;
  // frames_ of the reference: [0] = fixed reference (with points), back() = free target
  bool have_ref = false, have_tgt = false;
  AlignFrameDev ref{}, tgt{};
  int64_t ref_time = 0, tgt_time = 0;
  Rigid T_w_ref = rigidIdentity(), T_w_tgt = rigidIdentity();
  int n_points = 0;
  DeviceBuffer<double> d_u, d_v, d_id, d_int, d_partials[2];
  // the reference points the solve reads: the aligner's own buffers above or the per-level cache of a dsopp_hip_depth_maps
  const double *p_u = nullptr, *p_v = nullptr, *p_id = nullptr, *p_int = nullptr;
  AlignControl *h_ctrl = nullptr;  // pinned staging of the control block (upload + read-back)
  int lm_path = 0;                 // 0: automatic (single-workgroup loop for small point sets), 1: always one launch per iteration
  bool skip_covariance = false;    // estimate_pose: the per-level covariance is not read by the tracker loop
  bool pyramids_ordered = false;   // estimate_pose already ordered this stream behind both pyramids' builds (one wait per frame, not per level)
  // estimate_pose as one persistent launch over all levels (alignPyramidKernel)
  DeviceBuffer<double> d_pyr_partials;

  DeviceBuffer<AlignPyramidResult> d_pyr_out;
  AlignPyramidResult *h_pyr_out = nullptr;     // pinned
  int pyr_phase[kPyramidHypotheses] = {-1, -1, -1, -1, -1, -1, -1, -1};  // per hypothesis slot: buffer phase the next persistent launch starts
                                                                          // with; -1: the slot's exchange buffers have to be (re)armed by a fill
  int hypothesis_width = 0;  // initialisations per launch: 0 automatic (1 while tracking holds, 8 once a first try has failed), 1 .. 8 fixed
  int last_tries = 0;        // tries of the previous estimate_pose (automatic width)
  bool pyramid_kernel_disabled = false;        // a bounded spin timed out once (GPU shared with other work): stay on the launch-per-iteration path
  bool have_rotation_prior = false;  // setRotationPrior, cleared by reset() (eigen_pose_alignment.cpp:254-263)
  double rotation_prior[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  // launches the previous solve on a target level of this width needed: consecutive frames of a sequence need nearly the same
  // number of LM iterations per level, so the first batch is sized to that instead of a fixed 20 (launches past the end of
  // the loop are no-ops, but each still costs a dispatch)
  std::map<int, int> launches_needed;
  DeviceBuffer<int> d_rows;  // row counts / offsets of the device-side depth-map scan
  int *h_level_totals = nullptr;  // pinned: point counts of the levels scanned by one ensureAllLevelPoints
  DeviceBuffer<AlignControl> d_ctrl;
  std::map<int64_t, Rigid> known_poses;
};

namespace dsopp_hip {
namespace {

void setFrame(AlignFrameDev &f, const dsopp_hip_pyramid *p, int level, const double intr[4], double exposure, const double ab[2]) {
  const LevelView lv = p->view(level);
  f.texels = lv.texels;
  f.width = lv.width;
  f.height = lv.height;
  f.fx = intr[0];
  f.fy = intr[1];
  f.cx = intr[2];
  f.cy = intr[3];
  f.exposure = exposure;
  f.ab0[0] = ab[0];
  f.ab0[1] = ab[1];
}

void sampleReferenceIntensitiesImpl(hipStream_t st, const dsopp_hip_pyramid *pyr, int level, const double *u, const double *v, double *out, size_t n) {
  if (!n) return;
  const LevelView lv = pyr->view(level);
  const unsigned grid = static_cast<unsigned>((n + 255) / 256);
  if (pyr->dtype == DSOPP_HIP_F64)
    sampleReferenceKernel<double><<<grid, 256, 0, st>>>(static_cast<const Texel<double> *>(lv.texels), lv.width, u, v, out, static_cast<int>(n));
  else
    sampleReferenceKernel<float><<<grid, 256, 0, st>>>(static_cast<const Texel<float> *>(lv.texels), lv.width, u, v, out, static_cast<int>(n));
  HIP_CHECK(hipGetLastError());
}

void uploadPoints(dsopp_hip_aligner *a, const dsopp_hip_pyramid *pyr, int level, const std::vector<double> &u, const std::vector<double> &v,
                  const std::vector<double> &id) {
  const size_t n = u.size();
  hipStream_t st = a->sr.stream;
  a->d_u.reserve(std::max<size_t>(n, 1), 0, st);
  a->d_v.reserve(std::max<size_t>(n, 1), 0, st);
  a->d_id.reserve(std::max<size_t>(n, 1), 0, st);
  a->d_int.reserve(std::max<size_t>(n, 1), 0, st);
  a->d_u.upload(u.data(), n, 0, st);
  a->d_v.upload(v.data(), n, 0, st);
  a->d_id.upload(id.data(), n, 0, st);
  sampleReferenceIntensitiesImpl(st, pyr, level, a->d_u.ptr, a->d_v.ptr, a->d_int.ptr, n);
  a->sr.sync();
  a->n_points = static_cast<int>(n);
  a->p_u = a->d_u.ptr;
  a->p_v = a->d_v.ptr;
  a->p_id = a->d_id.ptr;
  a->p_int = a->d_int.ptr;
}

// ---- device-side scan of a reference depth map (LocalFrame depth-map ctor, PBA_INT/local_frame.hpp:367-392): a pixel
// becomes a reference point when it lies inside the 4-px border, its weight is positive and idepth = sum / weight >= 1e-6.
// The compaction keeps the row-major order of the reference's scan (= its summation order in the alignment).
__device__ inline bool depthMapPointValid(const double *idsum, const double *weight, int W, int H, int x, int y, double &id) {
  if (x < 4 || y < 4 || x >= W - 4 || y >= H - 4) return false;
  const size_t i = static_cast<size_t>(y) * W + x;
  const double w = weight[i];
  if (!(w > 0)) return false;
  id = idsum[i] / w;
  return !(id < 1e-6);
}

__global__ void countDepthMapRowsKernel(const double *__restrict__ idsum, const double *__restrict__ weight, int W, int H, int *row_count) {
  __shared__ int wave_sum[4];
  const int y = blockIdx.x;
  int c = 0;
  for (int x = threadIdx.x; x < W; x += 256) {
    double id;
    c += depthMapPointValid(idsum, weight, W, H, x, y, id) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) row_count[y] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

/** exclusive scan of the row counts (one workgroup of 256 threads, any H); row_offset[H] = total.  Every thread takes a run of consecutive
 *  rows (loads in flight together), the 256 run sums are scanned in LDS.  (Until round 6 one thread walked the rows: a dependent load per
 *  row, 47 us per level at 1280 x 1024 — 0.24 ms of the first estimatePose behind every keyframe.) */
__global__ void __launch_bounds__(256) scanDepthMapRowsKernel(const int *__restrict__ row_count, int H, int *__restrict__ row_offset) {
  __shared__ int part[256];
  const int tid = threadIdx.x;
  const int per = (H + 255) / 256, y0 = tid * per, y1 = min(y0 + per, H);
  int sum = 0;
  for (int y = y0; y < y1; ++y) sum += row_count[y];
  part[tid] = sum;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {  // Hillis-Steele inclusive scan
    const int v = tid >= o ? part[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int acc = part[tid] - sum;  // exclusive prefix of this thread's run
  for (int y = y0; y < y1; ++y) {
    row_offset[y] = acc;
    acc += row_count[y];
  }
  if (tid == 255) row_offset[H] = part[255];
}

__global__ void compactDepthMapRowsKernel(const double *__restrict__ idsum, const double *__restrict__ weight, int W, int H,
                                          const int *__restrict__ row_offset, double *u, double *v, double *idepth) {
  __shared__ int wave_cnt[4];
  __shared__ int running;
  const int y = blockIdx.x;
  if (threadIdx.x == 0) running = row_offset[y];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int x0 = 0; x0 < W; x0 += 256) {
    const int x = x0 + threadIdx.x;
    double id = 0;
    const bool ok = x < W && depthMapPointValid(idsum, weight, W, H, x, y, id);
    const unsigned long long m = __ballot(ok);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int base = running;
    for (int k = 0; k < wave; ++k) base += wave_cnt[k];
    if (ok) {
      const int dst = base + before;
      u[dst] = x;
      v[dst] = y;
      idepth[dst] = id;
    }
    __syncthreads();
    if (threadIdx.x == 0) running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
}


/** reference points of one level of device-resident depth maps (LocalFrame depth-map constructor, PBA_INT/local_frame.hpp:367-392),
 *  extracted on first use and cached with the maps: scan / compaction on the device in the reference's row-major order,
 *  intensities sampled from the keyframe's own pyramid */
dsopp_hip_depth_maps::LevelPoints &ensureLevelPoints(dsopp_hip_aligner *a, const dsopp_hip_depth_maps *maps, const dsopp_hip_pyramid *pyramid, int level) {
  hipStream_t st = a->sr.stream;
  const int W = pyramid->w(level), H = pyramid->h(level);
  dsopp_hip_depth_maps::LevelPoints &pts = maps->points[static_cast<size_t>(level)];
  if (pts.n < 0 || pts.pyramid != pyramid) {
    const double *idsum = maps->idepth_sum[static_cast<size_t>(level)].ptr, *wgt = maps->weight[static_cast<size_t>(level)].ptr;
    a->d_rows.reserve(2 * static_cast<size_t>(H) + 2, 0, st);
    int *row_count = a->d_rows.ptr, *row_offset = a->d_rows.ptr + H;
    countDepthMapRowsKernel<<<H, 256, 0, st>>>(idsum, wgt, W, H, row_count);
    scanDepthMapRowsKernel<<<1, 256, 0, st>>>(row_count, H, row_offset);
    int total = 0;
    HIP_CHECK(hipMemcpyAsync(&total, row_offset + H, sizeof(int), hipMemcpyDeviceToHost, st));
    a->sr.sync();
    const size_t n = static_cast<size_t>(total);
    pts.u.reserve(std::max<size_t>(n, 1), 0, st);
    pts.v.reserve(std::max<size_t>(n, 1), 0, st);
    pts.idepth.reserve(std::max<size_t>(n, 1), 0, st);
    pts.intensity.reserve(std::max<size_t>(n, 1), 0, st);
    if (n) compactDepthMapRowsKernel<<<H, 256, 0, st>>>(idsum, wgt, W, H, row_offset, pts.u.ptr, pts.v.ptr, pts.idepth.ptr);
    HIP_CHECK(hipGetLastError());
    sampleReferenceIntensitiesImpl(st, pyramid, level, pts.u.ptr, pts.v.ptr, pts.intensity.ptr, n);
    pts.markReady(st);
    a->sr.sync();
    pts.n = total;
    pts.pyramid = pyramid;
  }
  return pts;
}

/** ensureLevelPoints for all `levels` of the maps at once (estimatePose touches every level of fresh maps in the frame behind a keyframe):
 *  the scans of all stale levels are enqueued together, their totals come back with ONE copy and ONE synchronisation, then all
 *  compactions — instead of two synchronisations, a copy and a handful of allocations per level */
void ensureAllLevelPoints(dsopp_hip_aligner *a, const dsopp_hip_depth_maps *maps, const dsopp_hip_pyramid *pyramid, int levels) {
  hipStream_t st = a->sr.stream;
  int stale[DSOPP_HIP_MAX_LEVELS], n_stale = 0;
  size_t rows_total = 0;
  size_t first_row[DSOPP_HIP_MAX_LEVELS];
  for (int lvl = 0; lvl < levels; ++lvl) {
    if (pyramid->dtype != a->opt.dtype || pyramid->sr.device != a->sr.device || pyramid->w(lvl) != maps->width[static_cast<size_t>(lvl)] ||
        pyramid->h(lvl) != maps->height[static_cast<size_t>(lvl)])
      return;  // (the caller's per-level checks raise the error)
    const dsopp_hip_depth_maps::LevelPoints &pts = maps->points[static_cast<size_t>(lvl)];
    if (pts.n >= 0 && pts.pyramid == pyramid) continue;
    stale[n_stale++] = lvl;
    first_row[lvl] = rows_total;
    rows_total += 2 * static_cast<size_t>(pyramid->h(lvl)) + 2;
  }
  if (n_stale < 2) return;  // (one level: ensureLevelPoints does it as before)
  a->d_rows.reserve(rows_total, 0, st);
  if (!a->h_level_totals) HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&a->h_level_totals), DSOPP_HIP_MAX_LEVELS * sizeof(int), hipHostMallocDefault));
  for (int i = 0; i < n_stale; ++i) {
    const int lvl = stale[i], W = pyramid->w(lvl), H = pyramid->h(lvl);
    int *row_count = a->d_rows.ptr + first_row[lvl], *row_offset = row_count + H;
    countDepthMapRowsKernel<<<H, 256, 0, st>>>(maps->idepth_sum[static_cast<size_t>(lvl)].ptr, maps->weight[static_cast<size_t>(lvl)].ptr, W, H, row_count);
    scanDepthMapRowsKernel<<<1, 256, 0, st>>>(row_count, H, row_offset);
    HIP_CHECK(hipMemcpyAsync(a->h_level_totals + i, row_offset + H, sizeof(int), hipMemcpyDeviceToHost, st));
  }
  HIP_CHECK(hipGetLastError());
  a->sr.sync();
  for (int i = 0; i < n_stale; ++i) {
    const int lvl = stale[i], W = pyramid->w(lvl), H = pyramid->h(lvl);
    dsopp_hip_depth_maps::LevelPoints &pts = maps->points[static_cast<size_t>(lvl)];
    const int total = a->h_level_totals[i];
    const size_t n = static_cast<size_t>(total);
    pts.u.reserve(std::max<size_t>(n, 1), 0, st);
    pts.v.reserve(std::max<size_t>(n, 1), 0, st);
    pts.idepth.reserve(std::max<size_t>(n, 1), 0, st);
    pts.intensity.reserve(std::max<size_t>(n, 1), 0, st);
    const int *row_offset = a->d_rows.ptr + first_row[lvl] + H;
    if (n) compactDepthMapRowsKernel<<<H, 256, 0, st>>>(maps->idepth_sum[static_cast<size_t>(lvl)].ptr, maps->weight[static_cast<size_t>(lvl)].ptr, W, H, row_offset,
                                                         pts.u.ptr, pts.v.ptr, pts.idepth.ptr);
    HIP_CHECK(hipGetLastError());
    sampleReferenceIntensitiesImpl(st, pyramid, lvl, pts.u.ptr, pts.v.ptr, pts.intensity.ptr, n);
    pts.markReady(st);
    pts.n = total;
    pts.pyramid = pyramid;
  }
  // (no synchronisation: the consumers — the alignment kernels — are enqueued on this stream behind the compactions; d_rows is not reused
  // before the next call of this function or of ensureLevelPoints, both behind them on the same stream)
}

void checkPyramid(dsopp_hip_aligner *a, const dsopp_hip_pyramid *p, int level) {
  if (!p) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null pyramid");
  if (level < 0 || level >= p->levels) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "level out of range");
  if (p->dtype != a->opt.dtype) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "pyramid dtype differs from the aligner's");
  if (p->sr.device != a->sr.device) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "pyramid lives on another device");
  // dsopp_hip_pyramid_build_device only enqueues on the pyramid's stream: everything this aligner launches from here on must
  // see the finished texels
  if (!a->pyramids_ordered) p->waitReady(a->sr.stream);
}

}  // namespace
}  // namespace dsopp_hip

extern "C" {

int dsopp_hip_aligner_create(const dsopp_hip_options *options, int device, void *stream, dsopp_hip_aligner **out) {
  return guarded([&] {
    if (!options || !out) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    auto a = std::make_unique<dsopp_hip_aligner>();
    a->opt = *options;
    if (a->opt.dtype != DSOPP_HIP_F64 && a->opt.dtype != DSOPP_HIP_F32) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad dtype");
    a->sr.init(device, stream);
    a->d_ctrl.reserve(2, 0, a->sr.stream);
    a->sr.sync();
    *out = a.release();
  });
}

void dsopp_hip_aligner_destroy(dsopp_hip_aligner *a) {
  if (!a) return;
  (void)hipSetDevice(a->sr.device);
  if (a->sr.stream) (void)hipStreamSynchronize(a->sr.stream);
  if (a->h_ctrl) (void)hipHostFree(a->h_ctrl);
  if (a->h_pyr_out) (void)hipHostFree(a->h_pyr_out);
  if (a->h_level_totals) (void)hipHostFree(a->h_level_totals);
  StreamRef sr = a->sr;
  delete a;
  sr.destroy();
}

int dsopp_hip_aligner_reset(dsopp_hip_aligner *a) {
  return guarded([&] {
    if (!a) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null aligner");
    a->have_ref = a->have_tgt = false;
    a->n_points = 0;
    a->have_rotation_prior = false;  // prior_rotation_t_r_ = std::nullopt, eigen_pose_alignment.cpp:262
  });
}

int dsopp_hip_aligner_set_rotation_prior(dsopp_hip_aligner *a, const double *R_target_reference) {
  return guarded([&] {
    if (!a) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null aligner");
    a->have_rotation_prior = R_target_reference != nullptr;
    if (R_target_reference) {
      const double *R = R_target_reference;
      const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
      if (!(det > 0)) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "rotation prior has determinant %g", det);
      fitToSO3(R, a->rotation_prior);
    }
  });
}

int dsopp_hip_aligner_push_reference_points(dsopp_hip_aligner *a, int64_t timestamp, const double T_world_agent[7],
                                            const dsopp_hip_pyramid *pyramid, int level, const double intrinsics[4], int32_t n,
                                            const double *u, const double *v, const double *idepth, double exposure_time,
                                            const double affine_brightness[2]) {
  return guarded([&] {
    if (!a || !T_world_agent || !intrinsics || !affine_brightness || n < 0 || (n && (!u || !v || !idepth)))
      fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    checkPyramid(a, pyramid, level);
    if (a->have_ref || a->have_tgt) fail(DSOPP_HIP_ERR_ORDER, "the reference frame must be pushed first after reset()");
    a->sr.use();
    setFrame(a->ref, pyramid, level, intrinsics, exposure_time, affine_brightness);
    a->T_w_ref = rigidFromParams(T_world_agent);
    a->ref_time = timestamp;
    std::vector<double> uu(u, u + n), vv(v, v + n), dd(idepth, idepth + n);
    uploadPoints(a, pyramid, level, uu, vv, dd);
    a->have_ref = true;
  });
}

int dsopp_hip_aligner_push_reference_depth_map(dsopp_hip_aligner *a, int64_t timestamp, const double T_world_agent[7],
                                               const dsopp_hip_pyramid *pyramid, int level, const double intrinsics[4],
                                               const double *idepth_sum, const double *weight, double exposure_time,
                                               const double affine_brightness[2]) {
  return guarded([&] {
    if (!a || !T_world_agent || !intrinsics || !affine_brightness || !idepth_sum || !weight) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    checkPyramid(a, pyramid, level);
    if (a->have_ref || a->have_tgt) fail(DSOPP_HIP_ERR_ORDER, "the reference frame must be pushed first after reset()");
    a->sr.use();
    // LocalFrame depth-map ctor (PBA_INT/local_frame.hpp:367-392): row-major scan, 4-px border, weight > 0, idepth >= 1e-6.
    // The scan keeps the reference's point order (hence its summation order); intensities are sampled on the device.
    const int W = pyramid->w(level), H = pyramid->h(level);
    std::vector<double> uu, vv, dd;
    for (int y = 4; y < H - 4; ++y)
      for (int x = 4; x < W - 4; ++x) {
        const size_t i = static_cast<size_t>(y) * W + x;
        if (weight[i] > 0) {
          const double id = idepth_sum[i] / weight[i];
          if (id < 1e-6) continue;
          uu.push_back(x);
          vv.push_back(y);
          dd.push_back(id);
        }
      }
    setFrame(a->ref, pyramid, level, intrinsics, exposure_time, affine_brightness);
    a->T_w_ref = rigidFromParams(T_world_agent);
    a->ref_time = timestamp;
    uploadPoints(a, pyramid, level, uu, vv, dd);
    a->have_ref = true;
  });
}

int dsopp_hip_aligner_push_reference_depth_maps(dsopp_hip_aligner *a, int64_t timestamp, const double T_world_agent[7],
                                                const dsopp_hip_pyramid *pyramid, int level, const double intrinsics[4],
                                                const dsopp_hip_depth_maps *maps, double exposure_time, const double affine_brightness[2]) {
  return guarded([&] {
    if (!a || !T_world_agent || !intrinsics || !affine_brightness || !maps) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    checkPyramid(a, pyramid, level);
    if (a->have_ref || a->have_tgt) fail(DSOPP_HIP_ERR_ORDER, "the reference frame must be pushed first after reset()");
    if (level >= maps->levels) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "the depth maps have %d levels, level %d requested", maps->levels, level);
    if (maps->sr.device != a->sr.device) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "depth maps live on another device");
    const int W = pyramid->w(level), H = pyramid->h(level);
    if (W != maps->width[static_cast<size_t>(level)] || H != maps->height[static_cast<size_t>(level)])
      fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "depth map level %d is %d x %d, pyramid level is %d x %d", level, maps->width[static_cast<size_t>(level)],
           maps->height[static_cast<size_t>(level)], W, H);
    a->sr.use();
    hipStream_t st = a->sr.stream;
    if (maps->sr.stream != st) HIP_CHECK(hipStreamSynchronize(maps->sr.stream));  // producer finished (it synchronises at creation anyway)
    dsopp_hip_depth_maps::LevelPoints &pts = ensureLevelPoints(a, maps, pyramid, level);
    setFrame(a->ref, pyramid, level, intrinsics, exposure_time, affine_brightness);
    a->T_w_ref = rigidFromParams(T_world_agent);
    a->ref_time = timestamp;
    a->p_u = pts.u.ptr;
    a->p_v = pts.v.ptr;
    a->p_id = pts.idepth.ptr;
    a->p_int = pts.intensity.ptr;
    const int total = pts.n;
    a->n_points = total;
    a->have_ref = true;
  });
}

int dsopp_hip_aligner_push_target(dsopp_hip_aligner *a, int64_t timestamp, const double T_world_agent_init[7],
                                  const dsopp_hip_pyramid *pyramid, int level, const double intrinsics[4], double exposure_time,
                                  const double affine_brightness[2]) {
  return guarded([&] {
    if (!a || !T_world_agent_init || !intrinsics || !affine_brightness) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    checkPyramid(a, pyramid, level);
    if (a->have_ref && a->ref_time > timestamp) fail(DSOPP_HIP_ERR_ORDER, "frames must be processed in ascending order of time");
    setFrame(a->tgt, pyramid, level, intrinsics, exposure_time, affine_brightness);
    a->T_w_tgt = rigidFromParams(T_world_agent_init);
    a->tgt_time = timestamp;
    a->have_tgt = true;
  });
}

int dsopp_hip_aligner_push_known_pose(dsopp_hip_aligner *a, int64_t timestamp, const double T_world_agent[7]) {
  return guarded([&] {
    if (!a || !T_world_agent) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    a->known_poses[timestamp] = rigidFromParams(T_world_agent);
  });
}

int dsopp_hip_initialization_poses(const double T_world_previous[7], const double T_world_last[7], const double T_world_keyframe[7],
                                   int32_t capacity, double *poses, int32_t *n) {
  return guarded([&] {
    if (!poses || !n || capacity < 1) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    std::vector<Rigid> init;
    if (!T_world_previous || !T_world_last || !T_world_keyframe) {
      init.push_back(rigidIdentity());  // track.frames().size() < 2, monocular_tracker.cpp:138
    } else {
      const double kMin = 1. * M_PI / 180., kMax = 3. * M_PI / 180., kStep = 0.5 * M_PI / 180.;  // :139-141
      const Rigid t_w_r = rigidFromParams(T_world_last);
      const Rigid t_r_t_prev = rigidMul(rigidInverse(rigidFromParams(T_world_previous)), t_w_r);  // :145
      init.push_back(rigidMul(t_w_r, t_r_t_prev));                        // previous motion
      init.push_back(rigidMul(rigidMul(t_w_r, t_r_t_prev), t_r_t_prev));  // double previous motion (frame skipped)
      double xi[6];
      rigidLog(t_r_t_prev, xi);
      for (double &v : xi) v *= 0.5;
      init.push_back(rigidMul(t_w_r, rigidExp(xi)));                      // half motion
      init.push_back(t_w_r);                                              // zero motion
      init.push_back(rigidFromParams(T_world_keyframe));                  // zero motion from keyframe
      for (double delta = kMin; delta < kMax; delta += kStep)             // perturbed previous motion, :160-172
        for (double rx : {0.0, delta, -delta})
          for (double ry : {0.0, delta, -delta})
            for (double rz : {0.0, delta, -delta}) {
              const double rot[6] = {0, 0, 0, rx, ry, rz};
              init.push_back(rigidMul(init[0], rigidExp(rot)));
            }
    }
    *n = static_cast<int32_t>(init.size());
    for (size_t i = 0; i < init.size() && static_cast<int32_t>(i) < capacity; ++i) {
      Rigid T = init[i];
      rigidNormalize(T);
      rigidToParams(T, poses + 7 * i);
    }
  });
}

int dsopp_hip_aligner_set_lm_path(dsopp_hip_aligner *a, int path) {
  return guarded([&] {
    if (!a || path < 0 || path > 1) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad argument");
    a->lm_path = path;
  });
}

int dsopp_hip_aligner_num_points(dsopp_hip_aligner *a, int32_t *n) {
  return guarded([&] {
    if (!a || !n) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *n = a->n_points;
  });
}

int dsopp_hip_aligner_solve(dsopp_hip_aligner *a, dsopp_hip_align_result *result) {
  return guarded([&] {
    if (!a || !result) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (!a->have_tgt) fail(DSOPP_HIP_ERR_STATE, "no target frame pushed");
    std::memset(result, 0, sizeof(*result));
    // known pose short-cut — eigen_pose_alignment.cpp:281-288
    auto kp = a->known_poses.find(a->tgt_time);
    if (kp != a->known_poses.end()) {
      a->T_w_tgt = kp->second;
      result->rmse = -1;  // kZeroCost
      rigidToParams(a->T_w_tgt, result->T_world_target);
      result->affine_brightness[0] = a->tgt.ab0[0];
      result->affine_brightness[1] = a->tgt.ab0[1];
      return;
    }
    if (!a->have_ref) fail(DSOPP_HIP_ERR_STATE, "no reference frame pushed");
    a->sr.use();
    hipStream_t st = a->sr.stream;
    const int n = a->n_points;
    const int n_blocks = std::max(1, std::min(256, (n + kAlignThreads - 1) / kAlignThreads));
    a->d_partials[0].reserve(static_cast<size_t>(n_blocks) * kAlignPartial, 0, st);
    a->d_partials[1].reserve(static_cast<size_t>(n_blocks) * kAlignPartial, 0, st);
    AlignParams prm;
    prm.sigma_huber = a->opt.sigma_huber_loss;
    prm.affine_reg[0] = a->opt.affine_brightness_regularizer[0];
    prm.affine_reg[1] = a->opt.affine_brightness_regularizer[1];
    prm.function_tolerance = a->opt.function_tolerance;
    prm.parameter_tolerance = a->opt.parameter_tolerance;
    prm.decrease_on_accept = 2.0;  // eigen_pose_alignment.cpp:304-305
    prm.increase_on_reject = 2.0;
    prm.max_iterations = a->opt.max_iterations;
    prm.n_points = n;
    prm.n_blocks = n_blocks;
    if (!a->h_ctrl) HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&a->h_ctrl), 2 * sizeof(AlignControl), hipHostMallocDefault));
    AlignControl &c = a->h_ctrl[0];
    std::memset(&c, 0, sizeof(c));
    Rigid T_tr = rigidMul(rigidInverse(a->T_w_tgt), a->T_w_ref);  // eigen_pose_alignment.cpp:307-308
    if (a->have_rotation_prior)
      for (int i = 0; i < 9; ++i) T_tr.R[i] = a->rotation_prior[i];  // t_t_r.setRotationMatrix(prior), :309-311
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) c.T_tr[4 * i + j] = c.cand_T[4 * i + j] = T_tr.R[3 * i + j];
      c.T_tr[4 * i + 3] = c.cand_T[4 * i + 3] = T_tr.t[i];
    }
    c.lambda = 1.0 / a->opt.initial_trust_region_radius;
    c.active = 1;
    a->d_ctrl.reserve(2, 0, st);
    a->d_ctrl.upload(&c, 1, 0, st);  // pinned source: stream-ordered, no host wait
    const int total_launches = a->opt.max_iterations + 2;  // initial evaluation + one per iteration + final control pass
    int launch = 0;
    AlignControl &h = a->h_ctrl[1];
    const bool single_workgroup = n <= kAlignLoopMaxPoints && a->lm_path != 1;
    if (single_workgroup) {
      // the whole LM loop in one launch of one workgroup (alignLoopKernel): one enqueue, one read-back, one synchronisation
      const size_t smem = (static_cast<size_t>(kAlignPartial) * (kLoopCols + 2) + kAlignPartial) * sizeof(double);
      if (a->opt.dtype == DSOPP_HIP_F64)
        ensureDynamicLds(reinterpret_cast<const void *>(alignLoopKernel<double>), a->sr.device, 128 * 1024);
      else
        ensureDynamicLds(reinterpret_cast<const void *>(alignLoopKernel<float>), a->sr.device, 128 * 1024);
      if (a->opt.dtype == DSOPP_HIP_F64)
        alignLoopKernel<double><<<1, kLoopThreads, smem, st>>>(a->ref, a->tgt, a->p_u, a->p_v, a->p_id, a->p_int, a->d_ctrl.ptr, prm);
      else
        alignLoopKernel<float><<<1, kLoopThreads, smem, st>>>(a->ref, a->tgt, a->p_u, a->p_v, a->p_id, a->p_int, a->d_ctrl.ptr, prm);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipMemcpyAsync(&h, a->d_ctrl.ptr + 1, sizeof(AlignControl), hipMemcpyDeviceToHost, st));
      a->sr.sync();
    }
    // launches after the loop has ended are no-ops of ~2 us (plus the dispatch); the first batch covers what the previous
    // solve on this level needed + 2, follow-up batches are short; every batch ends with one read-back + sync
    int batch = 20;
    {
      auto it = a->launches_needed.find(a->tgt.width);
      if (it != a->launches_needed.end()) batch = std::max(6, it->second + 2);
    }
    while (!single_workgroup) {
      const int end = std::min(total_launches, launch + batch);
      batch = 8;
      for (; launch < end; ++launch) {
        const AlignControl *cin = a->d_ctrl.ptr + ((launch + 1) & 1);
        AlignControl *cout = a->d_ctrl.ptr + (launch & 1);
        if (launch == 0) cin = a->d_ctrl.ptr;  // host-prepared block
        const double *prev = a->d_partials[(launch + 1) & 1].ptr;
        double *cur = a->d_partials[launch & 1].ptr;
        if (a->opt.dtype == DSOPP_HIP_F64)
          alignIterationKernel<double><<<n_blocks, kAlignThreads, 0, st>>>(a->ref, a->tgt, a->p_u, a->p_v, a->p_id, a->p_int, cin, cout, prev, cur, prm, launch);
        else
          alignIterationKernel<float><<<n_blocks, kAlignThreads, 0, st>>>(a->ref, a->tgt, a->p_u, a->p_v, a->p_id, a->p_int, cin, cout, prev, cur, prm, launch);
      }
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipMemcpyAsync(&h, a->d_ctrl.ptr + ((launch - 1) & 1), sizeof(AlignControl), hipMemcpyDeviceToHost, st));
      a->sr.sync();
      if (!h.active || launch >= total_launches) break;
    }
    if (!single_workgroup) a->launches_needed[a->tgt.width] = h.iteration + 2;  // initial evaluation + iterations + closing control pass
    // result — eigen_pose_alignment.cpp:320-328
    Rigid Tfin;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) Tfin.R[3 * i + j] = h.T_tr[4 * i + j];
      Tfin.t[i] = h.T_tr[4 * i + 3];
    }
    rigidNormalize(Tfin);
    a->T_w_tgt = rigidMul(a->T_w_ref, rigidInverse(Tfin));
    a->tgt.ab0[0] += h.ab_eps[0];
    a->tgt.ab0[1] += h.ab_eps[1];
    result->energy = h.energy;
    result->n_valid = h.n_valid;
    result->iterations = h.iteration;
    result->rmse = std::sqrt(h.energy / static_cast<double>(h.n_valid) / 1.0);
    rigidToParams(a->T_w_tgt, result->T_world_target);
    result->affine_brightness[0] = a->tgt.ab0[0];
    result->affine_brightness[1] = a->tgt.ab0[1];
    if (!a->skip_covariance) {
      hostla::Mat Hm(h.H_used, h.H_used + 64);
      const hostla::Mat pinv = hostla::pinvRankRevealing(Hm, 8);
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) result->covariance[6 * i + j] = pinv[static_cast<size_t>(8 * i + j)];
    }
    std::memcpy(result->H, h.H_used, sizeof(h.H_used));
  });
}


int dsopp_hip_aligner_set_hypothesis_width(dsopp_hip_aligner *a, int32_t width) {
  return guarded([&] {
    if (!a) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null aligner");
    if (width < 0 || width > kPyramidHypotheses) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "hypothesis width must be 0 (automatic) .. %d", kPyramidHypotheses);
    a->hypothesis_width = width;
  });
}

int dsopp_hip_aligner_estimate_pose(dsopp_hip_aligner *a, int64_t reference_time, const double T_world_reference[7],
                                    const dsopp_hip_pyramid *reference_pyramid, const dsopp_hip_depth_maps *reference_depth_maps,
                                    double reference_exposure, const double reference_affine[2], int64_t target_time,
                                    const dsopp_hip_pyramid *target_pyramid, double target_exposure, const double intrinsics[4],
                                    int32_t n_initializations, const double *T_world_target_init, const double affine_init[2],
                                    double *rmse_last_pose_estimation, double T_world_target[7], double affine_brightness[2],
                                    int32_t *success_out, int32_t *tries_out, int32_t *lm_iterations_out) {
  if (!a || !T_world_reference || !reference_pyramid || !reference_depth_maps || !reference_affine || !target_pyramid || !intrinsics ||
      n_initializations < 1 || !T_world_target_init || !affine_init || !rmse_last_pose_estimation || !T_world_target || !affine_brightness) {
    return guarded([] { fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null / empty argument"); });
  }
  // estimatePose — src/tracker/tracker/src/monocular_tracker.cpp:179-245
  const double kEnergyRatioThreshold = 2.5;
  const int levels = target_pyramid->levels;
  if (levels > reference_depth_maps->levels || levels > reference_pyramid->levels)
    return guarded([&] { fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "target pyramid has %d levels, reference pyramid / depth maps fewer", levels); });
  std::vector<double> local_rmse(static_cast<size_t>(levels));
  struct SkipCov {
    dsopp_hip_aligner *a;
    explicit SkipCov(dsopp_hip_aligner *x) : a(x) { a->skip_covariance = true; }
    ~SkipCov() { a->skip_covariance = false; }
  } skip_cov(a);
  struct OrderedOnce {
    dsopp_hip_aligner *a;
    explicit OrderedOnce(dsopp_hip_aligner *x) : a(x) {}
    ~OrderedOnce() { a->pyramids_ordered = false; }
  } ordered_once(a);
  {
    const int rc = guarded([&] {
      a->sr.use();
      reference_pyramid->waitReady(a->sr.stream);
      target_pyramid->waitReady(a->sr.stream);
    });
    if (rc != DSOPP_HIP_OK) return rc;
    a->pyramids_ordered = true;
  }
  double T[7], ab[2], T_const[7] = {0, 0, 0, 1, 0, 0, 0}, ab_const[2] = {0, 0};
  bool success = false;
  int tries = 0, lm_iterations = 0;
  int try_number = 0;
  while (!success && try_number < n_initializations) {
    // ---- persistent launches over all levels (alignPyramidKernel), up to kPyramidHypotheses initialisations per launch; the
    // launch-per-iteration loop below is the fallback (lm_path 1, a known pose for this frame, or a spin time-out on a GPU shared with
    // other work).  Width of a batch: while tracking holds the first initialisation succeeds (previous motion), so it runs alone — one
    // XCD, nothing wasted; once a first try has failed (this call or the previous one) the remaining initialisations go out 8 at a time.
    if (a->lm_path == 0 && !a->pyramid_kernel_disabled && a->known_poses.find(target_time) == a->known_poses.end()) {
      const int auto_width = (try_number == 0 && a->last_tries <= 1) ? 1 : kPyramidHypotheses;
      const int width = a->hypothesis_width > 0 ? a->hypothesis_width : auto_width;
      const int nb = std::min(width, n_initializations - try_number);
      bool fast = false;  // the batch was decided by the persistent launch
      const int rc = guarded([&] {
        a->sr.use();
        hipStream_t st = a->sr.stream;
        AlignPyramidArgs args;
        std::memset(&args, 0, sizeof(args));
        int max_blocks = 1;
        const double zero_ab[2] = {0, 0};
        if (reference_depth_maps->sr.device == a->sr.device) {
          if (reference_depth_maps->sr.stream != st) HIP_CHECK(hipStreamSynchronize(reference_depth_maps->sr.stream));
          ensureAllLevelPoints(a, reference_depth_maps, reference_pyramid, levels);  // fresh maps: all levels in one round trip
        }
        for (int lvl = 0; lvl < levels; ++lvl) {
          checkPyramid(a, reference_pyramid, lvl);
          checkPyramid(a, target_pyramid, lvl);
          if (reference_depth_maps->sr.device != a->sr.device) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "depth maps live on another device");
          if (reference_pyramid->w(lvl) != reference_depth_maps->width[static_cast<size_t>(lvl)] ||
              reference_pyramid->h(lvl) != reference_depth_maps->height[static_cast<size_t>(lvl)])
            fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "depth map level %d does not match the pyramid level", lvl);
          if (reference_depth_maps->sr.stream != st) HIP_CHECK(hipStreamSynchronize(reference_depth_maps->sr.stream));
          const dsopp_hip_depth_maps::LevelPoints &pts = ensureLevelPoints(a, reference_depth_maps, reference_pyramid, lvl);
          const double sl = static_cast<double>(1 << lvl);  // CameraCalibration::cameraModel(level), camera_calibration.cpp:66-70
          const double intr[4] = {intrinsics[0] / sl, intrinsics[1] / sl, intrinsics[2] / sl, intrinsics[3] / sl};
          AlignLevelDev &L = args.level[lvl];
          setFrame(L.ref, reference_pyramid, lvl, intr, reference_exposure, reference_affine);
          setFrame(L.tgt, target_pyramid, lvl, intr, target_exposure, zero_ab);
          L.pu = pts.u.ptr;
          L.pv = pts.v.ptr;
          L.pid = pts.idepth.ptr;
          L.pint = pts.intensity.ptr;
          L.n_points = pts.n;
          // (every try starts from the caller's rmse_last_pose_estimation: monocular_tracker.cpp:198)
          L.rmse_limit = kEnergyRatioThreshold * rmse_last_pose_estimation[lvl];
          max_blocks = std::max(max_blocks, (pts.n + kAlignThreads - 1) / kAlignThreads);
        }
        const int G = std::min(kPyramidMaxWorkgroups, max_blocks);
        args.n_levels = levels;
        args.max_iterations = a->opt.max_iterations;
        args.sigma_huber = a->opt.sigma_huber_loss;
        args.affine_reg[0] = a->opt.affine_brightness_regularizer[0];
        args.affine_reg[1] = a->opt.affine_brightness_regularizer[1];
        args.function_tolerance = a->opt.function_tolerance;
        args.parameter_tolerance = a->opt.parameter_tolerance;
        args.decrease_on_accept = 2.0;  // eigen_pose_alignment.cpp:304-305
        args.increase_on_reject = 2.0;
        args.inv_decrease = 0.5;  // exactly 1 / decrease_on_accept
        args.lambda0 = 1.0 / a->opt.initial_trust_region_radius;
        args.n_hyp = nb;
        for (int h = 0; h < nb; ++h) {
          const Rigid T_tr = rigidMul(rigidInverse(rigidFromParams(T_world_target_init + 7 * (try_number + h))),
                                      rigidFromParams(T_world_reference));  // eigen_pose_alignment.cpp:307-308
          for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) args.T_tr0[h][4 * i + j] = T_tr.R[3 * i + j];
            args.T_tr0[h][4 * i + 3] = T_tr.t[i];
          }
        }
        args.ab0[0] = affine_init[0];
        args.ab0[1] = affine_init[1];
        static_assert(kAlignPartial == 48, "kPyramidSetDoubles");
        a->d_pyr_partials.reserve(kPyramidHypotheses * kPyramidSetDoubles, 0, st);
        if (!a->h_pyr_out)
          HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&a->h_pyr_out), kPyramidHypotheses * sizeof(AlignPyramidResult), hipHostMallocDefault));
        args.partials = a->d_pyr_partials.ptr;
        args.out = a->h_pyr_out;  // pinned host memory: the kernel leaves its results there itself (no copy kernel behind it)
        // every slot armed with the sentinel, the failed flag with the same (!= kPyramidFailed) pattern: one fill per hypothesis slot —
        // before its first launch and after a failed one only; a launch leaves the buffers armed for its successor (the pass counter
        // continues, the kernel re-arms by its rotation rule), which saves two fill kernels (10 us) per tracked frame
        for (int h = 0; h < nb; ++h) {
          if (a->pyr_phase[h] < 0) {
            HIP_CHECK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(a->d_pyr_partials.ptr + static_cast<size_t>(h) * kPyramidSetDoubles),
                                        static_cast<int>(kPyramidSentinelWord), 2 * kPyramidSetDoubles, st));
            a->pyr_phase[h] = 0;
          }
          args.start_phase[h] = a->pyr_phase[h];
          a->pyr_phase[h] = -1;  // (until this launch has reported how it left the buffers)
        }
#ifdef DSOPP_HIP_STAMPS
        static const bool check_sums = std::getenv("DSOPP_HIP_CHECK_SUMS") != nullptr;
        static double *dbg_dev = nullptr;
        constexpr size_t kDbgDoubles = 256 * kPyramidMaxWorkgroups * 4;
        if (check_sums) {
          if (!dbg_dev) HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&dbg_dev), kDbgDoubles * sizeof(double)));
          HIP_CHECK(hipMemsetAsync(dbg_dev, 0, kDbgDoubles * sizeof(double), st));
          args.debug_sums = dbg_dev;
        }
        static const bool poison = std::getenv("DSOPP_HIP_POISON_LDS") != nullptr;
        args.debug_poison_lds = poison ? 1 : 0;
#endif
        static const int spread_override = std::getenv("DSOPP_HIP_ALIGN_SPREAD") ? std::atoi(std::getenv("DSOPP_HIP_ALIGN_SPREAD")) : 0;  // tuning aid
        args.spread = (nb == 1 && spread_override > 0) ? spread_override : 8;
        if (a->opt.dtype == DSOPP_HIP_F64)
          alignPyramidKernel<double><<<G * args.spread, kAlignThreads, 0, st>>>(args);
        else
          alignPyramidKernel<float><<<G * args.spread, kAlignThreads, 0, st>>>(args);
        HIP_CHECK(hipGetLastError());
        a->sr.sync();
#ifdef DSOPP_HIP_STAMPS
        if (std::getenv("DSOPP_HIP_TRACE")) {
          const AlignPyramidResult &o = a->h_pyr_out[0];
          auto us = [&](int i, int j) { return (o.stamps[j] - o.stamps[i]) / 100.0; };
          std::fprintf(stderr, "alignPyramid pass (level 0, wave 0 of workgroup 0): set-up %.2f  gather + point %.2f  rows + matrix cores %.2f  wave sums + barrier %.2f  "
                               "workgroup sum + publish %.2f  poll + quarter sums %.2f  barrier + total %.2f  control step %.2f  drain + barrier %.2f  = %.2f us (G = %d, one XCD: %d)\n",
                       us(0, 1), us(1, 2), us(2, 3), us(3, 4), us(4, 5), us(5, 6), us(6, 7), us(7, 8), us(8, 9), us(0, 9), G, o.same_xcd);
        }
#endif
#ifdef DSOPP_HIP_STAMPS
        if (check_sums) {
          // every workgroup has to derive the same totals in every pass (a difference is a stale or missed value in the exchange), and a
          // repeated call on the same inputs has to reproduce the first call's partial sums and totals bit for bit
          std::vector<double> dbg(kDbgDoubles);
          HIP_CHECK(hipMemcpy(dbg.data(), dbg_dev, kDbgDoubles * sizeof(double), hipMemcpyDeviceToHost));
          static std::map<int, std::vector<double>> reference;  // by level count x participants
          for (int pass = 0; pass < 256; ++pass) {
            const double *row = dbg.data() + static_cast<size_t>(pass) * kPyramidMaxWorkgroups * 4;
            for (int w = 0; w < G; ++w)
              if (std::memcmp(row + 4 * w, row, 2 * sizeof(double)) != 0)
                std::fprintf(stderr, "SUMS DIFFER pass %d: workgroup %d has energy %.17g H00 %.17g, workgroup 0 %.17g %.17g\n", pass, w, row[4 * w], row[4 * w + 1], row[0], row[1]);
          }
          const int key = levels * 1000 + G;
          auto it = reference.find(key);
          if (it == reference.end()) {
            reference[key] = dbg;
          } else {
            bool reported = false;
            for (int pass = 0; pass < 256 && !reported; ++pass)
              for (int w = 0; w < G && !reported; ++w) {
                const double *x = dbg.data() + (static_cast<size_t>(pass) * kPyramidMaxWorkgroups + w) * 4, *y = it->second.data() + (x - dbg.data());
                if (std::memcmp(x, y, 4 * sizeof(double)) != 0) {
                  std::fprintf(stderr, "FIRST DIFFERENCE to the first call: pass %d workgroup %d: own partial energy %.17g (first call %.17g) candidate t_x %.17g (%.17g); total energy %.17g (%.17g) "
                                       "H00 %.17g (%.17g)\n", pass, w, x[2], y[2], x[3], y[3], x[0], y[0], x[1], y[1]);
                  // the whole pass: which workgroups' partials differ
                  for (int w2 = 0; w2 < G; ++w2) {
                    const double *x2 = dbg.data() + (static_cast<size_t>(pass) * kPyramidMaxWorkgroups + w2) * 4, *y2 = it->second.data() + (x2 - dbg.data());
                    if (x2[2] != y2[2] || x2[3] != y2[3])
                      std::fprintf(stderr, "   workgroup %d differs: partial energy %.17g (%.17g), its candidate's t_x %.17g (%.17g)\n", w2, x2[2], y2[2], x2[3], y2[3]);
                  }
                  reported = true;
                }
              }
          }
        }
#endif
        static const bool trace_levels = std::getenv("DSOPP_HIP_TRACE") != nullptr;  // debugging aid: what every level of hypothesis 0 did
        if (trace_levels) {
          const AlignPyramidResult &o = a->h_pyr_out[0];
          for (int lvl = levels - 1; lvl >= 0; --lvl)
            std::fprintf(stderr, "  level %d: %d points, %d iterations, n_valid %d, rmse %.12g (G = %d)\n", lvl, args.level[lvl].n_points, o.iterations[lvl],
                         o.n_valid[lvl], o.rmse[lvl], G);
        }
        for (int h = 0; h < nb; ++h)
          if (a->h_pyr_out[h].failed) {
            a->pyramid_kernel_disabled = true;  // not all workgroups were resident in time: this GPU is busy with something else
            return;
          }
        for (int h = 0; h < nb; ++h) a->pyr_phase[h] = a->h_pyr_out[h].end_phase;  // armed for a launch that continues the pass counter
        fast = true;
      });
      if (rc != DSOPP_HIP_OK) return rc;
      if (fast) {
        // the batch in the order of the sequential loop: the first success ends it
        for (int h = 0; h < nb && !success; ++h) {
          const AlignPyramidResult &o = a->h_pyr_out[h];
          ++tries;
          lm_iterations += o.lm_iterations;
          std::memcpy(T, T_world_target_init + 7 * (try_number + h), sizeof(T));
          ab[0] = affine_init[0];
          ab[1] = affine_init[1];
          const int accepted = o.success ? o.levels_done : o.levels_done - 1;  // the last level run failed its energy test
          if (accepted > 0) {  // (no level accepted: the initialisation itself is what the reference keeps)
            Rigid Tfin;
            for (int i = 0; i < 3; ++i) {
              for (int j = 0; j < 3; ++j) Tfin.R[3 * i + j] = o.T_tr[4 * i + j];
              Tfin.t[i] = o.T_tr[4 * i + 3];
            }
            rigidToParams(rigidMul(rigidFromParams(T_world_reference), rigidInverse(Tfin)), T);
            ab[0] = o.ab[0];
            ab[1] = o.ab[1];
          }
          if (try_number + h == 0) {
            std::memcpy(T_const, T, sizeof(T));
            ab_const[0] = ab[0];
            ab_const[1] = ab[1];
          }
          if (o.success) {
            success = true;
            std::copy(rmse_last_pose_estimation, rmse_last_pose_estimation + levels, local_rmse.begin());
            for (int k = 0; k < accepted; ++k) {
              const int lvl = levels - 1 - k;
              local_rmse[static_cast<size_t>(lvl)] = o.rmse[lvl];
            }
          }
        }
        try_number += nb;
        continue;
      }
    }
    // ---- one initialisation, one launch per Levenberg-Marquardt iteration
    ++tries;
    success = true;
    std::memcpy(T, T_world_target_init + 7 * try_number, sizeof(T));
    ab[0] = affine_init[0];
    ab[1] = affine_init[1];
    std::copy(rmse_last_pose_estimation, rmse_last_pose_estimation + levels, local_rmse.begin());
    for (int lvl = levels - 1; success && lvl >= 0; --lvl) {
      const double s = static_cast<double>(1 << lvl);  // CameraCalibration::cameraModel(level), camera_calibration.cpp:66-70
      const double intr[4] = {intrinsics[0] / s, intrinsics[1] / s, intrinsics[2] / s, intrinsics[3] / s};
      int rc = dsopp_hip_aligner_reset(a);
      if (rc == DSOPP_HIP_OK)
        rc = dsopp_hip_aligner_push_reference_depth_maps(a, reference_time, T_world_reference, reference_pyramid, lvl, intr, reference_depth_maps,
                                                         reference_exposure, reference_affine);
      if (rc == DSOPP_HIP_OK) rc = dsopp_hip_aligner_push_target(a, target_time, T, target_pyramid, lvl, intr, target_exposure, ab);
      dsopp_hip_align_result r;
      if (rc == DSOPP_HIP_OK) rc = dsopp_hip_aligner_solve(a, &r);
      if (rc != DSOPP_HIP_OK) return rc;
      lm_iterations += r.iterations;
      if (r.rmse < kEnergyRatioThreshold * local_rmse[static_cast<size_t>(lvl)]) {
        std::memcpy(T, r.T_world_target, sizeof(T));
        ab[0] = r.affine_brightness[0];
        ab[1] = r.affine_brightness[1];
        if (r.rmse != -1.0) local_rmse[static_cast<size_t>(lvl)] = r.rmse;  // kZeroCost
      } else {
        success = false;
      }
    }
    if (try_number == 0) {
      std::memcpy(T_const, T, sizeof(T));
      ab_const[0] = ab[0];
      ab_const[1] = ab[1];
    }
    ++try_number;
  }
  a->last_tries = tries;
  if (!success) {
    std::memcpy(T, T_const, sizeof(T));
    ab[0] = ab_const[0];
    ab[1] = ab_const[1];
    for (int l = 0; l < levels; ++l) rmse_last_pose_estimation[l] *= kEnergyRatioThreshold;  // pose invalid: assume maximum energy
  } else {
    std::copy(local_rmse.begin(), local_rmse.end(), rmse_last_pose_estimation);
  }
  std::memcpy(T_world_target, T, sizeof(T));
  affine_brightness[0] = ab[0];
  affine_brightness[1] = ab[1];
  if (success_out) *success_out = success ? 1 : 0;
  if (tries_out) *tries_out = tries;
  if (lm_iterations_out) *lm_iterations_out = lm_iterations;
  return DSOPP_HIP_OK;
}

}  // extern "C"
