// Reduction, Schur complement, dense solve and Levenberg-Marquardt control kernels (hot loops B, C, D of SURVEY.md §3.2
// and the LM driver, levenberg_marquardt_algorithm.hpp:77-128, kept on the device so that a whole solve is one stream
// of launches with no host round trip).
#pragma once
#include <cstddef>
#include <type_traits>
#include "pba_kernels.hpp"

namespace dsopp_hip {

// ---------------------------------------------------------------------------------------------------------------
// LM control block (lives in HBM; double-buffered by iteration parity so every workgroup of the decide kernel can read
// the incoming state while workgroup 0 writes the outgoing one)
// ---------------------------------------------------------------------------------------------------------------
// (struct LmControl: pba_types.hpp)

struct LmParams {
  double function_tolerance, parameter_tolerance;
  double decrease_on_accept, increase_on_reject;
  double lambda0;
  int max_iterations, min_iterations;
  int force_accept;
  int use_reduced_scalars;  // 1: (energy, n_valid, step^2, idepth.step) were summed (across ranks) into `scalars`;
                            // 2: `scalars` holds kScalarGroups group sums of them (sweepScalarGroupsKernel: large windows)
};

// ---------------------------------------------------------------------------------------------------------------
// K2: per-pair reduction + Schur complement
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSchurLandmarks = 64;
constexpr int kSchurThreads = 512;  // 8 lanes per landmark in the finalisation phase, 8 waves for the MFMA tiles
constexpr int kPairBlk = 144;  // per pair: TGT[64] | GT[64] | T^T q [8] | pad

/** LDS row stride (in doubles) for K-wide rows read as 16-lane x 4-row MFMA operands without bank conflicts:
 *  smallest s >= Kp with s % 32 == 16 (ds_read_b64 banks = (addr/4) % 64, two rows per 32-lane group) */
__host__ __device__ inline int schurRowStride(int Kp) {
  return ((Kp + 15) / 32) * 32 + 16;  // smallest s >= Kp with s % 32 == 16 (closed form: a search loop here gets strength-reduced into the kernel)
}

/** derived per-pair blocks from G, q:  TGT = T^T G T (H_rr contribution), GT = G T (H_rt = -(GT)^T), Tq = T^T q (b_r),
 *  T = blockdiag(Adj, 1, s0).  64 lanes. */
__device__ inline void derivePairBlocks(const double *__restrict__ G /* [48] */, const PairConst &P, double *__restrict__ out /* [kPairBlk] */,
                                        double *lds /* [64] scratch */, int lane) {
  const int i = lane >> 3, j = lane & 7;
  double gt;
  if (j < 6) {
    gt = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) gt += G[symIdx(i, k)] * P.Adj[6 * k + j];
  } else if (j == 6) {
    gt = G[symIdx(i, 6)];
  } else {
    gt = G[symIdx(i, 7)] * P.s0;
  }
  lds[lane] = gt;
  out[64 + lane] = gt;
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  double tgt;
  if (i < 6) {
    tgt = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) tgt += P.Adj[6 * k + i] * lds[8 * k + j];
  } else if (i == 6) {
    tgt = lds[48 + j];
  } else {
    tgt = P.s0 * lds[56 + j];
  }
  out[lane] = tgt;
  if (lane < 8) {
    const double *q = G + 36;
    double s;
    if (lane < 6) {
      s = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) s += P.Adj[6 * k + lane] * q[k];
    } else if (lane == 6) {
      s = q[6];
    } else {
      s = P.s0 * q[7];
    }
    out[128 + lane] = s;
  }
}

/** combined system layout (fused LM loop): 8 x 8 block (bi, bj), bj <= bi, at combBlockIndex * 64, row-major inside */
__host__ __device__ constexpr int combBlockIndex(int bi, int bj) { return bi * (bi + 1) / 2 + bj; }
__host__ __device__ constexpr int combBlockCount(int F) { return F * (F + 1) / 2; }
/** address of entry (row, col), row >= col or same diagonal block, in the combined system */
__device__ __forceinline__ int combIndex(int row, int col) { return combBlockIndex(row >> 3, col >> 3) * 64 + ((row & 7) << 3) + (col & 7); }

struct ReduceSchurArgs {
  const FrameDev *frames;
  const PairConst *pc;
  const SchurBlock *schur_table;
  const double *partials;
  const int *pair_first_block, *pair_num_blocks;
  double *Hpp, *bpp;  // pose-pose system without priors, accumulated with atomics (K x K full, K)
  double *Hsc, *bsc;  // Schur system, upper triangle accumulated with atomics
  const LmControl *ctrl;  // nullable: skip when !active or the linear system is still valid
  int F;
  int n_schur_blocks;
  int for_marginalized;
  int ublk_parity;  // which half of the double-buffered Schur rows the preceding sweep wrote
  // fused loop (nullable ctrl_out = plain reduction): the decision of the pending candidate is taken HERE, from the
  // energy the linearisation sweep at the candidate state produced (levenberg_marquardt_algorithm.hpp:93-123)
  LmControl *ctrl_out;
  WindowState *st;
  const double *scalars;  // multi-GPU: all-reduced {energy, n_valid, step^2, idepth.step}
  int n_sweep_blocks;
  int total_blocks;
  double *scalars_out;  // nullable: accumulate-only launches write {energy, n_valid, |step|^2, idepth.step} here
  int scalars_out_groups = 0;  // landmark shards: the sums are group 0 of kScalarGroups groups of four, the other groups are zeroed — the
                               // layout a shard on the two-stage path sends, so that every rank's collective has the same size and meaning
  // fused LM loop: ONE combined system instead of the four above (nullable = off).  Block-packed lower triangle of
  //   A = H_pp (1 + lambda on the diagonal) - H_schur / (1 + lambda)        (calculateStep, problem.hpp:347-351, without priors)
  // as combBlockCount(F) 8 x 8 blocks (bi >= bj) of 64 doubles, followed by b = b_pp - b_schur / (1 + lambda) (K doubles): the solve
  // kernel then loads 14 KB instead of two K x K matrices, and a sharded window all-reduces 14 KB instead of 51 KB
  double *comb;
  // comb_copies > 1 (single-GPU windows of the fused loop): workgroup b accumulates into copy b mod comb_copies — the first at `comb`,
  // the others comb_copy_stride doubles apart from comb + comb_copy_first — and the solve launch adds the copies while it loads them.
  // An f64 atomic add is carried out at the memory side, one after the other per address (~85 ns each: profiles/r05/atomic_fanin_probe.txt);
  // every entry of the 7-frame system receives one from each of the 35 landmark workgroups, and their queue — not the arithmetic — is the
  // last 2 - 3 us of the launch.
  int comb_copies = 1;
  int comb_copy_first = 0, comb_copy_stride = 0;
  double comb_lambda;  // lambda when neither control block is given (isolated timing launches)
  LmParams prm;
  long long *dbg;  // nullable tuning aid
  LmControl *ctrl_host = nullptr;  // nullable: pinned host copy of the published control block (closing round of a solve)
};
#define RS_STAMP(i) do { if (kStamps && a.dbg && threadIdx.x == 0 && blockIdx.x == 1) a.dbg[i] = wall_clock64(); } while (0)

using f64x4 = __attribute__((ext_vector_type(4))) double;

/**
 * Fused-loop prologue of the reduction kernel, executed identically by every workgroup: take the LM decision for the
 * pending candidate from the energy the linearisation sweep just produced at the candidate state, apply it to this
 * workgroup's landmark chunk (acceptStep / rejectStep, problem.hpp:366-402), workgroup 0 also moves the frame states and
 * publishes the outgoing control block.  Returns true when a linear system has to be built from the sweep's output.
 */
constexpr int kScalarGroups = 64;

struct ApplyRegs {
  uint8_t *p_status[2], *p_cand[2];
  uint8_t v_status[2], v_cand[2];
  double *p_idepth, *p_istep;
  double v_idepth, v_istep;
  double st_eps, st_step;
  const LmControl *out;  // LDS copy of the outgoing control block (workgroup 0 publishes it)
  int accept, pending, publish;
};

/** the stores of acceptStep / rejectStep decided by fusedDecideApply.  Issued at the END of the kernel: gfx9 counts
 *  stores in vmcnt, so any load wait that follows a store also waits for the store's round trip to HBM. */
__device__ inline void applyDecision(const ReduceSchurArgs &a, const ApplyRegs &ar) {
  const int tid = threadIdx.x;
  if (ar.pending) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (ar.p_status[h] == nullptr) continue;
      if (ar.accept)
        *ar.p_status[h] = ar.v_cand[h];
      else
        *ar.p_cand[h] = ar.v_status[h];
    }
    if (ar.p_idepth != nullptr) {
      if (ar.accept) *ar.p_idepth = ar.v_idepth + ar.v_istep;
      *ar.p_istep = 0;
    }
    if (blockIdx.x == 0 && tid < kBlk * a.F) {
      const int f = tid >> 3, c = tid & 7;
      if (ar.accept) a.st->eps[f][c] = ar.st_eps + ar.st_step;
      a.st->step[f][c] = 0;
    }
  }
  if (ar.publish && blockIdx.x == 0 && tid == 0) {
    *a.ctrl_out = *ar.out;
    if (a.ctrl_host) *a.ctrl_host = *ar.out;  // closing round: the solve's result goes straight into pinned host memory
  }
}

/** The Levenberg-Marquardt decision for the pending candidate (levenberg_marquardt_algorithm.hpp:93-123) from the four sums of the
 *  sweep that evaluated it: t = {energy, n_valid, |idepth step|^2, idepth . step} (already summed over all landmarks — and over all
 *  ranks of a sharded window).  Pure function of its arguments: every workgroup (and every rank) that evaluates it gets the same
 *  outgoing control block `c`, `accept` and `proceed` (a linear system has to be built / solved from the sweep's output). */
__device__ inline void lmDecision(const LmControl &cin, const double (&t)[4], const LmParams &prm, LmControl &c, int &accept, int &proceed) {
  c = cin;
  accept = 0;
  proceed = 0;
  const double eval_energy = t[0] + cin.cand_prior;
  const int n_valid = static_cast<int>(t[1] + 0.5);
  if (!cin.pending) {
    if (cin.relin) {
      // re-linearisation at the reverted state after a rejected step: nothing to decide
      c.relin = 0;
      proceed = 1;
    } else {
      // result = problem.calculateEnergy() before the loop (levenberg_marquardt_algorithm.hpp:82)
      c.idepth_sq = t[2];  // the opening sweep sums idepth^2 into the step-norm slot
      c.energy = eval_energy;
      c.n_valid = n_valid;
      c.active = (prm.max_iterations > 0 && n_valid > 0) ? 1 : 0;
      proceed = c.active;
    }
  } else {
    c.iteration = cin.iteration + 1;
    c.pending = 0;
    if (n_valid == 0) {
      c.active = 0;  // problem.rejectStep(); break;
      c.need_final_setup = 1;
    } else {
      if (fabs(cin.energy - eval_energy) / cin.energy < prm.function_tolerance) c.converged = 1;
      if (eval_energy < cin.energy || (prm.force_accept && cin.iteration < prm.min_iterations)) {
        accept = 1;
        const double state_sq = cin.frame_state_sq + cin.idepth_sq, step_sq = cin.frame_step_sq + t[2];
        if (step_sq < prm.parameter_tolerance * (state_sq + prm.parameter_tolerance)) c.converged = 1;
        c.energy = eval_energy;
        c.n_valid = n_valid;
        c.lambda = cin.lambda / prm.decrease_on_accept;
        c.idepth_sq = cin.idepth_sq + 2.0 * t[3] + t[2];
        c.need_final_setup = 0;
        proceed = 1;
      } else {
        c.need_final_setup = 1;
        if (prm.force_accept) {
          c.active = 0;  // problem.calculateEnergy(); return result;
        } else {
          c.lambda = cin.lambda * prm.increase_on_reject;
          c.relin = 1;  // the sweep's output belongs to the rejected state: re-linearise at the reverted one
        }
      }
      if (c.converged || c.iteration >= prm.max_iterations) c.active = 0;
    }
    if (!c.active) proceed = 0;
  }
}

__device__ inline bool fusedDecideApply(const ReduceSchurArgs &a, double *lds, long long *dbg_out, ApplyRegs &ar) {
  __shared__ LmControl s_out;
  __shared__ int s_accept, s_proceed;
  __shared__ long long s_dbg[2];
  const int tid = threadIdx.x;
  const LmControl cin = *a.ctrl;
  ar.out = &s_out;
  ar.pending = 0;
  ar.accept = 0;
  ar.publish = 0;
  // Everything the apply phase needs is fetched up front, so that the dependent chain table -> pointer -> value
  // overlaps the reduction instead of following the decision: 8 threads per landmark, thread `sub` owns the
  // targets t = sub and sub + 8.
  const bool is_schur = static_cast<int>(blockIdx.x) < a.n_schur_blocks;
  uint8_t *p_status[2] = {nullptr, nullptr}, *p_cand[2] = {nullptr, nullptr};
  uint8_t v_status[2] = {0, 0}, v_cand[2] = {0, 0};
  double *p_idepth = nullptr, *p_istep = nullptr;
  double v_idepth = 0, v_istep = 0;
  if (is_schur) {
    const SchurBlock &be = a.schur_table[blockIdx.x];
    const int l = tid >> 3, sub = tid & 7;
    const int i = be.offset + l;
    if (i < be.n) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int t = sub + 8 * h;
        if (t < a.F && be.status[t] != nullptr && i < be.n_res[t]) {
          p_status[h] = be.status[t] + i;
          p_cand[h] = be.cand[t] + i;
          v_status[h] = *p_status[h];
          v_cand[h] = *p_cand[h];
        }
      }
      if (sub == 0) {
        p_idepth = be.idepth + i;
        p_istep = be.idepth_step + i;
        v_idepth = *p_idepth;
        v_istep = *p_istep;
      }
    }
  }
  // frame states: read (and later written) by workgroup 0 only — the norms every workgroup needs for the decision come from the
  // control block (LmControl::frame_state_sq / frame_step_sq), a snapshot no workgroup of this launch modifies
  double st_eps = 0, st_step = 0;
  if (blockIdx.x == 0 && tid < kBlk * a.F) {
    const int f = tid >> 3, c = tid & 7;
    st_eps = a.st->eps[f][c];
    st_step = a.st->step[f][c];
  }
  // deterministic sums: energy, n_valid, |idepth step|^2, idepth . step (sweep partials)
  double v[4] = {0, 0, 0, 0};
  if (a.prm.use_reduced_scalars == 2) {
    if (tid < kScalarGroups) {
      const double *p = a.scalars + 4 * tid;
      v[0] = p[0];
      v[1] = p[1];
      v[2] = p[2];
      v[3] = p[3];
    }
  } else if (a.prm.use_reduced_scalars) {
    if (tid == 0) {
      v[0] = a.scalars[0];
      v[1] = a.scalars[1];
      v[2] = a.scalars[2];
      v[3] = a.scalars[3];
    }
  } else {
    for (int b = tid; b < a.n_sweep_blocks; b += kSchurThreads) {
      const double *p = a.partials + static_cast<size_t>(b) * kPartial;
      v[0] += p[44];
      v[1] += p[45];
      v[2] += p[46];
      v[3] += p[47];
    }
  }
  if (!cin.active) {  // (the loads above are speculative: they overlap the control block's round trip)
    if (blockIdx.x == 0 && tid == 0) {
      *a.ctrl_out = cin;
      if (a.ctrl_host) *a.ctrl_host = cin;
    }
    return false;
  }
  // landmark-sharded windows hand over the four sums ready-made (all-reduced): no reduction at all.  (Uniform branch: a launch
  // argument.)
  const bool direct = a.prm.use_reduced_scalars == 1;
  constexpr int RS = kSchurThreads + 2;
  static_assert(kSchurThreads == 512, "reduction tree below assumes 512 threads");
  double *l2 = lds + 6 * RS;  // [4][64]
  double *l3 = l2 + 6 * 64;   // [4][8]
  if (!direct) {
#pragma unroll
    for (int e = 0; e < 4; ++e) lds[e * RS + tid] = v[e];
    ldsBarrier();
    if (kStamps && a.dbg && tid == 0) s_dbg[0] = wall_clock64();
    // three-level tree (8-way per level), fixed order => deterministic
    if (tid < 4 * 64) {
      const int e = tid >> 6, j = tid & 63;
      double s = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += lds[e * RS + 8 * j + k];
      l2[e * 64 + j] = s;
    }
    ldsBarrier();
    if (tid < 4 * 8) {
      const int e = tid >> 3, j = tid & 7;
      double s = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) s += l2[e * 64 + 8 * j + k];
      l3[e * 8 + j] = s;
    }
    ldsBarrier();
  }
  if (tid == 0) {
    double t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      double s = 0;
      if (direct) {
        s = v[e];
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) s += l3[e * 8 + k];
      }
      t[e] = s;
    }
    LmControl c;
    int accept = 0, proceed = 0;
    lmDecision(cin, t, a.prm, c, accept, proceed);
    s_accept = accept;
    s_proceed = proceed;
    s_out = c;
  }
  ldsBarrier();
  if (kStamps && a.dbg && tid == 0) s_dbg[1] = wall_clock64();
  if (kStamps && a.dbg && tid == 0) { dbg_out[0] = s_dbg[0]; dbg_out[1] = s_dbg[1]; }
  ar.accept = s_accept;
  ar.pending = cin.pending;
  ar.publish = 1;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    ar.p_status[h] = p_status[h];
    ar.p_cand[h] = p_cand[h];
    ar.v_status[h] = v_status[h];
    ar.v_cand[h] = v_cand[h];
  }
  ar.p_idepth = p_idepth;
  ar.p_istep = p_istep;
  ar.v_idepth = v_idepth;
  ar.v_istep = v_istep;
  ar.st_eps = st_eps;
  ar.st_step = st_step;
  return s_proceed != 0;
}

/**
 * grid = n_schur_blocks + F*F workgroups of 256 threads.
 *  - workgroups [0, n_schur_blocks): evaluateLinearSystemPoseDepthSchurComplement (hessian_block_evaluation.hpp:169-236) for a chunk
 *    of 64 landmarks of one frame: phase 1 finalises the landmarks (reference-frame block sum_t T^T u, H_dd, b_d, inverse,
 *    caches), phase 2 accumulates H_schur += A^T W A with the f64 matrix cores (v_mfma_f64_16x16x4_f64): a 16x16 output
 *    tile takes 4 landmarks per instruction, rows staged in LDS with a conflict-free stride; upper-triangular tiles only.
 *  - workgroups [n_schur_blocks, +F*F): evaluateLinearSystemPosePose (hessian_block_evaluation.hpp:96-164) for one frame pair:
 *    deterministic sum of the sweep's per-block partials (G, q), the derived blocks T^T G T, G T, T^T q, and their
 *    scatter into H_pp / b_pp (H_rr += T^T G T, H_tt += G, H_rt = -(G T)^T, H_tr = -G T, b_r += T^T q, b_t -= q).
 */
/** (the leading arguments repeat members of `a`: the dispatcher preloads the first 16 argument words into scalar registers —
 *  build.sh: -amdgpu-kernarg-preload-count — but not the members of a by-value struct.  With them the control block and the descriptor
 *  are requested in the wave's first cycles, beside the rest of the argument block instead of behind it.) */
__global__ void __launch_bounds__(kSchurThreads) reduceSchurKernel(const LmControl *ctrl_p, const SchurBlock *table_p, const PairConst *pc_p,
                                                                   const double *partials_p, const int *pair_first_p, const int *pair_num_p,
                                                                   int n_schur_blocks_p, int F_p, ReduceSchurArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  a.ctrl = ctrl_p;
  a.schur_table = table_p;
  a.pc = pc_p;
  a.partials = partials_p;
  a.pair_first_block = pair_first_p;
  a.pair_num_blocks = pair_num_p;
  a.n_schur_blocks = n_schur_blocks_p;
  a.F = F_p;
  const long long rs_t0 = (kStamps && a.dbg) ? wall_clock64() : 0;
  long long rs_dbg[2] = {0, 0};
  // (inside the fused loop the LM decision for the pending candidate is the prologue of the SOLVE launch behind this one — it needs
  // the sums of all of the sweep's energy scalars, which the extra workgroup below produces while the others build the system)
  // every control-block word this launch needs is requested BEFORE the first one is tested: three dependent scalar round trips
  // (active -> linear_system_valid -> lambda) become one.  Damping of the system being built: the incoming control block (the PBA's
  // LM keeps lambda constant, eigen_photometric_bundle_adjustment.cpp:74-75), or the launch argument
  // Both are read through pointers that are valid whatever the launch passes (the argument block itself stands in for an absent control
  // block / an empty table), so that there is no branch in front of the loads: control block, descriptor head and the remaining
  // argument words leave together and are waited for once.
  // (the stand-in — the argument segment, of which ReduceSchurArgs is a part — has to cover every word read through it)
  static_assert(sizeof(ReduceSchurArgs) >= sizeof(LmControl), "the argument block stands in for an absent control block");
  static_assert(sizeof(ReduceSchurArgs) >= offsetof(SchurBlock, flags) + sizeof(SchurBlock::flags) && sizeof(ReduceSchurArgs) >= offsetof(SchurBlock, ublk) + sizeof(SchurBlock::ublk),
                "the argument block stands in for the head of an absent Schur descriptor");
  const void DSOPP_CONSTANT *any_words = (const void DSOPP_CONSTANT *)__builtin_amdgcn_kernarg_segment_ptr();
  const LmControl DSOPP_CONSTANT *cp = ctrl_p ? (const LmControl DSOPP_CONSTANT *)ctrl_p : (const LmControl DSOPP_CONSTANT *)any_words;
  int c_active = cp->active, c_lsv = cp->linear_system_valid;
  double comb_lam = cp->lambda;
  // ... and with them the head of this workgroup's Schur descriptor (the pair / scalar workgroups behind the Schur blocks fetch the
  // last one and ignore it): the descriptor's address depends on the launch arguments only, so it need not wait for the test
  struct SchurHead {
    int r, offset, n, cap;
  } hd = {0, 0, 0, 0};
  unsigned hd_conn = 0;
  decltype(SchurBlock::ublk) hd_ublk = nullptr;
  decltype(SchurBlock::flags) hd_flags = nullptr;
  {
    const SchurBlock DSOPP_CONSTANT *dp =
        n_schur_blocks_p > 0 ? (const SchurBlock DSOPP_CONSTANT *)table_p + min(static_cast<int>(blockIdx.x), n_schur_blocks_p - 1)
                             : (const SchurBlock DSOPP_CONSTANT *)any_words;
    const SchurBlock DSOPP_CONSTANT &d = *dp;
    hd = {d.r, d.offset, d.n, d.cap};
    hd_conn = d.conn_mask;
    hd_ublk = d.ublk;
    hd_flags = d.flags;
  }
  // the other argument words the kernel's head reads: one burst, in flight with the control block and the descriptor
  asm volatile("" ::"s"(a.frames), "s"(a.Hpp), "s"(a.bpp), "s"(a.Hsc), "s"(a.bsc), "s"(a.for_marginalized), "s"(a.ublk_parity), "s"(a.ctrl_out),
               "s"(a.st), "s"(a.scalars), "s"(a.n_sweep_blocks), "s"(a.scalars_out), "s"(a.scalars_out_groups), "s"(a.comb), "s"(a.comb_lambda), "s"(a.comb_copies),
               "s"(a.comb_copy_first), "s"(a.comb_copy_stride));
  asm volatile("" : "+s"(hd.r), "+s"(hd.offset), "+s"(hd.n), "+s"(hd.cap), "+s"(hd_conn), "+s"(hd_ublk), "+s"(hd_flags), "+s"(c_active), "+s"(c_lsv),
               "+s"(comb_lam));
  if (!ctrl_p) {
    c_active = 1;
    c_lsv = 0;
    comb_lam = a.comb_lambda;
  }
  if (n_schur_blocks_p <= 0) {
    hd = {0, 0, 0, 0};
    hd_conn = 0;
    hd_ublk = nullptr;
    hd_flags = nullptr;
  }
  if (!c_active || c_lsv) return;
  const int F = a.F, K = kBlk * F;
  // this workgroup's copy of the combined system (ReduceSchurArgs::comb_copies)
  double *const comb_w = (a.comb && a.comb_copies > 1 && (blockIdx.x % a.comb_copies) != 0)
                             ? a.comb + a.comb_copy_first + static_cast<size_t>(blockIdx.x % a.comb_copies - 1) * a.comb_copy_stride
                             : a.comb;
  if (a.scalars_out && static_cast<int>(blockIdx.x) == a.n_schur_blocks + F * F) {
    // ---- landmark-sharded windows: one extra workgroup sums the sweep's 4 energy scalars (fixed order) into the tail of
    // the reduction buffer, so that they travel in the same collective as the systems (no separate kernel for it)
    double *lds = reinterpret_cast<double *>(smem_raw);
    double v[4] = {0, 0, 0, 0};
    for (int b = threadIdx.x; b < a.n_sweep_blocks; b += kSchurThreads) {
      const double *p = a.partials + static_cast<size_t>(b) * kPartial + 44;
      v[0] += p[0];
      v[1] += p[1];
      v[2] += p[2];
      v[3] += p[3];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) lds[e * (kSchurThreads + 2) + threadIdx.x] = v[e];
    __syncthreads();
    if (threadIdx.x < 4) {
      double sacc = 0;
      for (int j = 0; j < kSchurThreads; ++j) sacc += lds[threadIdx.x * (kSchurThreads + 2) + j];
      a.scalars_out[threadIdx.x] = sacc;
    } else if (static_cast<int>(threadIdx.x) < 4 * a.scalars_out_groups) {
      a.scalars_out[threadIdx.x] = 0;
    }
    return;
  }
  if (static_cast<int>(blockIdx.x) >= a.n_schur_blocks) {
    // ---- pair block
    const int p = blockIdx.x - a.n_schur_blocks;
    const int r = p / F, t = p % F, pi = r * kMaxFrames + t;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const PairConst &P = a.pc[pi];
    const int valid = P.valid;
    double *lds = reinterpret_cast<double *>(smem_raw);  // [48] G/q + [64] scratch + [kPairBlk] derived + [8][48] wave sums
    double *wsum = lds + 48 + 64 + kPairBlk;
    // partial sums of the sweep's workgroups: wave w takes the blocks w, w + 8, ... (independent loads), the 8 wave sums
    // are combined in fixed order => deterministic
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
    ldsBarrier();
    if (threadIdx.x >= 64 || !valid) return;
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
    const int i = lane >> 3, j = lane & 7;
    if (a.comb) {
      // combined system: diagonal entries carry the (1 + lambda) of calculateStep; of the two off-diagonal blocks
      // H_rt = -(G T)^T and H_tr = -G T only the one below the diagonal is kept
      const double dl = i == j ? 1.0 + comb_lam : 1.0;
      atomicAdd(&comb_w[combBlockIndex(r, r) * 64 + lane], dl * drv[lane]);
      atomicAdd(&comb_w[combBlockIndex(t, t) * 64 + lane], dl * lds[symIdx(i, j)]);
      if (r > t)
        atomicAdd(&comb_w[combBlockIndex(r, t) * 64 + lane], -drv[64 + 8 * j + i]);
      else
        atomicAdd(&comb_w[combBlockIndex(t, r) * 64 + lane], -drv[64 + lane]);
      if (lane < 8) {
        double *rhs = comb_w + combBlockCount(F) * 64;
        atomicAdd(&rhs[kBlk * r + lane], drv[128 + lane]);
        atomicAdd(&rhs[kBlk * t + lane], -lds[36 + lane]);
      }
      return;
    }
    atomicAdd(&a.Hpp[(kBlk * r + i) * K + kBlk * r + j], drv[lane]);
    atomicAdd(&a.Hpp[(kBlk * t + i) * K + kBlk * t + j], lds[symIdx(i, j)]);
    atomicAdd(&a.Hpp[(kBlk * r + i) * K + kBlk * t + j], -drv[64 + 8 * j + i]);
    atomicAdd(&a.Hpp[(kBlk * t + i) * K + kBlk * r + j], -drv[64 + lane]);
    if (lane < 8) {
      atomicAdd(&a.bpp[kBlk * r + lane], drv[128 + lane]);
      atomicAdd(&a.bpp[kBlk * t + lane], -lds[36 + lane]);
    }
    return;
  }
  // ---- Schur block
  const double comb_sc = -1.0 / (1.0 + comb_lam);
  const int Kp = (K + 15) & ~15;
  const int stride = schurRowStride(Kp);
  double *hrow = reinterpret_cast<double *>(smem_raw);  // [kSchurLandmarks][stride]
  double *wgt = hrow + kSchurLandmarks * stride;        // inv per landmark (0 = excluded)
  double *wbd = wgt + kSchurLandmarks;                  // inv * bd
  const SchurBlock &be = a.schur_table[blockIdx.x];  // (head: hd, fetched above)
  const int r = hd.r;
  const bool bd_in_pad = K < Kp;  // H_schur^T W b_d as one more column of the SYRK when the tiles have a spare column
  if (kStamps && a.dbg && threadIdx.x == 0 && blockIdx.x == 1) { a.dbg[0] = rs_t0; a.dbg[6] = rs_dbg[0]; a.dbg[7] = rs_dbg[1]; }
  RS_STAMP(1);
  // phase 1: 8 threads per landmark, thread `sub` owns the targets t = sub, sub + 8, ...; the loads of its first target,
  // of the landmark flags and of this thread's share of the per-target constants T = blockdiag(Adj, 1, s0) are issued
  // before the LDS rows are cleared so that their latency overlaps.
  const int l = threadIdx.x >> 3, sub = threadIdx.x & 7;
  const int i = hd.offset + l;
  const unsigned conn = hd_conn & ~(1u << r);
  const size_t plane = ublkPlane(hd.cap);
  const double *ubase = hd_ublk + static_cast<size_t>(a.ublk_parity) * kMaxFrames * plane + static_cast<size_t>(i) * kUblk;
  bool take = false;
  uint8_t flg = 0;
  if (i < hd.n) {
    flg = hd_flags[i];
    take = a.for_marginalized ? (flg & kFlagToMarginalize) != 0 : (flg & kFlagMarginalized) == 0;
  }
  double ht[kUblk];
#pragma unroll
  for (int c = 0; c < kUblk; ++c) ht[c] = 0;
  const bool first = i < hd.n && sub < F && ((conn >> sub) & 1u);
  if (first) {
    const double *src = ubase + sub * plane;
#pragma unroll
    for (int c = 0; c < kUblk; ++c) ht[c] = src[c];
  }
  double tmv = 0;
  if (static_cast<int>(threadIdx.x) < F * 37) {
    const int t = threadIdx.x / 37, c = threadIdx.x - 37 * t;
    const PairConst &P = a.pc[r * kMaxFrames + t];
    tmv = c < 36 ? P.Adj[c] : P.s0;
  }
  // zero the tile rows (pad columns must be 0)
  for (int idx = threadIdx.x; idx < kSchurLandmarks * stride; idx += kSchurThreads) hrow[idx] = 0;
  double *Tm = wbd + kSchurLandmarks;  // [F][40]: Adj (36), s0
  if (static_cast<int>(threadIdx.x) < F * 37) {
    const int t = threadIdx.x / 37, c = threadIdx.x - 37 * t;
    Tm[t * 40 + c] = tmv;
  }
  for (int e = threadIdx.x + kSchurThreads; e < F * 37; e += kSchurThreads) {
    const int t = e / 37, c = e - 37 * t;
    const PairConst &P = a.pc[r * kMaxFrames + t];
    Tm[t * 40 + c] = c < 36 ? P.Adj[c] : P.s0;
  }
  ldsBarrier();
  RS_STAMP(2);
  {
    double hr[kBlk];
#pragma unroll
    for (int c = 0; c < kBlk; ++c) hr[c] = 0;
    double hdd = 0, bd = 0;
    double *row = hrow + l * stride;
    if (take) {
      for (int t = sub; t < F; t += 8) {
        if (!((conn >> t) & 1u)) continue;
        if (t != sub) {
          const double *src = ubase + t * plane;
#pragma unroll
          for (int c = 0; c < kUblk; ++c) ht[c] = src[c];
        }
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
        }
        hr[6] -= ht[6];
        hr[7] -= Tt[36] * ht[7];
      }
    }
    // combine the 8 sub-threads of a landmark (adjacent lanes)
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
          if (a.for_marginalized && be.fixed) hdd += 1e8;  // kScaleNullspaceRegularizer
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
      if (bd_in_pad && take) row[K] = bd;
    }
  }
  ldsBarrier();
  RS_STAMP(3);
  // phase 2: H_schur tiles with v_mfma_f64_16x16x4_f64.  A[i][k] = inv_k * h_k[16*ti + i], B[k][j] = h_k[16*tj + j];
  // lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]  (cdna_hip_programming.md §3)
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nt = Kp >> 4;
    const int n_tiles = nt * (nt + 1) / 2;
    const int li = lane & 15, lk = lane >> 4;
    for (int tile = wave; tile < n_tiles; tile += kSchurThreads / 64) {
      // decode (ti <= tj) from the linear upper-triangular index
      int ti = 0, rem = tile;
      while (rem >= nt - ti) {
        rem -= nt - ti;
        ++ti;
      }
      const int tj = ti + rem;
      f64x4 acc = {0, 0, 0, 0};
      const double *pa = hrow + lk * stride + 16 * ti + li;
      const double *pb = hrow + lk * stride + 16 * tj + li;
      // all 48 operand words of the tile are requested before the first matrix instruction: one LDS latency per tile instead
      // of one per instruction (the loop was LDS-latency-bound: 3.3 us for two tiles per wave)
      constexpr int kSteps = kSchurLandmarks / 4;
      double av[kSteps], bv[kSteps], wv[kSteps];
#pragma unroll
      for (int q = 0; q < kSteps; ++q) {
        wv[q] = wgt[4 * q + lk];
        av[q] = pa[4 * q * stride];
        bv[q] = pb[4 * q * stride];
      }
#pragma unroll
      for (int q = 0; q < kSteps; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wv[q] * av[q], bv[q], acc, 0, 0, 0);
      // C/D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int row = 16 * ti + lk + 4 * reg, col = 16 * tj + li;
        const double v = acc[reg];
        if (a.comb) {
          // symmetric: the upper-triangular tile entry (row, col) lands at (col, row) of the lower-packed combined system
          const double sv = v * comb_sc;
          if (row < K && col < K && col >= row && v != 0) atomicAdd(&comb_w[combIndex(col, row)], sv);
          if (bd_in_pad && row < K && col == K && v != 0) atomicAdd(&comb_w[combBlockCount(F) * 64 + row], sv);
          continue;
        }
        if (row < K && col < K && col >= row && v != 0) atomicAdd(&a.Hsc[row * K + col], v);
        if (bd_in_pad && row < K && col == K && v != 0) atomicAdd(&a.bsc[row], v);  // b_schur rides in the pad column
      }
    }
    RS_STAMP(4);
    for (int c = threadIdx.x; c < (bd_in_pad ? 0 : K); c += kSchurThreads) {
      double s = 0;
#pragma unroll 1
      for (int ll = 0; ll < kSchurLandmarks; ++ll) s += wbd[ll] * hrow[ll * stride + c];
      if (s != 0) {
        if (a.comb)
          atomicAdd(&comb_w[combBlockCount(F) * 64 + c], s * comb_sc);
        else
          atomicAdd(&a.bsc[c], s);
      }
    }
    RS_STAMP(5);
  }
}

/** The LM decision for the pending candidate + its accept / reject as a kernel of its own: the closing round of a solve (nothing
 *  is built or solved behind it).  grid = schur blocks (>= 1). */
__global__ void __launch_bounds__(kSchurThreads) decideApplyKernel(ReduceSchurArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  long long dbg_unused[2] = {0, 0};
  ApplyRegs ar;
  ar.pending = 0;
  ar.publish = 0;
  fusedDecideApply(a, reinterpret_cast<double *>(smem_raw), dbg_unused, ar);
  applyDecision(a, ar);
}

/** zeroes the Schur accumulation target unless the device-driven loop skips this linearisation */
__global__ void clearSchurKernel(double *buf, int n, const LmControl *ctrl) {
  if (ctrl && (!ctrl->active || ctrl->linear_system_valid)) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = 0;
}

// ---------------------------------------------------------------------------------------------------------------
// K3: assemble + solve (single workgroup)
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSolveThreads = 256;

struct SolveArgs {
  const FrameDev *frames;
  WindowState *st;
  PairConst *pc;
  const double *Hpp_raw, *bpp_raw;  // pose-pose sums without priors (K x K full, K)
  double *Hsc;                      // Schur system, upper triangle accumulated (symmetrised in place when stored)
  const double *bsc;
  double *Hpp_out, *bpp_out;        // optional store: system_pose with priors
  double *Hsc_copy = nullptr;       // optional (with store_system): full symmetric copy of H_schur for the covariance read-back
  const double *Hm, *bm;            // marginal prior
  double *step;                     // out: K
  LmControl *ctrl;                  // nullable (host-driven stages pass lambda explicitly)
  double lambda;
  double affine_reg[2];
  double fixed_reg;
  double energy_marginalized;
  int F;
  int fej;
  int do_solve;
  int store_system;  // write H_pp / b_pp (with priors) and the symmetrised H_schur back (stage API, marginalisation, covariance)
  int add_priors;
  int use_marginal;  // the marginal prior is non-zero
  long long *dbg_stamps;  // nullable: wall_clock64() stamps of the phases (tuning aid)
};
#define DSOPP_STAMP(i) do { if (kStamps && a.dbg_stamps && tid == 0) a.dbg_stamps[i] = wall_clock64(); } while (0)

/** prior + marginal energy terms of calculateEnergy (problem.hpp:293-312) for state x = eps (+ step); whole workgroup */
__device__ inline double priorEnergyBlock(const SolveArgs &a, bool with_step, double *lds /* K + 8 */, int tid) {
  const int F = a.F, K = kBlk * F;
  for (int c = tid; c < K; c += kSolveThreads) lds[c] = a.st->eps[c >> 3][c & 7] + (with_step ? a.st->step[c >> 3][c & 7] : 0.0);
  __syncthreads();
  double part = 0;
  for (int c = tid; c < K; c += kSolveThreads) {
    if (a.use_marginal) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += a.Hm[c * K + k] * lds[k];
      part += a.bm[c] * lds[c] + 0.5 * lds[c] * s;
    }
    if ((c & 7) >= 6) {
      const double ab = a.st->ab0[c >> 3][(c & 7) - 6] + lds[c];
      part += 0.5 * ab * a.affine_reg[(c & 7) - 6] * ab;
    }
  }
  part = waveSum(part);
  __syncthreads();
  if ((tid & 63) == 0) lds[K + (tid >> 6)] = part;
  __syncthreads();
  double total = a.energy_marginalized;
  for (int w = 0; w < kSolveThreads / 64; ++w) total += lds[K + w];
  __syncthreads();
  return total;
}

/** packed lower-triangular index of an 8x8 block */
__host__ __device__ constexpr int lowIdx(int i, int j) { return i * (i + 1) / 2 + j; }

/**
 * evaluateLinearSystemPrior (problem.hpp:37-77), calculateStep (problem.hpp:342-361) and NormalLinearSystem::solve
 * (normal_linear_system.cpp:10-16,52-59: Jacobi preconditioner + LDL^T; here a blocked Cholesky over the 8x8 frame blocks
 * with the right-hand side carried as an extra row, each thread factoring/inverting the current diagonal block in
 * registers so a block step costs two barriers) — one workgroup.  Also rebuilds the pair constants and the prior energy
 * for the candidate state eps + step, so the energy sweep can follow immediately.
 */
__global__ void __launch_bounds__(kSolveThreads, 1) assembleSolveKernel(SolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int F = a.F, K = kBlk * F;
  const int N = K + 1;   // augmented with the right-hand side row
  const int ld = N + 1;
  double *A = reinterpret_cast<double *>(smem_raw);  // N x ld (lower triangle used)
  double *pv = A + N * ld;                           // K preconditioner
  double *xs = pv + K;                               // K + 16 scratch
  double *Linv = xs + K + 16;                        // F x 36 inverses of the diagonal blocks
  double *epsl = Linv + 36 * kMaxFrames;             // K: state increment eps of every frame
  double *stpl = epsl + K;                           // K: the new step (-x), kept in LDS for the pair refresh / prior energy
  double *ab0l = stpl + K;                           // 2 F: affine brightness at the linearisation point
  const int tid = threadIdx.x;
  // Control block, pair-refresh inputs and the first tile batch are all requested before anything waits: the kernel
  // start costs one memory round trip.  (The early exit is taken after the first barrier, which keeps the loads above it.)
  int c_active = 1, c_relin = 0;
  double lam = a.lambda;
  if (a.ctrl) {
    c_active = a.ctrl->active;
    c_relin = a.ctrl->relin;
    lam = a.ctrl->lambda;
  }
  DSOPP_STAMP(0);
  // inputs of refreshPairCurrent for pair (r, t) = (tid / F, tid % F): constant over the solve.  Raw loads only: any
  // arithmetic on a loaded value here would make the compiler wait for the round trip before issuing the next loads.
  struct {
    int valid;
    Rigid T0;
    double fxr, fyr, cxr, cyr, fxt, fyt, cxt, cyt, exposure_r, exposure_t;
  } pp;
  pp.valid = 0;
  const bool fast_refresh = a.fej && a.do_solve && a.ctrl;
  if (fast_refresh && tid < F * F) {
    const int r = tid / F, t = tid - F * (tid / F);
    const PairConst &P = a.pc[r * kMaxFrames + t];
    pp.valid = P.valid;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) pp.T0.R[3 * i + j] = P.T0rel[4 * i + j];
      pp.T0.t[i] = P.T0rel[4 * i + 3];
    }
    const FrameDev &fr = a.frames[r];
    const FrameDev &ft = a.frames[t];
    pp.fxr = fr.fx;
    pp.fyr = fr.fy;
    pp.cxr = fr.cx;
    pp.cyr = fr.cy;
    pp.fxt = ft.fx;
    pp.fyt = ft.fy;
    pp.cxt = ft.cx;
    pp.cyt = ft.cy;
    pp.exposure_r = fr.exposure;
    pp.exposure_t = ft.exposure;
  }

  // ---- system_pose = sums + priors (problem.hpp:39-62)
  double eps_c = 0, ab0_c = 0, bpp_c = 0, bsc_c = 0, bm_c = 0;  // this thread's entry c = tid (K <= 128 < kSolveThreads)
  int fixed_c = 0, tomarg_c = 0;
  if (tid < K) {
    const int f = tid >> 3, i = tid & 7;
    eps_c = a.st->eps[f][i];
    fixed_c = a.frames[f].fixed;
    tomarg_c = a.frames[f].to_marginalize;
    ab0_c = a.st->ab0[f][i < 6 ? 0 : i - 6];
    bpp_c = a.bpp_raw[tid];
    bsc_c = a.bsc[tid];
    if (a.use_marginal) bm_c = a.bm[tid];
  }
  double *prior_diag = Linv;  // the Linv area is not in use yet
  // 16 x 16 thread tiles over the lower triangle (diagonal tiles included); the loads of a whole batch of tiles are issued
  // back to back so the phase costs ~one memory round trip per batch instead of one per tile
  const int tile_r = tid >> 4, tile_c = tid & 15;
  const int nt = (K + 15) >> 4, n_tiles = nt * (nt + 1) / 2;
  constexpr int kTileBatch = 12;
  double hp[kTileBatch], hs[kTileBatch], hm[kTileBatch];
  int rows[kTileBatch], cols[kTileBatch];
  auto loadBatch = [&](int base) {
#pragma unroll
    for (int u = 0; u < kTileBatch; ++u) {
      int tile = base + u, tr0 = 0;
      while (tile >= tr0 + 1) {  // tile row tr0 holds tr0 + 1 tiles
        tile -= tr0 + 1;
        ++tr0;
      }
      const int row = 16 * tr0 + tile_r, col = 16 * tile + tile_c;
      const bool in = (base + u) < n_tiles && row < K && col < K;
      rows[u] = in ? row : -1;
      cols[u] = col;
      // unconditional loads from clamped addresses (a select around a load makes hipcc branch and drain vmcnt per element)
      const int rc = min(row, K - 1), cc = min(col, K - 1);
      hp[u] = a.Hpp_raw[rc * K + cc];
      hs[u] = a.Hsc[min(rc, cc) * K + max(rc, cc)];
      hm[u] = 0;
    }
    if (a.use_marginal) {
#pragma unroll
      for (int u = 0; u < kTileBatch; ++u) {
        const int rc = min(rows[u] < 0 ? 0 : rows[u], K - 1), cc = min(cols[u], K - 1);
        hm[u] = a.Hm[rc * K + cc];
      }
    }
  };
  loadBatch(0);
  // opaque to the optimiser: stops it from testing these loaded flags (and waiting for them) above the tile loads
  asm volatile("" : "+v"(fixed_c), "+v"(tomarg_c), "+v"(pp.valid), "+v"(c_active), "+v"(c_relin));
  int prior_kind = 0;  // 0 none, 1 fixed frame, 2 affine brightness
  if (tid < K) {
    if (a.add_priors && !tomarg_c) prior_kind = fixed_c ? 1 : ((tid & 7) >= 6 ? 2 : 0);
    const double pd_c = prior_kind == 1 ? a.fixed_reg : (prior_kind == 2 ? a.affine_reg[(tid & 7) - 6] : 0.0);
    xs[tid] = eps_c;
    epsl[tid] = eps_c;
    prior_diag[tid] = pd_c;
    if ((tid & 7) >= 6) ab0l[2 * (tid >> 3) + (tid & 7) - 6] = ab0_c;
  }
  __syncthreads();  // fence: keeps every load above; prior_diag visible
  if (!c_active || c_relin) return;
  const double sc = -1.0 / (1.0 + lam);
  auto storeBatch = [&]() {
#pragma unroll
    for (int u = 0; u < kTileBatch; ++u) {
      const int row = rows[u], col = cols[u];
      if (row < 0) continue;
      double v = hp[u];
      if (row == col) v += prior_diag[row];
      if (a.store_system) {
        a.Hpp_out[row * K + col] = v;
        a.Hpp_out[col * K + row] = a.Hpp_raw[col * K + row] + (row == col ? prior_diag[row] : 0.0);
        if (col < row) a.Hsc[row * K + col] = hs[u];
        if (a.Hsc_copy && col <= row) {  // the symmetrised Schur system once more, right behind H_pp: one transfer for both
          a.Hsc_copy[row * K + col] = hs[u];
          a.Hsc_copy[col * K + row] = hs[u];
        }
      }
      if (col <= row) {
        // calculateStep — problem.hpp:347-351: H = H_pp + lam*diag(H_pp) + H_m - H_sc/(1+lam)
        if (row == col) v += v * lam;
        A[row * ld + col] = v + sc * hs[u] + hm[u];
      }
    }
  };
  storeBatch();
  for (int base = kTileBatch; base < n_tiles; base += kTileBatch) {
    loadBatch(base);
    storeBatch();
  }
  __syncthreads();
  if (tid < K) {
    const int c = tid;
    double v = bpp_c;
    if (prior_kind == 1)
      v += a.fixed_reg * eps_c;
    else if (prior_kind == 2)
      v += a.affine_reg[(c & 7) - 6] * (ab0_c + eps_c);
    if (a.store_system) a.bpp_out[c] = v;
    v += sc * bsc_c;
    if (a.use_marginal) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += a.Hm[c * K + k] * xs[k];
      v += bm_c + s;
    }
    A[K * ld + c] = v;
    // The reference solves the Jacobi-scaled system p H p, p = 1/sqrt(diag + 10) (normal_linear_system.cpp:10-16,52-59).  A
    // Cholesky factorisation is invariant under symmetric diagonal scaling (chol(S A S) = S chol(A), and every floating-point
    // operation keeps its relative error), so the scaling pass is skipped; only the zero-pivot guard refers to the scaled
    // pivot d / (diag + 10), kept here.
    pv[c] = A[c * ld + c] + 10.0;
  }
  if (tid == 0) A[K * ld + K] = 0;
  if (!a.do_solve) return;
  __syncthreads();
  DSOPP_STAMP(1);
  // ---- blocked Cholesky A = L L^T on the augmented (K+1) x (K+1) matrix: the last row of L becomes y^T = (L^-1 b)^T.
  // Look-ahead schedule, one barrier per 8x8 frame block: wave 0 ("panel wave") brings block column kb+1 up to date with
  // panel kb, factors its diagonal block in registers and solves the panel below it, WHILE waves 1..3 apply panel kb to
  // the rest of the trailing matrix (columns >= kb+2).  The sequential factor chain is thus off the other waves' path.
  DSOPP_STAMP(2);
  const int wave = tid >> 6, lane = tid & 63;
  auto readLane = [](double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
  };
  auto factorAndPanel = [&](int kb) {
    // wave 0 only.  Lane i owns row k0 + i of block column kb (the 8 rows of the diagonal block AND the panel rows below
    // it): one elimination loop does the Cholesky of the diagonal block and the triangular solve of the panel together.
    // Per pivot k: d = C[k][k] (v_readlane from lane k), l_ik = c_ik / sqrt(d) in every lane, then for the remaining columns
    // j the row-k factor l_jk is broadcast by v_readlane and every lane updates its own c_ij.  A wave issues one
    // instruction per ~4.7 cycles whether or not it depends on the previous one (measured), so what matters is the
    // instruction count: ~200 here against ~430 for a per-lane redundant 8x8 factorisation + per-row substitution.
    const int k0 = kb * kBlk;
    const int row = k0 + lane;
    const bool valid = row < N;
    double c[kBlk], invd[kBlk], lj[28];  // lj: strictly-lower factor entries l_jk of the diagonal block (uniform), for rows beyond 64
    {
      const double *src = A + (valid ? row : k0) * ld + k0;
#pragma unroll
      for (int j = 0; j < kBlk; ++j) c[j] = src[j];
    }
    double guard[kBlk];  // zero-pivot thresholds, fetched before the pivot chain starts
#pragma unroll
    for (int k = 0; k < kBlk; ++k) guard[k] = 1e-30 * pv[min(k0 + k, K - 1)];
    int e = 0;
#pragma unroll
    for (int k = 0; k < kBlk; ++k) {
      const double d = readLane(c[k], k);
      // inv = 1/sqrt(d) from the f32 estimate + two Newton steps in f64; pivots whose Jacobi-scaled value d / (diag + 10) is
      // below 1e-30 are treated as zero, as a rank-revealing factorisation would
      const bool okp = d > guard[k];
      double inv = static_cast<double>(__frsqrt_rn(static_cast<float>(okp ? d : 1.0)));
      inv = inv * (1.5 - 0.5 * d * inv * inv);
      inv = inv * (1.5 - 0.5 * d * inv * inv);
      inv = okp ? inv : 0.0;
      invd[k] = inv;
      const double l = c[k] * inv;  // lane k: sqrt(d); lanes i > k: l_ik
      c[k] = l;
#pragma unroll
      for (int j = k + 1; j < kBlk; ++j) {
        const double ljk = readLane(l, j);
        lj[e++] = ljk;
        c[j] -= l * ljk;
      }
    }
    if (valid) {
      double *dst = A + row * ld + k0;
#pragma unroll
      for (int j = 0; j < kBlk; ++j)
        if (lane >= kBlk || j <= lane) dst[j] = c[j];  // the diagonal block keeps its lower triangle only
    }
    // rows beyond the first 64 of this block column (windows of more than 7 frames): substitution with the broadcast factors
    for (int r2 = row + 64; r2 < N; r2 += 64) {
      double v[kBlk];
#pragma unroll
      for (int j = 0; j < kBlk; ++j) v[j] = A[r2 * ld + k0 + j];
      int e2 = 0;
#pragma unroll
      for (int k = 0; k < kBlk; ++k) {
        v[k] *= invd[k];
#pragma unroll
        for (int j = k + 1; j < kBlk; ++j) v[j] -= v[k] * lj[e2++];
      }
#pragma unroll
      for (int j = 0; j < kBlk; ++j) A[r2 * ld + k0 + j] = v[j];
    }
    if (lane == 0) {
#pragma unroll
      for (int cidx = 0; cidx < kBlk; ++cidx) Linv[kb * 36 + lowIdx(cidx, cidx)] = invd[cidx];  // diagonal of the inverse; completed below
    }
  };
  if (wave == 0) factorAndPanel(0);
  __syncthreads();
  DSOPP_STAMP(16);
  for (int kb = 0; kb < F; ++kb) {
    const int k0 = kb * kBlk, k1 = k0 + kBlk, k2 = k1 + kBlk;
    if (kb + 1 < F) {
      // all waves: block column kb+1 (rows k1 .. N-1, columns k1 .. k1+7) -= panel kb contribution (one element per thread)
      const int n_el = (N - k1) * kBlk;
      for (int e = tid; e < n_el; e += kSolveThreads) {
        const int row = k1 + (e >> 3), col = k1 + (e & 7);
        if (col > row) continue;
        const double *li = A + row * ld + k0, *lj = A + col * ld + k0;
        double sacc = 0;
#pragma unroll
        for (int c = 0; c < kBlk; ++c) sacc += li[c] * lj[c];
        A[row * ld + col] -= sacc;
      }
    }
    __syncthreads();
    DSOPP_STAMP(17 + 3 * kb);
    if (wave == 0) {
      if (kb + 1 < F) factorAndPanel(kb + 1);
      DSOPP_STAMP(18 + 3 * kb);
    } else {
      // trailing update of columns >= k2 with panel kb: A_ij -= sum_c L_ic L_jc  (192 threads as a 12 x 16 tile)
      const int t = tid - 64, tr = t >> 4, tc = t & 15;
      for (int row = k2 + tr; row < N; row += 12) {
        const double *li = A + row * ld + k0;
        double lic[kBlk];
#pragma unroll
        for (int c = 0; c < kBlk; ++c) lic[c] = li[c];
        for (int col = k2 + tc; col <= row; col += 16) {
          const double *lj = A + col * ld + k0;
          double sacc = 0;
#pragma unroll
          for (int c = 0; c < kBlk; ++c) sacc += lic[c] * lj[c];
          A[row * ld + col] -= sacc;
        }
      }
    }
    __syncthreads();
    DSOPP_STAMP(19 + 3 * kb);
  }
  // ---- back substitution x = L^-T y (y = row K of L), column-oriented on one wave: lane j carries y_j (and y_{j+64});
  // going down from k = K-1, x_k = y_k / L_kk is broadcast with v_readlane and every lane j < k takes y_j -= L_kj x_k.
  // 4-7 instructions per unknown, no LDS round trip or barrier inside the chain (L_kj is prefetched a frame block ahead).
  DSOPP_STAMP(3);
  if (wave == 0) {
    // No masking anywhere: lane j is consumed at step k = j (x_j = y_j / L_jj); whatever the later steps k < j add to it
    // (entries on / above the diagonal, uninitialised LDS) is never read again.  x_k leaves the chain as a wave-uniform
    // value and is written to LDS by lane 0, eight at a time.
    auto run = [&](auto two_tag) {
      constexpr bool TWO = decltype(two_tag)::value;
      const int j0 = lane, j1 = lane + 64;
      double y0 = j0 < K ? A[K * ld + j0] : 0.0, y1 = (TWO && j1 < K) ? A[K * ld + j1] : 0.0;
      const double gi0 = j0 < K ? Linv[(j0 >> 3) * 36 + lowIdx(j0 & 7, j0 & 7)] : 0.0;
      const double gi1 = (TWO && j1 < K) ? Linv[(j1 >> 3) * 36 + lowIdx(j1 & 7, j1 & 7)] : 0.0;
      double g0[kBlk], g1[kBlk], n0[kBlk], n1[kBlk];
      auto loadBlock = [&](int kb, double *o0, double *o1) {
#pragma unroll
        for (int c = 0; c < kBlk; ++c) {
          o0[c] = A[(kb * kBlk + c) * ld + j0];  // lanes beyond the row read into the next row: in bounds, never used
          if (TWO) o1[c] = j1 < K ? A[(kb * kBlk + c) * ld + j1] : 0.0;
        }
      };
      loadBlock(F - 1, g0, g1);
      for (int kb = F - 1; kb >= 0; --kb) {
        if (kb > 0) loadBlock(kb - 1, n0, n1);
        double xo[kBlk];
#pragma unroll
        for (int c = kBlk - 1; c >= 0; --c) {
          const int k = kb * kBlk + c;
          const double xk = (!TWO || k < 64) ? readLane(y0 * gi0, k & 63) : readLane(y1 * gi1, k & 63);
          xo[c] = xk;
          y0 -= g0[c] * xk;
          if (TWO) y1 -= g1[c] * xk;
        }
        if (lane == 0) {
#pragma unroll
          for (int c = 0; c < kBlk; ++c) xs[kb * kBlk + c] = xo[c];
        }
#pragma unroll
        for (int c = 0; c < kBlk; ++c) {
          g0[c] = n0[c];
          if (TWO) g1[c] = n1[c];
        }
      }
    };
    if (K > 64)
      run(std::true_type{});
    else
      run(std::false_type{});
  }
  __syncthreads();
  if (tid < K) {
    const double x = xs[tid];
    stpl[tid] = -x;
    a.step[tid] = x;
    a.st->step[tid >> 3][tid & 7] = -x;  // problem.hpp:353-357
  }
  DSOPP_STAMP(4);
  if (fast_refresh) {
    // FEJ: only the current reprojection / brightness constants move with the state; all inputs are in registers / LDS
    ldsBarrier();
    Rigid *E = reinterpret_cast<Rigid *>(A);  // [2][F]: exp(+xi_f), exp(-xi_f); A is free now
    if (tid < 2 * F) {
      const int f = tid < F ? tid : tid - F;
      const double sign = tid < F ? 1.0 : -1.0;
      double xi[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) xi[i] = sign * (epsl[kBlk * f + i] + stpl[kBlk * f + i]);
      E[tid] = rigidExp(xi);
    }
    ldsBarrier();
    if (tid < F * F && pp.valid) {
      const int r = tid / F, t = tid - F * (tid / F);
      PairConst &P = a.pc[r * kMaxFrames + t];
      const Rigid T_tr = rigidMul(E[F + t], rigidMul(pp.T0, E[r]));
      // ArrayReprojector ctor — camera_reproject.hpp:235-260 (as buildProjectionMatrices)
      const double ifx = 1.0 / pp.fxr, ify = 1.0 / pp.fyr;
      const double k02 = -pp.cxr * ifx, k12 = -pp.cyr * ify;
      double U[12];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        U[4 * i + 0] = T_tr.R[3 * i + 0] * ifx;
        U[4 * i + 1] = T_tr.R[3 * i + 1] * ify;
        U[4 * i + 2] = T_tr.R[3 * i + 0] * k02 + T_tr.R[3 * i + 1] * k12 + T_tr.R[3 * i + 2];
        U[4 * i + 3] = T_tr.t[i];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        P.M[0 + j] = pp.fxt * U[0 + j] + pp.cxt * U[8 + j];
        P.M[4 + j] = pp.fyt * U[4 + j] + pp.cyt * U[8 + j];
        P.M[8 + j] = U[8 + j];
      }
      const double a_r = ab0l[2 * r] + epsl[kBlk * r + 6] + stpl[kBlk * r + 6];
      const double a_t = ab0l[2 * t] + epsl[kBlk * t + 6] + stpl[kBlk * t + 6];
      P.s = (pp.exposure_t / pp.exposure_r) * exp(a_t - a_r);
      P.b_t = ab0l[2 * t + 1] + epsl[kBlk * t + 7] + stpl[kBlk * t + 7];
      P.b_r = ab0l[2 * r + 1] + epsl[kBlk * r + 7] + stpl[kBlk * r + 7];
    }
  } else {
    __syncthreads();  // the state written above is re-read from memory below
    Rigid *E = reinterpret_cast<Rigid *>(A);
    if (tid < 2 * F) E[tid] = frameIncrement(a.st, tid % F, tid < F ? 1.0 : -1.0);
    __syncthreads();
    // without first-estimate Jacobians every pair constant moves with the state: the host launches pairSetupKernel right
    // after this kernel (keeping that 15 KB routine and its stack frame out of here spares every launch the scratch setup)
    if (tid < F * F && a.fej) refreshPairCurrent(a.frames, a.st, a.pc, tid / F, tid % F, E[tid / F], E[F + tid % F]);
  }
  DSOPP_STAMP(5);
  if (a.ctrl) {
    // prior + marginal energy at the candidate state x = eps + step (calculateEnergy, problem.hpp:293-312)
    double part = 0;
    if (tid < K) {
      const double xc = epsl[tid] + stpl[tid];
      if (a.use_marginal) {
        double sacc = 0;
        for (int k = 0; k < K; ++k) sacc += a.Hm[tid * K + k] * (epsl[k] + stpl[k]);
        part += bm_c * xc + 0.5 * xc * sacc;
      }
      if ((tid & 7) >= 6) {
        const double ab = ab0_c + xc;
        part += 0.5 * ab * a.affine_reg[(tid & 7) - 6] * ab;
      }
    }
    // frame part of the norms acceptStep reports for this candidate (problem.hpp:366-388): read by the deciding kernel from the
    // control block
    double nstate = 0, nstep = 0;
    if (tid < K) {
      nstate = epsl[tid] * epsl[tid] + ((tid & 7) >= 6 ? ab0_c * ab0_c : 0.0);
      nstep = stpl[tid] * stpl[tid];
    }
    part = waveSum(part);
    nstate = waveSum(nstate);
    nstep = waveSum(nstep);
    ldsBarrier();  // E (in A) fully consumed before the scratch below is written; stpl visible
    if ((tid & 63) == 0) {
      xs[tid >> 6] = part;
      xs[4 + (tid >> 6)] = nstate;
      xs[8 + (tid >> 6)] = nstep;
    }
    ldsBarrier();
    if (tid == 0) {
      static_assert(kSolveThreads / 64 == 4, "three groups of four wave sums in xs");
      double total = a.energy_marginalized, s_state = 0, s_step = 0;
      for (int w = 0; w < kSolveThreads / 64; ++w) {
        total += xs[w];
        s_state += xs[4 + w];
        s_step += xs[8 + w];
      }
      a.ctrl->cand_prior = total;
      a.ctrl->frame_state_sq = s_state;
      a.ctrl->frame_step_sq = s_step;
      a.ctrl->pending = 1;
    }
  }
  DSOPP_STAMP(6);
}

// ---------------------------------------------------------------------------------------------------------------
// back-substitution, energy reduction, LM control
// ---------------------------------------------------------------------------------------------------------------
/** calculateIdepths — hessian_block_evaluation.hpp:238-263 */
__global__ void backsubKernel(const FrameDev *__restrict__ frames, const SchurBlock *__restrict__ table, const double *__restrict__ step,
                              double lambda, int F, const LmControl *ctrl, int ublk_parity, int gate_on_pending) {
  if (ctrl) {
    if (!ctrl->active || (gate_on_pending && !ctrl->pending)) return;
    lambda = ctrl->lambda;
  }
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
    const double *src = fr.ublk + (static_cast<size_t>(ublk_parity) * kMaxFrames + t) * ublkPlane(fr.cap) + static_cast<size_t>(i) * kUblk;
#pragma unroll
    for (int c = 0; c < kBlk; ++c) d += src[c] * step[kBlk * t + c];
  }
  const double s = (fr.b_d[i] - d) * (1.0 / (1.0 + lambda)) * fr.inv_hdd[i];
  fr.idepth_step[i] = -s;
}

/** sums the (energy, n_valid) partials of a sweep: out[0] = energy, out[1] = n_valid */
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
 *  norms[0] += sum idepth^2 (before), norms[1] += sum idepth_step^2. */
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
  __shared__ double lds[4];
  const int c = threadIdx.x;  // 128 threads
  double v[2] = {0, 0};
  if (c < kBlk * F) {
    const int f = c >> 3, a = c & 7;
    if (accept) {
      const double e = st->eps[f][a], s = st->step[f][a];
      v[0] = e * e;
      v[1] = s * s;
      if (a < 2) v[0] += st->ab0[f][a] * st->ab0[f][a];
      st->eps[f][a] = e + s;
    }
    st->step[f][a] = 0;
  }
  blockSum<2, 128>(v, lds);
  if (c == 0 && accept) {
    norms[0] = v[0];
    norms[1] = v[1];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// device-side Levenberg-Marquardt control — levenberg_marquardt_algorithm.hpp:77-128
// ---------------------------------------------------------------------------------------------------------------
struct LmInitArgs {
  SolveArgs sa;            // frames / st / Hm / bm / regularisers / energy_marginalized
  const double *partials;  // sweep partials of the initial energy evaluation
  const double *scalars;   // multi-GPU: all-reduced {energy, n_valid, -, -, idepth_sq}
  const SchurBlock *schur_table;
  int n_sweep_blocks, n_schur_blocks;
  LmControl *ctrl;  // [2]
  LmParams prm;
  // nullable: the pair constants of every ordered frame pair are set up here too (a solve that starts from a state whose constants
  // are not current: restore, new keyframe, accepted steps of the previous solve) — one launch less at the head of the solve
  const FrameDev *pair_frames = nullptr;
  PairConst *pair_pc = nullptr;
  int pair_fej = 0;
  // nullable: the solve starts from the window's snapshot (optimize_repeated) — workgroups >= 1 put the landmarks back (one chunk
  // each), the last workgroup copies the frame states, workgroup 0 reads them from the snapshot itself, so nothing in this launch waits for a copy
  // (sa.st then IS the snapshot — this kernel only reads the states — and restore_state the live block the copy goes to)
  WindowState *restore_state = nullptr;
};

/** sum of idepth^2 over this rank's landmarks (state norm of acceptStep, problem.hpp:379); grid = schur blocks */
__global__ void idepthNormKernel(const FrameDev *__restrict__ frames, const SchurBlock *__restrict__ table, double *out) {
  __shared__ double lds[kSchurThreads / 64];
  const SchurBlock be = table[blockIdx.x];
  const FrameDev &fr = frames[be.r];
  const int i = be.offset + threadIdx.x;
  double v[1] = {0};
  if (threadIdx.x < kSchurLandmarks && i < fr.n) v[0] = fr.idepth[i] * fr.idepth[i];
  blockSum<1, kSchurThreads>(v, lds);
  if (threadIdx.x == 0) atomicAdd(out, v[0]);
}

/** result = calculateEnergy() before the loop (levenberg_marquardt_algorithm.hpp:82); single workgroup */
__global__ void __launch_bounds__(kSolveThreads) lmInitKernel(LmInitArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double *lds = reinterpret_cast<double *>(smem_raw);
  __shared__ double red[(kSolveThreads / 64) * 2];
  const int tid = threadIdx.x;
  double v[2] = {0, 0};
  if (a.prm.use_reduced_scalars) {
    v[0] = tid == 0 ? a.scalars[0] : 0.0;
    v[1] = tid == 0 ? a.scalars[1] : 0.0;
  } else {
    for (int b = tid; b < a.n_sweep_blocks; b += kSolveThreads) {
      v[0] += a.partials[static_cast<size_t>(b) * kPartial + 44];
      v[1] += a.partials[static_cast<size_t>(b) * kPartial + 45];
    }
  }
  blockSum<2, kSolveThreads>(v, red);
  const double prior = priorEnergyBlock(a.sa, false, lds, tid);
  if (tid == 0) {
    LmControl c;
    c.lambda = a.prm.lambda0;
    c.energy = v[0] + prior;
    c.cand_prior = 0;
    c.frame_state_sq = c.frame_step_sq = 0;
    c.idepth_sq = a.scalars[4];
    c.n_valid = static_cast<int>(v[1] + 0.5);
    c.converged = 0;
    c.iteration = 0;
    c.linear_system_valid = 0;
    c.need_final_setup = 0;
    c.active = (a.prm.max_iterations > 0 && c.n_valid > 0) ? 1 : 0;
    c.pending = 0;
    c.relin = 0;
    a.ctrl[0] = c;
    a.ctrl[1] = c;
  }
}

struct LmDecideArgs {
  const FrameDev *frames;
  WindowState *st;
  const SchurBlock *schur_table;
  const double *partials;  // energy sweep partials: [44] energy, [45] n_valid, [46] sum step^2, [47] sum idepth*step
  const double *scalars;   // multi-GPU: the same four sums, all-reduced
  const LmControl *ctrl_in;
  LmControl *ctrl_out;
  int n_sweep_blocks, n_schur_blocks;
  int F;
  LmParams prm;
};

/**
 * One loop body's tail (levenberg_marquardt_algorithm.hpp:93-123): compare energies, accept or reject, update lambda and
 * the convergence flags, and apply the decision to the state (acceptStep / rejectStep, problem.hpp:366-402).
 * grid = one workgroup per landmark chunk (+ at least one): every workgroup derives the same decision from the same
 * inputs (deterministic reductions), applies it to its own landmarks; workgroup 0 also moves the frame states and
 * writes the outgoing control block.
 */
__global__ void __launch_bounds__(kSchurThreads) lmDecideKernel(LmDecideArgs a) {
  __shared__ double red[(kSchurThreads / 64) * 4];
  const LmControl cin = *a.ctrl_in;
  const int tid = threadIdx.x;
  if (!cin.active) {
    if (blockIdx.x == 0 && tid == 0) *a.ctrl_out = cin;
    return;
  }
  // energy, n_valid, step_sq(idepth), idepth.step; the frame part of the norms comes from the control block (written by the solve
  // kernel): workgroup 0 of THIS launch moves the frame states, later workgroups must not derive their decision from them
  double v[4] = {0, 0, 0, 0};
  if (a.prm.use_reduced_scalars) {
    if (tid == 0) {
      v[0] = a.scalars[0];
      v[1] = a.scalars[1];
      v[2] = a.scalars[2];
      v[3] = a.scalars[3];
    }
  } else {
    for (int b = tid; b < a.n_sweep_blocks; b += kSchurThreads) {
      const double *p = a.partials + static_cast<size_t>(b) * kPartial;
      v[0] += p[44];
      v[1] += p[45];
      v[2] += p[46];
      v[3] += p[47];
    }
  }
  blockSum<4, kSchurThreads>(v, red);
  __shared__ int s_accept;
  __shared__ LmControl s_out;
  if (tid == 0) {
    LmControl c = cin;
    const double next_energy = v[0] + cin.cand_prior;
    const int n_valid = static_cast<int>(v[1] + 0.5);
    int accept = 0;
    c.iteration = cin.iteration + 1;
    if (n_valid == 0) {
      // problem.rejectStep(); break;
      c.active = 0;
      c.need_final_setup = 1;
    } else {
      if (fabs(cin.energy - next_energy) / cin.energy < a.prm.function_tolerance) c.converged = 1;
      if (next_energy < cin.energy || (a.prm.force_accept && cin.iteration < a.prm.min_iterations)) {
        accept = 1;
        const double state_sq = cin.frame_state_sq + cin.idepth_sq, step_sq = cin.frame_step_sq + v[2];
        if (step_sq < a.prm.parameter_tolerance * (state_sq + a.prm.parameter_tolerance)) c.converged = 1;
        c.energy = next_energy;
        c.n_valid = n_valid;
        c.lambda = cin.lambda / a.prm.decrease_on_accept;
        c.linear_system_valid = 0;
        c.idepth_sq = cin.idepth_sq + 2.0 * v[3] + v[2];
        c.need_final_setup = 0;
      } else {
        c.need_final_setup = 1;
        if (a.prm.force_accept) {
          c.active = 0;  // problem.calculateEnergy(); return result;
        } else {
          c.lambda = cin.lambda * a.prm.increase_on_reject;
          c.linear_system_valid = 1;
        }
      }
      if (c.converged || c.iteration >= a.prm.max_iterations) c.active = 0;
    }
    s_accept = accept;
    s_out = c;
  }
  __syncthreads();
  const int accept = s_accept;
  // landmarks of this workgroup's chunk
  if (static_cast<int>(blockIdx.x) < a.n_schur_blocks) {
    const SchurBlock be = a.schur_table[blockIdx.x];
    const FrameDev &fr = a.frames[be.r];
    const int i = be.offset + tid;
    if (tid < kSchurLandmarks && i < fr.n) {
      if (accept) fr.idepth[i] += fr.idepth_step[i];
      fr.idepth_step[i] = 0;
      for (int t = 0; t < a.F; ++t) {
        if (fr.status[t] == nullptr || i >= fr.n_res[t]) continue;
        if (accept)
          fr.status[t][i] = fr.cand[t][i];
        else
          fr.cand[t][i] = fr.status[t][i];
      }
    }
  }
  if (blockIdx.x == 0) {
    if (tid < kBlk * a.F) {
      const int f = tid >> 3, c = tid & 7;
      if (accept) a.st->eps[f][c] += a.st->step[f][c];
      a.st->step[f][c] = 0;
    }
    if (tid == 0) *a.ctrl_out = s_out;
  }
}

/** one chunk of 64 landmarks back to the snapshot: idepths, flags and connection statuses (threads >= 64 idle) */
__device__ __forceinline__ void restoreLandmarkChunk(const FrameDev *__restrict__ frames, const SchurBlock *__restrict__ table, int F, int chunk) {
  const SchurBlock be = table[chunk];
  const FrameDev &fr = frames[be.r];
  const int i = be.offset + threadIdx.x;
  if (threadIdx.x >= kSchurLandmarks || i >= fr.n) return;
  fr.idepth[i] = fr.snap_idepth[i];
  fr.idepth_step[i] = 0;
  fr.flags[i] = fr.snap_flags[i];
  for (int t = 0; t < F; ++t) {
    if (fr.status[t] == nullptr || i >= fr.n_res[t]) continue;
    const uint8_t s = fr.snap_status[t][i];
    fr.status[t][i] = s;
    fr.cand[t][i] = s;
  }
}

__device__ __forceinline__ void restoreFrameStates(WindowState *state, const WindowState *state_snap) {
  constexpr int kWords = static_cast<int>(sizeof(WindowState) / sizeof(double));
  for (int k = threadIdx.x; k < kWords; k += blockDim.x) reinterpret_cast<double *>(state)[k] = reinterpret_cast<const double *>(state_snap)[k];
}

/** dsopp_hip_window_restore: idepths, flags and connection statuses back to the snapshot; grid = schur blocks */
__global__ void restoreKernel(const FrameDev *__restrict__ frames, const SchurBlock *__restrict__ table, int F, WindowState *state,
                              const WindowState *state_snap, int n_schur_blocks) {
  if (static_cast<int>(blockIdx.x) == n_schur_blocks) {
    // last workgroup: the frame states (one launch restores everything; no separate device-to-device copy)
    restoreFrameStates(state, state_snap);
    return;
  }
  restoreLandmarkChunk(frames, table, F, blockIdx.x);
}

/** fused loop: control block before the first sweep (prior energy of the initial state goes to cand_prior); one workgroup */
__global__ void __launch_bounds__(kSolveThreads) lmBeginKernel(LmInitArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  if (a.restore_state) {
    if (blockIdx.x > 0) {
      // (the frame states are copied by a workgroup of their own: stores in workgroup 0 would sit in front of its loads in vmcnt)
      if (static_cast<int>(blockIdx.x) <= a.n_schur_blocks)
        restoreLandmarkChunk(a.sa.frames, a.schur_table, a.sa.F, blockIdx.x - 1);
      else
        restoreFrameStates(a.restore_state, a.sa.st);
      return;
    }
  }
  const double idepth_sq = 0;  // set by the decide step of the opening round (sum idepth^2 rides in the sweep's partials)
  if (a.pair_frames) {
    // the 2 F frame increments exp(+-(eps + step)) once per frame, shared through LDS, instead of twice per ordered pair
    __shared__ Rigid s_inc[2 * kMaxFrames];
    const int F = a.sa.F, tid = threadIdx.x;
    if (tid < 2 * F) s_inc[tid] = frameIncrement(a.sa.st, tid < F ? tid : tid - F, tid < F ? 1.0 : -1.0);
    __syncthreads();
    if (2 * F * F <= kSolveThreads - 64) {
      // the two halves of a pair's constants on two different waves (SIMDs): wave 0.. the state-dependent half, from thread 64 +
      // ... the linearisation-point half (rounded up to a wave boundary so that no wave runs both instruction streams)
      const int second = (F * F + 63) & ~63;
      if (tid < F * F)
        computePairConst<1>(a.pair_frames, a.sa.st, a.pair_pc, tid / F, tid % F, F, a.pair_fej != 0, &s_inc[tid / F], &s_inc[F + tid % F]);
      else if (tid >= second && tid < second + F * F)
        computePairConst<2>(a.pair_frames, a.sa.st, a.pair_pc, (tid - second) / F, (tid - second) % F, F, a.pair_fej != 0, &s_inc[(tid - second) / F],
                            &s_inc[F + (tid - second) % F]);
    } else if (tid < F * F) {
      computePairConst<0>(a.pair_frames, a.sa.st, a.pair_pc, tid / F, tid % F, F, a.pair_fej != 0, &s_inc[tid / F], &s_inc[F + tid % F]);
    }
    __syncthreads();  // (priorEnergyBlock below has its own scratch; keep the phases apart)
  }
  const double prior = priorEnergyBlock(a.sa, false, reinterpret_cast<double *>(smem_raw), threadIdx.x);
  if (threadIdx.x == 0) {
    LmControl c;
    c.lambda = a.prm.lambda0;
    c.energy = 0;
    c.cand_prior = prior;
    c.frame_state_sq = c.frame_step_sq = 0;
    c.idepth_sq = idepth_sq;
    c.n_valid = 0;
    c.converged = 0;
    c.active = 1;
    c.linear_system_valid = 0;
    c.iteration = 0;
    c.need_final_setup = 0;
    c.pending = 0;
    c.relin = 0;
    a.ctrl[0] = c;
    a.ctrl[1] = c;
  }
}

/** large windows: the four per-block scalars of a sweep summed in kScalarGroups fixed groups (out[g][4]) — every workgroup of
 *  the decision kernel then adds 64 x 4 numbers instead of walking tens of thousands of sweep blocks itself */
__global__ void __launch_bounds__(256) sweepScalarGroupsKernel(const double *__restrict__ partials, int n_blocks, double *out, const LmControl *ctrl) {
  __shared__ double lds[(256 / 64) * 4];
  double v[4] = {0, 0, 0, 0};
  const bool live = !ctrl || ctrl->active;
  const int per = (n_blocks + kScalarGroups - 1) / kScalarGroups;
  const int b0 = blockIdx.x * per, b1 = min(b0 + per, n_blocks);
  if (live)
    for (int b = b0 + threadIdx.x; b < b1; b += 256) {
      const double *p = partials + static_cast<size_t>(b) * kPartial;
      v[0] += p[44];
      v[1] += p[45];
      v[2] += p[46];
      v[3] += p[47];
    }
  blockSum<4, 256>(v, lds);
  if (threadIdx.x == 0) {
    out[4 * blockIdx.x + 0] = v[0];
    out[4 * blockIdx.x + 1] = v[1];
    out[4 * blockIdx.x + 2] = v[2];
    out[4 * blockIdx.x + 3] = v[3];
  }
}

/** the group sums added up (fixed order) into out[0..3]: the sharded large-window path needs the four scalars contiguous for its
 *  collective */
__global__ void sweepScalarGroupsFinalKernel(const double *__restrict__ groups, double *out) {
  if (threadIdx.x < 4) {
    double s = 0;
    for (int g = 0; g < kScalarGroups; ++g) s += groups[4 * g + threadIdx.x];
    out[threadIdx.x] = s;
  }
}

/** sums the four per-block scalars of an energy sweep into out[0..3] (multi-GPU path, before the all-reduce) */
__global__ void sweepScalarsKernel(const double *__restrict__ partials, int n_blocks, double *out, const LmControl *ctrl) {
  __shared__ double lds[(256 / 64) * 4];
  double v[4] = {0, 0, 0, 0};
  const bool live = !ctrl || ctrl->active;
  if (live)
    for (int b = threadIdx.x; b < n_blocks; b += blockDim.x) {
      const double *p = partials + static_cast<size_t>(b) * kPartial;
      v[0] += p[44];
      v[1] += p[45];
      v[2] += p[46];
      v[3] += p[47];
    }
  blockSum<4, 256>(v, lds);
  if (threadIdx.x == 0) {
    out[0] = v[0];
    out[1] = v[1];
    out[2] = v[2];
    out[3] = v[3];
  }
}

}  // namespace dsopp_hip
