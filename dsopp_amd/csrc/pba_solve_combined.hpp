// K3 of the fused Levenberg-Marquardt loop: priors + dense solve of the combined system the reduction kernel emits
// (pba_solve_kernels.hpp: ReduceSchurArgs::comb), candidate pair constants, prior energy of the candidate — one workgroup; the
// launch's other workgroups apply the LM decision to the landmarks and back-substitute the inverse depths for the new step while the
// solving workgroup finishes (SolveCombArgs::bs_ticket).
//
// Replaces, like assembleSolveKernel (which stays for the stage API / marginalisation / covariance paths that need the four
// systems separately): evaluateLinearSystemPrior (problem.hpp:37-77), calculateStep (:342-361), NormalLinearSystem::solve
// (normal_linear_system.cpp:10-16,52-59).  What is different from assembleSolveKernel:
//   * input is ONE block-packed lower triangle (14 KB at 7 frames) instead of H_pp and H_schur as K x K matrices (2 x 25 KB) plus
//     their right-hand sides: the load phase is a single coalesced sweep, 7 loads per thread;
//   * the Jacobi guard (pivot threshold) is taken while the diagonal is written, not in a pass of its own.
// The factorisation itself (blocked Cholesky over the frame blocks with a look-ahead panel wave, right-hand side as an extra row,
// v_rsq_f64 seed + two Newton steps, pivots below 1e-30 of the Jacobi-scaled diagonal treated as zero) and the back-substitution
// are those of assembleSolveKernel.
#pragma once
#include "pba_solve_kernels.hpp"

namespace dsopp_hip {

struct SolveCombArgs {
  const FrameDev *frames;
  WindowState *st;
  PairConst *pc;
  const double *comb;  // combBlockCount(F) * 64 block entries, then K right-hand-side entries
  // the reduction launch spread its atomics over comb_copies copies of the system (ReduceSchurArgs::comb_copies): copy c > 0 starts
  // comb_copy_first + (c - 1) * comb_copy_stride doubles behind `comb`; the load phase adds them entry by entry (fixed order)
  int comb_copies = 1;
  int comb_copy_first = 0, comb_copy_stride = 0;
  const double *Hm, *bm;  // marginal prior
  // the marginal prior's matrix once more in the layout of `comb` (block-packed lower triangle, packed by the host with the prior:
  // pba.hip uploadMarginal): the load phase adds it entry by entry at the SAME offsets as the combined system — no per-entry block
  // decode (a square root and a dozen integer instructions per entry) and no second set of addresses in the prologue, whose register
  // pressure had the 512-thread kernel spill freshly loaded values (and wait for them) in front of the decision
  const double *HmPacked = nullptr;
  double *step;        // out: K
  LmControl *ctrl;     // nullable (isolated timing launches pass lambda explicitly)
  double lambda;
  double affine_reg[2];
  double fixed_reg;
  double energy_marginalized;
  int F;
  int fej;
  int use_marginal;
  long long *dbg_stamps;  // compiled in with -DDSOPP_HIP_STAMPS only
  // Landmark-sharded windows: the LM decision for the pending candidate runs HERE instead of in a kernel of its own between the
  // collective and the solve.  Its inputs are the incoming control block and the four all-reduced sums — no reduction — so every
  // workgroup of this launch evaluates it for itself (lmDecision): workgroup 0 moves the frame states, publishes the outgoing control
  // block (`ctrl`) and solves; workgroups 1 .. dec_blocks apply accept / reject to the landmarks of one 64-landmark chunk each.
  const LmControl *dec_in = nullptr;  // nullable = no decision here (single-GPU windows decide in the reduction kernel)
  const double *dec_scalars = nullptr;  // {energy, n_valid, |idepth step|^2, idepth . step}, summed over all ranks
  const SchurBlock *dec_table = nullptr;
  int dec_blocks = 0;
  int dec_groups = 0;  // 0: dec_scalars holds the four sums; kScalarGroups: it holds that many group sums [g][4] (added here, fixed order)
  LmParams dec_prm;
  // calculateIdepths for the step this launch solves for (hessian_block_evaluation.hpp:238-263), by the landmark workgroups: they
  // request their landmarks' Schur rows while workgroup 0 factorises, wait for its step and finish under its tail
  // (pair constants, prior energy) — the back-substitution costs neither a launch nor a pass over the rows on the critical path.
  // dec_chunks 64-landmark chunks are dealt out over the dec_blocks workgroups (all resident at once: sized by the host).
  // Hand-over: the step travels through one of two buffers whose slots hold a NaN pattern no step can take until the solver stores
  // into them (device scope): a value that is not the sentinel IS the step — the waiters poll the values themselves, one read from
  // memory instead of a flag and then the data.  Every launch re-arms the OTHER buffer (nobody reads it before the next launch).
  unsigned *bs_ticket = nullptr;   // nullable = no back-substitution here; the ticket counter (below)
  double *bs_hand = nullptr;       // [kBlk * kMaxFrames] this launch's hand-over slots (armed by the previous launch)
  double *bs_hand_next = nullptr;  // the other buffer: armed here
  int bs_parity = 0;            // which half of the double-buffered Schur rows this round's linearisation wrote
  int dec_chunks = 0;
  // Who solves is decided by arrival, not by block index: the first workgroup of the launch to draw a ticket (counting on from
  // bs_ticket_base) is "workgroup 0".  The waiting workgroups then wait for one that is certainly running — block 0 need not be: the
  // XCDs dispatch their shares of a grid independently, and with another process's (or stream's) kernels holding XCD 0 the blocks
  // 1, 2, ... were resident and waiting while block 0 was not (two ranks on one device deadlocked exactly so, with the other
  // rank's collective waiting for this rank in turn).
  unsigned bs_ticket_base = 0;
  // Where a landmark workgroup whose bounded wait for the step ran out says so: one int in pinned host memory (0 = fine).  The workgroup
  // then leaves its landmarks' idepth_step untouched and returns; the host finds the word set at the solve's synchronisation and fails
  // the solve with DSOPP_HIP_ERR_HIP.  (Until round 5 the waiter trapped, which takes the whole HIP context — every window, aligner and
  // group of the process — down with it.)
  int *bs_fault = nullptr;
};
/** what an armed hand-over slot holds: a quiet NaN with a payload no arithmetic produces */
__host__ __device__ inline double kHandOverSentinel() {
  union {
    unsigned long long u;
    double d;
  } c;
  c.u = 0x7FF8D50FF00DBEEFull;
  return c.d;
}
constexpr int kMaxCombCopies = 4;
#define SC_STAMP(i) do { if (kStamps && a.dbg_stamps && tid == 0) a.dbg_stamps[i] = wall_clock64(); } while (0)

/** block index b of the packed lower triangle -> (bi, bj), bj <= bi */
__device__ __forceinline__ void combBlockDecode(int b, int &bi, int &bj) {
  int r = static_cast<int>((__fsqrt_rn(8.0f * static_cast<float>(b) + 1.0f) - 1.0f) * 0.5f);
  if ((r + 1) * (r + 2) / 2 <= b) ++r;
  if (r * (r + 1) / 2 > b) --r;
  bi = r;
  bj = b - r * (r + 1) / 2;
}

/** THREADS = 256 up to 8 frames (K <= 64: one row per lane of the panel wave); 512 above: the trailing update of the early block
 *  columns of a 9..16-frame system is longer than the panel wave's factor step with three waves (12 frames: Cholesky 20 us), and the
 *  combined system arrives in two rounds of batched loads instead of three.  Every matrix entry is produced by one thread with a
 *  fixed summation order, so the two instantiations give bit-identical results. */
/** (the leading arguments repeat members of `a`: the dispatcher preloads the first 16 argument words into scalar registers — build.sh:
 *  -amdgpu-kernarg-preload-count — but not the members of a by-value struct.  With them the ticket and the solver's first operand loads
 *  leave in the wave's first cycles, beside the rest of the argument block instead of behind it.) */
/** COPIES: 1, or kMaxCombCopies for a system the reduction launch accumulated in several copies (SolveCombArgs::comb_copies of them are live) */
template <int THREADS, int COPIES = 1>
__global__ void __launch_bounds__(THREADS, 1) solveCombinedKernel(unsigned *bs_ticket_p, double *bs_hand_next_p, const LmControl *dec_in_p,
                                                                  const double *dec_scalars_p, const double *comb_p, const FrameDev *frames_p,
                                                                  WindowState *st_p, int F_p, SolveCombArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int F = F_p, K = kBlk * F;
  const int N = K + 1;   // augmented with the right-hand side row
  const int ld = N + 1;
  double *A = reinterpret_cast<double *>(smem_raw);  // N x ld (lower triangle used)
  double *pv = A + N * ld;                           // K pivot guards
  double *xs = pv + K;                               // K + 16 scratch
  double *Linv = xs + K + 16;                        // 36 x kMaxFrames doubles: prior diagonal during assembly, then pivots | reciprocals
  double *epsl = Linv + 36 * kMaxFrames;             // K: state increment eps of every frame
  double *stpl = epsl + K;                           // K: the new step (-x), kept in LDS for the pair refresh / prior energy
  double *ab0l = stpl + K;                           // 2 F: affine brightness at the linearisation point
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const long long sc_t_entry = (kStamps && a.dbg_stamps) ? wall_clock64() : 0;  // (tuning aid: the solving workgroup's first cycle)

  // ---- everything is requested before anything waits: the kernel start costs one memory round trip
  int c_active = 1, c_relin = 0;
  double lam = a.lambda;
  const bool decides = dec_in_p != nullptr;
  double dec_eps = 0, dec_step = 0;
  int dec_accept = 0;
  // Workgroup 0 requests everything that does not depend on the decision — pair constants, frame flags, right-hand side, the first
  // batch of the combined system — BEFORE it takes the decision: the decision's own loads, its LDS tree and its scalar chain then
  // run under these loads' round trip instead of in front of it (the other workgroups only apply the decision and leave).
  const bool ticketed = bs_ticket_p != nullptr;
  // The ticket is drawn here and NOT looked at before the decision below: the returned value stays in its register, un-waited-for, under
  // everything requested in between.  (Until round 5 the base was subtracted right here — the compiler then waits for the atomic's
  // return, a device-scope round trip of 1 - 2 us, in front of every operand load of the solver.)
  unsigned ticket_raw = 0;
  if (ticketed && tid == 0) {
    // (the address is made opaque: for a wave-uniform address the compiler rewrites the add as a wave scan whose v_readfirstlane
    // consumes — and waits for — the returned value on the spot)
    auto tp = glb(bs_ticket_p);
    asm volatile("" : "+v"(tp));
    ticket_raw = __hip_atomic_fetch_add(tp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (ticketed && tid < kBlk * kMaxFrames) bs_hand_next_p[tid] = kHandOverSentinel();  // (every workgroup: the same value; read by the next launch)
  // The decision's inputs — the incoming control block and the sweep's four sums — as SCALAR loads, requested in the launch's first
  // cycles (both addresses are preloaded arguments): scalar loads return on their own counter, so the decision below runs as soon as these
  // 30 words are here, under the flight of the solver's vector operands instead of behind it.  (Until round 5 they were vector loads issued
  // behind ~300 instructions of address arithmetic, and the four sums were fetched only after the control block had been tested.)
  // Absent control block (isolated timing launches): the argument block stands in, the values are not used.
  const void DSOPP_CONSTANT *any_words = (const void DSOPP_CONSTANT *)__builtin_amdgcn_kernarg_segment_ptr();
  const LmControl DSOPP_CONSTANT *cin_src = dec_in_p ? (const LmControl DSOPP_CONSTANT *)dec_in_p : (const LmControl DSOPP_CONSTANT *)any_words;
  // (the stand-in has to cover every word read through it, and the field-by-field copy below every field: `*a.ctrl = cin` writes all of it)
  static_assert(sizeof(a) >= sizeof(LmControl) && sizeof(a) >= 4 * sizeof(double), "the argument block stands in for an absent control block / scalar block");
  static_assert(sizeof(LmControl) == 6 * sizeof(double) + 8 * sizeof(int), "LmControl changed: extend the scalar copy below");
  LmControl cin;
  cin.lambda = cin_src->lambda;
  cin.energy = cin_src->energy;
  cin.cand_prior = cin_src->cand_prior;
  cin.idepth_sq = cin_src->idepth_sq;
  cin.frame_state_sq = cin_src->frame_state_sq;
  cin.frame_step_sq = cin_src->frame_step_sq;
  cin.n_valid = cin_src->n_valid;
  cin.converged = cin_src->converged;
  cin.active = cin_src->active;
  cin.linear_system_valid = cin_src->linear_system_valid;
  cin.iteration = cin_src->iteration;
  cin.need_final_setup = cin_src->need_final_setup;
  cin.pending = cin_src->pending;
  cin.relin = cin_src->relin;
  const double DSOPP_CONSTANT *t_src = dec_scalars_p ? (const double DSOPP_CONSTANT *)dec_scalars_p : (const double DSOPP_CONSTANT *)any_words;
  double t_early[4] = {t_src[0], t_src[1], t_src[2], t_src[3]};
  // Every other argument word the kernel's head reads is requested in ONE burst of scalar loads with one wait: left to itself the
  // compiler fetches each member where it is first used — a dozen s_load / s_waitcnt pairs in a row in front of the operand loads,
  // four of them on argument lines no wave of the launch had touched yet.
  // (the decision's inputs ride in the same burst and the same wait: pinned, because loads from the constant address space may otherwise
  // be sunk to their uses behind the branches below)
  asm volatile("" : "+s"(cin.lambda), "+s"(cin.energy), "+s"(cin.cand_prior), "+s"(cin.idepth_sq), "+s"(cin.frame_state_sq), "+s"(cin.frame_step_sq),
               "+s"(cin.n_valid), "+s"(cin.converged), "+s"(cin.active), "+s"(cin.linear_system_valid), "+s"(cin.iteration),
               "+s"(cin.need_final_setup), "+s"(cin.pending), "+s"(cin.relin), "+s"(t_early[0]), "+s"(t_early[1]), "+s"(t_early[2]), "+s"(t_early[3])
               : "s"(a.pc), "s"(a.Hm), "s"(a.bm), "s"(a.HmPacked), "s"(a.step), "s"(a.ctrl), "s"(a.fej), "s"(a.use_marginal), "s"(a.dec_table),
               "s"(a.dec_blocks), "s"(a.dec_groups), "s"(a.bs_hand), "s"(a.bs_parity), "s"(a.dec_chunks), "s"(a.bs_ticket_base), "s"(a.comb_copies),
               "s"(a.comb_copy_first), "s"(a.comb_copy_stride), "s"(a.lambda), "s"(a.affine_reg[0]), "s"(a.affine_reg[1]), "s"(a.fixed_reg),
               "s"(a.energy_marginalized));
  __shared__ unsigned s_vblock;
  // with tickets every workgroup requests the solver's operands (it does not know its role yet; the others drop them)
  bool main_wg = ticketed || blockIdx.x == 0;
  struct {
    int valid;
    Rigid T0;
    double fxr, fyr, cxr, cyr, fxt, fyt, cxt, cyt, exposure_r, exposure_t;
  } pp;
  pp.valid = 0;
  const bool fast_refresh = a.fej != 0;
  auto requestPairInputs = [&] {
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
      const FrameDev &fr = frames_p[r];
      const FrameDev &ft = frames_p[t];
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
  };
  // (with tickets the role is unknown here and these 22 doubles would be live across the landmark workgroups' register-hungry branch:
  // the compiler spilled them at the head and reloaded them in front of the pair refresh; the solver requests them once it knows
  // what it is — they are not needed before the kernel's tail)
  // (requested at ONE place, behind the decision — below: with a second request here for the launches without tickets the two sets of loaded
  // values met in different registers where the paths join, and the copies that unify them wait for the loads: a full memory round trip
  // in front of the assembly, on every launch)
  double eps_c = 0, ab0_c = 0, rhs_c = 0, bm_c = 0;  // this thread's entry c = tid (K <= 128 < THREADS)
  int fixed_c = 0, tomarg_c = 0;
  if (main_wg && tid < K) {
    const int f = tid >> 3, i = tid & 7;
    if (!decides) eps_c = st_p->eps[f][i];
    if (decides) {
      // the frame states the decision moves (read and written by the solving workgroup only): requested here with everything else — behind
      // the control block's arrival, where they used to be requested, their round trip was 0.7 us of waiting behind the decision
      dec_eps = st_p->eps[f][i];
      dec_step = st_p->step[f][i];
    }
    fixed_c = frames_p[f].fixed;
    tomarg_c = frames_p[f].to_marginalize;
    ab0_c = st_p->ab0[f][i < 6 ? 0 : i - 6];
    rhs_c = comb_p[combBlockCount(F) * 64 + tid];
    if (COPIES > 1) {
      for (int cp = 1; cp < a.comb_copies; ++cp)
        rhs_c += comb_p[a.comb_copy_first + static_cast<size_t>(cp - 1) * a.comb_copy_stride + combBlockCount(F) * 64 + tid];
    }
    if (a.use_marginal) bm_c = a.bm[tid];
  }
  // the prior energy of the candidate (kernel tail) is evaluated by the threads 64 .. 64 + K - 1, i.e. on other waves than the
  // pair-constant refresh it runs beside; their operands are requested here with everything else
  // (requested behind the factorisation — requestTailInputs below —: fetched here they were live, or spilled, across the whole solve)
  const int pc = tid - 64;
  const bool prior_thread = main_wg && pc >= 0 && pc < K;
  double ab0_p = 0, bm_p = 0;
  auto requestTailInputs = [&] {
    if (prior_thread) {
      ab0_p = st_p->ab0[pc >> 3][(pc & 7) < 6 ? 0 : (pc & 7) - 6];
      if (a.use_marginal) bm_p = a.bm[pc];
    }
  };
  if (!ticketed) requestTailInputs();
  // the block-packed lower triangle: entry e = tid + 256 u, coalesced.  kBatch loads are in flight per thread.
  const int n_entries = combBlockCount(F) * 64;
  // (512 threads: 6 per batch cover the 4992 entries of 12 frames in two rounds and the 8704 of 16 frames in three, as 8 would, with
  // eight registers fewer held across the decision)
  constexpr int kBatch = THREADS >= 512 ? 6 : 8;
  double hv[kBatch], hm[kBatch];
  auto loadBatch = [&](int base) {
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int e = base + tid + THREADS * u;
      hv[u] = comb_p[min(e, n_entries - 1)];  // clamped, unconditional (a select around a load makes hipcc branch per element)
      hm[u] = 0;
    }
    // (the other copies of a system accumulated in several: all requested before the first is added)
    if (COPIES > 1) {  // (a kernel of its own: the plain one carries neither the test nor the registers)
      double hc[COPIES > 1 ? COPIES - 1 : 1][kBatch];
#pragma unroll
      for (int cp = 1; cp < COPIES; ++cp) {
        const double *cpy = comb_p + a.comb_copy_first + static_cast<size_t>(cp - 1) * a.comb_copy_stride;
#pragma unroll
        for (int u = 0; u < kBatch; ++u) hc[cp - 1][u] = 0;
        if (cp < a.comb_copies) {
#pragma unroll
          for (int u = 0; u < kBatch; ++u) hc[cp - 1][u] = cpy[min(base + tid + THREADS * u, n_entries - 1)];
        }
      }
#pragma unroll
      for (int cp = 1; cp < COPIES; ++cp)
#pragma unroll
        for (int u = 0; u < kBatch; ++u) hv[u] += hc[cp - 1][u];
    }
    if (a.use_marginal) {
#pragma unroll
      for (int u = 0; u < kBatch; ++u) hm[u] = a.HmPacked[min(base + tid + THREADS * u, n_entries - 1)];
    }
  };
  if (main_wg) loadBatch(0);
  if (decides) {
    __shared__ LmControl s_dec_out;
    __shared__ int s_dec_accept, s_dec_proceed;
    double t[4] = {0, 0, 0, 0};
    if (a.dec_groups) {
      // 64 group sums per scalar -> 8 sums of 8 -> one (fixed order: identical in every workgroup and from run to run)
      __shared__ double s_grp[4 * 8];
      if (tid < 32) {
        const int e = tid >> 3, j = tid & 7;
        // (all eight requested before the first is added: written as one running sum the compiler issued them one by one, each behind
        // the previous one's arrival — seven memory round trips in front of every decision of a two-stage / sharded window)
        const auto gsrc = glb(dec_scalars_p);
        double gv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) gv[k] = gsrc[4 * (8 * j + k) + e];
        asm volatile("" : "+v"(gv[0]), "+v"(gv[1]), "+v"(gv[2]), "+v"(gv[3]), "+v"(gv[4]), "+v"(gv[5]), "+v"(gv[6]), "+v"(gv[7]));
        double sacc = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) sacc += gv[k];
        s_grp[e * 8 + j] = sacc;
      }
      ldsBarrier();
      if (tid == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          double sacc = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) sacc += s_grp[e * 8 + j];
          t[e] = sacc;
        }
      }
    } else {
      t[0] = t_early[0];
      t[1] = t_early[1];
      t[2] = t_early[2];
      t[3] = t_early[3];
    }
    if (!cin.active) {  // the loop has ended: the control block is handed on unchanged
      if (tid == 0 && (ticketed ? ticket_raw == a.bs_ticket_base : blockIdx.x == 0)) *a.ctrl = cin;
      return;
    }
    SC_STAMP(8);
    if (tid == 0) {
      LmControl c;
      int accept = 0, proceed = 0;
      lmDecision(cin, t, a.dec_prm, c, accept, proceed);
      s_dec_out = c;
      s_dec_accept = accept;
      s_dec_proceed = proceed;
      s_vblock = ticketed ? ticket_raw - a.bs_ticket_base : blockIdx.x;
    }
    ldsBarrier();
    SC_STAMP(9);
    dec_accept = s_dec_accept;
    const unsigned vblock = s_vblock;
    main_wg = vblock == 0;
    if (vblock > 0) {
      // ---- landmark workgroups.  4 lanes per landmark (lane `sub` owns the frame slots sub, sub + 4, ...), THREADS / 4 landmarks per
      // pass = CPP chunks of 64; workgroup b takes the passes b - 1, b - 1 + W, ...
      constexpr int LP = THREADS / 4, CPP = LP / kSchurLandmarks;
      static_assert(CPP >= 1, "a pass covers whole chunks");
      const int W = gridDim.x - 1, n_pass = (a.dec_chunks + CPP - 1) / CPP;
      const int sub = tid & 3, grp = tid >> 2, l = grp & (kSchurLandmarks - 1), chunk_in_pass = grp / kSchurLandmarks;
      if (cin.pending) {
        // acceptStep / rejectStep of the landmarks (problem.hpp:364-402)
        for (int p = static_cast<int>(vblock) - 1; p < n_pass; p += W) {
          const int chunk = p * CPP + chunk_in_pass;
          if (chunk >= a.dec_chunks) continue;
          const SchurBlock &be = a.dec_table[chunk];
          const int i = be.offset + l;
          if (i >= be.n) continue;
#pragma unroll
          for (int h = 0; h < 4; ++h) {
            const int t2 = sub + 4 * h;
            if (t2 < F && be.status[t2] != nullptr && i < be.n_res[t2]) {
              if (dec_accept)
                be.status[t2][i] = be.cand[t2][i];
              else
                be.cand[t2][i] = be.status[t2][i];
            }
          }
          if (sub == 0) {
            if (dec_accept) be.idepth[i] += be.idepth_step[i];
            be.idepth_step[i] = 0;
          }
        }
      }
      // what workgroup 0 does with this decision: no step -> nothing to substitute
      if (!ticketed || !s_dec_proceed || s_dec_out.relin) return;
      // (a) the rows of the first kPre passes are requested now and wait in registers (12 frames / 50 000 landmarks: 255 workgroups x
      // 2 passes x 128 landmarks hold all of them)
      auto backSubstitute = [&](auto ns_tag) {
      constexpr int NS = decltype(ns_tag)::value;  // frame slots per lane: 2 up to 8 frames, 3 up to 12, 4 up to 16
      // passes whose rows wait in registers: 2 x 128 landmarks at 512 threads (1 at 13 - 16 frames: the compiler spilled with two),
      // 2 x 64 at 256 threads — with 4 the 256-thread kernel needed 255 VGPRs + 38 AGPRs and its solving path slowed down (the
      // factorisation 8.2 -> 8.8 us at 7 frames: accumulator-register moves in the pivot loop); with 2 it is 217 VGPRs, two workgroups
      // per compute unit, and every window up to 8 frames / 65 000 landmarks still has all rows prefetched
      constexpr int kPre = THREADS >= 512 ? (NS >= 4 ? 1 : 2) : 2;
      double rows[kPre][NS][kBlk], bdv[kPre], ihv[kPre];
      hbm_f64 *dstp[kPre];
    #pragma unroll
      for (int u = 0; u < kPre; ++u) {
        dstp[u] = nullptr;
        bdv[u] = ihv[u] = 0;
    #pragma unroll
        for (int q = 0; q < NS; ++q)
    #pragma unroll
          for (int c = 0; c < kBlk; ++c) rows[u][q][c] = 0;
      }
      auto fetch = [&](int p, double (&rw)[NS][kBlk], double &bd, double &ih, hbm_f64 *&dst) {
        dst = nullptr;
        const int chunk = p * CPP + chunk_in_pass;
        if (chunk >= a.dec_chunks) return;
        const SchurBlock &be = a.dec_table[chunk];
        const int i = be.offset + l;
        if (i >= be.n) return;
        const uint8_t flg = be.flags[i];
        if (flg & (kFlagMarginalized | kFlagIllConditioned)) return;
        const size_t plane = ublkPlane(be.cap);
        const hbm_f64 *base = be.ublk + static_cast<size_t>(a.bs_parity) * kMaxFrames * plane + static_cast<size_t>(i) * kUblk;
    #pragma unroll
        for (int q = 0; q < NS; ++q) {
          int t = sub + 4 * q;
          asm volatile("" : "+v"(t));  // (opaque: otherwise the per-slot masks 1 << t are hoisted out of the pass loop and held — two of them spilled)
          if (t < F && (t == be.r || ((be.conn_mask >> t) & 1u))) {
    #pragma unroll
            for (int c = 0; c < kBlk; ++c) rw[q][c] = base[t * plane + c];
          }
        }
        bd = be.b_d[i];
        ih = be.inv_hdd[i];
        dst = be.idepth_step + i;
      };
    #pragma unroll
      for (int u = 0; u < kPre; ++u) {
        const int p = static_cast<int>(vblock) - 1 + u * W;
        if (p < n_pass) fetch(p, rows[u], bdv[u], ihv[u], dstp[u]);
      }
      // (b) wait for the solving workgroup's step: thread k polls slot k until it holds a value.  The solver drew its ticket before
      // this workgroup and waits for nobody; the bound only turns an accident into a loud failure instead of a hang.  The step goes
      // through LDS (this workgroup's share of the launch's dynamic allocation is otherwise unused): every lane reads its slots' eight
      // values per pass instead of keeping NS x 8 more doubles in registers next to the rows
      double *stp = reinterpret_cast<double *>(smem_raw);  // [kBlk * kMaxFrames], zero beyond K
      __shared__ int s_timed_out;
      if (tid == 0) s_timed_out = 0;
      ldsBarrier();
      if (tid < kBlk * kMaxFrames) {
        double v = 0;
        if (tid < K) {
          const long long t0 = wall_clock64();
          for (;;) {
            v = __hip_atomic_load(a.bs_hand + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__double_as_longlong(v) != __double_as_longlong(kHandOverSentinel())) break;
            __builtin_amdgcn_s_sleep(24);  // (255 workgroups poll the same 12 lines: a poll every ~0.7 us keeps that queue short)
            if (wall_clock64() - t0 > 200000000ll) {  // 2 s: report and leave (SolveCombArgs::bs_fault)
              s_timed_out = 1;
              if (a.bs_fault) __hip_atomic_store(a.bs_fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              break;
            }
          }
        }
        stp[tid] = v;
      }
      __syncthreads();
      if (s_timed_out) return;  // (uniform: read after the barrier)
      const double damp = 1.0 / (1.0 + s_dec_out.lambda);
      auto finish = [&](const double (&rw)[NS][kBlk], double bd, double ih, hbm_f64 *dst) {
        double d = 0;
    #pragma unroll
        for (int q = 0; q < NS; ++q) {
          const double *sq = stp + kBlk * (sub + 4 * q);
    #pragma unroll
          for (int c = 0; c < kBlk; ++c) d += rw[q][c] * sq[c];
        }
        d += dppMove<0xB1>(d);  // the landmark's four lanes: quad_perm [1,0,3,2], [2,3,0,1]
        d += dppMove<0x4E>(d);
        if (sub == 0 && dst) *dst = -((bd - d) * damp * ih);
      };
    #pragma unroll
      for (int u = 0; u < kPre; ++u) finish(rows[u], bdv[u], ihv[u], dstp[u]);
      // (c) passes beyond what the registers hold (very large windows): fetched behind the step
      for (int p = static_cast<int>(vblock) - 1 + kPre * W; p < n_pass; p += W) {
        double rw[NS][kBlk], bd = 0, ih = 0;
        hbm_f64 *dst = nullptr;
    #pragma unroll
        for (int q = 0; q < NS; ++q)
    #pragma unroll
          for (int c = 0; c < kBlk; ++c) rw[q][c] = 0;
        fetch(p, rw, bd, ih, dst);
        finish(rw, bd, ih, dst);
      }
      };
      if (THREADS < 512)
        backSubstitute(std::integral_constant<int, 2>{});
      else if (F <= 12)
        backSubstitute(std::integral_constant<int, 3>{});
      else
        backSubstitute(std::integral_constant<int, 4>{});
      return;
    }
    // workgroup 0: the frame states and the outgoing control block, then the solve at the decided state
    if (cin.pending && tid < K) {
      if (dec_accept) {
        dec_eps += dec_step;
        st_p->eps[tid >> 3][tid & 7] = dec_eps;
      }
      st_p->step[tid >> 3][tid & 7] = 0;
    }
    if (tid == 64) *a.ctrl = s_dec_out;  // (a lane of wave 1, which has nothing else to do here: wave 0 stores the decided states and requests the pair inputs)
    if (!s_dec_proceed) return;
    c_active = 1;
    c_relin = s_dec_out.relin;
    lam = s_dec_out.lambda;
  } else if (a.ctrl) {
    c_active = a.ctrl->active;
    c_relin = a.ctrl->relin;
    lam = a.ctrl->lambda;
  }
  if (decides) eps_c = dec_eps;  // (decided here: the accepted state is already in registers)
  SC_STAMP(0);
  if (kStamps && a.dbg_stamps && tid == 0) a.dbg_stamps[7] = sc_t_entry;
  // opaque to the optimiser: stops it from testing these loaded flags (and waiting for them) above the loads
  // (with tickets the pair inputs were requested a moment ago, behind the decision: they must NOT be waited for here — they are needed at
  // the kernel's tail and land under the assembly and the factorisation)
  asm volatile("" : "+v"(fixed_c), "+v"(tomarg_c), "+v"(c_active), "+v"(c_relin));
  int prior_kind = 0;  // 0 none, 1 fixed frame, 2 affine brightness (evaluateLinearSystemPrior, problem.hpp:39-62)
  double pd_c = 0;
  if (tid < K) {
    if (!tomarg_c) prior_kind = fixed_c ? 1 : ((tid & 7) >= 6 ? 2 : 0);
    pd_c = prior_kind == 1 ? a.fixed_reg : (prior_kind == 2 ? ((tid & 7) == 6 ? a.affine_reg[0] : a.affine_reg[1]) : 0.0);  // (no dynamic index: that is a vector load of the argument block)
    xs[tid] = eps_c;
    epsl[tid] = eps_c;
    Linv[tid] = pd_c;  // prior diagonal, parked in the Linv area until the factorisation starts
    if ((tid & 7) >= 6) ab0l[2 * (tid >> 3) + (tid & 7) - 6] = ab0_c;
  }
  // (an LDS-only barrier: __syncthreads() also waits for every outstanding global access — here the pair inputs just requested and the
  // stores of the decided frame states and the control block, a memory round trip of about 1 us in front of the assembly)
  ldsBarrier();  // keeps every load above; prior diagonal / eps visible
  SC_STAMP(10);
  if (!c_active || c_relin) return;
  const double *prior_diag = Linv;
  // Where entry e = base + tid + THREADS * u goes: its block index (e >> 6) is the same for the whole wave (THREADS and base are multiples
  // of 64) and advances by THREADS / 64 per u; the position inside the block depends on the lane alone.  The block's row and column are
  // therefore walked in scalar registers, one decode per batch (until round 5: a float square root and two dozen vector instructions per
  // entry — 1.2 of the 1.9 us between the decision and the factorisation at 7 frames).
  const int in_row = (tid >> 3) & 7, in_col = tid & 7;
  // (an entry's LDS address = the block's offset — scalar — + the thread's offset inside a block — computed once; the divergent tests for the
  // diagonal and the upper triangle exist only in diagonal blocks, behind a scalar branch.  The first scalar-walk version still spent ~45
  // instructions per entry — row x ld as a vector integer multiply, two exec-mask regions — i.e. 1 us per batch of 8 on an
  // instruction-bound workgroup.)
  double *const A_thr = A + in_row * ld + in_col;
  auto storeBatch = [&](int base) {
    int blk = __builtin_amdgcn_readfirstlane((base + tid) >> 6);
    int bi, bj;
    combBlockDecode(blk, bi, bj);
    bi = __builtin_amdgcn_readfirstlane(bi);
    bj = __builtin_amdgcn_readfirstlane(bj);
    const int n_blocks = n_entries >> 6;
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      if (blk < n_blocks) {  // (wave-uniform)
        const int blk_off = __builtin_amdgcn_readfirstlane(8 * bi * ld + 8 * bj);
        const double v = hv[u] + hm[u];
        if (bi != bj) {  // (wave-uniform)
          A_thr[blk_off] = v;
        } else if (in_col <= in_row) {  // diagonal block: the lower triangle is stored, the diagonal takes the prior and feeds the guard
          double vd = v;
          if (in_row == in_col) {
            const int row = 8 * bi + in_row;
            vd += prior_diag[row] * (1.0 + lam);  // the prior's diagonal takes the damping too (problem.hpp:347-349)
            // The reference solves the Jacobi-scaled system p H p, p = 1/sqrt(diag + 10) (normal_linear_system.cpp:10-16,52-59).  A
            // Cholesky factorisation is invariant under symmetric diagonal scaling, so only the zero-pivot guard refers to it
            pv[row] = vd + 10.0;
          }
          A_thr[blk_off] = vd;
        }
      }
      // next block of this wave: THREADS / 64 further along the packed lower triangle
      blk += THREADS / 64;
      bj += THREADS / 64;
      while (bj > bi) {
        bj -= bi + 1;
        ++bi;
      }
    }
  };
  storeBatch(0);
  for (int base = kBatch * THREADS; base < n_entries; base += kBatch * THREADS) {
    loadBatch(base);
    storeBatch(base);
  }
  SC_STAMP(11);
  if (tid < K) {
    double v = rhs_c;
    if (prior_kind == 1)
      v += a.fixed_reg * eps_c;
    else if (prior_kind == 2)
      v += ((tid & 7) == 6 ? a.affine_reg[0] : a.affine_reg[1]) * (ab0_c + eps_c);
    if (a.use_marginal) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += a.Hm[tid * K + k] * xs[k];
      v += bm_c + s;
    }
    A[K * ld + tid] = v;
  }
  if (tid == 0) A[K * ld + K] = 0;
  ldsBarrier();
  SC_STAMP(1);
  // The pair inputs (needed at the kernel's tail) are requested HERE: the factorisation and the back-substitution below touch LDS only, so
  // nothing waits for these loads before the tail.  Requested in the head — at either of the two places they have been — something always
  // did: the copies that unify two request sites' registers where the paths join, the test of P.valid (which the compiler evaluates as soon
  // as the word is loaded unless it is pinned at its use), a kernel-argument array read through a vector load with vmcnt(0) behind it —
  // each a memory round trip on the solving workgroup's path.
  if (main_wg) requestPairInputs();

  // (Measured alternatives, all slower on this part — scripts/probes/bcast_probe.hip, dbg_stamps.py: a single barrier per block
  // step with the panel wave applying the previous panel to its own column: 10.2 us against 9.2 us for the 7-frame window; L D L^T
  // with v_rcp_f64 pivots: 11.0 us; the diagonal block eliminated redundantly per lane from LDS broadcast reads: 11.5 us.  The
  // panel wave is bound by its instruction count (~4.7 cycles per instruction, v_readlane ~8) and by the LDS instructions a lone
  // wave can issue, not by dependent latency.)
  // ---- blocked Cholesky A = L L^T on the augmented (K+1) x (K+1) matrix: the last row of L becomes y^T = (L^-1 b)^T.
  // Look-ahead schedule, one barrier per 8x8 frame block: wave 0 ("panel wave") brings block column kb+1 up to date with
  // panel kb, factors its diagonal block in registers and solves the panel below it, WHILE waves 1..3 apply panel kb to
  // the rest of the trailing matrix (columns >= kb+2).  The sequential factor chain is thus off the other waves' path.
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
    // inverse square root of a pivot: the hardware estimate (v_rsq_f64, 2^-23 relative, straight on the f64 value: an f32 seed costs two
    // conversions on the dependent chain) + two Newton steps
    auto rsqrtRefined = [](double d) {
      const double hd = 0.5 * d;
      double inv = __builtin_amdgcn_rsq(d);
      inv = fma(inv, fma(-hd * inv, inv, 0.5), inv);
      inv = fma(inv, fma(-hd * inv, inv, 0.5), inv);
      return inv;
    };
#ifndef DSOPP_HIP_PAIRED_PIVOTS
#pragma unroll
    for (int k = 0; k < kBlk; ++k) {
#ifdef DSOPP_HIP_MARKS
      asm volatile("; ##PANEL_PIVOT");  // ISA reading aid (scripts/panel_pivot_isa.sh): where a pivot's instructions begin
#endif
      const double d = readLane(c[k], k);
      // pivots whose Jacobi-scaled value d / (diag + 10) is below 1e-30 are treated as zero, as a rank-revealing factorisation would
      const bool okp = d > guard[k];
      double inv = rsqrtRefined(d);
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
#else
    // (-DDSOPP_HIP_PAIRED_PIVOTS: built, parity-green, measured and NOT the default — round 5.)  Pivots eliminated in PAIRS, on the
    // assumption that the panel wave's time is the dependent chain pivot -> 1/sqrt (estimate + two Newton steps: 7 dependent operations)
    // -> column -> broadcast -> update -> next pivot.  It is not: the wave issues an instruction every ~4.7 cycles (a v_readlane ~8)
    // whether or not it depends on the previous one (scripts/probes/clock_probe.hip: 4.70 cycles per dependent f64 fma at 2.33 - 2.40
    // GHz), and this form has more instructions (the extra broadcast of b, a cc - b^2, the selects): stamps, factor + panel summed over the 7
    // block steps of a 7-frame window 7.8 - 8.3 us against 6.8 - 7.0 us; the solve launch in the loop 19.45 against 18.29 us
    // (profiles/r05/paired_pivots_ab.txt).  For the pivots k, k + 1 both inverse square roots can be started
    // from values that are known at the same time: with a = c_kk, b = c_(k+1)k, cc = c_(k+1)(k+1) the second pivot is
    // d2 = cc - b^2 / a = (a cc - b^2) / a, so 1/sqrt(d2) = sqrt(a) / sqrt(a cc - b^2) = (a / sqrt(a)) * rsqrt(a cc - b^2): the two
    // refinements run side by side and the chain is walked four times per block instead of eight.  The updates themselves are those of
    // the pivot-by-pivot loop, in the same order (l_(k+1)k = b / sqrt(a) is formed from the broadcast b instead of being broadcast).
    // Zero-pivot guards as before: a pivot whose Jacobi-scaled value d / (diag + 10) is below 1e-30 is treated as zero; for the second
    // one the test d2 > g reads a cc - b^2 > g a, and behind a zero first pivot (nothing is eliminated) d2 = cc.
    double ljm[kBlk][kBlk];  // l_jk of the diagonal block (uniform), for the rows beyond 64
#pragma unroll
    for (int k = 0; k < kBlk; k += 2) {
      const double a0 = readLane(c[k], k), b0 = readLane(c[k], k + 1), cc0 = readLane(c[k + 1], k + 1);
      const bool ok1 = a0 > guard[k];
      const double s0 = fma(a0, cc0, -(b0 * b0));
      const bool ok2 = ok1 ? s0 > guard[k + 1] * a0 : cc0 > guard[k + 1];
      double inv1 = rsqrtRefined(a0);
      double r2 = rsqrtRefined(ok1 ? s0 : cc0);
      inv1 = ok1 ? inv1 : 0.0;
      double inv2 = ok1 ? r2 * (a0 * inv1) : r2;
      inv2 = ok2 ? inv2 : 0.0;
      invd[k] = inv1;
      invd[k + 1] = inv2;
      const double l0 = c[k] * inv1;       // lane k: sqrt(a); lanes i > k: l_ik
      const double l10 = b0 * inv1;        // l_(k+1)k (uniform)
      ljm[k][k + 1] = l10;
      c[k] = l0;
      c[k + 1] -= l0 * l10;
      const double l1 = c[k + 1] * inv2;   // lane k + 1: sqrt(d2); lanes i > k + 1: l_i(k+1)
      c[k + 1] = l1;
#pragma unroll
      for (int j = k + 2; j < kBlk; ++j) {
        const double lj0 = readLane(l0, j), lj1 = readLane(l1, j);
        ljm[k][j] = lj0;
        ljm[k + 1][j] = lj1;
        c[j] -= l0 * lj0;
        c[j] -= l1 * lj1;
      }
    }
#pragma unroll
    for (int k = 0; k < kBlk; ++k)
#pragma unroll
      for (int j = k + 1; j < kBlk; ++j) lj[e++] = ljm[k][j];
#endif
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
  // ---- round 6 (-DDSOPP_HIP_DPP_PANEL: built, parity-green on 74 GPU tests, measured and NOT the default): the same elimination with the
  // row-k factors broadcast by the VALU's own cross-lane path instead of v_readlane.  Result: C1 37.1 - 37.2 us per iteration against 36.8,
  // the solve launch alone 16.47 against 16.28 us (12 KF / 50 k: 28.5 against 27.6; 15 KF / 5 k: 38.6 against 37.2) — profiles/r06/
  // solve_panel_ab.txt.  The second pass that 32 rows per pass force on the first block columns, the two wait states in front of every DPP
  // read and the serialisation of the inline assembly cost what the 84 v_readlane pairs per block did; the panel wave's LDS traffic (8 loads
  // + 8 stores per lane and pass) is untouched either way.
  // The 64-bit VALU operations take ONE DPP control, row_newbcast:k — lane k of every row of 16 lanes is the source for that row
  // (scripts/probes/dpp64_probe.hip) — so `c_j -= l l_jk` is ONE instruction (v_fmac_f64_dpp) where the loop above needs two v_readlane
  // (~8 cycles each on the lone panel wave) and an fma.  The price is the layout: a row of 16 lanes can only hear its own lanes, so each of
  // the wave's four rows carries the diagonal block in its lanes 0 .. 7 (eliminated four times over, redundantly — same instructions) and
  // eight panel rows in its lanes 8 .. 15: 32 panel rows per pass instead of 64, further passes substitute with the finished block's
  // factors (lane j, register k = l_jk) through the same broadcast.  Arithmetic per entry as above, operation for operation.
  auto factorAndPanelDpp = [&](int kb) {
    const int k0 = kb * kBlk, k1 = k0 + kBlk;
    const int sub = lane & 15, rowgrp = lane >> 4;
    const bool diag_lane = sub < kBlk;
    const int prow = rowgrp * kBlk + (sub - kBlk);  // panel row of this lane within a pass (lanes 8 .. 15 of every row of 16)
    double c[kBlk], invd[kBlk];
    {
      const int row = diag_lane ? k0 + sub : k1 + prow;
      const double *src = A + (row < N ? row : k0) * ld + k0;
#pragma unroll
      for (int j = 0; j < kBlk; ++j) c[j] = src[j];
    }
    double guard[kBlk];  // zero-pivot thresholds, fetched before the pivot chain starts
#pragma unroll
    for (int k = 0; k < kBlk; ++k) guard[k] = 1e-30 * pv[min(k0 + k, K - 1)];
    auto rsqrtRefined = [](double d) {
      const double hd = 0.5 * d;
      double inv = __builtin_amdgcn_rsq(d);
      inv = fma(inv, fma(-hd * inv, inv, 0.5), inv);
      inv = fma(inv, fma(-hd * inv, inv, 0.5), inv);
      return inv;
    };
    // (s_nop 1: a VALU result read through DPP needs two wait states, which the assembler does not insert inside inline assembly)
#define DSOPP_BCAST(dst, src, k) asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(src))
#define DSOPP_FMAC_BCAST(acc, src_bcast, mul, k) \
  asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src_bcast), "v"(mul))
#define DSOPP_PIVOT(k)                                                                                   \
  {                                                                                                      \
    double d;                                                                                            \
    DSOPP_BCAST(d, c[k], k);                                                                             \
    const bool okp = d > guard[k];                                                                       \
    double inv = rsqrtRefined(d);                                                                        \
    inv = okp ? inv : 0.0;                                                                               \
    invd[k] = inv;                                                                                       \
    l = c[k] * inv; /* lane k of a row: sqrt(d); diagonal lanes i > k and panel lanes: l_ik */           \
    c[k] = l;                                                                                            \
    nl = -l;                                                                                             \
  }
    double l, nl;
    DSOPP_PIVOT(0)
    DSOPP_FMAC_BCAST(c[1], l, nl, 1); DSOPP_FMAC_BCAST(c[2], l, nl, 2); DSOPP_FMAC_BCAST(c[3], l, nl, 3); DSOPP_FMAC_BCAST(c[4], l, nl, 4);
    DSOPP_FMAC_BCAST(c[5], l, nl, 5); DSOPP_FMAC_BCAST(c[6], l, nl, 6); DSOPP_FMAC_BCAST(c[7], l, nl, 7);
    DSOPP_PIVOT(1)
    DSOPP_FMAC_BCAST(c[2], l, nl, 2); DSOPP_FMAC_BCAST(c[3], l, nl, 3); DSOPP_FMAC_BCAST(c[4], l, nl, 4); DSOPP_FMAC_BCAST(c[5], l, nl, 5);
    DSOPP_FMAC_BCAST(c[6], l, nl, 6); DSOPP_FMAC_BCAST(c[7], l, nl, 7);
    DSOPP_PIVOT(2)
    DSOPP_FMAC_BCAST(c[3], l, nl, 3); DSOPP_FMAC_BCAST(c[4], l, nl, 4); DSOPP_FMAC_BCAST(c[5], l, nl, 5); DSOPP_FMAC_BCAST(c[6], l, nl, 6);
    DSOPP_FMAC_BCAST(c[7], l, nl, 7);
    DSOPP_PIVOT(3)
    DSOPP_FMAC_BCAST(c[4], l, nl, 4); DSOPP_FMAC_BCAST(c[5], l, nl, 5); DSOPP_FMAC_BCAST(c[6], l, nl, 6); DSOPP_FMAC_BCAST(c[7], l, nl, 7);
    DSOPP_PIVOT(4)
    DSOPP_FMAC_BCAST(c[5], l, nl, 5); DSOPP_FMAC_BCAST(c[6], l, nl, 6); DSOPP_FMAC_BCAST(c[7], l, nl, 7);
    DSOPP_PIVOT(5)
    DSOPP_FMAC_BCAST(c[6], l, nl, 6); DSOPP_FMAC_BCAST(c[7], l, nl, 7);
    DSOPP_PIVOT(6)
    DSOPP_FMAC_BCAST(c[7], l, nl, 7);
    DSOPP_PIVOT(7)
    {
      const int row = diag_lane ? k0 + sub : k1 + prow;
      if (row < N && (!diag_lane || rowgrp == 0)) {
        double *dst = A + row * ld + k0;
#pragma unroll
        for (int j = 0; j < kBlk; ++j)
          if (!diag_lane || j <= sub) dst[j] = c[j];  // the diagonal block keeps its lower triangle only
      }
    }
    // panel rows beyond the first 32 of this block column: substitution with the finished block's factors (lane j of a row, register k)
    for (int base = k1 + 32; base < N; base += 32) {
      const int r2 = base + prow;
      const bool live = !diag_lane && r2 < N;
      double v[kBlk];
      {
        const double *src = A + (live ? r2 : k0) * ld + k0;
#pragma unroll
        for (int j = 0; j < kBlk; ++j) v[j] = src[j];
      }
      double nv;
#define DSOPP_SUBST(k) \
  v[k] *= invd[k];     \
  nv = -v[k];
      DSOPP_SUBST(0)
      DSOPP_FMAC_BCAST(v[1], c[0], nv, 1); DSOPP_FMAC_BCAST(v[2], c[0], nv, 2); DSOPP_FMAC_BCAST(v[3], c[0], nv, 3); DSOPP_FMAC_BCAST(v[4], c[0], nv, 4);
      DSOPP_FMAC_BCAST(v[5], c[0], nv, 5); DSOPP_FMAC_BCAST(v[6], c[0], nv, 6); DSOPP_FMAC_BCAST(v[7], c[0], nv, 7);
      DSOPP_SUBST(1)
      DSOPP_FMAC_BCAST(v[2], c[1], nv, 2); DSOPP_FMAC_BCAST(v[3], c[1], nv, 3); DSOPP_FMAC_BCAST(v[4], c[1], nv, 4); DSOPP_FMAC_BCAST(v[5], c[1], nv, 5);
      DSOPP_FMAC_BCAST(v[6], c[1], nv, 6); DSOPP_FMAC_BCAST(v[7], c[1], nv, 7);
      DSOPP_SUBST(2)
      DSOPP_FMAC_BCAST(v[3], c[2], nv, 3); DSOPP_FMAC_BCAST(v[4], c[2], nv, 4); DSOPP_FMAC_BCAST(v[5], c[2], nv, 5); DSOPP_FMAC_BCAST(v[6], c[2], nv, 6);
      DSOPP_FMAC_BCAST(v[7], c[2], nv, 7);
      DSOPP_SUBST(3)
      DSOPP_FMAC_BCAST(v[4], c[3], nv, 4); DSOPP_FMAC_BCAST(v[5], c[3], nv, 5); DSOPP_FMAC_BCAST(v[6], c[3], nv, 6); DSOPP_FMAC_BCAST(v[7], c[3], nv, 7);
      DSOPP_SUBST(4)
      DSOPP_FMAC_BCAST(v[5], c[4], nv, 5); DSOPP_FMAC_BCAST(v[6], c[4], nv, 6); DSOPP_FMAC_BCAST(v[7], c[4], nv, 7);
      DSOPP_SUBST(5)
      DSOPP_FMAC_BCAST(v[6], c[5], nv, 6); DSOPP_FMAC_BCAST(v[7], c[5], nv, 7);
      DSOPP_SUBST(6)
      DSOPP_FMAC_BCAST(v[7], c[6], nv, 7);
      DSOPP_SUBST(7)
#undef DSOPP_SUBST
      if (live) {
        double *dst = A + r2 * ld + k0;
#pragma unroll
        for (int j = 0; j < kBlk; ++j) dst[j] = v[j];
      }
    }
#undef DSOPP_PIVOT
#undef DSOPP_FMAC_BCAST
#undef DSOPP_BCAST
    if (lane == 0) {
#pragma unroll
      for (int cidx = 0; cidx < kBlk; ++cidx) Linv[kb * 36 + lowIdx(cidx, cidx)] = invd[cidx];  // diagonal of the inverse; completed below
    }
  };
  (void)factorAndPanelDpp;
  // (tuning aid, stamps build: where the panel wave's time goes over the block steps — slots 12 column update, 13 its barrier, 14 factor +
  // panel, 15 the barrier behind it)
  long long cs_acc[4] = {0, 0, 0, 0}, cs_t = (kStamps && a.dbg_stamps) ? wall_clock64() : 0;
  auto csMark = [&](int slot) {
    if (kStamps && a.dbg_stamps) {
      const long long now = wall_clock64();
      cs_acc[slot] += now - cs_t;
      cs_t = now;
    }
  };
#ifdef DSOPP_HIP_DPP_PANEL
  if (wave == 0) factorAndPanelDpp(0);
#else
  if (wave == 0) factorAndPanel(0);
#endif
  ldsBarrier();
  csMark(2);
  for (int kb = 0; kb < F; ++kb) {
    const int k0 = kb * kBlk, k1 = k0 + kBlk, k2 = k1 + kBlk;
    if (kb + 1 < F) {
      // all waves: block column kb+1 (rows k1 .. N-1, columns k1 .. k1+7) -= panel kb contribution (one element per thread)
      const int n_el = (N - k1) * kBlk;
      for (int e = tid; e < n_el; e += THREADS) {
        const int row = k1 + (e >> 3), col = k1 + (e & 7);
        if (col > row) continue;
        const double *li = A + row * ld + k0, *lj = A + col * ld + k0;
        double sacc = 0;
#pragma unroll
        for (int c = 0; c < kBlk; ++c) sacc += li[c] * lj[c];
        A[row * ld + col] -= sacc;
      }
    }
    csMark(0);
    ldsBarrier();
    csMark(1);
    if (wave == 0) {
#ifdef DSOPP_HIP_DPP_PANEL
      if (kb + 1 < F) factorAndPanelDpp(kb + 1);
#else
      if (kb + 1 < F) factorAndPanel(kb + 1);
#endif
      csMark(2);
    } else {
      // trailing update of columns >= k2 with panel kb: A_ij -= sum_c L_ic L_jc  (the other waves as a 12 x 16 / 28 x 16 tile)
      const int t = tid - 64, tr = t >> 4, tc = t & 15;
      for (int row = k2 + tr; row < N; row += (THREADS - 64) / 16) {
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
    ldsBarrier();
    csMark(3);
  }
  if (kStamps && a.dbg_stamps && tid == 0) {
    for (int q = 0; q < 4; ++q) a.dbg_stamps[12 + q] = cs_acc[q];
  }
  SC_STAMP(2);
  // what only the kernel's tail reads (pair-constant refresh, prior energy of the candidate) is requested HERE, to land under the
  // back-substitution: requested at the head, these 24 doubles per thread were held — or spilled and reloaded — across the factorisation
  if (ticketed) requestTailInputs();
  // ---- back substitution x = L^-T y (y = row K of L), column-oriented on one wave: lane j carries y_j (and y_{j+64});
  // going down from k = K-1, x_k = y_k / L_kk is broadcast with v_readlane and every lane j < k takes y_j -= L_kj x_k.
  // 4-7 instructions per unknown, no LDS round trip or barrier inside the chain (L_kj is prefetched a frame block ahead).
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
      // Unknowns 64 .. K-1 (windows of more than 8 frames) are carried by y1: while they are eliminated (frame blocks >= 8) every
      // step updates both halves; from block 7 down nothing above lane 63 is read again, so the loop is the one-register loop of
      // a small window (no second load, multiply, broadcast pair or update per step: 9.0 -> about 5 us at 12 frames).
      auto sweepBlocks = [&](auto upper_tag, int kb_from, int kb_to) {
        constexpr bool UPPER = decltype(upper_tag)::value;
        for (int kb = kb_from; kb >= kb_to; --kb) {
          if (kb > 0) {
#pragma unroll
            for (int c = 0; c < kBlk; ++c) {
              n0[c] = A[((kb - 1) * kBlk + c) * ld + j0];  // lanes beyond the row read into the next row: in bounds, never used
              if (UPPER) n1[c] = j1 < K ? A[((kb - 1) * kBlk + c) * ld + j1] : 0.0;
            }
          }
          double xo[kBlk];
#pragma unroll
          for (int c = kBlk - 1; c >= 0; --c) {
            const int k = kb * kBlk + c;
            const double xk = UPPER ? readLane(y1 * gi1, k & 63) : readLane(y0 * gi0, k);
            xo[c] = xk;
            y0 -= g0[c] * xk;
            if (UPPER) y1 -= g1[c] * xk;
          }
          if (lane == 0) {
#pragma unroll
            for (int c = 0; c < kBlk; ++c) xs[kb * kBlk + c] = xo[c];
          }
#pragma unroll
          for (int c = 0; c < kBlk; ++c) {
            g0[c] = n0[c];
            if (UPPER) g1[c] = n1[c];
          }
        }
      };
      loadBlock(F - 1, g0, g1);
      if (TWO) {
        sweepBlocks(std::true_type{}, F - 1, 8);
        sweepBlocks(std::false_type{}, 7, 0);
      } else {
        sweepBlocks(std::false_type{}, F - 1, 0);
      }
    };
    // (Measured and dropped: back-substitution by frame blocks with inverted diagonal blocks, x_blk = W^T y_blk then y -= L_blk^T x_blk,
    // which halves the dependent chain but doubles the broadcasts: 3.2 us against 2.6 us.  A v_readlane pair costs as much as three
    // dependent f64 FMAs here (scripts/probes/bcast_probe.hip), so the count of broadcasts decides, not the chain length.)
    if (K > 64)
      run(std::true_type{});
    else
      run(std::false_type{});
    SC_STAMP(6);
  }
  __syncthreads();
  if (tid < K) {
    const double x = xs[tid];
    stpl[tid] = -x;
    // (a device-scope store: written through to where the other XCDs read — no flag, no fence: the slot's sentinel gives way to the value)
    if (a.bs_hand) __hip_atomic_store(a.bs_hand + tid, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the waiting workgroups' copy
    a.step[tid] = x;
    st_p->step[tid >> 3][tid & 7] = -x;  // problem.hpp:353-357
  }
  SC_STAMP(3);
  // prior + marginal energy at the candidate state x = eps + step (calculateEnergy, problem.hpp:293-312) and the frame part of the
  // norms acceptStep reports for this candidate (problem.hpp:366-388; read by the next decision from the control block): one
  // entry per thread of the waves behind wave 0, which meanwhile refreshes the pair constants
  double part = 0, nstate = 0, nstep = 0;
  auto priorMath = [&] {
    if (!a.ctrl || !prior_thread) return;
    const double xc = epsl[pc] + stpl[pc];
    if (a.use_marginal) {
      double sacc = 0;
      for (int k = 0; k < K; ++k) sacc += a.Hm[pc * K + k] * (epsl[k] + stpl[k]);
      part += bm_p * xc + 0.5 * xc * sacc;
    }
    if ((pc & 7) >= 6) {
      const double ab = ab0_p + xc;
      part += 0.5 * ab * ((pc & 7) == 6 ? a.affine_reg[0] : a.affine_reg[1]) * ab;
    }
    nstate = epsl[pc] * epsl[pc] + ((pc & 7) >= 6 ? ab0_p * ab0_p : 0.0);
    nstep = stpl[pc] * stpl[pc];
  };
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
    asm volatile("" : "+v"(pp.valid));  // (tested here, not where it was loaded)
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
    priorMath();  // (waves 1..: beside the pair constants of wave 0)
  } else {
    ldsBarrier();  // the new step is visible
    priorMath();
  }
  // (without first-estimate Jacobians every pair constant moves with the state: the host launches pairSetupKernel behind this
  // kernel, as it does behind assembleSolveKernel)
  SC_STAMP(4);
  if (a.ctrl && K <= 64) {
    // windows of up to 8 keyframes: the K prior threads are ONE wave (wave 1), which sums and writes the control block on its own while
    // wave 0 is still refreshing pair constants — no barrier, nothing behind the refresh on the workgroup's critical path
    // (bitwise the sums of the general form below: the other waves contribute exact zeros)
    // (wave 1 alone: the other waves hold exact zeros — and wave 0, which has just finished the pair constants, would spend another
    // 0.45 us of the launch's tail on three wave sums of zeros: it is the last wave to finish)
    if (wave == 1) {
      part = waveSum(part);
      nstate = waveSum(nstate);
      nstep = waveSum(nstep);
      if (tid == 64) {
        a.ctrl->cand_prior = a.energy_marginalized + part;
        a.ctrl->frame_state_sq = nstate;
        a.ctrl->frame_step_sq = nstep;
        a.ctrl->pending = 1;
      }
    }
  } else if (a.ctrl) {
    part = waveSum(part);
    nstate = waveSum(nstate);
    nstep = waveSum(nstep);
    ldsBarrier();  // E (in A) fully consumed before the scratch below is written; stpl visible
    if ((tid & 63) == 0) {
      xs[tid >> 6] = part;
      xs[THREADS / 64 + (tid >> 6)] = nstate;
      xs[2 * (THREADS / 64) + (tid >> 6)] = nstep;
    }
    ldsBarrier();
    // (thread 64 — the thread that stored the whole control block after the decision, line ~505: two stores of one thread to the same words
    // stay in order; with thread 0 here the struct store of wave 1 could land behind these fields, the barriers in between wait for LDS only)
    if (tid == 64) {
      constexpr int kWaves = THREADS / 64;  // three groups of kWaves wave sums in xs
      double total = a.energy_marginalized, s_state = 0, s_step = 0;
      for (int w = 0; w < kWaves; ++w) {
        total += xs[w];
        s_state += xs[kWaves + w];
        s_step += xs[2 * kWaves + w];
      }
      a.ctrl->cand_prior = total;
      a.ctrl->frame_state_sq = s_state;
      a.ctrl->frame_step_sq = s_step;
      a.ctrl->pending = 1;
    }
  }
  SC_STAMP(5);
}

}  // namespace dsopp_hip
