// Device-side data model of the sliding window (DESIGN.md §Data layout in HBM).
#pragma once
#include <cstdint>

#include "pyramid.hpp"
#include "se3_math.hpp"

namespace dsopp_hip {

constexpr int kMaxFrames = DSOPP_HIP_MAX_FRAMES;
constexpr int kBlk = DSOPP_HIP_BLOCK_SIZE;  // 8 = 6 pose + 2 affine
constexpr int kPat = DSOPP_HIP_PATTERN_SIZE;
constexpr int kUblk = 10;                   // per (landmark, slot): u[8], hdd, bd
constexpr int kUblkPad = 272;               // doubles between two slot planes of the Schur rows (see ublkPlane)
/** Plane stride (in doubles) of the Schur rows ublk[parity][slot][landmark][kUblk].  The landmark capacity is a power of two, so
 *  un-padded planes start at multiples of cap x 80 bytes (655 360 B at 8192 landmarks): the 8..12 slot planes a landmark's lanes
 *  touch together then fall onto the same memory channels.  2 176 bytes of padding per plane spreads them. */
__host__ __device__ inline size_t ublkPlane(int cap) { return static_cast<size_t>(cap) * kUblk + kUblkPad; }
constexpr int kPartial = 48;                // per sweep block: G (36 upper) + q (8) + energy + n_valid + pad
constexpr int kSweepThreads = 128;        // 16 (landmark, target) items x 8 pattern pixels

// landmark flag bits (LocalFrame::Landmark booleans, PBA_INT/local_frame.hpp:276-293)
constexpr uint8_t kFlagMarginalized = 1, kFlagOutlier = 2, kFlagToMarginalize = 4, kFlagIllConditioned = 8;

/** per-frame topology + landmark SoA + residual tables; uploaded by the host whenever it changes.
 *  Replaces LocalFrame (PBA_INT/local_frame.hpp:232-584) minus the materialised ResidualPoint Jacobians:
 *  of ResidualPoint's 259 scalars only status, candidate status, energy and the FEJ validity bit are kept. */
struct FrameDev {
  const hbm_void *texels;  // Texel<S>* of the level this frame was pushed with
  const hbm_void *iplane;  // tiled intensity plane of that level (nullptr: none), itiles tiles per row
  int itiles, pad_i;
  int width, height;
  double fx, fy, cx, cy;
  double exposure;
  int fixed, is_marginalized, to_marginalize;
  int n;    // landmarks
  int cap;  // landmark capacity (stride of the per-slot planes of ublk)
  int first_conn;  // first connected target slot (-1 = none): the thread of that pair owns per-landmark sums
  int pad;
  hbm_f64 *uv, *idepth, *idepth_step, *idepth_fej, *patch;
  hbm_f64 *inv_hdd, *b_d, *relative_baseline;
  hbm_i32 *n_inliers;
  hbm_u8 *flags;
  hbm_f64 *ublk;  // [kMaxFrames][cap][kUblk]: slot t != r: {-u_pt (= h_p block t), hdd_pt, bd_pt}; slot r: {h_p block r, -, -}
  hbm_u8 *status[kMaxFrames], *cand[kMaxFrames], *fej_valid[kMaxFrames];  // by target slot; nullptr = no connection
  hbm_f64 *energy[kMaxFrames];
  int n_res[kMaxFrames];
  // device-side snapshot (dsopp_hip_window_snapshot / _restore)
  hbm_f64 *snap_idepth;
  hbm_u8 *snap_flags;
  hbm_u8 *snap_status[kMaxFrames];
};

/** dynamic state of the window, lives in HBM and is advanced by the kernels */
struct WindowState {
  double T0_R[kMaxFrames][9];
  double T0_t[kMaxFrames][3];
  double ab0[kMaxFrames][2];
  double eps[kMaxFrames][kBlk];
  double step[kMaxFrames][kBlk];
};

/** constants of one ordered frame pair (reference r -> target t), rebuilt whenever the state moves */
struct PairConst {
  double M[12];    // current reproject_ = K_t [R|t] Kinv_r at eps+step   (camera_reproject.hpp:256)
  double U[12];    // transform_unproject_ = [R|t] Kinv_r used for the Jacobians: linearisation point when FEJ (:258)
  double tl[3];    // translation of the transform behind U (:259)
  double Adj[36];  // rightLogTransformer: Adj(T_tr0) when FEJ else Adj(T_tr) (evaluate_jacobians.hpp:62-64)
  double fxt, fyt, cxt, cyt;
  double s;        // current brightness_change_scale (evaluate_jacobians.hpp:56-57)
  double s0;       // residual.brightness_change_scale cached by firstEstimateJacobians_ (== s when not FEJ)
  double sigma_r;  // scale behind landmark.corrected_intensities: the LAST connected target's s0 (first_estimate_jacobians.hpp:57-62)
  double b_t, b_r; // current affine offsets
  double b_r0;     // reference offset at the linearisation point (== b_r when not FEJ)
  int valid;       // connection exists
  int pad;
  double T0rel[12];  // T_tr0 = inv(T_t0) * T_r0 as [R | t] rows (3 x 4): lets the solve kernel refresh M without the frame table
};

/** LM control block (lives in HBM; double-buffered by iteration parity so every workgroup of the decide step can read the
 *  incoming state while workgroup 0 writes the outgoing one) */
struct LmControl {
  double lambda;
  double energy;           // result.energy
  double cand_prior;       // prior + marginal energy of the candidate state eps + step (written by the solve kernel)
  double idepth_sq;        // running sum of idepth^2 over all landmarks of this rank (state norm part)
  // frame part of acceptStep's norms (problem.hpp:366-388) for the PENDING candidate: |eps|^2 + |ab0|^2 and |step|^2, written by the
  // solve kernel next to the step itself.  The decision reads them here and not from WindowState, which workgroup 0 of the deciding
  // launch overwrites (eps += step, step = 0) while later workgroups of the same launch may not have started yet.
  double frame_state_sq, frame_step_sq;
  int n_valid;             // result.number_of_valid_residuals
  int converged;
  int active;              // loop still running           (sweep kernels read {active, linear_system_valid} as int[2])
  int linear_system_valid;
  int iteration;           // loop bodies executed
  int need_final_setup;    // last step was rejected: pair constants must be rebuilt before the closing energy sweep
  int pending;             // fused loop: a candidate step is waiting for its energy (0 in the first round / after a re-linearisation)
  int relin;               // fused loop: the last candidate was rejected, the next sweep re-linearises at the reverted state
};

/** one thread block of a sweep = a chunk of landmarks of one ordered pair.  Like SchurBlock the descriptor repeats every
 *  pointer the kernel needs: table -> data is one dependent load instead of table -> FrameDev -> data. */
struct SweepBlock {
  int r, t;    // frame slots
  int offset;  // first landmark
  int n_res;   // residuals of the pair (== landmarks of r that have a residual in t)
  int cap;     // landmark capacity of frame r (plane stride of ublk)
  int owns_landmark_sums;  // t is the first connected target of r: its items own the per-landmark sums / writes
  int width_r, height_r, width_t, height_t;
  unsigned conn_mask;  // bit k: frame r has residuals in target slot k
  int n_groups;        // groups of 16 items this workgroup sweeps (1 except in the coarse table of large windows)
  const hbm_f64 *uv, *idepth, *patch, *idepth_fej, *b_d, *inv_hdd;
  hbm_f64 *idepth_step, *ublk, *energy;
  const hbm_u8 *flags, *status, *fej_valid;
  hbm_u8 *cand;
  const hbm_void *texels_t;  // Texel<S>* of the target frame's level
  // tiled intensity plane of the target level (pyramid.hpp; nullptr: none — f32 storage or switched off): what residual-only
  // sweeps sample instead of the texels; itiles_t = 4 x 2 tiles per image row
  const hbm_void *iplane_t;
  int itiles_t;
  int partial_row;     // row of the sweep's per-workgroup sums this entry writes (= its index in the pair-contiguous numbering the pair
                       // reductions walk; differs from the launch index only under the XCD-banded launch order, an experiment)
};

/** one thread block of the Schur kernel = a chunk of landmarks of one frame.  The descriptor repeats the frame's
 *  pointers so that a kernel reaches its data with one dependent load (table -> data) instead of two (table -> FrameDev -> data). */
struct SchurBlock {
  int r;
  int offset;
  int n, cap;  // landmarks / landmark capacity of frame r
  int fixed;
  unsigned conn_mask;  // bit t: frame r has residuals in target slot t
  hbm_f64 *idepth, *idepth_step, *inv_hdd, *b_d, *ublk;
  hbm_u8 *flags;
  hbm_u8 *status[kMaxFrames], *cand[kMaxFrames];
  int n_res[kMaxFrames];
};

struct SolveParams {
  double lambda;
  double affine_reg[2];
  double fixed_reg;
  int F;
  int n_sweep_blocks;
  int use_marginal;
};

}  // namespace dsopp_hip
