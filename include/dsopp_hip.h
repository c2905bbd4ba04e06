/* dsopp_hip.h — C-ABI of the MI355X-native photometric bundle adjustment / direct image alignment hot path.
 *
 * This is the drop-in boundary: a thin `extern "C"` layer with plain pointers and sizes that a
 * `solver: hip` backend of DSOPP binds to.  The reference itself has no FFI; its plugin API is a pair of C++ abstract
 * class templates selected by a YAML string in a factory (src/tracker/tracker/src/fabric.cpp:58-180).  Each entry point
 * below cites the reference member function whose work it replaces.  INTEGRATION.md shows the reference-side adapter
 * classes (`HipPhotometricBundleAdjustment`, `HipPoseAlignment`, mirrored in dsopp_amd/host/) that call these.
 *
 * Path shorthands used in citations:
 *   PBA_INC  = src/energy/problems/include/energy/problems/photometric_bundle_adjustment
 *   PBA_INT  = src/energy/problems/internal/energy/problems/photometric_bundle_adjustment
 *   PA_INC   = src/energy/problems/include/energy/problems/pose_alignment
 *   PROB_SRC = src/energy/problems/src
 *
 * Conventions
 *   - every function returns DSOPP_HIP_OK (0) or a negative error code; dsopp_hip_last_error() gives the message
 *     (thread-local).  No exceptions cross the boundary.
 *   - all host buffers are caller-owned and only read/written during the call; handles are opaque and used from one
 *     thread at a time (the reference calls its solvers from the single tracker thread only).
 *   - poses are 7 doubles in Sophus::SE3 storage order (qx, qy, qz, qw, tx, ty, tz); tangent vectors are
 *     (translation, rotation); per-frame state blocks are 8 doubles (6 pose + affine a, b); K = 8 * number of frames.
 *   - matrices are row-major doubles.
 *   - there is NO CPU fallback: every entry point that computes fails with DSOPP_HIP_ERR_HIP when no gfx950 device
 *     is available.
 */
#ifndef DSOPP_HIP_H
#define DSOPP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSOPP_HIP_PATTERN_SIZE 8 /* Pattern::kSize — src/common/pattern/include/common/pattern/pattern.hpp:17 */
#define DSOPP_HIP_BLOCK_SIZE 8   /* Motion::DoF + 2 */
#define DSOPP_HIP_MAX_FRAMES 16  /* window capacity (reference configs use 5..15 keyframes) */
#define DSOPP_HIP_MAX_LEVELS 5   /* PixelDataFrame::kMaxPyramidDepth — src/features/include/features/camera/pixel_data_frame.hpp:26 */

enum {
  DSOPP_HIP_OK = 0,
  DSOPP_HIP_ERR_INVALID_ARGUMENT = -1,
  DSOPP_HIP_ERR_NOT_FOUND = -2,
  DSOPP_HIP_ERR_ORDER = -3,    /* frames must be pushed in ascending timestamp order (PROB_SRC/photometric_bundle_adjustment.cpp:101-102) */
  DSOPP_HIP_ERR_HIP = -4,      /* HIP runtime failure / no device */
  DSOPP_HIP_ERR_CAPACITY = -5, /* window full */
  DSOPP_HIP_ERR_STATE = -6     /* call sequence violation (e.g. stage call before begin) */
};

/* track::PointConnectionStatus — src/track/connections/include/track/connections/frame_connection.hpp:19-25 */
enum { DSOPP_HIP_STATUS_OK = 0, DSOPP_HIP_STATUS_OUTLIER = 1, DSOPP_HIP_STATUS_OCCLUDED = 2, DSOPP_HIP_STATUS_OOB = 3, DSOPP_HIP_STATUS_UNKNOWN = 4 };

/* storage / evaluation scalar of the device images and sweeps.  F64 matches the reference's default build
 * (Precision = double, src/common/include/common/settings.hpp:10-14); F32 matches its -DUSE_FLOAT=ON build.
 * Normal equations are accumulated in fp64 in both. */
enum { DSOPP_HIP_F64 = 0, DSOPP_HIP_F32 = 1 };

const char *dsopp_hip_last_error(void);
int dsopp_hip_device_count(int *count);
const char *dsopp_hip_version(void);

/* TrustRegionPhotometricBundleAdjustmentOptions (PBA_INC/trust_region_photometric_bundle_adjustment_options.hpp:14-52)
 * + the EigenPhotometricBundleAdjustment ctor flags (PROB_SRC/eigen_photometric_bundle_adjustment.cpp:47-57)
 * + the class template switches FIRST_ESTIMATE_JACOBIANS / OPTIMIZE_IDEPTHS */
typedef struct dsopp_hip_options {
  int32_t max_iterations;
  double initial_trust_region_radius;
  double function_tolerance;
  double parameter_tolerance;
  double affine_brightness_regularizer[2];
  double fixed_state_regularizer;
  double sigma_huber_loss;
  int32_t estimate_uncertainty;
  int32_t force_accept;
  int32_t first_estimate_jacobians;
  int32_t optimize_idepths;
  int32_t dtype; /* DSOPP_HIP_F64 | DSOPP_HIP_F32 */
} dsopp_hip_options;

/* production values of createPhotometricBundleAdjustment / createPoseAlignment — src/tracker/tracker/src/fabric.cpp:63-79,127-142 */
void dsopp_hip_default_pba_options(dsopp_hip_options *o);
void dsopp_hip_default_align_options(dsopp_hip_options *o);

/* ------------------------------------------------------------------------------------------------------------------
 * Image pyramid of one frame, resident in HBM (replaces features::PixelDataFrame / PixelMap<1> levels + CameraMask)
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct dsopp_hip_pyramid dsopp_hip_pyramid;

/* allocate `levels` (<= DSOPP_HIP_MAX_LEVELS) texel images of width>>l x height>>l on `device`.
 * `stream` is a hipStream_t (NULL = the library creates its own non-blocking stream). */
int dsopp_hip_pyramid_create(int device, void *stream, int width, int height, int levels, int dtype, dsopp_hip_pyramid **out);
void dsopp_hip_pyramid_destroy(dsopp_hip_pyramid *p);
/* PixelDataFrame ctor (src/features/src/pixel_data_frame.cpp:12-31): photometric correction LUT[u8] * vmax/(vignette+1)
 * (src/features/src/photometrically_corrected_image.cpp:9-29), 2x2 box pyramid (downscale_image.hpp:16-33), per-level
 * (I, dI/dx, dI/dy) (src/features/src/calculate_pixelinfo.cpp:340-374).  lut256 / vignetting may be NULL.
 * The host arrays are consumed before the call returns (the image through a pinned copy: its upload and the build run behind the call on
 * the pyramid's stream; every consumer of the library orders itself behind them). */
int dsopp_hip_pyramid_build(dsopp_hip_pyramid *p, const uint8_t *image_host, const double *lut256, const uint8_t *vignetting_host);
/* same, the u8 image (and vignette) already in HBM: no PCIe transfer of pixels inside the call and no host sync
 * (the call only enqueues work on the pyramid's stream).  vignetting_max = max over the vignette image
 * (cv::minMaxLoc in the reference, a per-camera constant); ignored when vignetting_dev is NULL. */
int dsopp_hip_pyramid_build_device(dsopp_hip_pyramid *p, const void *image_dev, const double *lut256, const void *vignetting_dev,
                                   double vignetting_max);
/* adopt a level built by the reference's host code: pixelinfo = H_l x W_l x (I, dx, dy) doubles = PixelInfo<1>::data_
 * (src/features/include/features/camera/pixel_map.hpp:79-132) */
int dsopp_hip_pyramid_set_level(dsopp_hip_pyramid *p, int level, const double *pixelinfo_host);
/* CameraMask of a level (src/sensors/camera_calibration/include/sensors/camera_calibration/mask/camera_mask.hpp:48-89);
 * NULL = all valid */
int dsopp_hip_pyramid_set_mask(dsopp_hip_pyramid *p, int level, const uint8_t *mask_host);
int dsopp_hip_pyramid_get_level(dsopp_hip_pyramid *p, int level, double *pixelinfo_host);
int dsopp_hip_pyramid_level_size(dsopp_hip_pyramid *p, int level, int *width, int *height);

/* ------------------------------------------------------------------------------------------------------------------
 * Sliding-window photometric bundle adjustment
 * (replaces EigenPhotometricBundleAdjustment<SE3, PinholeCamera, 8, PixelMap, true, true, true, 1>)
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct dsopp_hip_window dsopp_hip_window;

int dsopp_hip_window_create(const dsopp_hip_options *options, int device, void *stream, dsopp_hip_window **out);
void dsopp_hip_window_destroy(dsopp_hip_window *w);

/* pushFrame(ActiveKeyframe, level, model, parameterization) — PBA_INC/photometric_bundle_adjustment.hpp:55-56,
 * PROB_SRC/eigen_photometric_bundle_adjustment.cpp:119-141: when the window already holds > 1 frame, first folds the
 * landmarks/frames flagged for marginalisation into the marginal prior (updateMarginalizedLinearSystem,
 * PBA_INT/eigen_photometric_bundle_adjustment_problem.hpp:146-203), then appends the frame.
 * The pyramid is BORROWED: it must outlive the frame's stay in the window (the reference keeps raw pointers to the
 * keyframe's PixelMap levels, PBA_INT/local_frame.hpp:323-325).  intrinsics = (fx, fy, cx, cy) of that level. */
int dsopp_hip_window_push_frame(dsopp_hip_window *w, int32_t frame_id, int64_t timestamp, const dsopp_hip_pyramid *pyramid,
                                int level, const double intrinsics[4], const double T_world_agent[7], double exposure_time,
                                const double affine_brightness[2], int fixed, int is_marginalized);
/* LocalFrame ctor landmark copy + LocalFrame::update (PBA_INT/local_frame.hpp:309-335,484-505).  n_total >= current count;
 * existing landmarks only get their flags refreshed (to_marginalize = newly marginalised && !outlier), new ones are
 * appended.  uv = landmark.projection(), patch = 8 intensities of the host level-0 image (src/track/frames/src/
 * active_keyframe.cpp:96-110).  flags bit0 = isMarginalized, bit1 = isOutlier.
 * The arrays are copied before the call returns; the device side of set_landmarks / set_connection is QUEUED and applied — one transfer,
 * one launch for everything queued — in front of the next call that uses the window's device state (any solve / stage / getter /
 * push_frame / activation call): a keyframe step's ~120 appends cost host time only.  A device error of a queued append is reported by
 * that later call. */
int dsopp_hip_window_set_landmarks(dsopp_hip_window *w, int32_t frame_id, int32_t n_total, const double *uv, const double *idepth,
                                   const double *patch, const uint8_t *flags);
/* residual lists from FrameConnection statuses (PROB_SRC/photometric_bundle_adjustment.cpp:109-123, local_frame.hpp:507-519):
 * appends entries [current size, n) of the (reference, target) connection.  A target that is not in the window (not pushed yet, or
 * folded into the prior) is legal, as in LocalFrame::update, which keeps the list of every connection: the statuses are kept on the
 * host as given — they become the device list when a frame with that id is pushed, and they are what get_frame_update returns
 * for such a target. */
int dsopp_hip_window_set_connection(dsopp_hip_window *w, int32_t reference_id, int32_t target_id, int32_t n, const uint8_t *statuses);
/* updateLocalFrame's frame flags (PROB_SRC/eigen_photometric_bundle_adjustment.cpp:106-113) */
int dsopp_hip_window_mark_frame_marginalized(dsopp_hip_window *w, int32_t frame_id);
int dsopp_hip_window_num_frames(dsopp_hip_window *w, int32_t *n);
/* ids of the frames currently in the window, oldest first (frames_ of PBA_INC/photometric_bundle_adjustment.hpp:181 after
 * the fold-in of pushFrame erased the marginalised ones): a caller that owns the borrowed pyramids learns here which of them the
 * window has let go of.  At most `capacity` ids are written, *n receives the count. */
int dsopp_hip_window_frame_ids(dsopp_hip_window *w, int32_t capacity, int32_t *ids, int32_t *n);

/* solve(number_of_threads) -> final energy — PBA_INC/photometric_bundle_adjustment.hpp:154,
 * PROB_SRC/eigen_photometric_bundle_adjustment.cpp:61-101 (FEJ, LM loop on device, relinearise, covariances, point statuses) */
int dsopp_hip_window_solve(dsopp_hip_window *w, double *energy, int32_t *iterations, int32_t *n_valid);

/* stage-level entry points = PhotometricBundleAdjustmentProblem methods (PBA_INT/eigen_photometric_bundle_adjustment_problem.hpp:290-402)
 * for host-driven LM and stage-by-stage parity checks */
int dsopp_hip_window_begin(dsopp_hip_window *w);                                           /* problem ctor + firstEstimateJacobians */
int dsopp_hip_window_calculate_energy(dsopp_hip_window *w, double *energy, int32_t *n_valid); /* :290-317 */
int dsopp_hip_window_linearize(dsopp_hip_window *w);                                        /* :322-336 */
int dsopp_hip_window_get_system(dsopp_hip_window *w, double *H_pp, double *b_pp, double *H_schur, double *b_schur);
int dsopp_hip_window_calculate_step(dsopp_hip_window *w, double lambda, double *step);     /* :342-361 */
int dsopp_hip_window_accept_step(dsopp_hip_window *w, double *state_sq, double *step_sq);  /* :366-388 */
int dsopp_hip_window_reject_step(dsopp_hip_window *w);                                      /* :392-402 */
int dsopp_hip_window_update_point_statuses(dsopp_hip_window *w);                            /* PROB_SRC/photometric_bundle_adjustment.cpp:321-406 */

/* read-back = PhotometricBundleAdjustment::updateFrame / getPose / getAffineBrightness
 * (PROB_SRC/photometric_bundle_adjustment.cpp:156-264) */
int dsopp_hip_window_get_frame_state(dsopp_hip_window *w, int32_t frame_id, double T0[7], double ab0[2], double eps[8], double step[8]);
int dsopp_hip_window_get_pose(dsopp_hip_window *w, int32_t frame_id, double T_world_agent[7], double affine_brightness[2]);
int dsopp_hip_window_num_landmarks(dsopp_hip_window *w, int32_t frame_id, int32_t *n);
/* updateFrame in one transfer (PROB_SRC/photometric_bundle_adjustment.cpp:182-264 reads, per keyframe: idepths, H_dd^-1 for the
 * idepth variance, relative baselines, inlier counts, outlier flags and the connection statuses towards every other frame):
 * a gather kernel packs everything of one frame, one pinned device-to-host copy, one synchronisation.  Any landmark array may
 * be NULL; statuses receives n_targets rows of n bytes in the order of target_ids.  (The per-array getters below cost one
 * copy + wait each.) */
int dsopp_hip_window_get_frame_update(dsopp_hip_window *w, int32_t frame_id, double *idepth, double *inv_hessian_idepth, double *relative_baseline,
                                      int32_t *n_inliers, uint8_t *flags_out, int32_t n_targets, const int32_t *target_ids, uint8_t *statuses);
/* any output may be NULL.  flags_out bit0 marginalized, bit1 outlier, bit2 to_marginalize, bit3 ill_conditioned;
 * hpib (hessian_poses_idepth_block) is n x K */
int dsopp_hip_window_get_landmarks(dsopp_hip_window *w, int32_t frame_id, double *idepth, double *idepth_step, double *inv_hessian_idepth,
                                   double *b_idepth, double *relative_baseline, int32_t *n_inliers, uint8_t *flags_out, double *hpib);
int dsopp_hip_window_get_residuals(dsopp_hip_window *w, int32_t reference_id, int32_t target_id, int32_t n, uint8_t *status,
                                   uint8_t *candidate, double *energy);
int dsopp_hip_window_get_marginalized(dsopp_hip_window *w, double *H, double *b, double *energy, int32_t *size);
int dsopp_hip_window_get_covariance(dsopp_hip_window *w, int32_t reference_id, int32_t target_id, double cov[36]);

/* multi-GPU: landmarks are sharded across ranks by the caller (each rank uploads only its shard; frames and images are
 * replicated).  The library calls `allreduce_sum` on its stream whenever partial sums over landmarks must be combined
 * (the reduced normal equations once per linearisation, energy + valid count once per energy sweep).  `device_buffer`
 * is device memory holding `count` doubles to be summed in place across ranks.  With no callback the window is single-GPU. */
typedef int (*dsopp_hip_allreduce_fn)(void *user, void *device_buffer, size_t count, void *stream);
int dsopp_hip_window_set_allreduce(dsopp_hip_window *w, dsopp_hip_allreduce_fn fn, void *user, int rank, int world_size);

/* The same exchange as native code: a communicator of one rank per GPU (one process per GPU) whose all-reduce the library
 * enqueues itself — ncclAllReduce(sum, double) over RCCL / xGMI on the window's stream, no callback into the host language.
 * Replaces the mutex reduction of the per-thread partial systems (PBA_INT/hessian_block_evaluation.hpp:101-145,178-235) across
 * GPUs.  Rank 0 obtains an id (dsopp_hip_comm_unique_id), hands it to the other ranks by any means (MPI, a file, a socket),
 * then EVERY rank calls dsopp_hip_comm_create (collective).  dsopp_hip_comm_adopt wraps an ncclComm_t the host program
 * already owns (not destroyed with the wrapper): the communicator must come from the SAME loaded librccl this library resolves
 * (the instance already mapped into the process — e.g. PyTorch's — else /opt/rocm/lib/librccl.so.1); a handle of another RCCL
 * build (statically linked, differently named) must not be adopted.  `device` is the device the communicator's rank runs on.
 * librccl is loaded on first use only. */
#define DSOPP_HIP_COMM_ID_BYTES 128
typedef struct dsopp_hip_comm dsopp_hip_comm;
int dsopp_hip_comm_unique_id(uint8_t id[DSOPP_HIP_COMM_ID_BYTES]);
int dsopp_hip_comm_create(const uint8_t id[DSOPP_HIP_COMM_ID_BYTES], int rank, int world_size, int device, dsopp_hip_comm **out);
int dsopp_hip_comm_adopt(void *nccl_comm, int device, dsopp_hip_comm **out);
void dsopp_hip_comm_destroy(dsopp_hip_comm *c);
/* ncclCommAbort: releases collectives the other ranks have enqueued and that this rank will never join (it failed between two of
 * them); the handle can only be destroyed afterwards, a window it is attached to fails its next collective with an error instead of
 * enqueueing it.  What dsopp_hip_window_group does with every shard's communicator when one shard fails. */
int dsopp_hip_comm_abort(dsopp_hip_comm *c);
int dsopp_hip_comm_rank(const dsopp_hip_comm *c, int *rank, int *world_size);
/* in-place sum of `count` doubles in device memory across the ranks, ordered on `stream` (a hipStream_t) */
int dsopp_hip_comm_allreduce(dsopp_hip_comm *c, void *device_buffer, size_t count, void *stream);
/* attach (NULL: detach) a communicator: the window sums its partial systems / energy scalars through it.  The communicator
 * must outlive its use by the window; landmarks are sharded by the caller exactly as with dsopp_hip_window_set_allreduce. */
int dsopp_hip_window_set_comm(dsopp_hip_window *w, dsopp_hip_comm *comm);

/* ------------------------------------------------------------------------------------------------------------------
 * Window group: ONE host process, n landmark shards of one sliding window on n devices
 *
 * The reference builds ONE solver object in ONE process (createPhotometricBundleAdjustment, src/tracker/tracker/src/fabric.cpp:58-121,
 * moved into the tracker that src/application/dsopp_main.cpp:114-119 runs), so the multi-GPU form of the drop-in has to live behind
 * one object too.  A group has the calls of dsopp_hip_window (same argument meaning and error codes; every dsopp_hip_window_group_X
 * replaces what dsopp_hip_window_X replaces) and does the sharding itself:
 *   - frames, images, poses, the marginal prior: replicated on every device (dsopp_hip_pyramid_group = one pyramid per distinct device);
 *   - landmarks and their connection statuses: landmark j of a keyframe lives on shard j % n at local index j / n (balanced to +-1 and
 *     stable under the appends of PROB_SRC/photometric_bundle_adjustment.cpp:109-123); the getters interleave them back;
 *   - every Gauss-Newton iteration sums the partial combined systems with ONE collective (F(F+1)/2 * 64 + K + 4 doubles), every shard
 *     then takes the same LM decision, solves the same K x K system and back-substitutes its own landmarks.
 * One worker thread per shard enqueues that shard's launches.  transport: RCCL = ncclAllReduce on the shards' streams over xGMI, one
 * rank per device (all device ids distinct); LOCAL = an event-ordered sum kernel inside the process (peer access between the
 * devices; the only choice when shards share a device, e.g. to exercise the sharded path on a single GPU); AUTO = RCCL when the ids
 * are distinct (falling back to LOCAL when no communicator can be created, e.g. no librccl on the node — an explicit RCCL request
 * fails instead), else LOCAL.  P2P = the one-shot all-reduce: every shard stores its partial sums (15 .. 41 KB per Gauss-Newton
 * iteration) straight into every peer's receive area over the direct xGMI links and adds up what the others stored into its own — one
 * kernel per shard and collective, no ring, no host barrier (fine-grained device memory, system-scope flags, every wait bounded: a
 * time-out is reported as an error and leaves the group unusable); buffers beyond the receive area take the LOCAL reducer.  Never
 * chosen by AUTO.  EXPERIMENTAL across distinct devices: that branch has never run on a multi-GPU node, so a P2P request with distinct
 * device ids is refused (DSOPP_HIP_ERR_INVALID_ARGUMENT) unless the process sets DSOPP_HIP_P2P_EXPERIMENTAL=1; with all shards on one
 * device (what the one-GPU tests execute) it needs no opt-in.
 * dsopp_hip_window_group_size reports the transport in use.  A group of one shard is a plain window.
 * ---------------------------------------------------------------------------------------------------------------- */
enum { DSOPP_HIP_TRANSPORT_AUTO = 0, DSOPP_HIP_TRANSPORT_RCCL = 1, DSOPP_HIP_TRANSPORT_LOCAL = 2, DSOPP_HIP_TRANSPORT_P2P = 3 };
typedef struct dsopp_hip_window_group dsopp_hip_window_group;
typedef struct dsopp_hip_pyramid_group dsopp_hip_pyramid_group;
typedef struct dsopp_hip_depth_maps dsopp_hip_depth_maps;

int dsopp_hip_window_group_create(const dsopp_hip_options *options, const int32_t *device_ids, int32_t n, int32_t transport,
                                  dsopp_hip_window_group **out);
void dsopp_hip_window_group_destroy(dsopp_hip_window_group *g);
int dsopp_hip_window_group_size(const dsopp_hip_window_group *g, int32_t *n, int32_t *transport);
/* introspection: the shard's own window (owned by the group; for read-only calls and tests) and its device */
int dsopp_hip_window_group_shard(dsopp_hip_window_group *g, int32_t shard, dsopp_hip_window **window, int32_t *device);

/* the keyframe's image pyramid on every device of the group (PixelDataFrame is one host object in the reference; here one
 * device-resident copy per distinct device, each built on its own device from the same u8 image / adopted host levels) */
int dsopp_hip_pyramid_group_create(dsopp_hip_window_group *g, int width, int height, int levels, dsopp_hip_pyramid_group **out);
void dsopp_hip_pyramid_group_destroy(dsopp_hip_pyramid_group *pg);
int dsopp_hip_pyramid_group_build(dsopp_hip_pyramid_group *pg, const uint8_t *image_host, const double *lut256, const uint8_t *vignetting_host);
int dsopp_hip_pyramid_group_set_level(dsopp_hip_pyramid_group *pg, int level, const double *pixelinfo_host);
int dsopp_hip_pyramid_group_set_mask(dsopp_hip_pyramid_group *pg, int level, const uint8_t *mask_host);
int dsopp_hip_pyramid_group_get(dsopp_hip_pyramid_group *pg, int32_t shard, dsopp_hip_pyramid **pyramid);

/* pushFrame / LocalFrame landmark copy / residual lists / updateLocalFrame flags — as the dsopp_hip_window_ calls of the same name;
 * landmark arrays and status lists are the WHOLE keyframe's (the group deals them out) */
int dsopp_hip_window_group_push_frame(dsopp_hip_window_group *g, int32_t frame_id, int64_t timestamp, const dsopp_hip_pyramid_group *pyramids,
                                      int level, const double intrinsics[4], const double T_world_agent[7], double exposure_time,
                                      const double affine_brightness[2], int fixed, int is_marginalized);
int dsopp_hip_window_group_set_landmarks(dsopp_hip_window_group *g, int32_t frame_id, int32_t n_total, const double *uv, const double *idepth,
                                         const double *patch, const uint8_t *flags);
int dsopp_hip_window_group_set_connection(dsopp_hip_window_group *g, int32_t reference_id, int32_t target_id, int32_t n, const uint8_t *statuses);
int dsopp_hip_window_group_mark_frame_marginalized(dsopp_hip_window_group *g, int32_t frame_id);
int dsopp_hip_window_group_num_frames(dsopp_hip_window_group *g, int32_t *n);
int dsopp_hip_window_group_frame_ids(dsopp_hip_window_group *g, int32_t capacity, int32_t *ids, int32_t *n);
int dsopp_hip_window_group_num_landmarks(dsopp_hip_window_group *g, int32_t frame_id, int32_t *n);
/* solve(number_of_threads) and the stage entry points (PBA_INT/eigen_photometric_bundle_adjustment_problem.hpp:290-402) */
int dsopp_hip_window_group_solve(dsopp_hip_window_group *g, double *energy, int32_t *iterations, int32_t *n_valid);
int dsopp_hip_window_group_optimize(dsopp_hip_window_group *g, double *energy, int32_t *iterations, int32_t *n_valid);
int dsopp_hip_window_group_begin(dsopp_hip_window_group *g);
int dsopp_hip_window_group_calculate_energy(dsopp_hip_window_group *g, double *energy, int32_t *n_valid);
int dsopp_hip_window_group_linearize(dsopp_hip_window_group *g);
int dsopp_hip_window_group_get_system(dsopp_hip_window_group *g, double *H_pp, double *b_pp, double *H_schur, double *b_schur);
int dsopp_hip_window_group_calculate_step(dsopp_hip_window_group *g, double lambda, double *step);
int dsopp_hip_window_group_accept_step(dsopp_hip_window_group *g, double *state_sq, double *step_sq);
int dsopp_hip_window_group_reject_step(dsopp_hip_window_group *g);
int dsopp_hip_window_group_update_point_statuses(dsopp_hip_window_group *g);
/* updateFrame / getPose / getAffineBrightness read-back; landmark arrays come back in the keyframe's own landmark order */
int dsopp_hip_window_group_get_frame_state(dsopp_hip_window_group *g, int32_t frame_id, double T0[7], double ab0[2], double eps[8], double step[8]);
int dsopp_hip_window_group_get_pose(dsopp_hip_window_group *g, int32_t frame_id, double T_world_agent[7], double affine_brightness[2]);
int dsopp_hip_window_group_get_landmarks(dsopp_hip_window_group *g, int32_t frame_id, double *idepth, double *idepth_step, double *inv_hessian_idepth,
                                         double *b_idepth, double *relative_baseline, int32_t *n_inliers, uint8_t *flags_out, double *hpib);
int dsopp_hip_window_group_get_frame_update(dsopp_hip_window_group *g, int32_t frame_id, double *idepth, double *inv_hessian_idepth,
                                            double *relative_baseline, int32_t *n_inliers, uint8_t *flags_out, int32_t n_targets,
                                            const int32_t *target_ids, uint8_t *statuses);
int dsopp_hip_window_group_get_residuals(dsopp_hip_window_group *g, int32_t reference_id, int32_t target_id, int32_t n, uint8_t *status,
                                         uint8_t *candidate, double *energy);
int dsopp_hip_window_group_get_marginalized(dsopp_hip_window_group *g, double *H, double *b, double *energy, int32_t *size);
int dsopp_hip_window_group_get_covariance(dsopp_hip_window_group *g, int32_t reference_id, int32_t target_id, double cov[36]);
/* createReferenceDepthMaps over all shards: the level-0 splat planes are summed across the shards, the maps returned live on
 * device_ids[0] (where the tracker's aligner runs); refill needs maps this group created */
int dsopp_hip_window_group_create_reference_depth_maps(dsopp_hip_window_group *g, int32_t levels, dsopp_hip_depth_maps **out);
int dsopp_hip_window_group_refill_reference_depth_maps(dsopp_hip_window_group *g, dsopp_hip_depth_maps *maps);
/* settings / measurement aids, as the dsopp_hip_window_ calls of the same name */
int dsopp_hip_window_group_set_lm_mode(dsopp_hip_window_group *g, int mode);
int dsopp_hip_window_group_set_deterministic(dsopp_hip_window_group *g, int enable);
int dsopp_hip_window_group_set_max_iterations(dsopp_hip_window_group *g, int32_t max_iterations);
int dsopp_hip_window_group_snapshot(dsopp_hip_window_group *g);
int dsopp_hip_window_group_restore(dsopp_hip_window_group *g);
int dsopp_hip_window_group_optimize_repeated(dsopp_hip_window_group *g, int32_t iterations_target, int32_t *iterations_done, double *last_energy);
int dsopp_hip_window_group_last_solve_ms(dsopp_hip_window_group *g, float *ms);

/* the Levenberg-Marquardt loop of solve() alone (firstEstimateJacobians + levenberg_marquardt_algorithm::solve,
 * PROB_SRC/eigen_photometric_bundle_adjustment.cpp:83-86) without the post-processing (relinearise, covariance, statuses).
 * One loop body = one Gauss-Newton iteration = linearize + calculateStep + calculateEnergy + accept/reject. */
int dsopp_hip_window_optimize(dsopp_hip_window *w, double *energy, int32_t *iterations, int32_t *n_valid);
/* the same split in two for callers that drive several independent windows from one host thread (a mapping server with many
 * sessions per GPU): _async enqueues the whole LM loop on the window's stream and returns, _wait is the one host
 * synchronisation and returns the results.  One window is latency-bound (three dependent launches per iteration), so
 * windows on different streams overlap: 8 C1 windows reach ~3x the single-window rate on one MI355X (bench.py,
 * "concurrent_windows").  No other call on this window between the two. */
int dsopp_hip_window_optimize_async(dsopp_hip_window *w);
int dsopp_hip_window_optimize_wait(dsopp_hip_window *w, double *energy, int32_t *iterations, int32_t *n_valid);
/* 0 (default): fused device-side LM loop — 3 launches per Gauss-Newton iteration, one read-back per solve;
 * 1: control flow on the host through the stage entry points (one small read-back per energy evaluation);
 * 2: unfused device-side loop (5 launches per iteration).  Same arithmetic in all three; 1 and 2 are kept for debugging and
 * as parity cross-checks of the fused control logic */
int dsopp_hip_window_set_lm_mode(dsopp_hip_window *w, int mode);
/* Summation order of the reduced normal equations inside the fused LM loop.  0 (default): windows of up to 192 chunks of 64
 * landmarks accumulate H_schur with fp64 atomics (fewest launches; the sum order, hence the last bits, vary from run to run — the
 * reference's own reduction under a mutex, hessian_block_evaluation.hpp:101-145, has the same property), larger windows use the
 * two-stage build (per-workgroup partial systems, then one ordered sum per entry: no atomics, bit-reproducible, and faster there).
 * 1: the two-stage build at every size — the fused LM loop (lm_mode 0, the default), the host-driven loop (lm_mode 1) and the stage
 * entry point dsopp_hip_window_linearize are then bit-reproducible from run to run, at the cost of two more launches per Gauss-Newton
 * iteration.  Not covered: the unfused device loop (lm_mode 2, a debugging aid) and the fold-in of marginalised landmarks inside
 * push_frame, which keep the atomic accumulation. */
int dsopp_hip_window_set_deterministic(dsopp_hip_window *w, int enable);
/* TrustRegion...Options::max_iterations of an existing window */
int dsopp_hip_window_set_max_iterations(dsopp_hip_window *w, int32_t max_iterations);
/* Measurement aid (bench.py): repeats { dsopp_hip_window_restore; dsopp_hip_window_optimize } from the snapshot until exactly
 * `iterations_target` Gauss-Newton iterations have run (the last solve's max_iterations is capped accordingly, then the
 * window's setting is put back).  Same work as calling the two entry points in a loop, without the caller's per-call
 * overhead between solves; every solve still ends with its own read-back + stream synchronisation. */
int dsopp_hip_window_optimize_repeated(dsopp_hip_window *w, int32_t iterations_target, int32_t *iterations_done, double *last_energy);

/* device-side snapshot / restore of the mutable solver state (poses, affine, idepths, flags, connection statuses);
 * device-to-device copies only.  Lets a caller re-run a solve from the same starting point (benchmark loops, the
 * tracker's re-tracking tries) without re-uploading the window. */
int dsopp_hip_window_snapshot(dsopp_hip_window *w);
int dsopp_hip_window_restore(dsopp_hip_window *w);

/* timing / introspection for bench.py */
int dsopp_hip_window_last_solve_ms(dsopp_hip_window *w, float *ms); /* HIP-event time of the last LM loop */
enum {
  DSOPP_HIP_KERNEL_PAIR_SETUP = 0,
  DSOPP_HIP_KERNEL_FEJ,
  DSOPP_HIP_KERNEL_SWEEP_LINEARIZE,
  DSOPP_HIP_KERNEL_SWEEP_ENERGY,
  DSOPP_HIP_KERNEL_SCHUR,
  DSOPP_HIP_KERNEL_ASSEMBLE,
  DSOPP_HIP_KERNEL_ASSEMBLE_SOLVE,
  DSOPP_HIP_KERNEL_BACKSUB,
  DSOPP_HIP_KERNEL_ENERGY_REDUCE,
  DSOPP_HIP_KERNEL_ACCEPT,
  DSOPP_HIP_KERNEL_SWEEP_LINEARIZE_LOOP, /* the linearisation sweep as the fused LM loop runs it: back-substitution of the pending step
                                          * (calculateIdepths) + linearisation at the candidate state in one pass */
  DSOPP_HIP_NUM_KERNEL_CLASSES
};
/* when enabled every kernel launch is bracketed by HIP events on the window's stream; get_profile returns the summed
 * device time and launch count of one kernel class since profiling was (re-)enabled */
int dsopp_hip_window_set_profiling(dsopp_hip_window *w, int enable);
int dsopp_hip_window_get_profile(dsopp_hip_window *w, int kernel_class, double *total_ms, int64_t *launches);
const char *dsopp_hip_kernel_class_name(int kernel_class);
/* average duration (microseconds) of `repeats` back-to-back launches of one kernel class at the window's current state,
 * bracketed by ONE pair of HIP events on the window's stream (amortises the ~5 us an event pair costs around a single
 * short kernel).  Supported: SWEEP_LINEARIZE, SWEEP_LINEARIZE_LOOP, SWEEP_ENERGY, SCHUR, ASSEMBLE_SOLVE.  The window state is
 * left unchanged. */
int dsopp_hip_window_time_kernel(dsopp_hip_window *w, int kernel_class, int repeats, double *avg_us);

/* ---- reference depth maps of the newest keyframe (row a21) ----
 * createReferenceDepthMaps (src/tracker/tracker/src/create_depth_maps.cpp:124-147) on the device, from the window's own
 * state: the active landmarks of every older keyframe (connection status kOk towards the newest keyframe, not outlier, not
 * marginalized — :36-38) are reprojected into the newest keyframe and splatted with weight sqrt(1e-3 / (variance + 1e-12))
 * (:51-53; variance = H_dd^-1 of the last linearisation when estimate_uncertainty, else 1e-5 —
 * PROB_SRC/photometric_bundle_adjustment.cpp:252-254), sum-pooled to `levels` pyramid levels (:70-88) and dilated (:90-122).
 * The maps stay in HBM; dsopp_hip_aligner_push_reference_depth_maps hands a level to the tracker without a host round trip. */
/* (typedef struct dsopp_hip_depth_maps dsopp_hip_depth_maps: declared with the window group above) */
int dsopp_hip_window_create_reference_depth_maps(dsopp_hip_window *w, int32_t levels, dsopp_hip_depth_maps **out);
void dsopp_hip_depth_maps_destroy(dsopp_hip_depth_maps *m);
int dsopp_hip_depth_maps_level_size(const dsopp_hip_depth_maps *m, int32_t level, int32_t *width, int32_t *height);
/* copies one level to the host: two row-major H x W planes (energy::problem::DepthMap::map(x, y).{idepth, weight}) */
/* the same into an existing object of the same image size (the tracker keeps ONE reference_frame_depth_map_ and reassigns it
 * after every keyframe, monocular_tracker.cpp:465,509): no allocation, cached reference points are invalidated */
int dsopp_hip_window_refill_reference_depth_maps(dsopp_hip_window *w, dsopp_hip_depth_maps *maps);
/* calculateMeanSquareOpticalFlow (src/tracker/tracker/src/monocular_tracker.cpp:104-134) of one level of the device-resident
 * maps — the parallax measure the keyframe strategy reads for every tracked frame (:474-479: once for t_t_r, once for t_t_r
 * with the rotation removed) — for n_transforms <= 4 relative poses T_target_reference (7 each) in one pass.
 * flow[i] = sqrt(mean |bearing(pixel) - bearing(reprojection)|^2) over the map's pixels that reproject (NaN for none). */
int dsopp_hip_depth_maps_mean_square_optical_flow(const dsopp_hip_depth_maps *m, int32_t level, const double intrinsics[4], int32_t n_transforms,
                                                  const double *T_target_reference, double *flow);
int dsopp_hip_depth_maps_get_level(const dsopp_hip_depth_maps *m, int32_t level, double *idepth_sum, double *weight);

/* ---- depth estimation of immature landmarks (row f-1) ----
 * DepthEstimation::estimate (src/tracker/depth_estimators/src/depth_estimation.cpp:363-381) for the immature landmarks of one
 * keyframe against level `level` (0 in the tracker, monocular_tracker.cpp:98-100) of a new frame's device pyramid:
 * epipolar segment (epipolar_line_builder_pinhole_se3.hpp:296-372), discrete search (findBest, :36-76), sub-pixel
 * refinement on the epipolar tangent (refine, :184-221), uniqueness, error model and re-triangulated [idepth_min, idepth_max]
 * (estimateLandmark, :223-357).  T_target_reference = T_new_frame^-1 * T_keyframe.  The landmark arrays are the
 * struct-of-arrays view of track::landmarks::ImmatureTrackingLandmark and are updated in place:
 * status: 0 good, 1 out of boundary, 2 outlier, 3 skipped, 4 ill conditioned, 5 uninitialized, 6 delete
 * (immature_tracking_landmark.hpp:14-22).  One wavefront per landmark. */
/* the same on a device-resident landmark set (the immature landmarks of a keyframe persist over many frames: only their
 * estimator state changes).  create() initialises the state to the ImmatureTrackingLandmark constructor defaults;
 * estimate() is asynchronous on the set's stream; upload / download move the state (any pointer may be NULL). */
typedef struct dsopp_hip_immature_set dsopp_hip_immature_set;
int dsopp_hip_immature_set_create(int device, void *stream, int32_t n, const double *projection, const double *direction, const double *patch,
                                  const double *gradient, dsopp_hip_immature_set **out);
void dsopp_hip_immature_set_destroy(dsopp_hip_immature_set *s);
int dsopp_hip_immature_set_upload_state(dsopp_hip_immature_set *s, const double *idepth_min, const double *idepth_max, const double *uniqueness,
                                        const double *search_pixel_interval, const uint8_t *status, const uint8_t *traced);
int dsopp_hip_immature_set_download_state(dsopp_hip_immature_set *s, double *idepth_min, double *idepth_max, double *uniqueness,
                                          double *search_pixel_interval, uint8_t *status, uint8_t *traced);
int dsopp_hip_immature_set_estimate(dsopp_hip_immature_set *s, const dsopp_hip_pyramid *target_pyramid, int level, const double intrinsics[4],
                                    const double T_target_reference[7], double reference_exposure, const double reference_affine[2],
                                    double target_exposure, const double target_affine[2], double sigma_huber_loss);
/* estimateDepths of the tracker (monocular_tracker.cpp:74-102: the estimator runs for EVERY keyframe of the window on every
 * frame) as one call and one launch over n_sets sets: T_target_reference 7 per set, reference_exposure 1 per set,
 * reference_affine 2 per set.  Runs on the first set's stream; asynchronous when all sets share it. */
int dsopp_hip_immature_sets_estimate(int32_t n_sets, dsopp_hip_immature_set *const *sets, const dsopp_hip_pyramid *target_pyramid, int level,
                                     const double intrinsics[4], const double *T_target_reference, const double *reference_exposure,
                                     const double *reference_affine, double target_exposure, const double target_affine[2],
                                     double sigma_huber_loss);
/* one-shot form over host arrays (temporary set: upload, estimate, download) */
int dsopp_hip_estimate_depths(const dsopp_hip_pyramid *target_pyramid, int level, const double intrinsics[4],
                              const double T_target_reference[7], double reference_exposure, const double reference_affine[2],
                              double target_exposure, const double target_affine[2], double sigma_huber_loss, int32_t n,
                              const double *projection /* 2n */, const double *direction /* 3n */, const double *patch /* 8n */,
                              const double *gradient /* 2n */, double *idepth_min, double *idepth_max, double *uniqueness,
                              double *search_pixel_interval, uint8_t *status, uint8_t *traced);

/* ---- activation of immature landmarks (row f-3) ----
 * LandmarksActivator<SE3, PinholeCamera, PixelMap, 1, REFINE>::activate (src/tracker/landmarks_activator/src/landmarks_activator.cpp:351-391),
 * called once per new keyframe after pushNewKeyframe and before the keyframe enters the bundle adjustment
 * (monocular_tracker.cpp:491-497).  track.activeFrames() = the listed window keyframes (oldest first; poses, affine
 * brightness, exposure, level-0 images and active landmarks are the window's own) + the newest keyframe given explicitly.
 *   1. reprojectActivePoints (:51-87) into the newest keyframe at pyramid level 1, number_of_active_points;
 *   2. recalculateMinDistanceToNeighbor (:29-39): *min_distance_to_neighbor is LandmarksActivator::min_distance_to_neighbor_ (in/out);
 *   3. activationStatus (:89-126) of every immature landmark in keyframe / landmark order, the greedy sparsity test
 *      haveNoNeighbors (:41-49) through a uniform grid (identical result to the sequential O(n^2) loop);
 *   4. refine != 0: optimizeImmatureLandmark (:279-311), a 3-iteration LM on the inverse depth over all other keyframes.
 * activation_status[k][i]: 0 activate, 1 skip, 2 delete (ActiveKeyframe::ImmatureLandmarkActivationStatus, active_keyframe.hpp:40-44);
 * idepth[k][i] = landmark.idepth() after the call (refined for activated landmarks).  Either array (or an entry) may be NULL.
 * The sets are updated the way applyImmatureLandmarkActivationStatuses (active_keyframe.cpp:209-239) leaves the immature
 * landmarks: idepth_min = idepth_max = refined value for activated ones, status := delete for activated and deleted ones.
 * Creating the ActiveTrackingLandmark objects (and dsopp_hip_window_set_landmarks for them) stays with the caller. */
typedef struct dsopp_hip_activation_result {
  int32_t number_of_active_points;
  int32_t n_activated, n_skipped, n_deleted;
  int32_t selection_rounds; /* parallel rounds the greedy selection needed */
  double min_distance_to_neighbor;
} dsopp_hip_activation_result;
int dsopp_hip_window_activate_landmarks(dsopp_hip_window *w, int32_t n_keyframes, const int32_t *frame_ids,
                                        dsopp_hip_immature_set *const *immature /* n_keyframes, entries may be NULL */,
                                        const dsopp_hip_pyramid *newest_pyramid, const double T_world_newest[7], double exposure_newest,
                                        const double affine_newest[2], int32_t number_of_desired_points, double *min_distance_to_neighbor,
                                        int32_t refine, double sigma_huber_loss, uint8_t *const *activation_status, double *const *idepth,
                                        dsopp_hip_activation_result *result);

/* ------------------------------------------------------------------------------------------------------------------
 * Two-frame direct image alignment of one pyramid level
 * (replaces EigenPoseAlignment<SE3, PinholeCamera, 1, PixelMap, 1, true>, PROB_SRC/eigen_pose_alignment.cpp:26-329)
 * ---------------------------------------------------------------------------------------------------------------- */
typedef struct dsopp_hip_aligner dsopp_hip_aligner;

typedef struct dsopp_hip_align_result {
  double rmse; /* sqrt(E / n_valid / PatternSize) — eigen_pose_alignment.cpp:328; -1 (kZeroCost) for a known pose */
  double energy;
  int32_t n_valid;
  int32_t iterations;
  double T_world_target[7];
  double affine_brightness[2];
  double covariance[36]; /* tTargetReferenceCovariance — eigen_pose_alignment.cpp:320-323 */
  double H[64];
} dsopp_hip_align_result;

int dsopp_hip_aligner_create(const dsopp_hip_options *options, int device, void *stream, dsopp_hip_aligner **out);
void dsopp_hip_aligner_destroy(dsopp_hip_aligner *a);
/* reset() — PA_INC/pose_alignment.hpp, eigen_pose_alignment.cpp:261-264 */
int dsopp_hip_aligner_reset(dsopp_hip_aligner *a);
/* pushFrame(timestamp, pose, pyramids, masks, depth maps, ...) for the fixed reference frame
 * (PROB_SRC/photometric_bundle_adjustment.cpp:58-74 + LocalFrame depth-map ctor PBA_INT/local_frame.hpp:350-393): every
 * pixel with weight > 0 and idepth_sum/weight >= 1e-6 inside the 4-px border becomes a reference point whose intensity
 * is sampled from the reference pyramid level on the device.  depth maps are H_l x W_l row-major host arrays. */
int dsopp_hip_aligner_push_reference_depth_map(dsopp_hip_aligner *a, int64_t timestamp, const double T_world_agent[7],
                                               const dsopp_hip_pyramid *pyramid, int level, const double intrinsics[4],
                                               const double *idepth_sum, const double *weight, double exposure_time,
                                               const double affine_brightness[2]);
/* same, from device-resident maps (dsopp_hip_window_create_reference_depth_maps): the scan / compaction of the LocalFrame
 * depth-map constructor runs on the device and keeps the reference's row-major point order */
int dsopp_hip_aligner_push_reference_depth_maps(dsopp_hip_aligner *a, int64_t timestamp, const double T_world_agent[7],
                                                const dsopp_hip_pyramid *pyramid, int level, const double intrinsics[4],
                                                const dsopp_hip_depth_maps *maps, double exposure_time, const double affine_brightness[2]);
/* same with an explicit point list (u, v, idepth); intensity sampled on the device */
int dsopp_hip_aligner_push_reference_points(dsopp_hip_aligner *a, int64_t timestamp, const double T_world_agent[7],
                                            const dsopp_hip_pyramid *pyramid, int level, const double intrinsics[4], int32_t n,
                                            const double *u, const double *v, const double *idepth, double exposure_time,
                                            const double affine_brightness[2]);
/* pushFrame(timestamp, initial pose, pyramids, masks, ...) for the free target frame (photometric_bundle_adjustment.cpp:131-154) */
int dsopp_hip_aligner_push_target(dsopp_hip_aligner *a, int64_t timestamp, const double T_world_agent_init[7],
                                  const dsopp_hip_pyramid *pyramid, int level, const double intrinsics[4], double exposure_time,
                                  const double affine_brightness[2]);
/* pushKnownPose — eigen_pose_alignment.cpp:268-271 */
int dsopp_hip_aligner_push_known_pose(dsopp_hip_aligner *a, int64_t timestamp, const double T_world_agent[7]);
/* setRotationPrior — eigen_pose_alignment.cpp:254-257: the rotation of t_target_reference is replaced by fitToSO3(R) (3x3
 * row-major, det > 0) at the start of solve (:309-311); reset() clears it (:262), as does a NULL pointer */
int dsopp_hip_aligner_set_rotation_prior(dsopp_hip_aligner *a, const double *R_target_reference);
/* solve -> rmse (or kZeroCost = -1) — eigen_pose_alignment.cpp:275-329 */
int dsopp_hip_aligner_solve(dsopp_hip_aligner *a, dsopp_hip_align_result *result);
/* initializationPoses (src/tracker/tracker/src/monocular_tracker.cpp:136-176): the pose hypotheses estimatePose tries in turn —
 * previous motion, doubled, halved (exp(log/2)), zero motion, the keyframe's pose, then the previous motion perturbed by
 * rotations of 1, 1.5, 2, 2.5 degrees about every axis combination: 5 + 4 * 27 = 113 poses.  T_world_previous / T_world_last are
 * track.getFrame(-2) / (-1); any of the three NULL = fewer than two frames in the track -> the single identity pose.
 * Host arithmetic (no device work); *n receives the count, at most `capacity` poses (7 doubles each) are written. */
int dsopp_hip_initialization_poses(const double T_world_previous[7], const double T_world_last[7], const double T_world_keyframe[7],
                                   int32_t capacity, double *poses, int32_t *n);
/* Coarse-to-fine pose estimation of a new frame against the last keyframe: estimatePose of the tracker
 * (src/tracker/tracker/src/monocular_tracker.cpp:179-245).  For every initialisation in turn (until one succeeds): from the
 * coarsest level of the target pyramid down to level 0 { reset; push the keyframe's depth map of that level; push the target
 * with the current estimate; solve; accept the level when rmse < 2.5 * rmse_last_pose_estimation[level] }.  The camera model
 * of level l is the level-0 one scaled by 2^-l.  rmse_last_pose_estimation (one per level) is updated as the reference
 * does (multiplied by 2.5 when every initialisation failed; then the result of the FIRST initialisation is returned). */
int dsopp_hip_aligner_estimate_pose(dsopp_hip_aligner *a, int64_t reference_time, const double T_world_reference[7],
                                    const dsopp_hip_pyramid *reference_pyramid, const dsopp_hip_depth_maps *reference_depth_maps,
                                    double reference_exposure, const double reference_affine[2], int64_t target_time,
                                    const dsopp_hip_pyramid *target_pyramid, double target_exposure, const double intrinsics[4],
                                    int32_t n_initializations, const double *T_world_target_init, const double affine_init[2],
                                    double *rmse_last_pose_estimation, double T_world_target[7], double affine_brightness[2],
                                    int32_t *success, int32_t *tries, int32_t *lm_iterations);
/* Initialisations estimate_pose evaluates per launch.  The reference tries them one after the other until one passes its per-level
 * energy gates (monocular_tracker.cpp:193-243); they are independent of each other, so up to 8 run concurrently — one per XCD of the
 * device, each with its own exchange buffers — and the lowest-index success is taken: pose, `tries`, `lm_iterations` and the updated
 * rmse_last_pose_estimation are those of the sequential loop.  0 (default): automatic — the first initialisation alone while
 * tracking holds (it succeeds; nothing is computed in vain), 8 at a time as soon as a first try failed in this or the previous call
 * (re-localisation: up to 113 initialisations); 1: strictly one per launch; 2 .. 8: that many per launch from the first one on. */
int dsopp_hip_aligner_set_hypothesis_width(dsopp_hip_aligner *a, int32_t width);
/* 0 (default): the whole LM loop of a solve runs in ONE launch of one workgroup when the reference has at most 16384 points
 * (a launch per iteration costs more than the iteration at that size); 1: always one launch per LM iteration (multi-workgroup).
 * Same state machine and arithmetic on both paths (parity cross-check in tests/test_gpu_tracker.py). */
int dsopp_hip_aligner_set_lm_path(dsopp_hip_aligner *a, int path);
int dsopp_hip_aligner_num_points(dsopp_hip_aligner *a, int32_t *n);

#ifdef __cplusplus
}
#endif
#endif /* DSOPP_HIP_H */
