/* ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
 *
 * C interface of the CPU oracle (restatement of the reference's src/energy + src/features hot path, see the headers of
 * pba.hpp / pose_alignment.hpp / pyramid.hpp for the file:line citations).  It deliberately has the same shape as the
 * product's C-ABI (include/dsopp_hip.h) so tests can drive both with identical call sequences and compare outputs.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * PARITY UNPINNED: no reference build / golden vectors exist for this path in this environment. */
#ifndef DSOPP_ORACLE_H
#define DSOPP_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
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
} orc_options;

void orc_default_pba_options(orc_options *o);
void orc_default_align_options(orc_options *o);
void orc_set_threads(int n);
int orc_get_threads(void);

/* ---- sliding-window photometric bundle adjustment ---- */
typedef struct orc_window orc_window;
orc_window *orc_window_create(const orc_options *o);
void orc_window_destroy(orc_window *w);
/* image / mask buffers are BORROWED (the reference keeps raw pointers to the keyframe's PixelMap, local_frame.hpp:323-325) */
int orc_window_push_frame(orc_window *w, int frame_id, int64_t timestamp, int width, int height, const double *pixelinfo,
                          const uint8_t *mask, const double intrinsics[4], const double T_w_agent[7], double exposure_time,
                          const double affine_brightness[2], int fixed, int is_marginalized);
/* flags bit0 = is_marginalized, bit1 = is_outlier.  n_total >= current count: existing landmarks get their
 * marginalisation flags updated (LocalFrame::update, local_frame.hpp:484-505), new ones are appended */
int orc_window_set_landmarks(orc_window *w, int frame_id, int n_total, const double *uv, const double *idepth,
                             const double *patch, const uint8_t *flags);
/* appends ResidualPoints for landmarks [current size, n) of the (ref, tgt) connection with the given statuses */
int orc_window_set_connection(orc_window *w, int ref_id, int tgt_id, int n, const uint8_t *statuses);
int orc_window_mark_frame_marginalized(orc_window *w, int frame_id);
int orc_window_num_frames(orc_window *w);

/* stage-level API (PhotometricBundleAdjustmentProblem methods) */
int orc_window_begin(orc_window *w); /* construct the problem (+ firstEstimateJacobians when FEJ) */
int orc_window_calculate_energy(orc_window *w, double *energy, int *n_valid);
int orc_window_linearize(orc_window *w);
int orc_window_get_system(orc_window *w, double *H_pp, double *b_pp, double *H_schur, double *b_schur);
int orc_window_calculate_step(orc_window *w, double lambda, double *step);
int orc_window_accept_step(orc_window *w, double *state_sq, double *step_sq);
int orc_window_reject_step(orc_window *w);
int orc_window_update_point_statuses(orc_window *w);
/* whole solve (EigenPhotometricBundleAdjustment::solve) */
int orc_window_solve(orc_window *w, double *energy, int *iterations, int *n_valid);
/* LM loop only (no relinearisation / covariance / statuses) */
int orc_window_optimize(orc_window *w, double *energy, int *iterations, int *n_valid);
int orc_window_reset_state(orc_window *w, int frame_id, const double T_w_agent[7], const double ab[2], const double *idepth);

/* getters */
int orc_window_get_frame_state(orc_window *w, int frame_id, double T0[7], double ab0[2], double eps[8], double step[8]);
int orc_window_get_pose(orc_window *w, int frame_id, double T_w_agent[7], double affine_brightness[2]);
int orc_window_num_landmarks(orc_window *w, int frame_id);
/* per landmark: idepth, idepth_step, inv_hessian_idepth_idepth, b_idepth_block, relative_baseline (any may be NULL);
 * flags_out bit0 marginalized, bit1 outlier, bit2 to_marginalize, bit3 ill_conditioned; hpib is n x K */
int orc_window_get_landmarks(orc_window *w, int frame_id, double *idepth, double *idepth_step, double *inv_hdd, double *b_d,
                             double *relative_baseline, int32_t *n_inliers, uint8_t *flags_out, double *hpib);
int orc_window_get_residuals(orc_window *w, int ref_id, int tgt_id, uint8_t *status, uint8_t *candidate, double *energy,
                             double *huber_weight, double *residuals8, double *J_ref64, double *J_tgt64, double *J_idepth8);
int orc_window_get_marginalized(orc_window *w, double *H, double *b, double *energy);
int orc_window_get_covariance(orc_window *w, int ref_id, int tgt_id, double cov[36]);

/* ---- two-frame direct alignment (one pyramid level) ---- */
typedef struct {
  double rmse;
  double energy;
  int32_t n_valid;
  int32_t iterations;
  double T_w_target[7];
  double affine_brightness[2];
  double covariance[36];
  double H[64];
} orc_align_result;
/* reference points from a depth map (LocalFrame depth-map ctor): returns count, fills up to cap entries */
int orc_points_from_depth_map(int width, int height, const double *pixelinfo, const double *idepth_sum, const double *weight,
                              int cap, double *u, double *v, double *idepth, double *intensity);
int orc_align_solve(const orc_options *o, int n, const double *u, const double *v, const double *idepth,
                    const double *intensity, const double ref_intrinsics[4], int ref_width, int ref_height,
                    const double T_w_ref[7], double ref_exposure, const double ref_ab[2], const double tgt_intrinsics[4],
                    int tgt_width, int tgt_height, const double *tgt_pixelinfo, const uint8_t *tgt_mask,
                    const double T_w_tgt_init[7], double tgt_exposure, const double tgt_ab[2], orc_align_result *out);
/* the same with setRotationPrior (eigen_pose_alignment.cpp:254-257,309-311): the rotation of t_t_r is replaced by
 * fitToSO3(prior_rotation_t_r) (3x3 row-major, nullable) before the solve */
int orc_align_solve_with_prior(const orc_options *o, int n, const double *u, const double *v, const double *idepth,
                               const double *intensity, const double ref_intrinsics[4], int ref_width, int ref_height,
                               const double T_w_ref[7], double ref_exposure, const double ref_ab[2], const double tgt_intrinsics[4],
                               int tgt_width, int tgt_height, const double *tgt_pixelinfo, const uint8_t *tgt_mask,
                               const double T_w_tgt_init[7], double tgt_exposure, const double tgt_ab[2],
                               const double *prior_rotation_t_r, orc_align_result *out);

/* createReferenceDepthMaps (create_depth_maps.cpp:18-147) over plain arrays: n_sources older keyframes, source s holds
 * counts[s] landmarks starting at offsets into the concatenated arrays (uv 2 per landmark).  Outputs: per level row-major
 * H_l x W_l planes, concatenated level after level (level sizes halve, floor). */
int orc_create_reference_depth_maps(int n_sources, const double *T_w_sources /* 7 each */, const int32_t *counts, const double *uv,
                                    const double *idepth, const double *variance, const uint8_t *skip, const uint8_t *status,
                                    const double T_w_newest[7], const double intrinsics[4], int width, int height, int levels,
                                    double *idepth_sum_out, double *weight_out);

/* DepthEstimation::estimate (depth_estimation.cpp:363-381) for the immature landmarks of one keyframe against a new frame.
 * Landmark arrays are in/out (struct-of-arrays view of ImmatureTrackingLandmark). */
int orc_estimate_depths(int width, int height, const double *target_pixelinfo, const uint8_t *mask, const double intrinsics[4],
                        const double T_target_reference[7], double reference_exposure, const double reference_affine[2],
                        double target_exposure, const double target_affine[2], double sigma_huber_loss, int n,
                        const double *projection, const double *direction, const double *patch, const double *gradient,
                        double *idepth_min, double *idepth_max, double *uniqueness, double *search_pixel_interval,
                        uint8_t *status, uint8_t *traced);
/* LandmarksActivator::activate (landmarks_activator.cpp:351-391) + the immature-landmark side of
 * applyImmatureLandmarkActivationStatuses.  Frames oldest first, the newest keyframe last (it has no landmarks yet); all
 * per-landmark arrays are the concatenation over frames 0..n_frames-2.  idepth_min/max and status are in/out;
 * activation_status: 0 activate, 1 skip, 2 delete.  Returns number_of_active_points. */
int orc_activate_landmarks(int n_frames, int width, int height, const double *const *pixelinfo, const uint8_t *const *mask0,
                           const uint8_t *mask_sparsity_newest, const double *T_w, const double *exposure, const double *affine,
                           const double intrinsics[4], const int32_t *n_active, const double *active_uv, const double *active_idepth,
                           const uint8_t *active_skip, const int32_t *n_immature, const double *projection, const double *patch,
                           double *idepth_min, double *idepth_max, const double *uniqueness, const double *search_pixel_interval,
                           uint8_t *status, const uint8_t *traced, double sigma_huber_loss, int number_of_desired_points,
                           double *min_distance_to_neighbor, int refine, uint8_t *activation_status);
/* initializationPoses (monocular_tracker.cpp:136-176): returns the number of hypotheses, fills up to cap (7 doubles each) */
int orc_initialization_poses(int have_two_frames, const double T_w_previous[7], const double T_w_last[7], const double T_w_keyframe[7],
                             int cap, double *poses);
void orc_se3_log(const double T[7], double xi[6]);
/* calculateMeanSquareOpticalFlow (monocular_tracker.cpp:104-134) of one depth-map level (row-major H x W planes) */
double orc_mean_square_optical_flow(int width, int height, const double *idepth_sum, const double *weight, const double intrinsics[4],
                                    const double T_target_reference[7]);
/* EpipolarLineBuilder::buildSegment: returns the number of points, fills up to cap (projection 2 each, reference idepth) */
int orc_build_epipolar_segment(int width, int height, const double intrinsics[4], const double T_target_reference[7],
                               const double observed[2], double idepth_min, double idepth_max, int cap, double *projections,
                               double *idepths);

/* ---- pyramid ---- */
/* pixelinfo_out[l] must hold 3*w_l*h_l doubles; plane_out[l] (optional) w_l*h_l */
int orc_build_pyramid(const uint8_t *image, int width, int height, const double *lut256, const uint8_t *vignetting,
                      int levels, double **pixelinfo_out, double **plane_out);

/* ---- primitives exposed for identity tests ---- */
void orc_se3_exp(const double xi[6], double T[7]);
void orc_se3_mul(const double A[7], const double B[7], double C[7]);
void orc_se3_inverse(const double A[7], double B[7]);
void orc_se3_adj(const double A[7], double adj36[36]);
int orc_reproject_pattern(const double ref_intr[4], int ref_w, int ref_h, const double tgt_intr[4], int tgt_w, int tgt_h,
                          const double T_t_r[7], int n, const double *u, const double *v, double idepth, int with_jacobians,
                          double *tu, double *tv, double *d_u_idepth, double *d_v_idepth, double *d_u_T, double *d_v_T);
int orc_solve_system(int n, const double *H, const double *b, double *x);
int orc_reduce_system(int n, double *H, double *b, int n_elim, const int32_t *elim, double *H_out, double *b_out);
int orc_pinv_drop(int n, const double *H, int nullspaces, double *out);
/* covariance of the relative pose of two frames from the blocks of their joint covariance — se3_motion.hpp:151-158 */
int orc_relative_transformation_uncertainty(const double T_w_1[7], const double T_w_2[7], const double *sigma_11, const double *sigma_22,
                                            const double *sigma_12, double *out36);
/* third-party restatements on their own (tests/test_oracle_thirdparty.py): Eigen::LDLT solve (normal_linear_system.cpp:57),
 * completeOrthogonalDecomposition().pseudoInverse() (:35-37), PixelMap bilinear sampler (pixel_map.hpp:20-40), CameraMask lookup at the
 * rounded position (camera_mask.hpp:64-66) */
int orc_ldlt_solve(int n, const double *A, const double *b, double *x);
int orc_pinv_cod(int n, const double *H, double *out);
int orc_interpolate_linear(int width, int height, const double *pixelinfo, int n, const double *x, const double *y, double *out3);
int orc_mask_valid(int width, int height, const uint8_t *mask, int n, const double *x, const double *y, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
