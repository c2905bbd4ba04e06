// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
// C interface over the CPU restatement (see oracle.h).
#include "oracle.h"

#include <cstring>
#include <memory>

#include "pba.hpp"
#include "pose_alignment.hpp"
#include "depth_maps.hpp"
#include "depth_estimation.hpp"
#include "landmarks_activator.hpp"
#include "pyramid.hpp"

using namespace oracle;

struct orc_window {
  PbaWindow win;
  std::unique_ptr<PbaProblem> problem;
};

static PbaOptions toOptions(const orc_options *o) {
  PbaOptions p;
  p.max_iterations = o->max_iterations;
  p.initial_trust_region_radius = o->initial_trust_region_radius;
  p.function_tolerance = o->function_tolerance;
  p.parameter_tolerance = o->parameter_tolerance;
  p.affine_brightness_regularizer[0] = o->affine_brightness_regularizer[0];
  p.affine_brightness_regularizer[1] = o->affine_brightness_regularizer[1];
  p.fixed_state_regularizer = o->fixed_state_regularizer;
  p.sigma_huber_loss = o->sigma_huber_loss;
  p.estimate_uncertainty = o->estimate_uncertainty != 0;
  p.force_accept = o->force_accept != 0;
  p.first_estimate_jacobians = o->first_estimate_jacobians != 0;
  p.optimize_idepths = o->optimize_idepths != 0;
  return p;
}

extern "C" {

void orc_default_pba_options(orc_options *o) {
  // createPhotometricBundleAdjustment — src/tracker/tracker/src/fabric.cpp:63-79
  o->max_iterations = 7;
  o->initial_trust_region_radius = 1e5;
  o->function_tolerance = 1e-8;
  o->parameter_tolerance = 1e-8;
  o->affine_brightness_regularizer[0] = 1e12;
  o->affine_brightness_regularizer[1] = 1e8;
  o->fixed_state_regularizer = 1e16;
  o->sigma_huber_loss = 20;
  o->estimate_uncertainty = 1;
  o->force_accept = 1;
  o->first_estimate_jacobians = 1;
  o->optimize_idepths = 1;
}
void orc_default_align_options(orc_options *o) {
  // createPoseAlignment — fabric.cpp:127-142
  orc_default_pba_options(o);
  o->max_iterations = 50;
  o->initial_trust_region_radius = 1e2;
  o->function_tolerance = 1e-5;
  o->parameter_tolerance = 1e-5;
  o->force_accept = 0;
}
void orc_set_threads(int n) { ThreadPool::instance().setThreads(n); }
int orc_get_threads(void) { return ThreadPool::instance().threads(); }

orc_window *orc_window_create(const orc_options *o) {
  auto *w = new orc_window();
  w->win.opt = toOptions(o);
  return w;
}
void orc_window_destroy(orc_window *w) { delete w; }

int orc_window_push_frame(orc_window *w, int frame_id, int64_t timestamp, int width, int height, const double *pixelinfo,
                          const uint8_t *mask, const double intrinsics[4], const double T_w_agent[7], double exposure_time,
                          const double affine_brightness[2], int fixed, int is_marginalized) {
  auto f = std::make_unique<LocalFrame>();
  f->id = frame_id;
  f->timestamp = timestamp;
  f->T_w_agent_linearization_point = SE3::fromParams(T_w_agent);
  f->exposure_time = exposure_time;
  f->affine_brightness0[0] = affine_brightness[0];
  f->affine_brightness0[1] = affine_brightness[1];
  f->model.width = width;
  f->model.height = height;
  f->model.fx = intrinsics[0];
  f->model.fy = intrinsics[1];
  f->model.cx = intrinsics[2];
  f->model.cy = intrinsics[3];
  f->grid = PixelMapView{pixelinfo, width, height};
  f->mask = MaskView{mask, width, height};
  f->fixed = fixed != 0;
  f->is_marginalized = is_marginalized != 0;
  w->problem.reset();
  return w->win.pushFrame(std::move(f));
}

int orc_window_set_landmarks(orc_window *w, int frame_id, int n_total, const double *uv, const double *idepth,
                             const double *patch, const uint8_t *flags) {
  LocalFrame *f = w->win.getLocalFrame(frame_id);
  if (!f) return -2;
  const size_t old = f->active_landmarks.size();
  if (static_cast<size_t>(n_total) < old) return -1;
  for (size_t i = 0; i < old; ++i) {
    Landmark &lm = f->active_landmarks[i];
    const bool marg = flags[i] & 1, outl = flags[i] & 2;
    lm.to_marginalize = !lm.is_marginalized && marg && !outl;
    lm.is_marginalized = marg;
  }
  for (size_t i = old; i < static_cast<size_t>(n_total); ++i)
    f->active_landmarks.emplace_back(uv + 2 * i, idepth[i], patch + 8 * i, (flags[i] & 1) != 0, (flags[i] & 2) != 0);
  w->problem.reset();
  return 0;
}

int orc_window_set_connection(orc_window *w, int ref_id, int tgt_id, int n, const uint8_t *statuses) {
  LocalFrame *f = w->win.getLocalFrame(ref_id);
  if (!f) return -2;
  auto &res = f->residuals[tgt_id];
  for (size_t i = res.size(); i < static_cast<size_t>(n); ++i) res.emplace_back(statuses[i]);
  w->problem.reset();
  return 0;
}

int orc_window_mark_frame_marginalized(orc_window *w, int frame_id) {
  // EigenPhotometricBundleAdjustment::updateLocalFrame — eigen_photometric_bundle_adjustment.cpp:106-113
  LocalFrame *f = w->win.getLocalFrame(frame_id);
  if (!f) return -2;
  f->to_marginalize = !f->is_marginalized;
  f->is_marginalized = true;
  return 0;
}
int orc_window_num_frames(orc_window *w) { return static_cast<int>(w->win.frames.size()); }

int orc_window_begin(orc_window *w) {
  w->problem = std::make_unique<PbaProblem>(w->win.frames, w->win.opt, w->win.system_marginalized, w->win.energy_marginalized);
  if (w->win.opt.first_estimate_jacobians) firstEstimateJacobians(w->win.frames);
  return 0;
}
int orc_window_calculate_energy(orc_window *w, double *energy, int *n_valid) {
  if (!w->problem) return -6;
  auto r = w->problem->calculateEnergy();
  *energy = r.first;
  *n_valid = r.second;
  return 0;
}
int orc_window_linearize(orc_window *w) {
  if (!w->problem) return -6;
  w->problem->linearize();
  return 0;
}
int orc_window_get_system(orc_window *w, double *H_pp, double *b_pp, double *H_schur, double *b_schur) {
  const NormalLinearSystem *sp = w->problem ? &w->problem->system_pose : &w->win.last_system_pose;
  const NormalLinearSystem *ss = w->problem ? &w->problem->system_schur : &w->win.last_system_schur;
  if (H_pp) std::memcpy(H_pp, sp->H.a.data(), sp->H.a.size() * sizeof(double));
  if (b_pp) std::memcpy(b_pp, sp->b.data(), sp->b.size() * sizeof(double));
  if (H_schur) std::memcpy(H_schur, ss->H.a.data(), ss->H.a.size() * sizeof(double));
  if (b_schur) std::memcpy(b_schur, ss->b.data(), ss->b.size() * sizeof(double));
  return 0;
}
int orc_window_calculate_step(orc_window *w, double lambda, double *step) {
  if (!w->problem) return -6;
  w->problem->calculateStep(lambda);
  if (step) std::memcpy(step, w->problem->last_step.data(), w->problem->last_step.size() * sizeof(double));
  return 0;
}
int orc_window_accept_step(orc_window *w, double *state_sq, double *step_sq) {
  if (!w->problem) return -6;
  auto r = w->problem->acceptStep();
  if (state_sq) *state_sq = r.first;
  if (step_sq) *step_sq = r.second;
  return 0;
}
int orc_window_reject_step(orc_window *w) {
  if (!w->problem) return -6;
  w->problem->rejectStep();
  return 0;
}
int orc_window_update_point_statuses(orc_window *w) {
  updatePointStatuses(w->win.frames, 1, w->win.opt.sigma_huber_loss);
  return 0;
}
int orc_window_solve(orc_window *w, double *energy, int *iterations, int *n_valid) {
  w->problem.reset();
  const double e = w->win.solve();
  if (energy) *energy = e;
  if (iterations) *iterations = w->win.last_result.iterations;
  if (n_valid) *n_valid = w->win.last_result.number_of_valid_residuals;
  return 0;
}

int orc_window_optimize(orc_window *w, double *energy, int *iterations, int *n_valid) {
  w->problem.reset();
  const double e = w->win.optimize();
  if (energy) *energy = e;
  if (iterations) *iterations = w->win.last_result.iterations;
  if (n_valid) *n_valid = w->win.last_result.number_of_valid_residuals;
  return 0;
}
/* overwrite the mutable solver state (poses as linearisation points with zero eps, idepths, statuses = kOk) so a
 * timing loop can restart from the same point */
int orc_window_reset_state(orc_window *w, int frame_id, const double T_w_agent[7], const double ab[2], const double *idepth) {
  LocalFrame *f = w->win.getLocalFrame(frame_id);
  if (!f) return -2;
  f->T_w_agent_linearization_point = SE3::fromParams(T_w_agent);
  f->affine_brightness0[0] = ab[0];
  f->affine_brightness0[1] = ab[1];
  for (double &v : f->state_eps) v = 0;
  for (double &v : f->state_eps_step) v = 0;
  for (size_t i = 0; i < f->active_landmarks.size(); ++i) {
    f->active_landmarks[i].idepth = idepth[i];
    f->active_landmarks[i].idepth_step = 0;
    f->active_landmarks[i].is_outlier = false;
  }
  for (auto &kv : f->residuals)
    for (auto &r : kv.second) r = ResidualPoint(kOk);
  return 0;
}

int orc_window_get_frame_state(orc_window *w, int frame_id, double T0[7], double ab0[2], double eps[8], double step[8]) {
  LocalFrame *f = w->win.getLocalFrame(frame_id);
  if (!f) return -2;
  if (T0) f->T_w_agent_linearization_point.toParams(T0);
  if (ab0) {
    ab0[0] = f->affine_brightness0[0];
    ab0[1] = f->affine_brightness0[1];
  }
  if (eps) std::memcpy(eps, f->state_eps, sizeof(f->state_eps));
  if (step) std::memcpy(step, f->state_eps_step, sizeof(f->state_eps_step));
  return 0;
}
int orc_window_get_pose(orc_window *w, int frame_id, double T_w_agent[7], double affine_brightness[2]) {
  LocalFrame *f = w->win.getLocalFrame(frame_id);
  if (!f) return -2;
  if (T_w_agent) f->tWorldAgent().toParams(T_w_agent);
  if (affine_brightness) f->affineBrightness(affine_brightness);
  return 0;
}
int orc_window_num_landmarks(orc_window *w, int frame_id) {
  LocalFrame *f = w->win.getLocalFrame(frame_id);
  return f ? static_cast<int>(f->active_landmarks.size()) : -2;
}
int orc_window_get_landmarks(orc_window *w, int frame_id, double *idepth, double *idepth_step, double *inv_hdd, double *b_d,
                             double *relative_baseline, int32_t *n_inliers, uint8_t *flags_out, double *hpib) {
  LocalFrame *f = w->win.getLocalFrame(frame_id);
  if (!f) return -2;
  const size_t K = 8 * w->win.frames.size();
  for (size_t i = 0; i < f->active_landmarks.size(); ++i) {
    const Landmark &lm = f->active_landmarks[i];
    if (idepth) idepth[i] = lm.idepth;
    if (idepth_step) idepth_step[i] = lm.idepth_step;
    if (inv_hdd) inv_hdd[i] = lm.inv_hessian_idepth_idepth;
    if (b_d) b_d[i] = lm.b_idepth_block;
    if (relative_baseline) relative_baseline[i] = lm.relative_baseline;
    if (n_inliers) n_inliers[i] = static_cast<int32_t>(lm.number_of_inlier_residuals);
    if (flags_out)
      flags_out[i] = static_cast<uint8_t>((lm.is_marginalized ? 1 : 0) | (lm.is_outlier ? 2 : 0) | (lm.to_marginalize ? 4 : 0) |
                                          (lm.ill_conditioned ? 8 : 0));
    if (hpib) {
      for (size_t k = 0; k < K; ++k)
        hpib[i * K + k] = k < lm.hessian_poses_idepth_block.size() ? lm.hessian_poses_idepth_block[k] : 0.0;
    }
  }
  return 0;
}
int orc_window_get_residuals(orc_window *w, int ref_id, int tgt_id, uint8_t *status, uint8_t *candidate, double *energy,
                             double *huber_weight, double *residuals8, double *J_ref64, double *J_tgt64, double *J_idepth8) {
  LocalFrame *f = w->win.getLocalFrame(ref_id);
  if (!f) return -2;
  auto it = f->residuals.find(tgt_id);
  if (it == f->residuals.end()) return -2;
  for (size_t i = 0; i < it->second.size(); ++i) {
    const ResidualPoint &r = it->second[i];
    if (status) status[i] = r.connection_status;
    if (candidate) candidate[i] = r.connection_status_candidate;
    if (energy) energy[i] = r.energy;
    if (huber_weight) huber_weight[i] = r.huber_weight;
    if (residuals8) std::memcpy(residuals8 + 8 * i, r.residuals, sizeof(r.residuals));
    if (J_ref64) std::memcpy(J_ref64 + 64 * i, r.d_reference_state_eps, sizeof(r.d_reference_state_eps));
    if (J_tgt64) std::memcpy(J_tgt64 + 64 * i, r.d_target_state_eps, sizeof(r.d_target_state_eps));
    if (J_idepth8) std::memcpy(J_idepth8 + 8 * i, r.d_idepth, sizeof(r.d_idepth));
  }
  return static_cast<int>(it->second.size());
}
int orc_window_get_marginalized(orc_window *w, double *H, double *b, double *energy) {
  const NormalLinearSystem &m = w->win.system_marginalized;
  if (H) std::memcpy(H, m.H.a.data(), m.H.a.size() * sizeof(double));
  if (b) std::memcpy(b, m.b.data(), m.b.size() * sizeof(double));
  if (energy) *energy = w->win.energy_marginalized;
  return m.size();
}
int orc_window_get_covariance(orc_window *w, int ref_id, int tgt_id, double cov[36]) {
  LocalFrame *f = w->win.getLocalFrame(ref_id);
  if (!f) return -2;
  auto it = f->covariance_matrices.find(tgt_id);
  if (it == f->covariance_matrices.end()) return -2;
  std::memcpy(cov, it->second.data(), 36 * sizeof(double));
  return 0;
}

int orc_points_from_depth_map(int width, int height, const double *pixelinfo, const double *idepth_sum, const double *weight,
                              int cap, double *u, double *v, double *idepth, double *intensity) {
  auto pts = pointsFromDepthMap(PixelMapView{pixelinfo, width, height}, idepth_sum, weight);
  const int n = static_cast<int>(pts.size());
  for (int i = 0; i < std::min(n, cap); ++i) {
    u[i] = pts[static_cast<size_t>(i)].u;
    v[i] = pts[static_cast<size_t>(i)].v;
    idepth[i] = pts[static_cast<size_t>(i)].idepth;
    intensity[i] = pts[static_cast<size_t>(i)].intensity;
  }
  return n;
}

int orc_estimate_depths(int width, int height, const double *target_pixelinfo, const uint8_t *mask, const double intrinsics[4],
                        const double T_target_reference[7], double reference_exposure, const double reference_affine[2],
                        double target_exposure, const double target_affine[2], double sigma_huber_loss, int n,
                        const double *projection, const double *direction, const double *patch, const double *gradient,
                        double *idepth_min, double *idepth_max, double *uniqueness, double *search_pixel_interval,
                        uint8_t *status, uint8_t *traced) {
  DepthEstimationFrame f;
  f.target = PixelMapView{target_pixelinfo, width, height};
  f.mask = MaskView{mask, width, height};
  f.model = PinholeModel{static_cast<double>(width), static_cast<double>(height), intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  f.t_t_r = SE3::fromParams(T_target_reference);
  f.reference_exposure_time = reference_exposure;
  f.target_exposure_time = target_exposure;
  for (int i = 0; i < 2; ++i) {
    f.reference_affine[i] = reference_affine[i];
    f.target_affine[i] = target_affine[i];
  }
  f.sigma_huber_loss = sigma_huber_loss;
  for (int i = 0; i < n; ++i) {
    ImmatureLandmark lm;
    for (int k = 0; k < 2; ++k) {
      lm.projection[k] = projection[2 * i + k];
      lm.gradient[k] = gradient[2 * i + k];
    }
    for (int k = 0; k < 3; ++k) lm.direction[k] = direction[3 * i + k];
    for (int k = 0; k < kPatternSize; ++k) lm.patch[k] = patch[kPatternSize * i + k];
    lm.idepth_min = idepth_min[i];
    lm.idepth_max = idepth_max[i];
    lm.uniqueness = uniqueness[i];
    lm.search_pixel_interval = search_pixel_interval[i];
    lm.status = status[i];
    lm.traced = traced[i] != 0;
    estimateLandmark(f, lm);
    idepth_min[i] = lm.idepth_min;
    idepth_max[i] = lm.idepth_max;
    uniqueness[i] = lm.uniqueness;
    search_pixel_interval[i] = lm.search_pixel_interval;
    status[i] = lm.status;
    traced[i] = lm.traced ? 1 : 0;
  }
  return n;
}

int orc_activate_landmarks(int n_frames, int width, int height, const double *const *pixelinfo, const uint8_t *const *mask0,
                           const uint8_t *mask_sparsity_newest, const double *T_w, const double *exposure, const double *affine,
                           const double intrinsics[4], const int32_t *n_active, const double *active_uv, const double *active_idepth,
                           const uint8_t *active_skip, const int32_t *n_immature, const double *projection, const double *patch,
                           double *idepth_min, double *idepth_max, const double *uniqueness, const double *search_pixel_interval,
                           uint8_t *status, const uint8_t *traced, double sigma_huber_loss, int number_of_desired_points,
                           double *min_distance_to_neighbor, int refine, uint8_t *activation_status) {
  std::vector<ActivationKeyframe> frames(static_cast<size_t>(n_frames));
  size_t oa = 0, oi = 0;
  for (int f = 0; f < n_frames; ++f) {
    ActivationKeyframe &k = frames[static_cast<size_t>(f)];
    k.t_world_agent = SE3::fromParams(T_w + 7 * f);
    k.exposure_time = exposure[f];
    k.affine_brightness[0] = affine[2 * f];
    k.affine_brightness[1] = affine[2 * f + 1];
    k.level0 = PixelMapView{pixelinfo[f], width, height};
    k.mask0 = MaskView{mask0 ? mask0[f] : nullptr, width, height};
    k.mask_sparsity = MaskView{f + 1 == n_frames ? mask_sparsity_newest : nullptr, width / 2, height / 2};
    if (f + 1 == n_frames) break;
    k.n_active = n_active[f];
    k.active_uv = active_uv + 2 * oa;
    k.active_idepth = active_idepth + oa;
    k.active_skip = active_skip + oa;
    oa += static_cast<size_t>(n_active[f]);
    for (int i = 0; i < n_immature[f]; ++i, ++oi) {
      ImmatureLandmark lm;
      lm.projection[0] = projection[2 * oi];
      lm.projection[1] = projection[2 * oi + 1];
      for (int c = 0; c < kPatternSize; ++c) lm.patch[c] = patch[kPatternSize * oi + c];
      lm.idepth_min = idepth_min[oi];
      lm.idepth_max = idepth_max[oi];
      lm.uniqueness = uniqueness[oi];
      lm.search_pixel_interval = search_pixel_interval[oi];
      lm.status = status[oi];
      lm.traced = traced[oi] != 0;
      k.immature.push_back(lm);
    }
  }
  const PinholeModel model{static_cast<double>(width), static_cast<double>(height), intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  const ActivationResult r = activateLandmarks(frames, model, sigma_huber_loss, static_cast<size_t>(number_of_desired_points),
                                               *min_distance_to_neighbor, refine != 0);
  oi = 0;
  for (int f = 0; f + 1 < n_frames; ++f)
    for (size_t i = 0; i < frames[static_cast<size_t>(f)].immature.size(); ++i, ++oi) {
      const ImmatureLandmark &lm = frames[static_cast<size_t>(f)].immature[i];
      idepth_min[oi] = lm.idepth_min;
      idepth_max[oi] = lm.idepth_max;
      status[oi] = lm.status;
      activation_status[oi] = r.statuses[static_cast<size_t>(f)][i];
    }
  return static_cast<int>(r.number_of_active_points);
}

int orc_build_epipolar_segment(int width, int height, const double intrinsics[4], const double T_target_reference[7],
                               const double observed[2], double idepth_min, double idepth_max, int cap, double *projections,
                               double *idepths) {
  const PinholeModel model{static_cast<double>(width), static_cast<double>(height), intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  const SE3 t = SE3::fromParams(T_target_reference);
  const EpipolarLineBuilder builder(model, t);
  const EpipolarLine line = builder.buildSegment(observed, idepth_min, idepth_max);
  const int n = static_cast<int>(line.points.size());
  for (int i = 0; i < std::min(n, cap); ++i) {
    projections[2 * i] = line.points[static_cast<size_t>(i)].projection[0];
    projections[2 * i + 1] = line.points[static_cast<size_t>(i)].projection[1];
    idepths[i] = line.points[static_cast<size_t>(i)].reference_idepth;
  }
  return n;
}

int orc_create_reference_depth_maps(int n_sources, const double *T_w_sources, const int32_t *counts, const double *uv,
                                    const double *idepth, const double *variance, const uint8_t *skip, const uint8_t *status,
                                    const double T_w_newest[7], const double intrinsics[4], int width, int height, int levels,
                                    double *idepth_sum_out, double *weight_out) {
  std::vector<DepthMapSource> sources(static_cast<size_t>(n_sources));
  size_t off = 0;
  for (int s = 0; s < n_sources; ++s) {
    DepthMapSource &src = sources[static_cast<size_t>(s)];
    src.t_world_agent = SE3::fromParams(T_w_sources + 7 * s);
    src.n = counts[s];
    src.uv = uv + 2 * off;
    src.idepth = idepth + off;
    src.variance = variance + off;
    src.skip = skip + off;
    src.status = status + off;
    off += static_cast<size_t>(counts[s]);
  }
  const PinholeModel model{static_cast<double>(width), static_cast<double>(height), intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  const auto maps = createReferenceDepthMaps(sources, SE3::fromParams(T_w_newest), model, levels);
  size_t o = 0;
  for (const DepthMapLevel &m : maps) {
    std::copy(m.idepth.begin(), m.idepth.end(), idepth_sum_out + o);
    std::copy(m.weight.begin(), m.weight.end(), weight_out + o);
    o += m.idepth.size();
  }
  return static_cast<int>(maps.size());
}

int orc_initialization_poses(int have_two_frames, const double T_w_previous[7], const double T_w_last[7], const double T_w_keyframe[7],
                             int cap, double *poses) {
  const auto v = initializationPoses(have_two_frames != 0, have_two_frames ? SE3::fromParams(T_w_previous) : SE3(),
                                     have_two_frames ? SE3::fromParams(T_w_last) : SE3(), have_two_frames ? SE3::fromParams(T_w_keyframe) : SE3());
  for (size_t i = 0; i < v.size() && static_cast<int>(i) < cap; ++i) v[i].toParams(poses + 7 * i);
  return static_cast<int>(v.size());
}
void orc_se3_log(const double T[7], double xi[6]) { SE3::fromParams(T).log(xi); }

double orc_mean_square_optical_flow(int width, int height, const double *idepth_sum, const double *weight, const double intrinsics[4],
                                    const double T_target_reference[7]) {
  DepthMapLevel m(width, height);
  std::copy(idepth_sum, idepth_sum + static_cast<size_t>(width) * height, m.idepth.begin());
  std::copy(weight, weight + static_cast<size_t>(width) * height, m.weight.begin());
  const PinholeModel model{static_cast<double>(width), static_cast<double>(height), intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]};
  return calculateMeanSquareOpticalFlow(m, SE3::fromParams(T_target_reference), model);
}

int orc_align_solve(const orc_options *o, int n, const double *u, const double *v, const double *idepth,
                    const double *intensity, const double ref_intrinsics[4], int ref_width, int ref_height,
                    const double T_w_ref[7], double ref_exposure, const double ref_ab[2], const double tgt_intrinsics[4],
                    int tgt_width, int tgt_height, const double *tgt_pixelinfo, const uint8_t *tgt_mask,
                    const double T_w_tgt_init[7], double tgt_exposure, const double tgt_ab[2], orc_align_result *out) {
  return orc_align_solve_with_prior(o, n, u, v, idepth, intensity, ref_intrinsics, ref_width, ref_height, T_w_ref, ref_exposure, ref_ab,
                                    tgt_intrinsics, tgt_width, tgt_height, tgt_pixelinfo, tgt_mask, T_w_tgt_init, tgt_exposure, tgt_ab,
                                    nullptr, out);
}

int orc_align_solve_with_prior(const orc_options *o, int n, const double *u, const double *v, const double *idepth,
                               const double *intensity, const double ref_intrinsics[4], int ref_width, int ref_height,
                               const double T_w_ref[7], double ref_exposure, const double ref_ab[2], const double tgt_intrinsics[4],
                               int tgt_width, int tgt_height, const double *tgt_pixelinfo, const uint8_t *tgt_mask,
                               const double T_w_tgt_init[7], double tgt_exposure, const double tgt_ab[2],
                               const double *prior_rotation_t_r, orc_align_result *out) {
  std::vector<AlignPoint> pts(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) pts[static_cast<size_t>(i)] = {u[i], v[i], idepth[i], intensity[i]};
  AlignFrame rf, tf;
  rf.T_w_agent = SE3::fromParams(T_w_ref);
  rf.exposure_time = ref_exposure;
  rf.affine_brightness0[0] = ref_ab[0];
  rf.affine_brightness0[1] = ref_ab[1];
  rf.model = PinholeModel{static_cast<double>(ref_width), static_cast<double>(ref_height), ref_intrinsics[0],
                          ref_intrinsics[1], ref_intrinsics[2], ref_intrinsics[3]};
  tf.T_w_agent = SE3::fromParams(T_w_tgt_init);
  tf.exposure_time = tgt_exposure;
  tf.affine_brightness0[0] = tgt_ab[0];
  tf.affine_brightness0[1] = tgt_ab[1];
  tf.model = PinholeModel{static_cast<double>(tgt_width), static_cast<double>(tgt_height), tgt_intrinsics[0],
                          tgt_intrinsics[1], tgt_intrinsics[2], tgt_intrinsics[3]};
  tf.grid = PixelMapView{tgt_pixelinfo, tgt_width, tgt_height};
  tf.mask = MaskView{tgt_mask, tgt_width, tgt_height};
  AlignResult r = alignSolve(rf, tf, pts, toOptions(o), prior_rotation_t_r);
  out->rmse = r.rmse;
  out->energy = r.lm.energy;
  out->n_valid = r.lm.number_of_valid_residuals;
  out->iterations = r.lm.iterations;
  r.T_w_target.toParams(out->T_w_target);
  out->affine_brightness[0] = r.affine_brightness[0];
  out->affine_brightness[1] = r.affine_brightness[1];
  std::memcpy(out->covariance, r.covariance, sizeof(r.covariance));
  std::memcpy(out->H, r.H, sizeof(r.H));
  return 0;
}

int orc_build_pyramid(const uint8_t *image, int width, int height, const double *lut256, const uint8_t *vignetting,
                      int levels, double **pixelinfo_out, double **plane_out) {
  Pyramid p = buildPyramid(image, width, height, lut256, vignetting, levels);
  for (size_t l = 0; l < p.planes.size(); ++l) {
    if (pixelinfo_out && pixelinfo_out[l]) std::memcpy(pixelinfo_out[l], p.pixelinfo[l].data(), p.pixelinfo[l].size() * sizeof(double));
    if (plane_out && plane_out[l]) std::memcpy(plane_out[l], p.planes[l].data(), p.planes[l].size() * sizeof(double));
  }
  return static_cast<int>(p.planes.size());
}

void orc_se3_exp(const double xi[6], double T[7]) { SE3::exp(xi).toParams(T); }
void orc_se3_mul(const double A[7], const double B[7], double C[7]) { (SE3::fromParams(A) * SE3::fromParams(B)).toParams(C); }
void orc_se3_inverse(const double A[7], double B[7]) { SE3::fromParams(A).inverse().toParams(B); }
void orc_se3_adj(const double A[7], double adj36[36]) { SE3::fromParams(A).Adj(adj36); }

int orc_reproject_pattern(const double ref_intr[4], int ref_w, int ref_h, const double tgt_intr[4], int tgt_w, int tgt_h,
                          const double T_t_r[7], int n, const double *u, const double *v, double idepth, int with_jacobians,
                          double *tu, double *tv, double *d_u_idepth, double *d_v_idepth, double *d_u_T, double *d_v_T) {
  if (n != 8 && n != 1) return -1;
  PinholeModel rm{static_cast<double>(ref_w), static_cast<double>(ref_h), ref_intr[0], ref_intr[1], ref_intr[2], ref_intr[3]};
  PinholeModel tm{static_cast<double>(tgt_w), static_cast<double>(tgt_h), tgt_intr[0], tgt_intr[1], tgt_intr[2], tgt_intr[3]};
  ArrayReprojector<true> rp(rm, tm, SE3::fromParams(T_t_r));
  bool ok;
  if (with_jacobians) {
    ok = (n == 8) ? rp.reprojectPattern<8>(u, v, idepth, tu, tv, d_u_idepth, d_v_idepth, d_u_T, d_v_T)
                  : rp.reprojectPattern<1>(u, v, idepth, tu, tv, d_u_idepth, d_v_idepth, d_u_T, d_v_T);
  } else {
    ok = (n == 8) ? rp.reprojectPattern<8>(u, v, idepth, tu, tv) : rp.reprojectPattern<1>(u, v, idepth, tu, tv);
  }
  return ok ? 1 : 0;
}

int orc_solve_system(int n, const double *H, const double *b, double *x) {
  NormalLinearSystem s(n);
  std::memcpy(s.H.a.data(), H, sizeof(double) * static_cast<size_t>(n) * n);
  std::memcpy(s.b.data(), b, sizeof(double) * static_cast<size_t>(n));
  Vec r = s.solve();
  std::memcpy(x, r.data(), sizeof(double) * static_cast<size_t>(n));
  return 0;
}
int orc_reduce_system(int n, double *H, double *b, int n_elim, const int32_t *elim, double *H_out, double *b_out) {
  NormalLinearSystem s(n);
  std::memcpy(s.H.a.data(), H, sizeof(double) * static_cast<size_t>(n) * n);
  std::memcpy(s.b.data(), b, sizeof(double) * static_cast<size_t>(n));
  std::vector<int> e(elim, elim + n_elim);
  s.reduceSystem(e);
  std::memcpy(H_out, s.H.a.data(), sizeof(double) * s.H.a.size());
  std::memcpy(b_out, s.b.data(), sizeof(double) * s.b.size());
  return s.size();
}
int orc_pinv_drop(int n, const double *H, int nullspaces, double *out) {
  Mat m(n, n);
  std::memcpy(m.a.data(), H, sizeof(double) * static_cast<size_t>(n) * n);
  Mat r = pseudoInverseDropSmallest(m, nullspaces);
  std::memcpy(out, r.a.data(), sizeof(double) * r.a.size());
  return 0;
}
/* relativeTransformationUncertainty (se3_motion.hpp:151-158) on its own, so that tests/test_oracle_uncertainty.py can hold it against
 * a finite-difference statement of the same propagation */
int orc_relative_transformation_uncertainty(const double T_w_1[7], const double T_w_2[7], const double *sigma_11, const double *sigma_22,
                                            const double *sigma_12, double *out36) {
  relativeTransformationUncertainty(SE3::fromParams(T_w_1), SE3::fromParams(T_w_2), sigma_11, sigma_22, sigma_12, out36);
  return 0;
}

/* The restated third-party arithmetic on its own (Eigen LDLT / completeOrthogonalDecomposition().pseudoInverse(), the bilinear
 * sampler of pixel_map.hpp:20-40), so that tests/test_oracle_thirdparty.py can hold each against a third-party implementation
 * (scipy / numpy) instead of against a second statement by the same author */
int orc_ldlt_solve(int n, const double *A, const double *b, double *x) {
  Mat m(n, n);
  std::memcpy(m.a.data(), A, sizeof(double) * static_cast<size_t>(n) * n);
  Vec r = ldltSolve(m, Vec(b, b + n));
  std::memcpy(x, r.data(), sizeof(double) * static_cast<size_t>(n));
  return 0;
}
int orc_pinv_cod(int n, const double *H, double *out) {
  Mat m(n, n);
  std::memcpy(m.a.data(), H, sizeof(double) * static_cast<size_t>(n) * n);
  Mat r = pseudoInverseCOD(m);
  std::memcpy(out, r.a.data(), sizeof(double) * r.a.size());
  return 0;
}
int orc_interpolate_linear(int width, int height, const double *pixelinfo, int n, const double *x, const double *y, double *out3) {
  PixelMapView map;
  map.data = pixelinfo;
  map.width = width;
  map.height = height;
  for (int i = 0; i < n; ++i) interpolateLinear3(map, x[i], y[i], out3 + 3 * i);
  return 0;
}
int orc_mask_valid(int width, int height, const uint8_t *mask, int n, const double *x, const double *y, uint8_t *out) {
  MaskView m;
  m.data = mask;
  m.width = width;
  m.height = height;
  for (int i = 0; i < n; ++i) out[i] = m.valid(x[i], y[i]) ? 1 : 0;
  return 0;
}

}  // extern "C"
