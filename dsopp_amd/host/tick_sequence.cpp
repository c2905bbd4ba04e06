// Native driver of a tracked SEQUENCE over the host mirror (dsopp_hip_solvers.hpp): MonocularTracker::tick as the reference runs it
// (src/tracker/tracker/src/monocular_tracker.cpp:425-525), in C++, on the C-ABI — BASELINE.json's second metric (frame-tracking
// ms / frame) without a Python interpreter between the calls.
//
//   per frame   (tick, :425-470):  image -> pyramid -> initializationPoses (:136-176) -> estimatePose (:179-245) against the
//                                  device-resident reference depth maps -> calculateMeanSquareOpticalFlow with / without rotation
//                                  (:104-134) -> estimateDepths of every keyframe's immature landmarks (:74-102) -> keyframe decision
//                                  (mean_square_optical_flow_and_rmse_keyframe_strategy.cpp:14-48)
//   per keyframe (:471-525):       LandmarksActivator::activate -> applyImmatureLandmarkActivationStatuses (activated landmarks join
//                                  the window) -> pushFrame -> solve (refinePoses) -> updateFrame of every keyframe -> marginalisation
//                                  of the oldest free keyframe beyond `max_keyframes` -> createReferenceDepthMaps
//
// The sequence (8-bit images, candidate pixels per frame, the two bootstrap keyframes) is read from a file that
// scripts/tick_sequence.py --export writes: rendering, the feature extractor's choice of candidate pixels and the initializer are outside
// the hot path and outside the timed regions, exactly as in the Python driver, whose HIP run this program reproduces
// call for call (bench.py compares the poses of the two).
//
//   g++ -std=c++17 -O2 tick_sequence.cpp -L../lib -ldsopp_hip -Wl,-rpath,$PWD/../lib -o tick_sequence
//   ./tick_sequence sequence.bin [poses_out.txt]        -> one JSON line on stdout
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <numeric>

#include "dsopp_hip_solvers.hpp"

using namespace dsopp_hip_host;

namespace {

struct Sequence {
  int32_t width = 0, height = 0, levels = 0, n_frames = 0, n_boot = 0, n_immature = 0, desired_points = 0, max_keyframes = 0, first_kf_gap = 0;
  double kf_factor = 0;
  double intrinsics[4] = {0, 0, 0, 0};
  std::vector<std::vector<uint8_t>> images;       // n_frames x (H * W)
  std::vector<Motion> poses_gt;                   // n_frames
  std::vector<std::vector<double>> candidates;    // n_frames x (n_immature * 2): candidate pixels of the frame, should it become a keyframe
  std::vector<double> boot_uv[2], boot_idepth[2]; // the two bootstrap keyframes (frames 0 and first_kf_gap): active landmarks near the truth
  Motion boot_pose[2];
};

template <typename T>
void readInto(std::ifstream &in, T *dst, size_t count) {
  in.read(reinterpret_cast<char *>(dst), static_cast<std::streamsize>(sizeof(T) * count));
  if (!in) throw std::runtime_error("sequence file is truncated");
}

Sequence readSequence(const char *path) {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error(std::string("cannot open ") + path);
  char magic[8];
  readInto(in, magic, 8);
  if (std::memcmp(magic, "DSOPTICK", 8) != 0) throw std::runtime_error("not a tick-sequence file");
  Sequence s;
  int32_t head[9];
  readInto(in, head, 9);
  s.width = head[0], s.height = head[1], s.levels = head[2], s.n_frames = head[3], s.n_boot = head[4], s.n_immature = head[5];
  s.desired_points = head[6], s.max_keyframes = head[7], s.first_kf_gap = head[8];
  readInto(in, &s.kf_factor, 1);
  readInto(in, s.intrinsics, 4);
  const size_t px = static_cast<size_t>(s.width) * s.height;
  s.images.resize(static_cast<size_t>(s.n_frames));
  for (auto &im : s.images) {
    im.resize(px);
    readInto(in, im.data(), px);
  }
  s.poses_gt.resize(static_cast<size_t>(s.n_frames));
  for (auto &T : s.poses_gt) readInto(in, T.data(), 7);
  for (int b = 0; b < 2; ++b) {
    s.boot_uv[b].resize(2 * static_cast<size_t>(s.n_boot));
    s.boot_idepth[b].resize(static_cast<size_t>(s.n_boot));
    readInto(in, s.boot_uv[b].data(), s.boot_uv[b].size());
    readInto(in, s.boot_idepth[b].data(), s.boot_idepth[b].size());
    readInto(in, s.boot_pose[b].data(), 7);
  }
  s.candidates.resize(static_cast<size_t>(s.n_frames));
  for (auto &c : s.candidates) {
    c.resize(2 * static_cast<size_t>(s.n_immature));
    readInto(in, c.data(), c.size());
  }
  return s;
}

// ---- the little SE3 arithmetic the tracker itself does on the host between solver calls (Sophus storage: qx qy qz qw tx ty tz)
struct Mat34 {
  double R[9], t[3];
};
Mat34 toMat(const Motion &T) {
  const double x = T[0], y = T[1], z = T[2], w = T[3];
  Mat34 m;
  m.R[0] = 1 - 2 * (y * y + z * z), m.R[1] = 2 * (x * y - z * w), m.R[2] = 2 * (x * z + y * w);
  m.R[3] = 2 * (x * y + z * w), m.R[4] = 1 - 2 * (x * x + z * z), m.R[5] = 2 * (y * z - x * w);
  m.R[6] = 2 * (x * z - y * w), m.R[7] = 2 * (y * z + x * w), m.R[8] = 1 - 2 * (x * x + y * y);
  m.t[0] = T[4], m.t[1] = T[5], m.t[2] = T[6];
  return m;
}
Motion fromMat(const Mat34 &m) {
  // rotation matrix -> unit quaternion (Shepperd's method), w >= 0
  const double *R = m.R;
  double q[4];  // x y z w
  const double tr = R[0] + R[4] + R[8];
  if (tr > 0) {
    const double s = std::sqrt(tr + 1.0) * 2;
    q[3] = 0.25 * s, q[0] = (R[7] - R[5]) / s, q[1] = (R[2] - R[6]) / s, q[2] = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const double s = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2;
    q[3] = (R[7] - R[5]) / s, q[0] = 0.25 * s, q[1] = (R[1] + R[3]) / s, q[2] = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const double s = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2;
    q[3] = (R[2] - R[6]) / s, q[0] = (R[1] + R[3]) / s, q[1] = 0.25 * s, q[2] = (R[5] + R[7]) / s;
  } else {
    const double s = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2;
    q[3] = (R[3] - R[1]) / s, q[0] = (R[2] + R[6]) / s, q[1] = (R[5] + R[7]) / s, q[2] = 0.25 * s;
  }
  if (q[3] < 0)
    for (double &v : q) v = -v;
  return Motion{q[0], q[1], q[2], q[3], m.t[0], m.t[1], m.t[2]};
}
/** inv(A) * B */
Mat34 relative(const Mat34 &A, const Mat34 &B) {
  Mat34 out;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += A.R[3 * k + i] * B.R[3 * k + j];
      out.R[3 * i + j] = s;
    }
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int k = 0; k < 3; ++k) s += A.R[3 * k + i] * (B.t[k] - A.t[k]);
    out.t[i] = s;
  }
  return out;
}

constexpr int kPatX[8] = {0, -1, 1, -2, 0, 2, -1, 0}, kPatY[8] = {2, 1, 1, 0, 0, 0, -1, -2};  // common/pattern/pattern.hpp:21-32

struct Keyframe {
  KeyframeView view;
  std::unique_ptr<DevicePyramid> pyramid;
  std::vector<ImmatureLandmarkView> immature;       // host copy of what the candidate selection produced (projection, patch)
  std::unique_ptr<DeviceImmatureSet> immature_set;  // their estimator state lives on the device
  int frame_index = 0;
};

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

int main(int argc, char **argv) {
  // DSOPP_TICK_PHASE_LOG=<file>: the phase boundaries of every frame / keyframe as CLOCK_MONOTONIC seconds — what scripts/hip_api_breakdown.py
  // buckets the calls of a rocprofv3 --hip-trace run by
  std::FILE *phase_log = std::getenv("DSOPP_TICK_PHASE_LOG") ? std::fopen(std::getenv("DSOPP_TICK_PHASE_LOG"), "w") : nullptr;
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s sequence.bin [poses_out.txt]\n", argv[0]);
    return 64;
  }
  int32_t n_devices = 0;
  if (dsopp_hip_device_count(&n_devices) != DSOPP_HIP_OK || n_devices < 1) {
    std::printf("no HIP device: this library has no CPU fallback\n");
    return 2;
  }
  try {
    const Sequence seq = readSequence(argv[1]);
    const int W = seq.width, H = seq.height, levels = seq.levels;
    const PinholeModel model{seq.intrinsics[0], seq.intrinsics[1], seq.intrinsics[2], seq.intrinsics[3]};
    // production settings (fabric.cpp:63-79,127-131): 7 iterations, lambda0 = 1e-5, force_accept; alignment: 50 iterations, 1e-2
    const TrustRegionOptions pba_opt{7, 1e5, 1e-8, 1e-8, {1e12, 1e8}, 1e16, 20};
    const TrustRegionOptions align_opt{50, 1e2, 1e-5, 1e-5, {1e12, 1e8}, 1e16, 20};
    HipPhotometricBundleAdjustment pba(pba_opt, /*estimate_uncertainty=*/true, /*force_accept=*/true);
    HipPoseAlignment aligner(align_opt);
    HipLandmarksActivator<true> activator(20.0, static_cast<size_t>(seq.desired_points));
    std::vector<std::unique_ptr<Keyframe>> alive, retired, retired_before;
    std::vector<Motion> est(static_cast<size_t>(seq.n_frames));
    std::vector<char> have(static_cast<size_t>(seq.n_frames), 0);
    std::vector<double> t_frame, t_keyframe;
    double phase[5] = {0, 0, 0, 0, 0};  // per frame: pyramid object, pyramid build, estimatePose (+ hypotheses), optical flow, depth estimation
    double kphase[6] = {0, 0, 0, 0, 0, 0};  // per keyframe: activation + appends, pushFrame, solve, updateFrame x window, marginalisation, depth maps
    int activated = 0, marginalised = 0, solves = 0, tries_max = 0;
    long lm_iterations = 0;

    // Device pyramids are recycled: allocating the five levels of a frame afresh (and freeing them, a device-wide synchronisation)
    // measured 2.2 ms per frame in this process — more than everything the frame computes.
    // The pool is filled before the clock starts (a tracker owns its ring of frame buffers from the start): window + the keyframe waiting
    // for its fold-in + the frame being tracked + one spare; should a sequence need more, takePyramid allocates inside the timed region.
    std::vector<std::unique_ptr<DevicePyramid>> pyramid_pool;
    for (int i = 0; i < seq.max_keyframes + 4; ++i) pyramid_pool.push_back(std::make_unique<DevicePyramid>(W, H, levels));
    auto takePyramid = [&] {
      if (pyramid_pool.empty()) return std::make_unique<DevicePyramid>(W, H, levels);
      auto p = std::move(pyramid_pool.back());
      pyramid_pool.pop_back();
      return p;
    };
    auto image = [&](int k) -> const std::vector<uint8_t> & { return seq.images[static_cast<size_t>(k)]; };
    auto patchAt = [&](int k, double u, double v, std::array<double, 8> &patch) {
      const int ui = static_cast<int>(u), vi = static_cast<int>(v);
      for (int p = 0; p < 8; ++p) patch[static_cast<size_t>(p)] = image(k)[static_cast<size_t>(vi + kPatY[p]) * W + ui + kPatX[p]];
    };
    auto newKeyframe = [&](int k, std::unique_ptr<DevicePyramid> pyr) {
      auto kf = std::make_unique<Keyframe>();
      kf->frame_index = k;
      kf->pyramid = std::move(pyr);
      kf->view.keyframe_id = k;
      kf->view.timestamp = 1000 * (static_cast<int64_t>(k) + 1);
      kf->view.exposure_time = 1;
      kf->view.affine_brightness = {0, 0};
      kf->view.is_marginalized = false;
      kf->view.pyramids = kf->pyramid.get();
      // pushImmatureLandmarks: the feature extractor's candidates with direction, patch and image gradient (active_keyframe.cpp:96-110)
      const std::vector<double> &c = seq.candidates[static_cast<size_t>(k)];
      kf->immature.resize(static_cast<size_t>(seq.n_immature));
      for (int i = 0; i < seq.n_immature; ++i) {
        ImmatureLandmarkView &lm = kf->immature[static_cast<size_t>(i)];
        const double u = c[2 * static_cast<size_t>(i)], v = c[2 * static_cast<size_t>(i) + 1];
        lm.projection = {u, v};
        lm.direction = {(u - model.cx) / model.fx, (v - model.cy) / model.fy, 1.0};
        patchAt(k, u, v, lm.patch);
        const int ui = static_cast<int>(u), vi = static_cast<int>(v);
        const auto I = [&](int x, int y) { return static_cast<double>(image(k)[static_cast<size_t>(y) * W + x]); };
        lm.gradient = {0.5 * (I(ui + 1, vi) - I(ui - 1, vi)), 0.5 * (I(ui, vi + 1) - I(ui, vi - 1))};
      }
      kf->immature_set = std::make_unique<DeviceImmatureSet>(kf->immature);
      return kf;
    };
    auto pushKeyframe = [&](std::unique_ptr<Keyframe> kf, const Motion &T_w, const Vector2 &affine, bool fixed) {
      kf->view.t_world_agent = T_w;
      kf->view.affine_brightness = affine;
      for (auto &h : alive) {  // residual lists between the new frame and every frame of the window, both directions (:98-124)
        h->view.reprojection_statuses[kf->view.keyframe_id].assign(h->view.active_landmarks.size(), DSOPP_HIP_STATUS_OK);
        kf->view.reprojection_statuses[h->view.keyframe_id].assign(kf->view.active_landmarks.size(), DSOPP_HIP_STATUS_OK);
      }
      pba.pushFrame(kf->view, 0, model, fixed ? FrameParameterization::kFixed : FrameParameterization::kFree);
      for (auto &h : alive) pba.updateLocalFrame(h->view);
      alive.push_back(std::move(kf));
    };

    // ---- bootstrap (the reference's initializer is outside the hot path): two keyframes with active landmarks near the truth
    const int boot_frames[2] = {0, seq.first_kf_gap};
    for (int b = 0; b < 2; ++b) {
      const int k = boot_frames[b];
      auto pyr = std::make_unique<DevicePyramid>(W, H, levels);
      pyr->build(image(k).data());
      auto kf = newKeyframe(k, std::move(pyr));
      for (int i = 0; i < seq.n_boot; ++i) {
        LandmarkView lm;
        lm.projection = {seq.boot_uv[b][2 * static_cast<size_t>(i)], seq.boot_uv[b][2 * static_cast<size_t>(i) + 1]};
        lm.idepth = seq.boot_idepth[b][static_cast<size_t>(i)];
        patchAt(k, lm.projection[0], lm.projection[1], lm.patch);
        lm.is_marginalized = lm.is_outlier = false;
        kf->view.active_landmarks.push_back(lm);
      }
      pushKeyframe(std::move(kf), seq.boot_pose[b], Vector2{0, 0}, b == 0);
    }
    pba.solve(1);
    for (auto &kf : alive) pba.updateFrame(kf->view);
    DeviceDepthMaps maps = pba.createReferenceDepthMaps(levels);
    for (int k = 0; k <= seq.first_kf_gap; ++k) est[static_cast<size_t>(k)] = seq.poses_gt[static_cast<size_t>(k)], have[static_cast<size_t>(k)] = 1;
    est[static_cast<size_t>(seq.first_kf_gap)] = alive.back()->view.t_world_agent;
    std::vector<double> rmse_last(static_cast<size_t>(levels), 1e10);
    Vector2 affine_prev{0, 0};
    double rmse_ref = -1;

    for (int k = seq.first_kf_gap + 1; k < seq.n_frames; ++k) {
      const double t0 = now();
      auto pyr = takePyramid();
      const double t_a = now();
      pyr->build(image(k).data());
      const double t_b = now();
      Keyframe &ref = *alive.back();
      const Motion T_ref = ref.view.t_world_agent;
      const Vector2 ab_ref = ref.view.affine_brightness;
      const std::vector<Motion> hyp = initializationPoses(&est[static_cast<size_t>(k - 2)], &est[static_cast<size_t>(k - 1)], &T_ref);
      const PoseEstimate pe = aligner.estimatePose(ref.view.timestamp, T_ref, *ref.pyramid, maps, 1.0, ab_ref, 1000 * (static_cast<int64_t>(k) + 1), *pyr, 1.0,
                                                   model, hyp, affine_prev, rmse_last);
      const double t_c = now();
      if (!pe.success) throw std::runtime_error("frame " + std::to_string(k) + ": tracking lost");
      const Motion T_new = pe.t_world_target;
      est[static_cast<size_t>(k)] = T_new, have[static_cast<size_t>(k)] = 1;
      affine_prev = pe.affine_brightness;
      tries_max = std::max(tries_max, pe.tries);
      lm_iterations += pe.lm_iterations;
      const double rmse0 = rmse_last[0];
      // calculateMeanSquareOpticalFlow with and without rotation (:104-134)
      const Mat34 M_new = toMat(T_new);
      const Mat34 t_t_r = relative(M_new, toMat(T_ref));
      Mat34 t_nr = t_t_r;
      for (int i = 0; i < 9; ++i) t_nr.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
      const std::vector<double> flow = maps.meanSquareOpticalFlow(0, {fromMat(t_t_r), fromMat(t_nr)}, model);
      const double t_d = now();
      // estimateDepths: every keyframe's immature landmarks against the new frame, one launch (:74-102)
      {
        std::vector<DeviceImmatureSet *> sets;
        std::vector<Motion> rel;
        std::vector<double> exposures;
        std::vector<Vector2> affines;
        for (auto &kf : alive) {
          sets.push_back(kf->immature_set.get());
          rel.push_back(fromMat(relative(M_new, toMat(kf->view.t_world_agent))));
          exposures.push_back(1.0);
          affines.push_back(kf->view.affine_brightness);
        }
        estimateDepthsOfWindow(*pyr, sets, rel, exposures, affines, 1.0, affine_prev, model, 20.0);
        sets.front()->synchronize();
      }
      // keyframe decision: mean_square_optical_flow_and_rmse_keyframe_strategy.cpp:14-48 (exposures are 1 in this sequence)
      if (rmse_ref < 0) rmse_ref = rmse0;
      const bool need_kf = seq.kf_factor * (4.5 * flow[0] + 9.0 * flow[1] + 2.0 * std::abs(affine_prev[0] - ab_ref[0])) > 1.0 || rmse0 / rmse_ref > 4.0;
      const double t_e = now();
      t_frame.push_back(t_e - t0);
      phase[0] += t_a - t0, phase[1] += t_b - t_a, phase[2] += t_c - t_b, phase[3] += t_d - t_c, phase[4] += t_e - t_d;
      if (phase_log) std::fprintf(phase_log, "frame %d %.9f %.9f %.9f %.9f %.9f %.9f\n", k, t0, t_a, t_b, t_c, t_d, t_e);
      if (!need_kf) {
        pyramid_pool.push_back(std::move(pyr));  // the frame's images go back to the pool (a tracker keeps a ring of them)
        continue;
      }

      // ================= new keyframe (:471-525) =================
      rmse_ref = -1;
      const double t_before_candidates = now();
      auto fresh = newKeyframe(k, std::move(pyr));  // candidate pixels -> immature landmarks: the feature extractor's job, not counted
      const double t1 = now();
      const double t_candidates = t1 - t_before_candidates;
      (void)t_candidates;
      fresh->view.t_world_agent = T_new;
      fresh->view.affine_brightness = affine_prev;
      // LandmarksActivator::activate + applyImmatureLandmarkActivationStatuses (:491-497)
      HipLandmarksActivator<true>::Track track{&pba, {}, {}, &fresh->view};
      for (auto &kf : alive) {
        track.keyframe_ids.push_back(kf->view.keyframe_id);
        track.immature.push_back(kf->immature_set.get());
      }
      std::vector<std::vector<double>> new_idepths;
      const auto statuses = activator.activate(track, &new_idepths);
      for (size_t a = 0; a < alive.size(); ++a) {
        Keyframe &kf = *alive[a];
        bool any = false;
        for (size_t i = 0; i < statuses[a].size(); ++i) {
          if (statuses[a][i] != ImmatureLandmarkActivationStatus::kActivate) continue;
          LandmarkView lm;
          lm.projection = kf.immature[i].projection;
          lm.idepth = new_idepths[a][i];
          lm.patch = kf.immature[i].patch;
          lm.is_marginalized = lm.is_outlier = false;
          kf.view.active_landmarks.push_back(lm);
          any = true;
          ++activated;
        }
        if (!any) continue;
        // the new landmarks get a residual towards every other frame of the window (LocalFrame::update, :109-123): lists grow to the new size
        for (auto &kv : kf.view.reprojection_statuses) kv.second.resize(kf.view.active_landmarks.size(), DSOPP_HIP_STATUS_OK);
        pba.updateLocalFrame(kf.view);
      }
      const double t2 = now();
      pushKeyframe(std::move(fresh), T_new, affine_prev, false);
      // the keyframe marginalised at the previous keyframe is folded into the prior now: its connections leave the frames (until here the
      // landmarks activated above still got their residual towards it, as LocalFrame::update gives them one in every connection)
      for (auto &gone : retired_before)
        for (auto &kf : alive) kf->view.reprojection_statuses.erase(gone->view.keyframe_id);
      const double t3 = now();
      pba.solve(1);  // refinePoses
      ++solves;
      const double t4 = now();
      for (auto &kf : alive) pba.updateFrame(kf->view);  // poses, affine brightness, inverse depths, statuses, covariances back into the frames
      est[static_cast<size_t>(k)] = alive.back()->view.t_world_agent;
      const double t5 = now();
      if (static_cast<int>(alive.size()) > seq.max_keyframes) {
        // marginalisation strategy (outside the hot path): the oldest free keyframe leaves; its landmarks are flagged, every frame is updated
        Keyframe &victim = *alive[1];
        for (LandmarkView &lm : victim.view.active_landmarks) lm.is_marginalized = true;
        victim.view.is_marginalized = true;
        for (auto &kf : alive) pba.updateLocalFrame(kf->view);
        // The solver keeps the marginalised frame — and BORROWS its image — until the next pushFrame folds it into the prior
        // (eigen_photometric_bundle_adjustment.cpp:119-141): its pyramid must outlive that call, as the keyframe's PixelMap does in the
        // reference (unloadMarginalizedResources runs behind the bundle adjustment, monocular_tracker.cpp:504)
        retired.push_back(std::move(alive[1]));
        alive.erase(alive.begin() + 1);
        ++marginalised;
      }
      const double t6 = now();
      pba.createReferenceDepthMaps(maps);  // refill of the tracker's map object
      std::fill(rmse_last.begin(), rmse_last.end(), 1e10);
      const double t7 = now();
      t_keyframe.push_back(t7 - t1);
      kphase[0] += t2 - t1, kphase[1] += t3 - t2, kphase[2] += t4 - t3, kphase[3] += t5 - t4, kphase[4] += t6 - t5, kphase[5] += t7 - t6;
      if (phase_log) std::fprintf(phase_log, "keyframe %d %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n", k, t1, t2, t3, t4, t5, t6, t7);
      // frames marginalised at the PREVIOUS keyframe were folded into the prior by this keyframe's pushFrame: their resources go now
      // (images back to the pool; unloadMarginalizedResources, monocular_tracker.cpp:504) — device memory management, not tracker time
      for (auto &old_kf : retired_before) pyramid_pool.push_back(std::move(old_kf->pyramid));
      retired_before.clear();
      retired_before.swap(retired);
    }

    if (phase_log) std::fclose(phase_log);
    // ---- report
    auto stat = [](std::vector<double> v, double q) {
      if (v.empty()) return 0.0;
      std::sort(v.begin(), v.end());
      const double pos = q * static_cast<double>(v.size() - 1);
      const size_t lo = static_cast<size_t>(pos);
      const size_t hi = std::min(lo + 1, v.size() - 1);
      return v[lo] + (pos - static_cast<double>(lo)) * (v[hi] - v[lo]);
    };
    const double sum_f = std::accumulate(t_frame.begin(), t_frame.end(), 0.0), sum_k = std::accumulate(t_keyframe.begin(), t_keyframe.end(), 0.0);
    double final_err = 0;
    {
      const Mat34 E = relative(toMat(seq.poses_gt.back()), toMat(est.back()));
      final_err = std::sqrt(E.t[0] * E.t[0] + E.t[1] * E.t[1] + E.t[2] * E.t[2]);
    }
    size_t window_landmarks = 0;
    for (auto &kf : alive) window_landmarks += kf->view.active_landmarks.size();
    if (argc > 2) {
      std::FILE *f = std::fopen(argv[2], "w");
      if (!f) throw std::runtime_error("cannot write the pose file");
      for (int k = 0; k < seq.n_frames; ++k)
        if (have[static_cast<size_t>(k)]) {
          const Motion &T = est[static_cast<size_t>(k)];
          std::fprintf(f, "%d %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", k, T[0], T[1], T[2], T[3], T[4], T[5], T[6]);
        }
      std::fclose(f);
    }
    std::printf("{\"driver\": \"dsopp_amd/host/tick_sequence.cpp (C++ over dsopp_hip_solvers.hpp)\", \"frames\": %zu, \"keyframes\": %zu, "
                "\"ms_per_frame_mean\": %.4f, \"ms_per_frame_median\": %.4f, \"ms_per_frame_p95\": %.4f, \"ms_per_keyframe_mean\": %.4f, "
                "\"ms_per_keyframe_p95\": %.4f, \"ms_per_frame_including_keyframe_work\": %.4f, \"lm_iterations_per_frame\": %.2f, "
                "\"hypotheses_tried_max\": %d, \"activated\": %d, \"marginalised\": %d, \"solves\": %d, \"window_landmarks_end\": %zu, "
                "\"translation_error_final\": %.6g, \"ms_per_frame_by_phase\": {\"pyramid_object\": %.4f, \"pyramid_build\": %.4f, \"estimate_pose\": %.4f, "
                "\"optical_flow\": %.4f, \"depth_estimation\": %.4f}, \"ms_per_keyframe_by_phase\": {\"activation_and_appends\": %.4f, \"push_frame\": %.4f, "
                "\"solve\": %.4f, \"update_frames\": %.4f, \"marginalisation\": %.4f, \"depth_maps\": %.4f}}\n",
                t_frame.size(), t_keyframe.size(), 1e3 * sum_f / std::max<size_t>(1, t_frame.size()), 1e3 * stat(t_frame, 0.5), 1e3 * stat(t_frame, 0.95),
                1e3 * sum_k / std::max<size_t>(1, t_keyframe.size()), 1e3 * stat(t_keyframe, 0.95), 1e3 * (sum_f + sum_k) / std::max<size_t>(1, t_frame.size()),
                static_cast<double>(lm_iterations) / std::max<size_t>(1, t_frame.size()), tries_max, activated, marginalised, solves, window_landmarks, final_err,
                1e3 * phase[0] / std::max<size_t>(1, t_frame.size()), 1e3 * phase[1] / std::max<size_t>(1, t_frame.size()),
                1e3 * phase[2] / std::max<size_t>(1, t_frame.size()), 1e3 * phase[3] / std::max<size_t>(1, t_frame.size()),
                1e3 * phase[4] / std::max<size_t>(1, t_frame.size()), 1e3 * kphase[0] / std::max<size_t>(1, t_keyframe.size()),
                1e3 * kphase[1] / std::max<size_t>(1, t_keyframe.size()), 1e3 * kphase[2] / std::max<size_t>(1, t_keyframe.size()),
                1e3 * kphase[3] / std::max<size_t>(1, t_keyframe.size()), 1e3 * kphase[4] / std::max<size_t>(1, t_keyframe.size()),
                1e3 * kphase[5] / std::max<size_t>(1, t_keyframe.size()));
    return 0;
  } catch (const std::exception &e) {
    std::fprintf(stderr, "tick_sequence: %s\n", e.what());
    return 1;
  }
}
