// Self-contained C++ driver of the host mirror classes (dsopp_hip_solvers.hpp): a 3-keyframe window over an analytic
// scene (fronto-parallel textured plane at z = 4, camera translating along x), solved with
// HipPhotometricBundleAdjustment, then one HipPoseAlignment of the third frame against the first, the reference depth
// maps, and the depth estimation + activation of immature landmarks when a fourth keyframe arrives.
// Exit code 0 = energies decreased and the perturbed poses moved toward the ground truth.
//   g++ -std=c++17 example_solvers.cpp -L../lib -ldsopp_hip -Wl,-rpath,$PWD/../lib -o example_solvers
#include <cmath>
#include <cstdio>
#include <random>

#include "dsopp_hip_solvers.hpp"

using namespace dsopp_hip_host;

namespace {
constexpr int W = 320, H = 240;
constexpr double fx = 224, fy = 224, cx = 160, cy = 120, Z = 4.0;

double texture(double X, double Y) {  // radiance of the plane point (X, Y, Z)
  return 127.5 + 55 * std::sin(9.0 * X) * std::cos(7.0 * Y) + 35 * std::sin(23.0 * X + 1.0) * std::sin(19.0 * Y) + 20 * std::cos(41.0 * (X + Y));
}

std::vector<uint8_t> render(double tx) {  // camera at (tx, 0, 0), identity rotation
  std::vector<uint8_t> img(static_cast<size_t>(W) * H);
  for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) {
      const double X = (u - cx) / fx * Z + tx, Y = (v - cy) / fy * Z;
      const double val = std::min(255.0, std::max(0.0, texture(X, Y)));
      img[static_cast<size_t>(v) * W + u] = static_cast<uint8_t>(std::lround(val));
    }
  return img;
}

/** The back end exactly as MonocularTracker::tick drives it for every new keyframe (monocular_tracker.cpp:497-507):
 *    pushFrame(new keyframe) -> updateSolver [updateLocalFrame of every active frame] -> refinePoses [solve, then updateFrame of
 *    every active frame] -> marginalize [the strategy flags a keyframe and landmarks] -> updateSolver again,
 *  over six keyframes through a four-frame window: the same call order the reference-side adapter
 *  (dsopp_amd/host/reference_adapter/hip_photometric_bundle_adjustment.cpp) forwards, including the fold-in of the
 *  marginalised frame by the next pushFrame. */
bool trackerCallOrder(const std::vector<int> &devices, std::vector<double> *final_tx) {
  const int kFrames = 6, kWindow = 4, kLandmarks = 80;
  TrustRegionOptions opt{7, 1e5, 1e-8, 1e-8, {1e12, 1e8}, 1e16, 20};
  // one device: a plain window.  Several entries: the single-process multi-device form (the YAML `devices` list of fabric_hip.patch);
  // entries naming the same device share it through the in-process reducer (how a one-GPU box exercises the sharded path)
  const bool sharded = devices.size() > 1;
  HipPhotometricBundleAdjustment pba(opt, true, true, devices, sharded ? DSOPP_HIP_TRANSPORT_LOCAL : DSOPP_HIP_TRANSPORT_AUTO);
  std::vector<std::unique_ptr<DevicePyramidGroup>> pyramid_groups;
  const PinholeModel model{fx, fy, cx, cy};
  std::mt19937 rng(7);
  std::uniform_int_distribution<int> ux(20, W - 21), uy(20, H - 21);
  std::vector<std::unique_ptr<DevicePyramid>> pyramids;
  std::vector<std::vector<uint8_t>> images;
  std::vector<KeyframeView> track(kFrames);
  std::vector<int> active;  // indices into `track`, oldest first
  bool ok = true;
  for (int i = 0; i < kFrames; ++i) {
    const double tx_gt = 0.08 * i, tx_init = tx_gt + (i ? 0.012 * ((i % 2) ? 1 : -1) : 0.0);
    images.push_back(render(tx_gt));
    pyramids.push_back(std::make_unique<DevicePyramid>(W, H, 1));
    pyramids.back()->build(images.back().data());
    if (sharded) {
      pyramid_groups.push_back(std::make_unique<DevicePyramidGroup>(pba.group(), W, H, 1));
      pyramid_groups.back()->build(images.back().data());
    }
    KeyframeView &f = track[static_cast<size_t>(i)];
    f.keyframe_id = i;
    f.timestamp = 1000 * (i + 1);
    f.t_world_agent = {0, 0, 0, 1, tx_init, 0, 0};
    f.exposure_time = 1;
    f.affine_brightness = {0, 0};
    f.is_marginalized = false;
    f.pyramids = pyramids.back().get();
    f.pyramid_group = sharded ? pyramid_groups.back().get() : nullptr;
    for (int k = 0; k < kLandmarks; ++k) {
      LandmarkView lm;
      const int u = ux(rng), v = uy(rng);
      lm.projection = {static_cast<double>(u), static_cast<double>(v)};
      lm.idepth = 1.0 / Z * (1 + 0.002 * ((k % 5) - 2));
      static const int px[8] = {0, -1, 1, -2, 0, 2, -1, 0}, py[8] = {2, 1, 1, 0, 0, 0, -1, -2};
      for (int p = 0; p < 8; ++p) lm.patch[static_cast<size_t>(p)] = images.back()[static_cast<size_t>(v + py[p]) * W + u + px[p]];
      lm.is_marginalized = lm.is_outlier = false;
      f.active_landmarks.push_back(lm);
    }
    // pushNewKeyframe: connections between the new keyframe and every active one (both directions)
    for (int j : active) {
      f.reprojection_statuses[j] = std::vector<uint8_t>(f.active_landmarks.size(), DSOPP_HIP_STATUS_OK);
      track[static_cast<size_t>(j)].reprojection_statuses[i] = std::vector<uint8_t>(track[static_cast<size_t>(j)].active_landmarks.size(), DSOPP_HIP_STATUS_OK);
    }
    active.push_back(i);
    pba.pushFrame(f, 0, model, i == 0 ? FrameParameterization::kFixed : FrameParameterization::kFree);  // :497
    for (int j : active) pba.updateLocalFrame(track[static_cast<size_t>(j)]);                           // updateSolver, :499
    if (active.size() < 2) continue;
    const double energy = pba.solve(1);                                                                  // refinePoses, :501
    for (int j : active) pba.updateFrame(track[static_cast<size_t>(j)]);
    ok = ok && std::isfinite(energy);
    const double err = std::abs(f.t_world_agent[4] - tx_gt);
    std::printf("keyframe %d: window of %zu, energy %.2f, |tx error| of the new keyframe %.5f\n", i, active.size(), energy, err);
    ok = ok && err < 0.012 * 0.6 + 2e-3;
    // marginalize (:503): once the window is full the strategy retires the oldest free keyframe and a few landmarks elsewhere
    if (static_cast<int>(active.size()) == kWindow) {
      const int victim = active[1];
      KeyframeView &v = track[static_cast<size_t>(victim)];
      v.is_marginalized = true;
      for (auto &lm : v.active_landmarks) lm.is_marginalized = true;
      for (int j : active)
        if (j != victim)
          for (size_t k = 0; k < track[static_cast<size_t>(j)].active_landmarks.size(); k += 9) track[static_cast<size_t>(j)].active_landmarks[k].is_marginalized = true;
      for (int j : active) pba.updateLocalFrame(track[static_cast<size_t>(j)]);                         // updateSolver, :505
      active.erase(active.begin() + 1);  // the next pushFrame folds it into the prior and drops it from the window
    }
  }
  int32_t in_window = 0;
  ok = ok && dsopp_hip_window_group_num_frames(pba.group(), &in_window) == DSOPP_HIP_OK && in_window == static_cast<int32_t>(active.size()) + 1;
  ok = ok && (pba.handle() != nullptr) == !sharded && pba.numDevices() == static_cast<int>(devices.size());
  std::printf("tracker call order on %zu device shard(s): %d frames in the solver's window (one of them awaiting its fold-in)\n", devices.size(), in_window);
  if (final_tx)
    for (const KeyframeView &f : track) final_tx->push_back(f.t_world_agent[4]);
  return ok;
}
}  // namespace

int main() {
  int n_dev = 0;
  dsopp_hip_device_count(&n_dev);
  if (n_dev < 1) {
    std::printf("no GPU: the HIP backend has no CPU fallback\n");
    return 2;
  }
  const double tx_gt[3] = {0.0, 0.10, 0.20};
  const double tx_init[3] = {0.0, 0.115, 0.18};
  std::vector<std::unique_ptr<DevicePyramid>> pyramids;
  std::vector<std::vector<uint8_t>> images;
  for (int i = 0; i < 3; ++i) {
    images.push_back(render(tx_gt[i]));
    pyramids.push_back(std::make_unique<DevicePyramid>(W, H, 3));
    pyramids.back()->build(images.back().data());
  }
  // production options of createPhotometricBundleAdjustment (src/tracker/tracker/src/fabric.cpp:63-79)
  TrustRegionOptions opt{7, 1e5, 1e-8, 1e-8, {1e12, 1e8}, 1e16, 20};
  HipPhotometricBundleAdjustment pba(opt, true, true);
  std::mt19937 rng(1);
  std::uniform_int_distribution<int> ux(20, W - 21), uy(20, H - 21);
  std::vector<KeyframeView> frames(3);
  const PinholeModel model{fx, fy, cx, cy};
  for (int i = 0; i < 3; ++i) {
    KeyframeView &f = frames[static_cast<size_t>(i)];
    f.keyframe_id = i;
    f.timestamp = 1000 * (i + 1);
    f.t_world_agent = {0, 0, 0, 1, tx_init[i], 0, 0};
    f.exposure_time = 1;
    f.affine_brightness = {0, 0};
    f.is_marginalized = false;
    f.pyramids = pyramids[static_cast<size_t>(i)].get();
    for (int k = 0; k < 120; ++k) {
      LandmarkView lm;
      const int u = ux(rng), v = uy(rng);
      lm.projection = {static_cast<double>(u), static_cast<double>(v)};
      lm.idepth = 1.0 / Z * (1 + 0.002 * ((k % 7) - 3));
      static const int px[8] = {0, -1, 1, -2, 0, 2, -1, 0}, py[8] = {2, 1, 1, 0, 0, 0, -1, -2};
      for (int p = 0; p < 8; ++p) lm.patch[static_cast<size_t>(p)] = images[static_cast<size_t>(i)][static_cast<size_t>(v + py[p]) * W + u + px[p]];
      lm.is_marginalized = lm.is_outlier = false;
      f.active_landmarks.push_back(lm);
    }
    for (int j = 0; j < i; ++j) {
      f.reprojection_statuses[j] = std::vector<uint8_t>(f.active_landmarks.size(), DSOPP_HIP_STATUS_OK);
      frames[static_cast<size_t>(j)].reprojection_statuses[i] = std::vector<uint8_t>(frames[static_cast<size_t>(j)].active_landmarks.size(), DSOPP_HIP_STATUS_OK);
    }
    pba.pushFrame(f, 0, model, i == 0 ? FrameParameterization::kFixed : FrameParameterization::kFree);
    for (int j = 0; j < i; ++j) pba.updateLocalFrame(frames[static_cast<size_t>(j)]);  // new connections of the older frames
  }
  const double energy = pba.solve(1);
  bool ok = std::isfinite(energy);
  for (int i = 1; i < 3; ++i) {
    pba.updateFrame(frames[static_cast<size_t>(i)]);
    const double err0 = std::abs(tx_init[i] - tx_gt[i]), err1 = std::abs(frames[static_cast<size_t>(i)].t_world_agent[4] - tx_gt[i]);
    std::printf("frame %d: |tx error| %.5f -> %.5f\n", i, err0, err1);
    ok = ok && err1 < 0.6 * err0 + 2e-3;
  }
  std::printf("final PBA energy %.3f\n", energy);

  // coarse alignment of frame 2 against frame 0 at level 1 with a sparse reference depth map
  TrustRegionOptions aopt{50, 1e2, 1e-5, 1e-5, {1e12, 1e8}, 1e16, 20};
  HipPoseAlignment align(aopt);
  const int w1 = W / 2, h1 = H / 2;
  std::vector<double> idsum(static_cast<size_t>(w1) * h1, 0.0), weight(static_cast<size_t>(w1) * h1, 0.0);
  for (int k = 0; k < 900; ++k) {
    const int u = ux(rng) / 2, v = uy(rng) / 2;
    idsum[static_cast<size_t>(v) * w1 + u] = 1.0 / Z;
    weight[static_cast<size_t>(v) * w1 + u] = 1.0;
  }
  const PinholeModel model1{fx / 2, fy / 2, cx / 2, cy / 2};
  align.reset();
  align.pushFrame(1000, Motion{0, 0, 0, 1, 0, 0, 0}, *pyramids[0], idsum.data(), weight.data(), 1.0, Vector2{0, 0}, 1, model1);
  align.pushFrame(3000, Motion{0, 0, 0, 1, 0.17, 0, 0}, *pyramids[2], 1.0, Vector2{0, 0}, 1, model1);
  const double rmse = align.solve(1);
  const Motion Ta = align.getPose(3000);
  std::printf("alignment: rmse %.3f, tx 0.17000 -> %.5f (ground truth 0.20000)\n", rmse, Ta[4]);

  // reference depth maps of the newest keyframe straight from the window (createReferenceDepthMaps), consumed on the device
  DeviceDepthMaps maps = pba.createReferenceDepthMaps(1);
  std::vector<double> dm_id, dm_w;
  int dm_width = 0, dm_height = 0;
  maps.level(0, dm_id, dm_w, dm_width, dm_height);
  int filled = 0;
  for (double wv : dm_w) filled += wv > 0 ? 1 : 0;
  std::printf("reference depth map %d x %d: %d cells carry depth\n", dm_width, dm_height, filled);
  ok = ok && filled > 0;
  // the keyframe strategy's parallax measure, straight from the device-resident map (with and without rotation, as the tracker asks)
  const std::vector<double> flow = maps.meanSquareOpticalFlow(0, {Motion{0, 0, 0, 1, -0.05, 0, 0}, Motion{0, 0, 0, 1, 0, 0, 0}}, model);
  std::printf("mean square optical flow for a 5 cm step: %.5f (analytic %.5f), for no motion: %.1e\n", flow[0], 0.05 / Z, flow[1]);
  ok = ok && std::abs(flow[0] - 0.05 / Z) < 2e-3 && flow[1] < 1e-12;
  const Motion T_prev{0, 0, 0, 1, 0.1, 0, 0}, T_last{0, 0, 0, 1, 0.2, 0, 0};
  const std::vector<Motion> hypotheses = initializationPoses(&T_prev, &T_last, &T_prev);
  std::printf("%zu pose hypotheses, constant-motion guess tx = %.3f\n", hypotheses.size(), hypotheses[0][4]);
  ok = ok && hypotheses.size() == 113 && std::abs(hypotheses[0][4] - 0.3) < 1e-12 && std::abs(hypotheses[2][4] - 0.25) < 1e-12;
  HipPoseAlignment align2(aopt);
  align2.reset();
  const KeyframeView &newest = frames[2];
  align2.pushFrame(newest.timestamp, newest.t_world_agent, *pyramids[2], maps, 1.0, Vector2{0, 0}, 0, model);
  align2.pushFrame(newest.timestamp + 500, newest.t_world_agent, *pyramids[2], 1.0, Vector2{0, 0}, 0, model);
  const double rmse2 = align2.solve(1);
  std::printf("alignment of the newest keyframe against its own depth map: rmse %.3f\n", rmse2);
  ok = ok && rmse2 >= 0 && rmse2 < 5;
  ok = ok && rmse > 0 && std::abs(Ta[4] - 0.2) < 0.015;

  // a fourth keyframe arrives: trace the immature landmarks of keyframe 0 in it (depth estimation on the resident set), then
  // let the activator pick and refine the ones that become active (LandmarksActivator::activate call order of the tracker)
  const double tx_new = 0.3;
  auto pyr_new = std::make_unique<DevicePyramid>(W, H, 2);
  const std::vector<uint8_t> image_new = render(tx_new);
  pyr_new->build(image_new.data());
  std::vector<ImmatureLandmarkView> immature;
  for (int k = 0; k < 200; ++k) {
    ImmatureLandmarkView lm;
    const int u = ux(rng), v = uy(rng);
    lm.projection = {static_cast<double>(u), static_cast<double>(v)};
    lm.direction = {(u - cx) / fx, (v - cy) / fy, 1.0};
    static const int px[8] = {0, -1, 1, -2, 0, 2, -1, 0}, py[8] = {2, 1, 1, 0, 0, 0, -1, -2};
    for (int p = 0; p < 8; ++p) lm.patch[static_cast<size_t>(p)] = images[0][static_cast<size_t>(v + py[p]) * W + u + px[p]];
    const auto I0 = [&](int x, int y) { return static_cast<double>(images[0][static_cast<size_t>(y) * W + x]); };
    lm.gradient = {0.5 * (I0(u + 1, v) - I0(u - 1, v)), 0.5 * (I0(u, v + 1) - I0(u, v - 1))};
    lm.idepth_min = 0.8 / Z;  // as an earlier observation would have left it (the analytic texture is periodic: a search over
    lm.idepth_max = 1.25 / Z;  // the whole epipolar line would lock onto the wrong period)
    immature.push_back(lm);
  }
  DeviceImmatureSet dset(immature);
  pba.updateFrame(frames[0]);
  const Motion T0 = frames[0].t_world_agent;  // identity rotation in this example: T_new^-1 T_0 is a pure translation
  // (the tracker's estimateDepths loop over all keyframes of the window is one call: here the window contributes one set)
  estimateDepthsOfWindow(*pyr_new, {&dset}, {Motion{0, 0, 0, 1, T0[4] - tx_new, T0[5], T0[6]}}, {1.0}, {Vector2{0, 0}}, 1.0, Vector2{0, 0}, model, 20.0);
  pba.createReferenceDepthMaps(maps);  // refill of the map object the tracker keeps (no allocation)
  KeyframeView fourth{};
  fourth.keyframe_id = 3;
  fourth.timestamp = 4000;
  fourth.t_world_agent = {0, 0, 0, 1, tx_new, 0, 0};
  fourth.exposure_time = 1;
  fourth.affine_brightness = {0, 0};
  fourth.pyramids = pyr_new.get();
  HipLandmarksActivator<true> activator(20.0, 2000);
  HipLandmarksActivator<true>::Track track{&pba, {0, 1, 2}, {&dset, nullptr, nullptr}, &fourth};
  std::vector<std::vector<double>> new_idepths;
  const auto act = activator.activate(track, &new_idepths);
  int n_act = 0;
  double worst = 0;
  for (size_t i = 0; i < act[0].size(); ++i)
    if (act[0][i] == ImmatureLandmarkActivationStatus::kActivate) {
      ++n_act;
      worst = std::max(worst, std::abs(new_idepths[0][i] - 1.0 / Z) * Z);
    }
  std::printf("activation: %d of %zu immature landmarks activated (%d skipped, %d deleted), worst relative idepth error %.4f, distance %.3f\n", n_act,
              act[0].size(), activator.lastResult().n_skipped, activator.lastResult().n_deleted, worst, activator.minDistanceToNeighbor());
  ok = ok && n_act > 20 && worst < 0.05;
  std::vector<double> tx_single, tx_sharded;
  ok = trackerCallOrder({0}, &tx_single) && ok;
  // the same sequence with the window's landmarks sharded three ways behind the same solver object
  ok = trackerCallOrder({0, 0, 0}, &tx_sharded) && ok;
  double worst_shard = 0;
  for (size_t i = 0; i < tx_single.size() && i < tx_sharded.size(); ++i) worst_shard = std::max(worst_shard, std::abs(tx_single[i] - tx_sharded[i]));
  std::printf("sharded vs single-device solver over the sequence: max |tx difference| %.2e\n", worst_shard);
  ok = ok && tx_single.size() == tx_sharded.size() && worst_shard < 1e-7;
  return ok ? 0 : 1;
}
