// dsopp_hip_window_*: host side of the sliding-window photometric bundle adjustment.
//
// Mirrors EigenPhotometricBundleAdjustment (PROB_SRC/eigen_photometric_bundle_adjustment.cpp:47-141) and its base class
// (PROB_SRC/photometric_bundle_adjustment.cpp:21-435): the window owns per-frame landmark / residual tables in HBM,
// launches the sweeps, and keeps the small dense pieces the reference also keeps in double on the host
// (marginal prior, covariance pseudo-inverse).  There is no CPU compute path: every stage is a HIP kernel.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <map>
#include <memory>
#include <mutex>

#include "host_linalg.hpp"
#include "pba_solve_kernels.hpp"
#include "pba_solve_combined.hpp"
#include "pba_schur_two_stage.hpp"
#include "depth_map_kernels.hpp"
#include "point_status_kernels.hpp"
#include "depth_maps.hpp"
#include "activation_kernels.hpp"
#include "immature_set.hpp"

namespace dsopp_hip {
namespace {

struct ResidualTable {
  DeviceBuffer<uint8_t> status, cand, fej_valid, snap_status;
  DeviceBuffer<double> energy;
  int n = 0, snap_n = 0;
};

struct HostFrame {
  int id = 0;
  int64_t timestamp = 0;
  const dsopp_hip_pyramid *pyramid = nullptr;
  unsigned pyramid_generation = 0;  // the pyramid's rewrite count when the frame's sweep descriptors (texels, intensity plane) were built
  int level = 0;
  double intr[4] = {0, 0, 0, 0};
  double exposure = 1;
  bool fixed = false, is_marginalized = false, to_marginalize = false;
  int n = 0;
  int cap = 0;
  // statuses (caller's order) of connections whose target is NOT in the window: kept on the host, as the reference keeps every residual list
  // of LocalFrame::update whether or not the solver holds the target (local_frame.hpp:507-519) — they become a device table when a frame
  // with that id is pushed (a tracker declares its connections towards a frame that was just folded into the prior once more, and towards
  // a new frame possibly before it is pushed: neither needs device memory)
  std::map<int, std::vector<uint8_t>> pending;
  std::vector<uint8_t> flags;  // host mirror of the landmark flags as last uploaded
  DeviceBuffer<double> uv, idepth, idepth_step, idepth_fej, patch, inv_hdd, b_d, relative_baseline, ublk;
  DeviceBuffer<int32_t> n_inliers;
  DeviceBuffer<uint8_t> dflags, snap_flags;
  DeviceBuffer<double> snap_idepth;
  int snap_n = 0;
  std::map<int, std::unique_ptr<ResidualTable>> residuals;  // by target frame id
  std::map<int, std::array<double, 36>> covariance;
  // Internal order of the landmarks.  The C-ABI speaks the CALLER's indices; on the device every appended batch of
  // landmarks is held sorted by 32 x 32-pixel tile of its projection (raster inside a tile): neighbouring items of a sweep workgroup then
  // sample neighbouring texels, i.e. share 64-byte granules — the sweep is bound by the rate of those requests.  Invariant that keeps the
  // residual tables' prefix meaning ("landmark i has a residual in target t iff i < n_res[t]"): for every entry e of batch_end the device
  // indices [0, e) hold exactly the caller indices [0, e).  A connection that ends INSIDE a batch splits it (splitBatchAt).
  std::vector<int> to_internal, to_caller;  // caller index <-> device index; empty = identity (order kept: DSOPP_HIP_LANDMARK_ORDER=caller)
  std::vector<int> batch_end;
  DeviceBuffer<int> d_to_internal;          // the caller -> device map on the device (export kernels un-permute there)
  bool permuted() const { return !to_internal.empty(); }
  int internalOf(int caller) const { return to_internal.empty() ? caller : to_internal[static_cast<size_t>(caller)]; }
};

}  // namespace
}  // namespace dsopp_hip

using namespace dsopp_hip;

struct dsopp_hip_window {
  StreamRef sr;
  dsopp_hip_options opt;
  std::vector<std::unique_ptr<HostFrame>> frames;
  // host mirror of the dynamic state
  WindowState hst;
  // marginal prior (system_marginalized_, energy_marginalized_ — PBA_INC/eigen_photometric_bundle_adjustment.hpp:68-70)
  std::vector<double> Hm, bm;
  double energy_marginalized = 0;
  int marg_size = 0;
  // device
  DeviceBuffer<FrameDev> d_frames;
  DeviceBuffer<WindowState> d_state, d_state_snap;
  DeviceBuffer<PairConst> d_pc;
  DeviceBuffer<SweepBlock> d_sweep_table;   // the sweeps' table: one entry (= one workgroup, one row of d_partials) per n_groups x 16 items
  DeviceBuffer<SweepBlock> d_fine_table;    // one entry per 16 items (first-estimate and point-status kernels); the same when n_groups == 1
  int n_fine_blocks = 0;
  const SweepBlock *fineTable() const { return d_fine_table.ptr ? d_fine_table.ptr : d_sweep_table.ptr; }
  DeviceBuffer<SchurBlock> d_schur_table;
  DeviceBuffer<int> d_pair_first, d_pair_count;
  // d_reduce = [Hpp K*K | bpp K | Hsc K*K | bsc K] (no priors): everything a multi-GPU run must sum across ranks, contiguous
  DeviceBuffer<double> d_partials, d_reduce, d_Hpp, d_bpp, d_Hm, d_Hm_packed, d_bm, d_step, d_scalars, d_gather;
  bool marg_nonzero = false;
  // two-stage (atomic-free, order-deterministic) build of the combined system: pba_schur_two_stage.hpp
  DeviceBuffer<double> d_schur_partials, d_pair_out;
  bool deterministic = false;  // dsopp_hip_window_set_deterministic: the two-stage build at every window size
  /** chunks of 64 landmarks above which the atomic-free two-stage build replaces the atomic accumulation (its three small launches
   *  more per iteration — scalar groups, ordered sum, separate back-substitution — are outweighed by the atomics' contention only
   *  from about 12 000 landmarks; the same bound serves landmark shards: shard of 6 250 landmarks / 12 frames 108 us two-stage) */
  int twoStageMinChunks() const {
    static const int override_chunks = std::getenv("DSOPP_HIP_TWO_STAGE_MIN_CHUNKS") ? std::atoi(std::getenv("DSOPP_HIP_TWO_STAGE_MIN_CHUNKS")) : 0;  // tuning aid
    if (override_chunks > 0) return override_chunks;
    return kTwoStageMinChunks;
  }
  bool twoStage() const { return deterministic || n_schur_blocks > twoStageMinChunks(); }
  /** chunks above which calculateIdepths runs as its own kernel in front of the sweep instead of inside it (the fused form re-reads a
   *  landmark's Schur row once per (landmark, target) item) */
  int backsubSplitMinChunks() const {
    static const int override_chunks = std::getenv("DSOPP_HIP_BACKSUB_SPLIT_MIN_CHUNKS") ? std::atoi(std::getenv("DSOPP_HIP_BACKSUB_SPLIT_MIN_CHUNKS")) : 0;  // tuning aid
    return override_chunks > 0 ? override_chunks : twoStageMinChunks();
  }
  unsigned *d_bs_flag = nullptr;  // ticket counter + hand-over buffers of the back-substitution inside the solve launch (pba_solve_combined.hpp)
  unsigned bs_seq = 0, bs_ticket_base = 0;
  int *h_bs_fault = nullptr;  // pinned: raised by a landmark workgroup of the solve launch whose bounded wait for the step ran out (checkSolveLaunchFault)
  long long *dbg_stamps = nullptr;
  long long *dbg_sweep = nullptr;
  bool dbg_sweep_lin = true;
  DeviceBuffer<LmControl> d_ctrl;
  const LmControl *fused_final_ctrl = nullptr;  // control block the enqueued fused loop ends in
  bool async_pending = false;                   // dsopp_hip_window_optimize_async enqueued, _wait not called yet
  bool begun_with_first_estimate = true;        // the solve in progress started with firstEstimate() (stageBegin) rather than fusedBegin
  int lm_mode = 0;  // 0: fused device loop (3 launches / iteration), 1: host-driven stages, 2: unfused device loop (5 launches)
  double *dHppRaw() const { return d_reduce.ptr; }
  double *dbppRaw() const { return d_reduce.ptr + static_cast<size_t>(K()) * K(); }
  double *dHsc() const { return dbppRaw() + K(); }
  double *dbsc() const { return dHsc() + static_cast<size_t>(K()) * K(); }
  void d_HscDownload(double *host, size_t n, size_t off, hipStream_t st) const {
    HIP_CHECK(hipMemcpyAsync(host, dHsc() + off, n * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  void d_bscDownload(double *host, size_t n, size_t off, hipStream_t st) const {
    HIP_CHECK(hipMemcpyAsync(host, dbsc() + off, n * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  size_t reduceCount() const { return 2 * (static_cast<size_t>(K()) * K() + K()); }
  /** fused LM loop: the combined block-packed system [blocks | rhs] at the head of d_reduce (pba_solve_kernels.hpp: ReduceSchurArgs::comb) */
  size_t combCount() const { return static_cast<size_t>(combBlockCount(F())) * 64 + static_cast<size_t>(K()); }
  // copies of the combined system the reduction launch spreads its atomics over (ReduceSchurArgs::comb_copies); > 1 only while the fused
  // loop of an unsharded window of up to 8 keyframes on the atomics path is being enqueued
  int comb_copies_active = 1;
  size_t combCopyFirst() const { return (combCount() + 4 * kScalarGroups + 8 + 1) & ~static_cast<size_t>(1); }
  size_t combCopyStride() const { return (combCount() + 1) & ~static_cast<size_t>(1); }
  int n_sweep_blocks = 0, n_schur_blocks = 0;
  bool topology_dirty = true;
  bool frames_dirty = false;          // only the frames' marginalisation flags changed since the tables were built: the frame table is patched
  std::vector<FrameDev> h_frames;     // the frame table as last uploaded (what such a patch starts from)
  bool state_dirty = true;   // host mirror newer than device
  bool host_stale = false;   // device state newer than the host mirror (after a device-driven solve): see downloadState
  LmControl *h_ctrl = nullptr;  // pinned read-back buffer of the solve result
  void *h_uncertainty = nullptr;          // pinned destination of estimateUncertainty's systems (+ frame states)
  size_t h_uncertainty_bytes = 0;
  hipEvent_t uncertainty_ready = nullptr; // recorded behind that transfer
  bool restore_in_begin = false;      // optimize_repeated: the restore to the snapshot rides in the next solve's opening kernel
  DeviceBuffer<LmControl> d_results;  // optimize_repeated: one result slot per solve of a batch ...
  LmControl *h_results = nullptr;     // ... fetched together into pinned memory
  LmControl *result_device = nullptr; // set while such a solve is enqueued: where its closing kernel leaves the control block
  DeviceBuffer<SelectState> d_select;   // radix-select state of updatePointStatuses
  DeviceBuffer<double> d_pair_dist;     // camera-centre distances of all frame pairs
  DeviceBuffer<double> d_export;        // packed per-frame read-back (get_frame_update): 4 n doubles, then (1 + targets) n bytes
  void *h_export = nullptr;             // its pinned host staging
  size_t h_export_bytes = 0;
  // Pinned bump allocator for the per-keyframe uploads (landmarks, connection statuses, flags): the caller's arrays are
  // copied here and leave with asynchronous transfers, so set_landmarks / set_connection need no synchronisation of their own.
  // When the ring wraps the stream is synchronised once (every transfer that read the old contents has finished then).
  struct StageRing {
    char *base = nullptr;
    size_t capacity = 0, offset = 0;
  } stage;
  // Appends of a keyframe step (dsopp_hip_window_set_landmarks / _set_connection: ~120 calls per keyframe from the tracker, each of which used
  // to cost one or more pinned-ring copies and a small kernel — 0.6 ms of host time per keyframe in the native driver, rocprofv3 --hip-trace,
  // profiles/r06/keyframe_hip_trace_breakdown.json) are QUEUED here and leave as ONE copy + ONE kernel in front of the next call that touches the
  // device (flushAppends): `blob` = the callers' data back to back, `ops` = what to do with it.
  struct AppendOp {
    int kind;                    // 0: copy; 1: merge landmark flags (+ clear the solver state of new landmarks); 2: new connection entries; 3: clear flag bits (mask in `a`); 4: sweep-table entries of a frame pair from its template
    int n;                       // elements (> 0)
    int a;                       // kind 0: bytes per element (8, 4, 1); kind 1: n_old; kind 2: keep (first new entry)
    int first_block;             // first workgroup of the launch that works on this operation
    unsigned long long src_off;  // byte offset of the operation's data in the blob
    void *dst;                   // kind 0: destination; kind 1: device flags; kind 2: statuses
    void *p[6];                  // kind 1: idepth_step, idepth_fej, inv_hdd, b_d, relative_baseline, n_inliers (null: no new landmarks); kind 2: cand, fej_valid, energy
  };
  std::vector<uint8_t> append_blob;
  std::vector<AppendOp> append_ops;
  DeviceBuffer<uint8_t> d_append;
  // updateFrame read-back prefetched by solve(): the tracker calls updateFrame for every keyframe right after the solve
  // (refinePoses), so solve() packs all frames behind its own final synchronisation and the getters become host copies.
  // Valid until the next call that can change the window (every such entry point clears it).
  struct ExportEntry {
    int frame_id, n;
    size_t word_offset;               // into h_update (doubles)
    std::vector<int> target_ids;      // status rows in this order
  };
  std::vector<ExportEntry> export_entries;
  DeviceBuffer<double> d_update;
  void *h_update = nullptr;
  size_t h_update_bytes = 0;
  bool export_valid = false;
  // device buffers of keyframes / connections that left the window, kept for the next keyframe: a new keyframe needs ~12
  // landmark arrays and 12 connection tables of 5 arrays each — about 70 hipMallocs (0.5 ms) when allocated afresh
  std::vector<std::unique_ptr<HostFrame>> frame_pool;
  std::vector<std::unique_ptr<ResidualTable>> table_pool;
  DeviceBuffer<double> dm_tmp;  // undilated reference depth maps, all levels (temporaries of createReferenceDepthMaps)
  std::vector<dsopp_hip_depth_maps *> live_maps;          // maps this window produced and that still borrow its stream
  struct ActivationScratch {            // work buffers of dsopp_hip_window_activate_landmarks
    DeviceBuffer<ActKeyframe> keyframes;
    DeviceBuffer<ActPair> pairs;
    DeviceBuffer<const void *> texels0;
    DeviceBuffer<double> px, py;
    DeviceBuffer<double> sx, sy;
    DeviceBuffer<int> state, cell_cursor, sid, accepted, nbr, nbr_count;
    // results and counters in one buffer: [inverse depths nI | distance | statuses nI bytes | counters 8 ints | cell_start]: what the host reads
    // back is one contiguous range (one copy), what a call clears (counters, cell_start) another (one fill)
    DeviceBuffer<double> pack;
  } act;
  bool marg_dirty = true;
  bool pair_valid = false;   // pair constants match the device state
  bool begun = false;
  bool linearized = false;
  bool reduced_by_collective = false;
  double last_lambda = 0;
  std::vector<double> last_step;
  float last_solve_ms = 0;
  bool solve_events_pending = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  dsopp_hip_allreduce_fn allreduce = nullptr;
  void *allreduce_user = nullptr;
  int rank = 0, world = 1;
  // per-kernel-class HIP-event timing (bench.py's live roofline measurement)
  bool profiling = false;
  struct TimedLaunch {
    int cls;
    hipEvent_t a, b;
  };
  std::vector<TimedLaunch> timed;
  std::vector<hipEvent_t> event_pool;
  double prof_ms[DSOPP_HIP_NUM_KERNEL_CLASSES] = {0};
  int64_t prof_count[DSOPP_HIP_NUM_KERNEL_CLASSES] = {0};
  // device-side snapshot of the mutable solver state (bench loops / tracker retries): see dsopp_hip_window_snapshot
  WindowState snap_state;
  bool snap_valid = false;
  int snap_F = 0;

  int F() const { return static_cast<int>(frames.size()); }
  int K() const { return kBlk * F(); }
  bool fej() const { return opt.first_estimate_jacobians != 0; }
  int slotOf(int id) const {
    for (size_t i = 0; i < frames.size(); ++i)
      if (frames[i]->id == id) return static_cast<int>(i);
    return -1;
  }
  HostFrame &frameById(int id) {
    const int s = slotOf(id);
    if (s < 0) fail(DSOPP_HIP_ERR_NOT_FOUND, "frame %d is not in the window", id);
    return *frames[static_cast<size_t>(s)];
  }
};

namespace dsopp_hip {
namespace {

using W = dsopp_hip_window;

}  // namespace
int nativeAllreduce(void *user, void *device_buffer, size_t count, void *stream);  // comm.hip
namespace {

hipEvent_t takeEvent(W &w) {
  if (!w.event_pool.empty()) {
    hipEvent_t e = w.event_pool.back();
    w.event_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  HIP_CHECK(hipEventCreate(&e));
  return e;
}

/** brackets the launches issued by `body` with HIP events on the window's stream when profiling is on */
template <typename Body>
void timedLaunch(W &w, int cls, Body &&body) {
  if (!w.profiling) {
    body();
    return;
  }
  hipEvent_t a = takeEvent(w), b = takeEvent(w);
  HIP_CHECK(hipEventRecord(a, w.sr.stream));
  body();
  HIP_CHECK(hipEventRecord(b, w.sr.stream));
  w.timed.push_back({cls, a, b});
}

void collectTimings(W &w) {
  if (w.timed.empty()) return;
  w.sr.sync();
  for (auto &t : w.timed) {
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, t.a, t.b));
    w.prof_ms[t.cls] += ms;
    w.prof_count[t.cls] += 1;
    w.event_pool.push_back(t.a);
    w.event_pool.push_back(t.b);
  }
  w.timed.clear();
}

/** does the device hold the landmarks in its own (spatial) order?  DSOPP_HIP_LANDMARK_ORDER=caller keeps the caller's (A/B aid) */
bool sortLandmarksInternally() {
  static const bool keep = std::getenv("DSOPP_HIP_LANDMARK_ORDER") != nullptr && std::string(std::getenv("DSOPP_HIP_LANDMARK_ORDER")) == "caller";
  return !keep;
}

/** rows of `row_bytes` bytes gathered from absolute row indices: dst[k] = src[rows[k]] (the rare re-ordering of a landmark batch) */
__global__ void gatherRowsKernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, const int *__restrict__ rows, int count, int row_bytes) {
  const long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (e >= static_cast<long long>(count) * row_bytes) return;
  const int k = static_cast<int>(e / row_bytes), b = static_cast<int>(e % row_bytes);
  dst[e] = src[static_cast<size_t>(rows[k]) * row_bytes + b];
}

/**
 * A residual list of frame `f` is about to end at caller index n, INSIDE a batch the device holds in its own order: the batch is split —
 * the landmarks with caller index < n move in front of the others (each part keeps its order) — so that device indices [0, n) again hold
 * exactly the caller indices [0, n), which is what `i < n_res[t]` means in every kernel.  Every per-landmark array of the frame and of its
 * residual tables is re-ordered on the device.  Rare: the tracker's lists always end where a batch ends (LocalFrame::update appends
 * landmarks and their residuals together, local_frame.hpp:484-521); ragged test windows come through here.
 */
void flushAppends(W &w, const char *why = "flush: in front of an entry point");

void splitBatchAt(W &w, HostFrame &f, int n) {
  if (!f.permuted() || n <= 0 || n >= f.n) return;
  int b0 = 0, b1 = f.n;
  for (int e : f.batch_end) {
    if (e == n) return;  // already a boundary
    if (e < n) b0 = std::max(b0, e);
    if (e > n) b1 = std::min(b1, e);
  }
  flushAppends(w, "flush: splitBatchAt");  // the rows about to be permuted may still be on their way (queued appends)
  const int count = b1 - b0;
  std::vector<int> new_to_old;  // absolute device rows
  new_to_old.reserve(static_cast<size_t>(count));
  for (int p = b0; p < b1; ++p)
    if (f.to_caller[static_cast<size_t>(p)] < n) new_to_old.push_back(p);
  for (int p = b0; p < b1; ++p)
    if (f.to_caller[static_cast<size_t>(p)] >= n) new_to_old.push_back(p);
  hipStream_t st = w.sr.stream;
  DeviceBuffer<int> d_rows;
  DeviceBuffer<uint8_t> tmp;
  d_rows.reserve(static_cast<size_t>(count), 0, st);
  d_rows.upload(new_to_old.data(), new_to_old.size(), 0, st);
  auto permute = [&](void *base, size_t row_bytes) {
    if (!base) return;
    const size_t bytes = static_cast<size_t>(count) * row_bytes;
    tmp.reserve(bytes, 0, st);
    gatherRowsKernel<<<static_cast<unsigned>((bytes + 255) / 256), 256, 0, st>>>(static_cast<const uint8_t *>(base), tmp.ptr, d_rows.ptr, count,
                                                                                 static_cast<int>(row_bytes));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(static_cast<uint8_t *>(base) + static_cast<size_t>(b0) * row_bytes, tmp.ptr, bytes, hipMemcpyDeviceToDevice, st));
  };
  permute(f.uv.ptr, 2 * sizeof(double));
  permute(f.idepth.ptr, sizeof(double));
  permute(f.idepth_step.ptr, sizeof(double));
  permute(f.idepth_fej.ptr, sizeof(double));
  permute(f.patch.ptr, kPat * sizeof(double));
  permute(f.inv_hdd.ptr, sizeof(double));
  permute(f.b_d.ptr, sizeof(double));
  permute(f.relative_baseline.ptr, sizeof(double));
  permute(f.n_inliers.ptr, sizeof(int32_t));
  permute(f.dflags.ptr, 1);
  if (f.ublk.ptr)
    for (int pl = 0; pl < 2 * kMaxFrames; ++pl) permute(f.ublk.ptr + static_cast<size_t>(pl) * ublkPlane(f.cap), kUblk * sizeof(double));
  for (auto &kv : f.residuals) {
    ResidualTable &rt = *kv.second;
    if (rt.n < b1) continue;  // (ends at or before b0: every list ends on a boundary)
    permute(rt.status.ptr, 1);
    permute(rt.cand.ptr, 1);
    permute(rt.fej_valid.ptr, 1);
    permute(rt.energy.ptr, sizeof(double));
  }
  // the device-side snapshot was taken in the old order: it is void (dsopp_hip_window_restore reports the missing snapshot)
  f.snap_n = 0;
  for (auto &kv : f.residuals) kv.second->snap_n = 0;
  w.snap_valid = false;
  HIP_CHECK(hipStreamSynchronize(st));  // (d_rows / tmp go out of scope)
  const std::vector<int> old_to_caller(f.to_caller.begin() + b0, f.to_caller.begin() + b1);
  for (int k = 0; k < count; ++k) {
    const int c = old_to_caller[static_cast<size_t>(new_to_old[static_cast<size_t>(k)] - b0)];
    f.to_caller[static_cast<size_t>(b0 + k)] = c;
    f.to_internal[static_cast<size_t>(c)] = b0 + k;
  }
  f.batch_end.push_back(n);
  f.d_to_internal.upload(f.to_internal.data(), f.to_internal.size(), 0, st);
  HIP_CHECK(hipStreamSynchronize(st));
  w.linearized = false;
  w.export_valid = false;
}

// First capacity of a frame's landmark arrays and of a connection table.  The reference's configurations ask for 2000 points over 7 keyframes
// (test/test_data/tummono/*.yaml) and the first keyframe of a sequence holds most of them: at 1024 three frames of the 200-frame sequence
// grew once — 11 arrays + 4 per connection table re-allocated (malloc, fill, copy, synchronise, free), ~120 of those over the run
// (DSOPP_HIP_HOST_TIMES=1 counts them per call site).  2048 entries cost 4.3 MB per frame, most of it the per-target ublk planes.
constexpr int kFirstLandmarkCapacity = 2048;

void ensureLandmarkCapacity(W &w, HostFrame &f, int n) {
  if (n <= f.cap) return;
  flushAppends(w, "flush: landmark arrays grow");  // the arrays move: queued appends hold their old addresses
  // (first allocation for kFirstLandmarkCapacity landmarks: a keyframe of the tracker gains its landmarks over several keyframes — from 256 every frame grew its
  // ~34 device arrays twice on the way, a malloc + fill + copy + synchronisation + free each)
  int cap = f.cap ? f.cap : kFirstLandmarkCapacity;
  while (cap < n) cap *= 2;
  hipStream_t st = w.sr.stream;
  const size_t keep = static_cast<size_t>(f.n);
  f.uv.reserve(2 * static_cast<size_t>(cap), 2 * keep, st);
  f.idepth.reserve(cap, keep, st);
  f.idepth_step.reserve(cap, keep, st);
  f.idepth_fej.reserve(cap, keep, st);
  f.patch.reserve(static_cast<size_t>(kPat) * cap, kPat * keep, st);
  f.inv_hdd.reserve(cap, keep, st);
  f.b_d.reserve(cap, keep, st);
  f.relative_baseline.reserve(cap, keep, st);
  f.n_inliers.reserve(cap, keep, st);
  f.dflags.reserve(cap, keep, st);
  f.ublk.reserve(2 * static_cast<size_t>(kMaxFrames) * ublkPlane(cap), 0, st);  // double-buffered (fused LM loop)
  for (auto &kv : f.residuals) {
    ResidualTable &rt = *kv.second;
    const size_t k = static_cast<size_t>(rt.n);
    rt.status.reserve(cap, k, st);
    rt.cand.reserve(cap, k, st);
    rt.fej_valid.reserve(cap, k, st);
    rt.energy.reserve(cap, k, st);
  }
  f.cap = cap;
  w.topology_dirty = true;
}

/** pinned read-back buffers grow geometrically (a window gains landmarks with every keyframe: sized exactly, the per-keyframe read-backs
 *  paid a hipHostFree + hipHostMalloc — 0.17 ms — at every keyframe of the native driver; rocprofv3 --hip-trace, profiles/r06) */
void growPinned(void *&ptr, size_t &have, size_t need) {
  if (have >= need) return;
  if (ptr) (void)hipHostFree(ptr);
  ptr = nullptr;
  size_t cap = std::max<size_t>(have, size_t(1) << 16);
  while (cap < need) cap *= 2;
  HIP_CHECK(hipHostMalloc(&ptr, cap, hipHostMallocDefault));
  have = cap;
}

/** `bytes` of pinned staging whose previous contents are no longer in flight */
void *stageAcquire(W &w, size_t bytes) {
  bytes = (bytes + 63) & ~static_cast<size_t>(63);
  if (bytes > w.stage.capacity) {
    if (w.stage.base) {
      w.sr.sync();
      (void)hipHostFree(w.stage.base);
      w.stage.base = nullptr;
    }
    const size_t cap = std::max<size_t>(bytes * 2, size_t(1) << 21);
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&w.stage.base), cap, hipHostMallocDefault));
    w.stage.capacity = cap;
    w.stage.offset = 0;
  }
  if (w.stage.offset + bytes > w.stage.capacity) {
    w.sr.sync();
    w.stage.offset = 0;
  }
  void *p = w.stage.base + w.stage.offset;
  w.stage.offset += bytes;
  return p;
}
constexpr int kAppendBlockElems = 2048;  // elements (kind 0: 8-byte words) one workgroup of applyAppendsKernel handles

/** the queued appends of a keyframe step in one launch: workgroup b works on the operation whose block range holds b */
__global__ void __launch_bounds__(256) applyAppendsKernel(const uint8_t *__restrict__ blob, const W::AppendOp *__restrict__ ops, int n_ops) {
  __shared__ int s_op;
  if (threadIdx.x == 0) {
    int lo = 0;
    for (int i = 1; i < n_ops; ++i)
      if (ops[i].first_block <= static_cast<int>(blockIdx.x)) lo = i;
    s_op = lo;
  }
  __syncthreads();
  const W::AppendOp op = ops[s_op];
  const int chunk = static_cast<int>(blockIdx.x) - op.first_block;
  const uint8_t *src = blob + op.src_off;
  const int e0 = chunk * kAppendBlockElems, e1 = e0 + kAppendBlockElems < op.n ? e0 + kAppendBlockElems : op.n;
  if (op.kind == 0) {
    // n elements of op.a bytes each (8: coordinates, inverse depths, patches; 4: row permutations); the blob entry is 8-byte aligned, the
    // destination is aligned to its element size
    if (op.a == 8) {
      for (int i = e0 + static_cast<int>(threadIdx.x); i < e1; i += 256) static_cast<unsigned long long *>(op.dst)[i] = reinterpret_cast<const unsigned long long *>(src)[i];
    } else if (op.a == 4) {
      for (int i = e0 + static_cast<int>(threadIdx.x); i < e1; i += 256) static_cast<unsigned *>(op.dst)[i] = reinterpret_cast<const unsigned *>(src)[i];
    } else {
      for (int i = e0 + static_cast<int>(threadIdx.x); i < e1; i += 256) static_cast<uint8_t *>(op.dst)[i] = src[i];
    }
    return;
  }
  if (op.kind == 1) {
    // landmark flags: {marginalized, to_marginalize} are decided on the host (LocalFrame::update), {outlier, ill_conditioned} live on the
    // device (point statuses, Schur kernel): old landmarks keep the device bits, new ones take the host's outlier bit and a cleared solver state
    uint8_t *dflags = static_cast<uint8_t *>(op.dst);
    for (int i = e0 + static_cast<int>(threadIdx.x); i < e1; i += 256) {
      const uint8_t h = src[i];
      if (i < op.a) {
        dflags[i] = static_cast<uint8_t>((dflags[i] & (kFlagOutlier | kFlagIllConditioned)) | (h & (kFlagMarginalized | kFlagToMarginalize)));
      } else {
        dflags[i] = h;
        if (op.p[0]) {
          static_cast<double *>(op.p[0])[i] = 0;
          static_cast<double *>(op.p[1])[i] = 0;
          static_cast<double *>(op.p[2])[i] = 0;
          static_cast<double *>(op.p[3])[i] = 0;
          static_cast<double *>(op.p[4])[i] = 0;
          static_cast<int32_t *>(op.p[5])[i] = 0;
        }
      }
    }
    return;
  }
  if (op.kind == 4) {
    // sweep-table entries of ONE frame pair written from the pair's template (syncTopology): entry k covers the landmarks from
    // k * step * kItemsPerBlock on, sums into row first + k, and — in the coarse table of a large window — sweeps up to `step` groups
    const SweepBlock tmpl = *reinterpret_cast<const SweepBlock *>(src);
    const int step = op.a, first = static_cast<int>(reinterpret_cast<intptr_t>(op.p[0])), n_all = static_cast<int>(reinterpret_cast<intptr_t>(op.p[1]));
    SweepBlock *out = static_cast<SweepBlock *>(op.dst);
    for (int k = e0 + static_cast<int>(threadIdx.x); k < e1; k += 256) {
      SweepBlock sb = tmpl;
      sb.offset = k * step * kItemsPerBlock;
      sb.n_groups = n_all ? (step < n_all - k * step ? step : n_all - k * step) : 1;
      sb.partial_row = first + k;
      out[k] = sb;
    }
    return;
  }
  if (op.kind == 3) {  // landmark flag bits cleared (to_marginalize after the fold-in: one operation per frame, one launch for all of them)
    uint8_t *dflags = static_cast<uint8_t *>(op.dst);
    for (int i = e0 + static_cast<int>(threadIdx.x); i < e1; i += 256) dflags[i] &= static_cast<uint8_t>(~op.a);
    return;
  }
  // kind 2 — new entries [keep, keep + n) of a connection: status from the caller, candidate = status, no FEJ cache, zero energy
  for (int i = e0 + static_cast<int>(threadIdx.x); i < e1; i += 256) {
    const uint8_t st = src[i];
    const size_t j = static_cast<size_t>(op.a) + static_cast<size_t>(i);
    static_cast<uint8_t *>(op.dst)[j] = st;
    static_cast<uint8_t *>(op.p[0])[j] = st;
    static_cast<uint8_t *>(op.p[1])[j] = 0;
    static_cast<double *>(op.p[2])[j] = 0;
  }
}

/** everything queued by set_landmarks / set_connection goes to the device: one pinned-ring copy, one launch.  Called in front of every
 *  entry point that enqueues device work on the window or reads its device arrays, and before a queued destination is reallocated. */
void flushAppends(W &w, const char *why) {
  if (w.append_ops.empty()) return;
  HostTimes ht_(why);
  const int n_ops = static_cast<int>(w.append_ops.size());
  int blocks = 0;
  for (W::AppendOp &op : w.append_ops) {
    op.first_block = blocks;
    blocks += (op.n + kAppendBlockElems - 1) / kAppendBlockElems;
  }
  const size_t data_bytes = (w.append_blob.size() + 15) & ~static_cast<size_t>(15), table_bytes = static_cast<size_t>(n_ops) * sizeof(W::AppendOp);
  hipStream_t st = w.sr.stream;
  w.d_append.reserve(data_bytes + table_bytes, 0, st);
  uint8_t *p = static_cast<uint8_t *>(stageAcquire(w, data_bytes + table_bytes));
  std::memcpy(p, w.append_blob.data(), w.append_blob.size());
  std::memcpy(p + data_bytes, w.append_ops.data(), table_bytes);
  HIP_CHECK(hipMemcpyAsync(w.d_append.ptr, p, data_bytes + table_bytes, hipMemcpyHostToDevice, st));
  applyAppendsKernel<<<static_cast<unsigned>(blocks), 256, 0, st>>>(w.d_append.ptr, reinterpret_cast<const W::AppendOp *>(w.d_append.ptr + data_bytes), n_ops);
  HIP_CHECK(hipGetLastError());
  w.append_blob.clear();
  w.append_ops.clear();
}

/** queue entry: data into the blob (8-byte aligned), flushing first when the new operation writes where a queued one does (operations
 *  of ONE launch run concurrently: two of them must never touch the same array) */
size_t queueAppendData(W &w, const void *host, size_t bytes, const void *dst_a, const void *dst_b = nullptr) {
  for (const W::AppendOp &op : w.append_ops)
    if (op.dst == dst_a || (dst_b && op.dst == dst_b)) {
      flushAppends(w, "flush: a queued operation writes the same array");
      break;
    }
  const size_t off = (w.append_blob.size() + 7) & ~static_cast<size_t>(7);
  w.append_blob.resize(off + bytes);
  if (bytes) std::memcpy(w.append_blob.data() + off, host, bytes);
  return off;
}

/** queued upload of `bytes` of host memory to a device address (applyAppendsKernel, kind 0): the descriptor tables, frame states and the
 *  marginal prior of prepare() travel with the appends — one pinned copy and one launch instead of ten copies from pageable memory and three
 *  synchronisations per prepare() (twice per keyframe: pushFrame's fold-in and solve) */
void uploadStagedBytes(W &w, void *dst, const void *host, size_t bytes) {
  if (!bytes) return;
  W::AppendOp op{};
  op.kind = 0;
  const bool words = (bytes & 7) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0;
  const bool half = (bytes & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 3) == 0;
  op.a = words ? 8 : (half ? 4 : 1);
  op.n = static_cast<int>(bytes / static_cast<size_t>(op.a));
  op.src_off = queueAppendData(w, host, bytes, dst);
  op.dst = dst;
  w.append_ops.push_back(op);
}

/** queued upload of caller memory (applyAppendsKernel, kind 0).  (Destinations of one frame's arrays are distinct allocations: the base
 *  pointer of the buffer identifies the array in the conflict test.) */
template <typename T>
void uploadStaged(W &w, DeviceBuffer<T> &dst, const T *host, size_t count, size_t offset) {
  if (!count) return;
  static_assert(sizeof(T) == 8 || sizeof(T) == 4 || sizeof(T) == 1, "element size of a queued copy");
  W::AppendOp op{};
  op.kind = 0;
  op.n = static_cast<int>(count);
  op.a = static_cast<int>(sizeof(T));
  op.src_off = queueAppendData(w, host, count * sizeof(T), dst.ptr + offset);
  op.dst = dst.ptr + offset;
  w.append_ops.push_back(op);
}


/** rebuild the FrameDev table, the sweep / Schur block tables and upload them */
void syncTopology(W &w) {
  HostTimes ht_("syncTopology");
  if (!w.topology_dirty) {
    if (w.frames_dirty && w.h_frames.size() == static_cast<size_t>(kMaxFrames) && w.d_frames.ptr) {
      // a keyframe was marked marginalised (dsopp_hip_window_mark_frame_marginalized): nothing the sweep / Schur tables hold has changed —
      // they carry pointers and counts — only two flags of the frame table: patched and sent alone (a full rebuild is 25 us on the host
      // and 400 KB of tables through the queue, once per keyframe in front of the reference depth maps)
      for (int r = 0; r < w.F(); ++r) {
        w.h_frames[static_cast<size_t>(r)].is_marginalized = w.frames[static_cast<size_t>(r)]->is_marginalized;
        w.h_frames[static_cast<size_t>(r)].to_marginalize = w.frames[static_cast<size_t>(r)]->to_marginalize;
      }
      uploadStagedBytes(w, w.d_frames.ptr, w.h_frames.data(), static_cast<size_t>(kMaxFrames) * sizeof(FrameDev));
    }
    w.frames_dirty = false;
    return;
  }
  hipStream_t st = w.sr.stream;
  const int F = w.F();
  // (scratch kept per thread: the tables are rebuilt twice per keyframe — pushFrame's fold-in and solve — and 500 vector constructions +
  // their growth were a good part of the 125 us a rebuild cost on the host; DSOPP_HIP_HOST_TIMES)
  static thread_local std::vector<FrameDev> fd;
  static thread_local std::vector<SweepBlock> sweep;  // (listed on the host only by the XCD-banded launch order experiment)
  static thread_local std::vector<SchurBlock> schur;
  static thread_local std::vector<SweepBlock> pair_tmpl;  // one template per frame pair (slot r * kMaxFrames + t)
  static thread_local std::vector<int> pair_items;       // its entries of kItemsPerBlock items (0: no residuals)
  fd.assign(static_cast<size_t>(kMaxFrames), FrameDev{});
  std::memset(fd.data(), 0, fd.size() * sizeof(FrameDev));
  sweep.clear();
  schur.clear();
  pair_tmpl.resize(kMaxFrames * kMaxFrames);
  pair_items.assign(kMaxFrames * kMaxFrames, 0);
  // large windows: a sweep workgroup takes 4 groups of 16 items (pba_kernels.hpp: SweepBlock::n_groups)
  size_t total_items = 0;
  for (int r = 0; r < F; ++r)
    for (const auto &kv : w.frames[static_cast<size_t>(r)]->residuals) total_items += static_cast<size_t>(kv.second->n);
  static const int groups_override = std::getenv("DSOPP_HIP_SWEEP_GROUPS") ? std::atoi(std::getenv("DSOPP_HIP_SWEEP_GROUPS")) : 0;  // tuning aid
  // measured: 7 KF / 20k points (120k items) 36.8 / 33.1 / 38.8 us at 2 / 4 / 6 groups; 12 KF / 50k points (550k items) 143 / 125 / 119 us
  const int groups = groups_override > 0 ? groups_override : (total_items >= 300000 ? 6 : (total_items >= 30000 ? 4 : 1));
  std::vector<int> pair_first(kMaxFrames * kMaxFrames, -1), pair_count(kMaxFrames * kMaxFrames, 0);
  for (int r = 0; r < F; ++r) {
    HostFrame &f = *w.frames[static_cast<size_t>(r)];
    FrameDev &d = fd[static_cast<size_t>(r)];
    const LevelView lv = f.pyramid->view(f.level);
    d.texels = hbm(lv.texels);
    static const bool no_iplane = std::getenv("DSOPP_HIP_NO_IPLANE") != nullptr;  // tuning aid (A/B of the counter traffic)
    d.iplane = no_iplane ? nullptr : hbm(f.pyramid->intensityPlane(f.level, st));
    f.pyramid_generation = f.pyramid->generation;
    d.itiles = f.pyramid->itilesX(f.level);
    d.width = lv.width;
    d.height = lv.height;
    d.fx = f.intr[0];
    d.fy = f.intr[1];
    d.cx = f.intr[2];
    d.cy = f.intr[3];
    d.exposure = f.exposure;
    d.fixed = f.fixed;
    d.is_marginalized = f.is_marginalized;
    d.to_marginalize = f.to_marginalize;
    d.n = f.n;
    d.cap = f.cap;
    d.uv = hbm(f.uv.ptr);
    d.idepth = hbm(f.idepth.ptr);
    d.idepth_step = hbm(f.idepth_step.ptr);
    d.idepth_fej = hbm(f.idepth_fej.ptr);
    d.patch = hbm(f.patch.ptr);
    d.inv_hdd = hbm(f.inv_hdd.ptr);
    d.b_d = hbm(f.b_d.ptr);
    d.relative_baseline = hbm(f.relative_baseline.ptr);
    d.n_inliers = hbm(f.n_inliers.ptr);
    d.flags = hbm(f.dflags.ptr);
    d.ublk = hbm(f.ublk.ptr);
    d.snap_idepth = hbm(f.snap_idepth.ptr);
    d.snap_flags = hbm(f.snap_flags.ptr);
    d.first_conn = -1;
    for (int t = 0; t < F; ++t) {
      if (t == r) continue;
      auto it = f.residuals.find(w.frames[static_cast<size_t>(t)]->id);
      if (it == f.residuals.end() || it->second->n == 0) continue;
      ResidualTable &rt = *it->second;
      d.status[t] = hbm(rt.status.ptr);
      d.cand[t] = hbm(rt.cand.ptr);
      d.fej_valid[t] = hbm(rt.fej_valid.ptr);
      d.energy[t] = hbm(rt.energy.ptr);
      d.n_res[t] = rt.n;
      d.snap_status[t] = hbm(rt.snap_status.ptr);
      if (d.first_conn < 0) d.first_conn = t;
      // the pair's entries differ in three fields (first landmark, row of the sums, groups of the coarse table): the host states ONE
      // template per pair, the entries are written from it — on the device (applyAppendsKernel, kind 4), or below for the experiment that
      // re-orders them.  (Listing them here and sending them whole was 25 us of host time and 350 KB through the queue per rebuild, twice per
      // keyframe of the tracker.)
      SweepBlock sb;
      std::memset(&sb, 0, sizeof(sb));
      sb.r = r;
      sb.t = t;
      sb.n_res = rt.n;
      sb.cap = f.cap;
      sb.owns_landmark_sums = d.first_conn == t ? 1 : 0;
      sb.width_r = lv.width;
      sb.height_r = lv.height;
      sb.uv = d.uv;
      sb.idepth = d.idepth;
      sb.patch = d.patch;
      sb.idepth_fej = d.idepth_fej;
      sb.b_d = d.b_d;
      sb.inv_hdd = d.inv_hdd;
      sb.idepth_step = d.idepth_step;
      sb.ublk = d.ublk;
      sb.energy = hbm(rt.energy.ptr);
      sb.flags = d.flags;
      sb.status = hbm(rt.status.ptr);
      sb.fej_valid = hbm(rt.fej_valid.ptr);
      sb.cand = hbm(rt.cand.ptr);
      sb.n_groups = 1;
      pair_tmpl[static_cast<size_t>(r * kMaxFrames + t)] = sb;
      pair_items[static_cast<size_t>(r * kMaxFrames + t)] = (rt.n + kItemsPerBlock - 1) / kItemsPerBlock;
    }
    for (int off = 0; off < f.n; off += kSchurLandmarks) {
      SchurBlock sb;
      std::memset(&sb, 0, sizeof(sb));
      sb.r = r;
      sb.offset = off;
      sb.n = d.n;
      sb.cap = d.cap;
      sb.fixed = d.fixed;
      sb.idepth = d.idepth;
      sb.idepth_step = d.idepth_step;
      sb.inv_hdd = d.inv_hdd;
      sb.b_d = d.b_d;
      sb.ublk = d.ublk;
      sb.flags = d.flags;
      for (int t = 0; t < F; ++t) {
        if (d.status[t] == nullptr) continue;
        sb.conn_mask |= 1u << t;
        sb.status[t] = d.status[t];
        sb.cand[t] = d.cand[t];
        sb.n_res[t] = d.n_res[t];
      }
      schur.push_back(sb);
    }
  }
  // Order of the sweep table: TARGET-major — all pairs that sample the same target image run back to back, so that an XCD's L2 (4 MB,
  // against 9.8 MB of texels per 640 x 480 image) sees the second and later visits of a texel line while it may still hold it: the
  // sweep is bound by the rate at which the fabric serves randomly placed 64-byte requests, and every L2 hit is one
  // request less.  (Reference-major until round 4: consecutive pairs switched the image.)  DSOPP_HIP_SWEEP_ORDER=rt restores it (A/B).
  static const bool reference_major = std::getenv("DSOPP_HIP_SWEEP_ORDER") != nullptr && std::string(std::getenv("DSOPP_HIP_SWEEP_ORDER")) == "rt";
  // entry k of a pair in the sweep table / in the fine table (what applyAppendsKernel's kind 4 writes on the device)
  const int step = groups > 1 ? groups : 1;
  auto sweepEntry = [&](const SweepBlock &tmpl, int n_all, int first, int k) {
    SweepBlock sb = tmpl;
    sb.offset = k * step * kItemsPerBlock;
    sb.n_groups = groups > 1 ? std::min(step, n_all - k * step) : 1;
    sb.partial_row = first + k;
    return sb;
  };
  struct PairRange {
    size_t pi;
    int first_sweep, n_sweep, first_fine, n_fine;
  };
  static thread_local std::vector<PairRange> ranges;
  ranges.clear();
  int total_sweep = 0, total_fine = 0;
  for (int outer = 0; outer < F; ++outer)
    for (int inner = 0; inner < F; ++inner) {
      const int r = reference_major ? outer : inner, t = reference_major ? inner : outer;
      const size_t pi = static_cast<size_t>(r * kMaxFrames + t);
      const int n_all = pair_items[pi];
      if (n_all == 0) continue;
      SweepBlock &tm = pair_tmpl[pi];
      const FrameDev &dr = fd[static_cast<size_t>(r)], &dt = fd[static_cast<size_t>(t)];
      tm.width_t = dt.width;
      tm.height_t = dt.height;
      tm.texels_t = dt.texels;
      tm.iplane_t = dt.iplane;
      tm.itiles_t = dt.itiles;
      for (int k = 0; k < F; ++k)
        if (dr.status[k] != nullptr) tm.conn_mask |= 1u << k;
      const int n_sweep = (n_all + step - 1) / step;
      pair_first[pi] = total_sweep;
      pair_count[pi] = n_sweep;
      ranges.push_back({pi, total_sweep, n_sweep, total_fine, groups > 1 ? n_all : 0});
      total_sweep += n_sweep;
      total_fine += groups > 1 ? n_all : 0;
    }
  // EXPERIMENT (DSOPP_HIP_SWEEP_XCD_BANDS=1, off by default): launch order such that XCD x (= blockIdx % 8 under round-robin dispatch)
  // sweeps the x-th eighth of every pair's landmarks.  With landmarks in a spatial order (rows of the image) an XCD then samples one
  // band of a target image from all reference frames — 1.2 MB of texels, which its 4 MB L2 holds — instead of the whole image.
  // Measures how much of the 2.6-fold re-use of texel lines across pairs an L2-aware order could turn into hits.
  // A measurement aid for the sweep only: the point-status kernels walk the same table, padding entries included, and solve()'s statuses /
  // inlier counts then differ from the checker's (tests/test_gpu_fullres.py::test_full_solve_parity_at_1280x1024 fails under the switch,
  // as it did when the experiment was built in round 5).
  static const bool xcd_bands = std::getenv("DSOPP_HIP_SWEEP_XCD_BANDS") != nullptr && std::atoi(std::getenv("DSOPP_HIP_SWEEP_XCD_BANDS")) != 0;
  if (xcd_bands)  // (this experiment re-orders the entries: they are listed on the host, as all tables were until round 6)
    for (const PairRange &pr : ranges)
      for (int k = 0; k < pr.n_sweep; ++k) sweep.push_back(sweepEntry(pair_tmpl[pr.pi], pair_items[pr.pi], pr.first_sweep, k));
  if (xcd_bands && !sweep.empty()) {
    std::vector<std::vector<int>> queue(8);
    for (size_t b = 0; b < sweep.size(); ++b) {
      const SweepBlock &sb = sweep[b];
      const int band = std::min(7, static_cast<int>((static_cast<long long>(sb.offset) * 8) / std::max(1, sb.n_res)));
      queue[static_cast<size_t>(band)].push_back(static_cast<int>(b));
    }
    size_t longest = 0;
    for (const auto &q : queue) longest = std::max(longest, q.size());
    std::vector<SweepBlock> launch;
    int pad_row = static_cast<int>(sweep.size());
    for (size_t k = 0; k < longest; ++k)
      for (int x = 0; x < 8; ++x) {
        if (k < queue[static_cast<size_t>(x)].size()) {
          launch.push_back(sweep[static_cast<size_t>(queue[static_cast<size_t>(x)][k])]);
        } else {  // a no-op entry keeps the XCD's turn (zero sums into a row of its own)
          SweepBlock sb = sweep[0];
          sb.n_groups = 0;
          sb.partial_row = pad_row++;
          launch.push_back(sb);
        }
      }
    sweep.swap(launch);
  }
  w.d_frames.reserve(kMaxFrames, 0, st);
  uploadStagedBytes(w, w.d_frames.ptr, fd.data(), (kMaxFrames) * sizeof(*w.d_frames.ptr));
  w.h_frames = fd;
  w.frames_dirty = false;
  // one queued operation per pair and table: its template travels (176 bytes), the device writes the entries
  auto queuePairEntries = [&](SweepBlock *table, const SweepBlock &tmpl, int first, int count, int entry_step, int n_all_for_groups) {
    W::AppendOp op{};
    op.kind = 4;
    op.n = count;
    op.a = entry_step;
    op.src_off = queueAppendData(w, &tmpl, sizeof(SweepBlock), table + first);
    op.dst = table + first;
    op.p[0] = reinterpret_cast<void *>(static_cast<intptr_t>(first));
    op.p[1] = reinterpret_cast<void *>(static_cast<intptr_t>(n_all_for_groups));
    w.append_ops.push_back(op);
  };
  const size_t n_sweep_entries = xcd_bands ? sweep.size() : static_cast<size_t>(total_sweep);
  w.n_sweep_blocks = static_cast<int>(n_sweep_entries);
  w.n_schur_blocks = static_cast<int>(schur.size());
  if (n_sweep_entries > w.d_sweep_table.capacity || (groups > 1 && static_cast<size_t>(total_fine) > w.d_fine_table.capacity))
    flushAppends(w, "flush: sweep tables grow");  // (nothing queued may still point into a table that moves)
  w.d_sweep_table.reserve(std::max<size_t>(1, n_sweep_entries), 0, st);
  if (xcd_bands) {
    uploadStagedBytes(w, w.d_sweep_table.ptr, sweep.data(), (sweep.size()) * sizeof(*w.d_sweep_table.ptr));
  } else {
    for (const PairRange &pr : ranges)
      queuePairEntries(w.d_sweep_table.ptr, pair_tmpl[pr.pi], pr.first_sweep, pr.n_sweep, step, groups > 1 ? pair_items[pr.pi] : 0);
  }
  if (groups > 1) {
    w.d_fine_table.reserve(std::max<size_t>(1, static_cast<size_t>(total_fine)), 0, st);
    for (const PairRange &pr : ranges) queuePairEntries(w.d_fine_table.ptr, pair_tmpl[pr.pi], pr.first_fine, pr.n_fine, 1, 0);
    w.n_fine_blocks = total_fine;
  } else {
    w.d_fine_table.release();
    w.n_fine_blocks = static_cast<int>(n_sweep_entries);
  }
  w.d_schur_table.reserve(std::max<size_t>(1, schur.size()), 0, st);
  uploadStagedBytes(w, w.d_schur_table.ptr, schur.data(), (schur.size()) * sizeof(*w.d_schur_table.ptr));
  w.d_pair_first.reserve(kMaxFrames * kMaxFrames, 0, st);
  uploadStagedBytes(w, w.d_pair_first.ptr, pair_first.data(), (pair_first.size()) * sizeof(*w.d_pair_first.ptr));
  w.d_pair_count.reserve(kMaxFrames * kMaxFrames, 0, st);
  uploadStagedBytes(w, w.d_pair_count.ptr, pair_count.data(), (pair_count.size()) * sizeof(*w.d_pair_count.ptr));
  w.d_partials.reserve(std::max<size_t>(1, n_sweep_entries) * kPartial, 0, st);
  w.d_pc.reserve(kMaxFrames * kMaxFrames, 0, st);
  w.d_ctrl.reserve(2, 0, st);
  const size_t KK = static_cast<size_t>(kBlk * kMaxFrames);
  w.d_Hpp.reserve(2 * KK * KK, 0, st);  // (second half: full symmetric copy of H_schur for the covariance read-back, makeSolveArgs)
  w.d_bpp.reserve(KK, 0, st);
  // + tail: the sweep's energy scalars sit behind the combined system (4 sums, or 4 x kScalarGroups group sums) and ride in the same
  // collective on landmark-sharded windows
  w.d_reduce.reserve(2 * (KK * KK + KK) + 8 + 4 * kScalarGroups + 16 + static_cast<size_t>(kMaxCombCopies - 1) * (static_cast<size_t>(combBlockCount(kMaxFrames)) * 64 + KK + 2), 0, st);
  w.d_Hm.reserve(KK * KK, 0, st);
  w.d_bm.reserve(KK, 0, st);
  w.d_step.reserve(KK, 0, st);
  w.d_scalars.reserve(16 + 4 * kScalarGroups, 0, st);
  // (the tables above are queued uploads — copied into the append blob — and leave with prepare()'s flush: no synchronisation here)
  w.topology_dirty = false;
  w.pair_valid = false;
}

void uploadState(W &w) {
  if (!w.state_dirty) return;
  w.d_state.reserve(1, 0, w.sr.stream);
  uploadStagedBytes(w, w.d_state.ptr, &w.hst, sizeof(WindowState));  // (queued: leaves with prepare()'s flush)
  w.state_dirty = false;
  w.pair_valid = false;
}

void downloadState(W &w) {
  if (w.state_dirty || !w.d_state.ptr || !w.host_stale) return;
  w.host_stale = false;
  w.d_state.download(&w.hst, 1, 0, w.sr.stream);
  w.sr.sync();
}

void uploadMarginal(W &w) {
  if (!w.marg_dirty) return;
  const int K = w.K();
  if (w.marg_size != K) fail(DSOPP_HIP_ERR_STATE, "marginal prior has size %d, window %d", w.marg_size, K);
  w.marg_nonzero = w.energy_marginalized != 0;
  for (double v : w.Hm) w.marg_nonzero = w.marg_nonzero || v != 0;
  for (double v : w.bm) w.marg_nonzero = w.marg_nonzero || v != 0;
  uploadStagedBytes(w, w.d_Hm.ptr, w.Hm.data(), static_cast<size_t>(K) * K * sizeof(double));
  uploadStagedBytes(w, w.d_bm.ptr, w.bm.data(), static_cast<size_t>(K) * sizeof(double));
  // ... and the matrix in the combined system's own layout (8 x 8 blocks of the lower triangle, row-major inside: combBlockIndex), which
  // the fused loop's solve launch adds while it loads the system (SolveCombArgs::HmPacked)
  const int F = w.F();
  std::vector<double> packed(static_cast<size_t>(combBlockCount(F)) * 64);
  for (int bi = 0; bi < F; ++bi)
    for (int bj = 0; bj <= bi; ++bj)
      for (int i = 0; i < kBlk; ++i)
        for (int j = 0; j < kBlk; ++j)
          packed[static_cast<size_t>(combBlockIndex(bi, bj)) * 64 + static_cast<size_t>(i * kBlk + j)] =
              w.Hm[static_cast<size_t>(kBlk * bi + i) * K + static_cast<size_t>(kBlk * bj + j)];
  w.d_Hm_packed.reserve(static_cast<size_t>(combBlockCount(kMaxFrames)) * 64, 0, w.sr.stream);
  uploadStagedBytes(w, w.d_Hm_packed.ptr, packed.data(), packed.size() * sizeof(double));
  w.marg_dirty = false;
}

/** a pyramid a keyframe borrows was rewritten (build / set_level / set_mask) since the sweep descriptors were made: the residual-only
 *  sweeps would keep sampling the old intensity plane while the linearising sweeps read the new texels — rebuild the descriptors */
void checkPyramidGenerations(W &w) {
  for (const auto &fp : w.frames)
    if (fp->pyramid && fp->pyramid->generation != fp->pyramid_generation) w.topology_dirty = true;
}

void prepare(W &w) {
  w.sr.use();
  // (queued landmark / connection appends are NOT flushed here: nothing below reads the device, and what it queues — tables, states, the prior —
  // goes to other arrays; everything leaves with the one flush at the end.  A flush at entry as well was a second copy + launch per call)
  checkPyramidGenerations(w);
  downloadState(w);  // no-op unless a device-driven solve left the host mirror behind
  syncTopology(w);
  uploadState(w);
  uploadMarginal(w);
  flushAppends(w, "flush: prepare()");  // the appends of the keyframe step + the tables, states and prior queued by the three calls above
}

/** prepare() for steps that only touch device state: when nothing on the host is newer than the device (no pending state,
 *  topology or prior upload) the device is current as it is, and the host mirror may stay behind a device-driven solve —
 *  its refresh (a read-back + synchronisation) is left to the first reader */
void prepareDevice(W &w) {
  w.sr.use();
  flushAppends(w);
  checkPyramidGenerations(w);
  if (w.state_dirty || w.topology_dirty || w.frames_dirty || w.marg_dirty || !w.d_state.ptr) prepare(w);
}

void ensurePairConstants(W &w) {
  if (w.pair_valid) return;
  const int F = w.F();
  timedLaunch(w, DSOPP_HIP_KERNEL_PAIR_SETUP, [&] {
    pairSetupKernel<<<1, kMaxFrames * kMaxFrames, 0, w.sr.stream>>>(w.d_frames.ptr, w.d_state.ptr, w.d_pc.ptr, F, w.fej() ? 1 : 0, nullptr);
  });
  HIP_CHECK(hipGetLastError());
  w.pair_valid = true;
}

template <typename S>
void launchFej(W &w) {
  if (!w.n_sweep_blocks) return;
  timedLaunch(w, DSOPP_HIP_KERNEL_FEJ,
              [&] {
                const int per_wave = 64 / kItemsPerBlock;
                fejKernel<S><<<(w.n_fine_blocks + per_wave - 1) / per_wave, 64, 0, w.sr.stream>>>(w.d_frames.ptr, w.d_pc.ptr, w.fineTable(),
                                                                                                  w.n_fine_blocks);
              });
  HIP_CHECK(hipGetLastError());
}

/** firstEstimateJacobians — PROB_SRC/photometric_bundle_adjustment.cpp:302-305 */
void firstEstimate(W &w) {
  ensurePairConstants(w);
  if (w.opt.dtype == DSOPP_HIP_F64)
    launchFej<double>(w);
  else
    launchFej<float>(w);
}

/** extras of the fused device loop (all zero / null for the plain launches) */
struct SweepExtras {
  int ublk_read = 0, ublk_write = 0;
  bool gate_on_pending = false;  // fused loop: BACKSUB only when a candidate step is pending (LmControl::pending)
  const int *run_flag = nullptr;
  bool fused_lin_backsub = false;
  bool combined = false;  // the following reduction builds the combined system: only that much has to be zeroed
  bool external_backsub = false;  // calculateIdepths ran in backsubKernel in front of this sweep (large windows)
  bool write_fej = false;         // opening linearisation of a fused solve: the sweep takes the first-estimate snapshot itself
};

template <typename S>
void launchSweepTyped(W &w, bool lin, bool huber, bool for_marg, const LmControl *ctrl, bool backsub, double lambda, const SweepExtras &ex) {
  if (!w.n_sweep_blocks) {
    if (lin) HIP_CHECK(hipMemsetAsync(w.d_reduce.ptr, 0, w.reduceCount() * sizeof(double), w.sr.stream));
    return;
  }
  SweepParams prm;
  prm.sigma_huber = w.opt.sigma_huber_loss;
  prm.for_marginalized = for_marg ? 1 : 0;
  prm.use_fej_flag = w.fej() ? 1 : 0;
  prm.ctrl = ctrl;
  prm.clear_buf = lin ? w.d_reduce.ptr : nullptr;
  prm.clear_count = static_cast<int>(ex.combined ? w.combCount() + 4 : w.reduceCount());
  if (ex.combined && w.comb_copies_active > 1)  // (the copies lie behind the scalar groups' slots: everything up to the last one)
    prm.clear_count = static_cast<int>(w.combCopyFirst() + (w.comb_copies_active - 1) * w.combCopyStride());
  prm.step = w.d_step.ptr;
  prm.lambda = lambda;
  prm.F = w.F();
  prm.ublk_read = ex.ublk_read;
  prm.ublk_write = ex.ublk_write;
  prm.gate_on_pending = ex.gate_on_pending ? 1 : 0;
  prm.run_flag = ex.run_flag;
  prm.external_backsub = ex.external_backsub ? 1 : 0;
  prm.dbg = (w.dbg_sweep && lin == w.dbg_sweep_lin) ? w.dbg_sweep : nullptr;
  dim3 grid(static_cast<unsigned>(w.n_sweep_blocks)), block(kSweepThreads);
  hipStream_t st = w.sr.stream;
  const FrameDev *fr = w.d_frames.ptr;
  const PairConst *pc = w.d_pc.ptr;
  const SweepBlock *tb = w.d_sweep_table.ptr;
  double *pa = w.d_partials.ptr;
  const bool opening = lin && huber && ex.write_fej && w.fej();
  if (opening) prm.external_backsub = 1;  // (nothing is pending in the opening round: only the state norms are recorded)
  timedLaunch(w, lin ? DSOPP_HIP_KERNEL_SWEEP_LINEARIZE : DSOPP_HIP_KERNEL_SWEEP_ENERGY, [&] {
    if (opening) {
      // opening linearisation of a fused solve: first-estimate snapshot taken by the sweep itself, no back-substitution
      sweepKernel<S, true, true, true, false, true><<<grid, block, 0, st>>>(fr, pc, tb, pa, prm.ctrl, prm.run_flag, prm);
    } else if (lin && ex.fused_lin_backsub) {
      // fused LM loop: linearisation at the candidate state = its energy evaluation + calculateIdepths in one pass
      if (w.fej())
        sweepKernel<S, true, true, true, true><<<grid, block, 0, st>>>(fr, pc, tb, pa, prm.ctrl, prm.run_flag, prm);
      else
        sweepKernel<S, true, false, true, true><<<grid, block, 0, st>>>(fr, pc, tb, pa, prm.ctrl, prm.run_flag, prm);
    } else if (!lin && backsub) {
      if (w.fej())
        sweepKernel<S, false, true, true, true><<<grid, block, 0, st>>>(fr, pc, tb, pa, prm.ctrl, prm.run_flag, prm);
      else
        sweepKernel<S, false, false, true, true><<<grid, block, 0, st>>>(fr, pc, tb, pa, prm.ctrl, prm.run_flag, prm);
    } else if (!lin) {
      if (w.fej())
        sweepKernel<S, false, true, true><<<grid, block, 0, st>>>(fr, pc, tb, pa, prm.ctrl, prm.run_flag, prm);
      else
        sweepKernel<S, false, false, true><<<grid, block, 0, st>>>(fr, pc, tb, pa, prm.ctrl, prm.run_flag, prm);
    } else if (w.fej()) {
      if (huber)
        sweepKernel<S, true, true, true><<<grid, block, 0, st>>>(fr, pc, tb, pa, prm.ctrl, prm.run_flag, prm);
      else
        sweepKernel<S, true, true, false><<<grid, block, 0, st>>>(fr, pc, tb, pa, prm.ctrl, prm.run_flag, prm);
    } else {
      if (huber)
        sweepKernel<S, true, false, true><<<grid, block, 0, st>>>(fr, pc, tb, pa, prm.ctrl, prm.run_flag, prm);
      else
        sweepKernel<S, true, false, false><<<grid, block, 0, st>>>(fr, pc, tb, pa, prm.ctrl, prm.run_flag, prm);
    }
  });
  HIP_CHECK(hipGetLastError());
}

void launchSweep(W &w, bool lin, bool huber, bool for_marg, const LmControl *ctrl = nullptr, bool backsub = false, double lambda = 0,
                 const SweepExtras &ex = SweepExtras()) {
  if (!ctrl && !ex.run_flag) ensurePairConstants(w);
  backsub = backsub && w.opt.optimize_idepths;
  if (w.opt.dtype == DSOPP_HIP_F64)
    launchSweepTyped<double>(w, lin, huber, for_marg, ctrl, backsub, lambda, ex);
  else
    launchSweepTyped<float>(w, lin, huber, for_marg, ctrl, backsub, lambda, ex);
}

void allreduceIfNeeded(W &w, double *dev, size_t count) {
  if (!w.allreduce) return;
  const int rc = w.allreduce(w.allreduce_user, dev, count, w.sr.stream);
  if (rc != 0) fail(DSOPP_HIP_ERR_HIP, "allreduce callback failed with %d", rc);
}

size_t schurSmemBytes(int K) {
  const int Kp = (K + 15) & ~15;
  return (static_cast<size_t>(kSchurLandmarks) * schurRowStride(Kp) + 2 * kSchurLandmarks + 40 * static_cast<size_t>(kMaxFrames)) * sizeof(double);
}

struct FusedReduce {
  int ublk_parity;
  LmControl *ctrl_out;
  LmParams prm;
  bool combined = false;   // emit the combined block-packed system instead of H_pp / H_schur (fused loop)
  double comb_lambda = 0;  // its damping when no control block is given
  const double *scalars = nullptr;  // decide-only launches: where the (group) sums of the sweep's scalars are
  LmControl *ctrl_host = nullptr;   // closing round: pinned host destination of the final control block
};

/** K2: per-pair reduction + Schur complement (+ the cross-rank sum of everything that is a sum over landmarks);
 *  in the fused loop it starts with the LM decision for the pending candidate */
enum class ReduceMode { kFused, kAccumulateOnly, kDecideOnly };
constexpr size_t kDecideSmemBytes = size_t((6 * (kSchurThreads + 2) + 6 * 72) * 8);

/** arguments of the LM decision (fusedDecideApply + applyDecision) when it runs as the prologue of another kernel */
void launchTwoStage(W &w, const LmControl *ctrl, int ublk_parity, double lambda, bool dense, double *group_sums);
void launchReduceSchur(W &w, bool for_marg, const LmControl *ctrl, const FusedReduce *fused = nullptr, ReduceMode mode = ReduceMode::kFused) {
  const int K = w.K(), F = w.F();
  hipStream_t st = w.sr.stream;
  if (!fused && !for_marg && !ctrl && w.deterministic) {
    // stage API / host-driven loop under dsopp_hip_window_set_deterministic: the four dense arrays without atomics
    launchTwoStage(w, nullptr, 0, 0.0, true, nullptr);
    allreduceIfNeeded(w, w.d_reduce.ptr, w.reduceCount());
    return;
  }
  ensureDynamicLds(reinterpret_cast<const void *>(reduceSchurKernel), w.sr.device, 96 * 1024);
  ReduceSchurArgs a;
  a.frames = w.d_frames.ptr;
  a.pc = w.d_pc.ptr;
  a.schur_table = w.d_schur_table.ptr;
  a.partials = w.d_partials.ptr;
  a.pair_first_block = w.d_pair_first.ptr;
  a.pair_num_blocks = w.d_pair_count.ptr;
  a.Hpp = w.dHppRaw();
  a.bpp = w.dbppRaw();
  a.Hsc = w.dHsc();
  a.bsc = w.dbsc();
  a.ctrl = ctrl;
  a.F = F;
  a.n_schur_blocks = w.opt.optimize_idepths ? w.n_schur_blocks : 0;
  a.for_marginalized = for_marg ? 1 : 0;
  a.ublk_parity = fused ? fused->ublk_parity : 0;
  a.ctrl_out = (fused && mode != ReduceMode::kAccumulateOnly) ? fused->ctrl_out : nullptr;
  a.st = w.d_state.ptr;
  const bool combined = fused && fused->combined;
  const size_t reduce_count = combined ? w.combCount() : w.reduceCount();
  a.comb = combined ? w.d_reduce.ptr : nullptr;
  a.comb_lambda = fused ? fused->comb_lambda : 0.0;
  if (combined && mode == ReduceMode::kAccumulateOnly && w.comb_copies_active > 1) {
    a.comb_copies = w.comb_copies_active;
    a.comb_copy_first = static_cast<int>(w.combCopyFirst());
    a.comb_copy_stride = static_cast<int>(w.combCopyStride());
  }
  a.scalars = mode == ReduceMode::kDecideOnly ? w.d_reduce.ptr + reduce_count : w.d_scalars.ptr;
  a.n_sweep_blocks = w.n_sweep_blocks;
  a.total_blocks = a.n_schur_blocks + F * F;
  a.scalars_out = mode == ReduceMode::kAccumulateOnly ? w.d_reduce.ptr + reduce_count : nullptr;
  // landmark shards: whether a shard builds with atomics or in two stages is decided by ITS chunk count, which may differ between
  // ranks by a chunk — so both paths hand the collective the same thing: [system | kScalarGroups groups of four sums]
  const bool grouped_tail = mode == ReduceMode::kAccumulateOnly && w.allreduce != nullptr;
  a.scalars_out_groups = grouped_tail ? kScalarGroups : 0;
  static_assert(4 * kScalarGroups <= kSchurThreads, "one workgroup writes the grouped tail");
  if (fused) a.prm = fused->prm;
  if (fused && fused->scalars) a.scalars = fused->scalars;
  a.ctrl_host = fused ? fused->ctrl_host : nullptr;
  a.dbg = w.dbg_stamps ? w.dbg_stamps + 24 : nullptr;
  const size_t decide_smem = kDecideSmemBytes;
  if (mode == ReduceMode::kDecideOnly) {
    timedLaunch(w, DSOPP_HIP_KERNEL_ACCEPT,
                [&] { decideApplyKernel<<<std::max(1, a.n_schur_blocks), kSchurThreads, decide_smem, st>>>(a); });
    HIP_CHECK(hipGetLastError());
    return;
  }
  timedLaunch(w, DSOPP_HIP_KERNEL_SCHUR, [&] {
    // both systems are accumulated with atomics into d_reduce, which the preceding linearisation sweep zeroed
    reduceSchurKernel<<<a.n_schur_blocks + F * F + (a.scalars_out ? 1 : 0), kSchurThreads, std::max(schurSmemBytes(K), decide_smem), st>>>(
        a.ctrl, a.schur_table, a.pc, a.partials, a.pair_first_block, a.pair_num_blocks, a.n_schur_blocks, a.F, a);
  });
  HIP_CHECK(hipGetLastError());
  // multi-GPU: landmarks are sharded, so both systems are partial sums: one collective over one contiguous buffer
  // (in the fused loop the 4 energy scalars of the sweep sit right behind the systems and travel with them)
  allreduceIfNeeded(w, w.d_reduce.ptr, reduce_count + (mode == ReduceMode::kAccumulateOnly ? (grouped_tail ? 4 * kScalarGroups : 4) : 0));
  (void)K;
}

SolveArgs makeSolveArgs(W &w) {
  SolveArgs a;
  a.frames = w.d_frames.ptr;
  a.st = w.d_state.ptr;
  a.pc = w.d_pc.ptr;
  a.Hpp_raw = w.dHppRaw();
  a.bpp_raw = w.dbppRaw();
  a.Hpp_out = w.d_Hpp.ptr;
  a.Hsc_copy = w.d_Hpp.ptr + static_cast<size_t>(w.K()) * w.K();  // written with store_system: [H_pp | H_schur] leave in one transfer
  a.bpp_out = w.d_bpp.ptr;
  a.Hsc = w.dHsc();
  a.bsc = w.dbsc();
  a.use_marginal = w.marg_nonzero ? 1 : 0;
  a.dbg_stamps = w.dbg_stamps;
  a.Hm = w.d_Hm.ptr;
  a.bm = w.d_bm.ptr;
  a.step = w.d_step.ptr;
  a.ctrl = nullptr;
  a.lambda = 0;
  a.affine_reg[0] = w.opt.affine_brightness_regularizer[0];
  a.affine_reg[1] = w.opt.affine_brightness_regularizer[1];
  a.fixed_reg = w.opt.fixed_state_regularizer;
  a.energy_marginalized = w.energy_marginalized;
  a.F = w.F();
  a.fej = w.fej() ? 1 : 0;
  a.do_solve = 1;
  a.store_system = 0;
  a.add_priors = 1;
  return a;
}

size_t solveSmemBytes(int K) {
  const size_t N = static_cast<size_t>(K) + 1;
  return (N * (N + 1) + 4 * static_cast<size_t>(K) + 32 + 38 * static_cast<size_t>(kMaxFrames)) * sizeof(double);
}

/** K3 */
void launchAssemble(W &w, double lambda, bool do_solve, bool add_priors, bool store_system, LmControl *ctrl) {
  ensureDynamicLds(reinterpret_cast<const void *>(assembleSolveKernel), w.sr.device, 150 * 1024);
  SolveArgs a = makeSolveArgs(w);
  a.lambda = lambda;
  a.ctrl = ctrl;
  a.do_solve = do_solve ? 1 : 0;
  a.add_priors = add_priors ? 1 : 0;
  a.store_system = store_system ? 1 : 0;
  timedLaunch(w, do_solve ? DSOPP_HIP_KERNEL_ASSEMBLE_SOLVE : DSOPP_HIP_KERNEL_ASSEMBLE,
              [&] { assembleSolveKernel<<<1, kSolveThreads, solveSmemBytes(w.K()), w.sr.stream>>>(a); });
  if (do_solve && !w.fej()) {
    // no first-estimate Jacobians: all pair constants follow the candidate state eps + step the solve just wrote
    pairSetupKernel<<<1, kMaxFrames * kMaxFrames, 0, w.sr.stream>>>(w.d_frames.ptr, w.d_state.ptr, w.d_pc.ptr, w.F(), 0, nullptr);
  }
  HIP_CHECK(hipGetLastError());
}

/** two-stage build of the combined system (large windows / deterministic mode): partial systems without atomics, then one
 *  ordered sum per entry.  `ctrl` (nullable) is the control block whose `active` gates both launches and whose lambda damps the
 *  system; the LM decision is NOT taken here (decideApplyKernel runs in front of / behind it). */
void launchTwoStage(W &w, const LmControl *ctrl, int ublk_parity, double lambda, bool dense = false, double *group_sums = nullptr) {
  const int F = w.F();
  hipStream_t st = w.sr.stream;
  using TwoStageKernel = void (*)(TwoStageArgs);
  static const TwoStageKernel kernels[6] = {nullptr, schurTwoStageKernel<2>, schurTwoStageKernel<2>, schurTwoStageKernel<3>, schurTwoStageKernel<4>,
                                            schurTwoStageKernel<5>};
  const int tpw = twoStageTilesPerWave(F);
  if (tpw < 1 || tpw > kMaxTilesPerWave) fail(DSOPP_HIP_ERR_CAPACITY, "window of %d frames exceeds the Schur kernel's tile budget", F);
  const TwoStageKernel two_stage_kernel = kernels[tpw];
  ensureDynamicLds(reinterpret_cast<const void *>(two_stage_kernel), w.sr.device, 96 * 1024);
  const int n_chunks = w.opt.optimize_idepths ? w.n_schur_blocks : 0;
  // two workgroups fit per compute unit (64 landmarks x K doubles of LDS each): twice as many workgroups as the chip has units,
  // each taking its share of the chunks
  static const int chunks_override = std::getenv("DSOPP_HIP_SCHUR_CHUNKS") ? std::atoi(std::getenv("DSOPP_HIP_SCHUR_CHUNKS")) : 0;  // tuning aid
  const int chunks_per_wg = chunks_override > 0 ? chunks_override : std::max(1, (n_chunks + 511) / 512);
  // chunks are dealt out evenly over at most 512 workgroups (n / W each, the first n mod W one more): ceil(n / 512) chunks for every
  // workgroup left e.g. 313 workgroups of 2 for 625 chunks — 12 KF / 40 000 points: 38.5 -> 31.7 us, 12 KF / 100 000: 62.9 -> 57.1 us;
  // no change where the old split was balanced (782 chunks).  DSOPP_HIP_SCHUR_EVEN=0 / DSOPP_HIP_SCHUR_CHUNKS=n: the old split (A/B aids)
  static const int even_env = std::getenv("DSOPP_HIP_SCHUR_EVEN") ? std::atoi(std::getenv("DSOPP_HIP_SCHUR_EVEN")) : 1;
  const bool even = even_env != 0 && chunks_override <= 0;
  const int n_wgs = even ? std::max(1, std::min(512, n_chunks)) : (n_chunks + chunks_per_wg - 1) / chunks_per_wg;
  w.d_schur_partials.reserve(std::max<size_t>(1, static_cast<size_t>(n_wgs)) * static_cast<size_t>(twoStagePartialCount(F)), 0, st);
  w.d_pair_out.reserve(static_cast<size_t>(kMaxFrames) * kMaxFrames * kPairOut, 0, st);
  TwoStageArgs a;
  a.frames = w.d_frames.ptr;
  a.pc = w.d_pc.ptr;
  a.schur_table = w.d_schur_table.ptr;
  a.partials = w.d_partials.ptr;
  a.dbg = w.dbg_stamps ? w.dbg_stamps + 32 : nullptr;
  a.pair_first_block = w.d_pair_first.ptr;
  a.pair_num_blocks = w.d_pair_count.ptr;
  a.ctrl = ctrl;
  a.schur_partials = w.d_schur_partials.ptr;
  a.pair_out = w.d_pair_out.ptr;
  a.F = F;
  a.n_chunks = n_chunks;
  a.chunks_per_wg = even ? 0 : chunks_per_wg;
  a.n_schur_wgs = n_wgs;
  a.ublk_parity = ublk_parity;
  a.group_sums = group_sums;
  a.n_sweep_blocks = w.n_sweep_blocks;
  CombineArgs c;
  c.pc = w.d_pc.ptr;
  c.schur_partials = w.d_schur_partials.ptr;
  c.pair_out = w.d_pair_out.ptr;
  c.ctrl = ctrl;
  c.lambda = lambda;
  c.comb = w.d_reduce.ptr;
  c.dense = dense ? w.d_reduce.ptr : nullptr;
  c.F = F;
  c.n_schur_wgs = n_wgs;
  const size_t pair_smem = (48 + 64 + kPairBlk + 8 * 48) * sizeof(double);
  timedLaunch(w, DSOPP_HIP_KERNEL_SCHUR, [&] {
    two_stage_kernel<<<n_wgs + F * F + (group_sums ? kScalarGroups : 0), kSchurThreads, std::max(twoStageSmemBytes(F), pair_smem), st>>>(a);
    combineSystemKernel<<<static_cast<unsigned>((twoStagePartialCount(F) + kCombineEntries - 1) / kCombineEntries), kCombineEntries * kCombineSlices, 0, st>>>(c);
  });
  HIP_CHECK(hipGetLastError());
}

/** K3 of the fused loop: priors + solve of the combined system launchReduceSchur(combined) left at the head of d_reduce */
/** how many workgroups of the solve launch a device holds at once (its landmark workgroups wait for workgroup 0: all of them resident) */
int solveResidentWorkgroups(W &w, bool wide) {
  static std::map<std::pair<int, size_t>, int> cache;
  static std::mutex mu;
  const size_t smem = solveSmemBytes(w.K());
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(w.sr.device * 2 + (wide ? 1 : 0), smem);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int per_cu = 0, cus = 0;
  if (wide)
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, solveCombinedKernel<512>, 512, smem));
  else
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, solveCombinedKernel<256>, 256, smem));
  HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, w.sr.device));
  const int n = std::max(2, per_cu * cus);
  cache[key] = n;
  return n;
}

void launchSolveCombined(W &w, double lambda, LmControl *ctrl, const LmControl *decide_from = nullptr, const LmParams *decide_prm = nullptr,
                         bool decide_from_groups = false, bool backsub = false, int ublk_parity = 0) {
  ensureDynamicLds(reinterpret_cast<const void *>(solveCombinedKernel<256>), w.sr.device, 150 * 1024);
  ensureDynamicLds(reinterpret_cast<const void *>(solveCombinedKernel<256, kMaxCombCopies>), w.sr.device, 150 * 1024);
  ensureDynamicLds(reinterpret_cast<const void *>(solveCombinedKernel<512>), w.sr.device, 150 * 1024);
  SolveCombArgs a;
  if (decide_from) {
    // decision + accept / reject as the prologue of this launch (workgroups 1 ..: the landmarks), then calculateIdepths for the new step
    a.dec_in = decide_from;
    a.dec_scalars = w.d_reduce.ptr + w.combCount();  // the four sums travelled behind the combined system in the collective
    a.dec_table = w.d_schur_table.ptr;
    a.dec_chunks = w.n_schur_blocks;
    // (landmark passes: 128 landmarks = two chunks at 512 threads, one chunk at 256 — pba_solve_combined.hpp)
    const bool wide = w.F() > 8;
    a.dec_blocks = std::min(wide ? (w.n_schur_blocks + 1) / 2 : w.n_schur_blocks, solveResidentWorkgroups(w, wide) - 1);
    a.dec_groups = decide_from_groups ? kScalarGroups : 0;
    a.dec_prm = *decide_prm;
    if (backsub && a.dec_blocks > 0) {
      if (!w.d_bs_flag) {
        // [ticket counter | pad] + two hand-over buffers of kBlk * kMaxFrames doubles, both armed
        constexpr size_t kHand = static_cast<size_t>(kBlk) * kMaxFrames;
        HIP_CHECK(hipMalloc(&w.d_bs_flag, 16 + 2 * kHand * sizeof(double)));
        std::vector<double> arm(2 * kHand, kHandOverSentinel());
        HIP_CHECK(hipMemsetAsync(w.d_bs_flag, 0, 16, w.sr.stream));
        HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char *>(w.d_bs_flag) + 16, arm.data(), arm.size() * sizeof(double), hipMemcpyHostToDevice, w.sr.stream));
        HIP_CHECK(hipStreamSynchronize(w.sr.stream));  // (`arm` is a pageable host buffer)
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&w.h_bs_fault), sizeof(int), hipHostMallocDefault));
        *w.h_bs_fault = 0;
      }
      {
        constexpr size_t kHand = static_cast<size_t>(kBlk) * kMaxFrames;
        double *hand = reinterpret_cast<double *>(reinterpret_cast<char *>(w.d_bs_flag) + 16);
        const unsigned n = w.bs_seq;
        a.bs_ticket = w.d_bs_flag;
        a.bs_hand = hand + (n & 1u) * kHand;
        a.bs_hand_next = hand + ((n + 1) & 1u) * kHand;
      }
      a.bs_parity = ublk_parity;
      a.bs_ticket_base = w.bs_ticket_base;  // (both mirrors advance behind the launch, once it is known to have been accepted)
      a.bs_fault = w.h_bs_fault;
    }
  }
  a.frames = w.d_frames.ptr;
  a.st = w.d_state.ptr;
  a.pc = w.d_pc.ptr;
  a.comb = w.d_reduce.ptr;
  if (w.comb_copies_active > 1) {
    a.comb_copies = w.comb_copies_active;
    a.comb_copy_first = static_cast<int>(w.combCopyFirst());
    a.comb_copy_stride = static_cast<int>(w.combCopyStride());
  }
  a.Hm = w.d_Hm.ptr;
  a.HmPacked = w.d_Hm_packed.ptr;
  a.bm = w.d_bm.ptr;
  a.step = w.d_step.ptr;
  a.ctrl = ctrl;
  a.lambda = lambda;
  a.affine_reg[0] = w.opt.affine_brightness_regularizer[0];
  a.affine_reg[1] = w.opt.affine_brightness_regularizer[1];
  a.fixed_reg = w.opt.fixed_state_regularizer;
  a.energy_marginalized = w.energy_marginalized;
  a.F = w.F();
  a.fej = w.fej() ? 1 : 0;
  a.use_marginal = w.marg_nonzero ? 1 : 0;
  a.dbg_stamps = w.dbg_stamps;
  timedLaunch(w, DSOPP_HIP_KERNEL_ASSEMBLE_SOLVE,
              [&] {
                if (w.F() > 8)
                  solveCombinedKernel<512><<<1 + a.dec_blocks, 512, solveSmemBytes(w.K()), w.sr.stream>>>(
                      a.bs_ticket, a.bs_hand_next, a.dec_in, a.dec_scalars, a.comb, a.frames, a.st, a.F, a);
                else if (a.comb_copies > 1)
                  solveCombinedKernel<256, kMaxCombCopies><<<1 + a.dec_blocks, 256, solveSmemBytes(w.K()), w.sr.stream>>>(
                      a.bs_ticket, a.bs_hand_next, a.dec_in, a.dec_scalars, a.comb, a.frames, a.st, a.F, a);
                else
                  solveCombinedKernel<256><<<1 + a.dec_blocks, 256, solveSmemBytes(w.K()), w.sr.stream>>>(
                      a.bs_ticket, a.bs_hand_next, a.dec_in, a.dec_scalars, a.comb, a.frames, a.st, a.F, a);
              });
  HIP_CHECK(hipGetLastError());
  if (a.bs_ticket) {
    // the launch was accepted: every one of its workgroups draws exactly one ticket and arms the other hand-over buffer.  (Advancing
    // these before the launch was known to exist left every later launch without a ticket 0 — all workgroups waiting for a solver
    // that is none of them — after a single failed launch.)
    w.bs_seq++;
    w.bs_ticket_base += static_cast<unsigned>(1 + a.dec_blocks);
  }
  if (!w.fej()) {
    // no first-estimate Jacobians: all pair constants follow the candidate state eps + step the solve just wrote
    pairSetupKernel<<<1, kMaxFrames * kMaxFrames, 0, w.sr.stream>>>(w.d_frames.ptr, w.d_state.ptr, w.d_pc.ptr, w.F(), 0, nullptr);
    HIP_CHECK(hipGetLastError());
  }
}

/** behind a synchronisation of a fused solve: did a landmark workgroup of a solve launch give up waiting for the step?  Then the inverse
 *  depths of that round were not back-substituted and the solve's result is not the algorithm's: the hand-over state is rebuilt and the
 *  call fails (the window stays usable). */
void checkSolveLaunchFault(W &w) {
  if (!w.h_bs_fault || !*w.h_bs_fault) return;
  *w.h_bs_fault = 0;
  constexpr size_t kHand = static_cast<size_t>(kBlk) * kMaxFrames;
  std::vector<double> arm(2 * kHand, kHandOverSentinel());
  HIP_CHECK(hipMemsetAsync(w.d_bs_flag, 0, 16, w.sr.stream));
  HIP_CHECK(hipMemcpyAsync(reinterpret_cast<char *>(w.d_bs_flag) + 16, arm.data(), arm.size() * sizeof(double), hipMemcpyHostToDevice, w.sr.stream));
  HIP_CHECK(hipStreamSynchronize(w.sr.stream));
  w.bs_seq = 0;
  w.bs_ticket_base = 0;
  fail(DSOPP_HIP_ERR_HIP, "a solve launch's landmark workgroups waited 2 s for the pose step (device shared with a kernel that never yields?): "
                          "the inverse depths of that iteration were not updated, the solve is void");
}

void launchBacksub(W &w, double lambda, const LmControl *ctrl, int ublk_parity = 0, bool gate_on_pending = false) {
  if (!w.opt.optimize_idepths || !w.n_schur_blocks) return;
  timedLaunch(w, DSOPP_HIP_KERNEL_BACKSUB, [&] {
    backsubKernel<<<w.n_schur_blocks, kSchurLandmarks, 0, w.sr.stream>>>(w.d_frames.ptr, w.d_schur_table.ptr, w.d_step.ptr, lambda, w.F(), ctrl,
                                                                         ublk_parity, gate_on_pending ? 1 : 0);
  });
  HIP_CHECK(hipGetLastError());
}

// ---- stages ---------------------------------------------------------------------------------------------------

/** prior / marginal terms of calculateEnergy — problem.hpp:293-312 — evaluated on the host mirror of the state */
double priorEnergy(const W &w, bool with_step) {
  const int F = w.F(), K = w.K();
  std::vector<double> state(static_cast<size_t>(K));
  for (int f = 0; f < F; ++f)
    for (int a = 0; a < kBlk; ++a) state[static_cast<size_t>(kBlk * f + a)] = w.hst.eps[f][a] + (with_step ? w.hst.step[f][a] : 0.0);
  double e = w.energy_marginalized;
  double quad = 0, lin = 0;
  for (int i = 0; i < K; ++i) {
    lin += w.bm[static_cast<size_t>(i)] * state[static_cast<size_t>(i)];
    double s = 0;
    for (int j = 0; j < K; ++j) s += w.Hm[static_cast<size_t>(i) * K + j] * state[static_cast<size_t>(j)];
    quad += state[static_cast<size_t>(i)] * s;
  }
  e += lin + quad / 2;
  for (int f = 0; f < F; ++f) {
    double t = 0;
    for (int a = 0; a < 2; ++a) {
      const double ab = w.hst.ab0[f][a] + state[static_cast<size_t>(kBlk * f + 6 + a)];
      t += ab * w.opt.affine_brightness_regularizer[a] * ab;
    }
    e += t / 2;
  }
  return e;
}

void stageBegin(W &w) {
  if (w.F() == 0) fail(DSOPP_HIP_ERR_STATE, "window is empty");
  prepare(w);
  if (w.fej()) firstEstimate(w);
  w.begun_with_first_estimate = true;
  w.begun = true;
  w.linearized = false;
}

/** begin of a fused solve (lm_mode 0): firstEstimateJacobians is folded into the solve's own first kernels (lmSolveFusedEnqueue)
 *  whenever there is a linearisation to fold it into */
void fusedBegin(W &w) {
  if (w.F() == 0) fail(DSOPP_HIP_ERR_STATE, "window is empty");
  if (!(w.fej() && w.opt.max_iterations > 0)) {
    stageBegin(w);
    return;
  }
  prepare(w);
  w.begun_with_first_estimate = false;
  w.begun = true;
  w.linearized = false;
}

std::pair<double, int> stageEnergy(W &w) {
  if (!w.begun) fail(DSOPP_HIP_ERR_STATE, "call begin first");
  launchSweep(w, false, true, false);
  timedLaunch(w, DSOPP_HIP_KERNEL_ENERGY_REDUCE,
              [&] { energyReduceKernel<<<1, 256, 0, w.sr.stream>>>(w.d_partials.ptr, w.n_sweep_blocks, w.d_scalars.ptr); });
  HIP_CHECK(hipGetLastError());
  allreduceIfNeeded(w, w.d_scalars.ptr, 2);
  double out[2];
  w.d_scalars.download(out, 2, 0, w.sr.stream);
  w.sr.sync();
  return {out[0] + priorEnergy(w, true), static_cast<int>(out[1] + 0.5)};
}

void stageLinearize(W &w, bool huber = true, bool for_marg = false, bool add_priors = true) {
  if (!w.begun) fail(DSOPP_HIP_ERR_STATE, "call begin first");
  launchSweep(w, true, huber, for_marg);
  launchReduceSchur(w, for_marg, nullptr);
  launchAssemble(w, 0.0, false, add_priors, /*store_system=*/true, nullptr);
  w.linearized = true;
}

void stageStep(W &w, double lambda) {
  if (!w.linearized) fail(DSOPP_HIP_ERR_STATE, "call linearize first");
  // the per-pair blocks and the Schur system are still resident: re-run the (cheap) assembly with the requested lambda
  launchAssemble(w, lambda, true, true, false, nullptr);
  w.pair_valid = true;  // the solve kernel rebuilt the pair constants for eps + step
  launchBacksub(w, lambda, nullptr);
  const int K = w.K();
  w.last_step.resize(static_cast<size_t>(K));
  w.d_step.download(w.last_step.data(), static_cast<size_t>(K), 0, w.sr.stream);
  w.sr.sync();
  for (int f = 0; f < w.F(); ++f)
    for (int a = 0; a < kBlk; ++a) w.hst.step[f][a] = -w.last_step[static_cast<size_t>(kBlk * f + a)];
  w.last_lambda = lambda;
}

std::pair<double, double> stageAccept(W &w, bool accept) {
  HIP_CHECK(hipMemsetAsync(w.d_scalars.ptr + 4, 0, 4 * sizeof(double), w.sr.stream));
  timedLaunch(w, DSOPP_HIP_KERNEL_ACCEPT, [&] {
    if (w.n_schur_blocks) {
      acceptLandmarksKernel<<<w.n_schur_blocks, kSchurThreads, 0, w.sr.stream>>>(w.d_frames.ptr, w.d_schur_table.ptr, w.F(), accept ? 1 : 0,
                                                                                  w.d_scalars.ptr + 4);
    }
    acceptFramesKernel<<<1, 128, 0, w.sr.stream>>>(w.d_state.ptr, w.F(), accept ? 1 : 0, w.d_scalars.ptr + 6);
  });
  HIP_CHECK(hipGetLastError());
  // the idepth part of the norms is a sum over this rank's landmark shard, the frame part is replicated
  if (accept) allreduceIfNeeded(w, w.d_scalars.ptr + 4, 2);
  double n4[4];
  w.d_scalars.download(n4, 4, 4, w.sr.stream);
  w.sr.sync();
  const double norms[2] = {n4[0] + n4[2], n4[1] + n4[3]};
  // host mirror
  for (int f = 0; f < w.F(); ++f)
    for (int a = 0; a < kBlk; ++a) {
      if (accept) w.hst.eps[f][a] += w.hst.step[f][a];
      w.hst.step[f][a] = 0;
    }
  if (!accept) w.pair_valid = false;  // constants were built for eps + step
  return {norms[0], norms[1]};
}

/**
 * levenberg_marquardt_algorithm::solve (levenberg_marquardt_algorithm.hpp:77-128) with the control flow ON THE DEVICE:
 * the host enqueues max_iterations loop bodies (5 launches each) plus the closing energy evaluation and reads the
 * control block back once; kernels of loop bodies after termination return immediately.
 */
void lmSolveDevice(W &w, double &energy_out, int &iterations, int &n_valid_out) {
  hipStream_t st = w.sr.stream;
  const int F = w.F();
  LmParams prm;
  prm.function_tolerance = w.opt.function_tolerance;
  prm.parameter_tolerance = w.opt.parameter_tolerance;
  prm.decrease_on_accept = 1.0;  // eigen_photometric_bundle_adjustment.cpp:74-75
  prm.increase_on_reject = 1.0;
  prm.lambda0 = 1.0 / w.opt.initial_trust_region_radius;
  prm.max_iterations = w.opt.max_iterations;
  prm.min_iterations = 3;
  prm.force_accept = w.opt.force_accept;
  prm.use_reduced_scalars = w.allreduce ? 1 : 0;
  LmControl *ctrl = w.d_ctrl.ptr;
  // result = problem.calculateEnergy()
  launchSweep(w, false, true, false);
  HIP_CHECK(hipMemsetAsync(w.d_scalars.ptr, 0, 8 * sizeof(double), st));
  if (w.n_schur_blocks) idepthNormKernel<<<w.n_schur_blocks, kSchurThreads, 0, st>>>(w.d_frames.ptr, w.d_schur_table.ptr, w.d_scalars.ptr + 4);
  if (w.allreduce) {
    sweepScalarsKernel<<<1, 256, 0, st>>>(w.d_partials.ptr, w.n_sweep_blocks, w.d_scalars.ptr, nullptr);
    allreduceIfNeeded(w, w.d_scalars.ptr, 5);
  }
  {
    LmInitArgs ia;
    ia.sa = makeSolveArgs(w);
    ia.partials = w.d_partials.ptr;
    ia.scalars = w.d_scalars.ptr;
    ia.schur_table = w.d_schur_table.ptr;
    ia.n_sweep_blocks = w.n_sweep_blocks;
    ia.n_schur_blocks = w.n_schur_blocks;
    ia.ctrl = ctrl;
    ia.prm = prm;
    lmInitKernel<<<1, kSolveThreads, (static_cast<size_t>(w.K()) + 16) * sizeof(double), st>>>(ia);
  }
  HIP_CHECK(hipGetLastError());
  for (int it = 0; it < w.opt.max_iterations; ++it) {
    LmControl *cin = ctrl + (it & 1), *cout = ctrl + ((it + 1) & 1);
    launchSweep(w, true, true, false, cin);          // linearize: evaluateJacobians ...
    launchReduceSchur(w, false, cin);                //            ... pose-pose blocks + Schur complement
    launchAssemble(w, 0.0, true, true, false, cin);  // calculateStep (lambda from the control block)
    launchSweep(w, false, true, false, cin, /*backsub=*/true);  // calculateIdepths + calculateEnergy at the candidate state
    if (w.allreduce) {
      sweepScalarsKernel<<<1, 256, 0, st>>>(w.d_partials.ptr, w.n_sweep_blocks, w.d_scalars.ptr, cin);
      allreduceIfNeeded(w, w.d_scalars.ptr, 4);
    }
    LmDecideArgs da;
    da.frames = w.d_frames.ptr;
    da.st = w.d_state.ptr;
    da.schur_table = w.d_schur_table.ptr;
    da.partials = w.d_partials.ptr;
    da.scalars = w.d_scalars.ptr;
    da.ctrl_in = cin;
    da.ctrl_out = cout;
    da.n_sweep_blocks = w.n_sweep_blocks;
    da.n_schur_blocks = w.n_schur_blocks;
    da.F = F;
    da.prm = prm;
    timedLaunch(w, DSOPP_HIP_KERNEL_ACCEPT,
                [&] { lmDecideKernel<<<std::max(1, w.n_schur_blocks), kSchurThreads, 0, st>>>(da); });
    HIP_CHECK(hipGetLastError());
  }
  // closing problem.calculateEnergy() at the final state (candidate statuses / energies of the accepted state); the pair
  // constants are rebuilt unconditionally: they are stale exactly when the last step was rejected
  const LmControl *cfin = ctrl + (w.opt.max_iterations & 1);
  w.pair_valid = false;
  ensurePairConstants(w);
  launchSweep(w, false, true, false);
  // one small read-back into pinned memory (a pageable destination makes the copy synchronous and staged); the host
  // mirror of the frame states is refreshed lazily, by the first reader (downloadState)
  if (!w.h_ctrl) HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&w.h_ctrl), sizeof(LmControl), hipHostMallocDefault));
  HIP_CHECK(hipMemcpyAsync(w.h_ctrl, cfin, sizeof(LmControl), hipMemcpyDeviceToHost, st));
  w.host_stale = true;
  w.sr.sync();  // the only host synchronisation of the solve
  energy_out = w.h_ctrl->energy;
  iterations = w.h_ctrl->iteration;
  n_valid_out = w.h_ctrl->n_valid;
}


/** dsopp_hip_window_restore, host half: checks that the snapshot matches the window and puts the host-side mirror back */
void restoreHostSide(W &w) {
  if (!w.snap_valid || w.snap_F != w.F()) fail(DSOPP_HIP_ERR_STATE, "no snapshot matching the current window");
  w.sr.use();
  flushAppends(w);
  for (auto &fp : w.frames) {
    if (fp->snap_n != fp->n) fail(DSOPP_HIP_ERR_STATE, "frame %d changed since the snapshot", fp->id);
    for (auto &kv : fp->residuals)
      if (kv.second->snap_n != kv.second->n) fail(DSOPP_HIP_ERR_STATE, "connection of frame %d changed since the snapshot", fp->id);
  }
  syncTopology(w);
  flushAppends(w);  // (its tables are queued uploads)
  w.hst = w.snap_state;
  w.state_dirty = false;
  w.host_stale = false;
  w.pair_valid = false;
  w.begun = false;
}

/** ... device half (optimize_repeated lets the opening kernel of the next solve do this instead: lmBeginKernel) */
void launchRestore(W &w) {
  restoreKernel<<<w.n_schur_blocks + 1, kSchurLandmarks, 0, w.sr.stream>>>(w.d_frames.ptr, w.d_schur_table.ptr, w.F(), w.d_state.ptr, w.d_state_snap.ptr,
                                                                        w.n_schur_blocks);
  HIP_CHECK(hipGetLastError());
}

/**
 * Same algorithm, three launches per Gauss-Newton iteration.  A linearisation sweep at the candidate state x + step
 * already contains the candidate's energy (NEW_EVALUATION_POINT) and, if the step is accepted, IS the next linearisation:
 *   round r:  K1  linearise at eps + step_r (step_0 = 0)
 *             K2  pose-pose + Schur reduction into the combined system (lambda of the incoming control block: constant),
 *                 + one workgroup (64 on the two-stage path) that adds up K1's energy scalars
 *             K3  prologue: decide step_r from those sums, apply accept / reject (the landmark workgroups);
 *                 the solving workgroup: solve -> step_{r+1}; the landmark workgroups: calculateIdepths for step_{r+1} as soon as it is
 *                 published, under the solver's tail (round 4; until then a kernel of its own in front of K1 on large windows, fused into
 *                 K1 on small ones: DSOPP_HIP_K3_BACKSUB=0)
 * (large windows: K2 = partial systems + ordered sum, two launches)
 * max_iterations + 1 rounds, one host synchronisation.  A rejected step without force_accept costs one extra round (the
 * sweep re-linearises at the reverted state, which reproduces the system the reference keeps via linear_system_valid).
 * The Schur rows are double-buffered because K1 reads the previous round's rows while writing this round's.
 */
void lmSolveFusedEnqueue(W &w) {
  hipStream_t st = w.sr.stream;
  LmParams prm;
  prm.function_tolerance = w.opt.function_tolerance;
  prm.parameter_tolerance = w.opt.parameter_tolerance;
  prm.decrease_on_accept = 1.0;  // eigen_photometric_bundle_adjustment.cpp:74-75
  prm.increase_on_reject = 1.0;
  prm.lambda0 = 1.0 / w.opt.initial_trust_region_radius;
  prm.max_iterations = w.opt.max_iterations;
  prm.min_iterations = 3;
  prm.force_accept = w.opt.force_accept;
  prm.use_reduced_scalars = w.allreduce ? 1 : 0;
  LmControl *ctrl = w.d_ctrl.ptr;
  // With first-estimate Jacobians the opening linearisation takes the snapshot (idepth, reprojection validity) for its own items
  // and the pair constants are set up by lmBeginKernel: a solve starts with ONE small kernel instead of three (pair set-up,
  // first-estimate kernel, LM initialisation).  fusedBegin() skipped firstEstimate() when this holds.
  const bool sweep_takes_fej = w.fej() && w.opt.max_iterations > 0 && !w.begun_with_first_estimate;
  const bool begin_sets_pairs = !w.pair_valid;
  {
    LmInitArgs ia;
    ia.pair_frames = begin_sets_pairs ? w.d_frames.ptr : nullptr;
    ia.pair_pc = w.d_pc.ptr;
    ia.pair_fej = w.fej() ? 1 : 0;
    w.pair_valid = true;
    ia.sa = makeSolveArgs(w);
    ia.partials = w.d_partials.ptr;
    ia.scalars = nullptr;
    ia.schur_table = w.d_schur_table.ptr;
    ia.n_sweep_blocks = w.n_sweep_blocks;
    ia.n_schur_blocks = w.n_schur_blocks;
    ia.ctrl = ctrl;
    ia.prm = prm;
    int begin_blocks = 1;
    if (w.restore_in_begin) {
      w.restore_in_begin = false;
      if (begin_sets_pairs) {
        ia.restore_state = w.d_state.ptr;
        ia.sa.st = w.d_state_snap.ptr;
        begin_blocks += w.n_schur_blocks + 1;
      } else {
        launchRestore(w);
      }
    }
    lmBeginKernel<<<begin_blocks, kSolveThreads, (static_cast<size_t>(w.K()) + 16) * sizeof(double), st>>>(ia);
  }
  HIP_CHECK(hipGetLastError());
  // one round per iteration + the opening evaluation; without force_accept a rejected step costs one more round (the
  // re-linearisation at the reverted state), so the budget doubles — rounds after the loop has ended are no-op launches
  const int rounds = (w.opt.force_accept ? w.opt.max_iterations : 2 * w.opt.max_iterations) + 1;
  bool result_written_by_kernel = false;
  // calculateIdepths inside the solve launch (its landmark workgroups wait for the step under the factorisation): no back-substitution
  // kernel in front of the sweeps of large windows, no Schur-row reads in the sweeps of small ones.  DSOPP_HIP_K3_BACKSUB=0: the round-3 flow
  static const int k3_env = std::getenv("DSOPP_HIP_K3_BACKSUB") ? std::atoi(std::getenv("DSOPP_HIP_K3_BACKSUB")) : 1;
  const bool k3_backsub = k3_env != 0 && w.opt.optimize_idepths && w.n_schur_blocks > 0;
  // atomics path of an unsharded window of up to 8 keyframes: the reduction launch accumulates into several copies of the combined system
  // and the solve launch adds them while loading (the queue of same-address f64 atomics is that launch's tail).  DSOPP_HIP_COMB_COPIES=1: off
  static const int comb_copies_env = std::getenv("DSOPP_HIP_COMB_COPIES") ? std::atoi(std::getenv("DSOPP_HIP_COMB_COPIES")) : 4;
  static const int comb_copies_min_chunks = std::getenv("DSOPP_HIP_COMB_COPIES_MIN_CHUNKS") ? std::atoi(std::getenv("DSOPP_HIP_COMB_COPIES_MIN_CHUNKS")) : 80;
  w.comb_copies_active = (!w.twoStage() && !w.allreduce && w.F() <= 7 && w.n_schur_blocks >= comb_copies_min_chunks)
                             ? std::max(1, std::min(comb_copies_env, kMaxCombCopies))
                             : 1;
  struct CopiesReset {
    W &w;
    ~CopiesReset() { w.comb_copies_active = 1; }
  } copies_reset{w};
  for (int r = 0; r < rounds; ++r) {
    LmControl *cin = ctrl + (r & 1), *cout = ctrl + ((r + 1) & 1);
    SweepExtras ex;
    ex.ublk_read = (r + 1) & 1;
    ex.ublk_write = r & 1;
    ex.gate_on_pending = true;
    ex.fused_lin_backsub = true;
    ex.combined = true;
    ex.write_fej = r == 0 && sweep_takes_fej;
    // the closing round only has to evaluate the last candidate (no linear system is built from it): residual-only sweep
    const bool large = w.n_schur_blocks > w.backsubSplitMinChunks();
    if (k3_backsub) {
      // calculateIdepths for this round's candidate ran in the solve launch that produced its step (launchSolveCombined): the sweep
      // only reads the inverse-depth steps
      ex.fused_lin_backsub = false;
      ex.external_backsub = true;
      launchSweep(w, /*lin=*/r + 1 < rounds, true, false, cin, false, 0.0, ex);
    } else if (large && r + 1 == rounds) {
      // closing round of a large window: the same split (the residual-only sweep with the back-substitution fused in took 115 us
      // at 12 frames / 50 000 landmarks against 10 + 47 us as two kernels)
      launchBacksub(w, 0.0, cin, ex.ublk_read, /*gate_on_pending=*/true);
      ex.fused_lin_backsub = false;
      ex.external_backsub = true;
      launchSweep(w, false, true, false, cin, false, 0.0, ex);
    } else if (large) {
      // large windows: the back-substitution fused into the sweep re-reads a landmark's whole Schur row for every one of its
      // (landmark, target) items — (F - 1) x the traffic (12 frames / 50 000 landmarks: 196 against 131 us).  One kernel per
      // landmark in front of the sweep instead.
      if (r > 0) launchBacksub(w, 0.0, cin, ex.ublk_read, /*gate_on_pending=*/true);  // (no step is pending in the opening round)
      ex.fused_lin_backsub = false;
      ex.external_backsub = true;
      launchSweep(w, true, true, false, cin, false, 0.0, ex);
    } else {
      launchSweep(w, /*lin=*/r + 1 < rounds, true, false, cin, true, 0.0, ex);
    }
    FusedReduce fr;
    fr.ublk_parity = r & 1;
    fr.ctrl_out = cout;
    fr.prm = prm;
    fr.combined = true;  // (the sharded accumulate pass reads lambda from the incoming control block: constant, decrease = increase = 1)
    bool decide_from_groups = false;
    if (w.twoStage() && r + 1 < rounds) {
      // large windows / deterministic mode: the combined system without atomics (pba_schur_two_stage.hpp).  kScalarGroups extra
      // workgroups of the same launch sum the sweep's energy scalars in fixed groups behind the combined system's slot; the
      // decision — a function of those sums and the control block alone — and its accept / reject are the prologue of the solve
      // launch (lambda is constant: decrease = increase = 1, so the ordered sum can damp with the incoming control block's).
      // Landmark shards send the group sums along in their ONE collective.
      double *groups = w.d_reduce.ptr + w.combCount();
      launchTwoStage(w, cin, fr.ublk_parity, 0.0, false, groups);
      allreduceIfNeeded(w, w.d_reduce.ptr, w.combCount() + 4 * kScalarGroups);
      decide_from_groups = true;
    } else if (w.allreduce && r + 1 == rounds) {
      // landmark shards, closing round: its sweep was residual-only, no system exists — only the four energy scalars of the last
      // candidate are summed across the shards (they sit where the decision expects them: behind the combined system's slot)
      if (large) {
        sweepScalarGroupsKernel<<<kScalarGroups, 256, 0, st>>>(w.d_partials.ptr, w.n_sweep_blocks, w.d_scalars.ptr + 16, cin);
        sweepScalarGroupsFinalKernel<<<1, 64, 0, st>>>(w.d_scalars.ptr + 16, w.d_reduce.ptr + w.combCount());
      } else {
        sweepScalarsKernel<<<1, 256, 0, st>>>(w.d_partials.ptr, w.n_sweep_blocks, w.d_reduce.ptr + w.combCount(), cin);
      }
      HIP_CHECK(hipGetLastError());
      allreduceIfNeeded(w, w.d_reduce.ptr + w.combCount(), 4);
      launchReduceSchur(w, false, cin, &fr, ReduceMode::kDecideOnly);
    } else if (r + 1 == rounds) {
      // the closing round only takes the decision for the last candidate (its sweep was residual-only: no system to build) and
      // leaves the solve's result in pinned host memory itself (a copy kernel behind it cost 4 us per solve)
      if (!w.h_ctrl) HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&w.h_ctrl), sizeof(LmControl), hipHostMallocDefault));
      fr.ctrl_host = w.result_device ? w.result_device : w.h_ctrl;  // (batched solves: a device slot, fetched with the batch)
      result_written_by_kernel = true;
      if (large) {
        // tens of thousands of sweep blocks: their scalars in 64 fixed groups first (every workgroup of the decision adds 64 x 4
        // numbers instead of walking all blocks: 31.5 -> about 10 us at 12 frames / 50 000 landmarks)
        sweepScalarGroupsKernel<<<kScalarGroups, 256, 0, st>>>(w.d_partials.ptr, w.n_sweep_blocks, w.d_scalars.ptr + 16, cin);
        HIP_CHECK(hipGetLastError());
        fr.prm.use_reduced_scalars = 2;
        fr.scalars = w.d_scalars.ptr + 16;
      }
      launchReduceSchur(w, false, cin, &fr, ReduceMode::kDecideOnly);
    } else {
      // K2: the local systems accumulated with atomics + one extra workgroup that sums the sweep's four energy scalars behind them
      // (landmark shards: ONE collective over [system | scalars]); K3 then takes the decision — a function of those sums and the
      // control block alone — as its prologue.  (Until round 3 the unsharded K2 decided in its own prologue: every workgroup waited
      // for a round trip over the sweep's scalars before it started to build; 41.4 -> 40.1 us per iteration at C1.)
      launchReduceSchur(w, false, cin, &fr, ReduceMode::kAccumulateOnly);
      decide_from_groups = w.allreduce != nullptr;  // (shards: the grouped tail, see launchReduceSchur)
    }
    if (r + 1 < rounds) launchSolveCombined(w, 0.0, cout, cin, &fr.prm, decide_from_groups, k3_backsub, fr.ublk_parity);  // K3: decision prologue + solve (+ calculateIdepths)
  }
  LmControl *cfin = ctrl + (rounds & 1);
  HIP_CHECK(hipGetLastError());
  w.pair_valid = false;
  // one small read-back into pinned memory (a pageable destination makes the copy synchronous and staged); the host
  // mirror of the frame states is refreshed lazily, by the first reader (downloadState)
  if (!w.h_ctrl) HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&w.h_ctrl), sizeof(LmControl), hipHostMallocDefault));
  if (!result_written_by_kernel) {
    if (w.result_device)  // batched solves of a sharded window: the slot is filled by a copy behind the closing round
      HIP_CHECK(hipMemcpyAsync(w.result_device, cfin, sizeof(LmControl), hipMemcpyDeviceToDevice, st));
    else
      HIP_CHECK(hipMemcpyAsync(w.h_ctrl, cfin, sizeof(LmControl), hipMemcpyDeviceToHost, st));
  }
  w.host_stale = true;
  w.fused_final_ctrl = cfin;
}

/** second half of the fused solve: the one host synchronisation, and the rare closing evaluation after a rejected last step */
void lmSolveFusedFinish(W &w, double &energy_out, int &iterations, int &n_valid_out) {
  hipStream_t st = w.sr.stream;
  const LmControl *cfin = w.fused_final_ctrl;
  w.sr.sync();  // the only host synchronisation of the solve (unless the last step was rejected, below)
  checkSolveLaunchFault(w);
  if (w.h_ctrl->need_final_setup) {
    // closing problem.calculateEnergy() at the final state: the last sweep already evaluated it unless the last step was
    // rejected — then the pair constants are rebuilt and a residual sweep re-evaluates energies / candidate statuses
    ensurePairConstants(w);
    launchSweep(w, false, true, false);
    HIP_CHECK(hipGetLastError());
    w.sr.sync();
  }
  (void)st;
  (void)cfin;
  energy_out = w.h_ctrl->energy;
  iterations = w.h_ctrl->iteration;
  n_valid_out = w.h_ctrl->n_valid;
}

void lmSolveFused(W &w, double &energy_out, int &iterations, int &n_valid_out) {
  lmSolveFusedEnqueue(w);
  lmSolveFusedFinish(w, energy_out, iterations, n_valid_out);
}

}  // namespace
}  // namespace dsopp_hip

// ---------------------------------------------------------------------------------------------------------------------
// point statuses, relinearisation, covariance, marginalisation, solve
// ---------------------------------------------------------------------------------------------------------------------
namespace dsopp_hip {
namespace {

Rigid poseOf(const W &w, int f) {
  Rigid T0;
  for (int i = 0; i < 9; ++i) T0.R[i] = w.hst.T0_R[f][i];
  for (int i = 0; i < 3; ++i) T0.t[i] = w.hst.T0_t[f][i];
  return rigidMul(T0, rigidExp(w.hst.eps[f]));  // tWorldAgent — local_frame.hpp:525-527
}

/** relinearizeSystem — PROB_SRC/photometric_bundle_adjustment.cpp:310-316 (on the device state; the host mirror follows lazily) */
void relinearize(W &w) {
  prepareDevice(w);
  relinearizeKernel<<<1, kMaxFrames * kMaxFrames, 0, w.sr.stream>>>(w.d_frames.ptr, w.d_state.ptr, w.d_pc.ptr, w.F(), w.fej() ? 1 : 0, w.F() - 1);
  HIP_CHECK(hipGetLastError());
  w.host_stale = true;
  w.pair_valid = true;  // (set up by the same launch)
}

/** updatePointStatuses on the device: exact 3rd-quartile order statistic by an 8-pass radix select over the energies where
 *  they lie, then one pass per landmark.  No host round trip (the sharded multi-rank case keeps the host gather below). */
void updatePointStatusesDevice(W &w) {
  prepareDevice(w);
  hipStream_t st = w.sr.stream;
  const int F = w.F();
  if (!w.d_select.ptr) {
    w.d_select.reserve(1, 0, st);
    selectInitKernel<<<1, 1024, 0, st>>>(w.d_select.ptr);  // once: every select leaves the state ready for the next (selectFinish)
  }
  w.d_pair_dist.reserve(static_cast<size_t>(kMaxFrames) * kMaxFrames, 0, st);
  const double half_sigma_sq = w.opt.sigma_huber_loss * w.opt.sigma_huber_loss / 2;
  if (w.n_sweep_blocks) {
    const int per_group = kSelectThreads / kItemsPerBlock;
    const int grid = (w.n_fine_blocks + per_group - 1) / per_group;
    for (int pass = kSelectPasses - 1; pass >= 0; --pass)  // each pass advances the select state from the previous pass's histogram itself
      selectHistKernel<<<grid, kSelectThreads, 0, st>>>(w.d_frames.ptr, w.fineTable(), w.n_fine_blocks, w.d_select.ptr, pass);
  }
  pairDistanceKernel<<<1, 256, 0, st>>>(w.d_state.ptr, F, w.d_pair_dist.ptr, w.n_sweep_blocks ? w.d_select.ptr : nullptr, half_sigma_sq);
  if (w.n_schur_blocks)
    applyPointStatusesKernel<<<w.n_schur_blocks, kSchurLandmarks, 0, st>>>(w.d_frames.ptr, w.d_schur_table.ptr, F, w.d_select.ptr, w.d_pair_dist.ptr);
  HIP_CHECK(hipGetLastError());
}

void updatePointStatuses(W &w) {
  if (!(w.allreduce && w.world > 1)) {
    updatePointStatusesDevice(w);
    return;
  }
  prepare(w);
  downloadState(w);
  const int F = w.F();
  hipStream_t st = w.sr.stream;
  struct Conn {
    int r, t;
    std::vector<uint8_t> status, cand;
    std::vector<double> energy;
  };
  std::vector<Conn> conns;
  std::vector<std::vector<uint8_t>> flags(static_cast<size_t>(F));
  std::vector<std::vector<double>> idepth(static_cast<size_t>(F)), baseline(static_cast<size_t>(F));
  for (int r = 0; r < F; ++r) {
    HostFrame &f = *w.frames[static_cast<size_t>(r)];
    flags[static_cast<size_t>(r)].resize(static_cast<size_t>(f.n));
    idepth[static_cast<size_t>(r)].resize(static_cast<size_t>(f.n));
    baseline[static_cast<size_t>(r)].resize(static_cast<size_t>(f.n));
    f.dflags.download(flags[static_cast<size_t>(r)].data(), static_cast<size_t>(f.n), 0, st);
    f.idepth.download(idepth[static_cast<size_t>(r)].data(), static_cast<size_t>(f.n), 0, st);
    f.relative_baseline.download(baseline[static_cast<size_t>(r)].data(), static_cast<size_t>(f.n), 0, st);
    for (int t = 0; t < F; ++t) {
      if (t == r) continue;
      HostFrame &g = *w.frames[static_cast<size_t>(t)];
      if (g.is_marginalized) continue;
      auto it = f.residuals.find(g.id);
      if (it == f.residuals.end() || it->second->n == 0) continue;
      Conn c;
      c.r = r;
      c.t = t;
      c.status.resize(static_cast<size_t>(it->second->n));
      c.cand.resize(static_cast<size_t>(it->second->n));
      c.energy.resize(static_cast<size_t>(it->second->n));
      it->second->status.download(c.status.data(), c.status.size(), 0, st);
      it->second->cand.download(c.cand.data(), c.cand.size(), 0, st);
      it->second->energy.download(c.energy.data(), c.energy.size(), 0, st);
      conns.push_back(std::move(c));
    }
  }
  w.sr.sync();
  std::vector<double> energies;
  for (const Conn &c : conns)
    for (size_t i = 0; i < c.status.size(); ++i) {
      if (flags[static_cast<size_t>(c.r)][i] & kFlagMarginalized) continue;
      if (c.status[i] == DSOPP_HIP_STATUS_OK) energies.push_back(c.energy[i]);
    }
  if (w.allreduce && w.world > 1) {
    // the 3rd-quartile threshold is a statistic of ALL kOk residual energies of the window, landmarks are sharded:
    // gather the per-rank energy lists with two sum-collectives (counts, then zero-padded slices)
    std::vector<double> counts(static_cast<size_t>(w.world), 0.0);
    counts[static_cast<size_t>(w.rank)] = static_cast<double>(energies.size());
    w.d_gather.reserve(static_cast<size_t>(w.world), 0, st);
    w.d_gather.upload(counts.data(), counts.size(), 0, st);
    allreduceIfNeeded(w, w.d_gather.ptr, counts.size());
    w.d_gather.download(counts.data(), counts.size(), 0, st);
    w.sr.sync();
    size_t total = 0, offset = 0;
    for (int r = 0; r < w.world; ++r) {
      if (r < w.rank) offset += static_cast<size_t>(counts[static_cast<size_t>(r)] + 0.5);
      total += static_cast<size_t>(counts[static_cast<size_t>(r)] + 0.5);
    }
    std::vector<double> all(total, 0.0);
    std::copy(energies.begin(), energies.end(), all.begin() + static_cast<long>(offset));
    if (total) {
      w.d_gather.reserve(total, 0, st);
      w.d_gather.upload(all.data(), total, 0, st);
      allreduceIfNeeded(w, w.d_gather.ptr, total);
      w.d_gather.download(all.data(), total, 0, st);
      w.sr.sync();
    }
    energies.swap(all);
  }
  double threshold = 0;
  if (!energies.empty()) {
    const size_t q = static_cast<size_t>(static_cast<double>(energies.size()) * 0.75);
    std::nth_element(energies.begin(), energies.begin() + static_cast<long>(q), energies.end());
    threshold = energies[q] + w.opt.sigma_huber_loss * w.opt.sigma_huber_loss / 2;
  }
  std::vector<std::vector<int32_t>> inliers(static_cast<size_t>(F));
  std::vector<std::vector<int>> valid(static_cast<size_t>(F));
  for (int r = 0; r < F; ++r) {
    inliers[static_cast<size_t>(r)].assign(static_cast<size_t>(w.frames[static_cast<size_t>(r)]->n), 0);
    valid[static_cast<size_t>(r)].assign(static_cast<size_t>(w.frames[static_cast<size_t>(r)]->n), 0);
  }
  for (Conn &c : conns) {
    const Rigid Tr = poseOf(w, c.r), Tt = poseOf(w, c.t);
    const double dx = Tr.t[0] - Tt.t[0], dy = Tr.t[1] - Tt.t[1], dz = Tr.t[2] - Tt.t[2];
    const double distance = std::sqrt(dx * dx + dy * dy + dz * dz);
    for (size_t i = 0; i < c.status.size(); ++i) {
      if (flags[static_cast<size_t>(c.r)][i] & kFlagMarginalized) continue;
      if (c.energy[i] > threshold) {  // residual = {kOutlier}: fresh ResidualPoint, energy 0
        c.status[i] = DSOPP_HIP_STATUS_OUTLIER;
        c.cand[i] = DSOPP_HIP_STATUS_OUTLIER;
        c.energy[i] = 0;
      }
      if (c.status[i] == DSOPP_HIP_STATUS_OK) {
        double &bl = baseline[static_cast<size_t>(c.r)][i];
        bl = std::max(bl, idepth[static_cast<size_t>(c.r)][i] * distance);
        valid[static_cast<size_t>(c.r)][i]++;
        inliers[static_cast<size_t>(c.r)][i]++;
      }
    }
    ResidualTable &rt = *w.frames[static_cast<size_t>(c.r)]->residuals[w.frames[static_cast<size_t>(c.t)]->id];
    rt.status.upload(c.status.data(), c.status.size(), 0, st);
    rt.cand.upload(c.cand.data(), c.cand.size(), 0, st);
    rt.energy.upload(c.energy.data(), c.energy.size(), 0, st);
  }
  for (int r = 0; r < F; ++r) {
    HostFrame &f = *w.frames[static_cast<size_t>(r)];
    bool any_conn = false;
    for (const Conn &c : conns) any_conn = any_conn || c.r == r;
    for (int i = 0; i < f.n; ++i) {
      uint8_t &fl = flags[static_cast<size_t>(r)][static_cast<size_t>(i)];
      if (fl & kFlagMarginalized) continue;
      if (valid[static_cast<size_t>(r)][static_cast<size_t>(i)] < 1) fl |= kFlagOutlier;  // minimum_valid_reprojections_num = 1
    }
    (void)any_conn;
    f.dflags.upload(flags[static_cast<size_t>(r)].data(), static_cast<size_t>(f.n), 0, st);
    f.relative_baseline.upload(baseline[static_cast<size_t>(r)].data(), static_cast<size_t>(f.n), 0, st);
    // number_of_inlier_residuals is only reset for non-marginalised landmarks; marginalised ones keep their value
    std::vector<int32_t> cur(static_cast<size_t>(f.n));
    f.n_inliers.download(cur.data(), cur.size(), 0, st);
    w.sr.sync();
    for (int i = 0; i < f.n; ++i)
      if (!(flags[static_cast<size_t>(r)][static_cast<size_t>(i)] & kFlagMarginalized)) cur[static_cast<size_t>(i)] = inliers[static_cast<size_t>(r)][static_cast<size_t>(i)];
    f.n_inliers.upload(cur.data(), cur.size(), 0, st);
    w.sr.sync();
    // (the host mirror is indexed as the caller lists the landmarks, the device arrays in the device's order)
    f.flags.resize(static_cast<size_t>(f.n));
    for (int c = 0; c < f.n; ++c) f.flags[static_cast<size_t>(c)] = flags[static_cast<size_t>(r)][static_cast<size_t>(f.internalOf(c))];
  }
  w.sr.sync();
}

/** covarianceMatrixPosePose + covarianceMatricesOfRelativePoses — problem.hpp:204-242,
 *  PBA_INT/covariance_matrices_of_relative_poses.hpp:23-62, se3_motion.hpp:151-158 */
/** device half of estimateUncertainty: linearisation without the robust weight at the current state and the transfer of both
 *  systems (and of the frame states, when a device-driven solve left the host mirror behind) into the pinned buffer
 *  `w.h_uncertainty`; `w.uncertainty_ready` is recorded behind it.  Returns whether the states travel too. */
bool estimateUncertaintyEnqueue(W &w, bool force_state) {
  prepareDevice(w);
  if (w.fej()) firstEstimate(w);
  w.begun = true;
  stageLinearize(w, /*huber=*/false, false, true);
  const int K = w.K();
  const size_t kk = static_cast<size_t>(K) * K;
  const bool want_state = force_state || (w.host_stale && !w.state_dirty);
  const size_t bytes = 2 * kk * sizeof(double) + sizeof(WindowState);
  // (a buffer of its own: the packed per-frame read-back of solve() uses h_export while the host is still working on this one)
  growPinned(w.h_uncertainty, w.h_uncertainty_bytes, bytes);
  if (!w.uncertainty_ready) HIP_CHECK(hipEventCreateWithFlags(&w.uncertainty_ready, hipEventDisableTiming));
  double *Hpp = static_cast<double *>(w.h_uncertainty), *Hsc = Hpp + kk;
  w.d_Hpp.download(Hpp, 2 * kk, 0, w.sr.stream);  // [H_pp | symmetric H_schur], stored back to back by the assemble kernel
  if (want_state) HIP_CHECK(hipMemcpyAsync(Hsc + kk, w.d_state.ptr, sizeof(WindowState), hipMemcpyDeviceToHost, w.sr.stream));
  HIP_CHECK(hipEventRecord(w.uncertainty_ready, w.sr.stream));
  return want_state;
}

void estimateUncertaintyHost(W &w, bool want_state);

void estimateUncertainty(W &w) {
  const bool want_state = estimateUncertaintyEnqueue(w, false);
  estimateUncertaintyHost(w, want_state);
}

/** host half: waits for the transfer only (the stream may already be busy with whatever was enqueued behind it), pseudo-inverse of
 *  the reduced system, relative covariances */
void estimateUncertaintyHost(W &w, bool want_state) {
  const int K = w.K(), F = w.F();
  const size_t kk = static_cast<size_t>(K) * K;
  double *Hpp = static_cast<double *>(w.h_uncertainty), *Hsc = Hpp + kk;
  HIP_CHECK(hipEventSynchronize(w.uncertainty_ready));
  if (want_state) {
    std::memcpy(&w.hst, Hsc + kk, sizeof(WindowState));
    w.host_stale = false;
  }
  hostla::Mat full(kk);
  for (size_t i = 0; i < full.size(); ++i) full[i] = Hpp[i] - Hsc[i] + w.Hm[i];
  const auto t_pinv0 = std::chrono::steady_clock::now();
  const hostla::Mat cov = hostla::pinvDropSmallest(full, K, w.opt.optimize_idepths ? 1 : 0);
  if (std::getenv("DSOPP_HIP_TRACE"))
    std::fprintf(stderr, "[dsopp_hip] estimateUncertainty: pinv of the %d x %d system took %.1f us on the host\n", K, K,
                 std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_pinv0).count());
  downloadState(w);
  for (int r = 0; r < F; ++r)
    for (int t = 0; t < F; ++t) {
      if (r == t) continue;
      double adj[36], s11[36], s22[36], s12[36];
      rigidAdj(rigidMul(rigidInverse(poseOf(w, t)), poseOf(w, r)), adj);
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
          s11[6 * i + j] = cov[static_cast<size_t>(8 * r + i) * K + 8 * r + j];
          s22[6 * i + j] = cov[static_cast<size_t>(8 * t + i) * K + 8 * t + j];
          s12[6 * i + j] = cov[static_cast<size_t>(8 * r + i) * K + 8 * t + j];
        }
      double a11[36], a12[36];
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
          double x = 0, y = 0;
          for (int k = 0; k < 6; ++k) {
            x += adj[6 * i + k] * s11[6 * k + j];
            y += adj[6 * i + k] * s12[6 * k + j];
          }
          a11[6 * i + j] = x;
          a12[6 * i + j] = y;
        }
      std::array<double, 36> out;
      for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
          double x = 0, y = 0;
          for (int k = 0; k < 6; ++k) {
            x += a11[6 * i + k] * adj[6 * j + k];
            y += s12[6 * k + i] * adj[6 * j + k];
          }
          out[static_cast<size_t>(6 * i + j)] = x - y - a12[6 * i + j] + s22[6 * i + j];
        }
      w.frames[static_cast<size_t>(r)]->covariance[w.frames[static_cast<size_t>(t)]->id] = out;
    }
}

/** several device arrays (8-byte words) gathered into ONE pinned host buffer by one launch: the stores go over the host link, the caller
 *  synchronises the stream and reads.  (Five hipMemcpyAsync read-backs of the fold-in ran one behind the other — three copy kernels and two
 *  DMA transfers, 40 us on the timeline — for 66 KB.) */
struct GatherPieces {
  const unsigned long long *src[6];
  int first_word[7];  // piece i lands at words [first_word[i], first_word[i + 1]) of the destination
  int n;
};
__global__ void __launch_bounds__(256) gatherToHostKernel(GatherPieces g, unsigned long long *__restrict__ dst) {
  const int total = g.first_word[g.n];
  for (int i = static_cast<int>(blockIdx.x) * 256 + static_cast<int>(threadIdx.x); i < total; i += static_cast<int>(gridDim.x) * 256) {
    int p = 0;
    while (p + 1 < g.n && i >= g.first_word[p + 1]) ++p;
    dst[i] = g.src[p][i - g.first_word[p]];
  }
}

/** updateMarginalizedLinearSystem — problem.hpp:146-203, called from pushFrame before the new frame is appended */
void foldMarginalized(W &w) {
  HostTimes ht_("foldMarginalized (total)");
  prepareDevice(w);
  firstEstimate(w);  // pushFrame calls firstEstimateJacobians unconditionally (eigen_photometric_bundle_adjustment.cpp:123)
  w.begun = true;
  // evaluateJacobians<..., true, true, true> then changeResidualStatuses (accept), :124-126
  launchSweep(w, true, true, /*for_marg=*/true);
  {
    HIP_CHECK(hipMemsetAsync(w.d_scalars.ptr + 4, 0, 2 * sizeof(double), w.sr.stream));
    // statuses <- candidates without touching idepths: accept with zero steps
    if (w.n_schur_blocks)
      acceptLandmarksKernel<<<w.n_schur_blocks, kSchurThreads, 0, w.sr.stream>>>(w.d_frames.ptr, w.d_schur_table.ptr, w.F(), 1, w.d_scalars.ptr + 4);
    HIP_CHECK(hipGetLastError());
  }
  launchReduceSchur(w, true, nullptr);
  launchAssemble(w, 0.0, false, /*add_priors=*/false, /*store_system=*/true, nullptr);
  energyReduceKernel<<<1, 256, 0, w.sr.stream>>>(w.d_partials.ptr, w.n_sweep_blocks, w.d_scalars.ptr);
  HIP_CHECK(hipGetLastError());
  allreduceIfNeeded(w, w.d_scalars.ptr, 2);
  const int K = w.K(), F = w.F();
  // both systems, the energy scalars and (if a device-driven solve left the host mirror behind) the frame states in ONE
  // synchronisation, through pinned staging (five pageable read-backs cost five staged transfers)
  const size_t kk = static_cast<size_t>(K) * K, k1 = static_cast<size_t>(K);
  const bool want_state = w.host_stale && !w.state_dirty;
  double *stg = static_cast<double *>(stageAcquire(w, (2 * kk + 2 * k1 + 2) * sizeof(double) + sizeof(WindowState)));
  double *Hpp = stg, *Hsc = Hpp + kk, *bpp = Hsc + kk, *bsc = bpp + k1, *scal = bsc + k1;
  {
    static_assert(sizeof(WindowState) % 8 == 0, "the frame states are gathered as 8-byte words");
    GatherPieces g;
    std::memset(&g, 0, sizeof(g));
    const void *src[6] = {w.d_Hpp.ptr, w.dHsc(), w.d_bpp.ptr, w.dbsc(), w.d_scalars.ptr, w.d_state.ptr};
    const size_t words[6] = {kk, kk, k1, k1, 2, sizeof(WindowState) / 8};
    g.n = want_state ? 6 : 5;
    for (int i = 0; i < g.n; ++i) {
      g.src[i] = static_cast<const unsigned long long *>(src[i]);
      g.first_word[i + 1] = g.first_word[i] + static_cast<int>(words[i]);
    }
    const int blocks = std::min(64, (g.first_word[g.n] + 1023) / 1024);
    gatherToHostKernel<<<static_cast<unsigned>(blocks), 256, 0, w.sr.stream>>>(g, reinterpret_cast<unsigned long long *>(stg));
    HIP_CHECK(hipGetLastError());
  }
  {
    HostTimes ht2_("foldMarginalized: wait for the device");
    w.sr.sync();
  }
  HostTimes ht3_("foldMarginalized: host arithmetic behind the wait");
  if (want_state) {
    std::memcpy(&w.hst, scal + 2, sizeof(WindowState));
    w.host_stale = false;
  }
  std::vector<double> state(static_cast<size_t>(K));
  for (int f = 0; f < F; ++f)
    for (int a = 0; a < kBlk; ++a) state[static_cast<size_t>(kBlk * f + a)] = w.hst.eps[f][a];
  std::vector<double> Hp(static_cast<size_t>(K) * K), bp(static_cast<size_t>(K));
  for (size_t i = 0; i < Hp.size(); ++i) Hp[i] = Hpp[i] - Hsc[i];
  for (int i = 0; i < K; ++i) bp[static_cast<size_t>(i)] = bpp[static_cast<size_t>(i)] - bsc[static_cast<size_t>(i)];
  std::vector<double> Hs(static_cast<size_t>(K), 0.0);
  for (int i = 0; i < K; ++i)
    for (int j = 0; j < K; ++j) Hs[static_cast<size_t>(i)] += Hp[static_cast<size_t>(i) * K + j] * state[static_cast<size_t>(j)];
  double sHs = 0, sb = 0;
  for (int i = 0; i < K; ++i) {
    sHs += state[static_cast<size_t>(i)] * Hs[static_cast<size_t>(i)];
    sb += state[static_cast<size_t>(i)] * bp[static_cast<size_t>(i)];
  }
  w.energy_marginalized += scal[0] + sHs - sb;  // eq. 8.15 of the DSO paper, problem.hpp:168-170
  for (int i = 0; i < K; ++i) bp[static_cast<size_t>(i)] -= Hs[static_cast<size_t>(i)];
  for (size_t i = 0; i < Hp.size(); ++i) w.Hm[i] += Hp[i];
  for (int i = 0; i < K; ++i) w.bm[static_cast<size_t>(i)] += bp[static_cast<size_t>(i)];
  // landmark.to_marginalize = false for all landmarks (:175-177)
  for (auto &fp : w.frames) {
    HostFrame &f = *fp;
    bool changed = false;
    for (auto &fl : f.flags)
      if (fl & kFlagToMarginalize) {
        fl &= static_cast<uint8_t>(~kFlagToMarginalize);
        changed = true;
      }
    if (changed && f.n)  // on the device too (its other bits — outlier, ill_conditioned — are the device's own)
      {
        W::AppendOp op{};  // queued: leaves with the next flush (prepare() of the solve behind this pushFrame, or any reader)
        op.kind = 3;
        op.n = f.n;
        op.a = kFlagToMarginalize;
        op.src_off = queueAppendData(w, nullptr, 0, f.dflags.ptr);
        op.dst = f.dflags.ptr;
        w.append_ops.push_back(op);
      }
  }
  std::vector<int> marginalized_part;
  for (int f = 0; f < F; ++f)
    if (w.frames[static_cast<size_t>(f)]->to_marginalize)
      for (int p = 0; p < kBlk; ++p) marginalized_part.push_back(kBlk * f + p);
  w.marg_dirty = true;
  if (marginalized_part.empty()) return;
  // prior of the frames being marginalised (evaluateLinearSystemPrior with for_marginalized = true, problem.hpp:193-197)
  std::vector<double> Hprior(static_cast<size_t>(K) * K, 0.0), bprior(static_cast<size_t>(K), 0.0);
  for (int f = 0; f < F; ++f) {
    HostFrame &hf = *w.frames[static_cast<size_t>(f)];
    if (!hf.to_marginalize) continue;
    if (hf.fixed) {
      for (int a = 0; a < kBlk; ++a) {
        Hprior[static_cast<size_t>(kBlk * f + a) * K + kBlk * f + a] += w.opt.fixed_state_regularizer;
        bprior[static_cast<size_t>(kBlk * f + a)] += w.opt.fixed_state_regularizer * w.hst.eps[f][a];
      }
    } else {
      for (int a = 0; a < 2; ++a) {
        const double ab = w.hst.ab0[f][a] + w.hst.eps[f][6 + a];
        Hprior[static_cast<size_t>(kBlk * f + 6 + a) * K + kBlk * f + 6 + a] += w.opt.affine_brightness_regularizer[a];
        bprior[static_cast<size_t>(kBlk * f + 6 + a)] += w.opt.affine_brightness_regularizer[a] * ab;
      }
    }
  }
  for (int i = 0; i < K; ++i) {
    double s = 0;
    for (int j = 0; j < K; ++j) s += Hprior[static_cast<size_t>(i) * K + j] * state[static_cast<size_t>(j)];
    bprior[static_cast<size_t>(i)] -= s;
  }
  for (size_t i = 0; i < Hprior.size(); ++i) w.Hm[i] += Hprior[i];
  for (int i = 0; i < K; ++i) w.bm[static_cast<size_t>(i)] += bprior[static_cast<size_t>(i)];
  hostla::reduceSystem(w.Hm, w.bm, K, marginalized_part);
  // erase the marginalised frames (:201-202) — state rows shift down
  WindowState ns = w.hst;
  std::vector<std::unique_ptr<HostFrame>> kept;
  std::vector<int> erased_ids;
  int dst = 0;
  for (int f = 0; f < F; ++f) {
    if (w.frames[static_cast<size_t>(f)]->to_marginalize) {
      // recycle the device buffers of the leaving frame: its own connection tables and landmark arrays
      std::unique_ptr<HostFrame> gone = std::move(w.frames[static_cast<size_t>(f)]);
      for (auto &kv : gone->residuals)
        if (kv.second) w.table_pool.push_back(std::move(kv.second));
      gone->residuals.clear();
      gone->pending.clear();
      gone->covariance.clear();
      erased_ids.push_back(gone->id);
      w.frame_pool.push_back(std::move(gone));
      continue;
    }
    std::memcpy(ns.T0_R[dst], w.hst.T0_R[f], sizeof(ns.T0_R[dst]));
    std::memcpy(ns.T0_t[dst], w.hst.T0_t[f], sizeof(ns.T0_t[dst]));
    std::memcpy(ns.ab0[dst], w.hst.ab0[f], sizeof(ns.ab0[dst]));
    std::memcpy(ns.eps[dst], w.hst.eps[f], sizeof(ns.eps[dst]));
    std::memcpy(ns.step[dst], w.hst.step[f], sizeof(ns.step[dst]));
    kept.push_back(std::move(w.frames[static_cast<size_t>(f)]));
    ++dst;
  }
  for (auto &kf : kept)  // ... and the remaining frames' tables towards it
    for (int id : erased_ids) {
      auto it = kf->residuals.find(id);
      if (it == kf->residuals.end()) continue;
      if (it->second) w.table_pool.push_back(std::move(it->second));
      kf->residuals.erase(it);
    }
  w.hst = ns;
  w.frames = std::move(kept);
  w.marg_size = kBlk * w.F();
  w.state_dirty = true;
  w.topology_dirty = true;
}

/** levenberg_marquardt_algorithm::solve (levenberg_marquardt_algorithm.hpp:77-128) driving the device stages.
 *  Round 1: control flow on the host (one small D2H per energy evaluation). */
void lmSolve(W &w, double &energy_out, int &iterations, int &n_valid_out) {
  const double lambda0 = 1.0 / w.opt.initial_trust_region_radius;
  const size_t max_it = static_cast<size_t>(w.opt.max_iterations), min_it = 3;
  const bool force_accept = w.opt.force_accept != 0;
  double lambda = lambda0;
  auto e0 = stageEnergy(w);
  double energy = e0.first;
  int n_valid = e0.second;
  bool converged = false, linear_system_valid = false;
  iterations = 0;
  bool early_return = false;
  for (size_t it = 0; it < max_it && !converged && n_valid > 0; ++it) {
    ++iterations;
    if (!linear_system_valid) stageLinearize(w);
    stageStep(w, lambda);
    auto en = stageEnergy(w);
    if (en.second == 0) {
      stageAccept(w, false);
      break;
    }
    converged |= std::abs(energy - en.first) / energy < w.opt.function_tolerance;
    if (en.first < energy || (force_accept && it < min_it)) {
      auto nr = stageAccept(w, true);
      converged |= nr.second < w.opt.parameter_tolerance * (nr.first + w.opt.parameter_tolerance);
      energy = en.first;
      n_valid = en.second;
      lambda /= 1.0;  // decrease_on_accept = 1 (eigen_photometric_bundle_adjustment.cpp:74)
      linear_system_valid = false;
    } else {
      stageAccept(w, false);
      if (force_accept) {
        stageEnergy(w);
        early_return = true;
        break;
      }
      lambda *= 1.0;
      linear_system_valid = true;
    }
  }
  if (!early_return) stageEnergy(w);
  energy_out = energy;
  n_valid_out = n_valid;
}

/** entries [current size, n) of the (f, target) connection: the table comes from the pool (or is allocated), the entries are queued */
void appendConnection(W &w, HostFrame &f, int target_id, int n, const uint8_t *statuses) {
  hipStream_t st = w.sr.stream;
  {
    // nothing to append to a list that exists: the call leaves the window as it is (no table rebuild behind it) — most of a keyframe step's
    // ~130 set_connection calls, since LocalFrame::update walks every connection of every keyframe
    const auto have = f.residuals.find(target_id);
    if (have != f.residuals.end() && have->second && n <= have->second->n) return;
  }
  auto &slot = f.residuals[target_id];
  if (!slot) {
    if (!w.table_pool.empty()) {
      // best fit: the smallest pooled table that holds the frame's capacity (the tracker's first keyframe carries more landmarks than the
      // others: its tables towards every new keyframe found a smaller pooled table on top and re-allocated all four arrays — a flush, four
      // malloc + fill + synchronise + free, ~0.1 ms per keyframe — while the table its last connection had returned lay further down)
      const size_t need = static_cast<size_t>(std::max(f.cap, kFirstLandmarkCapacity));
      size_t pick = w.table_pool.size() - 1;
      bool fits = false;
      for (size_t k = 0; k < w.table_pool.size(); ++k) {
        const size_t have = w.table_pool[k]->status.capacity;
        if (have < need) continue;
        if (!fits || have < w.table_pool[pick]->status.capacity) pick = k, fits = true;
      }
      slot = std::move(w.table_pool[pick]);
      w.table_pool.erase(w.table_pool.begin() + static_cast<std::ptrdiff_t>(pick));
      slot->n = 0;
      slot->snap_n = 0;
    } else {
      slot = std::make_unique<ResidualTable>();
    }
  }
  ResidualTable &rt = *slot;
  // (at least kFirstLandmarkCapacity entries: tables come back from the pool with whatever capacity their last owner needed, and a frame's
  // capacity starts there — a smaller pooled table cost a flush of the queue + four reallocations (malloc, fill, copy, synchronise, free) per connection)
  const size_t cap = static_cast<size_t>(std::max(f.cap, kFirstLandmarkCapacity));
  const size_t keep = static_cast<size_t>(rt.n);
  if (rt.status.ptr && cap > rt.status.capacity) flushAppends(w, "flush: connection table grows");  // the arrays move: queued appends hold their old addresses
  rt.status.reserve(cap, keep, st);
  rt.cand.reserve(cap, keep, st);
  rt.fej_valid.reserve(cap, keep, st);
  rt.energy.reserve(cap, keep, st);
  if (n > rt.n) {
    const size_t add = static_cast<size_t>(n - rt.n);
    W::AppendOp op{};
    op.kind = 2;
    op.n = static_cast<int>(add);
    op.a = static_cast<int>(keep);
    if (f.permuted()) {
      // the list grows by the caller's landmarks [rt.n, n): they must be the device's [rt.n, n) as well (splitBatchAt), in the device's order
      splitBatchAt(w, f, n);
      std::vector<uint8_t> staged(add);
      for (size_t k = 0; k < add; ++k) staged[k] = statuses[f.to_caller[keep + k]];
      op.src_off = queueAppendData(w, staged.data(), add, rt.status.ptr);
    } else {
      op.src_off = queueAppendData(w, statuses + rt.n, add, rt.status.ptr);
    }
    op.dst = rt.status.ptr;
    op.p[0] = rt.cand.ptr;
    op.p[1] = rt.fej_valid.ptr;
    op.p[2] = rt.energy.ptr;
    w.append_ops.push_back(op);
    rt.n = n;
  }
  w.topology_dirty = true;
  w.begun = false;
}

}  // namespace
}  // namespace dsopp_hip

// ---------------------------------------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------------------------------------
extern "C" {

void dsopp_hip_default_pba_options(dsopp_hip_options *o) {
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
  o->dtype = DSOPP_HIP_F64;
}

void dsopp_hip_default_align_options(dsopp_hip_options *o) {
  dsopp_hip_default_pba_options(o);
  o->max_iterations = 50;
  o->initial_trust_region_radius = 1e2;
  o->function_tolerance = 1e-5;
  o->parameter_tolerance = 1e-5;
  o->force_accept = 0;
}

int dsopp_hip_window_create(const dsopp_hip_options *options, int device, void *stream, dsopp_hip_window **out) {
  return guarded([&] {
    if (!options || !out) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    auto w = std::make_unique<dsopp_hip_window>();
    w->opt = *options;
    if (w->opt.dtype != DSOPP_HIP_F64 && w->opt.dtype != DSOPP_HIP_F32) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad dtype");
    w->sr.init(device, stream);
    std::memset(&w->hst, 0, sizeof(w->hst));
    HIP_CHECK(hipEventCreate(&w->ev0));
    HIP_CHECK(hipEventCreate(&w->ev1));
    *out = w.release();
  });
}

void dsopp_hip_window_destroy(dsopp_hip_window *w) {
  if (!w) return;
  (void)hipSetDevice(w->sr.device);
  if (w->sr.stream) (void)hipStreamSynchronize(w->sr.stream);
  if (w->ev0) (void)hipEventDestroy(w->ev0);
  if (w->ev1) (void)hipEventDestroy(w->ev1);
  if (w->h_ctrl) (void)hipHostFree(w->h_ctrl);
  if (w->h_results) (void)hipHostFree(w->h_results);
  if (w->h_uncertainty) (void)hipHostFree(w->h_uncertainty);
  if (w->uncertainty_ready) (void)hipEventDestroy(w->uncertainty_ready);
  if (w->h_export) (void)hipHostFree(w->h_export);
  if (w->stage.base) (void)hipHostFree(w->stage.base);
  if (w->h_update) (void)hipHostFree(w->h_update);
  if (w->d_bs_flag) (void)hipFree(w->d_bs_flag);
  if (w->h_bs_fault) (void)hipHostFree(w->h_bs_fault);
  w->frames.clear();
  for (dsopp_hip_depth_maps *m : w->live_maps) {  // maps may outlive the window (the tracker holds them): they lose the borrowed stream
    m->sr.stream = nullptr;
    m->owner = nullptr;
  }
  StreamRef sr = w->sr;
  delete w;
  sr.destroy();
}

int dsopp_hip_window_push_frame(dsopp_hip_window *w, int32_t frame_id, int64_t timestamp, const dsopp_hip_pyramid *pyramid, int level,
                                const double intrinsics[4], const double T_world_agent[7], double exposure_time,
                                const double affine_brightness[2], int fixed, int is_marginalized) {
  return guarded([&] {
    HostTimes ht_("push_frame (total)");
    if (w) w->export_valid = false;
    if (!w || !pyramid || !intrinsics || !T_world_agent || !affine_brightness) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (level < 0 || level >= pyramid->levels) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "level out of range");
    if (pyramid->dtype != w->opt.dtype) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "pyramid dtype differs from the window's");
    if (pyramid->sr.device != w->sr.device) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "pyramid lives on another device");
    if (!w->frames.empty() && !(w->frames.back()->timestamp < timestamp))
      fail(DSOPP_HIP_ERR_ORDER, "frames must be pushed in ascending order of time");
    if (w->slotOf(frame_id) >= 0) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "frame %d already in the window", frame_id);
    w->sr.use();
    flushAppends(*w);
    if (w->frames.size() > 1) foldMarginalized(*w);
    if (w->F() >= kMaxFrames) fail(DSOPP_HIP_ERR_CAPACITY, "window holds %d frames already", kMaxFrames);
    downloadState(*w);
    std::unique_ptr<HostFrame> f;
    if (!w->frame_pool.empty()) {  // device arrays of a keyframe that left the window (capacity kept, contents dead)
      f = std::move(w->frame_pool.back());
      w->frame_pool.pop_back();
      f->n = 0;
      f->snap_n = 0;
      f->flags.clear();
      f->to_internal.clear();
      f->to_caller.clear();
      f->batch_end.clear();
      f->to_marginalize = false;
    } else {
      f = std::make_unique<HostFrame>();
    }
    f->id = frame_id;
    f->timestamp = timestamp;
    f->pyramid = pyramid;
    pyramid->waitReady(w->sr.stream);  // a build_device still in flight on the pyramid's stream
    f->level = level;
    for (int i = 0; i < 4; ++i) f->intr[i] = intrinsics[i];
    f->exposure = exposure_time;
    f->fixed = fixed != 0;
    f->is_marginalized = is_marginalized != 0;
    const int s = w->F();
    const Rigid T = rigidFromParams(T_world_agent);
    for (int i = 0; i < 9; ++i) w->hst.T0_R[s][i] = T.R[i];
    for (int i = 0; i < 3; ++i) w->hst.T0_t[s][i] = T.t[i];
    w->hst.ab0[s][0] = affine_brightness[0];
    w->hst.ab0[s][1] = affine_brightness[1];
    for (int a = 0; a < kBlk; ++a) w->hst.eps[s][a] = w->hst.step[s][a] = 0;
    w->frames.push_back(std::move(f));
    for (auto &h : w->frames) {  // connections towards this id that were declared before the frame came
      auto it = h->pending.find(frame_id);
      if (it == h->pending.end()) continue;
      std::vector<uint8_t> statuses = std::move(it->second);
      h->pending.erase(it);
      if (!statuses.empty() && static_cast<int>(statuses.size()) <= h->n) appendConnection(*w, *h, frame_id, static_cast<int>(statuses.size()), statuses.data());
    }
    // system_marginalized_.resize(new_size) keeping old entries, new rows/cols zero (:134-140)
    const int Kn = w->K(), Ko = w->marg_size;
    std::vector<double> Hn(static_cast<size_t>(Kn) * Kn, 0.0), bn(static_cast<size_t>(Kn), 0.0);
    for (int i = 0; i < std::min(Ko, Kn); ++i) {
      bn[static_cast<size_t>(i)] = w->bm[static_cast<size_t>(i)];
      for (int j = 0; j < std::min(Ko, Kn); ++j) Hn[static_cast<size_t>(i) * Kn + j] = w->Hm[static_cast<size_t>(i) * Ko + j];
    }
    w->Hm.swap(Hn);
    w->bm.swap(bn);
    w->marg_size = Kn;
    w->marg_dirty = true;
    w->state_dirty = true;
    w->topology_dirty = true;
    w->begun = false;
  });
}

int dsopp_hip_window_set_landmarks(dsopp_hip_window *w, int32_t frame_id, int32_t n_total, const double *uv, const double *idepth,
                                   const double *patch, const uint8_t *flags) {
  return guarded([&] {
    HostTimes ht_("set_landmarks");
    if (w) w->export_valid = false;
    if (!w || n_total < 0 || (n_total && (!uv || !idepth || !patch || !flags))) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    w->sr.use();
    HostFrame &f = w->frameById(frame_id);
    if (n_total < f.n) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "landmarks can only be appended (%d < %d)", n_total, f.n);
    hipStream_t st = w->sr.stream;
    ensureLandmarkCapacity(*w, f, n_total);
    const int old = f.n;
    // existing landmarks: only flags move (LocalFrame::update, local_frame.hpp:492-497).  The host mirror holds the bits the
    // host decides (marginalized, to_marginalize); the device flags also carry outlier / ill_conditioned, which stay
    f.flags.resize(static_cast<size_t>(n_total));
    bool changed = false;
    for (int i = 0; i < old; ++i) {
      const bool was_marg = f.flags[static_cast<size_t>(i)] & kFlagMarginalized;
      const bool marg = flags[i] & 1, outl = flags[i] & 2;
      uint8_t v = 0;  // (to_marginalize is ASSIGNED by every update, local_frame.hpp:493-496: a pending one does not survive a second update)
      if (marg) v |= kFlagMarginalized;
      if (!was_marg && marg && !outl) v |= kFlagToMarginalize;
      changed = changed || v != f.flags[static_cast<size_t>(i)];
      f.flags[static_cast<size_t>(i)] = v;
    }
    // An update that brings no landmark and leaves every host-decided flag as it is changes nothing on the device: no operation is queued and
    // the window's tables stay valid.  (The tracker calls updateLocalFrame for every keyframe of the window three times per keyframe step —
    // after the activation, in pushFrame, at the marginalisation: two thirds of those 21 calls are of this kind.)
    if (n_total == old && !changed) return;
    for (int i = old; i < n_total; ++i)
      f.flags[static_cast<size_t>(i)] = static_cast<uint8_t>(((flags[i] & 1) ? kFlagMarginalized : 0) | ((flags[i] & 2) ? kFlagOutlier : 0));
    // the new batch in the device's order: by 32 x 32-pixel tile (rows of tiles, then tiles), raster inside a tile — what a grid-cell
    // feature extractor yields, and the order in which the sweep's neighbouring items share texel granules (HostFrame::to_internal)
    const size_t add = static_cast<size_t>(n_total - old);
    if (add && (f.permuted() || (old == 0 && sortLandmarksInternally()))) {
      std::vector<int> order(add);
      for (size_t k = 0; k < add; ++k) order[k] = old + static_cast<int>(k);
      // (experiment switches, defaults = what is described above: DSOPP_HIP_LANDMARK_TILE = log2 of the tile edge, DSOPP_HIP_LANDMARK_INNER =
      // raster | morton | snake (rows of a tile alternate direction, tile rows alternate direction))
      static const int tb = std::getenv("DSOPP_HIP_LANDMARK_TILE") ? std::max(2, std::min(8, std::atoi(std::getenv("DSOPP_HIP_LANDMARK_TILE")))) : 5;
      static const int inner = [] {
        const char *e = std::getenv("DSOPP_HIP_LANDMARK_INNER");
        return !e ? 0 : (std::string(e) == "morton" ? 1 : (std::string(e) == "snake" ? 2 : 0));
      }();
      auto key = [&](int c) {
        // (a coordinate that is not a number, negative or beyond any image is clamped instead of cast: the cast would be undefined behaviour,
        // and such a landmark is uploaded like any other — the sweep rejects it by its ROI test, as before the internal order existed)
        auto coord = [](double x) -> long long { return !(x >= 0.0) ? 0ll : (x > 65535.0 ? 65535ll : static_cast<long long>(x)); };
        const long long u = coord(uv[2 * c]), v = coord(uv[2 * c + 1]);
        const long long m = (1ll << tb) - 1;
        long long tu = u >> tb, tv = v >> tb, iu = u & m, iv = v & m, in;
        if (inner == 1) {
          in = 0;
          for (int b = 0; b < tb; ++b) in |= ((iu >> b) & 1) << (2 * b) | ((iv >> b) & 1) << (2 * b + 1);
        } else {
          if (inner == 2) {
            if (tv & 1) tu = 65535 - tu;
            if (iv & 1) iu = m - iu;
          }
          in = (iv << tb) + iu;
        }
        return ((tv * 65536 + tu) << (2 * tb)) + in;
      };
      // (ties — two landmarks on one pixel — are broken by their content, so that the device order, and with it every sum of the
      // deterministic build, depends on the SET of landmarks only, not on the order the caller lists them in)
      std::vector<long long> keys(add);  // computed once per landmark, not twice per comparison (the sort was 0.1 ms of a keyframe's appends)
      for (size_t k = 0; k < add; ++k) keys[k] = key(old + static_cast<int>(k));
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        const long long ka = keys[static_cast<size_t>(a - old)], kb = keys[static_cast<size_t>(b - old)];
        if (ka != kb) return ka < kb;
        // bit patterns, not values: a strict weak order whatever the numbers are (a NaN compares false both ways as a value, which
        // std::stable_sort must never be given)
        unsigned long long ia, ib;
        std::memcpy(&ia, idepth + a, sizeof(ia));
        std::memcpy(&ib, idepth + b, sizeof(ib));
        if (ia != ib) return ia < ib;
        return std::memcmp(patch + static_cast<size_t>(kPat) * a, patch + static_cast<size_t>(kPat) * b, sizeof(double) * kPat) < 0;
      });
      f.to_internal.resize(static_cast<size_t>(n_total));
      f.to_caller.resize(static_cast<size_t>(n_total));
      for (size_t k = 0; k < add; ++k) {
        f.to_internal[static_cast<size_t>(order[k])] = old + static_cast<int>(k);
        f.to_caller[static_cast<size_t>(old) + k] = order[k];
      }
      f.d_to_internal.reserve(static_cast<size_t>(std::max(f.cap, n_total)), static_cast<size_t>(old), st);
      uploadStaged(*w, f.d_to_internal, f.to_internal.data() + old, add, static_cast<size_t>(old));
    }
    if (add) f.batch_end.push_back(n_total);
    const bool permuted = f.permuted();
    if (n_total) {
      // flag bits in the device's landmark order -> queued merge (applyAppendsKernel, kind 1)
      W::AppendOp op{};
      op.kind = 1;
      op.n = n_total;
      op.a = old;
      if (permuted) {
        std::vector<uint8_t> staged(static_cast<size_t>(n_total));
        for (int p = 0; p < n_total; ++p) staged[static_cast<size_t>(p)] = f.flags[static_cast<size_t>(f.to_caller[static_cast<size_t>(p)])];
        op.src_off = queueAppendData(*w, staged.data(), staged.size(), f.dflags.ptr);
      } else {
        op.src_off = queueAppendData(*w, f.flags.data(), static_cast<size_t>(n_total), f.dflags.ptr);
      }
      op.dst = f.dflags.ptr;
      if (n_total > old) {
        op.p[0] = f.idepth_step.ptr;
        op.p[1] = f.idepth_fej.ptr;
        op.p[2] = f.inv_hdd.ptr;
        op.p[3] = f.b_d.ptr;
        op.p[4] = f.relative_baseline.ptr;
        op.p[5] = f.n_inliers.ptr;
      }
      w->append_ops.push_back(op);
    }
    for (int i = old; i < n_total; ++i) f.flags[static_cast<size_t>(i)] &= kFlagMarginalized;  // the mirror keeps host-decided bits only
    if (add && permuted) {
      std::vector<double> s_uv(2 * add), s_id(add), s_patch(static_cast<size_t>(kPat) * add);
      for (size_t k = 0; k < add; ++k) {
        const size_t c = static_cast<size_t>(f.to_caller[static_cast<size_t>(old) + k]);
        s_uv[2 * k] = uv[2 * c];
        s_uv[2 * k + 1] = uv[2 * c + 1];
        s_id[k] = idepth[c];
        std::memcpy(&s_patch[static_cast<size_t>(kPat) * k], patch + static_cast<size_t>(kPat) * c, kPat * sizeof(double));
      }
      uploadStaged(*w, f.uv, s_uv.data(), 2 * add, 2 * static_cast<size_t>(old));
      uploadStaged(*w, f.idepth, s_id.data(), add, static_cast<size_t>(old));
      uploadStaged(*w, f.patch, s_patch.data(), kPat * add, kPat * static_cast<size_t>(old));
    } else if (add) {
      uploadStaged(*w, f.uv, uv + 2 * old, 2 * add, 2 * static_cast<size_t>(old));
      uploadStaged(*w, f.idepth, idepth + old, add, static_cast<size_t>(old));
      uploadStaged(*w, f.patch, patch + kPat * old, kPat * add, kPat * static_cast<size_t>(old));
    }
    f.n = n_total;
    if (add) w->topology_dirty = true;  // (a flag update leaves counts and addresses — all the tables hold — as they are)
    w->begun = false;
  });
}

int dsopp_hip_window_set_connection(dsopp_hip_window *w, int32_t reference_id, int32_t target_id, int32_t n, const uint8_t *statuses) {
  return guarded([&] {
    HostTimes ht_("set_connection");
    if (w) w->export_valid = false;
    if (!w || n < 0 || (n && !statuses)) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    w->sr.use();
    HostFrame &f = w->frameById(reference_id);
    if (n > f.n) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "connection has %d entries, frame %d only %d landmarks", n, reference_id, f.n);
    if (w->slotOf(target_id) < 0 && f.residuals.find(target_id) == f.residuals.end()) {
      // the target is not in the window: the list waits on the host (materialised by the push_frame that brings the target)
      // (the whole list as the caller holds it now: for a target that LEFT the window these are the statuses the last updateFrame handed out —
      // what the reference's residual list towards a dropped frame keeps, and what get_frame_update returns for such a target)
      std::vector<uint8_t> &pv = f.pending[target_id];
      if (static_cast<size_t>(n) >= pv.size()) pv.assign(statuses, statuses + n);
      return;
    }
    appendConnection(*w, f, target_id, n, statuses);
  });
}

int dsopp_hip_window_mark_frame_marginalized(dsopp_hip_window *w, int32_t frame_id) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    HostFrame &f = w->frameById(frame_id);
    f.to_marginalize = !f.is_marginalized;
    f.is_marginalized = true;
    w->frames_dirty = true;
  });
}

int dsopp_hip_window_num_frames(dsopp_hip_window *w, int32_t *n) {
  return guarded([&] {
    if (!w || !n) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *n = w->F();
  });
}

int dsopp_hip_window_frame_ids(dsopp_hip_window *w, int32_t capacity, int32_t *ids, int32_t *n) {
  return guarded([&] {
    if (!w || !n || (capacity > 0 && !ids)) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *n = w->F();
    for (int i = 0; i < w->F() && i < capacity; ++i) ids[i] = w->frames[static_cast<size_t>(i)]->id;
  });
}

int dsopp_hip_window_begin(dsopp_hip_window *w) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    stageBegin(*w);
  });
}

int dsopp_hip_window_calculate_energy(dsopp_hip_window *w, double *energy, int32_t *n_valid) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    w->sr.use();
    flushAppends(*w);
    auto r = stageEnergy(*w);
    if (energy) *energy = r.first;
    if (n_valid) *n_valid = r.second;
  });
}

int dsopp_hip_window_linearize(dsopp_hip_window *w) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    w->sr.use();
    flushAppends(*w);
    stageLinearize(*w);
    w->sr.sync();
  });
}

int dsopp_hip_window_get_system(dsopp_hip_window *w, double *H_pp, double *b_pp, double *H_schur, double *b_schur) {
  return guarded([&] {
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    if (!w->linearized) fail(DSOPP_HIP_ERR_STATE, "no linearised system available");
    w->sr.use();
    flushAppends(*w);
    const size_t K = static_cast<size_t>(w->K());
    hipStream_t st = w->sr.stream;
    if (H_pp) w->d_Hpp.download(H_pp, K * K, 0, st);
    if (b_pp) w->d_bpp.download(b_pp, K, 0, st);
    if (H_schur) w->d_HscDownload(H_schur, K * K, 0, st);
    if (b_schur) w->d_bscDownload(b_schur, K, 0, st);
    w->sr.sync();
  });
}

int dsopp_hip_window_calculate_step(dsopp_hip_window *w, double lambda, double *step) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    w->sr.use();
    flushAppends(*w);
    stageStep(*w, lambda);
    if (step) std::memcpy(step, w->last_step.data(), w->last_step.size() * sizeof(double));
  });
}

int dsopp_hip_window_accept_step(dsopp_hip_window *w, double *state_sq, double *step_sq) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    w->sr.use();
    flushAppends(*w);
    auto r = stageAccept(*w, true);
    if (state_sq) *state_sq = r.first;
    if (step_sq) *step_sq = r.second;
  });
}

int dsopp_hip_window_reject_step(dsopp_hip_window *w) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    w->sr.use();
    flushAppends(*w);
    stageAccept(*w, false);
  });
}

int dsopp_hip_window_update_point_statuses(dsopp_hip_window *w) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    updatePointStatuses(*w);
  });
}

static void runOptimize(dsopp_hip_window *w, double &e, int &it, int &nv) {
  if (w->F() == 0) fail(DSOPP_HIP_ERR_STATE, "window is empty");
  w->sr.use();
  prepare(*w);
  HIP_CHECK(hipEventRecord(w->ev0, w->sr.stream));
  if (w->lm_mode == 0)
    fusedBegin(*w);
  else
    stageBegin(*w);
  if (w->lm_mode == 1)
    lmSolve(*w, e, it, nv);
  else if (w->lm_mode == 2)
    lmSolveDevice(*w, e, it, nv);
  else
    lmSolveFused(*w, e, it, nv);
  HIP_CHECK(hipEventRecord(w->ev1, w->sr.stream));
  w->solve_events_pending = true;  // the elapsed time is read when somebody asks for it (dsopp_hip_window_last_solve_ms)
  collectTimings(*w);
}

int dsopp_hip_window_optimize(dsopp_hip_window *w, double *energy, int32_t *iterations, int32_t *n_valid) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    double e = 0;
    int it = 0, nv = 0;
    runOptimize(w, e, it, nv);
    if (energy) *energy = e;
    if (iterations) *iterations = it;
    if (n_valid) *n_valid = nv;
  });
}

namespace {
/** packs the updateFrame data of every keyframe (exportFrameKernel per frame, statuses towards all other frames of the window)
 *  and starts ONE transfer into pinned memory; the caller synchronises */
void prefetchFrameUpdates(dsopp_hip_window &w) {
  hipStream_t st = w.sr.stream;
  w.export_entries.clear();
  size_t words = 0;
  for (auto &fp : w.frames) {
    HostFrame &f = *fp;
    if (f.n == 0) continue;
    dsopp_hip_window::ExportEntry e;
    e.frame_id = f.id;
    e.n = f.n;
    e.word_offset = words;
    for (auto &gp : w.frames) {
      if (gp.get() == &f) continue;
      auto it = f.residuals.find(gp->id);
      if (it != f.residuals.end() && it->second && it->second->n == f.n) e.target_ids.push_back(gp->id);
    }
    const size_t n = static_cast<size_t>(f.n);
    words += 4 * n + ((1 + e.target_ids.size()) * n + 7) / 8;
    w.export_entries.push_back(std::move(e));
  }
  if (!words) return;
  w.d_update.reserve(words, 0, st);
  FrameExportBatch batch;
  batch.n_frames = 0;
  int max_n = 0;
  for (const auto &e : w.export_entries) {
    HostFrame &f = w.frameById(e.frame_id);
    FrameExportArgs &a = batch.f[batch.n_frames++];
    max_n = std::max(max_n, f.n);
    a.idepth = f.idepth.ptr;
    a.inv_hdd = f.inv_hdd.ptr;
    a.relative_baseline = f.relative_baseline.ptr;
    a.n_inliers = f.n_inliers.ptr;
    a.flags = f.dflags.ptr;
    a.to_internal = f.permuted() ? f.d_to_internal.ptr : nullptr;
    a.n = f.n;
    a.n_targets = static_cast<int>(e.target_ids.size());
    for (int t = 0; t < a.n_targets; ++t) a.status[t] = f.residuals[e.target_ids[static_cast<size_t>(t)]]->status.ptr;
    a.out_d = w.d_update.ptr + e.word_offset;
    a.out_b = reinterpret_cast<uint8_t *>(a.out_d + 4 * static_cast<size_t>(f.n));
  }
  exportFramesKernel<<<dim3(static_cast<unsigned>((max_n + 255) / 256), static_cast<unsigned>(batch.n_frames)), 256, 0, st>>>(batch);
  HIP_CHECK(hipGetLastError());
  growPinned(w.h_update, w.h_update_bytes, words * 8);
  HIP_CHECK(hipMemcpyAsync(w.h_update, w.d_update.ptr, words * 8, hipMemcpyDeviceToHost, st));
}
}  // namespace

int dsopp_hip_window_optimize_async(dsopp_hip_window *w) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    if (w->F() == 0) fail(DSOPP_HIP_ERR_STATE, "window is empty");
    if (w->lm_mode != 0) fail(DSOPP_HIP_ERR_STATE, "the asynchronous solve exists for the fused device loop only (lm_mode 0)");
    if (w->async_pending) fail(DSOPP_HIP_ERR_STATE, "an asynchronous solve is already pending: call dsopp_hip_window_optimize_wait first");
    w->sr.use();
    flushAppends(*w);
    prepare(*w);
    fusedBegin(*w);
    lmSolveFusedEnqueue(*w);
    w->async_pending = true;
  });
}

int dsopp_hip_window_optimize_wait(dsopp_hip_window *w, double *energy, int32_t *iterations, int32_t *n_valid) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    if (!w->async_pending) fail(DSOPP_HIP_ERR_STATE, "no asynchronous solve is pending");
    w->sr.use();
    flushAppends(*w);
    double e = 0;
    int it = 0, nv = 0;
    w->async_pending = false;
    lmSolveFusedFinish(*w, e, it, nv);
    collectTimings(*w);
    if (energy) *energy = e;
    if (iterations) *iterations = it;
    if (n_valid) *n_valid = nv;
  });
}

int dsopp_hip_window_solve(dsopp_hip_window *w, double *energy, int32_t *iterations, int32_t *n_valid) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    double e = 0;
    int it = 0, nv = 0;
    const bool sharded = w->allreduce && w->world > 1;
    static const bool serial_solve = std::getenv("DSOPP_HIP_SERIAL_SOLVE") != nullptr;  // tuning aid: the step-by-step flow
    if (w->lm_mode == 0 && !sharded && !serial_solve && w->F() > 0) {
      // One enqueue of everything, host work under it: the LM loop leaves its result in pinned memory by itself, the closing
      // evaluation after a rejected last step is gated by the device-side flag, the covariance linearisation and the transfer
      // of its systems follow, then point statuses and the packed per-frame read-back — and while the GPU is busy with those the
      // host inverts the reduced system.  (Step by step the GPU idled for the host synchronisation behind the LM loop and for
      // the 0.1 ms of the pseudo-inverse: rocprofv3 kernel trace, scripts/trace_solve.py.)
      HostTimes ht_("solve (total)");
      w->sr.use();
      {
        HostTimes ht2_("solve: prepare");
        prepare(*w);  // (flushes the queued appends first)
      }
      HIP_CHECK(hipEventRecord(w->ev0, w->sr.stream));
      {
        HostTimes ht2_("solve: enqueue of the LM loop");
        fusedBegin(*w);
        lmSolveFusedEnqueue(*w);
      }
      {
        // closing problem.calculateEnergy() at the reverted state iff the last step was rejected (lmSolveFusedFinish does this
        // on the host's say-so; here the final control block's flag decides on the device)
        const int *flag = &w->fused_final_ctrl->need_final_setup;
        pairSetupKernel<<<1, kMaxFrames * kMaxFrames, 0, w->sr.stream>>>(w->d_frames.ptr, w->d_state.ptr, w->d_pc.ptr, w->F(), w->fej() ? 1 : 0, flag);
        SweepExtras ex;
        ex.run_flag = flag;
        launchSweep(*w, false, true, false, nullptr, false, 0.0, ex);
        w->pair_valid = false;
      }
      HIP_CHECK(hipEventRecord(w->ev1, w->sr.stream));
      w->solve_events_pending = true;
      {
        HostTimes ht2_("solve: enqueue of the re-linearisation");
        relinearize(*w);
      }
      bool want_state = false;
      {
        HostTimes ht2_("solve: enqueue of the covariance linearisation");
        if (w->opt.estimate_uncertainty) want_state = estimateUncertaintyEnqueue(*w, /*force_state=*/true);
      }
      {
        HostTimes ht2_("solve: enqueue of point statuses + frame export");
        updatePointStatusesDevice(*w);
        prefetchFrameUpdates(*w);
      }
      {
        HostTimes ht2_("solve: uncertainty on the host (under the device's work)");
        if (w->opt.estimate_uncertainty) estimateUncertaintyHost(*w, want_state);
      }
      {
        HostTimes ht2_("solve: final wait for the device");
        w->sr.sync();  // solve() is a blocking call: every result is in place when it returns
      }
      checkSolveLaunchFault(*w);
      e = w->h_ctrl->energy;
      it = w->h_ctrl->iteration;
      nv = w->h_ctrl->n_valid;
      w->export_valid = true;
      collectTimings(*w);
      w->begun = false;
      w->linearized = true;
      if (energy) *energy = e;
      if (iterations) *iterations = it;
      if (n_valid) *n_valid = nv;
      return;
    }
    runOptimize(w, e, it, nv);
    relinearize(*w);
    if (w->opt.estimate_uncertainty) estimateUncertainty(*w);
    updatePointStatuses(*w);
    const bool prefetch = !(w->allreduce && w->world > 1);
    if (prefetch) prefetchFrameUpdates(*w);
    w->sr.sync();  // solve() is a blocking call: every result is in place when it returns
    w->export_valid = prefetch;
    collectTimings(*w);
    w->begun = false;
    w->linearized = true;  // the last linearised system stays readable through get_system
    if (energy) *energy = e;
    if (iterations) *iterations = it;
    if (n_valid) *n_valid = nv;
  });
}

int dsopp_hip_window_get_frame_state(dsopp_hip_window *w, int32_t frame_id, double T0[7], double ab0[2], double eps[8], double step[8]) {
  return guarded([&] {
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    w->sr.use();
    flushAppends(*w);
    const int s = w->slotOf(frame_id);
    if (s < 0) fail(DSOPP_HIP_ERR_NOT_FOUND, "frame %d is not in the window", frame_id);
    downloadState(*w);
    if (T0) {
      Rigid T;
      for (int i = 0; i < 9; ++i) T.R[i] = w->hst.T0_R[s][i];
      for (int i = 0; i < 3; ++i) T.t[i] = w->hst.T0_t[s][i];
      rigidToParams(T, T0);
    }
    if (ab0) {
      ab0[0] = w->hst.ab0[s][0];
      ab0[1] = w->hst.ab0[s][1];
    }
    if (eps) std::memcpy(eps, w->hst.eps[s], sizeof(double) * kBlk);
    if (step) std::memcpy(step, w->hst.step[s], sizeof(double) * kBlk);
  });
}

int dsopp_hip_window_get_pose(dsopp_hip_window *w, int32_t frame_id, double T_world_agent[7], double affine_brightness[2]) {
  return guarded([&] {
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    w->sr.use();
    flushAppends(*w);
    const int s = w->slotOf(frame_id);
    if (s < 0) fail(DSOPP_HIP_ERR_NOT_FOUND, "frame %d is not in the window", frame_id);
    downloadState(*w);
    if (T_world_agent) rigidToParams(poseOf(*w, s), T_world_agent);
    if (affine_brightness) {
      affine_brightness[0] = w->hst.ab0[s][0] + w->hst.eps[s][6];
      affine_brightness[1] = w->hst.ab0[s][1] + w->hst.eps[s][7];
    }
  });
}

int dsopp_hip_window_num_landmarks(dsopp_hip_window *w, int32_t frame_id, int32_t *n) {
  return guarded([&] {
    if (!w || !n) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    *n = w->frameById(frame_id).n;
  });
}

int dsopp_hip_window_get_landmarks(dsopp_hip_window *w, int32_t frame_id, double *idepth, double *idepth_step, double *inv_hessian_idepth,
                                   double *b_idepth, double *relative_baseline, int32_t *n_inliers, uint8_t *flags_out, double *hpib) {
  return guarded([&] {
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    w->sr.use();
    flushAppends(*w);
    HostFrame &f = w->frameById(frame_id);
    hipStream_t st = w->sr.stream;
    const size_t n = static_cast<size_t>(f.n);
    if (idepth) f.idepth.download(idepth, n, 0, st);
    if (idepth_step) f.idepth_step.download(idepth_step, n, 0, st);
    if (inv_hessian_idepth) f.inv_hdd.download(inv_hessian_idepth, n, 0, st);
    if (b_idepth) f.b_d.download(b_idepth, n, 0, st);
    if (relative_baseline) f.relative_baseline.download(relative_baseline, n, 0, st);
    if (n_inliers) f.n_inliers.download(n_inliers, n, 0, st);
    if (flags_out) f.dflags.download(flags_out, n, 0, st);
    w->sr.sync();
    if (f.permuted()) {  // device order -> the caller's (HostFrame::to_internal)
      auto unpermute = [&](auto *a) {
        if (!a) return;
        std::vector<std::remove_reference_t<decltype(*a)>> tmp(a, a + n);
        for (size_t c = 0; c < n; ++c) a[c] = tmp[static_cast<size_t>(f.to_internal[c])];
      };
      unpermute(idepth);
      unpermute(idepth_step);
      unpermute(inv_hessian_idepth);
      unpermute(b_idepth);
      unpermute(relative_baseline);
      unpermute(n_inliers);
      unpermute(flags_out);
    }
    if (hpib && n) {
      const int F = w->F(), K = w->K();
      std::vector<double> blk(static_cast<size_t>(f.cap) * kUblk);
      for (int t = 0; t < F; ++t) {
        f.ublk.download(blk.data(), n * kUblk, static_cast<size_t>(t) * ublkPlane(f.cap), st);
        w->sr.sync();
        for (size_t i = 0; i < n; ++i) {
          const size_t p = static_cast<size_t>(f.internalOf(static_cast<int>(i)));
          for (int a = 0; a < kBlk; ++a) hpib[i * K + static_cast<size_t>(kBlk * t + a)] = blk[p * kUblk + static_cast<size_t>(a)];
        }
      }
    }
  });
}

int dsopp_hip_window_get_frame_update(dsopp_hip_window *w, int32_t frame_id, double *idepth, double *inv_hessian_idepth, double *relative_baseline,
                                      int32_t *n_inliers, uint8_t *flags_out, int32_t n_targets, const int32_t *target_ids, uint8_t *statuses) {
  return guarded([&] {
    if (!w || n_targets < 0 || n_targets > kMaxFrames || (n_targets && (!target_ids || !statuses))) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad argument");
    w->sr.use();
    flushAppends(*w);
    HostFrame &f = w->frameById(frame_id);
    const size_t n = static_cast<size_t>(f.n);
    if (n == 0) return;
    {
      // connections towards frames that are not in the window live on the host (set_connection): their rows are the statuses as declared
      std::vector<int32_t> held;
      bool any_pending = false;
      for (int t = 0; t < n_targets; ++t) {
        if (f.residuals.find(target_ids[t]) != f.residuals.end()) {
          held.push_back(target_ids[t]);
          continue;
        }
        const auto p = f.pending.find(target_ids[t]);
        if (p == f.pending.end()) fail(DSOPP_HIP_ERR_NOT_FOUND, "no connection %d -> %d", frame_id, target_ids[t]);
        if (p->second.size() != n) fail(DSOPP_HIP_ERR_STATE, "connection %d -> %d holds %zu residuals, frame has %d landmarks", frame_id, target_ids[t], p->second.size(), f.n);
        any_pending = true;
      }
      if (any_pending) {
        std::vector<uint8_t> rows(held.size() * n);
        const int rc = dsopp_hip_window_get_frame_update(w, frame_id, idepth, inv_hessian_idepth, relative_baseline, n_inliers, flags_out,
                                                         static_cast<int32_t>(held.size()), held.data(), rows.data());
        if (rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());
        size_t k = 0;
        for (int t = 0; t < n_targets; ++t) {
          const auto p = f.pending.find(target_ids[t]);
          if (p != f.pending.end() && f.residuals.find(target_ids[t]) == f.residuals.end())
            std::memcpy(statuses + static_cast<size_t>(t) * n, p->second.data(), n);
          else
            std::memcpy(statuses + static_cast<size_t>(t) * n, rows.data() + (k++) * n, n);
        }
        return;
      }
    }
    if (w->export_valid) {  // packed by the last solve(): a host copy
      for (const auto &e : w->export_entries) {
        if (e.frame_id != frame_id || e.n != f.n) continue;
        std::vector<int> row(static_cast<size_t>(n_targets), -1);
        bool all = true;
        for (int t = 0; t < n_targets && all; ++t) {
          auto it = std::find(e.target_ids.begin(), e.target_ids.end(), target_ids[t]);
          all = it != e.target_ids.end();
          if (all) row[static_cast<size_t>(t)] = static_cast<int>(it - e.target_ids.begin());
        }
        if (!all) break;
        const double *hd = static_cast<const double *>(w->h_update) + e.word_offset;
        const uint8_t *hb = reinterpret_cast<const uint8_t *>(hd + 4 * n);
        if (idepth) std::memcpy(idepth, hd, n * sizeof(double));
        if (inv_hessian_idepth) std::memcpy(inv_hessian_idepth, hd + n, n * sizeof(double));
        if (relative_baseline) std::memcpy(relative_baseline, hd + 2 * n, n * sizeof(double));
        if (n_inliers)
          for (size_t i = 0; i < n; ++i) n_inliers[i] = static_cast<int32_t>(hd[3 * n + i]);
        if (flags_out) std::memcpy(flags_out, hb, n);
        for (int t = 0; t < n_targets; ++t) std::memcpy(statuses + static_cast<size_t>(t) * n, hb + n * (1 + static_cast<size_t>(row[static_cast<size_t>(t)])), n);
        return;
      }
    }
    hipStream_t st = w->sr.stream;
    FrameExportArgs a;
    a.idepth = f.idepth.ptr;
    a.inv_hdd = f.inv_hdd.ptr;
    a.relative_baseline = f.relative_baseline.ptr;
    a.n_inliers = f.n_inliers.ptr;
    a.flags = f.dflags.ptr;
    a.to_internal = f.permuted() ? f.d_to_internal.ptr : nullptr;
    a.n = f.n;
    a.n_targets = n_targets;
    for (int t = 0; t < n_targets; ++t) {
      auto it = f.residuals.find(target_ids[t]);
      if (it == f.residuals.end()) fail(DSOPP_HIP_ERR_NOT_FOUND, "no connection %d -> %d", frame_id, target_ids[t]);
      if (it->second->n != f.n) fail(DSOPP_HIP_ERR_STATE, "connection %d -> %d holds %d residuals, frame has %d landmarks", frame_id, target_ids[t], it->second->n, f.n);
      a.status[t] = it->second->status.ptr;
    }
    const size_t n_bytes = static_cast<size_t>(1 + n_targets) * n;
    const size_t words = 4 * n + (n_bytes + 7) / 8;
    w->d_export.reserve(words, 0, st);
    a.out_d = w->d_export.ptr;
    a.out_b = reinterpret_cast<uint8_t *>(w->d_export.ptr + 4 * n);
    exportFrameKernel<<<static_cast<unsigned>((n + 255) / 256), 256, 0, st>>>(a);
    HIP_CHECK(hipGetLastError());
    growPinned(w->h_export, w->h_export_bytes, words * 8);
    HIP_CHECK(hipMemcpyAsync(w->h_export, w->d_export.ptr, words * 8, hipMemcpyDeviceToHost, st));
    w->sr.sync();
    const double *hd = static_cast<const double *>(w->h_export);
    const uint8_t *hb = reinterpret_cast<const uint8_t *>(hd + 4 * n);
    if (idepth) std::memcpy(idepth, hd, n * sizeof(double));
    if (inv_hessian_idepth) std::memcpy(inv_hessian_idepth, hd + n, n * sizeof(double));
    if (relative_baseline) std::memcpy(relative_baseline, hd + 2 * n, n * sizeof(double));
    if (n_inliers)
      for (size_t i = 0; i < n; ++i) n_inliers[i] = static_cast<int32_t>(hd[3 * n + i]);
    if (flags_out) std::memcpy(flags_out, hb, n);
    if (n_targets) std::memcpy(statuses, hb + n, static_cast<size_t>(n_targets) * n);
  });
}

int dsopp_hip_window_get_residuals(dsopp_hip_window *w, int32_t reference_id, int32_t target_id, int32_t n, uint8_t *status,
                                   uint8_t *candidate, double *energy) {
  return guarded([&] {
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    w->sr.use();
    flushAppends(*w);
    HostFrame &f = w->frameById(reference_id);
    auto it = f.residuals.find(target_id);
    if (it == f.residuals.end()) fail(DSOPP_HIP_ERR_NOT_FOUND, "no connection %d -> %d", reference_id, target_id);
    ResidualTable &rt = *it->second;
    if (n > rt.n) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "connection holds %d residuals, %d requested", rt.n, n);
    hipStream_t st = w->sr.stream;
    if (f.permuted()) {
      // the caller's first n landmarks sit anywhere among the device's first rt.n (n need not end a batch): whole list, then picked
      const size_t m = static_cast<size_t>(rt.n);
      std::vector<uint8_t> s_st(status ? m : 0), s_ca(candidate ? m : 0);
      std::vector<double> s_en(energy ? m : 0);
      if (status) rt.status.download(s_st.data(), m, 0, st);
      if (candidate) rt.cand.download(s_ca.data(), m, 0, st);
      if (energy) rt.energy.download(s_en.data(), m, 0, st);
      w->sr.sync();
      for (int i = 0; i < n; ++i) {
        const size_t p = static_cast<size_t>(f.to_internal[static_cast<size_t>(i)]);
        if (status) status[i] = s_st[p];
        if (candidate) candidate[i] = s_ca[p];
        if (energy) energy[i] = s_en[p];
      }
      return;
    }
    if (status) rt.status.download(status, static_cast<size_t>(n), 0, st);
    if (candidate) rt.cand.download(candidate, static_cast<size_t>(n), 0, st);
    if (energy) rt.energy.download(energy, static_cast<size_t>(n), 0, st);
    w->sr.sync();
  });
}

int dsopp_hip_window_get_marginalized(dsopp_hip_window *w, double *H, double *b, double *energy, int32_t *size) {
  return guarded([&] {
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    if (H) std::memcpy(H, w->Hm.data(), w->Hm.size() * sizeof(double));
    if (b) std::memcpy(b, w->bm.data(), w->bm.size() * sizeof(double));
    if (energy) *energy = w->energy_marginalized;
    if (size) *size = w->marg_size;
  });
}

int dsopp_hip_window_get_covariance(dsopp_hip_window *w, int32_t reference_id, int32_t target_id, double cov[36]) {
  return guarded([&] {
    if (!w || !cov) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    HostFrame &f = w->frameById(reference_id);
    auto it = f.covariance.find(target_id);
    if (it == f.covariance.end()) fail(DSOPP_HIP_ERR_NOT_FOUND, "no covariance %d -> %d", reference_id, target_id);
    std::memcpy(cov, it->second.data(), 36 * sizeof(double));
  });
}

namespace {
/** fills `maps` (allocated for `levels` levels of the newest keyframe's size) from the window's landmarks */
void fillReferenceDepthMaps(dsopp_hip_window *w, dsopp_hip_depth_maps *maps) {
  HostTimes ht_("fillReferenceDepthMaps");
  hipStream_t st = w->sr.stream;
  const int levels = maps->levels;
  const int F = w->F(), newest = F - 1;
  const HostFrame &fn = *w->frames[static_cast<size_t>(newest)];
  const LevelView lv = fn.pyramid->view(fn.level);
  // temporaries (the undilated maps) live with the window, all levels in one buffer — level l: [idepth sums | weights] — so that the planes
  // the splat accumulates into are cleared by ONE fill and completed across landmark shards by ONE collective
  DepthMapLevels L;
  std::memset(&L, 0, sizeof(L));
  L.levels = levels;
  size_t total = 0, first[kDepthMapLevels];
  int rows = 0;
  for (int l = 0; l < levels; ++l) {
    first[l] = total;
    L.width[l] = maps->width[static_cast<size_t>(l)];
    L.height[l] = maps->height[static_cast<size_t>(l)];
    L.first_row[l] = rows;
    rows += L.height[l];
    total += 2 * static_cast<size_t>(L.width[l]) * L.height[l];
    maps->points[static_cast<size_t>(l)].n = -1;  // cached reference points of an earlier fill are stale
    if (l && (L.width[l] != L.width[l - 1] / 2 || L.height[l] != L.height[l - 1] / 2)) fail(DSOPP_HIP_ERR_STATE, "depth-map level %d is not half of level %d", l, l - 1);
  }
  L.first_row[levels] = rows;
  w->dm_tmp.reserve(total, 0, st);
  for (int l = 0; l < levels; ++l) {
    L.id[l] = w->dm_tmp.ptr + first[l];
    L.w[l] = L.id[l] + static_cast<size_t>(L.width[l]) * L.height[l];
    L.out_id[l] = maps->idepth_sum[static_cast<size_t>(l)].ptr;
    L.out_w[l] = maps->weight[static_cast<size_t>(l)].ptr;
  }
  const size_t n0 = static_cast<size_t>(L.width[0]) * L.height[0];
  HIP_CHECK(hipMemsetAsync(L.id[0], 0, 2 * n0 * sizeof(double), st));  // the splat accumulates; every other plane is overwritten
  // fillFineDepthMap — :18-59 (into the temporaries; the dilation writes the final planes): every older keyframe in one launch
  const Rigid T_newest_inv = rigidInverse(poseOf(*w, newest));
  SplatBatch batch;
  int n_sources = 0, max_n = 0;
  for (int f = 0; f < newest; ++f) {
    const HostFrame &fr = *w->frames[static_cast<size_t>(f)];
    auto it = fr.residuals.find(fn.id);
    if (fr.n == 0 || it == fr.residuals.end() || it->second->n == 0) continue;
    const Rigid T = rigidMul(T_newest_inv, poseOf(*w, f));  // t_t_r, :28
    SplatArgs &a = batch.src[n_sources++];
    const double ifx = 1.0 / fr.intr[0], ify = 1.0 / fr.intr[1];
    const double k02 = -fr.intr[2] / fr.intr[0], k12 = -fr.intr[3] / fr.intr[1];
    double U[12];
    for (int i = 0; i < 3; ++i) {
      U[4 * i + 0] = T.R[3 * i + 0] * ifx;
      U[4 * i + 1] = T.R[3 * i + 1] * ify;
      U[4 * i + 2] = T.R[3 * i + 0] * k02 + T.R[3 * i + 1] * k12 + T.R[3 * i + 2];
      U[4 * i + 3] = T.t[i];
    }
    for (int j = 0; j < 4; ++j) {
      a.M[0 + j] = fn.intr[0] * U[0 + j] + fn.intr[2] * U[8 + j];
      a.M[4 + j] = fn.intr[1] * U[4 + j] + fn.intr[3] * U[8 + j];
      a.M[8 + j] = U[8 + j];
    }
    a.Tz[0] = T.R[6];
    a.Tz[1] = T.R[7];
    a.Tz[2] = T.R[8];
    a.Tz[3] = T.t[2];
    a.fx = fr.intr[0];
    a.fy = fr.intr[1];
    a.cx = fr.intr[2];
    a.cy = fr.intr[3];
    a.width = lv.width;
    a.height = lv.height;
    a.n = std::min(fr.n, it->second->n);
    a.use_variance = w->opt.estimate_uncertainty ? 1 : 0;
    a.uv = fr.uv.ptr;
    a.idepth = fr.idepth.ptr;
    a.inv_hdd = fr.inv_hdd.ptr;
    a.flags = fr.dflags.ptr;
    a.status = it->second->status.ptr;
    max_n = std::max(max_n, a.n);
  }
  if (n_sources) splatDepthMapsKernel<<<dim3(static_cast<unsigned>((max_n + 255) / 256), static_cast<unsigned>(n_sources)), 256, 0, st>>>(batch, L.id[0], L.w[0]);
  if (w->allreduce && w->world > 1) {
    // landmark shards: every shard splatted its own landmarks; the level-0 planes are sums over landmarks, so one collective over both
    // planes completes them on every shard (pooling and dilation below are then replicated work on the full maps)
    allreduceIfNeeded(*w, L.id[0], 2 * n0);
  }
  // fillCoarseDepthMaps — :70-88: all coarser levels from the level-0 tiles in one launch
  if (levels > 1) poolDepthMapsKernel<<<dim3(static_cast<unsigned>((L.width[0] + 15) / 16), static_cast<unsigned>((L.height[0] + 15) / 16)), 256, 0, st>>>(L);
  // dilateDepthMaps — :90-122 (after ALL levels were pooled from the undilated maps, as in the reference): all levels in one launch
  dilateDepthMapsKernel<<<dim3(static_cast<unsigned>((L.width[0] + 127) / 128), static_cast<unsigned>(rows)), 128, 0, st>>>(L);
  HIP_CHECK(hipGetLastError());
  w->sr.sync();  // consumers may live on other streams
}
}  // namespace

int dsopp_hip_window_create_reference_depth_maps(dsopp_hip_window *w, int32_t levels, dsopp_hip_depth_maps **out) {
  return guarded([&] {
    if (!w || !out) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (w->frames.empty()) fail(DSOPP_HIP_ERR_STATE, "the window holds no keyframe");
    if (levels < 1 || levels > 5) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "levels must be in [1, 5]");
    prepare(*w);  // topology / state on the device, host mirror of the poses current
    hipStream_t st = w->sr.stream;
    const HostFrame &fn = *w->frames.back();
    const LevelView lv = fn.pyramid->view(fn.level);
    auto maps = std::make_unique<dsopp_hip_depth_maps>();
    maps->sr = w->sr;
    maps->sr.owned = false;
    maps->levels = levels;
    maps->points.resize(static_cast<size_t>(levels));
    // initDepthMaps — create_depth_maps.cpp:62-68: one map per pyramid level of the newest keyframe
    int lw = lv.width, lh = lv.height;
    for (int l = 0; l < levels; ++l) {
      if (lw < 3 || lh < 3) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "level %d of a %d x %d image is too small", l, lv.width, lv.height);
      maps->width.push_back(lw);
      maps->height.push_back(lh);
      maps->idepth_sum.emplace_back();
      maps->weight.emplace_back();
      const size_t n = static_cast<size_t>(lw) * lh;
      maps->idepth_sum.back().reserve(n, 0, st);
      maps->weight.back().reserve(n, 0, st);
      lw /= 2;
      lh /= 2;
    }
    fillReferenceDepthMaps(w, maps.get());
    maps->owner = w;
    w->live_maps.push_back(maps.get());
    *out = maps.release();
  });
}

int dsopp_hip_window_refill_reference_depth_maps(dsopp_hip_window *w, dsopp_hip_depth_maps *maps) {
  return guarded([&] {
    if (!w || !maps) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (w->frames.empty()) fail(DSOPP_HIP_ERR_STATE, "the window holds no keyframe");
    const HostFrame &fn = *w->frames.back();
    const LevelView lv = fn.pyramid->view(fn.level);
    if (maps->levels < 1 || maps->width[0] != lv.width || maps->height[0] != lv.height)
      fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "the maps are %d x %d, the newest keyframe is %d x %d", maps->levels ? maps->width[0] : 0,
           maps->levels ? maps->height[0] : 0, lv.width, lv.height);
    if (maps->sr.device != w->sr.device) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "depth maps live on another device");
    prepare(*w);
    fillReferenceDepthMaps(w, maps);
  });
}

void dsopp_hip_depth_maps_destroy(dsopp_hip_depth_maps *m) {
  if (!m) return;
  (void)hipSetDevice(m->sr.device);
  if (m->sr.stream) (void)hipStreamSynchronize(m->sr.stream);
  else (void)hipDeviceSynchronize();
  if (m->owner) {
    auto &live = m->owner->live_maps;
    live.erase(std::remove(live.begin(), live.end(), m), live.end());
  }
  delete m;
}

int dsopp_hip_depth_maps_level_size(const dsopp_hip_depth_maps *m, int32_t level, int32_t *width, int32_t *height) {
  return guarded([&] {
    if (!m || level < 0 || level >= m->levels) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad argument");
    if (width) *width = m->width[static_cast<size_t>(level)];
    if (height) *height = m->height[static_cast<size_t>(level)];
  });
}

int dsopp_hip_depth_maps_get_level(const dsopp_hip_depth_maps *m, int32_t level, double *idepth_sum, double *weight) {
  return guarded([&] {
    if (!m || level < 0 || level >= m->levels || !idepth_sum || !weight) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad argument");
    m->sr.use();
    const size_t n = static_cast<size_t>(m->width[static_cast<size_t>(level)]) * m->height[static_cast<size_t>(level)];
    m->idepth_sum[static_cast<size_t>(level)].download(idepth_sum, n, 0, m->sr.stream);
    m->weight[static_cast<size_t>(level)].download(weight, n, 0, m->sr.stream);
    m->sr.sync();
  });
}

int dsopp_hip_window_set_allreduce(dsopp_hip_window *w, dsopp_hip_allreduce_fn fn, void *user, int rank, int world_size) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    if (fn && (world_size < 1 || rank < 0 || rank >= world_size)) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad rank %d / world %d", rank, world_size);
    w->allreduce = fn;
    w->allreduce_user = user;
    w->rank = fn ? rank : 0;
    w->world = fn ? world_size : 1;
  });
}

int dsopp_hip_window_set_deterministic(dsopp_hip_window *w, int enable) {
  return guarded([&] {
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    w->deterministic = enable != 0;
  });
}

int dsopp_hip_window_set_comm(dsopp_hip_window *w, dsopp_hip_comm *comm) {
  int rank = 0, world = 1;
  if (comm) {
    const int rc = dsopp_hip_comm_rank(comm, &rank, &world);
    if (rc != DSOPP_HIP_OK) return rc;
  }
  return dsopp_hip_window_set_allreduce(w, comm ? &dsopp_hip::nativeAllreduce : nullptr, comm, rank, world);
}

/* tuning aid (not declared in the public header): stamps of the sweep kernel (workgroup grid/2) */
int dsopp_hip_debug_sweep_stamps(dsopp_hip_window *w, int lin, long long *out16) {
  return guarded([&] {
    if (!kStamps) fail(DSOPP_HIP_ERR_STATE, "phase stamps are not compiled in (build with -DDSOPP_HIP_STAMPS)");
    if (!w->dbg_sweep) {
      HIP_CHECK(hipMalloc(&w->dbg_sweep, 16 * sizeof(long long)));
    } else {
      HIP_CHECK(hipMemcpy(out16, w->dbg_sweep, 16 * sizeof(long long), hipMemcpyDeviceToHost));
    }
    HIP_CHECK(hipMemset(w->dbg_sweep, 0, 16 * sizeof(long long)));
    const long long big = 0x7fffffffffffffffLL;
    HIP_CHECK(hipMemcpy(w->dbg_sweep + 8, &big, sizeof(long long), hipMemcpyHostToDevice));
    w->dbg_sweep_lin = lin != 0;
  });
}

/* tuning aid (not declared in the public header): wall_clock64 stamps of the solve kernel's phases (100 MHz ticks) */
int dsopp_hip_debug_solve_stamps(dsopp_hip_window *w, long long *out8) {
  return guarded([&] {
    if (!kStamps) fail(DSOPP_HIP_ERR_STATE, "phase stamps are not compiled in (build with -DDSOPP_HIP_STAMPS)");
    if (!w->dbg_stamps) {
      HIP_CHECK(hipMalloc(&w->dbg_stamps, 64 * sizeof(long long)));
      HIP_CHECK(hipMemset(w->dbg_stamps, 0, 64 * sizeof(long long)));
    }
    HIP_CHECK(hipMemcpy(out8, w->dbg_stamps, 64 * sizeof(long long), hipMemcpyDeviceToHost));
  });
}

int dsopp_hip_window_time_kernel(dsopp_hip_window *w, int kernel_class, int repeats, double *avg_us) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w || !avg_us || repeats < 1) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad argument");
    w->sr.use();
    flushAppends(*w);
    if (!w->begun) stageBegin(*w);
    const bool saved = w->profiling;
    w->profiling = false;
    ensurePairConstants(*w);
    auto once = [&] {
      switch (kernel_class) {
        case DSOPP_HIP_KERNEL_SWEEP_LINEARIZE: launchSweep(*w, true, true, false); break;
        case DSOPP_HIP_KERNEL_SWEEP_ENERGY: launchSweep(*w, false, true, false); break;
        case DSOPP_HIP_KERNEL_SWEEP_LINEARIZE_LOOP: {
          // exactly the launch of lmSolveFusedEnqueue's rounds >= 1: linearises at the candidate state whose inverse-depth steps the
          // solve launch in front of it left (DSOPP_HIP_K3_BACKSUB=0: the round-3 flow — the sweep reads the Schur rows / pose step of
          // the previous round and back-substitutes itself)
          static const int k3_env = std::getenv("DSOPP_HIP_K3_BACKSUB") ? std::atoi(std::getenv("DSOPP_HIP_K3_BACKSUB")) : 1;
          SweepExtras ex;
          ex.ublk_read = 0;
          ex.ublk_write = 1;
          ex.fused_lin_backsub = k3_env == 0;
          ex.external_backsub = k3_env != 0;
          launchSweep(*w, true, true, false, nullptr, k3_env == 0, 1e-5, ex);
          break;
        }
        case DSOPP_HIP_KERNEL_SCHUR: {  // as the fused loop launches it (combined system), without the decision prologue
          if (w->twoStage()) {
            launchTwoStage(*w, nullptr, 0, 1e-5);
            break;
          }
          FusedReduce fr;
          fr.ublk_parity = 0;
          fr.ctrl_out = nullptr;
          fr.combined = true;
          fr.comb_lambda = 1e-5;
          launchReduceSchur(*w, false, nullptr, &fr, ReduceMode::kAccumulateOnly);  // (accumulates on top of the previous repeat: timing only)
          break;
        }
        case DSOPP_HIP_KERNEL_ASSEMBLE_SOLVE: launchSolveCombined(*w, 1e-5, nullptr); break;
        default: fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "kernel class %d cannot be timed in isolation", kernel_class);
      }
    };
    const bool needs_step = kernel_class == DSOPP_HIP_KERNEL_ASSEMBLE_SOLVE || kernel_class == DSOPP_HIP_KERNEL_SWEEP_LINEARIZE_LOOP;
    if (kernel_class == DSOPP_HIP_KERNEL_SCHUR || needs_step) launchSweep(*w, true, true, false);
    if (needs_step) {
      FusedReduce fr;
      fr.ublk_parity = 0;
      fr.ctrl_out = nullptr;
      fr.combined = true;
      fr.comb_lambda = 1e-5;
      HIP_CHECK(hipMemsetAsync(w->d_reduce.ptr, 0, (w->combCount() + 4) * sizeof(double), w->sr.stream));
      launchReduceSchur(*w, false, nullptr, &fr, ReduceMode::kAccumulateOnly);
    }
    if (kernel_class == DSOPP_HIP_KERNEL_SWEEP_LINEARIZE_LOOP) launchSolveCombined(*w, 1e-5, nullptr);  // a pose step to back-substitute
    once();  // warm
    HIP_CHECK(hipEventRecord(w->ev0, w->sr.stream));
    for (int i = 0; i < repeats; ++i) once();
    HIP_CHECK(hipEventRecord(w->ev1, w->sr.stream));
    HIP_CHECK(hipEventSynchronize(w->ev1));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, w->ev0, w->ev1));
    *avg_us = static_cast<double>(ms) * 1e3 / repeats;
    w->profiling = saved;
    // the repeated solve launches moved the candidate step: drop it so the window state is unchanged
    if (needs_step) {
      stageAccept(*w, false);
      w->pair_valid = false;
    }
    w->linearized = false;
  });
}

int dsopp_hip_window_set_lm_mode(dsopp_hip_window *w, int host_driven) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    w->lm_mode = host_driven;
  });
}

namespace {
/** optimize_repeated with the host running ahead: the solves of a batch are enqueued back to back (restore -> begin -> the fused
 *  LM rounds, every solve's closing kernel leaving its control block in its own device slot) and their results are fetched with
 *  one copy and one synchronisation per batch, so the stream never drains between solves.  Sequentially every solve ends with a host synchronisation and the next one starts with ~25 us of
 *  enqueue latency in front of its first kernels: ~60 us of an idle GPU per 7 iterations (rocprofv3 kernel trace), which belongs to
 *  this helper's bookkeeping, not to a Gauss-Newton iteration.  A solve never runs more iterations than its budget, so the total
 *  cannot overshoot; solves that stop early are made up for by further ones, exactly as in the sequential loop. */
void optimizeRepeatedPipelined(dsopp_hip_window &w, int target, int &done, double &energy) {
  w.sr.use();
  flushAppends(w);
  hipStream_t st = w.sr.stream;
  const int configured = w.opt.max_iterations;
  constexpr int kSlots = 32;  // solves per batch: their results are fetched with ONE copy and ONE synchronisation
  w.d_results.reserve(kSlots, 0, st);
  if (!w.h_results) HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&w.h_results), kSlots * sizeof(LmControl), hipHostMallocDefault));
  done = 0;
  bool stalled = false, last_needs_closing = false;
  auto cleanUp = [&] {
    w.opt.max_iterations = configured;
    w.result_device = nullptr;
    w.restore_in_begin = false;
  };
  try {
    while (done < target && !stalled) {
      int n = 0, planned = 0;
      while (n < kSlots && done + planned < target) {
        const int budget = std::min(configured, target - done - planned);
        restoreHostSide(w);
        w.opt.max_iterations = budget;
        // nothing is enqueued between here and the solve's opening kernel when the solve takes the first-estimate path and no
        // upload is pending: the device half of the restore then rides in that kernel (one launch less per solve)
        const bool ride = w.fej() && budget > 0 && !w.topology_dirty && !w.marg_dirty && w.d_state.ptr != nullptr;
        if (!ride) launchRestore(w);
        prepare(w);
        fusedBegin(w);
        w.restore_in_begin = ride;
        w.result_device = w.d_results.ptr + n;  // the closing kernel of this solve leaves its control block here
        lmSolveFusedEnqueue(w);
        planned += budget;
        ++n;
      }
      w.result_device = nullptr;
      HIP_CHECK(hipMemcpyAsync(w.h_results, w.d_results.ptr, static_cast<size_t>(n) * sizeof(LmControl), hipMemcpyDeviceToHost, st));
      w.sr.sync();
      checkSolveLaunchFault(w);
      for (int k = 0; k < n; ++k) {
        const LmControl &r = w.h_results[k];
        if (r.iteration <= 0) stalled = true;  // no progress: leave (iterations_done < target)
        done += r.iteration;
        energy = r.energy;
        last_needs_closing = r.need_final_setup != 0;
      }
    }
    if (last_needs_closing) {
      // the last solve ended on a rejected step: closing evaluation at the reverted state, as lmSolveFusedFinish does
      ensurePairConstants(w);
      launchSweep(w, false, true, false);
      HIP_CHECK(hipGetLastError());
      w.sr.sync();
    }
  } catch (...) {
    (void)hipStreamSynchronize(st);
    cleanUp();
    throw;
  }
  cleanUp();
  w.begun = false;
}
}  // namespace

int dsopp_hip_window_optimize_repeated(dsopp_hip_window *w, int32_t iterations_target, int32_t *iterations_done, double *last_energy) {
  if (!w || iterations_target < 0) return dsopp_hip_window_restore(nullptr);  // reports the invalid argument
  static const bool no_pipeline = std::getenv("DSOPP_HIP_NO_PIPELINE") != nullptr;  // tuning aid: one solve at a time
  // (landmark-sharded windows too: their collectives are ordered on the stream, and every rank plans the same solves because the
  // decisions are taken from all-reduced sums)
  if (!no_pipeline && w->lm_mode == 0 && w->opt.force_accept && w->F() > 0 && !w->async_pending) {
    int done = 0;
    double e = 0;
    const int rc = guarded([&] {
      w->export_valid = false;
      optimizeRepeatedPipelined(*w, iterations_target, done, e);
    });
    if (iterations_done) *iterations_done = done;
    if (last_energy) *last_energy = e;
    return rc;
  }
  const int configured = w->opt.max_iterations;
  int done = 0, rc = DSOPP_HIP_OK;
  double e = 0;
  while (rc == DSOPP_HIP_OK && done < iterations_target) {
    rc = dsopp_hip_window_restore(w);
    if (rc != DSOPP_HIP_OK) break;
    rc = dsopp_hip_window_set_max_iterations(w, std::min(configured, iterations_target - done));
    if (rc != DSOPP_HIP_OK) break;
    int32_t it = 0;
    rc = dsopp_hip_window_optimize(w, &e, &it, nullptr);
    if (rc == DSOPP_HIP_OK && it <= 0) break;  // no progress: leave the loop, the caller sees iterations_done < target
    done += it;
  }
  const int rc2 = dsopp_hip_window_set_max_iterations(w, configured);
  if (iterations_done) *iterations_done = done;
  if (last_energy) *last_energy = e;
  return rc != DSOPP_HIP_OK ? rc : rc2;
}

int dsopp_hip_window_set_max_iterations(dsopp_hip_window *w, int32_t max_iterations) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w || max_iterations < 0) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad argument");
    w->opt.max_iterations = max_iterations;
  });
}

int dsopp_hip_window_snapshot(dsopp_hip_window *w) {
  return guarded([&] {
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    w->sr.use();
    flushAppends(*w);
    prepare(*w);
    downloadState(*w);
    hipStream_t st = w->sr.stream;
    for (auto &fp : w->frames) {
      HostFrame &f = *fp;
      f.snap_idepth.reserve(static_cast<size_t>(std::max(f.cap, 1)), 0, st);
      f.snap_flags.reserve(static_cast<size_t>(std::max(f.cap, 1)), 0, st);
      if (f.n) {
        HIP_CHECK(hipMemcpyAsync(f.snap_idepth.ptr, f.idepth.ptr, static_cast<size_t>(f.n) * sizeof(double), hipMemcpyDeviceToDevice, st));
        HIP_CHECK(hipMemcpyAsync(f.snap_flags.ptr, f.dflags.ptr, static_cast<size_t>(f.n), hipMemcpyDeviceToDevice, st));
      }
      f.snap_n = f.n;
      for (auto &kv : f.residuals) {
        ResidualTable &rt = *kv.second;
        rt.snap_status.reserve(static_cast<size_t>(std::max(f.cap, 1)), 0, st);
        if (rt.n) HIP_CHECK(hipMemcpyAsync(rt.snap_status.ptr, rt.status.ptr, static_cast<size_t>(rt.n), hipMemcpyDeviceToDevice, st));
        rt.snap_n = rt.n;
      }
    }
    w->d_state_snap.reserve(1, 0, st);
    HIP_CHECK(hipMemcpyAsync(w->d_state_snap.ptr, w->d_state.ptr, sizeof(WindowState), hipMemcpyDeviceToDevice, st));
    w->snap_state = w->hst;
    w->snap_F = w->F();
    w->snap_valid = true;
    w->sr.sync();
    w->topology_dirty = true;  // the frame table must carry the snapshot pointers
    syncTopology(*w);
    flushAppends(*w);  // (its tables are queued uploads)
  });
}

int dsopp_hip_window_restore(dsopp_hip_window *w) {
  return guarded([&] {
    if (w) w->export_valid = false;
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    restoreHostSide(*w);
    // one enqueue, no host synchronisation: one kernel over all landmarks + the frame-state block
    launchRestore(*w);
  });
}

int dsopp_hip_window_set_profiling(dsopp_hip_window *w, int enable) {
  return guarded([&] {
    if (!w) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window");
    collectTimings(*w);
    w->profiling = enable != 0;
    for (int i = 0; i < DSOPP_HIP_NUM_KERNEL_CLASSES; ++i) {
      w->prof_ms[i] = 0;
      w->prof_count[i] = 0;
    }
  });
}

int dsopp_hip_window_get_profile(dsopp_hip_window *w, int kernel_class, double *total_ms, int64_t *launches) {
  return guarded([&] {
    if (!w || kernel_class < 0 || kernel_class >= DSOPP_HIP_NUM_KERNEL_CLASSES) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad kernel class");
    collectTimings(*w);
    if (total_ms) *total_ms = w->prof_ms[kernel_class];
    if (launches) *launches = w->prof_count[kernel_class];
  });
}

const char *dsopp_hip_kernel_class_name(int kernel_class) {
  static const char *names[DSOPP_HIP_NUM_KERNEL_CLASSES] = {"pair_setup", "fej", "sweep_linearize", "sweep_energy", "schur",
                                                            "assemble", "assemble_solve", "backsub", "energy_reduce", "accept_decide",
                                                            "sweep_linearize_loop"};
  return (kernel_class >= 0 && kernel_class < DSOPP_HIP_NUM_KERNEL_CLASSES) ? names[kernel_class] : "?";
}

int dsopp_hip_window_last_solve_ms(dsopp_hip_window *w, float *ms) {
  return guarded([&] {
    if (!w || !ms) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (w->solve_events_pending) {
      w->sr.use();
    flushAppends(*w);
      HIP_CHECK(hipEventSynchronize(w->ev1));
      HIP_CHECK(hipEventElapsedTime(&w->last_solve_ms, w->ev0, w->ev1));
      w->solve_events_pending = false;
    }
    *ms = w->last_solve_ms;
  });
}

}  // extern "C"

int dsopp_hip_depth_maps_mean_square_optical_flow(const dsopp_hip_depth_maps *m, int32_t level, const double intrinsics[4], int32_t n_transforms,
                                                  const double *T_target_reference, double *flow) {
  return guarded([&] {
    if (!m || !intrinsics || !T_target_reference || !flow) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (level < 0 || level >= m->levels) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "level %d out of range (%d levels)", level, m->levels);
    if (n_transforms < 1 || n_transforms > kMaxFlowTransforms) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "n_transforms must be in [1, %d]", kMaxFlowTransforms);
    m->sr.use();
    hipStream_t st = m->sr.stream;
    FlowArgs a;
    std::memset(&a, 0, sizeof(a));
    const double fx = intrinsics[0], fy = intrinsics[1], cx = intrinsics[2], cy = intrinsics[3];
    for (int t = 0; t < n_transforms; ++t) {
      const Rigid T = rigidFromParams(T_target_reference + 7 * t);
      const double ifx = 1.0 / fx, ify = 1.0 / fy, k02 = -cx / fx, k12 = -cy / fy;
      double U[12];
      for (int i = 0; i < 3; ++i) {
        U[4 * i + 0] = T.R[3 * i + 0] * ifx;
        U[4 * i + 1] = T.R[3 * i + 1] * ify;
        U[4 * i + 2] = T.R[3 * i + 0] * k02 + T.R[3 * i + 1] * k12 + T.R[3 * i + 2];
        U[4 * i + 3] = T.t[i];
      }
      for (int j = 0; j < 4; ++j) {
        a.M[t][0 + j] = fx * U[0 + j] + cx * U[8 + j];
        a.M[t][4 + j] = fy * U[4 + j] + cy * U[8 + j];
        a.M[t][8 + j] = U[8 + j];
      }
    }
    a.cx = cx;
    a.cy = cy;
    a.ifx = 1 / fx;
    a.ify = 1 / fy;
    a.width = m->width[static_cast<size_t>(level)];
    a.height = m->height[static_cast<size_t>(level)];
    a.n_transforms = n_transforms;
    const dim3 grid(static_cast<unsigned>((a.width + 255) / 256), static_cast<unsigned>((a.height + kFlowRows - 1) / kFlowRows));
    const size_t n_blocks = static_cast<size_t>(grid.x) * grid.y;
    // the level's reference points, when the tracker has extracted them from THESE maps (estimatePose does, in front of the flows of the same
    // frame; a refill marks them stale): the list holds exactly the pixels the dense pass keeps.  DSOPP_HIP_FLOW_DENSE=1: always the dense pass
    static const bool dense_only = std::getenv("DSOPP_HIP_FLOW_DENSE") != nullptr;
    dsopp_hip_depth_maps::LevelPoints &pts = m->points[static_cast<size_t>(level)];
    const bool by_points = !dense_only && pts.n >= 0;
    const size_t point_blocks = static_cast<size_t>(std::max(1, (pts.n + kFlowPointThreads - 1) / kFlowPointThreads));
    // scratch: [ticket of the point pass (zero-filled with the buffer, re-armed by the kernel) | 7 unused | partials per workgroup]
    m->flow_scratch.reserve(8 + std::max(n_blocks, by_points ? point_blocks : 0) * 2 * kMaxFlowTransforms, 0, st);
    if (!m->h_flow) HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&m->h_flow), 8 * sizeof(double), hipHostMallocDefault));
    double *out = m->h_flow, *partials = m->flow_scratch.ptr + 8;
    if (by_points) {
      pts.orderBehind(st);
      opticalFlowPointsKernel<<<static_cast<unsigned>(point_blocks), kFlowPointThreads, 0, st>>>(pts.u.ptr, pts.v.ptr, pts.idepth.ptr, pts.n, a, partials,
                                                                                                 reinterpret_cast<unsigned *>(m->flow_scratch.ptr), out);
    } else {
      opticalFlowPartialsKernel<<<grid, 256, 0, st>>>(m->idepth_sum[static_cast<size_t>(level)].ptr, m->weight[static_cast<size_t>(level)].ptr, a, partials);
      opticalFlowFinishKernel<<<1, 256, 0, st>>>(partials, static_cast<int>(n_blocks), n_transforms, out);
    }
    HIP_CHECK(hipGetLastError());
    m->sr.sync();  // (the closing workgroup's stores to the pinned result are visible once its kernel has completed)
    std::memcpy(flow, out, sizeof(double) * static_cast<size_t>(n_transforms));
  });
}

// ---------------------------------------------------------------------------------------------------------------------
// activation of immature landmarks (row f-3)
// ---------------------------------------------------------------------------------------------------------------------
int dsopp_hip_window_activate_landmarks(dsopp_hip_window *w, int32_t n_keyframes, const int32_t *frame_ids,
                                        dsopp_hip_immature_set *const *immature, const dsopp_hip_pyramid *newest_pyramid,
                                        const double T_world_newest[7], double exposure_newest, const double affine_newest[2],
                                        int32_t number_of_desired_points, double *min_distance_to_neighbor, int32_t refine,
                                        double sigma_huber_loss, uint8_t *const *activation_status, double *const *idepth,
                                        dsopp_hip_activation_result *result) {
  return guarded([&] {
    HostTimes ht_("activate_landmarks (total)");
    if (!w || !frame_ids || !immature || !newest_pyramid || !T_world_newest || !affine_newest || !min_distance_to_neighbor)
      fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (n_keyframes < 1 || n_keyframes > kMaxFrames - 1) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "n_keyframes must be in [1, %d]", kMaxFrames - 1);
    if (number_of_desired_points < 0) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "number_of_desired_points < 0");
    w->sr.use();
    prepare(*w);  // queued appends flushed, device state current, host mirror of the poses current
    hipStream_t st = w->sr.stream;
    const int F = n_keyframes + 1;
    std::vector<int> slots(static_cast<size_t>(n_keyframes));
    for (int k = 0; k < n_keyframes; ++k) {
      slots[static_cast<size_t>(k)] = w->slotOf(frame_ids[k]);
      if (slots[static_cast<size_t>(k)] < 0) fail(DSOPP_HIP_ERR_NOT_FOUND, "frame %d is not in the window", frame_ids[k]);
      if (k && slots[static_cast<size_t>(k)] <= slots[static_cast<size_t>(k - 1)])
        fail(DSOPP_HIP_ERR_ORDER, "keyframes must be listed oldest first (frame %d)", frame_ids[k]);
    }
    const HostFrame &f0 = *w->frames[static_cast<size_t>(slots[0])];
    const dsopp_hip_pyramid *p0 = f0.pyramid;
    if (f0.level != 0) fail(DSOPP_HIP_ERR_STATE, "the window must hold level-0 frames");
    if (newest_pyramid->levels < 2) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "the newest keyframe needs pyramid level 1 (sparsity level)");
    if (newest_pyramid->width != p0->width || newest_pyramid->height != p0->height || newest_pyramid->dtype != p0->dtype)
      fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "the newest keyframe's pyramid does not match the window's");
    newest_pyramid->waitReady(st);  // its build_device may still be in flight on the pyramid's stream
    for (int k = 0; k < n_keyframes; ++k) {
      const HostFrame &fr = *w->frames[static_cast<size_t>(slots[static_cast<size_t>(k)])];
      if (fr.pyramid->width != p0->width || fr.pyramid->height != p0->height || fr.pyramid->dtype != p0->dtype || fr.level != 0)
        fail(DSOPP_HIP_ERR_STATE, "frame %d: pyramid size / dtype / level differs", fr.id);
      if (immature[k] && immature[k]->sr.stream != st) immature[k]->sr.sync();  // the depth estimator may still be writing the state
    }
    const double *intr = f0.intr;
    // poses and photometric parameters of track.activeFrames(): window frames from the solver state, the newest as given
    std::vector<Rigid> T(static_cast<size_t>(F));
    std::vector<double> expo(static_cast<size_t>(F)), aa(static_cast<size_t>(F)), ab(static_cast<size_t>(F));
    for (int k = 0; k < n_keyframes; ++k) {
      const int s = slots[static_cast<size_t>(k)];
      T[static_cast<size_t>(k)] = poseOf(*w, s);
      expo[static_cast<size_t>(k)] = w->frames[static_cast<size_t>(s)]->exposure;
      aa[static_cast<size_t>(k)] = w->hst.ab0[s][0] + w->hst.eps[s][6];
      ab[static_cast<size_t>(k)] = w->hst.ab0[s][1] + w->hst.eps[s][7];
    }
    T[static_cast<size_t>(F - 1)] = rigidFromParams(T_world_newest);
    expo[static_cast<size_t>(F - 1)] = exposure_newest;
    aa[static_cast<size_t>(F - 1)] = affine_newest[0];
    ab[static_cast<size_t>(F - 1)] = affine_newest[1];
    // [R|t] K^-1 and K [R|t] K^-1 for intrinsics (fx, fy, cx, cy) — camera_reproject.hpp:250-258
    auto reprojectors = [](const Rigid &Ttr, double fx, double fy, double cx, double cy, double *M, double *U) {
      const double ifx = 1.0 / fx, ify = 1.0 / fy, k02 = -cx / fx, k12 = -cy / fy;
      double Ul[12];
      for (int i = 0; i < 3; ++i) {
        Ul[4 * i + 0] = Ttr.R[3 * i + 0] * ifx;
        Ul[4 * i + 1] = Ttr.R[3 * i + 1] * ify;
        Ul[4 * i + 2] = Ttr.R[3 * i + 0] * k02 + Ttr.R[3 * i + 1] * k12 + Ttr.R[3 * i + 2];
        Ul[4 * i + 3] = Ttr.t[i];
      }
      for (int j = 0; j < 4; ++j) {
        M[0 + j] = fx * Ul[0 + j] + cx * Ul[8 + j];
        M[4 + j] = fy * Ul[4 + j] + cy * Ul[8 + j];
        M[8 + j] = Ul[8 + j];
      }
      if (U) std::memcpy(U, Ul, sizeof(Ul));
    };
    // ---- tables
    std::vector<ActKeyframe> kfs(static_cast<size_t>(n_keyframes));
    std::vector<ActPair> pairs(static_cast<size_t>(F) * F);
    std::vector<const void *> tex(static_cast<size_t>(F));
    const Rigid T_newest_inv = rigidInverse(T[static_cast<size_t>(F - 1)]);
    int n_immature = 0, max_items = 0, n_active_cap = 0;
    for (int k = 0; k < n_keyframes; ++k) {
      const HostFrame &fr = *w->frames[static_cast<size_t>(slots[static_cast<size_t>(k)])];
      ActKeyframe &a = kfs[static_cast<size_t>(k)];
      std::memset(&a, 0, sizeof(a));
      reprojectors(rigidMul(T_newest_inv, T[static_cast<size_t>(k)]), intr[0] / 2, intr[1] / 2, intr[2] / 2, intr[3] / 2, a.M, nullptr);  // cameraModel(1)
      a.n_active = fr.n;
      a.active_uv = hbm(fr.uv.ptr);
      a.active_idepth = hbm(fr.idepth.ptr);
      a.active_flags = hbm(fr.dflags.ptr);
      a.frame_slot = k;
      a.immature_offset = n_immature;
      if (const dsopp_hip_immature_set *s = immature[k]) {
        const size_t N = static_cast<size_t>(s->n);
        a.n_immature = s->n;
        a.projection = hbm(s->d_in.ptr);
        a.patch = hbm(s->d_in.ptr + 5 * N);
        a.idepth_min = hbm(s->d_io.ptr);
        a.idepth_max = hbm(s->d_io.ptr + N);
        a.uniqueness = hbm(s->d_io.ptr + 2 * N);
        a.search_pixel_interval = hbm(s->d_io.ptr + 3 * N);
        a.status = hbm(s->d_flags.ptr);
        a.traced = hbm(s->d_flags.ptr + N);
      }
      n_immature += a.n_immature;
      n_active_cap += a.n_active;
      max_items = std::max(max_items, std::max(a.n_active, a.n_immature));
      tex[static_cast<size_t>(k)] = fr.pyramid->texels[0];
    }
    tex[static_cast<size_t>(F - 1)] = newest_pyramid->texels[0];
    for (int r = 0; r < F; ++r)
      for (int t = 0; t < F; ++t) {
        ActPair &pc = pairs[static_cast<size_t>(r) * F + t];
        std::memset(&pc, 0, sizeof(pc));
        if (r == t) continue;
        const Rigid Ttr = rigidMul(rigidInverse(T[static_cast<size_t>(t)]), T[static_cast<size_t>(r)]);  // :169
        reprojectors(Ttr, intr[0], intr[1], intr[2], intr[3], pc.M, pc.U);
        for (int i = 0; i < 3; ++i) pc.t[i] = Ttr.t[i];
        pc.scale = (expo[static_cast<size_t>(t)] / expo[static_cast<size_t>(r)]) * std::exp(aa[static_cast<size_t>(t)] - aa[static_cast<size_t>(r)]);  // :166-167
        pc.b_t = ab[static_cast<size_t>(t)];
        pc.b_r = ab[static_cast<size_t>(r)];
      }
    auto &S = w->act;
    ActArgs a;
    std::memset(&a, 0, sizeof(a));
    a.swd = static_cast<double>(p0->width) / 2.0;  // CameraCalibration::cameraModel(level): image size / 2^level
    a.shd = static_cast<double>(p0->height) / 2.0;
    a.grid_cap = (static_cast<int>(a.swd / kActMinCell) + 1) * (static_cast<int>(a.shd / kActMinCell) + 1);
    const size_t nI = static_cast<size_t>(std::max(n_immature, 1)), nP = static_cast<size_t>(n_active_cap) + nI;
    S.keyframes.reserve(static_cast<size_t>(n_keyframes), 0, st);
    S.pairs.reserve(pairs.size(), 0, st);
    S.texels0.reserve(static_cast<size_t>(F), 0, st);
    S.px.reserve(nP, 0, st);
    S.py.reserve(nP, 0, st);
    const size_t status_words = (nI + 7) / 8, cell_words = (static_cast<size_t>(a.grid_cap) + 2) / 2;
    S.pack.reserve(nI + 1 + status_words + 4 + cell_words, 0, st);
    double *const d_idepth_out = S.pack.ptr, *const d_distance = S.pack.ptr + nI;
    uint8_t *const d_act_status = reinterpret_cast<uint8_t *>(S.pack.ptr + nI + 1);
    int *const d_counters = reinterpret_cast<int *>(S.pack.ptr + nI + 1 + status_words), *const d_cell_start = d_counters + 8;
    S.state.reserve(nI, 0, st);
    S.cell_cursor.reserve(static_cast<size_t>(a.grid_cap) + 1, 0, st);
    S.sx.reserve(nP, 0, st);
    S.sy.reserve(nP, 0, st);
    S.sid.reserve(nP, 0, st);
    S.nbr.reserve(nI * kActNbrCap, 0, st);
    S.nbr_count.reserve(nI, 0, st);
    S.accepted.reserve(nI, 0, st);
    // the three tables travel like a keyframe step's appends: one pinned copy + one scatter launch (three copies from pageable memory before)
    uploadStagedBytes(*w, S.keyframes.ptr, kfs.data(), kfs.size() * sizeof(ActKeyframe));
    uploadStagedBytes(*w, S.pairs.ptr, pairs.data(), pairs.size() * sizeof(ActPair));
    uploadStagedBytes(*w, S.texels0.ptr, tex.data(), tex.size() * sizeof(const void *));
    flushAppends(*w, "flush: activation tables");
    HIP_CHECK(hipMemsetAsync(d_counters, 0, (8 + static_cast<size_t>(a.grid_cap) + 1) * sizeof(int), st));
    a.distance_in = *min_distance_to_neighbor;
    a.keyframes = S.keyframes.ptr;
    a.pairs = S.pairs.ptr;
    a.texels0 = S.texels0.ptr;
    a.newest_sparsity = newest_pyramid->texels[1];
    a.n_keyframes = n_keyframes;
    a.n_frames = F;
    a.n_immature = n_immature;
    a.active_cap = n_active_cap;
    a.width = p0->width;
    a.height = p0->height;
    a.sw = newest_pyramid->w(1);
    a.sh = newest_pyramid->h(1);
    a.fx = intr[0];
    a.fy = intr[1];
    a.cx = intr[2];
    a.cy = intr[3];
    a.sigma = sigma_huber_loss;
    a.desired = number_of_desired_points;
    a.minimum_inliers = std::min(1, F - 1);  // kMinimumInliers, :334
    a.refine = refine ? 1 : 0;
    static const int work_cap_env = std::getenv("DSOPP_HIP_ACT_WORK_CAP") ? std::atoi(std::getenv("DSOPP_HIP_ACT_WORK_CAP")) : kActWorkCap;
    a.work_cap = std::max(0, work_cap_env);
    a.px = S.px.ptr;
    a.py = S.py.ptr;
    a.counters = d_counters;
    a.distance = d_distance;
    a.state = S.state.ptr;
    a.cell_start = d_cell_start;
    a.cell_cursor = S.cell_cursor.ptr;
    a.sx = S.sx.ptr;
    a.sy = S.sy.ptr;
    a.sid = S.sid.ptr;
    a.nbr = S.nbr.ptr;
    a.nbr_count = S.nbr_count.ptr;
    a.accepted = S.accepted.ptr;
    a.act_status = d_act_status;
    a.idepth_out = d_idepth_out;
    const bool f64 = p0->dtype == DSOPP_HIP_F64;
    if (max_items > 0) {
      const dim3 grid(static_cast<unsigned>((max_items + 255) / 256), static_cast<unsigned>(n_keyframes));
      if (f64) activationProjectKernel<double><<<grid, 256, 0, st>>>(a);
      else activationProjectKernel<float><<<grid, 256, 0, st>>>(a);
    }
    {
      const unsigned pts = static_cast<unsigned>((n_active_cap + n_immature + 255) / 256);
      if (pts) activationCellCountKernel<<<pts, 256, 0, st>>>(a);
      activationCellScanKernel<<<1, kActSelectThreads, 0, st>>>(a);
      if (pts) activationCellFillKernel<<<pts, 256, 0, st>>>(a);
      if (n_immature > 0) activationNeighboursKernel<<<static_cast<unsigned>((n_immature + 255) / 256), 256, 0, st>>>(a);
      activationResolveKernel<<<1, kActSelectThreads, 0, st>>>(a);
    }
    if (n_immature > 0) {
      if (f64) activationRefineKernel<double><<<static_cast<unsigned>(n_immature), 64, 0, st>>>(a);
      else activationRefineKernel<float><<<static_cast<unsigned>(n_immature), 64, 0, st>>>(a);
    }
    HIP_CHECK(hipGetLastError());
    // ---- one read-back of the contiguous results: idepth (nI doubles) | distance | statuses (nI bytes, padded to words) | counters (8 ints)
    const size_t bytes = (nI + 1 + status_words + 4) * sizeof(double);
    growPinned(w->h_export, w->h_export_bytes, bytes);
    char *h = static_cast<char *>(w->h_export);
    HIP_CHECK(hipMemcpyAsync(h, S.pack.ptr, bytes, hipMemcpyDeviceToHost, st));
    w->sr.sync();
    const double *h_id = reinterpret_cast<const double *>(h);
    const int *h_cnt = reinterpret_cast<const int *>(h + (nI + 1 + status_words) * sizeof(double));
    const uint8_t *h_st = reinterpret_cast<const uint8_t *>(h + (nI + 1) * sizeof(double));
    *min_distance_to_neighbor = h_id[nI];
    dsopp_hip_activation_result res;
    std::memset(&res, 0, sizeof(res));
    res.number_of_active_points = h_cnt[0];
    res.selection_rounds = h_cnt[3];
    res.min_distance_to_neighbor = *min_distance_to_neighbor;
    for (int k = 0; k < n_keyframes; ++k) {
      const ActKeyframe &kf = kfs[static_cast<size_t>(k)];
      for (int i = 0; i < kf.n_immature; ++i) {
        const uint8_t s = h_st[kf.immature_offset + i];
        (s == kActActivate ? res.n_activated : s == kActSkip ? res.n_skipped : res.n_deleted)++;
      }
      if (activation_status && activation_status[k]) std::memcpy(activation_status[k], h_st + kf.immature_offset, static_cast<size_t>(kf.n_immature));
      if (idepth && idepth[k]) std::memcpy(idepth[k], h_id + kf.immature_offset, static_cast<size_t>(kf.n_immature) * 8);
    }
    if (result) *result = res;
  });
}
