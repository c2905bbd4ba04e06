// dsopp_hip_window_group_*: ONE host process, n landmark shards of one sliding window on n devices.
//
// Why it exists: the reference creates ONE solver object in ONE process (src/tracker/tracker/src/fabric.cpp:58-121, moved into the
// tracker of src/application/dsopp_main.cpp:114-119), so a drop-in that wants more than one GPU has to own all of them behind that
// one object — it cannot ask the host application to become a launcher of one process per GPU.  A window group is that object: it
// has the calls of dsopp_hip_window (same argument meaning, same error codes), replicates frames / images / poses on every device,
// deals the landmarks of every keyframe round-robin over the shards (stable under the appends of
// PROB_SRC/photometric_bundle_adjustment.cpp:109-123: landmark j of a frame lives on shard j % n at local index j / n) and lets
// every shard run the unchanged sharded solve of pba.hip (partial systems -> ONE collective per Gauss-Newton iteration -> replicated
// decision and solve -> back-substitution of the own landmarks).
//
// Host side: one worker thread per shard.  A group call posts the same job to every worker and waits; a shard's launches are
// enqueued by its own thread on its own device, so the host cost of an iteration does not grow with the device count (a single
// enqueueing thread would need n x 4 launches per iteration: at 8 devices that is longer than the iteration itself).
//
// Exchange: RCCL by default when the shards sit on distinct devices — every worker owns one rank of a communicator created from one
// unique id (ncclCommInitRank from n threads of one process), ncclAllReduce on the shard's stream, exactly the native path of
// comm.hip.  Shards that share a device (RCCL refuses two ranks per device: the single-GPU test set-up) or DSOPP_HIP_TRANSPORT_LOCAL
// use the in-process reducer below: every shard records an event behind its partial sums, shard 0's stream waits for all of them,
// ONE kernel adds the n buffers in shard order and writes the sum back into every buffer (peer access across devices), the other
// streams wait for its event.  No spinning kernels, no host synchronisation: stream dependencies only.  Sums in shard order, so the
// result is bit-reproducible and identical on every shard.
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <thread>

#include "common.hpp"

using namespace dsopp_hip;

namespace dsopp_hip {
namespace {

constexpr int kMaxShards = 16;
constexpr int kBlkConst = DSOPP_HIP_BLOCK_SIZE;

struct ShardBuffers {
  double *p[kMaxShards];
};

/** out-of-place-free all-reduce inside one process: every buffer receives the sum of all buffers, added in shard order */
__global__ void sumShardBuffersKernel(ShardBuffers b, int n, size_t count) {
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < count; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    double v[kMaxShards];
#pragma unroll
    for (int r = 0; r < kMaxShards; ++r) v[r] = r < n ? b.p[r][i] : 0.0;
    double s = 0;
#pragma unroll
    for (int r = 0; r < kMaxShards; ++r) s += v[r];
#pragma unroll
    for (int r = 0; r < kMaxShards; ++r)
      if (r < n) b.p[r][i] = s;
  }
}

/** host barrier of the shard workers that gives up when the group is aborting (a shard failed outside the collective) */
struct AbortableBarrier {
  std::mutex m;
  std::condition_variable cv;
  int n = 1, arrived = 0;
  unsigned generation = 0;
  std::atomic<bool> *abort = nullptr;
  bool wait() {
    std::unique_lock<std::mutex> lk(m);
    const unsigned gen = generation;
    if (++arrived == n) {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    cv.wait(lk, [&] { return generation != gen || abort->load(); });
    return generation != gen;
  }
  void wake() {
    std::lock_guard<std::mutex> lk(m);
    cv.notify_all();
  }
  /** between two jobs of the pool (no worker is inside wait()): workers that gave up during an aborted job left their arrivals behind */
  void reset() {
    std::lock_guard<std::mutex> lk(m);
    arrived = 0;
  }
};

struct LocalReducer {
  int n = 0;
  std::vector<int> device;
  std::vector<hipEvent_t> ev_in;
  hipEvent_t ev_out = nullptr;
  struct Slot {
    double *buf = nullptr;
    size_t count = 0;
    hipStream_t stream = nullptr;
  };
  std::vector<Slot> slot;
  AbortableBarrier bar;
  std::atomic<int> failed{0};
};

struct ShardUser {
  LocalReducer *red;
  int shard;
};

/** dsopp_hip_allreduce_fn of a shard (runs on that shard's worker thread, between the launches of its solve) */
int localAllreduce(void *user, void *device_buffer, size_t count, void *stream) {
  auto *u = static_cast<ShardUser *>(user);
  LocalReducer &r = *u->red;
  const int i = u->shard;
  hipStream_t st = static_cast<hipStream_t>(stream);
  r.slot[static_cast<size_t>(i)] = {static_cast<double *>(device_buffer), count, st};
  if (hipEventRecord(r.ev_in[static_cast<size_t>(i)], st) != hipSuccess) r.failed.store(1);
  if (!r.bar.wait()) return -2;
  if (i == 0) {
    ShardBuffers b{};
    bool ok = true;
    for (int j = 0; j < r.n; ++j) {
      const LocalReducer::Slot &s = r.slot[static_cast<size_t>(j)];
      ok = ok && s.count == count && s.buf != nullptr;
      b.p[j] = s.buf;
      if (j > 0 && hipStreamWaitEvent(st, r.ev_in[static_cast<size_t>(j)], 0) != hipSuccess) ok = false;
    }
    if (ok && count) {
      const unsigned grid = static_cast<unsigned>(std::min<size_t>((count + 255) / 256, 1024));
      sumShardBuffersKernel<<<grid, 256, 0, st>>>(b, r.n, count);
      ok = hipGetLastError() == hipSuccess;
    }
    if (hipEventRecord(r.ev_out, st) != hipSuccess) ok = false;
    if (!ok) r.failed.store(1);
  }
  if (!r.bar.wait()) return -2;
  if (i != 0 && hipStreamWaitEvent(st, r.ev_out, 0) != hipSuccess) r.failed.store(1);
  return r.failed.load() ? -1 : 0;
}

// ---- DSOPP_HIP_TRANSPORT_P2P: the one-shot all-reduce ----------------------------------------------------------------------------
// The per-iteration collective is 15 .. 41 KB: a ring over xGMI (RCCL) pays its latency n - 1 times, the in-process reducer above a
// host barrier.  Here every shard stores its partial sums straight into EVERY peer's receive area (xGMI is point to point: 7 direct
// links per device, one write latency) and adds up what the others stored into its own — one kernel per shard, no host involvement:
//   workgroup w of shard s:  push slice w of the buffer into recv[p][s] of every shard p  ->  system-scope release of flag[p][s][w]
//                            ->  wait (system-scope acquire) for flag[s][q][w] of every shard q  ->  slice w of the sum, added in
//                            shard order (deterministic, identical on every shard), back into the buffer.
// Slice w of the result needs slice w of every source only, so the workgroups never meet.  Receive areas and flags are double
// buffered by the collective's parity: a shard can be at most one collective ahead of the slowest (it needs everybody's pushes to
// finish its own).  Flags carry the collective's number and are never reset.  The memory is fine-grained (coherent across devices
// while kernels run).  Every spin is bounded: a time-out raises an error word in pinned memory that the group reports after the call.
// Buffers larger than the receive area (the energies of the point statuses, the depth-map planes: once per solve / keyframe) take the
// in-process reducer above.  Only executable on one device here (repeated ids): peers are then the device itself.
constexpr int kP2PBlocks = 32, kP2PThreads = 256;
constexpr size_t kP2PCapacity = 2 * (static_cast<size_t>(kBlkConst * DSOPP_HIP_MAX_FRAMES) * (kBlkConst * DSOPP_HIP_MAX_FRAMES) + kBlkConst * DSOPP_HIP_MAX_FRAMES) + 8 + 4 * 64;

struct P2PArgs {
  double *recv_of[kMaxShards];              // per shard: [2 parities][n sources][kP2PCapacity]
  unsigned long long *flags_of[kMaxShards]; // per shard: [2 parities][n sources][kP2PBlocks]
  double *buf_of[kMaxShards];               // the shards' buffers (distinct devices: only [shard] is used)
  int *error;                               // pinned host words, one per shard
  unsigned long long generation;
  int shard, n;                             // shard < 0: ONE launch plays every shard, shard = blockIdx.y (all shards on one device)
  unsigned count;
};

__global__ void __launch_bounds__(kP2PThreads) p2pAllreduceKernel(P2PArgs a) {
  __shared__ int s_timeout;
  const int w = blockIdx.x, tid = threadIdx.x;
  // Shards on distinct devices launch one kernel each.  Shards that share a device cannot: kernels of different streams are not
  // guaranteed to run concurrently (streams may share a hardware queue), so one waiting for the other can wait for ever — there the
  // shards' workgroups are the rows of ONE launch (blockIdx.y), resident together, and run the same protocol.
  const int shard = a.shard < 0 ? static_cast<int>(blockIdx.y) : a.shard;
  double *const buf = a.buf_of[shard];
  int *const error = a.error + shard;
  const unsigned per = ((a.count + kP2PBlocks - 1) / kP2PBlocks + 1u) & ~1u;
  const unsigned lo = min(a.count, w * per), hi = min(a.count, lo + per);
  const size_t parity = a.generation & 1ull;
  if (tid == 0) s_timeout = 0;
  // push this workgroup's slice into every shard's receive area (its own included: one code path, sums in shard order)
  for (int p = 0; p < a.n; ++p) {
    double *dst = a.recv_of[p] + (parity * a.n + shard) * kP2PCapacity;
    for (unsigned i = lo + tid; i < hi; i += kP2PThreads) dst[i] = buf[i];
  }
  __threadfence_system();
  __syncthreads();
  if (tid < a.n)
    __hip_atomic_store(a.flags_of[tid] + (parity * a.n + shard) * kP2PBlocks + w, a.generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // wait for the slice of every source
  if (tid < a.n) {
    const unsigned long long *f = a.flags_of[shard] + (parity * a.n + tid) * kP2PBlocks + w;
    unsigned spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != a.generation) {
      if (++spins > (1u << 22) || __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) {
        s_timeout = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(8);
    }
  }
  __syncthreads();
  if (s_timeout) {
    if (tid == 0) __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  __threadfence_system();  // (acquire side for the lanes that did not poll)
  const double *mine = a.recv_of[shard] + parity * a.n * kP2PCapacity;
  for (unsigned i = lo + tid; i < hi; i += kP2PThreads) {
    double sum = 0;
    for (int q = 0; q < a.n; ++q) sum += mine[static_cast<size_t>(q) * kP2PCapacity + i];
    buf[i] = sum;
  }
}

struct P2PReducer {
  int n = 0;
  std::vector<double *> recv;
  std::vector<unsigned long long *> flags;
  std::vector<unsigned long long> generation;  // per shard: collectives issued so far (all shards issue the same sequence)
  int *error = nullptr;                        // pinned host memory, one word per shard
  std::vector<int> device;
  bool one_device = false;                     // every shard on the same device: one launch plays all shards (see the kernel)
};

struct P2PUser {
  P2PReducer *p2p;
  ShardUser *local;  // buffers beyond the receive area take the in-process reducer
  int shard;
};

int p2pAllreduce(void *user, void *device_buffer, size_t count, void *stream) {
  auto *u = static_cast<P2PUser *>(user);
  if (count > kP2PCapacity) return localAllreduce(u->local, device_buffer, count, stream);
  P2PReducer &r = *u->p2p;
  hipStream_t st = static_cast<hipStream_t>(stream);
  P2PArgs a;
  for (int p = 0; p < r.n; ++p) {
    a.recv_of[p] = r.recv[static_cast<size_t>(p)];
    a.flags_of[p] = r.flags[static_cast<size_t>(p)];
    a.buf_of[p] = nullptr;
  }
  a.error = r.error;
  a.n = r.n;
  a.count = static_cast<unsigned>(count);
  if (!r.one_device) {
    a.buf_of[u->shard] = static_cast<double *>(device_buffer);
    a.generation = ++r.generation[static_cast<size_t>(u->shard)];
    a.shard = u->shard;
    p2pAllreduceKernel<<<kP2PBlocks, kP2PThreads, 0, st>>>(a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
  }
  // all shards on one device: event-ordered like the in-process reducer (ev_in -> host barrier -> shard 0 launches -> ev_out), but what
  // shard 0 launches is the peer-to-peer kernel with one row of workgroups per shard
  LocalReducer &lr = *u->local->red;
  const int i = u->shard;
  lr.slot[static_cast<size_t>(i)] = {static_cast<double *>(device_buffer), count, st};
  if (hipEventRecord(lr.ev_in[static_cast<size_t>(i)], st) != hipSuccess) lr.failed.store(1);
  if (!lr.bar.wait()) return -2;
  if (i == 0) {
    bool ok = true;
    for (int j = 0; j < r.n; ++j) {
      const LocalReducer::Slot &sl = lr.slot[static_cast<size_t>(j)];
      ok = ok && sl.count == count && sl.buf != nullptr;
      a.buf_of[j] = sl.buf;
      if (j > 0 && hipStreamWaitEvent(st, lr.ev_in[static_cast<size_t>(j)], 0) != hipSuccess) ok = false;
    }
    if (ok) {
      a.generation = ++r.generation[0];
      a.shard = -1;
      p2pAllreduceKernel<<<dim3(kP2PBlocks, static_cast<unsigned>(r.n)), kP2PThreads, 0, st>>>(a);
      ok = hipGetLastError() == hipSuccess;
    }
    if (hipEventRecord(lr.ev_out, st) != hipSuccess) ok = false;
    if (!ok) lr.failed.store(1);
  }
  if (!lr.bar.wait()) return -2;
  if (i != 0 && hipStreamWaitEvent(st, lr.ev_out, 0) != hipSuccess) lr.failed.store(1);
  return lr.failed.load() ? -1 : 0;
}

/** one worker thread per shard; run() executes the same job on all of them and returns the per-shard status codes */
struct ShardPool {
  int n = 0;
  std::vector<std::thread> threads;
  std::mutex m;
  std::condition_variable cv_job, cv_done;
  std::function<int(int)> job;
  unsigned generation = 0;
  int remaining = 0;
  bool quit = false;
  std::vector<int> rc;
  std::vector<std::string> err;
  std::vector<char> secondary;  // the shard failed AFTER another one had raised the abort flag: a consequence, not the cause
  std::atomic<bool> abort{false};
  AbortableBarrier *barrier = nullptr;
  std::function<void()> on_abort;  // first failure of a job: release what the other shards may be blocked in (RCCL: abort the communicators)
  std::function<void()> on_reset;  // in front of the job that follows an aborted one

  void start(int count, const std::vector<int> &devices) {
    n = count;
    rc.assign(static_cast<size_t>(n), 0);
    err.assign(static_cast<size_t>(n), std::string());
    secondary.assign(static_cast<size_t>(n), 0);
    for (int i = 0; i < n; ++i)
      threads.emplace_back([this, i, dev = devices[static_cast<size_t>(i)]] {
        (void)hipSetDevice(dev);
        unsigned seen = 0;
        for (;;) {
          std::function<int(int)> fn;
          {
            std::unique_lock<std::mutex> lk(m);
            cv_job.wait(lk, [&] { return quit || generation != seen; });
            if (quit) return;
            seen = generation;
            fn = job;
          }
          int code = DSOPP_HIP_OK;
          std::string msg;
          try {
            code = fn(i);
            if (code != DSOPP_HIP_OK) msg = dsopp_hip_last_error();
          } catch (const Error &e) {
            code = e.code;
            msg = e.what();
          } catch (const std::exception &e) {
            code = DSOPP_HIP_ERR_INVALID_ARGUMENT;
            msg = e.what();
          }
          bool was_aborting = false;
          if (code != DSOPP_HIP_OK) {
            // the other shards may be waiting for this one inside a collective: let their barrier give up
            was_aborting = abort.exchange(true);
            if (barrier) barrier->wake();
            if (!was_aborting && on_abort) on_abort();
          }
          {
            std::lock_guard<std::mutex> lk(m);
            rc[static_cast<size_t>(i)] = code;
            err[static_cast<size_t>(i)] = msg;
            secondary[static_cast<size_t>(i)] = was_aborting ? 1 : 0;
            if (--remaining == 0) cv_done.notify_all();
          }
        }
      });
  }
  /** throws the first shard's failure (shard order) */
  void run(std::function<int(int)> fn) {
    {
      std::lock_guard<std::mutex> lk(m);
      job = std::move(fn);
      remaining = n;
      if (abort.load()) {  // the previous job was aborted inside a collective
        if (barrier) barrier->reset();
        if (on_reset) on_reset();
      }
      abort.store(false);
      ++generation;
    }
    cv_job.notify_all();
    {
      std::unique_lock<std::mutex> lk(m);
      cv_done.wait(lk, [&] { return remaining == 0; });
    }
    // a shard that gave up inside a collective because ANOTHER shard failed reports the callback error: prefer the root cause (the
    // shard that raised the abort flag; every later failure of the job is marked secondary)
    int first = -1;
    for (int i = 0; i < n; ++i)
      if (rc[static_cast<size_t>(i)] != DSOPP_HIP_OK && (first < 0 || (secondary[static_cast<size_t>(first)] && !secondary[static_cast<size_t>(i)]))) first = i;
    if (first >= 0) fail(rc[static_cast<size_t>(first)], "shard %d: %s", first, err[static_cast<size_t>(first)].c_str());
  }
  void stop() {
    {
      std::lock_guard<std::mutex> lk(m);
      quit = true;
    }
    cv_job.notify_all();
    for (auto &t : threads)
      if (t.joinable()) t.join();
    threads.clear();
  }
};

/** landmarks of a frame on shard s when the frame holds n_total: j = s, s + S, ... */
inline int shardCount(int n_total, int s, int S) { return n_total > s ? (n_total - s + S - 1) / S : 0; }

}  // namespace
}  // namespace dsopp_hip

struct dsopp_hip_window_group {
  int n = 0;
  int transport = DSOPP_HIP_TRANSPORT_RCCL;
  dsopp_hip_options opt;
  std::vector<int> device;
  std::vector<dsopp_hip_window *> win;
  std::vector<dsopp_hip_comm *> comm;
  // depth maps of the shards > 0 (their side of the collective fill), keyed by the shard-0 object the caller holds: a refill of maps from
  // an EARLIER create call must meet the shadows made with them (their levels / sizes), not the latest ones.  The caller destroys the
  // shard-0 object itself; its shadows are dropped when the group goes, or when more than kMaxShadowSets sets have accumulated.
  struct ShadowSet {
    dsopp_hip_depth_maps *key = nullptr;
    std::vector<dsopp_hip_depth_maps *> of_shard;
  };
  std::vector<ShadowSet> shadow_sets;
  static constexpr size_t kMaxShadowSets = 4;
  LocalReducer reducer;
  std::vector<ShardUser> users;
  P2PReducer p2p;
  std::vector<P2PUser> p2p_users;
  ShardPool pool;
  bool pooled = false;               // calls go through the worker threads (n > 1, or DSOPP_HIP_GROUP_FORCE_POOL for a group of one)
  std::atomic<bool> poisoned{false};  // RCCL transport: a shard failed and the communicators were aborted — the group only waits to be destroyed
  // per-shard marshalling buffers (one set per worker thread, reused between calls)
  struct Scratch {
    std::vector<double> a, b, c, d, e, h;
    std::vector<int32_t> i32;
    std::vector<uint8_t> u8, v8;
  };
  std::vector<Scratch> scratch;
};

struct dsopp_hip_pyramid_group {
  dsopp_hip_window_group *group = nullptr;
  std::vector<int> device;                  // distinct devices of the group, in order of first use
  std::vector<dsopp_hip_pyramid *> pyramid;  // one per distinct device: shards that share a device share the image
  std::vector<int> of_shard;                 // shard -> index into `pyramid`
};

namespace {

using G = dsopp_hip_window_group;

/** runs body(shard, window) on every shard's worker thread; any failure is raised on the calling thread */
template <typename Body>
void fanOut(G &g, Body &&body) {
  if (g.poisoned.load())
    fail(DSOPP_HIP_ERR_STATE, "this window group is unusable: a shard failed outside a collective and the RCCL communicators were aborted (destroy the group)");
  if (!g.pooled) {  // a group of one is a plain window: no worker thread, the call runs where the caller is
    const int rc = body(0, g.win[0]);
    if (rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());
    return;
  }
  g.pool.run([&](int s) -> int { return body(s, g.win[static_cast<size_t>(s)]); });
  if (g.p2p.error) {  // DSOPP_HIP_TRANSPORT_P2P: a bounded wait of the one-shot all-reduce ran out (raised by the kernel in pinned memory)
    for (int s = 0; s < g.n; ++s)
      if (g.p2p.error[s]) {
        g.poisoned.store(true);
        fail(DSOPP_HIP_ERR_HIP, "shard %d: the peer-to-peer all-reduce timed out waiting for another shard's partial sums (its results are invalid; destroy the group)", s);
      }
  }
}

void checkGroup(const G *g) {
  if (!g) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null window group");
}

}  // namespace

extern "C" {

int dsopp_hip_window_group_create(const dsopp_hip_options *options, const int32_t *device_ids, int32_t n, int32_t transport,
                                  dsopp_hip_window_group **out) {
  return guarded([&] {
    if (!options || !device_ids || !out) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (n < 1 || n > kMaxShards) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "a window group holds 1 .. %d shards, %d requested", kMaxShards, n);
    if (transport != DSOPP_HIP_TRANSPORT_AUTO && transport != DSOPP_HIP_TRANSPORT_RCCL && transport != DSOPP_HIP_TRANSPORT_LOCAL && transport != DSOPP_HIP_TRANSPORT_P2P)
      fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "unknown transport %d", transport);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) fail(DSOPP_HIP_ERR_HIP, "no HIP device available (this library has no CPU fallback)");
    bool distinct = true;
    for (int i = 0; i < n; ++i) {
      if (device_ids[i] < 0 || device_ids[i] >= count) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", device_ids[i], count);
      for (int j = 0; j < i; ++j) distinct = distinct && device_ids[j] != device_ids[i];
    }
    const bool automatic = transport == DSOPP_HIP_TRANSPORT_AUTO;
    if (automatic) transport = (distinct && n > 1) ? DSOPP_HIP_TRANSPORT_RCCL : DSOPP_HIP_TRANSPORT_LOCAL;
    if (transport == DSOPP_HIP_TRANSPORT_RCCL && !distinct)
      fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "RCCL needs one device per shard (two shards share a device): use DSOPP_HIP_TRANSPORT_LOCAL");
    auto g = std::make_unique<G>();
    g->n = n;
    g->transport = transport;
    g->opt = *options;
    g->device.assign(device_ids, device_ids + n);
    g->win.assign(static_cast<size_t>(n), nullptr);
    g->comm.assign(static_cast<size_t>(n), nullptr);
    g->scratch.resize(static_cast<size_t>(n));
    g->users.resize(static_cast<size_t>(n));
    g->pool.barrier = &g->reducer.bar;
    g->reducer.bar.abort = &g->pool.abort;
    // DSOPP_HIP_GROUP_FORCE_POOL: a group of ONE shard goes through the worker thread and, with DSOPP_HIP_TRANSPORT_RCCL, through a
    // one-rank communicator (ncclCommInitRank from the worker, ncclAllReduce per Gauss-Newton iteration) — the code a multi-device group
    // runs, executable on a one-GPU box (tests/test_gpu_window_group.py)
    static const bool force_pool = std::getenv("DSOPP_HIP_GROUP_FORCE_POOL") != nullptr && std::atoi(std::getenv("DSOPP_HIP_GROUP_FORCE_POOL")) != 0;
    g->pooled = n > 1 || force_pool;
    if (g->pooled) g->pool.start(n, g->device);
    G *const gp = g.get();
    g->pool.on_reset = [gp] { gp->reducer.failed.store(0); };
    struct Cleanup {  // a failure below must not leak threads / windows
      std::unique_ptr<G> &g;
      bool armed = true;
      ~Cleanup() {
        if (armed && g) dsopp_hip_window_group_destroy(g.release());
      }
    } cleanup{g};
    fanOut(*g, [&](int s, dsopp_hip_window *) {
      return dsopp_hip_window_create(options, g->device[static_cast<size_t>(s)], nullptr, &g->win[static_cast<size_t>(s)]);
    });
    if (g->pooled && transport == DSOPP_HIP_TRANSPORT_RCCL) {
      try {
        uint8_t id[DSOPP_HIP_COMM_ID_BYTES];
        if (const int rc = dsopp_hip_comm_unique_id(id); rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());
        // ncclCommInitRank blocks until every rank has joined: all workers call it together
        fanOut(*g, [&](int s, dsopp_hip_window *w) {
          const int rc = dsopp_hip_comm_create(id, s, n, g->device[static_cast<size_t>(s)], &g->comm[static_cast<size_t>(s)]);
          return rc != DSOPP_HIP_OK ? rc : dsopp_hip_window_set_comm(w, g->comm[static_cast<size_t>(s)]);
        });
      } catch (const Error &) {
        // AUTO only: a node without a usable librccl (the library is dlopen'ed) still gets its group — the in-process reducer
        // over peer access.  An explicit DSOPP_HIP_TRANSPORT_RCCL request fails loudly instead.
        if (!automatic) throw;
        fanOut(*g, [&](int s, dsopp_hip_window *w) {
          (void)dsopp_hip_window_set_comm(w, nullptr);
          if (g->comm[static_cast<size_t>(s)]) dsopp_hip_comm_destroy(g->comm[static_cast<size_t>(s)]);
          g->comm[static_cast<size_t>(s)] = nullptr;
          return static_cast<int>(DSOPP_HIP_OK);
        });
        transport = DSOPP_HIP_TRANSPORT_LOCAL;
        g->transport = transport;
      }
      if (transport == DSOPP_HIP_TRANSPORT_RCCL) {
        // a shard that fails BETWEEN two collectives never enqueues its side of the next one: the other shards' ncclAllReduce kernels
        // would wait for it for ever and their stream synchronisation with them.  The first failure aborts every communicator
        // (ncclCommAbort releases the enqueued kernels), the call returns that shard's error and the group is unusable from then on.
        g->pool.on_abort = [gp] {
          gp->poisoned.store(true);
          for (dsopp_hip_comm *c : gp->comm)
            if (c) (void)dsopp_hip_comm_abort(c);
        };
      }
    }
    if (n > 1 && (transport == DSOPP_HIP_TRANSPORT_LOCAL || transport == DSOPP_HIP_TRANSPORT_P2P)) {
      LocalReducer &r = g->reducer;
      r.n = n;
      r.device = g->device;
      r.slot.resize(static_cast<size_t>(n));
      r.ev_in.assign(static_cast<size_t>(n), nullptr);
      r.bar.n = n;
      // shard 0's device adds the buffers of all shards: it needs peer access to every other device of the group
      HIP_CHECK(hipSetDevice(g->device[0]));
      for (int i = 1; i < n; ++i) {
        if (g->device[static_cast<size_t>(i)] == g->device[0]) continue;
        int can = 0;
        HIP_CHECK(hipDeviceCanAccessPeer(&can, g->device[0], g->device[static_cast<size_t>(i)]));
        if (!can) fail(DSOPP_HIP_ERR_HIP, "device %d cannot access device %d: the in-process reducer needs peer access", g->device[0], g->device[static_cast<size_t>(i)]);
        const hipError_t e = hipDeviceEnablePeerAccess(g->device[static_cast<size_t>(i)], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(e);
        (void)hipGetLastError();
      }
      HIP_CHECK(hipEventCreateWithFlags(&r.ev_out, hipEventDisableTiming));
      const bool p2p = transport == DSOPP_HIP_TRANSPORT_P2P;
      if (p2p) {
        // every shard writes into every shard's receive area: peer access between all pairs of distinct devices
        for (int i = 0; i < n; ++i) {
          HIP_CHECK(hipSetDevice(g->device[static_cast<size_t>(i)]));
          for (int j = 0; j < n; ++j) {
            if (g->device[static_cast<size_t>(i)] == g->device[static_cast<size_t>(j)]) continue;
            int can = 0;
            HIP_CHECK(hipDeviceCanAccessPeer(&can, g->device[static_cast<size_t>(i)], g->device[static_cast<size_t>(j)]));
            if (!can) fail(DSOPP_HIP_ERR_HIP, "device %d cannot access device %d: the peer-to-peer all-reduce needs peer access between all shards",
                           g->device[static_cast<size_t>(i)], g->device[static_cast<size_t>(j)]);
            const hipError_t e = hipDeviceEnablePeerAccess(g->device[static_cast<size_t>(j)], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_CHECK(e);
            (void)hipGetLastError();
          }
        }
        P2PReducer &q = g->p2p;
        q.n = n;
        q.device = g->device;
        bool same = true;
        for (int i = 1; i < n; ++i) same = same && g->device[static_cast<size_t>(i)] == g->device[0];
        if (!same && !distinct)
          fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "the peer-to-peer transport takes shards on all-distinct devices, or all on one device (kernels of shards that share a "
                                               "device cannot wait for each other)");
        // The distinct-device branch (one kernel per shard, remote xGMI stores into fine-grained receive areas, generation-valued
        // flags) has never executed: no box this library was built on had two GPUs.  Until one has, it is an opt-in experiment —
        // an explicit request without the opt-in is refused instead of running untested synchronisation in a caller's process.
        static const bool p2p_experimental = std::getenv("DSOPP_HIP_P2P_EXPERIMENTAL") != nullptr && std::atoi(std::getenv("DSOPP_HIP_P2P_EXPERIMENTAL")) != 0;
        if (!same && !p2p_experimental)
          fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "DSOPP_HIP_TRANSPORT_P2P across distinct devices is experimental (never run on a multi-GPU node): set "
                                               "DSOPP_HIP_P2P_EXPERIMENTAL=1 to opt in, or use DSOPP_HIP_TRANSPORT_RCCL / _LOCAL");
        q.one_device = same;
        q.recv.assign(static_cast<size_t>(n), nullptr);
        q.flags.assign(static_cast<size_t>(n), nullptr);
        q.generation.assign(static_cast<size_t>(n), 0);
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&q.error), sizeof(int) * static_cast<size_t>(n), hipHostMallocDefault));
        for (int i = 0; i < n; ++i) q.error[i] = 0;
        g->p2p_users.resize(static_cast<size_t>(n));
      }
      fanOut(*g, [&](int s, dsopp_hip_window *w) {
        if (hipEventCreateWithFlags(&r.ev_in[static_cast<size_t>(s)], hipEventDisableTiming) != hipSuccess) {
          lastError() = "hipEventCreate failed";
          return static_cast<int>(DSOPP_HIP_ERR_HIP);
        }
        g->users[static_cast<size_t>(s)] = {&r, s};
        if (!p2p) return dsopp_hip_window_set_allreduce(w, &localAllreduce, &g->users[static_cast<size_t>(s)], s, n);
        // receive area and flags of this shard, on its device, fine-grained (coherent between devices while kernels run), zeroed
        P2PReducer &q = g->p2p;
        const size_t recv_bytes = 2 * static_cast<size_t>(n) * kP2PCapacity * sizeof(double);
        const size_t flag_bytes = 2 * static_cast<size_t>(n) * kP2PBlocks * sizeof(unsigned long long);
        void *rp = nullptr, *fp = nullptr;
        if (hipExtMallocWithFlags(&rp, recv_bytes, hipDeviceMallocFinegrained) != hipSuccess ||
            hipExtMallocWithFlags(&fp, flag_bytes, hipDeviceMallocFinegrained) != hipSuccess || hipMemset(fp, 0, flag_bytes) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) {
          lastError() = "allocation of the peer-to-peer receive area failed";
          return static_cast<int>(DSOPP_HIP_ERR_HIP);
        }
        q.recv[static_cast<size_t>(s)] = static_cast<double *>(rp);
        q.flags[static_cast<size_t>(s)] = static_cast<unsigned long long *>(fp);
        g->p2p_users[static_cast<size_t>(s)] = {&q, &g->users[static_cast<size_t>(s)], s};
        return dsopp_hip_window_set_allreduce(w, &p2pAllreduce, &g->p2p_users[static_cast<size_t>(s)], s, n);
      });
    }
    cleanup.armed = false;
    *out = g.release();
  });
}

void dsopp_hip_window_group_destroy(dsopp_hip_window_group *g) {
  if (!g) return;
  auto release = [&](int s) -> int {
    const size_t i = static_cast<size_t>(s);
    for (auto &set : g->shadow_sets)
      if (i > 0 && i < set.of_shard.size() && set.of_shard[i]) dsopp_hip_depth_maps_destroy(set.of_shard[i]);  // ([0] is the caller's)
    if (g->win[i]) dsopp_hip_window_destroy(g->win[i]);
    if (g->comm[i]) dsopp_hip_comm_destroy(g->comm[i]);
    if (static_cast<size_t>(s) < g->reducer.ev_in.size() && g->reducer.ev_in[i]) (void)hipEventDestroy(g->reducer.ev_in[i]);
    if (i < g->p2p.recv.size()) {
      if (g->p2p.recv[i]) (void)hipFree(g->p2p.recv[i]);
      if (g->p2p.flags[i]) (void)hipFree(g->p2p.flags[i]);
    }
    return DSOPP_HIP_OK;
  };
  try {
    g->poisoned.store(false);  // (the release job itself must run)
    if (g->p2p.error)
      for (int s = 0; s < g->n; ++s) g->p2p.error[s] = 0;
    if (!g->pool.threads.empty())
      g->pool.run(release);
    else
      for (int s = 0; s < g->n; ++s) release(s);
  } catch (...) {
  }
  g->pool.stop();
  if (g->p2p.error) (void)hipHostFree(g->p2p.error);
  if (g->reducer.ev_out) {
    (void)hipSetDevice(g->device[0]);
    (void)hipEventDestroy(g->reducer.ev_out);
  }
  delete g;
}

int dsopp_hip_window_group_size(const dsopp_hip_window_group *g, int32_t *n, int32_t *transport) {
  return guarded([&] {
    checkGroup(g);
    if (n) *n = g->n;
    if (transport) *transport = g->pooled ? g->transport : DSOPP_HIP_TRANSPORT_LOCAL;
  });
}

int dsopp_hip_window_group_shard(dsopp_hip_window_group *g, int32_t shard, dsopp_hip_window **window, int32_t *device) {
  return guarded([&] {
    checkGroup(g);
    if (shard < 0 || shard >= g->n) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "shard %d out of range (%d shards)", shard, g->n);
    if (window) *window = g->win[static_cast<size_t>(shard)];
    if (device) *device = g->device[static_cast<size_t>(shard)];
  });
}

// ---- images: one pyramid per distinct device of the group --------------------------------------------------------------------------

int dsopp_hip_pyramid_group_create(dsopp_hip_window_group *g, int width, int height, int levels, dsopp_hip_pyramid_group **out) {
  return guarded([&] {
    checkGroup(g);
    if (!out) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    auto pg = std::make_unique<dsopp_hip_pyramid_group>();
    pg->group = g;
    pg->of_shard.resize(static_cast<size_t>(g->n));
    for (int s = 0; s < g->n; ++s) {
      const int dev = g->device[static_cast<size_t>(s)];
      int k = -1;
      for (size_t j = 0; j < pg->device.size(); ++j)
        if (pg->device[j] == dev) k = static_cast<int>(j);
      if (k < 0) {
        k = static_cast<int>(pg->device.size());
        pg->device.push_back(dev);
        pg->pyramid.push_back(nullptr);
      }
      pg->of_shard[static_cast<size_t>(s)] = k;
    }
    for (size_t k = 0; k < pg->device.size(); ++k) {
      const int rc = dsopp_hip_pyramid_create(pg->device[k], nullptr, width, height, levels, g->opt.dtype, &pg->pyramid[k]);
      if (rc != DSOPP_HIP_OK) {
        const std::string msg = dsopp_hip_last_error();
        for (auto *p : pg->pyramid) dsopp_hip_pyramid_destroy(p);
        fail(rc, "%s", msg.c_str());
      }
    }
    *out = pg.release();
  });
}

void dsopp_hip_pyramid_group_destroy(dsopp_hip_pyramid_group *pg) {
  if (!pg) return;
  for (auto *p : pg->pyramid) dsopp_hip_pyramid_destroy(p);
  delete pg;
}

#define DSOPP_PYRAMID_GROUP_FORALL(call)                                                      \
  return guarded([&] {                                                                        \
    if (!pg) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null pyramid group");                      \
    for (dsopp_hip_pyramid * p : pg->pyramid) {                                               \
      const int rc = (call);                                                                  \
      if (rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());                         \
    }                                                                                         \
  })

int dsopp_hip_pyramid_group_build(dsopp_hip_pyramid_group *pg, const uint8_t *image_host, const double *lut256, const uint8_t *vignetting_host) {
  // the same u8 image goes to every device and every device builds its own levels (1.3 MB over PCIe + a 35 us kernel chain per
  // device at 1280 x 1024) — cheaper than building once and broadcasting 56 MB of texels over xGMI
  DSOPP_PYRAMID_GROUP_FORALL(dsopp_hip_pyramid_build(p, image_host, lut256, vignetting_host));
}
int dsopp_hip_pyramid_group_set_level(dsopp_hip_pyramid_group *pg, int level, const double *pixelinfo_host) {
  DSOPP_PYRAMID_GROUP_FORALL(dsopp_hip_pyramid_set_level(p, level, pixelinfo_host));
}
int dsopp_hip_pyramid_group_set_mask(dsopp_hip_pyramid_group *pg, int level, const uint8_t *mask_host) {
  DSOPP_PYRAMID_GROUP_FORALL(dsopp_hip_pyramid_set_mask(p, level, mask_host));
}
int dsopp_hip_pyramid_group_get(dsopp_hip_pyramid_group *pg, int32_t shard, dsopp_hip_pyramid **pyramid) {
  return guarded([&] {
    if (!pg || !pyramid) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (shard < 0 || shard >= pg->group->n) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "shard %d out of range", shard);
    *pyramid = pg->pyramid[static_cast<size_t>(pg->of_shard[static_cast<size_t>(shard)])];
  });
}

// ---- window calls: replicated -------------------------------------------------------------------------------------------------------

int dsopp_hip_window_group_push_frame(dsopp_hip_window_group *g, int32_t frame_id, int64_t timestamp, const dsopp_hip_pyramid_group *pyramids,
                                      int level, const double intrinsics[4], const double T_world_agent[7], double exposure_time,
                                      const double affine_brightness[2], int fixed, int is_marginalized) {
  return guarded([&] {
    checkGroup(g);
    if (!pyramids || pyramids->group != g) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "the pyramid group belongs to another window group");
    // (with shards, the fold-in of the marginalised frames inside pushFrame is itself a collective: every shard takes part)
    fanOut(*g, [&](int s, dsopp_hip_window *w) {
      return dsopp_hip_window_push_frame(w, frame_id, timestamp, pyramids->pyramid[static_cast<size_t>(pyramids->of_shard[static_cast<size_t>(s)])], level,
                                         intrinsics, T_world_agent, exposure_time, affine_brightness, fixed, is_marginalized);
    });
  });
}

#define DSOPP_GROUP_REPLICATED(expr)             \
  return guarded([&] {                           \
    checkGroup(g);                               \
    fanOut(*g, [&](int s, dsopp_hip_window *w) { \
      (void)s;                                   \
      return (expr);                             \
    });                                          \
  })

int dsopp_hip_window_group_mark_frame_marginalized(dsopp_hip_window_group *g, int32_t frame_id) {
  DSOPP_GROUP_REPLICATED(dsopp_hip_window_mark_frame_marginalized(w, frame_id));
}
int dsopp_hip_window_group_begin(dsopp_hip_window_group *g) { DSOPP_GROUP_REPLICATED(dsopp_hip_window_begin(w)); }
int dsopp_hip_window_group_linearize(dsopp_hip_window_group *g) { DSOPP_GROUP_REPLICATED(dsopp_hip_window_linearize(w)); }
int dsopp_hip_window_group_reject_step(dsopp_hip_window_group *g) { DSOPP_GROUP_REPLICATED(dsopp_hip_window_reject_step(w)); }
int dsopp_hip_window_group_update_point_statuses(dsopp_hip_window_group *g) { DSOPP_GROUP_REPLICATED(dsopp_hip_window_update_point_statuses(w)); }
int dsopp_hip_window_group_set_lm_mode(dsopp_hip_window_group *g, int mode) { DSOPP_GROUP_REPLICATED(dsopp_hip_window_set_lm_mode(w, mode)); }
int dsopp_hip_window_group_set_deterministic(dsopp_hip_window_group *g, int enable) { DSOPP_GROUP_REPLICATED(dsopp_hip_window_set_deterministic(w, enable)); }
int dsopp_hip_window_group_set_max_iterations(dsopp_hip_window_group *g, int32_t max_iterations) {
  DSOPP_GROUP_REPLICATED(dsopp_hip_window_set_max_iterations(w, max_iterations));
}
int dsopp_hip_window_group_snapshot(dsopp_hip_window_group *g) { DSOPP_GROUP_REPLICATED(dsopp_hip_window_snapshot(w)); }
int dsopp_hip_window_group_restore(dsopp_hip_window_group *g) { DSOPP_GROUP_REPLICATED(dsopp_hip_window_restore(w)); }

/** calls whose scalar results are identical on every shard (they follow the collective): shard 0's are returned */
int dsopp_hip_window_group_solve(dsopp_hip_window_group *g, double *energy, int32_t *iterations, int32_t *n_valid) {
  return guarded([&] {
    checkGroup(g);
    fanOut(*g, [&](int s, dsopp_hip_window *w) { return dsopp_hip_window_solve(w, s ? nullptr : energy, s ? nullptr : iterations, s ? nullptr : n_valid); });
  });
}
int dsopp_hip_window_group_optimize(dsopp_hip_window_group *g, double *energy, int32_t *iterations, int32_t *n_valid) {
  return guarded([&] {
    checkGroup(g);
    fanOut(*g, [&](int s, dsopp_hip_window *w) { return dsopp_hip_window_optimize(w, s ? nullptr : energy, s ? nullptr : iterations, s ? nullptr : n_valid); });
  });
}
int dsopp_hip_window_group_optimize_repeated(dsopp_hip_window_group *g, int32_t iterations_target, int32_t *iterations_done, double *last_energy) {
  return guarded([&] {
    checkGroup(g);
    fanOut(*g, [&](int s, dsopp_hip_window *w) {
      return dsopp_hip_window_optimize_repeated(w, iterations_target, s ? nullptr : iterations_done, s ? nullptr : last_energy);
    });
  });
}
int dsopp_hip_window_group_calculate_energy(dsopp_hip_window_group *g, double *energy, int32_t *n_valid) {
  return guarded([&] {
    checkGroup(g);
    fanOut(*g, [&](int s, dsopp_hip_window *w) { return dsopp_hip_window_calculate_energy(w, s ? nullptr : energy, s ? nullptr : n_valid); });
  });
}
int dsopp_hip_window_group_calculate_step(dsopp_hip_window_group *g, double lambda, double *step) {
  return guarded([&] {
    checkGroup(g);
    fanOut(*g, [&](int s, dsopp_hip_window *w) { return dsopp_hip_window_calculate_step(w, lambda, s ? nullptr : step); });
  });
}
int dsopp_hip_window_group_accept_step(dsopp_hip_window_group *g, double *state_sq, double *step_sq) {
  return guarded([&] {
    checkGroup(g);
    fanOut(*g, [&](int s, dsopp_hip_window *w) { return dsopp_hip_window_accept_step(w, s ? nullptr : state_sq, s ? nullptr : step_sq); });
  });
}

/** replicated state: shard 0 answers */
int dsopp_hip_window_group_num_frames(dsopp_hip_window_group *g, int32_t *n) {
  return guarded([&] {
    checkGroup(g);
    if (const int rc = dsopp_hip_window_num_frames(g->win[0], n); rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());
  });
}
int dsopp_hip_window_group_frame_ids(dsopp_hip_window_group *g, int32_t capacity, int32_t *ids, int32_t *n) {
  return guarded([&] {
    checkGroup(g);
    if (const int rc = dsopp_hip_window_frame_ids(g->win[0], capacity, ids, n); rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());
  });
}
int dsopp_hip_window_group_get_system(dsopp_hip_window_group *g, double *H_pp, double *b_pp, double *H_schur, double *b_schur) {
  return guarded([&] {
    checkGroup(g);
    if (const int rc = dsopp_hip_window_get_system(g->win[0], H_pp, b_pp, H_schur, b_schur); rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());
  });
}
int dsopp_hip_window_group_get_frame_state(dsopp_hip_window_group *g, int32_t frame_id, double T0[7], double ab0[2], double eps[8], double step[8]) {
  return guarded([&] {
    checkGroup(g);
    if (const int rc = dsopp_hip_window_get_frame_state(g->win[0], frame_id, T0, ab0, eps, step); rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());
  });
}
int dsopp_hip_window_group_get_pose(dsopp_hip_window_group *g, int32_t frame_id, double T_world_agent[7], double affine_brightness[2]) {
  return guarded([&] {
    checkGroup(g);
    if (const int rc = dsopp_hip_window_get_pose(g->win[0], frame_id, T_world_agent, affine_brightness); rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());
  });
}
int dsopp_hip_window_group_get_marginalized(dsopp_hip_window_group *g, double *H, double *b, double *energy, int32_t *size) {
  return guarded([&] {
    checkGroup(g);
    if (const int rc = dsopp_hip_window_get_marginalized(g->win[0], H, b, energy, size); rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());
  });
}
int dsopp_hip_window_group_get_covariance(dsopp_hip_window_group *g, int32_t reference_id, int32_t target_id, double cov[36]) {
  return guarded([&] {
    checkGroup(g);
    if (const int rc = dsopp_hip_window_get_covariance(g->win[0], reference_id, target_id, cov); rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());
  });
}
int dsopp_hip_window_group_last_solve_ms(dsopp_hip_window_group *g, float *ms) {
  return guarded([&] {
    checkGroup(g);
    if (const int rc = dsopp_hip_window_last_solve_ms(g->win[0], ms); rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());
  });
}

// ---- window calls: dealt over the shards ------------------------------------------------------------------------------------------------

int dsopp_hip_window_group_num_landmarks(dsopp_hip_window_group *g, int32_t frame_id, int32_t *n) {
  return guarded([&] {
    checkGroup(g);
    if (!n) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    int total = 0;
    for (int s = 0; s < g->n; ++s) {
      int32_t k = 0;
      if (const int rc = dsopp_hip_window_num_landmarks(g->win[static_cast<size_t>(s)], frame_id, &k); rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());
      total += k;
    }
    *n = total;
  });
}

int dsopp_hip_window_group_set_landmarks(dsopp_hip_window_group *g, int32_t frame_id, int32_t n_total, const double *uv, const double *idepth,
                                         const double *patch, const uint8_t *flags) {
  return guarded([&] {
    checkGroup(g);
    if (n_total < 0 || (n_total && (!uv || !idepth || !patch || !flags))) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    const int S = g->n;
    fanOut(*g, [&](int s, dsopp_hip_window *w) {
      int32_t have = 0;
      if (const int rc = dsopp_hip_window_num_landmarks(w, frame_id, &have); rc != DSOPP_HIP_OK) return rc;
      const int ns = shardCount(n_total, s, S);
      G::Scratch &sc = g->scratch[static_cast<size_t>(s)];
      // the window reads flags of all its landmarks but coordinates / idepth / patch of the new ones [have, ns) only: marshal just those
      sc.u8.resize(static_cast<size_t>(ns));
      sc.a.resize(2 * static_cast<size_t>(ns));
      sc.b.resize(static_cast<size_t>(ns));
      sc.c.resize(static_cast<size_t>(DSOPP_HIP_PATTERN_SIZE) * static_cast<size_t>(ns));
      for (int k = 0; k < ns; ++k) sc.u8[static_cast<size_t>(k)] = flags[static_cast<size_t>(k) * S + s];
      for (int k = std::min<int>(have, ns); k < ns; ++k) {
        const size_t j = static_cast<size_t>(k) * S + s;
        sc.a[2 * static_cast<size_t>(k)] = uv[2 * j];
        sc.a[2 * static_cast<size_t>(k) + 1] = uv[2 * j + 1];
        sc.b[static_cast<size_t>(k)] = idepth[j];
        std::memcpy(&sc.c[static_cast<size_t>(DSOPP_HIP_PATTERN_SIZE) * k], patch + DSOPP_HIP_PATTERN_SIZE * j, DSOPP_HIP_PATTERN_SIZE * sizeof(double));
      }
      return dsopp_hip_window_set_landmarks(w, frame_id, ns, sc.a.data(), sc.b.data(), sc.c.data(), sc.u8.data());
    });
  });
}

int dsopp_hip_window_group_set_connection(dsopp_hip_window_group *g, int32_t reference_id, int32_t target_id, int32_t n, const uint8_t *statuses) {
  return guarded([&] {
    checkGroup(g);
    if (n < 0 || (n && !statuses)) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    const int S = g->n;
    fanOut(*g, [&](int s, dsopp_hip_window *w) {
      const int ns = shardCount(n, s, S);
      G::Scratch &sc = g->scratch[static_cast<size_t>(s)];
      sc.u8.resize(static_cast<size_t>(std::max(ns, 1)));
      for (int k = 0; k < ns; ++k) sc.u8[static_cast<size_t>(k)] = statuses[static_cast<size_t>(k) * S + s];
      return dsopp_hip_window_set_connection(w, reference_id, target_id, ns, sc.u8.data());
    });
  });
}

int dsopp_hip_window_group_get_landmarks(dsopp_hip_window_group *g, int32_t frame_id, double *idepth, double *idepth_step, double *inv_hessian_idepth,
                                         double *b_idepth, double *relative_baseline, int32_t *n_inliers, uint8_t *flags_out, double *hpib) {
  return guarded([&] {
    checkGroup(g);
    const int S = g->n;
    int32_t F = 0;
    if (const int rc = dsopp_hip_window_num_frames(g->win[0], &F); rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());
    const size_t K = static_cast<size_t>(DSOPP_HIP_BLOCK_SIZE) * static_cast<size_t>(F);
    fanOut(*g, [&](int s, dsopp_hip_window *w) {
      int32_t ns = 0;
      if (const int rc = dsopp_hip_window_num_landmarks(w, frame_id, &ns); rc != DSOPP_HIP_OK) return rc;
      const size_t n = static_cast<size_t>(ns);
      G::Scratch &sc = g->scratch[static_cast<size_t>(s)];
      sc.a.resize(n);
      sc.b.resize(n);
      sc.c.resize(n);
      sc.d.resize(n);
      sc.e.resize(n);
      sc.i32.resize(n);
      sc.u8.resize(n);
      if (hpib) sc.h.resize(n * K);
      const int rc = dsopp_hip_window_get_landmarks(w, frame_id, idepth ? sc.a.data() : nullptr, idepth_step ? sc.b.data() : nullptr,
                                                    inv_hessian_idepth ? sc.c.data() : nullptr, b_idepth ? sc.d.data() : nullptr,
                                                    relative_baseline ? sc.e.data() : nullptr, n_inliers ? sc.i32.data() : nullptr,
                                                    flags_out ? sc.u8.data() : nullptr, hpib ? sc.h.data() : nullptr);
      if (rc != DSOPP_HIP_OK) return rc;
      for (size_t k = 0; k < n; ++k) {
        const size_t j = k * static_cast<size_t>(S) + static_cast<size_t>(s);
        if (idepth) idepth[j] = sc.a[k];
        if (idepth_step) idepth_step[j] = sc.b[k];
        if (inv_hessian_idepth) inv_hessian_idepth[j] = sc.c[k];
        if (b_idepth) b_idepth[j] = sc.d[k];
        if (relative_baseline) relative_baseline[j] = sc.e[k];
        if (n_inliers) n_inliers[j] = sc.i32[k];
        if (flags_out) flags_out[j] = sc.u8[k];
        if (hpib) std::memcpy(hpib + j * K, sc.h.data() + k * K, K * sizeof(double));
      }
      return static_cast<int>(DSOPP_HIP_OK);
    });
  });
}

int dsopp_hip_window_group_get_frame_update(dsopp_hip_window_group *g, int32_t frame_id, double *idepth, double *inv_hessian_idepth,
                                            double *relative_baseline, int32_t *n_inliers, uint8_t *flags_out, int32_t n_targets,
                                            const int32_t *target_ids, uint8_t *statuses) {
  return guarded([&] {
    checkGroup(g);
    if (n_targets < 0 || n_targets > DSOPP_HIP_MAX_FRAMES || (n_targets && (!target_ids || !statuses))) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad argument");
    const int S = g->n;
    int32_t n_all = 0;
    if (const int rc = dsopp_hip_window_group_num_landmarks(g, frame_id, &n_all); rc != DSOPP_HIP_OK) fail(rc, "%s", dsopp_hip_last_error());
    fanOut(*g, [&](int s, dsopp_hip_window *w) {
      int32_t ns = 0;
      if (const int rc = dsopp_hip_window_num_landmarks(w, frame_id, &ns); rc != DSOPP_HIP_OK) return rc;
      const size_t n = static_cast<size_t>(ns);
      if (n == 0) return static_cast<int>(DSOPP_HIP_OK);
      G::Scratch &sc = g->scratch[static_cast<size_t>(s)];
      sc.a.resize(n);
      sc.c.resize(n);
      sc.e.resize(n);
      sc.i32.resize(n);
      sc.u8.resize(n);
      sc.v8.resize(n * static_cast<size_t>(std::max(n_targets, 1)));
      const int rc = dsopp_hip_window_get_frame_update(w, frame_id, sc.a.data(), sc.c.data(), sc.e.data(), sc.i32.data(), sc.u8.data(), n_targets, target_ids,
                                                       sc.v8.data());
      if (rc != DSOPP_HIP_OK) return rc;
      for (size_t k = 0; k < n; ++k) {
        const size_t j = k * static_cast<size_t>(S) + static_cast<size_t>(s);
        if (idepth) idepth[j] = sc.a[k];
        if (inv_hessian_idepth) inv_hessian_idepth[j] = sc.c[k];
        if (relative_baseline) relative_baseline[j] = sc.e[k];
        if (n_inliers) n_inliers[j] = sc.i32[k];
        if (flags_out) flags_out[j] = sc.u8[k];
        for (int t = 0; t < n_targets; ++t) statuses[static_cast<size_t>(t) * static_cast<size_t>(n_all) + j] = sc.v8[static_cast<size_t>(t) * n + k];
      }
      return static_cast<int>(DSOPP_HIP_OK);
    });
  });
}

int dsopp_hip_window_group_get_residuals(dsopp_hip_window_group *g, int32_t reference_id, int32_t target_id, int32_t n, uint8_t *status,
                                         uint8_t *candidate, double *energy) {
  return guarded([&] {
    checkGroup(g);
    if (n < 0) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad argument");
    const int S = g->n;
    fanOut(*g, [&](int s, dsopp_hip_window *w) {
      const int ns = shardCount(n, s, S);
      G::Scratch &sc = g->scratch[static_cast<size_t>(s)];
      sc.u8.resize(static_cast<size_t>(std::max(ns, 1)));
      sc.v8.resize(static_cast<size_t>(std::max(ns, 1)));
      sc.a.resize(static_cast<size_t>(std::max(ns, 1)));
      const int rc = dsopp_hip_window_get_residuals(w, reference_id, target_id, ns, status ? sc.u8.data() : nullptr, candidate ? sc.v8.data() : nullptr,
                                                    energy ? sc.a.data() : nullptr);
      if (rc != DSOPP_HIP_OK) return rc;
      for (int k = 0; k < ns; ++k) {
        const size_t j = static_cast<size_t>(k) * S + s;
        if (status) status[j] = sc.u8[static_cast<size_t>(k)];
        if (candidate) candidate[j] = sc.v8[static_cast<size_t>(k)];
        if (energy) energy[j] = sc.a[static_cast<size_t>(k)];
      }
      return static_cast<int>(DSOPP_HIP_OK);
    });
  });
}

// ---- reference depth maps of the newest keyframe: every shard splats its own landmarks, the level-0 planes are summed across the shards
//      (pba.hip: fillReferenceDepthMaps), every shard then holds the whole maps; the caller receives shard 0's (device_ids[0], where the
//      tracker's aligner lives) -------------------------------------------------------------------------------------------------------------

int dsopp_hip_window_group_create_reference_depth_maps(dsopp_hip_window_group *g, int32_t levels, dsopp_hip_depth_maps **out) {
  return guarded([&] {
    checkGroup(g);
    if (!out) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    std::vector<dsopp_hip_depth_maps *> made(static_cast<size_t>(g->n), nullptr);
    try {
      fanOut(*g, [&](int s, dsopp_hip_window *w) { return dsopp_hip_window_create_reference_depth_maps(w, levels, &made[static_cast<size_t>(s)]); });
    } catch (...) {
      for (auto *m : made) dsopp_hip_depth_maps_destroy(m);
      throw;
    }
    // kept: a refill of these maps needs its destination on every shard
    auto drop = [&](dsopp_hip_window_group::ShadowSet &set) {
      fanOut(*g, [&](int s, dsopp_hip_window *) {
        if (s && set.of_shard[static_cast<size_t>(s)]) dsopp_hip_depth_maps_destroy(set.of_shard[static_cast<size_t>(s)]);
        return static_cast<int>(DSOPP_HIP_OK);
      });
    };
    for (size_t k = 0; k < g->shadow_sets.size();) {  // (an address the allocator handed out again: the old object is gone)
      if (g->shadow_sets[k].key == made[0]) {
        drop(g->shadow_sets[k]);
        g->shadow_sets.erase(g->shadow_sets.begin() + static_cast<long>(k));
      } else {
        ++k;
      }
    }
    if (g->shadow_sets.size() >= dsopp_hip_window_group::kMaxShadowSets) {
      drop(g->shadow_sets.front());
      g->shadow_sets.erase(g->shadow_sets.begin());
    }
    g->shadow_sets.push_back({made[0], made});
    *out = made[0];
  });
}

int dsopp_hip_window_group_refill_reference_depth_maps(dsopp_hip_window_group *g, dsopp_hip_depth_maps *maps) {
  return guarded([&] {
    checkGroup(g);
    if (!maps) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    const dsopp_hip_window_group::ShadowSet *set = nullptr;
    for (const auto &c : g->shadow_sets)
      if (c.key == maps) set = &c;
    if (!set) fail(DSOPP_HIP_ERR_STATE, "refill needs maps made by dsopp_hip_window_group_create_reference_depth_maps of this group (one of its last %d calls)",
                   static_cast<int>(dsopp_hip_window_group::kMaxShadowSets));
    fanOut(*g, [&](int s, dsopp_hip_window *w) { return dsopp_hip_window_refill_reference_depth_maps(w, s ? set->of_shard[static_cast<size_t>(s)] : maps); });
  });
}

}  // extern "C"
