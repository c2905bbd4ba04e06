// dsopp_hip_comm_*: the multi-GPU exchange step of the hot path as native code — one RCCL communicator per process (one
// process per GPU), ncclAllReduce(sum, double) enqueued from C++ on the window's own stream.
//
// What it replaces: the reference reduces the per-thread partial systems of evaluateLinearSystemPosePose /
// ...PoseDepthSchurComplement under a mutex into one matrix (PBA_INT/hessian_block_evaluation.hpp:101-145,178-235).  With
// landmarks sharded across GPUs the same sum runs across ranks: ONE collective per Gauss-Newton iteration over the
// contiguous buffer [H_pp | b_pp | H_schur | b_schur | energy, n_valid, |step|^2, idepth.step] (pba.hip: launchReduceSchur).
//
// librccl is a 0.5 GB library: it is loaded lazily with dlopen on the first dsopp_hip_comm_* call, so a single-GPU process
// never maps it.  Inside a process that already holds an RCCL (PyTorch ships one with the same SONAME) the loaded instance
// is reused.
#include <atomic>
#include <chrono>
#include <thread>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <memory>
#include <mutex>

#include "common.hpp"

struct dsopp_hip_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  bool owned = true;
  // dsopp_hip_comm_abort is called from the failing shard's worker thread while other workers may be inside nativeAllreduce.
  // ncclCommAbort frees the communicator, so it must not run while another thread is between its `aborted` test and the return of its
  // ncclAllReduce: `in_flight` counts those threads; a thread that finds the flag set after it has announced itself backs out, and abort waits
  // for the count to drain before it hands the communicator to ncclCommAbort.  The wait is bounded (a worker can sit inside RCCL's enqueue
  // while a connection comes up: the abort exists to release exactly such workers) — after the bound the abort goes ahead, which is what
  // RCCL documents ncclCommAbort for.  `abort_ran` says whether RCCL really freed the communicator: destroy skips ncclCommDestroy only then.
  std::atomic<bool> aborted{false};
  std::atomic<bool> abort_ran{false};
  std::atomic<int> in_flight{0};
};

namespace dsopp_hip {
namespace {

struct RcclApi {
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  void *handle = nullptr;
};

RcclApi &rccl() {
  static RcclApi api;
  static std::once_flag once;
  static std::string load_error;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names)  // an instance the process already holds (PyTorch's) first
      if ((api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!api.handle)
      for (const char *n : names)
        if ((api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!api.handle) {
      load_error = std::string("librccl not found: ") + dlerror();
      return;
    }
    auto sym = [&](const char *name) {
      void *p = dlsym(api.handle, name);
      if (!p) load_error = std::string("librccl lacks ") + name;
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(api.handle, "ncclCommAbort"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
  });
  if (!load_error.empty()) fail(DSOPP_HIP_ERR_HIP, "%s", load_error.c_str());
  return api;
}

void rcclCheck(ncclResult_t r, const char *what) {
  if (r != ncclSuccess) fail(DSOPP_HIP_ERR_HIP, "%s failed: %s", what, rccl().GetErrorString(r));
}

}  // namespace

/** the dsopp_hip_allreduce_fn the window calls when a native communicator is attached (pba.hip: allreduceIfNeeded) */
int nativeAllreduce(void *user, void *device_buffer, size_t count, void *stream) {
  auto *c = static_cast<dsopp_hip_comm *>(user);
  struct InFlight {
    std::atomic<int> &n;
    explicit InFlight(std::atomic<int> &x) : n(x) { n.fetch_add(1, std::memory_order_acq_rel); }
    ~InFlight() { n.fetch_sub(1, std::memory_order_acq_rel); }
  } announced(c->in_flight);
  if (c->aborted.load(std::memory_order_acquire)) {  // (tested AFTER the announcement: abort either sees this thread or this thread sees the flag)
    lastError() = "communicator was aborted (dsopp_hip_comm_abort): the collective is not enqueued";
    return -3;
  }
  const ncclResult_t r = rccl().AllReduce(device_buffer, device_buffer, count, ncclDouble, ncclSum, c->comm, static_cast<hipStream_t>(stream));
  if (r != ncclSuccess) {
    lastError() = std::string("ncclAllReduce failed: ") + rccl().GetErrorString(r);
    return static_cast<int>(r);
  }
  return 0;
}

}  // namespace dsopp_hip

using namespace dsopp_hip;

extern "C" {

int dsopp_hip_comm_unique_id(uint8_t id[DSOPP_HIP_COMM_ID_BYTES]) {
  return guarded([&] {
    if (!id) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null id");
    static_assert(DSOPP_HIP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId uid;
    rcclCheck(rccl().GetUniqueId(&uid), "ncclGetUniqueId");
    std::memcpy(id, uid.internal, NCCL_UNIQUE_ID_BYTES);
  });
}

int dsopp_hip_comm_create(const uint8_t id[DSOPP_HIP_COMM_ID_BYTES], int rank, int world_size, int device, dsopp_hip_comm **out) {
  return guarded([&] {
    if (!id || !out) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (world_size < 1 || rank < 0 || rank >= world_size) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "bad rank %d / world %d", rank, world_size);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) fail(DSOPP_HIP_ERR_HIP, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= count) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", device, count);
    HIP_CHECK(hipSetDevice(device));
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    auto c = std::make_unique<dsopp_hip_comm>();
    rcclCheck(rccl().CommInitRank(&c->comm, world_size, uid, rank), "ncclCommInitRank");
    c->rank = rank;
    c->world = world_size;
    c->device = device;
    *out = c.release();
  });
}

int dsopp_hip_comm_adopt(void *nccl_comm, int device, dsopp_hip_comm **out) {
  return guarded([&] {
    if (!nccl_comm || !out) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) fail(DSOPP_HIP_ERR_HIP, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= count) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", device, count);
    HIP_CHECK(hipSetDevice(device));
    // The handle is passed to the librccl instance THIS library resolved (the one already loaded in the process, else /opt/rocm's): it
    // must have been created by that same instance — a communicator of a statically linked or differently named RCCL build is a
    // foreign object here (include/dsopp_hip.h says so)
    auto c = std::make_unique<dsopp_hip_comm>();
    c->comm = static_cast<ncclComm_t>(nccl_comm);
    c->owned = false;
    c->device = device;
    rcclCheck(rccl().CommCount(c->comm, &c->world), "ncclCommCount");
    rcclCheck(rccl().CommUserRank(c->comm, &c->rank), "ncclCommUserRank");
    *out = c.release();
  });
}

int dsopp_hip_comm_abort(dsopp_hip_comm *c) {
  return guarded([&] {
    if (!c) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null communicator");
    if (c->aborted.exchange(true, std::memory_order_acq_rel)) return;
    // ncclCommAbort releases the kernels of collectives the OTHER ranks already enqueued and that would wait for this rank for ever
    // (a rank that failed between two collectives never enqueues its side); afterwards the communicator can only be destroyed
    if (c->owned && c->comm && rccl().CommAbort) {
      // threads that passed the flag test before it was set are still inside ncclAllReduce with this handle: let them return first
      // (an enqueue takes microseconds; bounded, see the struct's comment)
      const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(2);
      while (c->in_flight.load(std::memory_order_acquire) > 0 && std::chrono::steady_clock::now() < deadline) std::this_thread::yield();
      (void)hipSetDevice(c->device);
      (void)rccl().CommAbort(c->comm);  // (ncclCommAbort frees the communicator's resources: destroy skips ncclCommDestroy)
      c->abort_ran.store(true, std::memory_order_release);
    }
  });
}

void dsopp_hip_comm_destroy(dsopp_hip_comm *c) {
  if (!c) return;
  // (every user of the handle has returned before the owner destroys it — the window / group detach first; a collective still inside
  // RCCL here would be a caller error, as with any handle of this API)
  if (c->owned && c->comm && !c->abort_ran.load(std::memory_order_acquire)) {  // aborted without ncclCommAbort in this RCCL: still ours to destroy
    (void)hipSetDevice(c->device);
    try {
      (void)rccl().CommDestroy(c->comm);
    } catch (...) {
    }
  }
  delete c;
}

int dsopp_hip_comm_rank(const dsopp_hip_comm *c, int *rank, int *world_size) {
  return guarded([&] {
    if (!c) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null communicator");
    if (rank) *rank = c->rank;
    if (world_size) *world_size = c->world;
  });
}

int dsopp_hip_comm_allreduce(dsopp_hip_comm *c, void *device_buffer, size_t count, void *stream) {
  return guarded([&] {
    if (!c || (!device_buffer && count)) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "null argument");
    if (nativeAllreduce(c, device_buffer, count, stream) != 0) fail(DSOPP_HIP_ERR_HIP, "%s", lastError().c_str());
  });
}

}  // extern "C"
