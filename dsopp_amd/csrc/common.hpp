// Common host-side helpers of the HIP library: error reporting, RAII device buffers.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdlib>
#include <map>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/dsopp_hip.h"

namespace dsopp_hip {

/** thread-local last error message returned by dsopp_hip_last_error() */
std::string &lastError();

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  throw Error(code, buf);
}

#define HIP_CHECK(expr)                                                                                     \
  do {                                                                                                      \
    hipError_t _e = (expr);                                                                                 \
    if (_e != hipSuccess)                                                                                   \
      ::dsopp_hip::fail(DSOPP_HIP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

/** hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): the attribute is per device, handles may live on
 *  different devices and be driven from different threads */
inline void ensureDynamicLds(const void *kernel, int device, int bytes) {
  static std::mutex mtx;
  static std::set<std::pair<const void *, int>> done;
  std::lock_guard<std::mutex> lock(mtx);
  if (done.count({kernel, device})) return;
  HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  done.insert({kernel, device});
}

/** run `body`, translate exceptions into the C-ABI's error codes */
template <typename F>
int guarded(F &&body) {
  try {
    body();
    return DSOPP_HIP_OK;
  } catch (const Error &e) {
    lastError() = e.what();
    return e.code;
  } catch (const std::exception &e) {
    lastError() = e.what();
    return DSOPP_HIP_ERR_INVALID_ARGUMENT;
  }
}

/** Tuning aid (DSOPP_HIP_HOST_TIMES=1): host wall time per named region of the library, summed over the process and printed at exit — where a
 *  blocking entry point's time goes between enqueueing, waiting for the device and host arithmetic (scripts/gpu_r6_keyframe_trace.sh).
 *  Without the variable a region costs one predictable branch. */
struct HostTimes {
  static bool on() {
    static const bool v = std::getenv("DSOPP_HIP_HOST_TIMES") != nullptr;
    return v;
  }
  struct Acc {
    double seconds = 0, longest = 0;
    long calls = 0;
  };
  static std::map<std::string, Acc> &table() {
    static std::map<std::string, Acc> *t = [] {
      auto *m = new std::map<std::string, Acc>();
      std::atexit([] {
        if (!HostTimes::on()) return;
        for (const auto &kv : HostTimes::table())
          std::fprintf(stderr, "[host times] %-44s %9.3f ms in %7ld calls = %8.2f us per call (longest %9.2f us, the others %8.2f us per call)\n",
                       kv.first.c_str(), kv.second.seconds * 1e3, kv.second.calls, kv.second.calls ? kv.second.seconds * 1e6 / kv.second.calls : 0.0,
                       kv.second.longest * 1e6, kv.second.calls > 1 ? (kv.second.seconds - kv.second.longest) * 1e6 / (kv.second.calls - 1) : 0.0);
      });
      return m;
    }();
    return *t;
  }
  static std::mutex &tableMutex() {  // (ONE lock for the table: regions close and growths are counted from a window group's worker threads too)
    static std::mutex m;
    return m;
  }
  /** one event without a duration (e.g. a device buffer growing: who asked, counted per call site) */
  static void count(const std::string &what, long by = 1) {
    if (!on()) return;
    std::lock_guard<std::mutex> lock(tableMutex());
    table()[what].calls += by;
  }
  const char *name;
  std::chrono::steady_clock::time_point t0;
  explicit HostTimes(const char *n) : name(n) {
    if (on()) t0 = std::chrono::steady_clock::now();
  }
  ~HostTimes() {
    if (!on()) return;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::lock_guard<std::mutex> lock(tableMutex());
    Acc &a = table()[name];
    a.seconds += dt;
    a.longest = dt > a.longest ? dt : a.longest;
    a.calls += 1;
  }
};

/** growable device buffer (never shrinks); contents preserved on growth */
template <typename T>
struct DeviceBuffer {
  T *ptr = nullptr;
  size_t capacity = 0;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer &) = delete;
  DeviceBuffer &operator=(const DeviceBuffer &) = delete;
  DeviceBuffer(DeviceBuffer &&o) noexcept : ptr(o.ptr), capacity(o.capacity) { o.ptr = nullptr; o.capacity = 0; }
  DeviceBuffer &operator=(DeviceBuffer &&o) noexcept {
    if (this != &o) {
      release();
      ptr = o.ptr;
      capacity = o.capacity;
      o.ptr = nullptr;
      o.capacity = 0;
    }
    return *this;
  }
  ~DeviceBuffer() { release(); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    capacity = 0;
  }
  /** ensure capacity >= n elements; keeps the first `keep` elements */
  void reserve(size_t n, size_t keep, hipStream_t stream, const char *who = __builtin_FUNCTION(), int line = __builtin_LINE()) {
    if (n <= capacity) return;
    if (HostTimes::on()) HostTimes::count(std::string("device buffer grows: ") + who + ":" + std::to_string(line) + (ptr ? " (re-allocation)" : " (first)"));
    size_t cap = capacity ? capacity : 64;
    while (cap < n) cap *= 2;
    T *np = nullptr;
    HIP_CHECK(hipMalloc(&np, cap * sizeof(T)));
    HIP_CHECK(hipMemsetAsync(np, 0, cap * sizeof(T), stream));
    if (ptr && keep) HIP_CHECK(hipMemcpyAsync(np, ptr, keep * sizeof(T), hipMemcpyDeviceToDevice, stream));
    if (ptr) {
      HIP_CHECK(hipStreamSynchronize(stream));
      (void)hipFree(ptr);
    }
    ptr = np;
    capacity = cap;
  }
  void upload(const T *host, size_t n, size_t offset, hipStream_t stream) {
    if (n) HIP_CHECK(hipMemcpyAsync(ptr + offset, host, n * sizeof(T), hipMemcpyHostToDevice, stream));
  }
  void download(T *host, size_t n, size_t offset, hipStream_t stream) const {
    if (n) HIP_CHECK(hipMemcpyAsync(host, ptr + offset, n * sizeof(T), hipMemcpyDeviceToHost, stream));
  }
};

/** owns (or borrows) a stream on a device */
struct StreamRef {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owned = false;
  void init(int dev, void *user_stream) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
      fail(DSOPP_HIP_ERR_HIP, "no HIP device available (this library has no CPU fallback)");
    if (dev < 0 || dev >= count) fail(DSOPP_HIP_ERR_INVALID_ARGUMENT, "device %d out of range (%d devices)", dev, count);
    device = dev;
    HIP_CHECK(hipSetDevice(dev));
    if (user_stream) {
      stream = static_cast<hipStream_t>(user_stream);
      owned = false;
    } else {
      HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
      owned = true;
    }
  }
  void destroy() {
    if (owned && stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
  }
  void use() const { HIP_CHECK(hipSetDevice(device)); }
  void sync() const { HIP_CHECK(hipStreamSynchronize(stream)); }
};

}  // namespace dsopp_hip
