// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the shipped product path.
//
// Minimal persistent thread pool standing in for the reference's TBB parallel_for / task_group
// (call sites listed in SURVEY.md §2).  Scheduling only — no arithmetic lives here.  The reference caps the
// worker count at clamp(hw,1,8)-1 (src/application/dsopp_main.cpp:114-119); the oracle takes the count as a knob.
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace oracle {

class ThreadPool {
 public:
  static ThreadPool &instance() {
    static ThreadPool pool;
    return pool;
  }
  /** number of threads participating in parallelFor (including the caller) */
  void setThreads(int n) {
    n = std::max(1, n);
    if (n == threads_) return;
    shutdown();
    threads_ = n;
    stop_ = false;
    for (int i = 0; i < n - 1; ++i) workers_.emplace_back([this] { workerLoop(); });
  }
  int threads() const { return threads_; }

  /** run fn(begin, end) over [0, n) split into chunks of `grain`, dynamically scheduled */
  void parallelFor(size_t n, size_t grain, const std::function<void(size_t, size_t)> &fn) {
    if (n == 0) return;
    if (threads_ <= 1 || n <= grain) {
      fn(0, n);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mutex_);
      job_ = &fn;
      job_n_ = n;
      job_grain_ = std::max<size_t>(1, grain);
      next_.store(0);
      pending_ = static_cast<int>(workers_.size());
      ++generation_;
    }
    cv_.notify_all();
    runChunks();
    std::unique_lock<std::mutex> lk(mutex_);
    done_cv_.wait(lk, [this] { return pending_ == 0; });
    job_ = nullptr;
  }
  ~ThreadPool() { shutdown(); }

 private:
  ThreadPool() = default;
  void runChunks() {
    while (true) {
      const size_t b = next_.fetch_add(job_grain_);
      if (b >= job_n_) break;
      (*job_)(b, std::min(job_n_, b + job_grain_));
    }
  }
  void workerLoop() {
    uint64_t seen = 0;
    while (true) {
      {
        std::unique_lock<std::mutex> lk(mutex_);
        cv_.wait(lk, [&] { return stop_ || generation_ != seen; });
        if (stop_) return;
        seen = generation_;
      }
      runChunks();
      {
        std::lock_guard<std::mutex> lk(mutex_);
        --pending_;
      }
      done_cv_.notify_one();
    }
  }
  void shutdown() {
    {
      std::lock_guard<std::mutex> lk(mutex_);
      stop_ = true;
    }
    cv_.notify_all();
    for (auto &w : workers_) w.join();
    workers_.clear();
  }
  int threads_ = 1;
  std::vector<std::thread> workers_;
  std::mutex mutex_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(size_t, size_t)> *job_ = nullptr;
  size_t job_n_ = 0, job_grain_ = 1;
  std::atomic<size_t> next_{0};
  int pending_ = 0;
  uint64_t generation_ = 0;
  bool stop_ = false;
};

inline void parallelFor(size_t n, size_t grain, const std::function<void(size_t, size_t)> &fn) {
  ThreadPool::instance().parallelFor(n, grain, fn);
}

}  // namespace oracle
