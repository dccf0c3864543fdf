// One thread that runs every Python callback of a native serving front.
//
// Reactor threads never take the GIL themselves: they queue closures here and go
// back to their sockets.  The funnel thread takes the GIL once per batch of queued
// closures, so N reactors under load cost one GIL hand-over per batch instead of
// one per request - with two reactors calling into Python directly, the hand-overs
// between them were measured to halve the throughput of microsecond-scale models.
// Closures run in the order they were queued (per-stream message order is kept).
#pragma once

#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

namespace tca {
namespace capi {

class GilFunnel {
 public:
  GilFunnel() : thread_([this] { Run(); }) {}
  ~GilFunnel() { Stop(); }

  // `fn` runs on the funnel thread with the GIL held.
  void Push(std::function<void()>&& fn) {
    bool wake;
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (stop_) return;
      wake = queue_.empty();
      queue_.push_back(std::move(fn));
    }
    if (wake) cv_.notify_one();
  }

  // Runs what is queued, then ends the thread.  Call WITHOUT the GIL held.
  void Stop() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (stop_) return;
      stop_ = true;
    }
    cv_.notify_one();
    if (thread_.joinable()) thread_.join();
  }

 private:
  void Run() {
    std::deque<std::function<void()>> batch;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
        if (queue_.empty()) return;
        batch.swap(queue_);
      }
      PyGILState_STATE gil = PyGILState_Ensure();
      for (auto& fn : batch) fn();
      batch.clear();  // closures may own Python references: drop them under the GIL
      PyGILState_Release(gil);
    }
  }

  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> queue_;
  bool stop_ = false;
  std::thread thread_;
};

}  // namespace capi
}  // namespace tca
