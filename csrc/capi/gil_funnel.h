// One thread that runs every Python callback of a native serving front.
//
// Reactor threads never take the GIL themselves: they queue closures here and go
// back to their sockets.  The funnel thread takes the GIL once per batch of queued
// closures, so N reactors under load cost one GIL hand-over per batch instead of
// one per request - with two reactors calling into Python directly, the hand-overs
// between them were measured to halve the throughput of microsecond-scale models.
// Closures run in the order they were queued (per-stream message order is kept).
//
// When nothing is queued and the funnel is idle, the calling reactor runs its closure
// itself (Run): a lone request does not pay a thread wake-up (measured: 527 us against
// 642 us per request at concurrency 1 through the funnel).  Inline runs and the funnel
// thread exclude each other, so at most one thread of the front is in Python at a time.
#pragma once

#define PY_SSIZE_T_CLEAN
#include <Python.h>

#include <atomic>
#include <condition_variable>
#include <cstdlib>
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

  // Runs `fn` with the GIL held: here, when the funnel has nothing queued or running, else queued.
  template <typename Fn>
  void Run(Fn&& fn) {
    bool idle = false;
    if (inline_ && busy_.compare_exchange_strong(idle, true, std::memory_order_acquire)) {
      bool empty;
      {
        std::lock_guard<std::mutex> lk(mu_);
        empty = queue_.empty() && !stop_;
      }
      if (empty) {
        PyGILState_STATE gil = PyGILState_Ensure();
        fn();
        PyGILState_Release(gil);
        busy_.store(false, std::memory_order_release);
        return;
      }
      busy_.store(false, std::memory_order_release);
    }
    Push(std::function<void()>(std::forward<Fn>(fn)));
  }

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
      }
      // an inline run may be in progress: it is a few tens of microseconds of Python
      for (bool idle = false; !busy_.compare_exchange_weak(idle, true, std::memory_order_acquire); idle = false)
        std::this_thread::yield();
      {
        std::lock_guard<std::mutex> lk(mu_);
        batch.swap(queue_);  // taken while busy_ is held: no inline run can overtake a queued closure
      }
      PyGILState_STATE gil = PyGILState_Ensure();
      for (auto& fn : batch) fn();
      batch.clear();  // closures may own Python references: drop them under the GIL
      PyGILState_Release(gil);
      busy_.store(false, std::memory_order_release);
    }
  }

  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> queue_;
  bool stop_ = false;
  std::atomic<bool> busy_{false};  // a closure batch (funnel thread) or an inline closure (a reactor) is running
  const bool inline_ = [] {  // TCA_FUNNEL_INLINE=0: every closure goes through the funnel thread (A/B switch)
    const char* env = std::getenv("TCA_FUNNEL_INLINE");
    return env == nullptr || env[0] != '0';
  }();
  std::thread thread_;
};

}  // namespace capi
}  // namespace tca
