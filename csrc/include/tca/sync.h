// Timed condition-variable wait that thread sanitizers can follow.
//
// `std::condition_variable::wait_for` waits on the steady clock, which libstdc++
// implements with pthread_cond_clockwait; the ThreadSanitizer runtime shipped
// with GCC 11 does not intercept that call, so it never sees the mutex being
// released during the wait and reports every such hand-over as a double lock
// plus data races.  Waiting until a system_clock time point goes through
// pthread_cond_timedwait, which is intercepted.  The timeouts in this code base
// are coarse safety nets (seconds), so a wall-clock adjustment during the wait
// is harmless.
#pragma once

#include <chrono>
#include <condition_variable>
#include <mutex>

namespace tca {

template <typename Rep, typename Period, typename Predicate>
bool WaitFor(std::condition_variable& cv, std::unique_lock<std::mutex>& lock,
             const std::chrono::duration<Rep, Period>& timeout, Predicate pred) {
  return cv.wait_until(lock, std::chrono::system_clock::now() + timeout, pred);
}

}  // namespace tca
