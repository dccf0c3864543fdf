// Error value type used by every native API in this tree.
//
// Parity: reference `src/c++/library/common.h:61-83` (`triton::client::Error`):
// a default-constructed / `Success` value means OK, anything else carries a
// message.  Here the HTTP-ish status code travels with the message so that the
// HTTP client can surface 4xx/5xx and the synthetic 499 deadline code.
#pragma once

#include <ostream>
#include <string>
#include <utility>

namespace tca {

class Error {
 public:
  Error() = default;
  explicit Error(std::string msg, int code = -1)
      : ok_(false), code_(code), msg_(std::move(msg)) {}

  static const Error& Success() {
    static const Error ok;
    return ok;
  }

  bool IsOk() const { return ok_; }
  explicit operator bool() const { return !ok_; }  // true == failure
  const std::string& Message() const { return msg_; }
  int Code() const { return code_; }

  friend std::ostream& operator<<(std::ostream& os, const Error& e) {
    return e.ok_ ? (os << "OK") : (os << "error: " << e.msg_);
  }

 private:
  bool ok_ = true;
  int code_ = 0;
  std::string msg_;
};

#define TCA_RETURN_IF_ERROR(expr)   \
  do {                              \
    const ::tca::Error _e = (expr); \
    if (!_e.IsOk()) return _e;      \
  } while (0)

}  // namespace tca
