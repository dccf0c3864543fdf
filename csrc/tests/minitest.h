// Minimal test harness (gtest is not part of this toolchain): TEST(name) {...}
// registers a case, EXPECT_* record failures, RunAllTests() runs everything
// (optionally filtered by a substring) and returns the process exit code.
#pragma once

#include <cstdio>
#include <cstring>
#include <functional>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace minitest {

struct Case {
  const char* name;
  std::function<void()> fn;
};
inline std::vector<Case>& Registry() {
  static std::vector<Case> r;
  return r;
}
inline int& Failures() {
  static int f = 0;
  return f;
}
struct Registrar {
  Registrar(const char* name, std::function<void()> fn) { Registry().push_back({name, std::move(fn)}); }
};

inline int RunAllTests(const char* filter = nullptr) {
  int failed_cases = 0, ran = 0;
  for (auto& c : Registry()) {
    if (filter && !std::strstr(c.name, filter)) continue;
    const int before = Failures();
    std::printf("[ RUN  ] %s\n", c.name);
    c.fn();
    ++ran;
    if (Failures() != before) {
      ++failed_cases;
      std::printf("[ FAIL ] %s\n", c.name);
    } else {
      std::printf("[  OK  ] %s\n", c.name);
    }
  }
  std::printf("%d case(s) run, %d failed\n", ran, failed_cases);
  return failed_cases == 0 ? 0 : 1;
}

}  // namespace minitest

#define TEST(name)                                                        \
  static void minitest_##name();                                          \
  static ::minitest::Registrar minitest_reg_##name(#name, minitest_##name); \
  static void minitest_##name()

#define EXPECT_TRUE(cond)                                                                  \
  do {                                                                                     \
    if (!(cond)) {                                                                         \
      ++::minitest::Failures();                                                            \
      std::cout << __FILE__ << ":" << __LINE__ << ": expected true: " #cond << std::endl;  \
    }                                                                                      \
  } while (0)

#define EXPECT_EQ(a, b)                                                                           \
  do {                                                                                            \
    auto&& _a = (a);                                                                              \
    auto&& _b = (b);                                                                              \
    if (!(_a == _b)) {                                                                            \
      ++::minitest::Failures();                                                                   \
      std::cout << __FILE__ << ":" << __LINE__ << ": " #a " == " #b " failed: [" << _a << "] vs [" \
                << _b << "]" << std::endl;                                                        \
    }                                                                                             \
  } while (0)

#define EXPECT_OK(expr)                                                                              \
  do {                                                                                               \
    ::tca::Error _e = (expr);                                                                        \
    if (!_e.IsOk()) {                                                                                \
      ++::minitest::Failures();                                                                      \
      std::cout << __FILE__ << ":" << __LINE__ << ": " #expr " failed: " << _e.Message() << std::endl; \
    }                                                                                                \
  } while (0)

#define ASSERT_OK(expr)                                                                              \
  do {                                                                                               \
    ::tca::Error _e = (expr);                                                                        \
    if (!_e.IsOk()) {                                                                                \
      ++::minitest::Failures();                                                                      \
      std::cout << __FILE__ << ":" << __LINE__ << ": " #expr " failed: " << _e.Message() << std::endl; \
      return;                                                                                        \
    }                                                                                                \
  } while (0)

#define EXPECT_ERROR_CONTAINS(expr, text)                                                             \
  do {                                                                                                \
    ::tca::Error _e = (expr);                                                                         \
    if (_e.IsOk() || _e.Message().find(text) == std::string::npos) {                                  \
      ++::minitest::Failures();                                                                       \
      std::cout << __FILE__ << ":" << __LINE__ << ": " #expr " should fail with '" << text << "', got: " \
                << (_e.IsOk() ? std::string("OK") : _e.Message()) << std::endl;                       \
    }                                                                                                 \
  } while (0)
