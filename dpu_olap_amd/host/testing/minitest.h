// minitest.h -- the handful of GoogleTest macros the operator tests use (TEST, EXPECT_EQ,
// EXPECT_TRUE, EXPECT_LE, EXPECT_GT, GTEST_SKIP).  GoogleTest is not in the image and cannot
// be fetched (the reference pulls it with CPM, Dependencies.cmake:10-21).
#pragma once

#include <functional>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace minitest {

struct TestCase {
  std::string suite, name;
  std::function<void()> body;
};

inline std::vector<TestCase>& registry() {
  static std::vector<TestCase> r;
  return r;
}

struct Current {
  int failures = 0;
  bool skipped = false;
  std::string skip_msg;
};

inline Current& current() {
  static Current c;
  return c;
}

struct Registrar {
  Registrar(const char* suite, const char* name, std::function<void()> body) {
    registry().push_back({suite, name, std::move(body)});
  }
};

struct SkipStream {
  std::ostringstream os;
  template <typename T>
  SkipStream& operator<<(const T& v) {
    os << v;
    return *this;
  }
  ~SkipStream() {
    current().skipped = true;
    current().skip_msg = os.str();
  }
};

inline void report_failure(const char* file, int line, const std::string& what) {
  ++current().failures;
  std::cerr << file << ":" << line << ": Failure\n  " << what << std::endl;
}

// returns the process exit code; argv may carry --gtest_filter=<substring> and --gtest_list_tests
inline int run_all(int argc, char** argv) {
  std::string filter;
  bool list = false;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a.rfind("--gtest_filter=", 0) == 0) filter = a.substr(15);
    if (a == "--gtest_list_tests") list = true;
  }
  int failed = 0, ran = 0, skipped = 0;
  for (auto& t : registry()) {
    const std::string full = t.suite + "." + t.name;
    if (!filter.empty() && filter != "*" && full.find(filter) == std::string::npos) continue;
    if (list) {
      std::cout << full << "\n";
      continue;
    }
    std::cout << "[ RUN      ] " << full << std::endl;
    current() = Current{};
    try {
      t.body();
    } catch (const std::exception& e) {
      report_failure(__FILE__, __LINE__, std::string("uncaught exception: ") + e.what());
    }
    ++ran;
    if (current().skipped) {
      ++skipped;
      std::cout << "[  SKIPPED ] " << full << " (" << current().skip_msg << ")" << std::endl;
    } else if (current().failures) {
      ++failed;
      std::cout << "[  FAILED  ] " << full << std::endl;
    } else {
      std::cout << "[       OK ] " << full << std::endl;
    }
  }
  if (!list) std::cout << "[==========] " << ran << " tests ran, " << failed << " failed, " << skipped << " skipped" << std::endl;
  return failed ? 1 : 0;
}

}  // namespace minitest

#define TEST(suite, name)                                                                   \
  static void suite##_##name##_body();                                                      \
  static ::minitest::Registrar suite##_##name##_reg(#suite, #name, suite##_##name##_body);  \
  static void suite##_##name##_body()

#define MINITEST_CMP(a, b, op, opname)                                                       \
  do {                                                                                       \
    auto&& va__ = (a);                                                                       \
    auto&& vb__ = (b);                                                                       \
    if (!(va__ op vb__)) {                                                                   \
      std::ostringstream os__;                                                               \
      os__ << "Expected: (" #a ") " opname " (" #b "), actual: " << va__ << " vs " << vb__;  \
      ::minitest::report_failure(__FILE__, __LINE__, os__.str());                            \
    }                                                                                        \
  } while (0)

#define EXPECT_EQ(a, b) MINITEST_CMP(a, b, ==, "==")
#define EXPECT_NE(a, b) MINITEST_CMP(a, b, !=, "!=")
#define EXPECT_LE(a, b) MINITEST_CMP(a, b, <=, "<=")
#define EXPECT_LT(a, b) MINITEST_CMP(a, b, <, "<")
#define EXPECT_GE(a, b) MINITEST_CMP(a, b, >=, ">=")
#define EXPECT_GT(a, b) MINITEST_CMP(a, b, >, ">")
#define EXPECT_TRUE(c)                                                                     \
  do {                                                                                     \
    if (!(c)) ::minitest::report_failure(__FILE__, __LINE__, "Value of: " #c "\n  Expected: true"); \
  } while (0)
#define EXPECT_FALSE(c) EXPECT_TRUE(!(c))
#define GTEST_SKIP() return (void)(::minitest::SkipStream())
