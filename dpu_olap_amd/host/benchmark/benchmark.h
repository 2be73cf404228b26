// benchmark.h -- a small, source-compatible subset of Google Benchmark (1.7.x), enough for
// the BM_* harness of this project.  Google Benchmark itself is not in the image and
// cannot be fetched (the reference pulls it with CPM, Dependencies.cmake:23-33), so the
// harness ships this runner: same registration macros, same `State` surface the
// reference's BM_* templates use (range(), iterations(), counters, SkipWithError,
// SetItemsProcessed / SetBytesProcessed), same CLI flags the reference's scripts pass
// (--benchmark_filter, --benchmark_out, --benchmark_out_format, --benchmark_repetitions,
// --benchmark_list_tests; README.md:62-65) and the same JSON layout, so GB-JSON tooling in
// the style of scripts/parse_results.py keeps working.
#pragma once

#include <chrono>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace benchmark {

enum TimeUnit { kNanosecond, kMicrosecond, kMillisecond, kSecond };

class Counter {
 public:
  enum Flags { kDefaults = 0, kIsRate = 1, kAvgThreads = 2, kAvgIterations = 4 };
  double value = 0.0;
  int flags = kDefaults;
  Counter(double v = 0.0, int f = kDefaults) : value(v), flags(f) {}
  operator double() const { return value; }
};

using UserCounters = std::map<std::string, Counter>;

class State {
 public:
  struct Iter {
    State* s;
    bool operator!=(const Iter&) const { return s->KeepRunning(); }
    void operator++() {}
    struct Unit {};
    Unit operator*() const { return {}; }
  };
  Iter begin() { return {this}; }
  Iter end() { return {this}; }

  int64_t range(size_t i = 0) const { return i < args_.size() ? args_[i] : 0; }
  int64_t iterations() const { return done_; }
  void SkipWithError(const char* msg);
  bool error_occurred() const { return error_; }
  void SetItemsProcessed(int64_t n) { items_ = n; }
  void SetBytesProcessed(int64_t n) { bytes_ = n; }
  int64_t items_processed() const { return items_; }
  int64_t bytes_processed() const { return bytes_; }
  void PauseTiming();
  void ResumeTiming();
  void SetLabel(const std::string& l) { label_ = l; }
  UserCounters counters;

  // runner side
  State(std::vector<int64_t> args, int64_t max_iters) : args_(std::move(args)), max_(max_iters) {}
  bool KeepRunning();
  void Finish();
  double real_seconds() const { return real_; }
  double cpu_seconds() const { return cpu_; }
  const std::string& error_message() const { return error_msg_; }
  const std::string& label() const { return label_; }

 private:
  std::vector<int64_t> args_;
  int64_t max_ = 1, done_ = 0, items_ = 0, bytes_ = 0;
  bool started_ = false, finished_ = false, error_ = false, paused_ = false;
  std::chrono::steady_clock::time_point t0_;
  double cpu0_ = 0, real_ = 0, cpu_ = 0;
  std::string error_msg_, label_;
};

class Fixture {
 public:
  virtual ~Fixture() = default;
  virtual void SetUp(State&) {}
  virtual void TearDown(State&) {}
  virtual void BenchmarkCase(State&) = 0;
  void Run(State& st) {
    SetUp(st);
    BenchmarkCase(st);
    TearDown(st);
  }
};

namespace internal {

class Benchmark {
 public:
  explicit Benchmark(std::string name, std::function<std::unique_ptr<Fixture>()> make)
      : name_(std::move(name)), make_(std::move(make)) {}
  Benchmark* ArgNames(const std::vector<std::string>& n) { arg_names_ = n; return this; }
  Benchmark* Args(const std::vector<int64_t>& a) { args_.push_back(a); return this; }
  Benchmark* Arg(int64_t a) { args_.push_back({a}); return this; }
  Benchmark* ArgsProduct(const std::vector<std::vector<int64_t>>& lists);
  Benchmark* Iterations(int64_t n) { iterations_ = n; return this; }
  Benchmark* Repetitions(int n) { repetitions_ = n; return this; }
  Benchmark* MinTime(double s) { min_time_ = s; return this; }
  Benchmark* UseRealTime() { use_real_time_ = true; return this; }
  Benchmark* MeasureProcessCPUTime() { process_cpu_ = true; return this; }
  Benchmark* Unit(TimeUnit u) { unit_ = u; return this; }

  std::string name_;
  std::function<std::unique_ptr<Fixture>()> make_;
  std::vector<std::string> arg_names_;
  std::vector<std::vector<int64_t>> args_;
  int64_t iterations_ = 0;
  int repetitions_ = 0;
  double min_time_ = 0.0;
  bool use_real_time_ = false, process_cpu_ = false;
  TimeUnit unit_ = kNanosecond;
  std::unique_ptr<Fixture> shared_fixture_;  // one fixture instance per registration, like GB
};

Benchmark* RegisterBenchmarkInternal(Benchmark* b);

}  // namespace internal

void AddCustomContext(const std::string& key, const std::string& value);
void Initialize(int* argc, char** argv);
bool ReportUnrecognizedArguments(int argc, char** argv);
size_t RunSpecifiedBenchmarks();
void Shutdown();

}  // namespace benchmark

#define BENCHMARK_PRIVATE_CONCAT2(a, b) a##b
#define BENCHMARK_PRIVATE_CONCAT(a, b) BENCHMARK_PRIVATE_CONCAT2(a, b)

// class Fixture_Name_Benchmark : public Fixture { void BenchmarkCase(State&) override; };
#define BENCHMARK_DEFINE_F(BaseClass, Method)                                   \
  class BaseClass##_##Method##_Benchmark : public BaseClass {                   \
   public:                                                                      \
    BaseClass##_##Method##_Benchmark() {}                                       \
                                                                                \
   protected:                                                                   \
    void BenchmarkCase(::benchmark::State&) override;                           \
  };                                                                            \
  void BaseClass##_##Method##_Benchmark::BenchmarkCase

#define BENCHMARK_REGISTER_F(BaseClass, Method)                                              \
  static ::benchmark::internal::Benchmark* BENCHMARK_PRIVATE_CONCAT(benchmark_reg_, __LINE__) \
      __attribute__((unused)) = ::benchmark::internal::RegisterBenchmarkInternal(            \
          new ::benchmark::internal::Benchmark(#BaseClass "/" #Method, [] {                  \
            return std::unique_ptr<::benchmark::Fixture>(new BaseClass##_##Method##_Benchmark()); \
          }))

// BENCHMARK(func): a plain function void func(benchmark::State&), name = "func" (no fixture prefix).
namespace benchmark {
namespace internal {
class FunctionFixture : public Fixture {
 public:
  explicit FunctionFixture(void (*fn)(State&)) : fn_(fn) {}

 protected:
  void BenchmarkCase(State& state) override { fn_(state); }

 private:
  void (*fn_)(State&);
};
}  // namespace internal
}  // namespace benchmark

#define BENCHMARK(func)                                                                        \
  static ::benchmark::internal::Benchmark* BENCHMARK_PRIVATE_CONCAT(benchmark_reg_, __LINE__)   \
      __attribute__((unused)) = ::benchmark::internal::RegisterBenchmarkInternal(              \
          new ::benchmark::internal::Benchmark(#func, [] {                                      \
            return std::unique_ptr<::benchmark::Fixture>(new ::benchmark::internal::FunctionFixture(func)); \
          }))

