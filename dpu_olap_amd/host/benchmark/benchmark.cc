// benchmark.cc -- runner of the Google-Benchmark-compatible subset in benchmark.h.
#include "benchmark/benchmark.h"

#include <sys/resource.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <regex>
#include <sstream>
#include <thread>

namespace benchmark {
namespace {

struct Flags {
  std::string filter = ".";
  std::string out, out_format = "json", format = "console";
  int repetitions = 1;
  double min_time = 0.5;
  bool list_tests = false;
};

Flags& flags() {
  static Flags f;
  return f;
}

std::vector<std::unique_ptr<internal::Benchmark>>& registry() {
  static std::vector<std::unique_ptr<internal::Benchmark>> r;
  return r;
}

std::vector<std::pair<std::string, std::string>>& custom_context() {
  static std::vector<std::pair<std::string, std::string>> c;
  return c;
}

std::vector<std::string>& unrecognized() {
  static std::vector<std::string> u;
  return u;
}

double process_cpu_seconds() {
  rusage ru;
  getrusage(RUSAGE_SELF, &ru);
  return ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6 + ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6;
}

double unit_multiplier(TimeUnit u) {
  switch (u) {
    case kSecond: return 1.0;
    case kMillisecond: return 1e3;
    case kMicrosecond: return 1e6;
    default: return 1e9;
  }
}

const char* unit_name(TimeUnit u) {
  switch (u) {
    case kSecond: return "s";
    case kMillisecond: return "ms";
    case kMicrosecond: return "us";
    default: return "ns";
  }
}

std::string json_escape(const std::string& s) {
  std::string o;
  for (char c : s) {
    if (c == '"' || c == '\\') { o += '\\'; o += c; }
    else if (c == '\n') o += "\\n";
    else o += c;
  }
  return o;
}

std::string fmt_double(double v) {
  char buf[64];
  std::snprintf(buf, sizeof buf, "%.17g", v);
  return buf;
}

struct RunResult {
  std::string name, run_name, error_message, label;
  bool error = false;
  int repetitions = 1, repetition_index = 0;
  int64_t iterations = 0;
  double real_time = 0, cpu_time = 0;  // per iteration, in `unit`
  TimeUnit unit = kNanosecond;
  double items_per_second = -1, bytes_per_second = -1;
  std::vector<std::pair<std::string, double>> counters;
};

}  // namespace

// ------------------------------------------------------------------ State
void State::SkipWithError(const char* msg) {
  error_ = true;
  error_msg_ = msg ? msg : "";
  Finish();
}

void State::PauseTiming() {
  if (started_ && !finished_ && !paused_) {
    real_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
    cpu_ += process_cpu_seconds() - cpu0_;
    paused_ = true;
  }
}

void State::ResumeTiming() {
  if (paused_) {
    t0_ = std::chrono::steady_clock::now();
    cpu0_ = process_cpu_seconds();
    paused_ = false;
  }
}

bool State::KeepRunning() {
  if (error_ || finished_) return false;
  if (!started_) {
    started_ = true;
    t0_ = std::chrono::steady_clock::now();
    cpu0_ = process_cpu_seconds();
    if (max_ <= 0) { Finish(); return false; }
    return true;
  }
  ++done_;
  if (done_ >= max_) {
    Finish();
    return false;
  }
  return true;
}

void State::Finish() {
  if (finished_) return;
  if (started_ && !paused_) {
    real_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count();
    cpu_ += process_cpu_seconds() - cpu0_;
  }
  finished_ = true;
}

// -------------------------------------------------------------- registration
namespace internal {

Benchmark* Benchmark::ArgsProduct(const std::vector<std::vector<int64_t>>& lists) {
  std::vector<std::vector<int64_t>> acc{{}};
  for (const auto& l : lists) {
    std::vector<std::vector<int64_t>> next;
    for (const auto& a : acc)
      for (int64_t v : l) {
        auto b = a;
        b.push_back(v);
        next.push_back(b);
      }
    acc.swap(next);
  }
  // Google Benchmark enumerates the first list fastest
  std::sort(acc.begin(), acc.end(), [](const auto& x, const auto& y) {
    for (size_t i = x.size(); i-- > 0;)
      if (x[i] != y[i]) return x[i] < y[i];
    return false;
  });
  for (auto& a : acc) args_.push_back(a);
  return this;
}

Benchmark* RegisterBenchmarkInternal(Benchmark* b) {
  registry().emplace_back(b);
  return b;
}

}  // namespace internal

void AddCustomContext(const std::string& key, const std::string& value) { custom_context().emplace_back(key, value); }

void Initialize(int* argc, char** argv) {
  Flags& f = flags();
  for (int i = 1; i < *argc; ++i) {
    std::string a = argv[i];
    auto val = [&](const char* name, std::string* out) {
      std::string p = std::string("--") + name + "=";
      if (a.rfind(p, 0) == 0) { *out = a.substr(p.size()); return true; }
      return false;
    };
    std::string v;
    if (val("benchmark_filter", &v)) f.filter = v;
    else if (val("benchmark_out", &v)) f.out = v;
    else if (val("benchmark_out_format", &v)) f.out_format = v;
    else if (val("benchmark_format", &v)) f.format = v;
    else if (val("benchmark_repetitions", &v)) f.repetitions = std::max(1, std::atoi(v.c_str()));
    else if (val("benchmark_min_time", &v)) f.min_time = std::atof(v.c_str());
    else if (a == "--benchmark_list_tests" || a == "--benchmark_list_tests=true") f.list_tests = true;
    else if (val("benchmark_counters_tabular", &v) || val("benchmark_color", &v)) {}
    else unrecognized().push_back(a);
  }
}

bool ReportUnrecognizedArguments(int, char**) {
  for (const auto& a : unrecognized()) std::cerr << "error: unrecognized command-line flag: " << a << "\n";
  return !unrecognized().empty();
}

static std::string instance_name(const internal::Benchmark& b, const std::vector<int64_t>& args) {
  std::ostringstream os;
  os << b.name_;
  for (size_t i = 0; i < args.size(); ++i) {
    os << "/";
    if (i < b.arg_names_.size() && !b.arg_names_[i].empty()) os << b.arg_names_[i] << ":";
    os << args[i];
  }
  if (b.iterations_) os << "/iterations:" << b.iterations_;
  if (b.process_cpu_) os << "/process_time";
  if (b.use_real_time_) os << "/real_time";
  return os.str();
}

static RunResult run_once(internal::Benchmark& b, const std::vector<int64_t>& args, const std::string& name, int rep, int reps) {
  RunResult r;
  r.name = r.run_name = name;
  r.repetitions = reps;
  r.repetition_index = rep;
  r.unit = b.unit_;
  if (!b.shared_fixture_) b.shared_fixture_ = b.make_();
  int64_t iters = b.iterations_ ? b.iterations_ : 1;
  const double min_time = b.min_time_ > 0 ? b.min_time_ : flags().min_time;
  for (;;) {
    State st(args, iters);
    b.shared_fixture_->Run(st);
    st.Finish();
    const double t = b.use_real_time_ ? st.real_seconds() : st.cpu_seconds();
    if (!st.error_occurred() && !b.iterations_ && t < min_time && iters < (int64_t)1e9) {
      // grow like Google Benchmark: aim past min_time with a 1.4x margin, at most x10 per round
      double mult = t > 1e-9 ? std::min(10.0, min_time * 1.4 / t) : 10.0;
      int64_t next = std::max<int64_t>(iters + 1, (int64_t)std::ceil(iters * std::max(mult, 1.5)));
      iters = next;
      continue;
    }
    r.error = st.error_occurred();
    r.error_message = st.error_message();
    r.label = st.label();
    r.iterations = st.iterations();
    const double n = std::max<int64_t>(st.iterations(), 1);
    r.real_time = st.real_seconds() / n * unit_multiplier(b.unit_);
    r.cpu_time = st.cpu_seconds() / n * unit_multiplier(b.unit_);
    if (st.items_processed() > 0 && t > 0) r.items_per_second = st.items_processed() / t;
    if (st.bytes_processed() > 0 && t > 0) r.bytes_per_second = st.bytes_processed() / t;
    for (const auto& kv : st.counters) {
      double v = kv.second.value;
      if (kv.second.flags & Counter::kAvgIterations) v /= n;
      if (kv.second.flags & Counter::kIsRate) v = t > 0 ? v / t : 0;
      r.counters.emplace_back(kv.first, v);
    }
    return r;
  }
}

static void print_console(const RunResult& r) {
  if (r.error) {
    std::printf("%-90s ERROR OCCURRED: '%s'\n", r.name.c_str(), r.error_message.c_str());
    return;
  }
  std::printf("%-90s %12.3f %s %12.3f %s %10lld", r.name.c_str(), r.real_time, unit_name(r.unit), r.cpu_time,
              unit_name(r.unit), (long long)r.iterations);
  if (r.bytes_per_second >= 0) std::printf(" bytes_per_second=%.4gG/s", r.bytes_per_second / (1024.0 * 1024 * 1024));
  if (r.items_per_second >= 0) std::printf(" items_per_second=%.4gM/s", r.items_per_second / 1e6);
  for (const auto& c : r.counters) std::printf(" %s=%.6g", c.first.c_str(), c.second);
  std::printf("\n");
  std::fflush(stdout);
}

static std::string to_json(const std::vector<RunResult>& results, const char* argv0) {
  std::ostringstream os;
  char host[256] = "unknown";
  gethostname(host, sizeof host);
  std::time_t now = std::time(nullptr);
  char date[64];
  std::strftime(date, sizeof date, "%Y-%m-%dT%H:%M:%S%z", std::localtime(&now));
  os << "{\n  \"context\": {\n";
  os << "    \"date\": \"" << date << "\",\n    \"host_name\": \"" << json_escape(host) << "\",\n";
  os << "    \"executable\": \"" << json_escape(argv0 ? argv0 : "") << "\",\n";
  os << "    \"num_cpus\": " << std::thread::hardware_concurrency() << ",\n";
  os << "    \"library_build_type\": \"release\",\n    \"library\": \"dpu_olap_amd mini-benchmark (Google Benchmark 1.7 compatible JSON)\"";
  for (const auto& kv : custom_context()) os << ",\n    \"" << json_escape(kv.first) << "\": \"" << json_escape(kv.second) << "\"";
  os << "\n  },\n  \"benchmarks\": [\n";
  for (size_t i = 0; i < results.size(); ++i) {
    const RunResult& r = results[i];
    os << "    {\n      \"name\": \"" << json_escape(r.name) << "\",\n      \"run_name\": \"" << json_escape(r.run_name)
       << "\",\n      \"run_type\": \"iteration\",\n      \"repetitions\": " << r.repetitions
       << ",\n      \"repetition_index\": " << r.repetition_index << ",\n      \"threads\": 1,\n";
    if (r.error) {
      os << "      \"error_occurred\": true,\n      \"error_message\": \"" << json_escape(r.error_message) << "\",\n";
    }
    os << "      \"iterations\": " << r.iterations << ",\n      \"real_time\": " << fmt_double(r.real_time)
       << ",\n      \"cpu_time\": " << fmt_double(r.cpu_time) << ",\n      \"time_unit\": \"" << unit_name(r.unit) << "\"";
    if (r.bytes_per_second >= 0) os << ",\n      \"bytes_per_second\": " << fmt_double(r.bytes_per_second);
    if (r.items_per_second >= 0) os << ",\n      \"items_per_second\": " << fmt_double(r.items_per_second);
    if (!r.label.empty()) os << ",\n      \"label\": \"" << json_escape(r.label) << "\"";
    for (const auto& c : r.counters) os << ",\n      \"" << json_escape(c.first) << "\": " << fmt_double(c.second);
    os << "\n    }" << (i + 1 < results.size() ? "," : "") << "\n";
  }
  os << "  ]\n}\n";
  return os.str();
}

static const char* g_argv0 = nullptr;

size_t RunSpecifiedBenchmarks() {
  const Flags& f = flags();
  std::regex re;
  bool negate = false;
  std::string pat = f.filter;
  if (!pat.empty() && pat[0] == '-') { negate = true; pat = pat.substr(1); }
  if (pat == "all" || pat.empty()) pat = ".";
  re = std::regex(pat);
  std::vector<RunResult> results;
  size_t count = 0;
  if (!f.list_tests && f.format == "console") {
    for (const auto& kv : custom_context()) std::printf("%s: %s\n", kv.first.c_str(), kv.second.c_str());
    std::printf("%-90s %15s %15s %10s\n", "Benchmark", "Time", "CPU", "Iterations");
  }
  for (auto& b : registry()) {
    auto arglists = b->args_;
    if (arglists.empty()) arglists.push_back({});
    for (const auto& args : arglists) {
      const std::string name = instance_name(*b, args);
      const bool match = std::regex_search(name, re);
      if (match == negate) continue;
      ++count;
      if (f.list_tests) {
        std::printf("%s\n", name.c_str());
        continue;
      }
      const int reps = b->repetitions_ > 0 ? b->repetitions_ : f.repetitions;
      for (int rep = 0; rep < reps; ++rep) {
        RunResult r = run_once(*b, args, name, rep, reps);
        if (f.format == "console") print_console(r);
        results.push_back(std::move(r));
      }
    }
    b->shared_fixture_.reset();
  }
  if (f.list_tests) return count;
  const std::string js = to_json(results, g_argv0);
  if (f.format == "json") std::cout << js;
  if (!f.out.empty()) {
    std::ofstream ofs(f.out);
    ofs << js;  // only the JSON format is implemented (the reference's scripts use json)
  }
  return count;
}

void Shutdown() {}

}  // namespace benchmark
