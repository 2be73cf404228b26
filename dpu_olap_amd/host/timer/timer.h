// timer.h -- named accumulating timers exported as benchmark counters; same surface as the
// reference's timer::Timer / timer::Timers (host/timer/timer.h:13-39): New/Get/get(),
// Start/Stop/Result per "rank" (here: per GPU).  GPU phases are measured by the C ABI with
// HIP events (olap_get_timers) and folded in with Add().
#pragma once

#include <chrono>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace timer {

class Timer {
 public:
  explicit Timer(uint32_t nb_ranks) : total_(nb_ranks ? nb_ranks : 1), begin_(nb_ranks ? nb_ranks : 1) {}
  void Start(uint32_t id = 0) { begin_.at(id) = clock::now(); }
  void Stop(uint32_t id = 0) {
    total_.at(id) += std::chrono::duration_cast<std::chrono::nanoseconds>(clock::now() - begin_.at(id));
  }
  void Add(std::chrono::nanoseconds d, uint32_t id = 0) { total_.at(id) += d; }
  std::chrono::nanoseconds Result() const {
    std::chrono::nanoseconds s{0};
    for (auto d : total_) s += d;
    return s;
  }

 private:
  using clock = std::chrono::high_resolution_clock;
  std::vector<std::chrono::nanoseconds> total_;
  std::vector<clock::time_point> begin_;
};

class Timers {
 public:
  explicit Timers(uint32_t nb_ranks) : nb_ranks_(nb_ranks) {}
  std::shared_ptr<Timer> New(const std::string& name) {
    auto t = std::make_shared<Timer>(nb_ranks_);
    timers_[name] = t;
    return t;
  }
  std::shared_ptr<Timer> Get(const std::string& name) {
    auto it = timers_.find(name);
    return it == timers_.end() ? New(name) : it->second;
  }
  const std::map<std::string, std::shared_ptr<Timer>>& get() const { return timers_; }

 private:
  uint32_t nb_ranks_;
  std::map<std::string, std::shared_ptr<Timer>> timers_;
};

}  // namespace timer
